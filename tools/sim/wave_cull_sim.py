#!/usr/bin/env python3
"""CPU simulation (numpy, statistics only): how many leaves does a wave of 64 rays marching in step have to evaluate per
iteration under different far tests?  A wave = the 64 pixels of an 8x8 tile at one AA sample (what the march kernel's ray pool
hands a wave), all rays advanced together, rays that end leave their lane idle.
  exact      a leaf is evaluated iff v_leaf <= thr for ANY live lane (per-lane, per-leaf test; the ideal of the threshold rule)
  pairs      the shipped tests: bounding sphere per pair of consecutive leaves, then a test per box
  ball       one test per leaf for the whole wave: |p* - c| - R - rho > max thr, p* a live lane's position, rho the wave's radius
usage: wave_cull_sim.py [scene] [W H] [n_tiles]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
from oracle import cbind  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "g32"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
NT = int(sys.argv[4]) if len(sys.argv) > 4 else 300
MAX_ITER, MIN_D, MAX_D = 256, 0.01, 100.0
nodes, root = getattr(scenes, scene)()


def flatten(nodes, root):
    """left-deep chain -> [(kind, params, op)] in evaluation order (op of leaf 0 = None)"""
    out = []

    def walk(i):
        kind, p, l, r = nodes[i]
        if kind in (scenes.SPHERE, scenes.BOX):
            return [(kind, p, None)]
        left = walk(l)
        right = walk(r)
        assert len(right) == 1, "chains only"
        return left + [(right[0][0], right[0][1], kind if kind != scenes.SMOOTH_UNION else ("m", p[0]))]
    return walk(root)


chain = flatten(nodes, root)
N = len(chain)
C = np.array([c[1][:3] for c in chain])
IS_S = np.array([c[0] == scenes.SPHERE for c in chain])
R_s = np.array([c[1][3] if c[0] == scenes.SPHERE else 0.0 for c in chain])
Hb = np.array([c[1][3:6] if c[0] == scenes.BOX else [0, 0, 0] for c in chain])
RB = np.where(IS_S, R_s, np.linalg.norm(Hb, axis=1))       # bounding radius


def leaf_values(p):     # p: (n,3) -> (n,N)
    d = p[:, None, :] - C[None]
    vs = np.linalg.norm(d, axis=2) - R_s[None]
    q = np.abs(d) - Hb[None]
    vb = np.linalg.norm(np.maximum(q, 0), axis=2) + np.minimum(q.max(axis=2), 0)
    return np.where(IS_S[None], vs, vb)


def fold(v):
    acc = v[:, 0].copy()
    for i in range(1, N):
        op = chain[i][2]
        if op == scenes.UNION:
            acc = np.minimum(acc, v[:, i])
        elif op == scenes.SUBTRACTION:
            acc = np.maximum(acc, -v[:, i])
        else:
            k = op[1]
            h = np.maximum(k - np.abs(acc - v[:, i]), 0) / k
            acc = np.minimum(acc, v[:, i]) - h * h * k * 0.25
    return acc


u, pos, q, orb = cbind.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
inv_proj = np.array(list(u.inv_proj)).reshape(4, 4).T
inv_view = np.array(list(u.inv_view)).reshape(4, 4).T
ro = (inv_view @ np.array([0, 0, 0, 1.0]))[:3]


def rays(px, py, s):
    i, j = s // 4, s % 4
    ox = ((i + 0.5) / 4 - 0.5) / W * 2.0
    oy = ((j + 0.5) / 4 - 0.5) / H * 2.0
    x = (px + 0.5) / W * 2 - 1 + ox
    y = 1 - (py + 0.5) / H * 2 + oy
    ndc = np.stack([x, y, -np.ones_like(x), np.ones_like(x)], axis=1)
    wv = (inv_view @ (inv_proj @ ndc.T)).T
    d = wv - np.concatenate([ro, [1.0]])[None]
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return d[:, :3]


pairs = [(2 * g, 2 * g + 1) for g in range(N // 2)]
PC = np.array([(C[a] + C[b]) / 2 for a, b in pairs])
PR = np.array([max(np.linalg.norm(C[a] - PC[g]) + RB[a], np.linalg.norm(C[b] - PC[g]) + RB[b]) for g, (a, b) in enumerate(pairs)])

rng = np.random.default_rng(1)
tiles_x, tiles_y = W // 8, H // 8
stat = {"iters": 0, "live": 0, "exact": 0, "pairs": 0, "ball": 0, "ball2": 0, "lane": 0}
done_tiles = 0
attempts = 0
while done_tiles < NT and attempts < 40 * NT:
    attempts += 1
    tx, ty = int(rng.integers(tiles_x)), int(rng.integers(tiles_y))
    s = int(rng.integers(16))
    px = (tx * 8 + np.arange(64) % 8).astype(float)
    py = (ty * 8 + np.arange(64) // 8).astype(float)
    d = rays(px, py, s)
    # keep rays that hit (the marched population is ~92 % hits); march all first to classify
    sc = np.zeros(64)
    live = np.ones(64, bool)
    hit = np.zeros(64, bool)
    hist = []
    thr = np.full(64, np.inf)
    for it in range(MAX_ITER):
        if not live.any():
            break
        p = ro[None] + d * sc[:, None]
        v = leaf_values(p)
        F = fold(v)
        hist.append((live.copy(), p.copy(), v.copy(), thr.copy()))
        h = live & (F < MIN_D)
        hit |= h
        esc = live & ~h & (F > MAX_D)
        go = live & ~h & ~esc
        sc = np.where(go, sc + F, sc)
        thr = np.where(go, 2 * np.abs(F), thr)
        live = go
    if hit.sum() < 8:
        continue
    done_tiles += 1
    keep = hit            # lanes of rays that are marched at all
    for live, p, v, thr in hist:
        l = live & keep
        if not l.any():
            continue
        stat["iters"] += 1
        stat["live"] += l.sum()
        near = (v[l] <= thr[l][:, None])                 # (n_live, N)
        stat["lane"] += near.sum() / l.sum()
        stat["exact"] += near.any(axis=0).sum()
        # shipped: pair spheres then boxes individually
        dp = np.linalg.norm(p[l][:, None, :] - PC[None], axis=2) - PR[None]
        pnear = (dp <= thr[l][:, None]).any(axis=0)
        cnt = 0
        for g, (a, b) in enumerate(pairs):
            if pnear[g]:
                for m in (a, b):
                    if IS_S[m]:
                        cnt += 1
                    else:
                        cnt += int(near[:, m].any())
        if N % 2:
            cnt += int(near[:, N - 1].any())
        stat["pairs"] += cnt
        # ball: reference = first live lane
        pl = p[l]
        pstar = pl[0]
        rho = np.linalg.norm(pl - pstar[None], axis=1).max()
        T = thr[l].max()
        lb = np.linalg.norm(C - pstar[None], axis=1) - RB - rho
        stat["ball"] += (lb <= T).sum()
        # ball around the centroid (smaller rho)
        pc = pl.mean(axis=0)
        rho2 = np.linalg.norm(pl - pc[None], axis=1).max()
        lb2 = np.linalg.norm(C - pc[None], axis=1) - RB - rho2
        stat["ball2"] += (lb2 <= T).sum()
it = stat["iters"]
print("%s %dx%d: %d tiles, %d wave-iterations, lane occupancy %.2f" % (scene, W, H, done_tiles, it, stat["live"] / (64.0 * it)))
print("leaves of %d evaluated per wave-iteration:  per-lane average %.2f | exact union %.2f | shipped pair+box tests %.2f | ball (a lane's position) %.2f | ball (centroid) %.2f"
      % (N, stat["lane"] / it, stat["exact"] / it, stat["pairs"] / it, stat["ball"] / it, stat["ball2"] / it))

# ---- programs that blend: the local rule, per lane and through the wave's ball -----------------------------------------
if any(isinstance(c[2], tuple) for c in chain):
    RIN = np.where(IS_S, R_s, Hb.min(axis=1))
    KS = np.array([c[2][1] if isinstance(c[2], tuple) else 0.0 for c in chain])
    KMAX = KS.max()
    ISSUB = np.array([c[2] == scenes.SUBTRACTION for c in chain])
    st2 = {"iters": 0, "lane_local": 0, "union_local": 0, "ball_seq": 0, "ball_par": 0, "lane_restart": 0, "union_restart": 0}
    rng = np.random.default_rng(1)
    done = 0
    while done < NT:
        tx, ty = int(rng.integers(tiles_x)), int(rng.integers(tiles_y))
        s = int(rng.integers(16))
        px = (tx * 8 + np.arange(64) % 8).astype(float)
        py = (ty * 8 + np.arange(64) // 8).astype(float)
        d = rays(px, py, s)
        sc = np.zeros(64); live = np.ones(64, bool); hit = np.zeros(64, bool); hist = []
        for it in range(MAX_ITER):
            if not live.any():
                break
            p = ro[None] + d * sc[:, None]
            v = leaf_values(p)
            # per-lane fold with per-unit accumulators
            accs = np.zeros((64, N)); acc = v[:, 0].copy(); accs[:, 0] = acc
            for i in range(1, N):
                accs[:, i] = acc            # accumulator the unit meets
                op = chain[i][2]
                if op == scenes.UNION: acc = np.minimum(acc, v[:, i])
                elif op == scenes.SUBTRACTION: acc = np.maximum(acc, -v[:, i])
                else:
                    k = op[1]; h = np.maximum(k - np.abs(acc - v[:, i]), 0) / k
                    acc = np.minimum(acc, v[:, i]) - h * h * k * 0.25
            F = acc
            hist.append((live.copy(), p.copy(), v.copy(), accs.copy()))
            h_ = live & (F < MIN_D); hit |= h_
            esc = live & ~h_ & (F > MAX_D); go = live & ~h_ & ~esc
            sc = np.where(go, sc + F, sc); live = go
        if hit.sum() < 8:
            continue
        done += 1
        for live, p, v, accs in hist:
            l = live & hit
            if not l.any():
                continue
            st2["iters"] += 1
            vl, al = v[l], accs[l]
            # local rule per lane: unit needed unless v >= acc + k (U/M) or v + acc >= 0 (SUB); unit 0 always
            need = np.ones_like(vl, bool)
            need[:, 1:] = np.where(ISSUB[None, 1:], vl[:, 1:] + al[:, 1:] < 0, vl[:, 1:] < al[:, 1:] + KS[None, 1:])
            st2["lane_local"] += need.sum() / l.sum()
            st2["union_local"] += need.any(axis=0).sum()
            # restart per lane: last U/M unit with v <= acc - k; units before it are dead
            rs = (~ISSUB[None, :]) & (vl <= al - KS[None, :]); rs[:, 0] = False
            jstar = np.where(rs.any(axis=1), N - 1 - np.argmax(rs[:, ::-1], axis=1), 0)
            need_r = need & (np.arange(N)[None, :] >= jstar[:, None])
            st2["lane_restart"] += need_r.sum() / l.sum()
            st2["union_restart"] += need_r.any(axis=0).sum()
            # the wave's ball
            pl = p[l]; pc = pl[0]
            rho = np.linalg.norm(pl - pc[None], axis=1).max()
            D = np.linalg.norm(C - pc[None], axis=1)
            L = D - RB - rho; Hh = D + rho - RIN
            # sequential bounds
            needs = [True]; alo, ahi = L[0], Hh[0]
            for uix in range(1, N):
                if ISSUB[uix]:
                    if L[uix] + alo >= 0: needs.append(False)
                    else:
                        needs.append(True); ahi = max(ahi, -L[uix]); alo = max(alo, -Hh[uix])
                else:
                    k = KS[uix]
                    if L[uix] >= ahi + k: needs.append(False)
                    elif Hh[uix] <= alo - k:
                        needs = [False] * len(needs) + [True]; alo, ahi = L[uix], Hh[uix]
                    else:
                        needs.append(True); ahi = min(ahi, Hh[uix]); alo = min(alo, L[uix]) - k / 4
            st2["ball_seq"] += sum(needs)
            # parallel: prefix minima, chain lemma for the lower bound, subtractions assumed skippable (checked)
            Hm = np.where(ISSUB, np.inf, Hh); Lm = np.where(ISSUB, np.inf, L)
            ahi_p = np.concatenate([[np.inf], np.minimum.accumulate(Hm)[:-1]])
            alo_p = np.concatenate([[np.inf], np.minimum.accumulate(Lm)[:-1]]) - KMAX
            skip = np.where(ISSUB, L + alo_p >= 0, L >= ahi_p + KS); skip[0] = False
            rst = (~ISSUB) & (Hh <= alo_p - KS); rst[0] = False
            r = (N - 1 - np.argmax(rst[::-1])) if rst.any() else 0
            needp = ~skip; needp[:r] = False
            bad_sub = np.where(ISSUB & ~skip & (np.arange(N) >= r))[0]
            if len(bad_sub):
                needp[bad_sub[0]:] = True
            st2["ball_par"] += needp.sum()
    it = st2["iters"]
    print("blend, units of %d needed per wave-iteration: local rule per lane %.2f, union over the wave %.2f | with restarts per lane %.2f, union %.2f | ball, sequential bounds %.2f | ball, parallel bounds %.2f"
          % (N, st2["lane_local"] / it, st2["union_local"] / it, st2["lane_restart"] / it, st2["union_restart"] / it, st2["ball_seq"] / it, st2["ball_par"] / it))
