"""rm_oracle_np.py -- second, independently written CPU restatement (numpy, float32).

TEST INFRASTRUCTURE (same rules as rm_oracle.c): imported only from tests/.  It exists to
cross-check the C oracle: the two were written separately from the reference text and must
agree bit for bit.  "parity unpinned" against the real wgpu render (see rm_oracle.c header).

Vectorised over rays: every numpy ufunc on float32 arrays is one IEEE-754 binary32 operation
per element (no contraction), np.sqrt and / are correctly rounded, np.rint is ties-to-even.

Reference text followed: src/ray_marching/ray_marching.wgsl (fs_main :36-76, ray_march :87-131,
calculate_normal :135-144, map_scene :187-203, eval_cmd* :205-252).
"""
import numpy as np

F = np.float32
CMD_SPHERE, CMD_BOX, CMD_UNION, CMD_SUBTRACTION = 0, 1, 100, 101
CMD_TRANSLATION_PUSH, CMD_TRANSLATION_POP, CMD_ROTATION_PUSH, CMD_ROTATION_POP, CMD_SCALE_PUSH, CMD_SCALE_POP = range(200, 206)
CMD_PLANE, CMD_CYLINDER, CMD_INTERSECTION, CMD_SMOOTH_UNION = 2, 10, 102, 110   # extensions (not in the reference)
CMD_MATERIAL = 300          # extension: tags the value on top of the stack with a material index (rm_oracle.c)


def _f(x):
    return np.asarray(x, dtype=F)


def _sign(a):
    return np.signbit(a)


def fmin(a, b):
    """minimumNumber: NaN loses, -0 < +0."""
    a, b = np.broadcast_arrays(_f(a), _f(b))
    with np.errstate(invalid="ignore"):
        r = np.where(a < b, a, b)
        r = np.where(a == b, np.where(_sign(a), a, b), r)
    r = np.where(np.isnan(b), a, r)
    r = np.where(np.isnan(a), b, r)
    return r.astype(F)


def fmax(a, b):
    a, b = np.broadcast_arrays(_f(a), _f(b))
    with np.errstate(invalid="ignore"):
        r = np.where(a > b, a, b)
        r = np.where(a == b, np.where(_sign(a), b, a), r)
    r = np.where(np.isnan(b), a, r)
    r = np.where(np.isnan(a), b, r)
    return r.astype(F)


def f2i(x):
    """WGSL i32(f32): truncate, clamp, NaN -> 0."""
    x = np.asarray(x, dtype=np.float64)          # f32 -> f64 is exact
    t = np.trunc(np.nan_to_num(x, nan=0.0, posinf=2.0**31, neginf=-(2.0**31)))
    return np.clip(t, -(2.0**31), 2.0**31 - 1).astype(np.int64).astype(np.int32)


def decode_words(words):
    return np.asarray(words, dtype=np.uint32)


def _wf(words, i):
    return words[i:i + 1].view(F)[0]


def map_scene(cmd_count, words, max_dist, px, py, pz, want_material=False):
    """wgsl:187-203 for arrays of positions.  want_material (extension): returns (distance, material index) -- every
    value carries the index of the operand that decided it; primitives carry 0, Material(i) overwrites the top."""
    if cmd_count == 0:
        d0 = np.full(px.shape, F(max_dist), dtype=F)
        return (d0, np.zeros(px.shape, dtype=np.uint32)) if want_material else d0
    words = decode_words(words)
    stack = []
    mats = []    # parallel to stack: uint32 arrays
    saved = []   # extension: (position, scale) saved by the transform pushes (opcodes 200-205)
    ptr = 0
    for _ in range(cmd_count):
        op = int(words[ptr]); ptr += 1
        if op == CMD_TRANSLATION_PUSH:
            tx, ty, tz = (_wf(words, ptr + k) for k in range(3)); ptr += 3
            saved.append((px, py, pz, None))
            px, py, pz = px - tx, py - ty, pz - tz
            continue
        if op == CMD_ROTATION_PUSH:
            qw, ax, ay, az = (_wf(words, ptr + k) for k in range(4)); ptr += 4
            saved.append((px, py, pz, None))
            cx, cy, cz = py * az - pz * ay, pz * ax - px * az, px * ay - py * ax
            tx, ty, tz = F(2) * cx, F(2) * cy, F(2) * cz
            ux, uy, uz = ty * az - tz * ay, tz * ax - tx * az, tx * ay - ty * ax
            px, py, pz = (px + qw * tx) + ux, (py + qw * ty) + uy, (pz + qw * tz) + uz
            continue
        if op == CMD_SCALE_PUSH:
            sf = _wf(words, ptr); ptr += 1
            saved.append((px, py, pz, sf))
            with np.errstate(divide="ignore", invalid="ignore"):
                px, py, pz = px / sf, py / sf, pz / sf
            continue
        if op == CMD_MATERIAL:
            mats[-1] = np.full(px.shape, int(words[ptr]), dtype=np.uint32); ptr += 1
            continue
        if op in (CMD_TRANSLATION_POP, CMD_ROTATION_POP, CMD_SCALE_POP):
            px, py, pz, sf = saved.pop()
            if op == CMD_SCALE_POP:
                stack[-1] = (stack[-1] * sf).astype(F)
            continue
        mat = np.zeros(px.shape, dtype=np.uint32)
        if op == CMD_SPHERE:
            cx, cy, cz, r = (_wf(words, ptr + k) for k in range(4)); ptr += 4
            dx, dy, dz = px - cx, py - cy, pz - cz
            val = np.sqrt((dx * dx + dy * dy) + dz * dz) - r
        elif op == CMD_BOX:
            cx, cy, cz, rx, ry, rz = (_wf(words, ptr + k) for k in range(6)); ptr += 6
            qx, qy, qz = np.abs(px - cx) - rx, np.abs(py - cy) - ry, np.abs(pz - cz) - rz
            mx, my, mz = fmax(qx, F(0)), fmax(qy, F(0)), fmax(qz, F(0))
            outside = np.sqrt((mx * mx + my * my) + mz * mz)
            inside = fmin(fmax(qx, fmax(qy, qz)), F(0))
            val = outside + inside
        elif op == CMD_PLANE:
            nx, ny, nz, h = (_wf(words, ptr + k) for k in range(4)); ptr += 4
            val = ((px * nx + py * ny) + pz * nz) + h
        elif op == CMD_CYLINDER:
            cx, cy, cz, r, hh = (_wf(words, ptr + k) for k in range(5)); ptr += 5
            dx, dz = px - cx, pz - cz
            qx = np.sqrt(dx * dx + dz * dz) - r
            qy = np.abs(py - cy) - hh
            mx, my = fmax(qx, F(0)), fmax(qy, F(0))
            val = fmin(fmax(qx, qy), F(0)) + np.sqrt(mx * mx + my * my)
        elif op == CMD_INTERSECTION:
            b = stack.pop(); a = stack.pop(); mb = mats.pop(); ma = mats.pop()
            val = fmax(a, b)
            with np.errstate(invalid="ignore"):
                mat = np.where(b > a, mb, ma)
        elif op == CMD_SMOOTH_UNION:
            k = _wf(words, ptr); ptr += 1
            b = stack.pop(); a = stack.pop(); mb = mats.pop(); ma = mats.pop()
            with np.errstate(invalid="ignore"):
                mat = np.where(b < a, mb, ma)
            val = fmin(a, b)
            if k > 0:
                with np.errstate(invalid="ignore"):
                    h = fmax(k - np.abs(a - b), F(0)) / k
                val = val - ((h * h) * k) * F(0.25)
        elif op == CMD_UNION:
            b = stack.pop(); a = stack.pop(); mb = mats.pop(); ma = mats.pop()
            val = fmin(a, b)
            with np.errstate(invalid="ignore"):
                mat = np.where(b < a, mb, ma)
        elif op == CMD_SUBTRACTION:
            b = stack.pop(); a = stack.pop(); mb = mats.pop(); ma = mats.pop()
            val = fmax(a, -b)
            with np.errstate(invalid="ignore"):
                mat = np.where(-b > a, mb, ma)
        else:
            val = np.zeros(px.shape, dtype=F)
        stack.append(val.astype(F))
        mats.append(mat.astype(np.uint32))
    return (stack.pop(), mats.pop()) if want_material else stack.pop()


def _normalize3(x, y, z):
    with np.errstate(invalid="ignore", divide="ignore"):
        l = np.sqrt((x * x + y * y) + z * z)
        return x / l, y / l, z / l


def ray_march(cmd_count, words, limits, ox, oy, oz, dx, dy, dz, materials=None):
    """wgsl:87-131 for arrays of rays.  Returns rgb arrays (linear).  materials (extension): (n, 3) albedo table."""
    min_dist, max_dist, max_iter = F(limits[0]), F(limits[1]), int(limits[2])
    n = dx.shape[0]
    col = np.zeros((3, n), dtype=F)
    dist = np.zeros(n, dtype=F)
    alive = np.arange(n)           # indices still marching
    hit_idx, hit_pos = [], []
    for _ in range(max_iter):
        if alive.size == 0:
            break
        d = dist[alive]
        px = ox[alive] + dx[alive] * d
        py = oy[alive] + dy[alive] * d
        pz = oz[alive] + dz[alive] * d
        s = map_scene(cmd_count, words, max_dist, px, py, pz)
        with np.errstate(invalid="ignore"):
            hit = s < min_dist
            esc = (~hit) & (s > max_dist)
        if hit.any():
            hit_idx.append(alive[hit])
            hit_pos.append((px[hit], py[hit], pz[hit]))
        cont = ~(hit | esc)
        dist[alive[cont]] = d[cont] + s[cont]
        alive = alive[cont]
    is_hit = np.zeros(n, dtype=bool)
    if hit_idx:
        hi = np.concatenate(hit_idx)
        hx = np.concatenate([p[0] for p in hit_pos])
        hy = np.concatenate([p[1] for p in hit_pos])
        hz = np.concatenate([p[2] for p in hit_pos])
        is_hit[hi] = True
        eps = F(0.0001)
        ks = [(1, -1, -1), (-1, -1, 1), (-1, 1, -1), (1, 1, 1)]
        acc = None
        for k in ks:
            kx, ky, kz = F(k[0]), F(k[1]), F(k[2])
            f = map_scene(cmd_count, words, max_dist, hx + kx * eps, hy + ky * eps, hz + kz * eps)
            term = (kx * f, ky * f, kz * f)
            acc = term if acc is None else (acc[0] + term[0], acc[1] + term[1], acc[2] + term[2])
        nx, ny, nz = _normalize3(*acc)
        lx, ly, lz = _normalize3(hx - F(2.0), hy - F(-5.0), hz - F(3.0))
        diffuse = fmax(F(0.02), (nx * lx + ny * ly) + nz * lz)
        if materials is None:
            col[0, hi] = F(0.4) * diffuse
            col[1, hi] = F(0.7) * diffuse
            col[2, hi] = F(0.1) * diffuse
        else:
            table = np.asarray(materials, dtype=F).reshape(-1, 3)
            _, m = map_scene(cmd_count, words, max_dist, hx, hy, hz, want_material=True)
            for ch in range(3):
                col[ch, hi] = table[m, ch] * diffuse
    miss = np.nonzero(~is_hit)[0]
    if miss.size:
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            fd = (F(-1.5) - oy[miss]) / dy[miss]
            on = fd > 0
            m = miss[on]
            t = fd[on]
            fx = ox[m] + dx[m] * t
            fz = oz[m] + dz[m] * t
            ix = f2i(np.rint(fx + F(0.5)))
            iz = f2i(np.rint(fz + F(0.5)))
        c = ((ix ^ iz) & 1).astype(F)
        g = F(0.2) * c
        col[0, m] = F(0.1) + g
        col[1, m] = F(0.1) + g
        col[2, m] = F(0.2) + g
    return col


def _matvec(m, x, y, z, w):
    """column-major 4x4 (flat 16) times vec4, ((c0*x + c1*y) + c2*z) + c3*w."""
    m = _f(m)
    return tuple(((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w for r in range(4))


def render(uniforms, limits, cmd_count, words, W, H, row0=0, rows=None, materials=None):
    """fs_main (wgsl:36-76) over rows [row0,row0+rows) -> (rows, W, 4) float32."""
    rows = H - row0 if rows is None else rows
    ve = _f(uniforms["viewport_extent"])
    inv_proj, inv_view = _f(uniforms["inv_proj"]), _f(uniforms["inv_view"])
    one, zero = F(1), F(0)
    ro = _matvec(inv_view, zero, zero, zero, one)
    pxs = np.arange(W, dtype=np.uint32)
    pys = np.arange(row0, row0 + rows, dtype=np.uint32)
    sx = ((pxs.astype(F) + F(0.5)) / F(W)) * F(2.0) - F(1.0)
    sy = F(1.0) - ((pys.astype(F) + F(0.5)) / F(H)) * F(2.0)
    SX, SY = np.meshgrid(sx, sy)      # (rows, W)
    SX, SY = SX.ravel(), SY.ravel()
    n = SX.size
    total = np.zeros((3, n), dtype=F)
    for i in range(4):
        for j in range(4):
            rx = (F(i) + F(0.5)) / F(4) - F(0.5)
            ry = (F(j) + F(0.5)) / F(4) - F(0.5)
            ox_s = rx / ve[0] * F(2.0)
            oy_s = ry / ve[1] * F(2.0)
            x = SX + ox_s
            y = SY + oy_s
            zc = np.full(n, F(-1.0)); wc = np.full(n, F(1.0))
            pv = _matvec(inv_proj, x, y, zc, wc)
            pw = _matvec(inv_view, *pv)
            d = [pw[k] - ro[k] for k in range(4)]
            ln = np.sqrt(((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) + d[3] * d[3])
            rd = [d[k] / ln for k in range(3)]
            o = [np.full(n, ro[k], dtype=F) for k in range(3)]
            c = ray_march(cmd_count, words, limits, o[0], o[1], o[2], rd[0], rd[1], rd[2], materials)
            total = total + np.sqrt(c)
    out = np.empty((rows * W, 4), dtype=F)
    out[:, 0] = total[0] / F(16)
    out[:, 1] = total[1] / F(16)
    out[:, 2] = total[2] / F(16)
    out[:, 3] = F(1.0)
    return out.reshape(rows, W, 4)
