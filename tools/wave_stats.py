#!/usr/bin/env python3
"""Diagnostics: per-wave timeline of one draw of the metric workload (RM_OPT_WAVE_STATS)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_marching_amd import _ffi, camera, csg, renderer  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--kernel", type=int, default=13)
p.add_argument("--width", type=int, default=1920)
p.add_argument("--height", type=int, default=1080)
p.add_argument("--scene", default="g32")
p.add_argument("--max-iter", type=int, default=256)
p.add_argument("--no-balance", action="store_true")
p.add_argument("--balance", type=int, default=-1)
p.add_argument("--wpt", type=int, default=4)
p.add_argument("--no-cull", action="store_true")
p.add_argument("--specialize", type=int, default=2)
p.add_argument("--prune", action="store_true", help="pruned specialised kernel; with RM_JIT_PRUNE_STATS=1 in the environment the "
               "'refills' field becomes the number of leaves evaluated")
p.add_argument("--refill-min", type=int, default=0)
p.add_argument("--interp-stats", action="store_true", help="a library built with -DRM_INTERP_STATS (tools/ab_build.sh): the 'refills' field is the "
               "number of records the interpreter's masked loops executed")
p.add_argument("--leaves", type=int, default=16, help="primitives of the scene (for the evaluated-leaf ratio)")
a = p.parse_args()
res = renderer.RayMarchingResources(0)
res.set_option(_ffi.RM_OPT_KERNEL, a.kernel)
res.set_option(_ffi.RM_OPT_SPECIALIZE, a.specialize)
res.set_option(_ffi.RM_OPT_PRUNE, 1 if a.prune else 0)
if a.refill_min:
    res.set_option(_ffi.RM_OPT_REFILL_MIN, a.refill_min)
res.set_option(_ffi.RM_OPT_BALANCE, a.balance if a.balance >= 0 else (0 if a.no_balance else 1))
res.set_option(_ffi.RM_OPT_CULL, 0 if a.no_cull else 1)
res.set_option(_ffi.RM_OPT_WAVES_PER_TILE, a.wpt)
res.set_limits(renderer.RayMarchLimits(0.01, 100.0, a.max_iter))
res.set_scene(csg.scene(a.scene))
ctl = camera.OrbitCameraController.new([0, 0, 0], 5.0)
ctl.update(camera.Orbit([35.0, -25.0]))
res.set_uniforms(renderer.prepare_uniforms((a.width, a.height), ctl.camera()))
res.draw(a.width, a.height)
res.set_option(_ffi.RM_OPT_WAVE_STATS, 1)
res.draw(a.width, a.height)
st = res.wave_stats()
t0, t1 = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64)
iters = (st[:, 2] & 0xFFFFFFFF).astype(np.int64)
live = (st[:, 3] & 0xFFFFFFFF).astype(np.int64)
refills = (st[:, 3] >> 32).astype(np.int64)
base = t0.min()
dur = (t1 - t0) / 100.0   # us
end = (t1 - base) / 100.0
start = (t0 - base) / 100.0
print("waves %d  kernel span %.1f us" % (len(st), end.max()))
print("wave duration us: mean %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" %
      (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
print("iterations: mean %.1f p50 %d p90 %d p99 %d max %d ; sum %d" %
      (iters.mean(), np.percentile(iters, 50), np.percentile(iters, 90), np.percentile(iters, 99), iters.max(), iters.sum()))
print("lane occupancy over iterations: %.3f ; refills/wave %.1f" % (live.sum() / max(1, iters.sum() * 64), refills.mean()))
if a.prune and os.environ.get("RM_JIT_PRUNE_STATS"):
    mode = os.environ["RM_JIT_PRUNE_STATS"]
    what = {"1": "leaves evaluated", "2": "leaf tests executed (leaves of near groups)", "3": "near groups",
            "4": "(group, live lane) pairs that are near, per 64 lanes"}.get(mode, "?")
    per = a.leaves if mode in ("1", "2") else a.leaves // 2
    if mode == "4":
        refills = refills / 64.0
    print("%s: %.2f per iteration = %.3f of %d (iterations include tap phases, which are not counted)" % (
        what, refills.sum() / max(1.0, float(iters.sum())), refills.sum() / max(1.0, float(iters.sum()) * per), per))
if a.interp_stats:
    print("records executed: %.2f per iteration (iterations include tap phases: one evaluation each)" % (refills.sum() / max(1.0, float(iters.sum()))))
k = np.argsort(-end)[:8]
for i in k:
    print("  late wave: slot %d tile %d start %.0f end %.0f dur %.0f iters %d live/iter %.1f" %
          (i, st[i, 2] >> 32, start[i], end[i], dur[i], iters[i], live[i] / max(1, iters[i])))
# concurrency over time
edges = np.linspace(0, end.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    act = ((start < hi) & (end > lo)).sum()
    print("  t=%6.0f..%6.0f us: %5d waves resident" % (lo, hi, act))
print("us per iteration of the longest waves: %.2f" % (dur[k] / np.maximum(1, iters[k])).mean())
