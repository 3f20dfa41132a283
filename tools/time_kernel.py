#!/usr/bin/env python3
"""Times the march kernel alone (library-side HIP events) without importing torch.
RM_HIP_RUNTIME=system makes the process use /opt/rocm's HIP runtime, hipRTC and comgr instead of the
ones bundled with the torch wheel: the specialised kernel is then compiled by a different compiler release."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_marching_amd import _ffi, camera, csg, renderer  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--width", type=int, default=1920)
p.add_argument("--height", type=int, default=1080)
p.add_argument("--scene", default="g32")
p.add_argument("--max-iter", type=int, default=256)
p.add_argument("--specialize", type=int, default=2)
p.add_argument("--steps", type=int, default=40)
p.add_argument("--strip-tags", action="store_true", help="render the scene without its Material tags (cost of the material phase)")
a = p.parse_args()
res = renderer.RayMarchingResources(0)
res.set_option(_ffi.RM_OPT_SPECIALIZE, a.specialize)
res.set_limits(renderer.RayMarchLimits(0.01, 100.0, a.max_iter))
import numpy as np  # noqa: E402
cc, words = csg.serialize(csg.scene(a.scene))
if a.strip_tags:
    n_par = {0: 4, 1: 6, 2: 4, 10: 5, 110: 1, 200: 3, 202: 4, 204: 1, 300: 1}
    kept, i, cc = [], 0, 0
    while i < len(words):
        n = 1 + n_par.get(int(words[i]), 0)
        if int(words[i]) != 300:
            kept += [int(x) for x in words[i:i + n]]
            cc += 1
        i += n
    words = np.array(kept, dtype=np.uint32)
res.set_materials([(0.4, 0.7, 0.1), (0.9, 0.15, 0.1), (0.1, 0.3, 0.9), (0.95, 0.9, 0.2), (0.8, 0.8, 0.8), (0.6, 0.1, 0.7), (0.2, 0.2, 0.2), (1.0, 1.0, 1.0)])
res.set_program(cc, words)
ctl = camera.OrbitCameraController.new([0, 0, 0], 5.0)
ctl.update(camera.Orbit([35.0, -25.0]))
res.set_uniforms(renderer.prepare_uniforms((a.width, a.height), ctl.camera()))
for _ in range(5):
    res.draw(a.width, a.height)
res.set_option(_ffi.RM_OPT_TIMING, 1)
for _ in range(a.steps):
    img = res.draw(a.width, a.height)
ms = res.info(_ffi.RM_INFO_KERNEL_MS)
maps = [l.split()[-1] for l in open("/proc/self/maps") if any(k in l for k in ("hiprtc", "comgr", "amdhip64"))]
print("march kernel %.4f ms  specialised=%d  jit_ms=%.0f  checksum %.6f" %
      (ms, res.info(_ffi.RM_INFO_SPECIALIZED), res.info(_ffi.RM_INFO_JIT_COMPILE_MS), float(img[..., :3].astype("float64").sum())))
print("libraries:", sorted(set(maps)))
