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
a = p.parse_args()
res = renderer.RayMarchingResources(0)
res.set_option(_ffi.RM_OPT_SPECIALIZE, a.specialize)
res.set_limits(renderer.RayMarchLimits(0.01, 100.0, a.max_iter))
res.set_scene(csg.scene(a.scene))
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
