#!/usr/bin/env python3
"""Experiment: K frames drawn through N contexts on N streams (frame f -> context f % N) against one context."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _ffi, camera, csg, renderer

W, H, K = 1920, 1080, 600
for n in (1, 2, 3, 4, 6, 8):
    ctx, streams, outs = [], [], []
    for i in range(n):
        r = renderer.RayMarchingResources(0)
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(renderer.RayMarchLimits(0.01, 100.0, 256))
        r.set_scene(csg.scene("g32"))
        ctl = camera.OrbitCameraController.new([0, 0, 0], 5.0)
        ctl.update(camera.Orbit([35.0, -25.0]))
        r.set_uniforms(renderer.prepare_uniforms((W, H), ctl.camera()))
        ctx.append(r)
        streams.append(torch.cuda.Stream())
        outs.append(torch.empty((H, W, 4), dtype=torch.float32, device="cuda"))
    for f in range(20):
        ctx[f % n].draw_device(W, H, outs[f % n].data_ptr(), stream=streams[f % n].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(K):
        ctx[f % n].draw_device(W, H, outs[f % n].data_ptr(), stream=streams[f % n].cuda_stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d context(s)/stream(s): %.1f Mpx/s  (%.3f ms per frame)  checksum %.6f" %
          (n, W * H * K / dt / 1e6, dt / K * 1e3, float(outs[0][..., :3].double().sum())))
    for r in ctx:
        r.close()
