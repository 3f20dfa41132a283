#!/usr/bin/env python3
"""Experiment: ONE frame at a time, drawn as P parts on P contexts / streams (interleaved 16-row strips, or row bands), all joined before the
next frame starts -- what would a single rm_draw gain if it overlapped its own pre-pass / sort / march chains?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _ffi, camera, csg, renderer, shard

W, H, K = 1920, 1080, 300
scene = sys.argv[1] if len(sys.argv) > 1 else "g32"


def make(n):
    ctx = []
    for i in range(n):
        r = renderer.RayMarchingResources(0)
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(renderer.RayMarchLimits(0.01, 100.0, 256))
        r.set_scene(csg.scene(scene))
        ctl = camera.OrbitCameraController.new([0, 0, 0], 5.0)
        ctl.update(camera.Orbit([35.0, -25.0]))
        r.set_uniforms(renderer.prepare_uniforms((W, H), ctl.camera()))
        ctx.append(r)
    return ctx


for parts, mode in ((1, "whole"), (2, "bands"), (2, "strips"), (3, "strips"), (4, "strips"), (4, "bands")):
    ctx = make(parts)
    streams = [torch.cuda.Stream() for _ in range(parts)]
    out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
    bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(parts)]

    def frame():
        if mode == "whole":
            ctx[0].draw_device(W, H, out.data_ptr(), stream=streams[0].cuda_stream)
        elif mode == "bands":
            rows = (H // parts + 7) // 8 * 8
            for p in range(parts):
                r0 = p * rows
                n = min(rows, H - r0)
                ctx[p].draw_device(W, H, out.data_ptr() + r0 * W * 16, row0=r0, rows=n, stream=streams[p].cuda_stream)
        else:
            for p in range(parts):
                ctx[p].draw_strips_device(W, H, 16, p, parts, bufs[p].data_ptr(), stream=streams[p].cuda_stream)
        torch.cuda.synchronize()

    for _ in range(20):
        frame()
    t0 = time.perf_counter()
    for _ in range(K):
        frame()
    dt = time.perf_counter() - t0
    print("%s, %d part(s) (%s): %.3f ms per frame, one frame at a time (host sync after every frame)" % (scene, parts, mode, dt / K * 1e3))
    for r in ctx:
        r.close()
