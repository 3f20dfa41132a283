#!/usr/bin/env python3
"""What ONE rank of an N-GPU tiled frame costs, measured on one GPU: rank r of N draws its interleaved 16-row strips
(rm_draw_strips) with four frames in flight, as bench.py's tile mode does.  W x H / (time per frame) is the job rate N such
GPUs would reach if nothing else got in the way (the strips interleave, so the ranks' shares are alike) -- it shows where
the fixed per-draw costs (three launches, pre-pass, sort) start to bound strong scaling.  With --gather each frame's strips
are also copied to page-locked host memory (the rank's part of the host-side gather)."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from ray_marching_amd import _ffi, camera, csg, renderer, shard  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--width", type=int, default=1920)
p.add_argument("--height", type=int, default=1080)
p.add_argument("--scene", default="g32")
p.add_argument("--max-iter", type=int, default=256)
p.add_argument("--steps", type=int, default=200)
p.add_argument("--frames-in-flight", type=int, default=4)
p.add_argument("--worlds", default="1,2,4,8")
p.add_argument("--gather", action="store_true")
p.add_argument("--waves-per-tile", type=int, default=0)
p.add_argument("--strip-rows", type=int, default=shard.DEFAULT_STRIP_ROWS)
a = p.parse_args()
W, H, K, F = a.width, a.height, a.steps, a.frames_in_flight
SR = a.strip_rows
cc, words = csg.serialize(csg.scene(a.scene))
ctl = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)
ctl.update(camera.Orbit([35.0, -25.0]))
u = renderer.prepare_uniforms((float(W), float(H)), ctl.camera())
for world in [int(x) for x in a.worlds.split(",")]:
    for rank in sorted({0, world - 1}):
        rows = shard.strip_row_count(H, SR, rank, world)
        ctxs = []
        for _ in range(F):
            r = renderer.RayMarchingResources(0)
            r.set_limits(renderer.RayMarchLimits(0.01, 100.0, a.max_iter))
            if a.waves_per_tile:
                r.set_option(_ffi.RM_OPT_WAVES_PER_TILE, a.waves_per_tile)
            r.set_program(cc, words)
            r.set_uniforms(u)
            ctxs.append(r)
        streams = [torch.cuda.Stream() for _ in range(F)]
        dev = [torch.empty((rows, W, 4), dtype=torch.float32, device="cuda") for _ in range(F)]
        host = [torch.empty((H, W, 4), dtype=torch.float32).pin_memory() for _ in range(2)] if a.gather else None

        def frame(k):
            i = k % F
            ctxs[i].set_uniforms(u)
            ctxs[i].set_program(cc, words)
            ctxs[i].draw_strips_device(W, H, SR, rank, world, dev[i].data_ptr(), stream=streams[i].cuda_stream)
            if a.gather:
                ctxs[i].gather_strips(W, H, SR, rank, world, dev[i].data_ptr(), host[k % 2].data_ptr(), stream=streams[i].cuda_stream)

        t_end = time.perf_counter() + 1.0        # a second of the same work first: clocks up, tile order learned
        k = 0
        while time.perf_counter() < t_end:
            for _ in range(4 * F):
                frame(k)
                k += 1
            torch.cuda.synchronize()
        runs, issue = [], []
        for _ in range(5):
            t0 = time.perf_counter()
            for k in range(K):
                frame(k)
            issue.append((time.perf_counter() - t0) / K * 1e3)   # host time to issue a frame (the queue absorbs it while it is shorter)
            torch.cuda.synchronize()
            runs.append((time.perf_counter() - t0) / K * 1e3)
        ms = sorted(runs)[len(runs) // 2]
        print("F=%d wpt=%d strips of %d rows, N=%d rank %d: %4d rows  %.3f ms per frame (median of 5 runs, min %.3f)%s -> %6.0f Mpx/s for the job  (host issue %.3f ms per frame)"
              % (F, a.waves_per_tile, SR, world, rank, rows, ms, min(runs), " incl. D2H of the strips" if a.gather else "", W * H / ms / 1e3, sorted(issue)[len(issue) // 2]), flush=True)
        for r in ctxs:
            r.close()
