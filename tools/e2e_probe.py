#!/usr/bin/env python3
"""Where does the end-to-end (draw + D2H) loop of bench.py lose its overlap?  One GPU, 1080p G32."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _ffi, camera, csg, renderer, shard

W, H, K = 1920, 1080, 60
cc, words = csg.serialize(csg.scene("g32"))
ctl = camera.OrbitCameraController.new([0, 0, 0], 5.0)
ctl.update(camera.Orbit([35.0, -25.0]))
u = renderer.prepare_uniforms((float(W), float(H)), ctl.camera())


def ctx():
    r = renderer.RayMarchingResources(0)
    r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
    r.set_limits(renderer.RayMarchLimits(0.01, 100.0, 256))
    r.set_program(cc, words)
    r.set_uniforms(u)
    return r


def run(label, D, copy, pinned_kind, copy_stream=False):
    cs = [ctx() for _ in range(D)]
    st = [torch.cuda.Stream() for _ in range(D)]
    cst = [torch.cuda.Stream() for _ in range(D)] if copy_stream else st
    dev = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(D)]
    if pinned_kind == "torch":
        host = [torch.empty((H, W, 4), dtype=torch.float32).pin_memory() for _ in range(D)]
        addr = [h.data_ptr() for h in host]
        img = None
    else:
        img = shard.SharedImage("rm_probe_%d" % os.getpid(), W, H, slots=D).open(0, 1, lambda: None)
        img.register()
        addr = [img.slot_address(i) for i in range(D)]
    done = [torch.cuda.Event() for _ in range(D)]
    drawn = [torch.cuda.Event() for _ in range(D)]
    for i in range(D):
        cs[i].draw_device(W, H, dev[i].data_ptr(), stream=st[i].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        i = k % D
        if k >= D:
            done[i].synchronize()
        cs[i].draw_device(W, H, dev[i].data_ptr(), stream=st[i].cuda_stream)
        if copy:
            if copy_stream:
                drawn[i].record(st[i])
                cst[i].wait_event(drawn[i])
            cs[i].gather_strips(W, H, 16, 0, 1, dev[i].data_ptr(), addr[i], stream=cst[i].cuda_stream)
        done[i].record(cst[i])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print("%-70s %.3f ms per frame  %.0f Mpx/s" % (label, dt * 1e3, W * H / dt / 1e6))
    for c in cs:
        c.close()
    if img is not None:
        img.close()


run("draw only, 1 frame in flight", 1, False, "torch")
run("draw only, 2 in flight", 2, False, "torch")
run("draw + D2H (torch pinned), 1 in flight", 1, True, "torch")
run("draw + D2H (torch pinned), 2 in flight", 2, True, "torch")
run("draw + D2H (registered shm), 2 in flight", 2, True, "shm")
run("draw + D2H (registered shm), 3 in flight", 3, True, "shm")
run("draw + D2H (registered shm) on a copy stream, 2 in flight", 2, True, "shm", copy_stream=True)
run("draw + D2H (registered shm) on a copy stream, 3 in flight", 3, True, "shm", copy_stream=True)
run("draw + D2H (torch pinned) on a copy stream, 3 in flight", 3, True, "torch", copy_stream=True)
