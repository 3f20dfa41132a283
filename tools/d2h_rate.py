#!/usr/bin/env python3
"""D2H copy rate of one 1080p RGBA32F frame (33 MB) into (a) torch pinned memory, (b) a POSIX shared-memory segment
page-locked with rm_host_register (what the multi-GPU gather of bench.py uses), (c) pageable memory."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ray_marching_amd import _ffi, shard

W, H = 1920, 1080
dev = torch.rand((H, W, 4), dtype=torch.float32, device="cuda")
nbytes = dev.numel() * 4
s = torch.cuda.Stream()


def rate(dst_tensor, label, n=40):
    with torch.cuda.stream(s):
        for _ in range(3):
            dst_tensor.copy_(dev, non_blocking=True)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            dst_tensor.copy_(dev, non_blocking=True)
        s.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-40s %.3f ms per frame  %.1f GB/s" % (label, dt * 1e3, nbytes / dt / 1e9))


rate(torch.empty((H, W, 4), dtype=torch.float32).pin_memory(), "torch pinned (hipHostMalloc)")
img = shard.SharedImage("rm_d2h_%d" % os.getpid(), W, H).open(0, 1, lambda: None)
img.register()
rate(torch.from_numpy(img.array), "shared memory + hipHostRegister")
img.close()
rate(torch.empty((H, W, 4), dtype=torch.float32), "pageable")
