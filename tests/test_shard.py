"""Multi-rank partitioning (ray-marching_amd/shard.py): index arithmetic, and the world_size-2
host-side gather over gloo on CPU.  The renderer is injected; here the ORACLE stands in for the
GPU (tests may use the oracle; the product's own draw_strips is the HIP kernel)."""
import os
import socket

import numpy as np
import pytest

import scenes
from ray_marching_amd import shard


def test_strip_partition_covers_image_once():
    for H in (1, 15, 16, 17, 270, 1080, 2160):
        for world in (1, 2, 3, 8):
            seen = np.zeros(H, dtype=int)
            for r in range(world):
                for r0, rows in shard.strips_of_rank(H, r, world):
                    seen[r0:r0 + rows] += 1
                assert shard.strip_row_count(H, 16, r, world) == sum(x[1] for x in shard.strips_of_rank(H, r, world))
            assert (seen == 1).all()
    assert shard.strips_of_rank(1080, 3, 8)[:2] == [(48, 16), (176, 16)]       # strip 3, 11, ...
    assert shard.strips_of_rank(1080, 3, 8)[-1] == (1072, 8)                    # 1080 = 67.5 strips: strip 67 is half


def test_frame_sharding():
    assert shard.frames_of_rank(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard.frames_of_rank(1024, r, 8) for r in range(8)), [])) == list(range(1024))


def test_scatter_roundtrip():
    H, W = 100, 7
    img = np.random.default_rng(0).random((H, W, 4)).astype(np.float32)
    out = np.zeros_like(img)
    for r in range(3):
        compact = np.concatenate([img[r0:r0 + n] for r0, n in shard.strips_of_rank(H, r, 3)])
        shard.scatter_strips(out, compact, H, r, 3)
    assert out.tobytes() == img.tobytes()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, W, H, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cbind
    cc, w = cbind.serialize(*scenes.g8())
    u, *_ = cbind.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 48)

    def draw_strips(strip_rows, first, stride):          # stand-in for RayMarchingResources.draw_strips
        parts = [cbind.render(u, lim, cc, w, W, H, row0=r0, rows=n) for r0, n in shard.strips_of_rank(H, first, stride, strip_rows)]
        return np.concatenate(parts) if parts else np.empty((0, W, 4), np.float32)

    img = shard.render_tiled(draw_strips, W, H, rank, world)
    frames = shard.frames_of_rank(5, rank, world)
    dist.barrier()
    # the same frame through the shared-memory gather: every rank ends up seeing the whole image
    shared = shard.SharedImage("rm_test_%d" % port, W, H).open(rank, world, dist.barrier)
    whole = shard.render_tiled_shared(draw_strips, shared, rank, world, dist.barrier)
    full = cbind.render(u, lim, cc, w, W, H)
    same_shared = whole.tobytes() == full.tobytes()
    shared.close(dist.barrier)
    if rank == 0:
        q.put((img.tobytes() == full.tobytes() and same_shared, frames))
    else:
        assert img is None and same_shared
    dist.destroy_process_group()


def test_two_rank_gloo_tiled_render_matches_single():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    W, H = 48, 72        # 4.5 strips of 16 rows: rank 0 gets 3 strips (one partial), rank 1 two
    procs = [ctx.Process(target=_worker, args=(r, 2, port, W, H, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, frames = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok and frames == [0, 2, 4]
