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


def test_shared_image_slots_counters_and_stale_segment():
    """The shared frame of the multi-GPU gather: frame slots, one completion counter per rank, and a segment left behind
    by a run that died is replaced instead of failing the next run."""
    import os
    import threading
    import time
    from multiprocessing import shared_memory
    name = "rm_test_shared_%d" % os.getpid()
    stale = shared_memory.SharedMemory(name=name, create=True, size=64)      # what a crashed run leaves in /dev/shm
    stale.close()
    img = shard.SharedImage(name, 8, 6, slots=2).open(0, 1, lambda: None)
    try:
        assert img.images.shape == (2, 6, 8, 4) and img.slot(3) is not None and img.slot_address(1) - img.slot_address(0) == 6 * 8 * 16
        assert img.flags.shape == (1,) and int(img.flags[0]) == 0
        img.slot(1)[:] = 7.0
        assert float(img.images[1].min()) == 7.0 and float(img.images[0].max()) == 0.0
        img.mark_done(0, 3)
        img.wait_all(3)                                                        # returns at once
        t = threading.Timer(0.2, lambda: img.mark_done(0, 4))
        t.start()
        t0 = time.monotonic()
        img.wait_all(4, timeout=10.0)                                          # waits for the counter
        assert 0.1 < time.monotonic() - t0 < 5.0
        with pytest.raises(TimeoutError):
            img.wait_all(5, timeout=0.2)
    finally:
        img.close()
    with pytest.raises(FileNotFoundError):
        shared_memory.SharedMemory(name=name)                                  # the owner unlinked it
