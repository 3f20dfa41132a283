"""Partitioning of the render over the GPUs of one node -- one process per GPU, no data-path
collective (tiles and frames are independent).

  image tiling   strips of `strip_rows` rows, strip s -> rank s % world (interleaved: the costly
                 centre of a frame is spread over all ranks); each rank renders its strips in one
                 launch (rm_draw_strips) into a compact buffer; a final HOST-side gather
                 reassembles the frame (gloo / plain copies -- never RCCL).
  frame sharding orbit batches: frame f -> rank f % world.

Everything here is index arithmetic plus an optional torch.distributed gather of CPU tensors;
the rendering itself is injected as a callable so that the logic is testable without a GPU.
"""
import numpy as np

DEFAULT_STRIP_ROWS = 16


def n_strips(H, strip_rows=DEFAULT_STRIP_ROWS):
    return (H + strip_rows - 1) // strip_rows


def strips_of_rank(H, rank, world, strip_rows=DEFAULT_STRIP_ROWS):
    """[(row0, rows), ...] of the strips rank, rank + world, ... in image order."""
    out = []
    for s in range(rank, n_strips(H, strip_rows), world):
        r0 = s * strip_rows
        out.append((r0, min(strip_rows, H - r0)))
    return out


def strip_row_count(H, strip_rows, first, stride):
    return sum(rows for _, rows in strips_of_rank(H, first, stride, strip_rows))


def frames_of_rank(n_frames, rank, world):
    return list(range(rank, n_frames, world))


def scatter_strips(image, compact, H, rank, world, strip_rows=DEFAULT_STRIP_ROWS):
    """Write one rank's compact strip buffer (rows, W, 4) into the full (H, W, 4) image."""
    at = 0
    for r0, rows in strips_of_rank(H, rank, world, strip_rows):
        image[r0:r0 + rows] = compact[at:at + rows]
        at += rows
    assert at == compact.shape[0]
    return image


def gather_image(compact, W, H, rank, world, strip_rows=DEFAULT_STRIP_ROWS, group=None, dst=0):
    """Host-side gather of every rank's strips into the full frame on rank `dst` (None elsewhere).
    `compact` is this rank's host array; the exchange uses CPU tensors (gloo)."""
    if world == 1:
        return scatter_strips(np.empty((H, W, 4), np.float32), compact, H, 0, 1, strip_rows)
    import torch
    import torch.distributed as dist
    max_rows = max(strip_row_count(H, strip_rows, r, world) for r in range(world))
    pad = torch.zeros((max_rows, W, 4), dtype=torch.float32)
    pad[:compact.shape[0]] = torch.from_numpy(np.ascontiguousarray(compact))
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    image = np.empty((H, W, 4), np.float32)
    for r in range(world):
        rows = strip_row_count(H, strip_rows, r, world)
        scatter_strips(image, bufs[r][:rows].numpy(), H, r, world, strip_rows)
    return image


class SharedImage:
    """One (H, W, 4) image in POSIX shared memory that every rank of the node maps: the host-side gather of a tiled
    frame becomes "each rank copies its own strips to their place" plus a barrier -- no rank relays another rank's
    pixels (gather_image moves every strip twice and serialises on the destination).  Rank `dst` creates the segment,
    the others attach after the barrier in open(); close() unlinks it on `dst`."""

    def __init__(self, name, W, H, dtype=np.float32):
        self.name, self.W, self.H, self.dtype = name, W, H, np.dtype(dtype)
        self._shm = None
        self.array = None

    def open(self, rank, world, barrier, dst=0):
        from multiprocessing import shared_memory
        nbytes = self.H * self.W * 4 * self.dtype.itemsize
        if rank == dst:
            self._shm = shared_memory.SharedMemory(name=self.name, create=True, size=max(nbytes, 1))
        if world > 1:
            barrier()
        if rank != dst:
            self._shm = shared_memory.SharedMemory(name=self.name)
            try:        # the creator owns the segment; keep this process's resource tracker from unlinking it at exit
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self.array = np.ndarray((self.H, self.W, 4), dtype=self.dtype, buffer=self._shm.buf)
        self._owner = rank == dst
        return self

    def put_strips(self, compact, rank, world, strip_rows=DEFAULT_STRIP_ROWS):
        scatter_strips(self.array, compact, self.H, rank, world, strip_rows)

    def close(self, barrier=None):
        self.array = None
        if barrier is not None:
            barrier()
        if self._shm is not None:
            self._shm.close()
            if self._owner:
                self._shm.unlink()
            self._shm = None


def render_tiled_shared(draw_strips, image, rank, world, barrier, strip_rows=DEFAULT_STRIP_ROWS):
    """draw_strips(strip_rows, first, stride) -> compact host array, written straight into the SharedImage; after the
    barrier every rank sees the whole frame in image.array."""
    image.put_strips(draw_strips(strip_rows, rank, world), rank, world, strip_rows)
    if world > 1:
        barrier()
    return image.array


def render_tiled(draw_strips, W, H, rank, world, strip_rows=DEFAULT_STRIP_ROWS, group=None, dst=0):
    """draw_strips(strip_rows, first, stride) -> compact host array; returns the frame on `dst`."""
    compact = draw_strips(strip_rows, rank, world)
    return gather_image(compact, W, H, rank, world, strip_rows, group, dst)
