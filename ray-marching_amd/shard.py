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
    frame becomes "each rank copies its own strips to their place" plus a completion flag per rank -- no rank relays
    another rank's pixels (gather_image moves every strip twice and serialises on the destination).  Rank `dst`
    creates the segment, the others attach after the barrier in open(); close() unlinks it on `dst`.

    Behind the image the segment holds one int64 per rank: the number of frames whose strips that rank has delivered
    (mark_done / wait_all: a frame is complete when every rank's counter has reached it; aligned 8-byte stores, one
    writer per word).  register() page-locks the mapping (rm_host_register) so that rm_gather_strips can copy device
    strips straight into it, asynchronously."""

    def __init__(self, name, W, H, dtype=np.float32, slots=1):
        self.name, self.W, self.H, self.dtype, self.slots = name, W, H, np.dtype(dtype), max(1, int(slots))
        self._shm = None
        self.array = None       # slot 0; frames in flight use slot(k % slots), one image each
        self.images = None      # (slots, H, W, 4)
        self.flags = None
        self._registered = False

    def open(self, rank, world, barrier, dst=0):
        from multiprocessing import shared_memory
        nbytes = self.slots * self.H * self.W * 4 * self.dtype.itemsize
        self._image_bytes = (nbytes + 4095) // 4096 * 4096
        total = self._image_bytes + 8 * max(world, 1)
        if rank == dst:
            try:
                self._shm = shared_memory.SharedMemory(name=self.name, create=True, size=total)
            except FileExistsError:      # left behind by a run that died: the name is ours (it carries our launcher's pid)
                stale = shared_memory.SharedMemory(name=self.name)
                stale.close()
                stale.unlink()
                self._shm = shared_memory.SharedMemory(name=self.name, create=True, size=total)
            np.ndarray((max(world, 1),), dtype=np.int64, buffer=self._shm.buf, offset=self._image_bytes)[:] = 0
        if world > 1:
            barrier()
        if rank != dst:
            self._shm = shared_memory.SharedMemory(name=self.name)
            try:        # the creator owns the segment; keep this process's resource tracker from unlinking it at exit
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self.images = np.ndarray((self.slots, self.H, self.W, 4), dtype=self.dtype, buffer=self._shm.buf)
        self.array = self.images[0]
        self.flags = np.ndarray((max(world, 1),), dtype=np.int64, buffer=self._shm.buf, offset=self._image_bytes)
        self._owner = rank == dst
        self._world = max(world, 1)
        return self

    @property
    def address(self):
        return self.images.ctypes.data

    def slot(self, i):
        return self.images[i % self.slots]

    def slot_address(self, i):
        return self.images.ctypes.data + (i % self.slots) * self.H * self.W * 4 * self.dtype.itemsize

    def register(self):
        """Page-lock the image for asynchronous device-to-host copies (needs the HIP library; GPU ranks only)."""
        from . import _ffi
        rc = _ffi.hip_lib().rm_host_register(self.address, self._image_bytes)
        if rc != _ffi.RM_OK:
            raise _ffi.RmError(rc, "rm_host_register of the shared frame failed")
        self._registered = True

    def put_strips(self, compact, rank, world, strip_rows=DEFAULT_STRIP_ROWS):
        scatter_strips(self.array, compact, self.H, rank, world, strip_rows)

    def mark_done(self, rank, frame_no):
        """This rank's strips of frames < frame_no are in place."""
        self.flags[rank] = frame_no

    def wait_all(self, frame_no, timeout=60.0):
        """Until every rank has delivered frame_no frames (the gather's only synchronisation)."""
        import time
        t0 = time.monotonic()
        spins = 0
        while int(self.flags.min()) < frame_no:
            spins += 1
            if spins > 200:
                time.sleep(0)   # yield: ranks may outnumber cores
            if time.monotonic() - t0 > timeout:
                raise TimeoutError("a rank did not deliver frame %d within %.0f s" % (frame_no, timeout))

    def close(self, barrier=None):
        if self._registered:
            from . import _ffi
            _ffi.hip_lib().rm_host_unregister(self.address)
            self._registered = False
        self.array = None
        self.images = None
        self.flags = None
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
