"""Python face of csrc/host/renderer.hpp: RayMarchingResources / RayMarchingCallback of
src/ray_marching/renderer.rs on top of the C ABI (include/rm_abi.h).  Device memory and
streams may come from torch; the render itself is always the HIP kernel (no CPU fallback)."""
import ctypes as C

import numpy as np

from . import _ffi
from . import csg as _csg
from ._ffi import Limits, Uniforms


def prepare_uniforms(viewport, camera):
    """The host half of prepare() (renderer.rs:205-222): the 144-byte Uniforms blob."""
    u = Uniforms()
    vp = (C.c_float * 2)(float(viewport[0]), float(viewport[1]))
    _ffi.host_lib().rmh_prepare_uniforms(vp, C.byref(camera._s), C.byref(u))
    return u


class RayMarchLimits:  # renderer.rs:36-41, defaults :133-137
    def __init__(self, min_dist=0.01, max_dist=100.0, max_iter=100):
        self.min_dist, self.max_dist, self.max_iter = float(min_dist), float(max_dist), int(max_iter)

    def as_struct(self):
        return Limits(self.min_dist, self.max_dist, self.max_iter)


class RayMarchingResources:
    """Long-lived GPU state (renderer.rs:43-49) = one rm_ctx on one GPU."""

    def __init__(self, device=0):
        self._L = _ffi.hip_lib()
        h = C.c_void_p()
        rc = self._L.rm_create(int(device), C.byref(h))
        if rc != _ffi.RM_OK:
            raise _ffi.RmError(rc, (self._L.rm_last_error(None) or b"").decode())
        self._h = h
        self.device = int(device)
        self._dtype = np.float32

    def close(self):
        if getattr(self, "_h", None):
            self._L.rm_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        return _ffi.check(self._h, rc)

    # -- the three buffers of the bind group ------------------------------------------------
    def write_buffer(self, buffer, offset, data):
        """Queue::write_buffer (renderer.rs:213,230,235); data: bytes-like."""
        b = bytes(data)
        self._check(self._L.rm_write_buffer(self._h, buffer, offset, b, len(b)))

    def set_limits(self, limits):
        s = limits.as_struct() if isinstance(limits, RayMarchLimits) else Limits(*limits)
        self._check(self._L.rm_set_limits(self._h, C.byref(s)))

    def set_uniforms(self, u):
        self._check(self._L.rm_set_uniforms(self._h, C.byref(u)))

    def set_program(self, cmd_count, words):
        w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
        n = int(w.size)
        ptr = w.ctypes.data_as(C.POINTER(C.c_uint32)) if n else None
        self._check(self._L.rm_set_program(self._h, int(cmd_count), ptr, n))

    def set_materials(self, rgb):
        """Material table (extension): (n, 3) albedo values, n in [1, 256]; entry i colours the surfaces tagged Material(i)."""
        m = np.ascontiguousarray(np.asarray(rgb, dtype=np.float32).reshape(-1, 3))
        self._check(self._L.rm_set_materials(self._h, len(m), m.ctypes.data_as(C.POINTER(C.c_float))))

    def set_scene(self, node):
        self.set_program(*_csg.serialize(node))

    def resize_command_buffer(self, nbytes):
        self._check(self._L.rm_resize_command_buffer(self._h, int(nbytes)))

    def validate(self):
        self._check(self._L.rm_validate(self._h))

    # -- draws -------------------------------------------------------------------------------
    def set_output_format(self, fmt):
        """RM_OPT_OUTPUT_FORMAT: _ffi.RM_FORMAT_RGBA32F (default) / RGBA8_UNORM / BGRA8_UNORM.  The host-array draw
        methods then return uint8 (..., 4) arrays; the *_device methods write 4 bytes per pixel."""
        self.set_option(_ffi.RM_OPT_OUTPUT_FORMAT, fmt)
        self._dtype = np.float32 if fmt == _ffi.RM_FORMAT_RGBA32F else np.uint8

    def draw(self, W, H, row0=0, rows=None):
        """Render rows [row0,row0+rows) into a new host array (rows, W, 4): float32, or uint8 for the 8-bit formats."""
        rows = H - row0 if rows is None else rows
        out = np.empty((max(rows, 0), W, 4), dtype=self._dtype)
        self._check(self._L.rm_draw(self._h, W, H, row0, rows, out.ctypes.data_as(C.c_void_p), 0, None))
        return out

    def draw_device(self, W, H, out_ptr, row0=0, rows=None, stream=None):
        """Asynchronous render into device memory (out_ptr: integer device address)."""
        rows = H - row0 if rows is None else rows
        self._check(self._L.rm_draw(self._h, W, H, row0, rows, C.c_void_p(out_ptr), 1,
                                    C.c_void_p(stream) if stream else None))

    def draw_strips(self, W, H, strip_rows, first, stride):
        """This GPU's interleaved strips of a W x H image (rm_draw_strips) -> (rows, W, 4) host array."""
        from . import shard
        rows = shard.strip_row_count(H, strip_rows, first, stride)
        out = np.empty((rows, W, 4), dtype=self._dtype)
        n = C.c_uint32(0)
        self._check(self._L.rm_draw_strips(self._h, W, H, strip_rows, first, stride,
                                           out.ctypes.data_as(C.c_void_p) if rows else None, 0, None, C.byref(n)))
        assert n.value == rows
        return out

    def draw_strips_device(self, W, H, strip_rows, first, stride, out_ptr, stream=None):
        n = C.c_uint32(0)
        self._check(self._L.rm_draw_strips(self._h, W, H, strip_rows, first, stride, C.c_void_p(out_ptr), 1,
                                           C.c_void_p(stream) if stream else None, C.byref(n)))
        return n.value

    def gather_strips(self, W, H, strip_rows, first, stride, strips_ptr, host_ptr, stream=None):
        """rm_gather_strips: this GPU's strips (device buffer rm_draw_strips filled) -> their rows of the full host image
        at address host_ptr (pinned / registered memory: asynchronous on `stream`)."""
        self._check(self._L.rm_gather_strips(self._h, W, H, strip_rows, first, stride, C.c_void_p(strips_ptr),
                                             C.c_void_p(host_ptr), C.c_void_p(stream) if stream else None))

    def draw_batch(self, frames, W, H):
        arr = (Uniforms * len(frames))(*frames)
        out = np.empty((len(frames), H, W, 4), dtype=self._dtype)
        self._check(self._L.rm_draw_batch(self._h, arr, len(frames), W, H, out.ctypes.data_as(C.c_void_p), 0, None))
        return out

    def draw_batch_device(self, frames, W, H, out_ptr, stream=None):
        arr = (Uniforms * len(frames))(*frames)
        self._check(self._L.rm_draw_batch(self._h, arr, len(frames), W, H, C.c_void_p(out_ptr), 1,
                                          C.c_void_p(stream) if stream else None))

    def sync(self):
        self._check(self._L.rm_sync(self._h))

    def sync_context(self):
        """Wait for this context's own stream (draws issued with stream=_ffi.RM_STREAM_OWN)."""
        self._check(self._L.rm_sync_context(self._h))

    # -- options / info ------------------------------------------------------------------------
    def set_option(self, key, value):
        self._check(self._L.rm_set_option(self._h, key, int(value)))

    def info(self, key):
        v = C.c_double()
        self._check(self._L.rm_get_info(self._h, key, C.byref(v)))
        return v.value

    def jit_log(self):
        """Compiler / loader messages of the structure specialiser for the current program ('' if none)."""
        buf = C.create_string_buffer(1 << 16)
        self._check(self._L.rm_jit_log(self._h, buf, len(buf)))
        return buf.value.decode(errors="replace")

    def wave_stats(self, max_waves=1 << 20):
        """Diagnostics: (n_waves, 4) uint64 array recorded by the last draw with RM_OPT_WAVE_STATS."""
        buf = np.zeros((max_waves, 4), dtype=np.uint64)
        n = C.c_uint64(0)
        self._check(self._L.rm_read_wave_stats(self._h, buf.ctypes.data_as(C.c_void_p), buf.nbytes, C.byref(n)))
        return buf[: n.value // 32]

    def measure_write_bandwidth(self, nbytes=1 << 30, iters=10):
        v = C.c_double()
        self._check(self._L.rm_measure_write_bandwidth(self._h, nbytes, iters, C.byref(v)))
        return v.value


class RayMarchingCallback:
    """Per-frame value object (renderer.rs:177-193) with the reference's prepare/paint split."""

    def __init__(self, time, csg_node, viewport, camera):
        self.time = time            # carried, unused (renderer.rs:178; main.rs:74 always passes 0.0)
        self.csg_node = csg_node    # Option<CSGNode>
        self.viewport = viewport
        self.camera = camera

    @classmethod
    def new(cls, time, csg_node, viewport, camera):
        return cls(time, csg_node, viewport, camera)

    def prepare(self, resources):
        """renderer.rs:196-242: uniforms write, then cmd_count and words writes."""
        u = prepare_uniforms(self.viewport, self.camera)
        resources.write_buffer(_ffi.RM_BUF_UNIFORMS, 0, bytes(u))
        cmd_count, words = _csg.serialize(self.csg_node)
        resources.write_buffer(_ffi.RM_BUF_COMMANDS, 0, np.uint32(cmd_count).tobytes())
        resources.write_buffer(_ffi.RM_BUF_COMMANDS, 4, words.tobytes())

    def paint(self, resources, width=None, height=None):
        """renderer.rs:244-255: one draw over the viewport; returns (H, W, 4) float32."""
        W = int(self.viewport[0]) if width is None else width
        H = int(self.viewport[1]) if height is None else height
        return resources.draw(W, H)


def validate_program(cmd_count, words):
    """rm_validate_program: (status, max_depth); pure host code, no GPU needed."""
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
    depth = C.c_uint32(0)
    ptr = w.ctypes.data_as(C.POINTER(C.c_uint32)) if w.size else None
    rc = _ffi.hip_lib().rm_validate_program(int(cmd_count), ptr, int(w.size), C.byref(depth))
    return rc, depth.value


PROGRAM_FACTS = ("records", "cones", "slabs", "subtracted_leaves", "groups", "spill_depth", "is_chain", "prunable", "bound_walk",
                 "has_xforms", "leaves", "auto_pruned")


def program_info(cmd_count, words):
    """rm_program_info: what the upload-time decoder makes of a command stream, as a dict (pure host code, no GPU needed)."""
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
    ptr = w.ctypes.data_as(C.POINTER(C.c_uint32)) if w.size else None
    out = (C.c_uint32 * len(PROGRAM_FACTS))()
    rc = _ffi.hip_lib().rm_program_info(int(cmd_count), ptr, int(w.size), out, len(PROGRAM_FACTS))
    if rc != _ffi.RM_OK:
        raise _ffi.RmError(rc, _ffi.hip_lib().rm_status_string(rc).decode())
    return dict(zip(PROGRAM_FACTS, [int(v) for v in out]))


def jit_source(cmd_count, words, waves_per_tile=4, prune=False, env=None):
    """rm_jit_source: the HIP source the structure specialiser generates for a command stream (no GPU needed).
    env: A/B knobs of the generator (RM_JIT_*), set in the process environment for this call only."""
    if env:
        import os
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            return jit_source(cmd_count, words, waves_per_tile, prune)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
    ptr = w.ctypes.data_as(C.POINTER(C.c_uint32)) if w.size else None
    L = _ffi.hip_lib()
    need = C.c_size_t(0)
    waves_per_tile = int(waves_per_tile) | (_ffi.RM_JIT_PRUNE if prune else 0)
    rc = L.rm_jit_source(int(cmd_count), ptr, int(w.size), int(waves_per_tile), None, 0, C.byref(need))
    if rc != _ffi.RM_OK:
        raise _ffi.RmError(rc, L.rm_status_string(rc).decode())
    buf = C.create_string_buffer(need.value)
    L.rm_jit_source(int(cmd_count), ptr, int(w.size), int(waves_per_tile), buf, need.value, None)
    return buf.value.decode()


def jit_compile(cmd_count, words, waves_per_tile=4, prune=False):
    """rm_jit_compile: compile the specialised kernel for gfx950 with hipRTC, without loading it (no GPU needed).
    Returns (status, compile_ms, code_bytes, log)."""
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
    ptr = w.ctypes.data_as(C.POINTER(C.c_uint32)) if w.size else None
    ms, nbytes = C.c_double(0.0), C.c_size_t(0)
    log = C.create_string_buffer(1 << 16)
    waves_per_tile = int(waves_per_tile) | (_ffi.RM_JIT_PRUNE if prune else 0)
    rc = _ffi.hip_lib().rm_jit_compile(int(cmd_count), ptr, int(w.size), int(waves_per_tile), C.byref(ms),
                                       C.byref(nbytes), log, len(log))
    return rc, ms.value, nbytes.value, log.value.decode(errors="replace")
