"""ctypes binding of oracle/librm_oracle.so (TEST INFRASTRUCTURE, see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librm_oracle.so")


class Uniforms(C.Structure):
    _fields_ = [("viewport_extent", C.c_float * 2), ("_pad", C.c_float * 2),
                ("inv_proj", C.c_float * 16), ("inv_view", C.c_float * 16)]


class Limits(C.Structure):
    _fields_ = [("min_dist", C.c_float), ("max_dist", C.c_float), ("max_iter", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("march_steps", "normal_taps", "rays", "hits", "floor_hits", "sky")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Node(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("p", C.c_float * 6), ("lhs", C.c_int32), ("rhs", C.c_int32)]


class Orbit(C.Structure):
    _fields_ = [("target", C.c_float * 3), ("pitch", C.c_float), ("yaw", C.c_float), ("radius", C.c_float),
                ("pan_speed", C.c_float), ("yaw_speed", C.c_float), ("pitch_speed", C.c_float),
                ("dolly_speed", C.c_float)]


def build(force=False):
    """Compile the oracle with gcc (building the checker is not using it)."""
    src = os.path.join(_HERE, "rm_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "librm_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u32, f32p, u32p = C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)
        L.rmo_validate_program.argtypes = [u32, u32p, u32, C.c_int, u32p]
        L.rmo_validate_program.restype = C.c_int
        L.rmo_map_scene.argtypes = [u32, u32p, u32, C.POINTER(Limits), f32p]
        L.rmo_map_scene.restype = C.c_float
        L.rmo_ray_march.argtypes = [u32, u32p, u32, C.POINTER(Limits), f32p, f32p, f32p]
        L.rmo_ray_march.restype = None
        L.rmo_render.argtypes = [C.POINTER(Uniforms), C.POINTER(Limits), u32, u32p, u32, u32, u32, u32, u32,
                                 f32p, C.POINTER(Counters)]
        L.rmo_render.restype = C.c_int
        L.rmo_render_mt.argtypes = L.rmo_render.argtypes + [u32]
        L.rmo_render_mt.restype = C.c_int
        L.rmo_render_mt_materials.argtypes = L.rmo_render_mt.argtypes + [f32p, u32]
        L.rmo_render_mt_materials.restype = C.c_int
        L.rmo_map_scene_material.argtypes = L.rmo_map_scene.argtypes
        L.rmo_map_scene_material.restype = u32
        L.rmo_quantize_unorm8.argtypes = [f32p, C.c_uint64, C.c_int, C.POINTER(C.c_uint8)]
        L.rmo_quantize_unorm8.restype = None
        L.rmo_builder_new.restype = C.c_void_p
        L.rmo_builder_free.argtypes = [C.c_void_p]
        L.rmo_builder_cmd_count.argtypes = [C.c_void_p]
        L.rmo_builder_cmd_count.restype = u32
        L.rmo_builder_n_words.argtypes = [C.c_void_p]
        L.rmo_builder_n_words.restype = u32
        L.rmo_builder_words.argtypes = [C.c_void_p]
        L.rmo_builder_words.restype = u32p
        L.rmo_build_commands.argtypes = [C.POINTER(Node), C.c_int32, C.c_void_p]
        L.rmo_build_commands.restype = None
        L.rmo_perspective_inverse.argtypes = [C.c_float] * 4 + [f32p]
        L.rmo_orbit_new.argtypes = [C.POINTER(Orbit), f32p, C.c_float]
        L.rmo_orbit_update.argtypes = [C.POINTER(Orbit), C.c_int, C.c_float, C.c_float]
        L.rmo_orbit_camera.argtypes = [C.POINTER(Orbit), f32p, f32p]
        L.rmo_camera_inv_view.argtypes = [f32p, f32p, f32p]
        L.rmo_camera_inv_view.restype = C.c_int
        L.rmo_prepare_uniforms.argtypes = [C.c_float, C.c_float, f32p, f32p, C.POINTER(Uniforms)]
        L.rmo_prepare_uniforms.restype = C.c_int
        L.rmo_sizeof_uniforms.restype = u32
        L.rmo_sizeof_limits.restype = u32
        _lib = L
    return _lib


def _f32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def _words(words):
    w = np.ascontiguousarray(np.asarray(words, dtype=np.uint32))
    if w.size == 0:
        w = np.zeros(1, dtype=np.uint32)
    return w


def serialize(nodes, root):
    """nodes: list of (kind, params[<=6], lhs, rhs).  Returns (cmd_count, np.uint32 words)."""
    L = lib()
    arr = (Node * len(nodes))()
    for i, (kind, p, lhs, rhs) in enumerate(nodes):
        arr[i].kind = kind
        for k, v in enumerate(p):
            arr[i].p[k] = v
        arr[i].lhs, arr[i].rhs = lhs, rhs
    b = L.rmo_builder_new()
    try:
        L.rmo_build_commands(arr, root, b)
        n = L.rmo_builder_n_words(b)
        cc = L.rmo_builder_cmd_count(b)
        ptr = L.rmo_builder_words(b)
        words = np.array([ptr[i] for i in range(n)], dtype=np.uint32)
    finally:
        L.rmo_builder_free(b)
    return int(cc), words


def validate(cmd_count, words, strict=False):
    w = _words(words)
    depth = C.c_uint32(0)
    rc = lib().rmo_validate_program(cmd_count, _u32p(w), len(np.asarray(words)), int(strict), C.byref(depth))
    return rc, depth.value


def map_scene(cmd_count, words, pos, limits=(0.01, 100.0, 100)):
    w = _words(words)
    lim = Limits(*limits)
    p = np.asarray(pos, dtype=np.float32)
    return float(lib().rmo_map_scene(cmd_count, _u32p(w), len(np.asarray(words)), C.byref(lim), _f32p(p)))


def map_scene_material(cmd_count, words, pos, limits=(0.01, 100.0, 100)):
    """Extension: the material index carried by map_scene's result at pos."""
    w = _words(words)
    lim = Limits(*limits)
    p = np.asarray(pos, dtype=np.float32)
    return int(lib().rmo_map_scene_material(cmd_count, _u32p(w), len(np.asarray(words)), C.byref(lim), _f32p(p)))


def ray_march(cmd_count, words, o, d, limits=(0.01, 100.0, 100)):
    w = _words(words)
    lim = Limits(*limits)
    o = np.asarray(o, dtype=np.float32)
    d = np.asarray(d, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().rmo_ray_march(cmd_count, _u32p(w), len(np.asarray(words)), C.byref(lim), _f32p(o), _f32p(d), _f32p(out))
    return out


def make_uniforms(viewport, inv_proj, inv_view):
    u = Uniforms()
    u.viewport_extent[0], u.viewport_extent[1] = viewport
    for i in range(16):
        u.inv_proj[i] = float(inv_proj[i])
        u.inv_view[i] = float(inv_view[i])
    return u


def uniforms_bytes(u):
    return bytes(u)


def orbit_uniforms(viewport, target=(0, 0, 0), radius=5.0, events=()):
    """OrbitCameraController::new(target, radius) + events -> prepared Uniforms (renderer.rs:205-222)."""
    L = lib()
    orb = Orbit()
    t = np.ascontiguousarray(np.asarray(target, dtype=np.float32))
    L.rmo_orbit_new(C.byref(orb), _f32p(t), radius)
    for ev, dx, dy in events:
        L.rmo_orbit_update(C.byref(orb), ev, dx, dy)
    pos = np.zeros(3, dtype=np.float32)
    q = np.zeros(4, dtype=np.float32)
    L.rmo_orbit_camera(C.byref(orb), _f32p(pos), _f32p(q))
    u = Uniforms()
    rc = L.rmo_prepare_uniforms(viewport[0], viewport[1], _f32p(pos), _f32p(q), C.byref(u))
    assert rc == 0
    return u, pos, q, orb


def render(u, limits, cmd_count, words, W, H, row0=0, rows=None, threads=1, want_counters=False, materials=None):
    """materials (extension): (n, 3) albedo table for programs with Material commands; None = {(0.4, 0.7, 0.1)}."""
    rows = H - row0 if rows is None else rows
    w = _words(words)
    lim = Limits(*limits)
    out = np.empty((rows, W, 4), dtype=np.float32)
    cnt = Counters()
    nw = len(np.asarray(words))
    if materials is not None:
        m = np.ascontiguousarray(np.asarray(materials, dtype=np.float32).reshape(-1, 3))
        rc = lib().rmo_render_mt_materials(C.byref(u), C.byref(lim), cmd_count, _u32p(w), nw, W, H, row0, rows,
                                           _f32p(out), C.byref(cnt), max(1, threads), _f32p(m), len(m))
    elif threads > 1:
        rc = lib().rmo_render_mt(C.byref(u), C.byref(lim), cmd_count, _u32p(w), nw, W, H, row0, rows,
                                 _f32p(out), C.byref(cnt), threads)
    else:
        rc = lib().rmo_render(C.byref(u), C.byref(lim), cmd_count, _u32p(w), nw, W, H, row0, rows,
                              _f32p(out), C.byref(cnt))
    if rc != 0:
        raise ValueError("oracle rejected program: rc=%d" % rc)
    return (out, cnt.as_dict()) if want_counters else out


def quantize_unorm8(img, bgra=False):
    """RGBA32F (..., 4) -> uint8 (..., 4) as an 8-bit UNORM colour target stores it (clamp, * 255, round to nearest even)."""
    a = np.ascontiguousarray(np.asarray(img, dtype=np.float32))
    out = np.empty(a.shape, dtype=np.uint8)
    lib().rmo_quantize_unorm8(_f32p(a), a.size // 4, int(bool(bgra)), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out
