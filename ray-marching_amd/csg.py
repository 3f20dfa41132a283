"""Python face of the C++ scene model (csrc/host/csg.hpp via include/rm_host.h), keeping the
reference's names: CSGCommandType, CSGCommandBufferBuilder (src/ray_marching/csg/builder.rs),
Sphere / Box / Union / Subtraction and build_commands (csg/mod.rs, primitives/, operations/).
All logic runs in librm_host.so; these classes only hold handles."""
import ctypes as C
import enum

import numpy as np

from . import _ffi


class CSGCommandType(enum.IntEnum):  # builder.rs:3-24
    Sphere = 0
    Box = 1
    Union = 100
    Subtraction = 101
    # extensions (not implemented by the reference)
    Plane = 2
    Cylinder = 10
    Intersection = 102
    SmoothUnion = 110
    TranslationPush, TranslationPop, RotationPush, RotationPop, ScalePush, ScalePop = 200, 201, 202, 203, 204, 205
    Material = 300


def _f3(v):
    a = (C.c_float * 3)(*[float(x) for x in v])
    return a


class CSGCommandBufferBuilder:  # builder.rs:26-62
    def __init__(self):
        self._L = _ffi.host_lib()
        self._h = self._L.rmh_builder_new()

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rmh_builder_free(self._h)
            self._h = None

    def push_command(self, cmd_type):
        self._L.rmh_builder_push_command(self._h, int(cmd_type))
        return self

    def push_param_vec3(self, value):
        self._L.rmh_builder_push_param_vec3(self._h, _f3(value))
        return self

    def push_param_float(self, value):
        self._L.rmh_builder_push_param_float(self._h, float(value))
        return self

    @property
    def cmd_count(self):
        return int(self._L.rmh_builder_cmd_count(self._h))

    @property
    def buffer(self):
        n = self._L.rmh_builder_len(self._h)
        p = self._L.rmh_builder_buffer(self._h)
        return np.array([p[i] for i in range(n)], dtype=np.uint32)


class CSGNode:
    """Owning handle of a C++ CSGNode tree (csg/mod.rs:28-45)."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("null CSGNode handle")
        self._L = _ffi.host_lib()
        self._h = handle

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.rmh_node_free(self._h)
            self._h = None

    def clone(self):
        return CSGNode(self._L.rmh_node_clone(self._h))

    def build_commands(self, builder):
        """BuildCommands::build_commands (csg/mod.rs:16-19)."""
        self._L.rmh_build_commands(self._h, builder._h)


def Sphere(center=(0.0, 0.0, 0.0), radius=1.0):  # sphere.rs:8-13
    return CSGNode(_ffi.host_lib().rmh_sphere(_f3(center), float(radius)))


def Box(center=(0.0, 0.0, 0.0), radius=(1.0, 1.0, 1.0)):  # box.rs:8-12
    return CSGNode(_ffi.host_lib().rmh_box(_f3(center), _f3(radius)))


def Union(lhs, rhs):  # operations/mod.rs:55
    return CSGNode(_ffi.host_lib().rmh_union(lhs._h, rhs._h))


def Subtraction(lhs, rhs):  # operations/mod.rs:56
    return CSGNode(_ffi.host_lib().rmh_subtraction(lhs._h, rhs._h))


# ---- extension node types (no counterpart in the reference; DESIGN.md "Extension node types") ----
def Plane(normal=(0.0, 1.0, 0.0), h=0.0):
    return CSGNode(_ffi.host_lib().rmh_plane(_f3(normal), float(h)))


def Cylinder(center=(0.0, 0.0, 0.0), radius=1.0, half_height=1.0):
    return CSGNode(_ffi.host_lib().rmh_cylinder(_f3(center), float(radius), float(half_height)))


def Intersection(lhs, rhs):
    return CSGNode(_ffi.host_lib().rmh_intersection(lhs._h, rhs._h))


def SmoothUnion(lhs, rhs, k=0.25):
    return CSGNode(_ffi.host_lib().rmh_smooth_union(lhs._h, rhs._h, float(k)))


def Translation(child, offset=(0.0, 0.0, 0.0)):
    """The child moved by `offset` (node type reserved by comment in csg/mod.rs:41; opcodes 200 / 201)."""
    return CSGNode(_ffi.host_lib().rmh_translation(child._h, _f3(offset)))


def Rotation(child, quaternion=(1.0, 0.0, 0.0, 0.0)):
    """The child rotated by the unit quaternion (w, i, j, k) (csg/mod.rs:42; opcodes 202 / 203)."""
    q = (C.c_float * 4)(*[float(x) for x in quaternion])
    return CSGNode(_ffi.host_lib().rmh_rotation(child._h, q))


def Scale(child, factor=1.0):
    """The child scaled uniformly by `factor` > 0 (csg/mod.rs:43; opcodes 204 / 205)."""
    return CSGNode(_ffi.host_lib().rmh_scale(child._h, float(factor)))


def Material(child, index=0):
    """The child with its surfaces tagged by entry `index` of the material table (RayMarchingResources.set_materials).
    The reference has no materials (README.md:11 lists them as future work); opcode 300, one u32 parameter."""
    return CSGNode(_ffi.host_lib().rmh_material(child._h, int(index)))


def scene(name):
    """Named synthetic scene (csrc/host/scenes.hpp): g1, g8, g32, g64, g32_balanced."""
    h = _ffi.host_lib().rmh_scene(name.encode())
    if not h:
        raise KeyError(name)
    return CSGNode(h)


def serialize(node):
    """(cmd_count, words) of `node` (None = empty scene, renderer.rs:224-227)."""
    b = CSGCommandBufferBuilder()
    if node is not None:
        node.build_commands(b)
    return b.cmd_count, b.buffer
