"""Deterministic synthetic scenes of SURVEY.md 8(d), written independently of the product's
scene generator (ray-marching_amd/csrc/host/scenes.hpp) so that the two can be compared
word for word.  A scene is (nodes, root) with nodes = [(kind, params, lhs, rhs), ...] in the
flat-table form oracle.cbind.serialize() takes."""
SPHERE, BOX, UNION, SUBTRACTION = 0, 1, 100, 101
PLANE, CYLINDER, INTERSECTION, SMOOTH_UNION = 2, 10, 102, 110    # extension node types
TRANSLATION, ROTATION, SCALE = 200, 202, 204                     # space transformations (push opcode; child = lhs)
MATERIAL = 300                                                   # material tag (extension; child = lhs, params = [index])


def _f32(x):
    import struct
    return struct.unpack("<f", struct.pack("<f", x))[0]


class _Tab:
    def __init__(self):
        self.nodes = []

    def sphere(self, c, r):
        self.nodes.append((SPHERE, [c[0], c[1], c[2], r], -1, -1))
        return len(self.nodes) - 1

    def box(self, c, r):
        self.nodes.append((BOX, [c[0], c[1], c[2], r[0], r[1], r[2]], -1, -1))
        return len(self.nodes) - 1

    def op(self, kind, a, b):
        self.nodes.append((kind, [], a, b))
        return len(self.nodes) - 1

    # extension node types
    def plane(self, n, h):
        self.nodes.append((PLANE, [n[0], n[1], n[2], h], -1, -1))
        return len(self.nodes) - 1

    def cylinder(self, c, r, half_h):
        self.nodes.append((CYLINDER, [c[0], c[1], c[2], r, half_h], -1, -1))
        return len(self.nodes) - 1

    def translation(self, child, offset):
        self.nodes.append((TRANSLATION, [offset[0], offset[1], offset[2]], child, -1))
        return len(self.nodes) - 1

    def rotation(self, child, q):
        self.nodes.append((ROTATION, [q[0], q[1], q[2], q[3]], child, -1))
        return len(self.nodes) - 1

    def scale(self, child, s):
        self.nodes.append((SCALE, [s], child, -1))
        return len(self.nodes) - 1

    def smooth_union(self, a, b, k):
        self.nodes.append((SMOOTH_UNION, [k], a, b))
        return len(self.nodes) - 1

    def material(self, child, index):
        self.nodes.append((MATERIAL, [index], child, -1))
        return len(self.nodes) - 1


def g1():
    t = _Tab()
    return t.nodes, t.sphere((0, 0, 0), 1.0)


def g8():
    t = _Tab()
    s0 = t.sphere((0, 0, 0), 1.0)
    b1 = t.box((0, 0, 0), (0.8, 0.8, 0.8))
    s2 = t.sphere((0.9, 0.5, 0.6), 0.6)
    b3 = t.box((0, -1.2, 0), (1.5, 0.1, 1.5))
    root = t.op(UNION, t.op(SUBTRACTION, t.op(UNION, s0, b1), s2), b3)
    return t.nodes, root


class _LCG:
    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFF

    def u(self):
        self.x = (1664525 * self.x + 1013904223) & 0xFFFFFFFF
        return (self.x >> 8) / 16777216.0


def _grid_prims(t, nx, nz, seed):
    rng = _LCG(seed)
    prims = []
    for k in range(nx * nz):
        ix, iz = k % nx, k // nx
        x = (ix - (nx - 1) / 2.0) * 1.1
        z = (iz - (nz - 1) / 2.0) * 1.1
        u1 = rng.u()
        u2 = rng.u()
        y = 0.3 * u1
        if (ix + iz) % 2 == 0:
            prims.append(t.sphere((x, y, z), 0.35 + 0.15 * u2))
        else:
            h = 0.3 + 0.15 * u2
            prims.append(t.box((x, y, z), (h, h, h)))
    return prims


def _fold_left(t, prims):
    acc = prims[0]
    for k in range(len(prims) - 1):
        acc = t.op(SUBTRACTION if k % 4 == 3 else UNION, acc, prims[k + 1])
    return acc


def g32():
    t = _Tab()
    return t.nodes, _fold_left(t, _grid_prims(t, 4, 4, 0x5DF00020))


def g64():
    t = _Tab()
    return t.nodes, _fold_left(t, _grid_prims(t, 8, 4, 0x5DF00040))


def g32_balanced():
    """Same primitives as g32, combined by a balanced binary tree (stack depth 5)."""
    t = _Tab()
    level = _grid_prims(t, 4, 4, 0x5DF00020)
    k = 0
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), 2):
            nxt.append(t.op(SUBTRACTION if k % 4 == 3 else UNION, level[i], level[i + 1]))
            k += 1
        level = nxt
    return t.nodes, level[0]


def right_deep(n):
    """Union(S0, Union(S1, Union(S2, ...))) -- needs a value stack n deep."""
    t = _Tab()
    prims = [t.sphere((0.7 * (i - (n - 1) / 2.0), 0.1 * (i % 3), 0.0), 0.3) for i in range(n)]
    acc = prims[-1]
    for i in range(n - 2, -1, -1):
        acc = t.op(UNION if i % 3 else SUBTRACTION, prims[i], acc)
    return t.nodes, acc


def g8x():
    """BASELINE config 2 as literally worded: sphere U box - cylinder (+ floor slab).  Extension."""
    t = _Tab()
    s0 = t.sphere((0, 0, 0), 1.0)
    b1 = t.box((0, 0, 0), (0.8, 0.8, 0.8))
    c2 = t.cylinder((0.9, 0.5, 0.6), 0.45, 0.9)
    b3 = t.box((0, -1.2, 0), (1.5, 0.1, 1.5))
    root = t.op(UNION, t.op(SUBTRACTION, t.op(UNION, s0, b1), c2), b3)
    return t.nodes, root


def g32s():
    """BASELINE config 3 as literally worded: G32 with Union -> SmoothUnion(k = 0.25).  Extension."""
    t = _Tab()
    prims = _grid_prims(t, 4, 4, 0x5DF00020)
    acc = prims[0]
    for k in range(len(prims) - 1):
        acc = t.op(SUBTRACTION, acc, prims[k + 1]) if k % 4 == 3 else t.smooth_union(acc, prims[k + 1], 0.25)
    return t.nodes, acc


def ext_mix():
    """Every extension node type in one tree: plane-cut, intersection, cylinder, smooth blends."""
    t = _Tab()
    a = t.smooth_union(t.sphere((-0.6, 0, 0), 0.7), t.cylinder((0.5, 0.0, 0.1), 0.4, 0.8), 0.3)
    b = t.op(INTERSECTION, t.box((0, 0, 0), (1.4, 0.9, 1.0)), t.plane((0.0, 1.0, 0.2), 0.35))
    c = t.smooth_union(a, t.op(SUBTRACTION, b, t.sphere((0.2, 0.3, 0.9), 0.5)), 0.15)
    return t.nodes, t.op(UNION, c, t.cylinder((-1.4, -0.6, -0.5), 0.25, 0.5))


def xform_mix():
    """Space transformations, nested, around primitives and around a sub-tree."""
    t = _Tab()
    h = 0.70710678
    a = t.translation(t.rotation(t.box((0, 0, 0), (0.9, 0.35, 0.5)), (h, 0, 0, h)), (-1.1, 0.2, 0.0))
    b = t.scale(t.op(UNION, t.sphere((0, 0, 0), 1.0), t.box((0.9, 0, 0), (0.5, 0.3, 0.3))), 0.6)
    c = t.translation(t.rotation(t.scale(t.op(SUBTRACTION, t.box((0, 0, 0), (1, 1, 1)), t.sphere((0.4, 0.4, 0.4), 0.9)), 0.45),
                                 (0.9238795, 0.2209424, 0.2209424, 0.2209424)), (1.2, -0.3, 0.4))
    d = t.rotation(t.cylinder((0.0, -0.9, -0.9), 0.3, 0.7), (h, h, 0, 0))
    return t.nodes, t.op(UNION, t.op(UNION, t.op(UNION, a, b), c), d)


MATERIAL_TABLE = [(0.4, 0.7, 0.1), (0.9, 0.15, 0.1), (0.1, 0.3, 0.9), (0.95, 0.9, 0.2), (0.8, 0.8, 0.8), (0.6, 0.1, 0.7)]


def mat_mix():
    """Material tags (extension) on leaves, on a sub-tree, inside transform scopes, under every operator: the carved
    surfaces of a subtraction show the subtractor's material, a blend switches where the operands cross."""
    t = _Tab()
    h = 0.70710678
    body = t.op(UNION, t.material(t.sphere((0, 0, 0), 1.0), 1), t.material(t.box((0, 0, 0), (0.8, 0.8, 0.8)), 2))
    carved = t.op(SUBTRACTION, body, t.material(t.sphere((0.9, 0.5, 0.6), 0.6), 3))
    slab = t.box((0, -1.2, 0), (1.5, 0.1, 1.5))                                    # untagged: material 0
    arm = t.translation(t.rotation(t.material(t.cylinder((0, 0, 0), 0.25, 0.9), 5), (h, 0, 0, h)), (-1.3, 0.4, 0.3))
    blob = t.material(t.smooth_union(t.material(t.sphere((1.3, 0.2, -0.6), 0.45), 1), t.sphere((1.7, 0.5, -0.3), 0.35), 0.3), 4)
    cut = t.op(INTERSECTION, t.material(t.box((-0.2, 1.3, -0.9), (0.5, 0.5, 0.5)), 2), t.material(t.sphere((-0.2, 1.3, -0.9), 0.62), 3))
    twin = t.scale(t.smooth_union(t.material(t.sphere((-2.0, -0.6, 1.6), 0.5), 3), t.material(t.box((-1.2, -0.6, 1.6), (0.4, 0.4, 0.4)), 5), 0.4), 0.8)
    root = t.op(UNION, t.op(UNION, t.op(UNION, t.op(UNION, t.op(UNION, carved, slab), arm), blob), cut), twin)
    return t.nodes, root


SCENES = {"g1": g1, "g8": g8, "g32": g32, "g64": g64, "g32_balanced": g32_balanced}
EXT_SCENES = {"g8x": g8x, "g32s": g32s, "ext_mix": ext_mix, "xform_mix": xform_mix}
MAT_SCENES = {"mat_mix": mat_mix}    # need MATERIAL_TABLE

# (events for OrbitCameraController::update) still camera of SURVEY 8(d): Orbit([35,-25])
STILL_CAMERA_EVENTS = [(1, 35.0, -25.0)]
LIMITS = {"g1": (0.01, 100.0, 64), "g8": (0.01, 100.0, 128), "g32": (0.01, 100.0, 256),
          "g64": (0.01, 100.0, 512), "g32_balanced": (0.01, 100.0, 256)}
