"""Extension node types (Plane, Cylinder, Intersection, SmoothUnion): NOT implemented by the
reference (it only reserves Plane/Intersection by comment); BASELINE.json configs 2-3 name
cylinder and smooth-min.  Their semantics are defined by this repo, so parity with
the reference is undefined; the oracle is pinned by closed-form values and C == numpy.  CPU only."""
import numpy as np
import pytest

import scenes
from oracle import rm_oracle_np as onp
from ray_marching_amd import csg, renderer

F = np.float32


def _words(t, root, oracle):
    return oracle.serialize(t.nodes, root)


def test_plane_and_intersection_values(oracle):
    t = scenes._Tab()
    cc, w = _words(t, t.plane((0.0, 1.0, 0.0), 0.5), oracle)
    assert list(w[:1]) == [2] and cc == 1 and len(w) == 5
    assert oracle.map_scene(cc, w, [3, 2, -7]) == 2.5                       # dot(p, n) + h
    t = scenes._Tab()
    cc, w = _words(t, t.op(scenes.INTERSECTION, t.sphere((0, 0, 0), 1.0), t.plane((0, 1, 0), 0.0)), oracle)
    assert int(w[-1]) == 102
    assert oracle.map_scene(cc, w, [0, 0.25, 0]) == 0.25                    # max(-0.75, 0.25)
    assert oracle.map_scene(cc, w, [0, -3.0, 0]) == 2.0                     # max(2, -3)


def test_cylinder_values(oracle):
    t = scenes._Tab()
    cc, w = _words(t, t.cylinder((0, 0, 0), 1.0, 2.0), oracle)
    assert int(w[0]) == 10 and len(w) == 6
    assert oracle.map_scene(cc, w, [3, 0, 0]) == 2.0                        # radial
    assert oracle.map_scene(cc, w, [0, 5, 0]) == 3.0                        # above the cap
    assert oracle.map_scene(cc, w, [0, 0, 0]) == -1.0                       # inside: max(-1, -2)
    assert oracle.map_scene(cc, w, [4, 6, 0]) == 5.0                        # rim: hypot(3, 4)


def test_smooth_union_values(oracle):
    t = scenes._Tab()
    a, b = t.sphere((0, 0, 0), 1.0), t.sphere((0, 0, 3), 1.0)              # at (0,0,1.5): both 0.5
    cc, w = _words(t, t.smooth_union(a, b, 1.0), oracle)
    assert int(w[-2]) == 110 and w[-1:].view(F)[0] == 1.0
    assert oracle.map_scene(cc, w, [0, 0, 1.5]) == 0.25                     # 0.5 - 1*1*1/4
    assert oracle.map_scene(cc, w, [0, 0, 7]) == 3.0                        # |a-b| >= k: plain min
    cc, w = _words(t, t.smooth_union(a, b, 0.0), oracle)                    # k <= 0: plain min
    assert oracle.map_scene(cc, w, [0, 0, 1.5]) == 0.5


@pytest.mark.parametrize("name", sorted(scenes.EXT_SCENES))
def test_extension_scenes_c_equals_numpy_and_host_mirror(oracle, name):
    cc, w = oracle.serialize(*scenes.EXT_SCENES[name]())
    hcc, hw = csg.serialize(csg.scene(name))
    assert hcc == cc and hw.tobytes() == w.tobytes()                        # C++ mirror emits the same words
    assert renderer.validate_program(cc, w) == oracle.validate(cc, w, strict=True)
    u, *_ = oracle.orbit_uniforms((36.0, 28.0), events=scenes.STILL_CAMERA_EVENTS)
    a = oracle.render(u, (0.01, 100.0, 96), cc, w, 36, 28)
    b = onp.render({"viewport_extent": list(u.viewport_extent), "inv_proj": list(u.inv_proj),
                    "inv_view": list(u.inv_view)}, (0.01, 100.0, 96), cc, w, 36, 28)
    assert a.tobytes() == b.tobytes()


def test_python_constructors_serialize():
    n = csg.SmoothUnion(csg.Intersection(csg.Cylinder((0, 0, 0), 0.5, 1.0), csg.Plane((0, 1, 0), 0.2)),
                        csg.Sphere((1, 0, 0), 0.5), 0.125)
    cc, w = csg.serialize(n)
    assert cc == 5
    assert [int(w[0]), int(w[6]), int(w[11]), int(w[12]), int(w[17])] == [10, 2, 102, 0, 110]
    assert w[18:19].view(F)[0] == F(0.125)


def test_truncated_extension_commands_are_rejected(oracle):
    for cc, words in [(1, [10, 0, 0, 0, 0]), (1, [2, 0, 0]), (3, [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 110]), (1, [110, 0])]:
        rc, _ = renderer.validate_program(cc, words)
        assert rc < 0 and rc == oracle.validate(cc, words, strict=True)[0]
