"""Material tags (opcode 300) and the material table: an extension -- the reference shades every hit with
(0.4, 0.7, 0.1) (wgsl:105) and lists a material system as future work (README.md:11).  The semantics are this repo's
(oracle/rm_oracle.c map_scene_impl), so parity with the reference is undefined; the oracle is pinned by hand-derived
values and by C == numpy, the host mirrors by word-for-word serialisation.  CPU only."""
import re

import numpy as np
import pytest

import scenes
from oracle import rm_oracle_np as onp
from ray_marching_amd import _ffi, csg, renderer

F = np.float32


def test_tag_serialisation_and_host_mirrors(oracle):
    t = scenes._Tab()
    cc, w = oracle.serialize(t.nodes, t.material(t.sphere((0, 0, 0), 1.0), 7))
    assert cc == 2 and [int(x) for x in w[-2:]] == [300, 7]          # the index is a plain integer word
    hcc, hw = csg.serialize(csg.Material(csg.Sphere((0, 0, 0), 1.0), 7))
    assert (hcc, hw.tobytes()) == (cc, w.tobytes())
    cc, w = oracle.serialize(*scenes.mat_mix())
    hcc, hw = csg.serialize(csg.scene("mat_mix"))                    # C++ mirror emits the same words
    assert (hcc, hw.tobytes()) == (cc, w.tobytes())
    assert renderer.validate_program(cc, w) == oracle.validate(cc, w, strict=True)


def test_which_operand_decides(oracle):
    t = scenes._Tab()
    a = t.material(t.sphere((0, 0, 0), 1.0), 1)
    b = t.material(t.sphere((3, 0, 0), 1.0), 2)
    cc, w = oracle.serialize(t.nodes, t.op(scenes.UNION, a, b))
    assert oracle.map_scene_material(cc, w, [0.5, 0, 0]) == 1          # nearer to a
    assert oracle.map_scene_material(cc, w, [2.5, 0, 0]) == 2          # nearer to b
    assert oracle.map_scene_material(cc, w, [1.5, 0, 0]) == 1          # tie: the operand pushed first (a)
    assert oracle.map_scene(cc, w, [4.5, 0, 0]) == 0.5                 # distances ignore the tags
    t = scenes._Tab()
    body = t.material(t.box((0, 0, 0), (1, 1, 1)), 1)
    hole = t.material(t.sphere((1, 0, 0), 0.5), 2)
    cc, w = oracle.serialize(t.nodes, t.op(scenes.SUBTRACTION, body, hole))
    assert oracle.map_scene_material(cc, w, [-1.2, 0, 0]) == 1         # outer face: a decides (max(a, -b) = a)
    assert oracle.map_scene_material(cc, w, [0.6, 0, 0]) == 2          # the carved surface shows the subtractor
    t = scenes._Tab()
    cc, w = oracle.serialize(t.nodes, t.op(scenes.INTERSECTION, t.material(t.box((0, 0, 0), (1, 1, 1)), 1),
                                           t.material(t.sphere((0, 0, 0), 1.2), 2)))
    assert oracle.map_scene_material(cc, w, [1.1, 0, 0]) == 1          # face of the box: 0.1 > -0.1
    assert oracle.map_scene_material(cc, w, [0.9, 0.9, 0.0]) == 2      # rounded corner: the sphere decides
    t = scenes._Tab()
    inner = t.op(scenes.UNION, t.material(t.sphere((0, 0, 0), 1.0), 1), t.sphere((3, 0, 0), 1.0))
    cc, w = oracle.serialize(t.nodes, t.material(inner, 4))            # an outer tag repaints the whole sub-tree
    assert oracle.map_scene_material(cc, w, [0.5, 0, 0]) == 4 and oracle.map_scene_material(cc, w, [2.5, 0, 0]) == 4
    t = scenes._Tab()
    cc, w = oracle.serialize(t.nodes, t.op(scenes.UNION, t.sphere((0, 0, 0), 1.0), t.material(t.sphere((3, 0, 0), 1.0), 2)))
    assert oracle.map_scene_material(cc, w, [0.5, 0, 0]) == 0          # untagged primitives carry material 0
    t = scenes._Tab()
    moved = t.translation(t.material(t.sphere((0, 0, 0), 1.0), 3), (5, 0, 0))
    cc, w = oracle.serialize(t.nodes, t.op(scenes.UNION, t.sphere((0, 0, 0), 1.0), moved))
    assert oracle.map_scene_material(cc, w, [4.5, 0, 0]) == 3 and oracle.map_scene_material(cc, w, [0.5, 0, 0]) == 0


def test_validation(oracle):
    f = lambda x: int(np.float32(x).view(np.uint32))
    sphere = [0, f(0), f(0), f(0), f(1)]
    cases = {
        "tag with nothing on the stack": (1, [300, 1], _ffi.RM_ERR_STACK_UNDERFLOW),
        "tag without its index": (2, sphere + [300], _ffi.RM_ERR_TRUNCATED),
        "index beyond the table's capacity": (2, sphere + [300, 256], _ffi.RM_ERR_MATERIAL),
        "tag inside a scope that has produced nothing yet": (5, sphere + [200, f(1), f(0), f(0), 300, 1] + sphere + [201, 100],
                                                             _ffi.RM_ERR_STACK_UNDERFLOW),
    }
    for name, (cc, words, want) in cases.items():
        w = np.array(words, dtype=np.uint32)
        assert renderer.validate_program(cc, w)[0] == want, name
        assert oracle.validate(cc, w, strict=True)[0] == want, name
    w = np.array(sphere + [300, 255], dtype=np.uint32)
    assert renderer.validate_program(2, w) == (0, 1) and oracle.validate(2, w, strict=True) == (0, 1)


def test_render_c_equals_numpy_and_needs_its_table(oracle):
    cc, w = oracle.serialize(*scenes.mat_mix())
    W, H = 40, 30
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 96)
    a = oracle.render(u, lim, cc, w, W, H, threads=4, materials=scenes.MATERIAL_TABLE)
    b = onp.render({"viewport_extent": list(u.viewport_extent), "inv_proj": list(u.inv_proj), "inv_view": list(u.inv_view)},
                   lim, cc, w, W, H, materials=scenes.MATERIAL_TABLE)
    assert a.tobytes() == b.tobytes()
    grey = oracle.render(u, lim, cc, w, W, H, threads=4, materials=[(0.4, 0.7, 0.1)] * 6)
    assert grey.tobytes() != a.tobytes()
    # every entry the reference colour: the image of the same tree without its tags (tags never move a surface)
    plain_w = []
    i = 0
    n_par = {0: 4, 1: 6, 2: 4, 10: 5, 110: 1, 200: 3, 202: 4, 204: 1, 300: 1}
    n_cmd = 0
    while i < len(w):
        op = int(w[i])
        n = 1 + n_par.get(op, 0)
        if op != 300:
            plain_w += [int(x) for x in w[i:i + n]]
            n_cmd += 1
        i += n
    plain = oracle.render(u, lim, n_cmd, np.array(plain_w, np.uint32), W, H, threads=4)
    assert plain.tobytes() == grey.tobytes()
    with pytest.raises(ValueError):                                     # index 5 with the one-entry default table
        oracle.render(u, lim, cc, w, W, H, threads=4)
    # a table changes nothing for a program without tags
    cc8, w8 = oracle.serialize(*scenes.g8())
    assert oracle.render(u, lim, cc8, w8, W, H, threads=4).tobytes() == \
        oracle.render(u, lim, cc8, w8, W, H, threads=4, materials=scenes.MATERIAL_TABLE).tobytes()


def test_generated_kernel_of_a_tagged_program():
    """A tagged program keeps the structure (and the map_scene code) of the untagged one; its kernel is compiled with
    the material phase.  hipRTC cross-compiles without a GPU."""
    from oracle import cbind
    cc, w = cbind.serialize(*scenes.mat_mix())
    src = renderer.jit_source(cc, w)
    assert "false, true, true>(L, work" in src
    cc8, w8 = cbind.serialize(*scenes.g8())
    assert "false, true, false>(L, work" in renderer.jit_source(cc8, w8)
    t = scenes._Tab()
    nodes, root = scenes.g8()
    tagged_nodes = list(nodes)
    tagged_nodes.append((scenes.MATERIAL, [3], root, -1))
    cct, wt = cbind.serialize(tagged_nodes, len(tagged_nodes) - 1)
    # the distance code (march function and four-tap function) is the untagged scene's, text for text ...
    def body(s):
        end = s.index("RM_DEV uint32_t map_scene_material_spec") if "map_scene_material_spec(const" in s else s.index('extern "C"')
        text = s[s.index("map_scene_spec"):end].replace("namespace rmk {\n", "").replace("template <bool FAST>\n", "").rstrip()
        return re.sub(r"\n__attribute__\(\(amdgpu_waves_per_eu\(\d, \d\)\)\)$", "", text)      # (a chain without materials is compiled for 7 waves per SIMD)
    tagged_src = renderer.jit_source(cct, wt)
    assert body(tagged_src) == body(renderer.jit_source(cc8, w8))
    # ... and the material walk follows it as straight-line code: (distance, index) pairs, the tag read as data
    assert "#define RM_JIT_MATERIAL_WALK 1" in tagged_src
    walk = tagged_src[tagged_src.index("RM_DEV uint32_t map_scene_material_spec"):tagged_src.index('extern "C"')]
    lines = [l.strip() for l in walk.splitlines()]
    assert "const float v1 = vmin(v0, sdf_box_t<FAST>(x0, y0, z0, mp[1].p, tiny)); const uint32_t m1 = v1 < v0 ? 0u : m0;" not in lines
    assert "const float v2 = vmin(v0, v1); const uint32_t m2 = v1 < v0 ? 0u : m0;" in lines        # (S u B): the box decides -> index 0
    assert "const float v4 = vmax_negb(v2, v3); const uint32_t m4 = -v3 > v2 ? 0u : m2;" in lines  # ... - S
    assert any(l.startswith("const float v7 = v6; const uint32_t m7 = __float_as_uint(mp[4].p[0]);") for l in lines)   # the root's tag
    assert lines[-3] == "return m7;"
    assert "#define RM_JIT_MATERIAL_WALK" not in renderer.jit_source(cc8, w8)
    rc, ms, nbytes, log = renderer.jit_compile(cc, w)
    if rc != _ffi.RM_OK and "could not be loaded" in log:
        pytest.skip("libhiprtc is not installed")
    assert rc == _ffi.RM_OK and nbytes > 4096, log
