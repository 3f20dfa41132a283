"""Parity of the HIP path with the oracle -- the tests proper.  Everything goes through the
C ABI (include/rm_abi.h via ctypes); the bar is BIT-EXACT equality of the RGBA32F image
(the stated contract is <= 1e-4 per channel; because hit/miss and the checkerboard are
discontinuous the only robust way to meet it is to execute the same IEEE op sequence)."""
import ctypes as C
import hashlib

import os

import numpy as np
import pytest

import golden_util as G
import scenes
from ray_marching_amd import _ffi, camera, csg, renderer

pytestmark = pytest.mark.gpu

TOLERANCE = 1e-4      # contract (BASELINE.json north_star); asserted bit-exact below
KERNELS = [_ffi.RM_KERNEL_PIXEL, _ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS]
# pseudo variant: the default kernel with its structure-specialised (hipRTC) march kernel, compiled synchronously;
# every explicitly named variant above runs with specialisation off, i.e. the interpreter kernels
KERNEL_SPEC = 1000
KERNEL_SPEC_PRUNE = 1001     # ... plus far-primitive pruning (RM_OPT_PRUNE = 1; opt-in)
KERNELS += [KERNEL_SPEC, KERNEL_SPEC_PRUNE]
KERNEL_IDS = ["pixel", "v5", "v5_lds", "v5_spec", "v5_spec_prune"]
IDX = G.index()


@pytest.fixture(scope="module")
def res():
    r = renderer.RayMarchingResources(0)
    yield r
    r.close()


def assert_same(img, ref):
    assert img.shape == ref.shape
    if img.tobytes() != ref.tobytes():
        d = np.abs(img.astype(np.float64) - ref.astype(np.float64))
        bad = np.argwhere(d.max(axis=-1) > 0)
        raise AssertionError("images differ: max abs diff %.3g (contract %.0e), %d/%d pixels, first at %s"
                             % (np.nanmax(d), TOLERANCE, len(bad), d.shape[0] * d.shape[1], bad[:4].tolist()))


def setup(res, e_or_none=None, *, cc=None, words=None, u=None, limits=None, kernel=_ffi.RM_KERNEL_DEFAULT, materials=None):
    spec = kernel in (KERNEL_SPEC, KERNEL_SPEC_PRUNE, _ffi.RM_KERNEL_DEFAULT)
    res.set_option(_ffi.RM_OPT_SPECIALIZE, 2 if spec else 0)
    res.set_option(_ffi.RM_OPT_PRUNE, 1 if kernel == KERNEL_SPEC_PRUNE else 0)
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_V5_LDS if kernel in (KERNEL_SPEC, KERNEL_SPEC_PRUNE) else kernel)
    res.set_limits(limits)
    res.set_uniforms(u)
    res.set_materials(materials if materials is not None else [(0.4, 0.7, 0.1)])
    res.set_program(cc, words)


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("name", sorted(n for n in IDX if "file" in IDX[n]))
def test_golden_fixtures(res, name, kernel):
    e = IDX[name]
    if (e["scene"] in scenes.EXT_SCENES or e["scene"] in scenes.MAT_SCENES) and kernel not in (_ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE):
        pytest.skip("only the v5 kernels render extension node types")
    u = _ffi.Uniforms.from_buffer_copy(G.uniforms_bytes(e))
    setup(res, cc=e["cmd_count"], words=G.words(e), u=u, limits=tuple(e["limits"]), kernel=kernel, materials=e.get("materials"))
    assert_same(res.draw(e["W"], e["H"]), G.load_image(e))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("name", sorted(n for n in IDX if "file" not in IDX[n]))
def test_golden_checksums(res, name, kernel):
    e = IDX[name]
    u = _ffi.Uniforms.from_buffer_copy(G.uniforms_bytes(e))
    setup(res, cc=e["cmd_count"], words=G.words(e), u=u, limits=tuple(e["limits"]), kernel=kernel)
    img = res.draw(e["W"], e["H"])
    assert hashlib.sha256(img.tobytes()).hexdigest() == e["sha256"]


def oracle_case(oracle, scene, W, H, limits, events=scenes.STILL_CAMERA_EVENTS):
    cc, w = oracle.serialize(*scene) if scene is not None else (0, np.zeros(0, np.uint32))
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
    return cc, w, u


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("W,H", [(1, 1), (7, 5), (16, 16), (17, 33), (63, 9), (130, 70)])
def test_ragged_sizes_vs_oracle(res, oracle, W, H, kernel):
    cc, w, u = oracle_case(oracle, scenes.g8(), W, H, None)
    lim = (0.01, 100.0, 96)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_limits_edge_cases(res, oracle, kernel):
    W, H = 40, 24
    cc, w, u = oracle_case(oracle, scenes.g8(), W, H, None)
    uu = _ffi.Uniforms.from_buffer_copy(bytes(u))
    for lim in [(0.01, 100.0, 0), (0.01, 100.0, 1), (0.01, 100.0, 3), (0.5, 3.0, 50), (1e-6, 1e6, 40),
                (5.0, 1.0, 10)]:   # max_iter 0; tiny budgets; coarse; min_dist > max_dist
        setup(res, cc=cc, words=w, u=uu, limits=lim, kernel=kernel)
        assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_deep_stack_programs(res, oracle, kernel):
    W, H = 48, 32
    for scene in (scenes.right_deep(6), scenes.right_deep(32), scenes.g32_balanced()):
        cc, w, u = oracle_case(oracle, scene, W, H, None)
        lim = (0.01, 100.0, 64)
        res.resize_command_buffer(4096)
        setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
        assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS], ids=["v5", "v5_lds"])
def test_interpreter_record_loops(res, oracle, kernel):
    """Which record loop the interpreter kernels take (RM_INFO_INTERPRETER_LOOP) and that each renders the oracle's image:
    chains "a op b op c ..." of 1..9 primitives -- the stack-free loop over the records the wave's unit mask names --
    seen from outside, from inside a primitive and from far away;
    other arrangements of reference nodes (a right-deep tree, operators on sub-trees) the tree loop -- from a dozen leaves on over
    the records that are left once operands without a needed leaf are dropped --, extension node types the general loop."""
    rng = np.random.default_rng(11)
    W, H = 56, 40
    lim = (0.01, 100.0, 96)
    res.resize_command_buffer(4096)
    cams = [scenes.STILL_CAMERA_EVENTS, [(2, -30.0, 0.0)] * 4, [(2, 60.0, 0.0), (1, 200.0, 30.0)]]   # Dolly in (among / inside the primitives), far out
    for n in (1, 2, 3, 4, 5, 8, 9):
        t = scenes._Tab()
        leaves = []
        for k in range(n):
            c = rng.uniform(-1.5, 1.5, 3)
            leaves.append(t.sphere(tuple(c), float(rng.uniform(0.3, 0.9))) if rng.random() < 0.5 else
                          t.box(tuple(c), tuple(rng.uniform(0.2, 0.7, 3))))
        acc = leaves[0]
        for k, leaf in enumerate(leaves[1:]):
            acc = t.op(scenes.SUBTRACTION if k % 3 == 2 else scenes.UNION, acc, leaf)
        cc, w = oracle.serialize(t.nodes, acc)
        for events in cams:
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
            # (chains of fewer than a dozen leaves walk every record: the mask would cost more than it saves)
            assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0 and res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == 1, n
    # a chain of 16: over the records the wave's unit mask names, where the unit records are at hand (LDS)
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
    assert res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == (2 if kernel == _ffi.RM_KERNEL_V5_LDS else 1)
    for scene, masked in ((scenes.right_deep(5), False), (scenes.g32_balanced(), kernel == _ffi.RM_KERNEL_V5_LDS)):
        cc, w, u = oracle_case(oracle, scene, W, H, None)
        setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
        assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
        # trees of reference nodes: one dispatch per record (3); a dozen leaves or more, unit records at hand: over the records
        # the wave's unit mask leaves (4)
        assert res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == (4 if masked else 3)
    # ... and anything with an extension node type the general loop
    cc, w, u = oracle_case(oracle, scenes.EXT_SCENES["ext_mix"](), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
    assert res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == 0
    # ... unless it is a chain of blends of eight leaves or more with its unit records at hand: the record machine over the units the
    # wave's mask names (5)
    cc, w, u = oracle_case(oracle, scenes.EXT_SCENES["g32s"](), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
    assert res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == (5 if kernel == _ffi.RM_KERNEL_V5_LDS else 0)
    # a specialised kernel reports 0 as well (the loop is the interpreter's)
    cc, w, u = oracle_case(oracle, scenes.g8(), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=KERNEL_SPEC)
    res.draw(W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1 and res.info(_ffi.RM_INFO_INTERPRETER_LOOP) == 0


def test_masked_tree_loop_of_the_interpreter(res, oracle):
    """Random trees of 12..48 spheres and boxes under Union / Subtraction -- balanced, left-deep, right-deep and mixed, with
    Subtractions whose left operand is a whole sub-tree (a wave that needs only the subtracted leaves forces one of that
    sub-tree's back: rm_kernel_v5.h tree_keep) -- through the interpreter's masked tree loop, from outside, from among the
    primitives and from far away: the oracle's image, bit for bit."""
    rng = np.random.default_rng(2024)
    W, H = 64, 48
    lim = (0.01, 100.0, 128)
    res.resize_command_buffer(8192)
    cams = [scenes.STILL_CAMERA_EVENTS, [(2, -30.0, 0.0)] * 4, [(2, 60.0, 0.0), (1, 200.0, 30.0)]]

    def build(t, n, shape, sub):
        if n == 1:
            c = rng.uniform(-2.5, 2.5, 3)
            return (t.sphere(tuple(c), float(rng.uniform(0.2, 0.8))) if rng.random() < 0.5 else
                    t.box(tuple(c), tuple(rng.uniform(0.15, 0.6, 3))))
        left = n // 2 if shape == 0 else n - 1 if shape == 1 else 1 if shape == 2 else int(rng.integers(1, n))
        a, b = build(t, left, shape, sub), build(t, n - left, shape, sub)
        return t.op(scenes.SUBTRACTION if rng.random() < sub else scenes.UNION, a, b)

    for case in range(10):
        t = scenes._Tab()
        n = int(rng.integers(12, 49))
        if case % 4 == 2:
            n = min(n, 28)      # (right-deep: the reference's value stack holds 32)
        root = build(t, n, case % 4, (0.0, 0.15, 0.35)[case % 3])
        cc, w = oracle.serialize(t.nodes, root)
        for events in cams:
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=_ffi.RM_KERNEL_V5_LDS)
            assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
            assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0
            assert res.info(_ffi.RM_INFO_INTERPRETER_LOOP) in ((2, 4) if case % 4 == 1 else (4,)), (case, n)      # (left-deep: a chain)


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_program_leaving_two_values_returns_top(res, oracle, kernel):
    # "S S" (no operator): the reference returns the top of stack (wgsl:202)
    t = scenes._Tab()
    a, b = t.sphere((0, 0, 0), 1.0), t.sphere((1.5, 0, 0), 0.7)
    _, wa = oracle.serialize(t.nodes, a)
    _, wb = oracle.serialize(t.nodes, b)
    w = np.concatenate([wa, wb])
    W, H = 40, 30
    _, _, u = oracle_case(oracle, None, W, H, None)
    lim = (0.01, 100.0, 64)
    setup(res, cc=2, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    assert_same(res.draw(W, H), oracle.render(u, lim, 2, w, W, H))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_row_bands_and_tiling_invariance(res, oracle, kernel):
    W, H = 96, 80
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    lim = (0.01, 100.0, 128)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    full = res.draw(W, H)
    assert_same(full, oracle.render(u, lim, cc, w, W, H, threads=4))
    for strip in (1, 7, 16, 33):
        parts = [res.draw(W, H, r0, min(strip, H - r0)) for r0 in range(0, H, strip)]
        assert np.concatenate(parts).tobytes() == full.tobytes()
    band = res.draw(W, H, 13, 29)
    assert_same(band, oracle.render(u, lim, cc, w, W, H, row0=13, rows=29))


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_batch_equals_single_draws(res, oracle, kernel):
    W, H = 64, 36
    cc, w = oracle.serialize(*scenes.g8())
    lim = (0.01, 100.0, 64)
    frames, refs = [], []
    for f in range(5):
        u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=[(1, 35.0 + 40.0 * f, -25.0)])
        frames.append(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        refs.append(oracle.render(u, lim, cc, w, W, H, threads=4))
    setup(res, cc=cc, words=w, u=frames[0], limits=lim, kernel=kernel)
    out = res.draw_batch(frames, W, H)
    for f in range(5):
        assert_same(out[f], refs[f])


CULL_CAMERAS = {
    "still": dict(events=scenes.STILL_CAMERA_EVENTS),
    "inside_solid": dict(events=[(2, -95.0, 0.0)]),
    "far": dict(events=[(1, 35.0, -25.0), (2, 300.0, 0.0)]),
    "top_down": dict(events=[(1, 10.0, -150.0)]),
    "from_below": dict(events=[(1, -80.0, 150.0)]),
    "panned_off_axis": dict(events=[(1, 35.0, -25.0), (0, 180.0, 90.0)]),
    "far_from_origin": dict(target=(1000.0, -2000.0, 500.0), events=[(1, 35.0, -25.0)]),
    "grazing_zoom": dict(events=[(1, 3.0, -6.0), (2, -60.0, 0.0), (0, 115.0, 12.0)]),
}


@pytest.mark.parametrize("cam", sorted(CULL_CAMERAS))
@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE],
                         ids=["v5", "v5_lds", "v5_spec", "v5_spec_prune"])
def test_miss_ray_culling_is_exact(res, oracle, cam, kernel):
    """The bounding-cone shortcut must never change a pixel: culling on == culling off == oracle,
    for cameras outside, inside, far from and grazing the scene, and for several min_dist."""
    W, H = 72, 48
    spec = CULL_CAMERAS[cam]
    for scene_name in ("g8", "g32"):
        nodes, root = scenes.SCENES[scene_name]()
        if "target" in spec:   # move the whole scene along with the camera target
            nodes = [(k, ([p[0] + spec["target"][0], p[1] + spec["target"][1], p[2] + spec["target"][2]] + list(p[3:]))
                      if k in (0, 1) else p, l, r) for (k, p, l, r) in nodes]
        cc, w = oracle.serialize(nodes, root)
        u, *_ = oracle.orbit_uniforms((float(W), float(H)), target=spec.get("target", (0, 0, 0)), events=spec["events"])
        uu = _ffi.Uniforms.from_buffer_copy(bytes(u))
        for lim in [(0.01, 100.0, 96), (0.6, 100.0, 40), (2.5, 50.0, 20)]:
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=uu, limits=lim, kernel=kernel)
            res.set_option(_ffi.RM_OPT_CULL, 0)
            off = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_CULL, 1)
            res.set_option(_ffi.RM_OPT_BALANCE, 0)
            on = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_BALANCE, 1)      # heaviest-tile-first order: same pixels
            bal = res.draw(W, H)
            assert_same(off, ref)
            assert_same(on, ref)
            assert_same(bal, ref)


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE],
                         ids=["v5_lds", "v5_spec", "v5_spec_prune"])
def test_culling_with_arbitrary_uniform_matrices(res, oracle, kernel):
    """The uniform block is three opaque blobs to the pipeline: the miss tests must hold for ANY matrices, not
    only the reference camera's.  wgsl:62 normalises a vec4, so whenever pt_world.w != ro.w the ray direction
    is shorter than 1 (inv_view with a non-affine last row; projections with znear != 1); viewport_extent
    need not match the target size (AA samples then leave their pixel); degenerate matrices give NaN rays."""
    W, H = 72, 48
    cc, w = oracle.serialize(*scenes.g32())
    base, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    L = oracle.lib()

    def variant(edit):
        u = type(base).from_buffer_copy(bytes(base))
        edit(u)
        return u

    def proj(znear):
        def edit(u):
            m = np.zeros(16, np.float32)
            L.rmo_perspective_inverse(W / H, 0.7853981633974483, znear, 10000.0, m.ctypes.data_as(C.POINTER(C.c_float)))
            for i in range(16):
                u.inv_proj[i] = float(m[i])
        return edit

    def set_view(idx, val):
        def edit(u):
            u.inv_view[idx] = val
        return edit

    def extent(ex, ey):
        def edit(u):
            u.viewport_extent[0], u.viewport_extent[1] = ex, ey
        return edit

    def zero_proj(u):
        for i in range(16):
            u.inv_proj[i] = 0.0

    cases = {
        "short_dir_a2": variant(set_view(11, 2.0)),       # pt_world.w = 1 - 2: |rd| ~ 0.45, same half-lines
        "short_dir_am": variant(set_view(11, -0.6)),
        "row3_x": variant(set_view(3, 0.8)),               # pt_world.w varies across the screen
        "znear_0.5": variant(proj(0.5)),
        "znear_1.5": variant(proj(1.5)),
        "znear_3": variant(proj(3.0)),
        "extent_small": variant(extent(9.0, 6.0)),         # AA offsets 8x the pixel pitch
        "extent_huge": variant(extent(1e6, 1e6)),          # all 16 samples coincide with the pixel centre
        "extent_negative": variant(extent(-float(W), float(H))),
        "nan_rays": variant(zero_proj),
    }
    for name, u in cases.items():
        uu = _ffi.Uniforms.from_buffer_copy(bytes(u))
        for lim in [(0.01, 100.0, 96), (0.4, 100.0, 48)]:
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=uu, limits=lim, kernel=kernel)
            res.set_option(_ffi.RM_OPT_CULL, 0)
            off = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_CULL, 1)
            on = res.draw(W, H)
            assert np.array_equal(off.view(np.uint32), ref.view(np.uint32)), (name, lim, "cull off")
            assert np.array_equal(on.view(np.uint32), ref.view(np.uint32)), (name, lim, "cull on")


def test_culling_degenerate_primitives_and_empty_scene(res, oracle):
    W, H = 56, 40
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    uu = _ffi.Uniforms.from_buffer_copy(bytes(u))
    t = scenes._Tab()
    weird = [t.sphere((0, 0, 0), -0.5), t.box((1.2, 0, 0), (0.0, 0.0, 0.0)), t.box((-1.2, 0, 0), (-0.3, 0.4, 0.2)),
             t.sphere((0, 1.0, 0), 0.0), t.sphere((0.3, -0.4, 0.2), 1e-3), t.box((0, 0, -1), (1e3, 1e-3, 1e-3))]
    cc, w = oracle.serialize(t.nodes, scenes._fold_left(t, weird))
    res.set_option(_ffi.RM_OPT_CULL, 1)
    for lim in [(0.01, 100.0, 64), (0.7, 100.0, 64)]:
        setup(res, cc=cc, words=w, u=uu, limits=lim)
        assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=4))
    # empty scene: map_scene == max_dist everywhere; with max_dist < min_dist every ray HITS at step 0
    for lim in [(0.01, 100.0, 12), (5.0, 1.0, 12), (5.0, 5.0, 3)]:
        setup(res, cc=0, words=[], u=uu, limits=lim)
        assert_same(res.draw(W, H), oracle.render(u, lim, 0, [], W, H, threads=4))


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_DEFAULT, _ffi.RM_KERNEL_PIXEL, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC],
                         ids=["default", "pixel", "v5_lds", "v5_spec"])
def test_interleaved_strips_reassemble_to_the_frame(res, oracle, kernel):
    """rm_draw_strips: the multi-GPU tiling partition.  Every rank's strips, scattered back,
    must reproduce the single-GPU frame byte for byte (tiling invariance)."""
    from ray_marching_amd import shard
    W, H = 120, 104       # 6.5 strips of 16 rows
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    lim = (0.01, 100.0, 96)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    full = res.draw(W, H)
    assert_same(full, oracle.render(u, lim, cc, w, W, H, threads=4))
    for world, sr in ((1, 16), (2, 16), (3, 16), (8, 16), (2, 8), (5, 8), (8, 8), (3, 24)):
        img = np.zeros_like(full)
        for rank in range(world):
            compact = res.draw_strips(W, H, sr, rank, world)
            assert compact.shape[0] == shard.strip_row_count(H, sr, rank, world)
            shard.scatter_strips(img, compact, H, rank, world, sr)
        assert img.tobytes() == full.tobytes(), (world, sr)
    with pytest.raises(_ffi.RmError):
        res.draw_strips(W, H, 12, 0, 2)          # strip height must be a multiple of 8 (the kernels' tile height)
    assert res.draw_strips(W, H, 32, 7, 8).shape[0] == 0   # more ranks than strips: empty share


EXT_KERNELS = [_ffi.RM_KERNEL_DEFAULT, _ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE]  # older kernels: reference nodes only
EXT_IDS = ["default", "v5", "v5_lds", "v5_spec", "v5_spec_prune"]


@pytest.mark.parametrize("kernel", EXT_KERNELS, ids=EXT_IDS)
@pytest.mark.parametrize("name", sorted(scenes.EXT_SCENES))
def test_extension_node_types_vs_oracle(res, oracle, name, kernel):
    """Plane / Cylinder / Intersection / SmoothUnion (not in the reference; BASELINE configs 2-3):
    bit-exact against the oracle, with miss-ray culling on and off, for several cameras."""
    W, H = 88, 56
    cc, w = oracle.serialize(*scenes.EXT_SCENES[name]())
    for events in (scenes.STILL_CAMERA_EVENTS, [(1, -70.0, 40.0), (2, 60.0, 0.0)], [(2, -93.0, 0.0)]):
        u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
        for lim in [(0.01, 100.0, 128), (0.4, 60.0, 48)]:
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            for cull in (0, 1):
                res.set_option(_ffi.RM_OPT_CULL, cull)
                assert_same(res.draw(W, H), ref)
    res.set_option(_ffi.RM_OPT_CULL, 1)


def test_reference_only_kernels_reject_extension_programs(res, oracle):
    cc, w = oracle.serialize(*scenes.g8x())
    u, *_ = oracle.orbit_uniforms((16.0, 16.0))
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=(0.01, 100.0, 16), kernel=_ffi.RM_KERNEL_PIXEL)
    with pytest.raises(_ffi.RmError) as e:
        res.draw(16, 16)
    assert e.value.status == _ffi.RM_ERR_ARG
    # the v2-v4 variants of ABI version 1 (2..11) are retired: selecting one is an argument error, nothing is launched
    for retired in range(2, 12):
        with pytest.raises(_ffi.RmError) as e:
            res.set_option(_ffi.RM_OPT_KERNEL, retired)
        assert e.value.status == _ffi.RM_ERR_ARG
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)


def test_baseline_config2_and_3_as_worded(res, oracle):
    """BASELINE.json configs[1] (1920x1080, sphere U box - cylinder, 128 steps) and configs[2]
    (32-node graph with smooth-min blends, 256 steps; at 1920x1080 here) on the default kernel:
    sampled row bands bit-exact against the oracle + full-frame sanity."""
    for name, lim in (("g8x", (0.01, 100.0, 128)), ("g32s", (0.01, 100.0, 256))):
        W, H = 1920, 1080
        cc, w = oracle.serialize(*scenes.EXT_SCENES[name]())
        u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
        setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim)
        full = res.draw(W, H)
        assert np.isfinite(full).all() and np.array_equal(full[..., 3], np.ones((H, W), np.float32))
        for r0, rows in [(300, 4), (520, 6), (700, 4)]:
            assert_same(full[r0:r0 + rows], oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=8))


def test_raw_write_buffer_path_and_stale_tail(res, oracle):
    """prepare()'s literal call sequence (renderer.rs:213-239): raw byte writes; words beyond
    the current program are stale and must be ignored (renderer.rs:230-239 never clears)."""
    W, H = 48, 40
    cc32, w32, u = oracle_case(oracle, scenes.g32(), W, H, None)
    cc1, w1 = oracle.serialize(*scenes.g1())
    lim = (0.01, 100.0, 64)
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)
    res.resize_command_buffer(1024)
    res.write_buffer(_ffi.RM_BUF_LIMITS, 0, bytes(_ffi.Limits(*lim)))
    res.write_buffer(_ffi.RM_BUF_UNIFORMS, 0, bytes(u))
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 0, np.uint32(cc32).tobytes())
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 4, w32.tobytes())
    assert_same(res.draw(W, H), oracle.render(u, lim, cc32, w32, W, H, threads=4))
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 0, np.uint32(cc1).tobytes())     # g32's tail stays behind
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 4, w1.tobytes())
    assert_same(res.draw(W, H), oracle.render(u, lim, cc1, w1, W, H, threads=4))
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 0, np.uint32(0).tobytes())       # csg_node == None
    assert_same(res.draw(W, H), oracle.render(u, lim, 0, [], W, H, threads=4))


def test_callback_prepare_paint_dropin(res, oracle):
    """RayMarchingCallback::new(time, node, viewport, camera).prepare()/paint() end to end,
    host mirror included, against the oracle's restatement of the same host code."""
    W, H = 80, 60
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)
    res.set_limits(renderer.RayMarchLimits())               # reference defaults {0.01, 100, 100}
    ctl = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)
    ctl.update(camera.Orbit([35.0, -25.0]))
    cb = renderer.RayMarchingCallback.new(0.0, csg.scene("g8"), [float(W), float(H)], ctl.camera())
    cb.prepare(res)
    img = cb.paint(res)
    cc, w, u = oracle_case(oracle, scenes.g8(), W, H, None)
    assert_same(img, oracle.render(u, (0.01, 100.0, 100), cc, w, W, H, threads=4))
    none_cb = renderer.RayMarchingCallback.new(0.0, None, [float(W), float(H)], ctl.camera())
    none_cb.prepare(res)
    assert_same(none_cb.paint(res), oracle.render(u, (0.01, 100.0, 100), 0, [], W, H, threads=4))


def test_abi_errors_on_device(res, oracle):
    L = _ffi.hip_lib()
    res.resize_command_buffer(1024)
    with pytest.raises(_ffi.RmError) as e:
        res.set_program(1, [0, 0, 0])
    assert e.value.status == _ffi.RM_ERR_TRUNCATED
    with pytest.raises(_ffi.RmError) as e:
        res.set_program(1, [9])
    assert e.value.status == _ffi.RM_ERR_OPCODE and "opcode" in str(e.value)
    cc, w = oracle.serialize(*scenes.g64())
    res.set_program(cc, w)                                     # 223 words fit the reference's 255
    t = scenes._Tab()
    prims = [t.box((i, 0, 0), (0.4, 0.4, 0.4)) for i in range(40)]
    cc, w = oracle.serialize(t.nodes, scenes._fold_left(t, prims))   # 319 words > 255
    with pytest.raises(_ffi.RmError) as e:
        res.set_program(cc, w)
    assert e.value.status == _ffi.RM_ERR_TOO_LARGE
    res.resize_command_buffer(2048)
    res.set_program(cc, w)
    # raw garbage is caught at draw time, not executed
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 0, np.uint32(3).tobytes())
    res.write_buffer(_ffi.RM_BUF_COMMANDS, 4, np.array([100, 100, 100], np.uint32).tobytes())
    with pytest.raises(_ffi.RmError) as e:
        res.draw(8, 8)
    assert e.value.status == _ffi.RM_ERR_STACK_UNDERFLOW
    with pytest.raises(_ffi.RmError) as e:
        res.write_buffer(_ffi.RM_BUF_UNIFORMS, 140, b"\0" * 8)
    assert e.value.status == _ffi.RM_ERR_TOO_LARGE
    res.set_program(0, [])
    for args in [(0, 8, 0, 8), (8, 8, 8, 1), (8, 8, 4, 5), (8, 8, 0, 0)]:
        with pytest.raises(_ffi.RmError) as e:
            res.draw(args[0], args[1], args[2], args[3])
        assert e.value.status == _ffi.RM_ERR_RANGE
    assert L.rm_draw(res._h, 8, 8, 0, 8, None, 0, None) == _ffi.RM_ERR_NULL


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
def test_metric_config_full_size_properties(res, oracle, kernel):
    """BASELINE metric config: 1920x1080, G32, 256 steps.  Size-independent properties plus a
    bit-exact comparison of sampled row bands against the oracle."""
    W, H = 1920, 1080
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    lim = (0.01, 100.0, 256)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
    full = res.draw(W, H)
    assert np.isfinite(full).all()
    assert np.array_equal(full[..., 3], np.ones((H, W), np.float32))          # alpha = 1 (wgsl:75)
    assert full[..., :3].min() >= 0.0 and full[..., :3].max() <= 1.0
    # tiling invariance: 8 interleaved 16-row strip sets (the multi-GPU partition) == full frame
    strips = np.concatenate([res.draw(W, H, r0, min(16, H - r0)) for r0 in range(0, H, 16)])
    assert hashlib.sha256(strips.tobytes()).digest() == hashlib.sha256(full.tobytes()).digest()
    for r0, rows in [(0, 4), (403, 6), (536, 8), (777, 5), (1076, 4)]:
        assert_same(full[r0:r0 + rows], oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=8))


def test_metric_config_full_frame_vs_oracle(res, oracle):
    """The whole 1920x1080 / G32 / 256-step frame, every pixel, default kernel vs oracle."""
    W, H = 1920, 1080
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    lim = (0.01, 100.0, 256)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim)
    img = res.draw(W, H)
    ref = oracle.render(u, lim, cc, w, W, H, threads=16)
    assert_same(img, ref)


def test_g64_512_steps_quarter_res(res, oracle):
    """BASELINE config 4's scene/limits (G64, 512 steps) at 960x540."""
    W, H = 960, 540
    cc, w, u = oracle_case(oracle, scenes.g64(), W, H, None)
    lim = (0.01, 100.0, 512)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim)
    assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=16))


def test_device_output_and_stream(res, oracle):
    torch = pytest.importorskip("torch")
    W, H = 128, 72
    cc, w, u = oracle_case(oracle, scenes.g8(), W, H, None)
    lim = (0.01, 100.0, 64)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim)
    ref = oracle.render(u, lim, cc, w, W, H, threads=4)
    out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e0.record(s)
        res.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)
        e1.record(s)
    s.synchronize()
    assert e0.elapsed_time(e1) > 0.0                # the events bracket a real kernel on that stream
    assert_same(out.cpu().numpy(), ref)
    out.zero_()
    torch.cuda.synchronize()
    res.draw_device(W, H, out.data_ptr())           # stream NULL = HIP's null stream
    res.sync()
    assert_same(out.cpu().numpy(), ref)


def test_write_bandwidth_calibration(res):
    gbps = res.measure_write_bandwidth(1 << 28, 5)
    assert 200.0 < gbps < 9000.0


def test_two_contexts_from_two_threads(oracle):
    """Different contexts may be used concurrently from different threads (rm_abi.h threading note)."""
    import threading
    W, H = 96, 64
    jobs = [("g8", (0.01, 100.0, 64)), ("g32", (0.01, 100.0, 96))]
    out, err = {}, []

    def work(i):
        try:
            name, lim = jobs[i]
            r = renderer.RayMarchingResources(0)
            cc, w = oracle.serialize(*scenes.SCENES[name]())
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
            r.set_limits(lim)
            r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
            r.set_program(cc, w)
            imgs = [r.draw(W, H) for _ in range(6)]
            out[i] = (imgs, oracle.render(u, lim, cc, w, W, H, threads=2))
            r.close()
        except Exception as e:   # noqa: BLE001
            err.append(e)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    for i in range(2):
        imgs, ref = out[i]
        for img in imgs:
            assert_same(img, ref)


def test_host_output_rate_is_reported(res, oracle):
    """PCIe-inclusive path (host destination): correct, and its rate is printed for DESIGN.md."""
    import time
    W, H = 1920, 1080
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=(0.01, 100.0, 256))
    res.draw(W, H)
    t0 = time.perf_counter()
    for _ in range(5):
        img = res.draw(W, H)
    dt = (time.perf_counter() - t0) / 5
    print("host-destination draw 1920x1080: %.2f ms = %.0f Mpixels/s (PCIe + pageable-host copy included)" % (dt * 1e3, W * H / dt / 1e6))
    assert img.shape == (H, W, 4)


@pytest.mark.parametrize("fmt,bgra", [(_ffi.RM_FORMAT_RGBA8_UNORM, False), (_ffi.RM_FORMAT_BGRA8_UNORM, True)], ids=["rgba8", "bgra8"])
@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC], ids=["v5_lds", "v5_spec"])
def test_8bit_output_formats(res, oracle, fmt, bgra, kernel):
    """Output stage: the 8-bit image is the UNORM8 quantisation of the RGBA32F image, byte for byte --
    single draws (ragged sizes, row bands), interleaved strips and batches, culled and marched tiles alike."""
    try:
        for W, H in [(64, 48), (37, 21)]:
            cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
            lim = (0.01, 100.0, 96)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            ref = oracle.quantize_unorm8(oracle.render(u, lim, cc, w, W, H, threads=4), bgra=bgra)
            res.set_output_format(fmt)
            img = res.draw(W, H)
            assert img.dtype == np.uint8 and img.shape == (H, W, 4)
            assert img.tobytes() == ref.tobytes()
            band = res.draw(W, H, row0=8, rows=9)
            assert band.tobytes() == ref[8:17].tobytes()
            res.set_output_format(_ffi.RM_FORMAT_RGBA32F)
        from ray_marching_amd import shard
        W, H = 64, 48
        cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
        setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=(0.01, 100.0, 96), kernel=kernel)
        res.set_output_format(fmt)
        full = np.zeros((H, W, 4), np.uint8)
        for rank in range(2):
            shard.scatter_strips(full, res.draw_strips(W, H, 16, rank, 2), H, rank, 2, 16)
        assert full.tobytes() == ref_of(oracle, scenes.g32(), W, H, (0.01, 100.0, 96), bgra).tobytes()
        frames = []
        for ev in ([(1, 35.0, -25.0)], [(1, -60.0, 30.0)]):
            uu, *_ = oracle.orbit_uniforms((float(W), float(H)), events=ev)
            frames.append(uu)
        batch = res.draw_batch([_ffi.Uniforms.from_buffer_copy(bytes(f)) for f in frames], W, H)
        cc, w = oracle.serialize(*scenes.g32())
        for i, f in enumerate(frames):
            want = oracle.quantize_unorm8(oracle.render(f, (0.01, 100.0, 96), cc, w, W, H, threads=4), bgra=bgra)
            assert batch[i].tobytes() == want.tobytes()
        # the v1 kernel does not have an output stage
        res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_PIXEL)
        with pytest.raises(_ffi.RmError) as e:
            res.draw(W, H)
        assert e.value.status == _ffi.RM_ERR_ARG
    finally:
        res.set_output_format(_ffi.RM_FORMAT_RGBA32F)
        res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)


def ref_of(oracle, scene, W, H, lim, bgra):
    cc, w, u = oracle_case(oracle, scene, W, H, None)
    return oracle.quantize_unorm8(oracle.render(u, lim, cc, w, W, H, threads=4), bgra=bgra)


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC], ids=["v5", "v5_lds", "v5_spec"])
def test_space_transformations_and_their_miss_tests(res, oracle, kernel):
    """Translation / Rotation / Scale (opcodes 200-205): interpreter and specialised kernels against the oracle;
    miss-ray culling of transformed primitives (world-space bounding spheres from the decoder) must never change a
    pixel -- including programs whose transforms are not similarities, where culling has to step aside."""
    import math
    W, H = 80, 56
    h = math.sqrt(0.5)

    def tree(variant):
        t = scenes._Tab()
        if variant == "mix":
            return scenes.xform_mix()
        if variant == "deep":          # eight nested scopes around one box, next to an untransformed sphere
            n = t.box((0.2, 0.0, 0.0), (0.5, 0.3, 0.2))
            for k in range(8):
                n = [t.translation(n, (0.1, -0.05, 0.02)), t.rotation(n, (math.cos(0.2), 0.0, math.sin(0.2), 0.0)),
                     t.scale(n, 1.1)][k % 3]
            return t.nodes, t.op(scenes.UNION, t.sphere((-1.2, 0, 0), 0.6), n)
        if variant == "far":           # the transform moves a primitive into view from far away
            return t.nodes, t.translation(t.sphere((100.0, 0, 0), 0.9), (-100.0, 0.2, 0.0))
        if variant == "not_unit":      # |q| = 1.2: not a rotation (distances are distorted): culling must veto
            return t.nodes, t.op(scenes.UNION, t.rotation(t.box((0, 0, 0), (0.7, 0.4, 0.3)), (1.2 * h, 0, 0, 1.2 * h)), t.sphere((1.5, 0, 0), 0.4))
        if variant == "negative_scale":
            return t.nodes, t.op(scenes.UNION, t.scale(t.sphere((0.5, 0.2, 0), 0.6), -1.5), t.box((-1.2, 0, 0), (0.3, 0.3, 0.3)))
        if variant == "zero_scale":    # division by zero: NaN positions inside the scope
            return t.nodes, t.op(scenes.UNION, t.scale(t.sphere((0, 0, 0), 0.6), 0.0), t.box((-1.2, 0, 0), (0.3, 0.3, 0.3)))
        raise KeyError(variant)

    res.resize_command_buffer(4096)
    for variant in ("mix", "deep", "far", "not_unit", "negative_scale", "zero_scale"):
        nodes, root = tree(variant)
        cc, w = oracle.serialize(nodes, root)
        for events in (scenes.STILL_CAMERA_EVENTS, [(1, 10.0, -150.0)], [(2, -95.0, 0.0)]):
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            lim = (0.01, 100.0, 80)
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            res.set_option(_ffi.RM_OPT_CULL, 0)
            off = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_CULL, 1)
            on = res.draw(W, H)
            assert off.tobytes() == ref.tobytes(), (variant, events, "cull off")
            assert on.tobytes() == ref.tobytes(), (variant, events, "cull on")
    res.resize_command_buffer(1024)
    # a malformed nest is refused with the decoder's status
    f = lambda x: int(np.float32(x).view(np.uint32))
    with pytest.raises(_ffi.RmError) as e:
        res.set_program(2, np.array([200, f(1), f(0), f(0), 0, f(0), f(0), f(0), f(1)], np.uint32))
    assert e.value.status == _ffi.RM_ERR_TRANSFORM
    cc, w = oracle.serialize(*scenes.g8())
    res.set_program(cc, w)


def test_frames_in_flight_on_context_owned_streams(oracle):
    """RM_STREAM_OWN / rm_sync_context: three contexts, frame f -> context f % 3, every frame on its context's own
    stream into its own device buffer, one wait per context at the end -- the images are the serial ones."""
    import torch
    W, H, F = 96, 64, 3
    cc, w = oracle.serialize(*scenes.g32())
    lim = (0.01, 100.0, 96)
    ctxs, bufs, frames = [], [], []
    for ev in ([(1, 35.0, -25.0)], [(1, 80.0, -10.0)], [(1, -60.0, 30.0)], [(1, 10.0, -150.0)], [(2, -95.0, 0.0)], [(1, 200.0, 5.0)]):
        uu, *_ = oracle.orbit_uniforms((float(W), float(H)), events=ev)
        frames.append(uu)
    try:
        for i in range(F):
            r = renderer.RayMarchingResources(0)
            r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
            r.set_limits(lim)
            r.set_program(cc, w)
            ctxs.append(r)
        bufs = [torch.empty((H, W, 4), dtype=torch.float32, device="cuda") for _ in frames]
        for f, uu in enumerate(frames):
            c = ctxs[f % F]
            c.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(uu)))
            c.draw_device(W, H, bufs[f].data_ptr(), stream=_ffi.RM_STREAM_OWN)
        for c in ctxs:
            c.sync_context()
        for f, uu in enumerate(frames):
            assert bufs[f].cpu().numpy().tobytes() == oracle.render(uu, lim, cc, w, W, H, threads=4).tobytes(), f
    finally:
        for c in ctxs:
            c.close()


def test_program_change_with_a_frame_in_flight(oracle):
    """wgpu's queue.write_buffer is ordered with the draws of the queue (renderer.rs:230-254): a program written for
    frame n+1 must not reach frame n, which may still be queued or running when rm_set_program + rm_draw of frame
    n+1 return.  One context, one stream, six frames back to back alternating between two scenes of the same
    structure (same specialised kernel, different parameters) and a third of another size, no wait in between; the
    batch uniforms of rm_draw_batch are uploaded the same way."""
    import torch
    W, H = 640, 360          # large enough for a frame to still be in flight when the next program arrives
    lim = (0.01, 100.0, 96)
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    nodes, root = scenes.g32()
    progs = [oracle.serialize(nodes, root)]
    cc, w = progs[0]
    moved = np.array(w, dtype=np.uint32).copy()
    fl = moved.view(np.float32)
    i = 0
    while i < len(moved):                                # same opcodes, every primitive lifted by 0.4
        op = int(moved[i])
        if op in (0, 1):
            fl[i + 2] += np.float32(0.4)
            i += 5 if op == 0 else 7
        else:
            i += 1
    progs.append((cc, moved))
    progs.append(oracle.serialize(*scenes.g8()))
    refs = [oracle.render(u, lim, c_, w_, W, H, threads=4).tobytes() for c_, w_ in progs]
    assert len(set(refs)) == 3
    r = renderer.RayMarchingResources(0)
    try:
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        for mode in (0, 2):
            r.set_option(_ffi.RM_OPT_SPECIALIZE, mode)
            order = [0, 1, 0, 2, 1, 0]
            if mode == 2:                                # compile outside of the pipelined part
                for k in (0, 2):
                    r.set_program(*progs[k])
                    r.draw(W, H)
            bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in order]
            for b, k in zip(bufs, order):
                r.set_program(*progs[k])
                r.draw_device(W, H, b.data_ptr(), stream=_ffi.RM_STREAM_OWN)
            r.sync_context()
            for n, (b, k) in enumerate(zip(bufs, order)):
                assert b.cpu().numpy().tobytes() == refs[k], (mode, n, k)
        # batches back to back: the second batch's cameras must not reach the first
        cams, ocams = [], []
        for ev in ([(1, 35.0, -25.0)], [(1, 80.0, -10.0)], [(1, -60.0, 30.0)], [(1, 10.0, -150.0)]):
            uu, *_ = oracle.orbit_uniforms((float(W), float(H)), events=ev)
            ocams.append(uu)
            cams.append(_ffi.Uniforms.from_buffer_copy(bytes(uu)))
        r.set_program(*progs[0])
        out = [torch.zeros((2, H, W, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
        r.draw_batch_device(cams[:2], W, H, out[0].data_ptr(), stream=_ffi.RM_STREAM_OWN)
        r.draw_batch_device(cams[2:], W, H, out[1].data_ptr(), stream=_ffi.RM_STREAM_OWN)
        r.sync_context()
        for j, cam in enumerate(ocams):
            ref = oracle.render(cam, lim, cc, w, W, H, threads=4)
            assert out[j // 2][j % 2].cpu().numpy().tobytes() == ref.tobytes(), j
    finally:
        r.close()


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE], ids=["v5_lds", "v5_spec", "v5_spec_prune"])
def test_smooth_union_slack_bound_is_safe(res, oracle, kernel):
    """The miss tests inflate every bound by how far SmoothUnion can pull the tree below its leaves (rm_decode.h:
    max k for a chain, + k/4 where two blended sub-trees meet).  Worst cases: many leaves at the SAME distance from
    one point, blended with a large k, grow a blob where no primitive is; culling must not remove the rays that hit
    it.  Chains, balanced trees, mixed k, a scaled chain."""
    import math
    W, H = 72, 48

    def ring(t, n, radius, r, y=0.0):
        return [t.sphere((radius * math.cos(2 * math.pi * i / n), y, radius * math.sin(2 * math.pi * i / n)), r) for i in range(n)]

    def chain(t, leaves, k):
        acc = leaves[0]
        for j, leaf in enumerate(leaves[1:]):
            acc = t.smooth_union(acc, leaf, k[j % len(k)] if isinstance(k, (list, tuple)) else k)
        return acc

    def balanced(t, level, k):
        while len(level) > 1:
            level = [t.smooth_union(level[i], level[i + 1], k) for i in range(0, len(level), 2)]
        return level[0]

    cases = {}
    t = scenes._Tab(); cases["ring_chain_k1.2"] = (t.nodes, chain(t, ring(t, 8, 1.0, 0.3), 1.2))
    t = scenes._Tab(); cases["ring_chain_mixed_k"] = (t.nodes, chain(t, ring(t, 12, 1.3, 0.25), [0.2, 1.5, 0.6]))
    t = scenes._Tab(); cases["ring_balanced_k0.9"] = (t.nodes, balanced(t, ring(t, 8, 1.0, 0.3), 0.9))
    t = scenes._Tab(); cases["two_rings_blended"] = (t.nodes, t.smooth_union(chain(t, ring(t, 6, 0.9, 0.25, 0.4), 0.8),
                                                                               chain(t, ring(t, 6, 0.9, 0.25, -0.4), 0.8), 1.0))
    t = scenes._Tab(); cases["scaled_chain"] = (t.nodes, t.scale(chain(t, ring(t, 8, 1.0, 0.3), 1.2), 1.7))
    res.resize_command_buffer(4096)
    for name, (nodes, root) in cases.items():
        cc, w = oracle.serialize(nodes, root)
        for events in (scenes.STILL_CAMERA_EVENTS, [(1, 10.0, -150.0)], [(1, 35.0, -25.0), (2, 120.0, 0.0)]):
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            for lim in [(0.01, 100.0, 96), (0.3, 100.0, 48)]:
                ref = oracle.render(u, lim, cc, w, W, H, threads=4)
                setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
                res.set_option(_ffi.RM_OPT_CULL, 0)
                off = res.draw(W, H)
                res.set_option(_ffi.RM_OPT_CULL, 1)
                on = res.draw(W, H)
                assert off.tobytes() == ref.tobytes(), (name, events, lim, "cull off")
                assert on.tobytes() == ref.tobytes(), (name, events, lim, "cull on")
    res.resize_command_buffer(1024)


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5, _ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE], ids=["v5", "v5_lds", "v5_spec", "v5_spec_prune"])
def test_miss_test_on_lower_bounds_is_exact(res, oracle, kernel):
    """Programs that blend (SmoothUnion) get a second, sharper miss test for the rays the inflated bounds cannot clear:
    the program run on lower bounds of its leaves along the ray (rm_kernel_v5.h "Miss test on lower bounds").  Random
    chains of spheres and boxes joined by SmoothUnion / Union / Subtraction / Intersection -- k small, large, zero,
    negative; flat, degenerate and negative sizes; everything 1000 units from the origin -- from eight cameras (inside a
    solid, grazing, top-down, far): culling on == culling off == oracle.  Cylinders (through their bounding box) and Planes
    (linear along a ray: bounded exactly when the ray points away) take part; a balanced tree of blends (needs a deeper
    value stack) keeps the plain tests and must still be right."""
    rng = np.random.default_rng(2024)
    W, H = 64, 40
    lim = (0.01, 100.0, 80)
    res.resize_command_buffer(4096)

    def random_chain(offset, n):
        t = scenes._Tab()
        acc = None
        for j in range(n):
            c = rng.uniform(-1.6, 1.6, 3) + np.array(offset)
            if rng.random() < 0.5:
                leaf = t.sphere(tuple(c), float(rng.choice([rng.uniform(0.2, 0.6), 0.0, -0.15], p=[0.8, 0.1, 0.1])))
            else:
                h = rng.uniform(0.1, 0.6, 3)
                if rng.random() < 0.2:
                    h[rng.integers(3)] = rng.choice([0.0, -0.1])
                leaf = t.box(tuple(c), tuple(h))
            if acc is None:
                acc = leaf
                continue
            r = rng.random()
            if r < 0.6:
                acc = t.smooth_union(acc, leaf, float(rng.choice([0.25, 0.05, 0.9, 0.0, -0.3])))
            elif r < 0.75:
                acc = t.op(scenes.UNION, acc, leaf)
            elif r < 0.9:
                acc = t.op(scenes.SUBTRACTION, acc, leaf)
            else:
                acc = t.op(scenes.INTERSECTION, acc, leaf)
        return t.nodes, acc

    programs = [("chain%d" % i, (0.0, 0.0, 0.0), random_chain((0.0, 0.0, 0.0), n)) for i, n in enumerate((2, 5, 9, 14))]
    programs.append(("far_from_origin", (1000.0, -2000.0, 500.0), random_chain((1000.0, -2000.0, 500.0), 8)))
    programs.append(("config3", (0.0, 0.0, 0.0), scenes.EXT_SCENES["g32s"]()))
    t = scenes._Tab()
    level = [t.sphere(tuple(rng.uniform(-1.5, 1.5, 3)), 0.4) for _ in range(8)]
    while len(level) > 1:
        level = [t.smooth_union(level[i], level[i + 1], 0.3) for i in range(0, len(level), 2)]
    programs.append(("balanced_blends", (0.0, 0.0, 0.0), (t.nodes, level[0])))
    t = scenes._Tab()    # a cylinder is bounded through its bounding box
    programs.append(("blended_cylinder", (0.0, 0.0, 0.0),
                     (t.nodes, t.smooth_union(t.sphere((-0.5, 0, 0), 0.6), t.cylinder((0.6, 0.0, 0.1), 0.4, 0.7), 0.3))))
    t = scenes._Tab()    # a ground Plane (|n| = 2: the value is not a distance) under two solids: rays that point away from it
    ground = t.plane((0.0, 2.0, 0.0), 2.4)
    programs.append(("solids_on_a_plane", (0.0, 0.0, 0.0),
                     (t.nodes, t.op(scenes.UNION, t.op(scenes.UNION, t.sphere((-0.7, 0.0, 0.0), 0.8), t.box((0.9, -0.4, 0.2), (0.5, 0.8, 0.5))), ground))))
    t = scenes._Tab()    # a solid cut by a tilted Plane (Intersection), blended with a sphere
    cutter = t.plane((0.3, 1.0, -0.2), 0.1)
    programs.append(("plane_cut_blend", (0.0, 0.0, 0.0),
                     (t.nodes, t.smooth_union(t.op(scenes.INTERSECTION, t.box((0, 0, 0), (1.0, 0.8, 0.9)), cutter), t.sphere((1.3, 0.4, 0.0), 0.5), 0.3))))
    for name, target, (nodes, root) in programs:
        cc, w = oracle.serialize(nodes, root)
        for cam in sorted(CULL_CAMERAS):
            spec = dict(CULL_CAMERAS[cam])
            if "target" not in spec:
                spec["target"] = target
            elif name != "far_from_origin":
                continue
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), **spec)
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            res.set_option(_ffi.RM_OPT_CULL, 0)
            off = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_CULL, 1)
            on = res.draw(W, H)
            assert off.tobytes() == ref.tobytes(), (name, cam, "cull off")
            assert on.tobytes() == ref.tobytes(), (name, cam, "cull on")
    res.resize_command_buffer(1024)


@pytest.mark.parametrize("kernel", [_ffi.RM_KERNEL_V5_LDS, KERNEL_SPEC, KERNEL_SPEC_PRUNE], ids=["v5_lds", "v5_spec", "v5_spec_prune"])
def test_subtracted_primitives_leave_the_miss_tests(res, oracle, kernel):
    """max(a, -b) >= a: a march position registers a hit only near a leaf of a Subtraction's LEFT operand, so the leaves of
    its right operand -- a single primitive or a whole sub-tree -- have no entry in the miss-test tables (rm_decode.h,
    RM_OP_NOCULL) and a ray that only comes near them is not marched.  Subtractors that carve visible holes, that float in
    empty space, that contain the camera; sub-trees on the right (a union, another subtraction: its own right operand then
    counts positively in the value but is still dropped from the tables, which is only conservative); culling on == off ==
    oracle from eight cameras."""
    W, H = 64, 40
    lim = (0.01, 100.0, 80)
    res.resize_command_buffer(4096)
    U, S, I = scenes.UNION, scenes.SUBTRACTION, scenes.INTERSECTION
    cases = {}
    t = scenes._Tab()   # a block with a spherical bite, plus a subtractor far from everything
    cases["bite_and_lonely_subtractor"] = (t.nodes, t.op(S, t.op(S, t.box((0, 0, 0), (1.0, 0.6, 0.8)), t.sphere((0.9, 0.5, 0.0), 0.7)),
                                                         t.sphere((-2.5, 1.0, 0.5), 0.8)))
    t = scenes._Tab()   # right operand is a union of two primitives
    cases["minus_union"] = (t.nodes, t.op(S, t.sphere((0, 0, 0), 1.2), t.op(U, t.box((0.8, 0, 0), (0.6, 0.3, 1.5)), t.sphere((-1.0, 0.8, 0), 0.6))))
    t = scenes._Tab()   # right operand is itself a subtraction: a - (b - c) = max(a, min(-b, c))
    cases["minus_subtraction"] = (t.nodes, t.op(S, t.box((0, 0, 0), (1.2, 1.0, 1.0)), t.op(S, t.sphere((0.6, 0.4, 0.2), 1.0), t.box((0.6, 0.4, 0.2), (0.4, 0.4, 0.4)))))
    t = scenes._Tab()   # a union whose second member is a carved block; the subtractor also pokes out into empty space
    cases["union_of_carved"] = (t.nodes, t.op(U, t.sphere((-1.4, 0, 0), 0.6), t.op(S, t.box((0.8, 0, 0), (0.7, 0.7, 0.7)), t.box((1.6, 0.5, 0), (0.9, 0.3, 0.3)))))
    t = scenes._Tab()   # intersection with a carved operand
    cases["intersection_of_carved"] = (t.nodes, t.op(I, t.sphere((0, 0, 0), 1.1), t.op(S, t.box((0, 0, 0), (0.9, 0.9, 0.9)), t.sphere((0, 1.0, 0), 0.6))))
    t = scenes._Tab()   # inside transform scopes (the tables then hold world-space bounding spheres: the subtracted one's clears everything)
    hq = 0.70710678
    carved = t.op(S, t.box((0, 0, 0), (0.9, 0.6, 0.7)), t.sphere((0.7, 0.5, 0.0), 0.6))
    cases["carved_in_transforms"] = (t.nodes, t.op(S, t.op(U, t.translation(t.rotation(carved, (hq, 0, hq, 0)), (-0.8, 0.2, 0.1)),
                                                          t.scale(t.sphere((1.2, 0, 0), 0.5), 1.4)),
                                                   t.translation(t.box((0, 0, 0), (0.4, 0.4, 0.4)), (2.6, 1.2, -0.5))))
    cases["g32"] = scenes.g32()
    cases["g8"] = scenes.g8()
    cases["xform_mix"] = scenes.EXT_SCENES["xform_mix"]()
    for name, (nodes, root) in cases.items():
        cc, w = oracle.serialize(nodes, root)
        for cam in sorted(CULL_CAMERAS):
            if "target" in CULL_CAMERAS[cam]:
                continue
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), **CULL_CAMERAS[cam])
            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
            setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=kernel)
            res.set_option(_ffi.RM_OPT_CULL, 0)
            off = res.draw(W, H)
            res.set_option(_ffi.RM_OPT_CULL, 1)
            on = res.draw(W, H)
            assert off.tobytes() == ref.tobytes(), (name, cam, "cull off")
            assert on.tobytes() == ref.tobytes(), (name, cam, "cull on")
    res.resize_command_buffer(1024)


def test_floor_and_sky_tiles_vs_oracle(res, oracle):
    """Tiles no primitive touches are finished in the pre-pass: sky pixels are zeros, a floor pixel whose sixteen samples
    provably land in one checker cell is sixteen equal terms, the pixels a cell edge crosses are sampled one by one (few per
    tile: sample-parallel; many: the plain loop).  Frames that are almost all floor and sky, from cameras that stress the
    bounds: close to the floor plane, grazing along it (cells shrink to sub-pixel size towards the horizon), straight down,
    from below the plane (nothing is assumed there), zoomed far in (one cell edge across the frame), off-axis; every pixel
    against the oracle."""
    W, H = 320, 200
    t = scenes._Tab()
    nodes, root = t.nodes, t.op(scenes.SUBTRACTION, t.sphere((0.0, 0.3, 0.0), 0.35), t.box((0.2, 0.4, 0.0), (0.2, 0.2, 0.2)))
    cc, w = oracle.serialize(nodes, root)
    cams = {
        "still": dict(events=scenes.STILL_CAMERA_EVENTS),
        "low_over_the_floor": dict(target=(0.0, -1.2, 0.0), events=[(1, 20.0, -4.0), (2, -40.0, 0.0)]),
        "grazing": dict(target=(0.0, -1.45, 0.0), events=[(1, 140.0, -1.0)]),
        "straight_down": dict(events=[(1, 0.0, -150.0)]),
        "from_below": dict(target=(0.0, -3.0, 0.0), events=[(1, 30.0, 60.0)]),
        "zoomed_in": dict(target=(0.3, -1.5, 0.2), events=[(1, 15.0, -60.0), (2, -97.0, 0.0)]),
        "panned": dict(events=[(1, 35.0, -25.0), (0, 400.0, -150.0)]),
        "far": dict(events=[(1, -60.0, -30.0), (2, 600.0, 0.0)]),
    }
    for fmt in (_ffi.RM_FORMAT_RGBA32F,):
        for name, spec in cams.items():
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), **spec)
            for lim in ((0.01, 100.0, 64), (0.01, 100.0, 0)):          # max_iter = 0: every ray is a miss, every tile is finished there
                setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim, kernel=_ffi.RM_KERNEL_DEFAULT)
                assert_same(res.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=8))


def test_draw_is_stream_capturable(oracle):
    """After its first (allocating, compiling) draw of a size, rm_draw with a device destination issues nothing but
    kernel launches on the caller's stream: it can be captured into a HIP graph and replayed."""
    import torch
    W, H = 96, 64
    cc, w = oracle.serialize(*scenes.g32())
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 96)
    ref = oracle.render(u, lim, cc, w, W, H, threads=4)
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_program(cc, w)
        out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        s = torch.cuda.Stream()
        r.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)       # warm: scratch buffers, compiled kernel
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin()
            r.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)
            g.capture_end()
        for _ in range(3):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert out.cpu().numpy().tobytes() == ref.tobytes()
    finally:
        r.close()


@pytest.mark.parametrize("form", ["generated", "generated_without_subtractor_tests", "interpreter"])
def test_culling_rules_of_blending_programs(oracle, form):
    """Programs that blend along a top-level chain are culled per wave (rm_kernel_v5.h "Wave-level culling", rm_units.h): a unit
    whose leaf is at least k above every value the accumulator can have is skipped, one at least k below restarts the chain,
    a subtractor or an intersection that cannot change the accumulator is skipped -- exact only if "at least k" is decided on
    the safe side of every rounding and every bound.  Scenes built to sit ON the boundaries and around them: concentric
    spheres whose radii differ by exactly k (the second leaf's value is the accumulator + k up to an ulp, everywhere),
    staircases of leaves each within k of the previous accumulator (every one matters although all but the last are far above
    the final value), k = 0 / negative / 1e-6 / larger than the scene, leaves that coincide or are 40 units apart, everything
    1000 units from the origin, Unions, Subtractions and an Intersection mixed into the chain, sub-trees and a cylinder and a
    plane as opaque units -- and the parameters (k included, through zero) moving under ONE compiled kernel."""
    import math
    rng = np.random.default_rng(4242)
    W, H = 80, 56
    r = renderer.RayMarchingResources(0)

    def chain(t, leaves, ks, ops=None):
        acc = leaves[0]
        for j, leaf in enumerate(leaves[1:]):
            op = ops[j % len(ops)] if ops else "m"
            if op == "m":
                acc = t.smooth_union(acc, leaf, ks[j % len(ks)])
            else:
                acc = t.op({"u": scenes.UNION, "s": scenes.SUBTRACTION, "i": scenes.INTERSECTION}[op], acc, leaf)
        return acc

    def cases(offset, kscale):
        ox, oy, oz = offset
        out = {}
        t = scenes._Tab()      # radii differ by exactly k: v_second = v_first + k up to an ulp, at every position
        out["concentric_plus_k"] = (t.nodes, chain(t, [t.sphere((ox, oy, oz), 1.0), t.sphere((ox, oy, oz), 0.75), t.sphere((ox + 0.5, oy, oz), 0.5),
                                                      t.sphere((ox + 0.5, oy, oz), 0.25), t.box((ox, oy + 0.2, oz), (0.3, 0.3, 0.3))], [0.25 * kscale]))
        t = scenes._Tab()      # a staircase along x: each leaf 0.8 k closer to the viewer's side than the one before
        k = 0.5 * kscale
        out["staircase"] = (t.nodes, chain(t, [t.sphere((ox - 1.6 + 0.4 * j, oy, oz + 0.1 * j), 0.3) if j % 2 else t.box((ox - 1.6 + 0.4 * j, oy, oz + 0.1 * j), (0.25, 0.3, 0.2))
                                                 for j in range(9)], [k]))
        t = scenes._Tab()      # every kind of k in one chain
        out["mixed_k"] = (t.nodes, chain(t, [t.sphere((ox + 0.7 * math.cos(j), oy + 0.3 * math.sin(2 * j), oz + 0.7 * math.sin(j)), 0.35) for j in range(10)],
                                           [0.3 * kscale, 0.0, -0.2, 1.0e-6, 7.0 * kscale, 0.05]))
        t = scenes._Tab()      # unions, subtractions and an intersection between the blends; boxes and spheres
        lv = [(t.sphere if j % 3 else t.box)((ox + rng.uniform(-1.5, 1.5), oy + rng.uniform(-0.6, 0.6), oz + rng.uniform(-1.5, 1.5)),
                                             0.4 if j % 3 else (0.3, 0.25, 0.35)) for j in range(14)]
        out["mixed_ops"] = (t.nodes, chain(t, lv, [0.3 * kscale, 0.6 * kscale], ops="mmusmmummsmim"))
        t = scenes._Tab()      # partners of a pair: coincident, and 40 units apart (a huge, never-far bounding sphere)
        lv = []
        for j in range(10):
            c = (ox + rng.uniform(-1.2, 1.2), oy + rng.uniform(-0.5, 0.5), oz + rng.uniform(-1.2, 1.2))
            if j % 4 == 2:
                c = tuple(t.nodes[lv[-1]][1][:3])
            if j % 4 == 0 and j:
                c = (c[0] + 40.0, c[1], c[2] - 25.0)
            lv.append(t.sphere(c, float(rng.choice([0.4, 0.0, -0.2], p=[0.8, 0.1, 0.1]))) if j % 2 else t.box(c, (0.3, 0.3, 0.3)))
        out["partners"] = (t.nodes, chain(t, lv, [0.35 * kscale]))
        t = scenes._Tab()      # blended sub-trees on both sides of a blend, and a cylinder and a plane (never skipped) in the chain
        left = chain(t, [t.sphere((ox - 0.8, oy, oz), 0.5), t.box((ox - 0.2, oy, oz), (0.3, 0.3, 0.3)), t.cylinder((ox - 0.5, oy + 0.5, oz), 0.2, 0.4)], [0.3 * kscale])
        right = chain(t, [t.box((ox + 0.8, oy, oz), (0.3, 0.4, 0.3)), t.sphere((ox + 0.4, oy + 0.3, oz), 0.3), t.plane((0.0, 1.0, 0.0), 1.2 - oy), t.sphere((ox + 1.2, oy, oz + 0.4), 0.3)],
                      [0.2 * kscale, 0.4 * kscale])
        out["subtrees"] = (t.nodes, t.smooth_union(left, right, 0.5 * kscale))
        return out

    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_option(_ffi.RM_OPT_PRUNE, 1)
        r.resize_command_buffer(4096)
        # a knob of the generator, read when a structure is generated (and part of the kernel cache's key): subtracted leaves
        # without their local test
        if form == "generated_without_subtractor_tests":
            os.environ["RM_JIT_SUB_TESTS"] = "0"
        if form == "interpreter":
            r.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
        for offset in ((0.0, 0.0, 0.0), (800.0, -300.0, 500.0)):
            for kscale in (1.0, 0.0, -1.0):                  # the same structures again with every k scaled: same compiled kernels
                for name, (nodes, root) in cases(offset, kscale).items():
                    cc, w = oracle.serialize(nodes, root)
                    info = renderer.program_info(cc, w)
                    assert info["auto_pruned"] in (0, 2) and info["prunable"] == 0, (name, info)
                    r.set_program(cc, w)
                    for events in (scenes.STILL_CAMERA_EVENTS, [(1, 170.0, 60.0), (2, -35.0, 0.0)]):
                        u, *_ = oracle.orbit_uniforms((float(W), float(H)), target=offset, events=events)
                        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
                        for lim in ((0.01, 100.0, 96), (0.002, 30.0, 200)):
                            r.set_limits(lim)
                            ref = oracle.render(u, lim, cc, w, W, H, threads=4)
                            for cull in (1, 0):
                                r.set_option(_ffi.RM_OPT_CULL, cull)
                                img = r.draw(W, H)
                                if form != "interpreter":
                                    assert r.info(_ffi.RM_INFO_SPECIALIZED) == 1 and r.info(_ffi.RM_INFO_PRUNED) == 2, (name, r.jit_log())
                                if img.tobytes() != ref.tobytes():
                                    bad = np.argwhere((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1))
                                    raise AssertionError("%s offset %s kscale %g events %s limits %s cull %d: %d pixels differ (first %s)"
                                                         % (name, offset, kscale, events, lim, cull, len(bad), bad[:3].tolist()))
    finally:
        os.environ.pop("RM_JIT_SUB_TESTS", None)
        r.close()


def test_host_draw_while_another_stream_still_captures_this_context(oracle):
    """Capture begun on stream S, a draw of the context captured there, and BEFORE the capture ends a host-destination draw of
    the same context (which runs on the context's own stream).  The context orders consecutive draws on different streams with
    an event -- but an event recorded on the capturing stream would become a node of its graph and the wait would drag the
    host draw's kernels, copy and synchronisation into the capture (or fail it).  Captured work has not run: the host draw
    simply runs, the capture stays valid, and both images are the oracle's."""
    import torch
    W, H = 96, 64
    cc, w = oracle.serialize(*scenes.g32())
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 96)
    ref = oracle.render(u, lim, cc, w, W, H, threads=4)
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_program(cc, w)
        out = torch.zeros((H, W, 4), dtype=torch.float32, device="cuda")
        s = torch.cuda.Stream()
        r.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)       # warm: scratch buffers, compiled kernel
        s.synchronize()
        assert r.draw(W, H).tobytes() == ref.tobytes()                   # warm the host path too (its device scratch)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            g.capture_begin(capture_error_mode="relaxed")
            r.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)
            host = r.draw(W, H)                                          # capture still open on s
            r.draw_device(W, H, out.data_ptr(), stream=s.cuda_stream)   # and back on the capturing stream
            g.capture_end()
        assert host.tobytes() == ref.tobytes()
        for _ in range(2):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert out.cpu().numpy().tobytes() == ref.tobytes()
        assert r.draw(W, H).tobytes() == ref.tobytes()
    finally:
        r.close()


def test_8k_config_row_bands_vs_oracle(res, oracle):
    """BASELINE configs[3] at its full size (7680x4320, 64-node graph, 512 steps; 518 400 tiles) on the default path:
    sampled row bands bit-exact against the oracle, alpha plane and finiteness of the whole frame."""
    W, H = 7680, 4320
    lim = (0.01, 100.0, 512)
    cc, w, u = oracle_case(oracle, scenes.g64(), W, H, None)
    setup(res, cc=cc, words=w, u=_ffi.Uniforms.from_buffer_copy(bytes(u)), limits=lim)
    full = res.draw(W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
    assert np.isfinite(full).all() and np.array_equal(full[..., 3], np.ones((H, W), np.float32))
    for r0, rows in [(1300, 2), (2161, 2), (3000, 1)]:
        assert_same(full[r0:r0 + rows], oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=16))
    del full


def orbit_frame_uniforms(oracle, W, H, f, n_frames=1024):
    """Uniforms of frame f of the orbit batch (BASELINE config 5): yaw 2 pi f / N, pitch -0.25, radius 5, through the
    oracle's controller and prepare()."""
    import math
    L = oracle.lib()
    orb = type(oracle.orbit_uniforms((1.0, 1.0))[3])()
    t = np.zeros(3, np.float32)
    f32p = C.POINTER(C.c_float)
    L.rmo_orbit_new(C.byref(orb), t.ctypes.data_as(f32p), 5.0)
    orb.yaw, orb.pitch, orb.radius = 2.0 * math.pi * f / n_frames, -0.25, 5.0
    pos, q = np.zeros(3, np.float32), np.zeros(4, np.float32)
    L.rmo_orbit_camera(C.byref(orb), pos.ctypes.data_as(f32p), q.ctypes.data_as(f32p))
    u = type(oracle.orbit_uniforms((1.0, 1.0))[0])()
    assert L.rmo_prepare_uniforms(float(W), float(H), pos.ctypes.data_as(f32p), q.ctypes.data_as(f32p), C.byref(u)) == 0
    return u


def test_library_defaults_full_frame_vs_oracle(oracle):
    """The metric frame exactly as bench.py draws it: a FRESH context, no rm_set_option call at all (so whatever the
    library's defaults are -- pruned + grouped far tests + four taps in one pass + temporal tile order, compiled in the
    background) against the oracle, every pixel; drawn until the specialised kernel has taken over, and once more so
    that the tile order comes from the previous frame's measurements."""
    import time
    W, H = 1920, 1080
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    lim = (0.01, 100.0, 256)
    ref = oracle.render(u, lim, cc, w, W, H, threads=16)
    r = renderer.RayMarchingResources(0)
    try:
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_program(cc, w)
        assert_same(r.draw(W, H), ref)                      # interpreter kernel while the compiler runs
        t0 = time.monotonic()
        while r.info(_ffi.RM_INFO_JIT_STATE) == 1.0 and time.monotonic() - t0 < 120:
            time.sleep(0.05)
        assert r.info(_ffi.RM_INFO_JIT_STATE) == 2.0, r.jit_log()
        for _ in range(2):
            assert_same(r.draw(W, H), ref)
            assert r.info(_ffi.RM_INFO_SPECIALIZED) == 1 and r.info(_ffi.RM_INFO_PRUNED) == 1
    finally:
        r.close()


@pytest.mark.parametrize("scene", ["g32", "g32s"])
def test_4k_configs_row_bands_vs_oracle(oracle, scene):
    """BASELINE configs[2] (3840x2160, 32-node graph with smooth-min blends, 256 steps) and configs[4] (the orbit batch
    at 3840x2160, G32) at their full size on the default path: two orbit-batch cameras each, sampled row bands bit-exact
    against the oracle, alpha plane and finiteness of the whole frame."""
    W, H = 3840, 2160
    lim = (0.01, 100.0, 256)
    cc, w = oracle.serialize(*{**scenes.SCENES, **scenes.EXT_SCENES}[scene]())
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(lim)
        r.set_program(cc, w)
        for f in (0, 341):                                   # frame 0 and one a third of the way round
            u = orbit_frame_uniforms(oracle, W, H, f)
            r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
            full = r.draw(W, H)
            assert r.info(_ffi.RM_INFO_SPECIALIZED) == 1
            assert np.isfinite(full).all() and np.array_equal(full[..., 3], np.ones((H, W), np.float32))
            for r0, rows in [(700, 2), (1081, 3), (1500, 2)]:
                assert_same(full[r0:r0 + rows], oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=16))
            del full
    finally:
        r.close()


def test_host_draw_is_ordered_with_draws_on_a_caller_stream(oracle):
    """Draws of one context share its scratch buffers (program copy, work list, counters).  A host-destination draw runs
    on the context's own stream: it must wait for device-destination draws the caller queued on ANOTHER stream, and a
    later draw on that stream must wait for it (rm_abi.h, order_with_previous)."""
    torch = pytest.importorskip("torch")
    W, H = 480, 270
    lim = (0.01, 100.0, 256)
    cc32, w32, u = oracle_case(oracle, scenes.g32(), W, H, None)
    cc8, w8 = oracle.serialize(*scenes.g8())
    uu = _ffi.Uniforms.from_buffer_copy(bytes(u))
    ref32 = oracle.render(u, lim, cc32, w32, W, H, threads=8)
    ref8 = oracle.render(u, lim, cc8, w8, W, H, threads=8)
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
        r.set_limits(lim)
        r.set_uniforms(uu)
        s = torch.cuda.Stream()
        outs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in range(6)]
        for rep in range(3):
            r.set_program(cc32, w32)
            for o in outs[:3]:                               # three frames queued on the caller's stream
                r.draw_device(W, H, o.data_ptr(), stream=s.cuda_stream)
            r.set_program(cc8, w8)                           # program change + host draw while they are in flight
            host = r.draw(W, H)
            r.set_program(cc32, w32)
            for o in outs[3:]:
                r.draw_device(W, H, o.data_ptr(), stream=s.cuda_stream)
            s.synchronize()
            assert_same(host, ref8)
            for o in outs:
                assert_same(o.cpu().numpy(), ref32)
                o.zero_()
    finally:
        r.close()


def test_64k_command_buffer_falls_back_to_the_scalar_cache_variant(res, oracle):
    """rm_resize_command_buffer admits 64 KB of commands; such a program's records do not fit a workgroup's LDS next to
    the ray buffers, so the draw reads it through the scalar cache instead of failing."""
    rng = np.random.default_rng(5)
    t = scenes._Tab()
    n = 2700                                              # 2700 spheres + 2699 unions = 16 199 words < 16 383
    prims = [t.sphere((float(rng.uniform(-3, 3)), float(rng.uniform(-1, 1.5)), float(rng.uniform(-3, 3))), float(rng.uniform(0.02, 0.08)))
             for _ in range(n)]
    cc, w = oracle.serialize(t.nodes, scenes._fold_left(t, prims))
    assert 4 * (len(w) + 1) <= 65536 and len(w) > 16000
    W, H = 48, 32
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 64)
    r = renderer.RayMarchingResources(0)
    try:
        r.resize_command_buffer(65536)
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_program(cc, w)
        assert_same(r.draw(W, H), oracle.render(u, lim, cc, w, W, H, threads=16))
    finally:
        r.close()


def test_gather_strips_reassembles_the_frame_on_the_host(oracle):
    """rm_gather_strips: every rank's strips, copied from its compact device buffer to their rows of ONE host frame
    (here: all 'ranks' in one process, pinned and pageable destinations), give the single-GPU frame byte for byte."""
    torch = pytest.importorskip("torch")
    from ray_marching_amd import shard
    W, H = 200, 104                                       # 6.5 strips of 16 rows
    lim = (0.01, 100.0, 96)
    cc, w, u = oracle_case(oracle, scenes.g32(), W, H, None)
    ref = oracle.render(u, lim, cc, w, W, H, threads=8)
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_program(cc, w)
        s = torch.cuda.Stream()
        for world, sr in ((1, 16), (2, 16), (3, 16), (8, 16), (3, 8), (8, 8)):
            pinned = torch.zeros((H, W, 4), dtype=torch.float32).pin_memory()
            pageable = np.zeros((H, W, 4), np.float32)
            for rank in range(world):
                rows = shard.strip_row_count(H, sr, rank, world)
                buf = torch.zeros((max(rows, 1), W, 4), dtype=torch.float32, device="cuda")
                assert r.draw_strips_device(W, H, sr, rank, world, buf.data_ptr(), stream=s.cuda_stream) == rows
                r.gather_strips(W, H, sr, rank, world, buf.data_ptr(), pinned.data_ptr(), stream=s.cuda_stream)
                r.gather_strips(W, H, sr, rank, world, buf.data_ptr(), pageable.ctypes.data, stream=s.cuda_stream)
                s.synchronize()
            assert_same(pinned.numpy(), ref)
            assert_same(pageable, ref)
        with pytest.raises(_ffi.RmError):
            r.gather_strips(W, H, 12, 0, 1, 1, 1)         # strip_rows must be a multiple of 8
        # BASELINE config 4's row width (7680 px = 1.97 MB per strip, destination pitch 15.7 MB for 8 GPUs): the pitched copy
        # of every rank's strips against the frame one GPU renders
        W2, H2 = 7680, 272                                # 17 strips: rank 0 of 8 owns three, the last of them ...
        cc8, w8, u8 = oracle_case(oracle, scenes.g8(), W2, H2, None)
        r.set_limits((0.01, 100.0, 48))
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u8)))
        r.set_program(cc8, w8)
        whole = r.draw(W2, H2)
        for world, H3 in ((8, H2), (8, H2 - 8)):         # ... full, then ragged (8 rows)
            full3 = whole if H3 == H2 else r.draw(W2, H3)
            pinned = torch.zeros((H3, W2, 4), dtype=torch.float32).pin_memory()
            for rank in range(world):
                rows = shard.strip_row_count(H3, 16, rank, world)
                buf = torch.zeros((max(rows, 1), W2, 4), dtype=torch.float32, device="cuda")
                assert r.draw_strips_device(W2, H3, 16, rank, world, buf.data_ptr(), stream=s.cuda_stream) == rows
                r.gather_strips(W2, H3, 16, rank, world, buf.data_ptr(), pinned.data_ptr(), stream=s.cuda_stream)
                s.synchronize()
            assert_same(pinned.numpy(), full3)
    finally:
        r.close()
