"""Structure specialisation (RM_OPT_SPECIALIZE, csrc/rm_jit.h): the hipRTC-compiled straight-line march
kernel must be indistinguishable from the interpreter kernel and the oracle -- bit for bit -- and the
machinery around it (cache per structure, background compilation, fallbacks) must behave as documented."""
import time

import os

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer

pytestmark = pytest.mark.gpu

LIM = (0.01, 100.0, 96)


@pytest.fixture(scope="module")
def res():
    r = renderer.RayMarchingResources(0)
    yield r
    r.close()


def case(oracle, scene, W, H):
    cc, w = oracle.serialize(*scene)
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    return cc, w, u


def load(res, cc, w, u, mode, prune=0):
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)
    res.set_option(_ffi.RM_OPT_SPECIALIZE, mode)
    res.set_option(_ffi.RM_OPT_PRUNE, prune)
    res.set_limits(LIM)
    res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
    res.set_program(cc, w)


ALL = dict(scenes.SCENES)
ALL.update(scenes.EXT_SCENES)


@pytest.mark.parametrize("name", sorted(ALL))
def test_specialised_equals_interpreter_equals_oracle(res, oracle, name):
    W, H = 96, 64
    cc, w, u = case(oracle, ALL[name](), W, H)
    res.resize_command_buffer(4096)
    load(res, cc, w, u, 0)
    interp = res.draw(W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0
    load(res, cc, w, u, 2)
    spec = res.draw(W, H)
    assert res.info(_ffi.RM_INFO_JIT_STATE) == 2, res.jit_log()
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
    assert res.info(_ffi.RM_INFO_JIT_COMPILE_MS) > 0
    load(res, cc, w, u, 2, prune=1)
    pruned = res.draw(W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1, res.jit_log()
    ref = oracle.render(u, LIM, cc, w, W, H, threads=4)
    assert spec.tobytes() == ref.tobytes()
    assert pruned.tobytes() == ref.tobytes()
    assert interp.tobytes() == ref.tobytes()


def test_moving_parameters_keeps_the_compiled_kernel(res, oracle):
    """The cache key is the structure: an animation that moves/resizes primitives must not recompile."""
    W, H = 64, 48
    nodes, root = scenes.g8()
    cc, w, u = case(oracle, (nodes, root), W, H)
    load(res, cc, w, u, 2)
    res.draw(W, H)
    ms0 = res.info(_ffi.RM_INFO_JIT_COMPILE_MS)
    assert ms0 > 0
    for step in range(1, 4):
        moved = [(k, ([p[0] + 0.1 * step, p[1], p[2] - 0.05 * step] + [x * (1.0 + 0.1 * step) for x in p[3:]])
                  if k in (0, 1) else p, l, r) for (k, p, l, r) in nodes]
        cc2, w2 = oracle.serialize(moved, root)
        res.set_program(cc2, w2)
        t0 = time.perf_counter()
        img = res.draw(W, H)
        dt = time.perf_counter() - t0
        assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
        assert res.info(_ffi.RM_INFO_JIT_COMPILE_MS) == ms0        # the same cache entry
        assert dt < 0.2                                             # no compiler run in this draw
        assert img.tobytes() == oracle.render(u, LIM, cc2, w2, W, H, threads=4).tobytes()


def test_background_mode_draws_correctly_before_and_after_the_switch(res, oracle):
    """Mode 1: frames keep coming from the interpreter kernel until the compiled kernel is ready."""
    W, H = 64, 48
    t = scenes._Tab()   # a structure no other test uses, so that this is a cache miss
    parts = [t.sphere((0.3 * i - 1.0, 0.1 * i, 0.0), 0.4) for i in range(7)] + [t.box((0, -1, 0), (2, 0.1, 2))]
    root = scenes._fold_left(t, parts)
    cc, w, u = case(oracle, (t.nodes, root), W, H)
    ref = oracle.render(u, LIM, cc, w, W, H, threads=4)
    load(res, cc, w, u, 1)
    deadline = time.time() + 60.0
    switched = False
    n = 0
    while time.time() < deadline:
        img = res.draw(W, H)
        n += 1
        assert img.tobytes() == ref.tobytes()
        if res.info(_ffi.RM_INFO_SPECIALIZED) == 1:
            switched = True
            break
        assert res.info(_ffi.RM_INFO_JIT_STATE) in (1, 2), res.jit_log()
        time.sleep(0.01)
    assert switched, "compiled kernel never became ready: %s" % res.jit_log()
    assert res.draw(W, H).tobytes() == ref.tobytes()


def test_programs_that_are_not_specialised(res, oracle):
    W, H = 40, 32
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    # empty program (csg_node == None)
    load(res, 0, np.zeros(0, np.uint32), u, 2)
    assert res.draw(W, H).tobytes() == oracle.render(u, LIM, 0, [], W, H, threads=2).tobytes()
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0 and res.info(_ffi.RM_INFO_JIT_STATE) == 0
    # more than 255 records: the interpreter kernel
    t = scenes._Tab()
    parts = [t.sphere((0.02 * i - 2.6, 0.3 * ((i * 7) % 5) - 0.6, 0.1 * ((i * 3) % 7) - 0.3), 0.15) for i in range(260)]
    root = scenes._fold_left(t, parts)
    cc, w = oracle.serialize(t.nodes, root)
    res.resize_command_buffer(16384)
    load(res, cc, w, u, 2)
    assert res.draw(W, H).tobytes() == oracle.render(u, LIM, cc, w, W, H, threads=4).tobytes()
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0
    # the explicitly selected scalar-cache variant has no specialised form
    cc, w, u = case(oracle, scenes.g8(), W, H)
    load(res, cc, w, u, 2)
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_V5)
    assert res.draw(W, H).tobytes() == oracle.render(u, LIM, cc, w, W, H, threads=2).tobytes()
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 0
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)
    res.resize_command_buffer(1024)
    with pytest.raises(_ffi.RmError):
        res.set_option(_ffi.RM_OPT_SPECIALIZE, 3)


def test_deep_and_long_programs_specialised(res, oracle):
    """Operands live in registers in the generated code: a 32-deep stack and a 127-record program."""
    W, H = 48, 32
    res.resize_command_buffer(4096)
    for scene in (scenes.right_deep(32), scenes.g64(), scenes.g32_balanced()):
        cc, w, u = case(oracle, scene, W, H)
        load(res, cc, w, u, 2)
        img = res.draw(W, H)
        assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1, res.jit_log()
        assert img.tobytes() == oracle.render(u, LIM, cc, w, W, H, threads=4).tobytes()
    res.resize_command_buffer(1024)


def test_waves_per_tile_variants_and_batch(res, oracle):
    W, H = 64, 40
    cc, w, u = case(oracle, scenes.g32(), W, H)
    ref = oracle.render(u, LIM, cc, w, W, H, threads=4)
    load(res, cc, w, u, 2)
    for wpt in (1, 2, 8, 4):
        res.set_option(_ffi.RM_OPT_WAVES_PER_TILE, wpt)
        assert res.draw(W, H).tobytes() == ref.tobytes()
        assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
    frames = []
    for ev in ([(1, 35.0, -25.0)], [(1, 80.0, -10.0), (2, 40.0, 0.0)], [(1, -60.0, 30.0)]):
        uu, *_ = oracle.orbit_uniforms((float(W), float(H)), events=ev)
        frames.append(uu)
    batch = res.draw_batch([_ffi.Uniforms.from_buffer_copy(bytes(f)) for f in frames], W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
    for i, f in enumerate(frames):
        assert batch[i].tobytes() == oracle.render(f, LIM, cc, w, W, H, threads=4).tobytes()
    with pytest.raises(_ffi.RmError):
        res.set_option(_ffi.RM_OPT_WAVES_PER_TILE, 3)
    res.set_option(_ffi.RM_OPT_WAVES_PER_TILE, 0)      # back to the default: by the size of the launch (8 here, 4 for a full frame)
    assert res.draw(W, H).tobytes() == ref.tobytes()


def test_full_size_metric_config_specialised_vs_interpreter(res, oracle):
    """BASELINE.json's metric configuration (1920x1080, 32 nodes, 256 steps): whole-frame identity of the two
    kernels plus sampled row bands against the oracle."""
    W, H = 1920, 1080
    lim = (0.01, 100.0, 256)
    cc, w, u = case(oracle, scenes.g32(), W, H)
    load(res, cc, w, u, 0)
    res.set_limits(lim)
    a = res.draw(W, H)
    load(res, cc, w, u, 2)
    res.set_limits(lim)
    b = res.draw(W, H)
    assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
    assert a.tobytes() == b.tobytes()
    load(res, cc, w, u, 2, prune=1)
    res.set_limits(lim)
    assert res.draw(W, H).tobytes() == a.tobytes()
    for r0, rows in [(200, 3), (540, 4), (901, 3)]:
        assert b[r0:r0 + rows].tobytes() == oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=8).tobytes()


def test_pruning_policy_by_primitive_count(oracle):
    """RM_OPT_PRUNE = 2 (the default): the pruned form of the generated kernel for programs with at least 12 spheres +
    boxes (G32: 16, G64: 32), the plain form below that (G8: 4); 0 / 1 force it; every form renders the oracle's image."""
    W, H = 64, 40
    r = renderer.RayMarchingResources(0)
    try:
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_limits(LIM)
        # g8x (4 leaves, a blend): the local rule of blending programs (RM_INFO_PRUNED = 2) only when forced; g32s (16): by default
        for name, want in (("g8", {2: 0, 1: 1, 0: 0}), ("g32", {2: 1, 1: 1, 0: 0}), ("g64", {2: 1, 1: 1, 0: 0}),
                           ("g32s", {2: 2, 1: 2, 0: 0}), ("ext_mix", {2: 0, 1: 2, 0: 0})):
            cc, w, u = case(oracle, ALL[name](), W, H)
            ref = oracle.render(u, LIM, cc, w, W, H, threads=4)
            r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
            r.set_program(cc, w)
            for mode in (2, 1, 0):           # a fresh context starts at 2
                if mode != 2:
                    r.set_option(_ffi.RM_OPT_PRUNE, mode)
                assert r.draw(W, H).tobytes() == ref.tobytes(), (name, mode)
                assert r.info(_ffi.RM_INFO_SPECIALIZED) == 1 and r.info(_ffi.RM_INFO_PRUNED) == want[mode], (name, mode)
            r.set_option(_ffi.RM_OPT_PRUNE, 2)
        with pytest.raises(_ffi.RmError):
            r.set_option(_ffi.RM_OPT_PRUNE, 3)
    finally:
        r.close()


@pytest.mark.parametrize("form", ["generated", "interpreter_lds", "interpreter_scalar_cache"])
def test_grouped_far_tests_far_from_the_origin_and_with_distant_partners(oracle, form):
    """The pruned form tests pairs of consecutive sphere / box leaves against a bounding sphere first (rm_decode.h:
    RmDecoded::groups).  Scenes that stress the bound: everything 1000 units from the origin (the members' own
    evaluation error is then ~1e-4), partners far apart (a huge group sphere), partners that coincide, degenerate radii,
    an odd number of leaves; and the same kernel after the parameters moved (group spheres follow the upload).  The
    interpreter kernels run the same pair tests in their chain loop (rm_interp.h map_scene_chain_pruned): variants 0-2 are
    chains, variant 3 (an Intersection in the middle) takes the general loop."""
    rng = np.random.default_rng(77)
    W, H = 64, 40
    lim = (0.01, 100.0, 80)
    r = renderer.RayMarchingResources(0)
    try:
        if form == "generated":
            r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
            r.set_option(_ffi.RM_OPT_PRUNE, 1)
        else:
            r.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
            r.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_V5_LDS if form == "interpreter_lds" else _ffi.RM_KERNEL_V5)
        r.set_limits(lim)
        r.resize_command_buffer(4096)
        for offset in ((0.0, 0.0, 0.0), (800.0, -300.0, 500.0)):
            for variant in range(4):
                t = scenes._Tab()
                leaves = []
                for k in range(13):
                    c = rng.uniform(-1.8, 1.8, 3) + np.array(offset)
                    if variant == 1 and k % 2 == 1:
                        c = c + rng.uniform(-40.0, 40.0, 3)            # the partner is far away
                    if variant == 2 and k % 2 == 1:
                        c = np.array(t.nodes[leaves[-1]][1][:3])        # the partner sits at the same place
                    if rng.random() < 0.5:
                        leaves.append(t.sphere(tuple(c), float(rng.choice([rng.uniform(0.2, 0.7), 0.0, -0.2], p=[0.8, 0.1, 0.1]))))
                    else:
                        leaves.append(t.box(tuple(c), tuple(rng.uniform(0.1, 0.6, 3))))
                acc = leaves[0]
                for k, leaf in enumerate(leaves[1:]):
                    acc = t.op(scenes.SUBTRACTION if k % 5 == 4 else scenes.INTERSECTION if variant == 3 and k % 4 == 1 else scenes.UNION, acc, leaf)
                cc, w = oracle.serialize(t.nodes, acc)
                for events in (scenes.STILL_CAMERA_EVENTS, [(1, 140.0, 40.0), (2, -30.0, 0.0)]):
                    u, *_ = oracle.orbit_uniforms((float(W), float(H)), target=offset, events=events)
                    r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
                    r.set_program(cc, w)
                    assert r.draw(W, H).tobytes() == oracle.render(u, lim, cc, w, W, H, threads=4).tobytes(), (offset, variant, events)
                    if form == "generated":
                        assert r.info(_ffi.RM_INFO_SPECIALIZED) == 1 and r.info(_ffi.RM_INFO_PRUNED) == 1
                    else:
                        assert r.info(_ffi.RM_INFO_SPECIALIZED) == 0
                        # chains: the stack-free loop, over the wave's unit mask where the unit records are at hand (LDS)
                        # chains (13 leaves): the stack-free loop, over the wave's unit mask where the unit records are at hand (LDS);
                        # variant 3 has an Intersection: an extension node, the general loop
                        assert r.info(_ffi.RM_INFO_INTERPRETER_LOOP) == (0 if variant == 3 else 2 if form == "interpreter_lds" else 1), variant
    finally:
        r.close()
