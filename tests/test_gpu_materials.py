"""Material tags + material table on the GPU (extension, see tests/test_materials_cpu.py): the interpreter kernels and
the specialised kernel against the oracle, bit for bit."""
import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer

pytestmark = pytest.mark.gpu

MODES = [("v5", _ffi.RM_KERNEL_V5, 0), ("v5_lds", _ffi.RM_KERNEL_V5_LDS, 0), ("v5_spec", _ffi.RM_KERNEL_DEFAULT, 2)]


@pytest.fixture(scope="module")
def res():
    r = renderer.RayMarchingResources(0)
    r.resize_command_buffer(8192)
    yield r
    r.close()


def select(res, mode):
    _, kernel, spec = mode
    res.set_option(_ffi.RM_OPT_KERNEL, kernel)
    res.set_option(_ffi.RM_OPT_SPECIALIZE, spec)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_tagged_scene_vs_oracle(res, oracle, mode):
    cc, w = oracle.serialize(*scenes.mat_mix())
    select(res, mode)
    res.set_materials(scenes.MATERIAL_TABLE)
    res.set_program(cc, w)
    for (W, H), events, lim in [((96, 64), scenes.STILL_CAMERA_EVENTS, (0.01, 100.0, 128)),
                                ((67, 45), [(1, 200.0, 60.0)], (0.01, 100.0, 96)),
                                ((64, 48), [(1, -80.0, -110.0), (2, -40.0, 0.0)], (0.05, 100.0, 64)),
                                ((40, 40), [(2, -95.0, 0.0)], (0.01, 100.0, 64))]:      # the last one: camera inside the solid
        u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
        ref = oracle.render(u, lim, cc, w, W, H, threads=4, materials=scenes.MATERIAL_TABLE)
        res.set_limits(lim)
        res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        for cull in (0, 1):
            res.set_option(_ffi.RM_OPT_CULL, cull)
            assert res.draw(W, H).tobytes() == ref.tobytes(), (events, cull)
        if mode[2]:
            assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1
        band = res.draw(W, H, 7, 19)
        assert band.tobytes() == ref[7:26].tobytes()
    res.set_option(_ffi.RM_OPT_CULL, 1)


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_deep_and_balanced_trees_with_tags(res, oracle, mode):
    """The material evaluation keeps (distance, index) pairs below the accumulator in LDS: right-deep chains 6 and 20
    deep with a tag on every leaf and on some operators, the balanced G32 tree with tags on sub-trees."""
    select(res, mode)
    table = [(0.4, 0.7, 0.1)] + [(0.025 * i, 1.0 - 0.025 * i, 0.5 + 0.01 * i) for i in range(1, 40)]
    res.set_materials(table)
    W, H = 56, 40
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 64)
    res.set_limits(lim)
    res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
    cases = []
    for n in (6, 20):
        t = scenes._Tab()
        prims = [t.material(t.sphere((0.7 * (i - (n - 1) / 2.0), 0.1 * (i % 3), 0.0), 0.3), 1 + i) for i in range(n)]
        acc = prims[-1]
        for i in range(n - 2, -1, -1):
            acc = t.op(scenes.UNION if i % 3 else scenes.SUBTRACTION, prims[i], acc)
            if i % 5 == 2:
                acc = t.material(acc, 25 + i % 5)
        cases.append((t.nodes, acc))
    t = scenes._Tab()
    level = scenes._grid_prims(t, 4, 4, 0x5DF00020)
    level = [t.material(p, 1 + i) if i % 3 else p for i, p in enumerate(level)]
    k = 0
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), 2):
            node = t.op(scenes.SUBTRACTION if k % 4 == 3 else scenes.UNION, level[i], level[i + 1])
            nxt.append(t.material(node, 20 + k) if k % 5 == 4 else node)
            k += 1
        level = nxt
    cases.append((t.nodes, level[0]))
    for nodes, root in cases:
        cc, w = oracle.serialize(nodes, root)
        ref = oracle.render(u, lim, cc, w, W, H, threads=4, materials=table)
        res.set_program(cc, w)
        assert res.draw(W, H).tobytes() == ref.tobytes()


def test_table_checks_and_untagged_programs(res, oracle):
    W, H = 48, 32
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = (0.01, 100.0, 64)
    select(res, MODES[2])
    res.set_limits(lim)
    res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
    cc, w = oracle.serialize(*scenes.mat_mix())
    res.set_program(cc, w)
    res.set_materials(scenes.MATERIAL_TABLE[:5])                 # the scene names material 5
    with pytest.raises(_ffi.RmError) as e:
        res.draw(W, H)
    assert e.value.status == _ffi.RM_ERR_MATERIAL
    res.set_materials(scenes.MATERIAL_TABLE)
    assert res.draw(W, H).tobytes() == oracle.render(u, lim, cc, w, W, H, threads=4, materials=scenes.MATERIAL_TABLE).tobytes()
    with pytest.raises(_ffi.RmError) as e:
        res.set_materials([(0.1, 0.2, 0.3)] * 257)
    assert e.value.status == _ffi.RM_ERR_MATERIAL
    # a program without tags is shaded with the reference colour whatever the table says
    cc8, w8 = oracle.serialize(*scenes.g8())
    res.set_materials([(0.9, 0.9, 0.9)] * 4)
    res.set_program(cc8, w8)
    assert res.draw(W, H).tobytes() == oracle.render(u, lim, cc8, w8, W, H, threads=4).tobytes()
    # the v1 kernel renders reference node types only
    res.set_program(cc, w)
    res.set_materials(scenes.MATERIAL_TABLE)
    res.set_option(_ffi.RM_OPT_SPECIALIZE, 0)
    for k in (_ffi.RM_KERNEL_PIXEL,):
        res.set_option(_ffi.RM_OPT_KERNEL, k)
        with pytest.raises(_ffi.RmError) as e:
            res.draw(W, H)
        assert e.value.status == _ffi.RM_ERR_ARG
    res.set_option(_ffi.RM_OPT_KERNEL, _ffi.RM_KERNEL_DEFAULT)


def test_table_writes_are_ordered_with_the_draws(oracle):
    """A table written for frame n+1 must not recolour frame n that is still in flight (same rule as the program);
    also: 8-bit output, batches and strips of a tagged scene."""
    import torch
    W, H = 512, 288
    lim = (0.01, 100.0, 96)
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    cc, w = oracle.serialize(*scenes.mat_mix())
    tables = [scenes.MATERIAL_TABLE, [tuple(reversed(m)) for m in scenes.MATERIAL_TABLE], [(0.5, 0.5, 0.5)] * 6]
    refs = [oracle.render(u, lim, cc, w, W, H, threads=4, materials=t).tobytes() for t in tables]
    r = renderer.RayMarchingResources(0)
    try:
        r.set_limits(lim)
        r.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
        r.set_option(_ffi.RM_OPT_SPECIALIZE, 2)
        r.set_program(cc, w)
        r.set_materials(tables[0])
        r.draw(W, H)                                          # compile outside of the pipelined part
        order = [0, 1, 2, 0, 2, 1]
        bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device="cuda") for _ in order]
        for b, k in zip(bufs, order):
            r.set_materials(tables[k])
            r.draw_device(W, H, b.data_ptr(), stream=_ffi.RM_STREAM_OWN)
        r.sync_context()
        for n, (b, k) in enumerate(zip(bufs, order)):
            assert b.cpu().numpy().tobytes() == refs[k], (n, k)
        # 8-bit output stage
        ref0 = np.frombuffer(refs[0], dtype=np.float32).reshape(H, W, 4)
        r.set_materials(tables[0])
        r.set_output_format(_ffi.RM_FORMAT_RGBA8_UNORM)
        assert r.draw(W, H).tobytes() == oracle.quantize_unorm8(ref0).tobytes()
        r.set_output_format(_ffi.RM_FORMAT_RGBA32F)
        # interleaved strips (multi-GPU tiling) of the tagged scene reassemble to the frame
        from ray_marching_amd import shard
        parts = [r.draw_strips(W, H, 16, k, 3) for k in range(3)]
        full = np.empty((H, W, 4), np.float32)
        for k, part in enumerate(parts):
            shard.scatter_strips(full, part, H, k, 3, strip_rows=16)
        assert full.tobytes() == refs[0]
    finally:
        r.close()
