"""Differential test of the miss tests: the same frame with culling on and with culling off must be the same bytes (culling off
marches every ray, which is what the oracle does; the oracle comparison itself is in test_gpu_parity.py / test_gpu_fuzz.py).  GPU
against GPU, so it can afford what the oracle cannot: larger frames, hundreds of random programs and cameras.  The programs are
left-deep chains of spheres and boxes (in two programs of five also cylinders and planes) under SmoothUnion / Union /
Subtraction / Intersection -- the shapes the structure-aware
tests act on (subtracted primitives out of the tables, the program run on lower bounds) -- plus right operands that are
sub-trees.  RM_CULL_SEEDS=N runs N programs (default 40)."""
import os

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer

pytestmark = pytest.mark.gpu

N = int(os.environ.get("RM_CULL_SEEDS", "40"))


def random_leaf(rng, t, spread, extended=False):
    c = rng.uniform(-spread, spread, 3)
    if extended and rng.random() < 0.25:
        if rng.random() < 0.5:
            return t.cylinder(tuple(c), float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.1, 0.8)))
        n = rng.normal(size=3) * rng.choice([1.0, 0.3, 2.5])          # |n| need not be 1
        return t.plane(tuple(n), float(rng.uniform(0.3, 2.5)))
    if rng.random() < 0.5:
        return t.sphere(tuple(c), float(rng.choice([rng.uniform(0.15, 0.7), 0.0, -0.2], p=[0.9, 0.05, 0.05])))
    h = rng.uniform(0.08, 0.7, 3)
    if rng.random() < 0.15:
        h[rng.integers(3)] = rng.choice([0.0, -0.1])
    return t.box(tuple(c), tuple(h))


def random_program(rng):
    t = scenes._Tab()
    spread = float(rng.choice([1.0, 1.8, 3.0]))
    blend = rng.random() < 0.7
    extended = rng.random() < 0.4       # cylinders and planes among the leaves
    acc = random_leaf(rng, t, spread)
    for _ in range(int(rng.integers(1, 18))):
        if rng.random() < 0.15:     # a sub-tree as right operand
            right = t.op(scenes.UNION if rng.random() < 0.6 else scenes.SUBTRACTION, random_leaf(rng, t, spread, extended), random_leaf(rng, t, spread, extended))
        else:
            right = random_leaf(rng, t, spread, extended)
        r = rng.random()
        if blend and r < 0.5:
            acc = t.smooth_union(acc, right, float(rng.choice([rng.uniform(0.02, 1.0), 0.0, -0.3], p=[0.9, 0.05, 0.05])))
        elif r < 0.7:
            acc = t.op(scenes.UNION, acc, right)
        elif r < 0.92:
            acc = t.op(scenes.SUBTRACTION, acc, right)
        else:
            acc = t.op(scenes.INTERSECTION, acc, right)
    return t.nodes, acc


@pytest.mark.parametrize("block", range((N + 9) // 10))
def test_culling_changes_no_pixel(oracle, block):
    W, H = 160, 96
    res = renderer.RayMarchingResources(0)
    try:
        res.resize_command_buffer(8192)
        for seed in range(block * 10, min(N, block * 10 + 10)):
            rng = np.random.default_rng(77000 + seed)
            nodes, root = random_program(rng)
            cc, w = oracle.serialize(nodes, root)
            info = renderer.program_info(cc, w)
            res.set_program(cc, w)
            for cam in range(3):
                events = [(1, float(rng.uniform(-300, 300)), float(rng.uniform(-140, 140))), (2, float(rng.uniform(-90, 120)), 0.0)]
                if cam == 2:
                    events.append((0, float(rng.uniform(-150, 150)), float(rng.uniform(-80, 80))))     # pan: off-axis views
                u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
                res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
                res.set_limits((float(rng.choice([0.01, 0.002, 0.15])), float(rng.choice([100.0, 6.0])), int(rng.choice([48, 128]))))
                for spec in (2, 0):
                    res.set_option(_ffi.RM_OPT_SPECIALIZE, spec)
                    res.set_option(_ffi.RM_OPT_CULL, 0)
                    off = res.draw(W, H)
                    res.set_option(_ffi.RM_OPT_CULL, 1)
                    on = res.draw(W, H)
                    if on.tobytes() != off.tobytes():
                        bad = np.argwhere((on.view(np.uint32) != off.view(np.uint32)).any(axis=-1))
                        raise AssertionError("seed %d camera %d specialise=%d: culling changes %d pixels (first %s); decoder: %s; program: cmd_count %d "
                                             "words %s; events %s" % (seed, cam, spec, len(bad), bad[:3].tolist(), info, cc, [int(x) for x in w], events))
    finally:
        res.close()
