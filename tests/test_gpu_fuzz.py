"""Seeded random CSG programs -- every node type, random nesting, degenerate parameters -- through the interpreter
kernel, the specialised kernel and the miss tests, against the oracle.  Exercises what the hand-written scenes do not:
arbitrary mixes of fused / unfused operators, spills, transform scopes around sub-trees, slack bookkeeping."""
import math

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer

pytestmark = pytest.mark.gpu


def random_tree(rng, t, depth, allow_plane, tags=False, lattice=False):
    """Returns a node index of a random sub-tree built into table t (tags: a quarter of the nodes get a material tag;
    lattice: bounded leaves and min / max operators only -- the programs far-primitive pruning applies to)."""
    node = _random_tree(rng, t, depth, allow_plane, tags, lattice)
    if tags and rng.random() < 0.25:
        node = t.material(node, int(rng.integers(0, 8)))
    return node


def _random_tree(rng, t, depth, allow_plane, tags, lattice):
    r = rng.random()
    if lattice and r >= 0.28 and r < 0.50:
        r = 0.9            # no transforms
    if depth == 0 or r < 0.28:
        kind = rng.integers(0, 4 if allow_plane else 3)
        c = rng.uniform(-1.6, 1.6, 3)
        if kind == 0:
            return t.sphere(tuple(c), float(rng.choice([rng.uniform(0.2, 0.8), 0.0, -0.3, 1e-4], p=[0.85, 0.05, 0.05, 0.05])))
        if kind == 1:
            return t.box(tuple(c), tuple(rng.uniform(0.1, 0.7, 3) * rng.choice([1.0, 0.0, -1.0], p=[0.9, 0.05, 0.05])))
        if kind == 2:
            return t.cylinder(tuple(c), float(rng.uniform(0.1, 0.5)), float(rng.uniform(0.1, 0.8)))
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        return t.plane(tuple(n), float(rng.uniform(0.5, 2.0)))
    if r < 0.50:   # a transform around a sub-tree
        child = random_tree(rng, t, depth - 1, allow_plane, tags, lattice)
        k = rng.integers(0, 3)
        if k == 0:
            return t.translation(child, tuple(rng.uniform(-0.8, 0.8, 3)))
        if k == 1:
            ax = rng.normal(size=3)
            ax /= np.linalg.norm(ax)
            ang = rng.uniform(-math.pi, math.pi)
            return t.rotation(child, (math.cos(ang / 2), *(math.sin(ang / 2) * ax)))
        return t.scale(child, float(rng.uniform(0.5, 1.8)))
    a = random_tree(rng, t, depth - 1, allow_plane, tags, lattice)
    b = random_tree(rng, t, depth - 1, allow_plane, tags, lattice)
    op = rng.choice(["u", "s", "i", "m"], p=[0.45, 0.25, 0.1, 0.2] if not lattice else [0.55, 0.3, 0.15, 0.0])
    if op == "m":
        return t.smooth_union(a, b, float(rng.choice([rng.uniform(0.05, 0.9), 0.0, -0.2], p=[0.9, 0.05, 0.05])))
    return t.op({"u": scenes.UNION, "s": scenes.SUBTRACTION, "i": scenes.INTERSECTION}[op], a, b)


import os

SEEDS = range(int(os.environ.get("RM_FUZZ_FIRST_SEED", "0")), int(os.environ.get("RM_FUZZ_FIRST_SEED", "0")) + int(os.environ.get("RM_FUZZ_SEEDS", "24")))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_programs_against_the_oracle(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    W, H = 48, 32
    res = renderer.RayMarchingResources(0)
    try:
        res.resize_command_buffer(8192)
        for k in range(3):
            t = scenes._Tab()
            tags = bool(rng.random() < 0.4)
            table = rng.uniform(0.0, 1.0, (8, 3)).astype(np.float32) if tags else None
            lattice = k == 2        # the third program of a seed is one the pruned kernel form applies to
            root = random_tree(rng, t, int(rng.integers(3, 6)) if lattice else int(rng.integers(1, 5)),
                               allow_plane=bool(rng.random() < 0.3) and not lattice, tags=tags, lattice=lattice)
            cc, w = oracle.serialize(t.nodes, root)
            rc, _ = oracle.validate(cc, w)
            prc, _ = renderer.validate_program(cc, w)
            assert rc == prc
            if rc != 0:      # e.g. nested more than 8 transforms deep / value stack too deep: both sides must agree
                continue
            events = [(1, float(rng.uniform(-300, 300)), float(rng.uniform(-140, 140))), (2, float(rng.uniform(-60, 150)), 0.0)]
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            lim = (float(rng.choice([0.01, 0.2])), 100.0, int(rng.choice([24, 64])))
            ref = oracle.render(u, lim, cc, w, W, H, threads=4, materials=table)
            res.set_materials(table if tags else [(0.4, 0.7, 0.1)])
            res.set_limits(lim)
            res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
            res.set_program(cc, w)
            # prune = 1: whichever skipping rule the program admits -- the threshold rule (lattice programs), the local rule
            # (programs that blend, without transforms), or none
            for spec, prune in ((0, 0), (2, 0), (2, 1)):
                res.set_option(_ffi.RM_OPT_SPECIALIZE, spec)
                res.set_option(_ffi.RM_OPT_PRUNE, prune)
                for cull in (0, 1):
                    res.set_option(_ffi.RM_OPT_CULL, cull)
                    img = res.draw(W, H)
                    if img.tobytes() != ref.tobytes():
                        bad = np.argwhere((img.view(np.uint32) != ref.view(np.uint32)).any(axis=-1))
                        raise AssertionError("seed %d: specialise=%d prune=%d cull=%d differs from the oracle at %d pixels (first %s); "
                                             "program: cmd_count %d words %s" % (seed, spec, prune, cull, len(bad), bad[:3].tolist(), cc,
                                                                               [int(x) for x in w]))
    finally:
        res.close()
