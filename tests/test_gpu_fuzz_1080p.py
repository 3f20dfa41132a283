"""Random programs x random cameras at 1920x1080 with the LIBRARY DEFAULTS, against the oracle.

The small-frame fuzz tests (test_gpu_fuzz.py: 48x32, test_gpu_cull_differential.py: 160x96) never see what only happens at full
size: pixel cones whose rho is a thousandth of a radian, tiles of pure sky and of one checker cell finished by the pre-pass,
the work list built by the last pre-pass workgroup, 8x8 tiles that lie entirely inside one primitive's far zone, the temporal
tile order of a second draw of the same shape.  Here every frame is drawn twice at 1920x1080 with nothing switched off
(specialised kernel, automatic pruning, culling, balance 3) and sampled row bands -- 64 rows per frame, tile-aligned and not --
are compared with the oracle bit for bit.  Programs: lattice trees and chains (min / max / subtraction over spheres and boxes:
the pruned kernels), and chains that blend with SmoothUnion (the lower-bound walk, the blend-aware pruning).
RM_FUZZ1080_SEEDS=N runs N programs (default 16)."""
import os

import numpy as np
import pytest

import scenes
from ray_marching_amd import _ffi, renderer
from test_gpu_cull_differential import random_leaf
from test_gpu_fuzz import random_tree

pytestmark = pytest.mark.gpu

N = int(os.environ.get("RM_FUZZ1080_SEEDS", "16"))
FIRST = int(os.environ.get("RM_FUZZ1080_FIRST_SEED", "0"))
W, H = 1920, 1080


def blended_chain(rng):
    """12-20 spheres and boxes folded left-deep under SmoothUnion (mostly), Union and Subtraction."""
    t = scenes._Tab()
    spread = float(rng.choice([1.2, 1.8, 2.6]))
    acc = random_leaf(rng, t, spread)
    k_common = float(rng.uniform(0.05, 0.5))
    for _ in range(int(rng.integers(11, 20))):
        right = random_leaf(rng, t, spread)
        r = rng.random()
        if r < 0.6:
            k = k_common if rng.random() < 0.7 else float(rng.choice([rng.uniform(0.02, 0.9), 0.0, -0.3], p=[0.9, 0.05, 0.05]))
            acc = t.smooth_union(acc, right, k)
        elif r < 0.8:
            acc = t.op(scenes.UNION, acc, right)
        else:
            acc = t.op(scenes.SUBTRACTION, acc, right)
    return t.nodes, acc


def lattice_program(rng):
    """A random min / max tree (depth 4-5: 12-32 leaves typically) or a left-deep chain of 12-24 leaves."""
    t = scenes._Tab()
    if rng.random() < 0.5:
        for _ in range(20):
            t = scenes._Tab()
            root = random_tree(rng, t, int(rng.integers(4, 6)), allow_plane=False, tags=False, lattice=True)
            if sum(1 for n in t.nodes if n[0] in (scenes.SPHERE, scenes.BOX)) >= 12:
                return t.nodes, root
        t = scenes._Tab()           # no tree with a dozen leaves came up: a chain then
    spread = float(rng.choice([1.2, 1.8, 2.6]))
    acc = random_leaf(rng, t, spread)
    for _ in range(int(rng.integers(11, 24))):
        acc = t.op(scenes.UNION if rng.random() < 0.7 else scenes.SUBTRACTION, acc, random_leaf(rng, t, spread))
    return t.nodes, acc


def row_bands(rng):
    """64 rows: eight bands of 8 rows, half of them aligned with the kernels' 8-row tiles, half straddling two."""
    bands = []
    for k in range(8):
        r0 = int(rng.integers(0, H - 8))
        r0 = (r0 // 8) * 8 + (0 if k % 2 == 0 else int(rng.integers(1, 8)))
        bands.append((min(r0, H - 8), 8))
    return bands


@pytest.mark.parametrize("block", range((N + 3) // 4))
def test_random_programs_and_cameras_at_1080p_with_library_defaults(oracle, block):
    res = renderer.RayMarchingResources(0)          # defaults: specialise in the background, prune by leaf count, cull, balance 3
    try:
        res.set_option(_ffi.RM_OPT_SPECIALIZE, 2)    # ... but wait for the compiler, so that the specialised kernel is what is tested
        res.resize_command_buffer(8192)
        for seed in range(FIRST + block * 4, FIRST + min(N, block * 4 + 4)):
            rng = np.random.default_rng(424200 + seed)
            blended = seed % 2 == 1
            nodes, root = blended_chain(rng) if blended else lattice_program(rng)
            cc, w = oracle.serialize(nodes, root)
            rc, _ = oracle.validate(cc, w)
            if rc != 0:
                continue
            info = renderer.program_info(cc, w)
            events = [(1, float(rng.uniform(-314, 314)), float(rng.uniform(-140, 60))), (2, float(rng.uniform(-40, 120)), 0.0)]
            if rng.random() < 0.3:
                events.append((0, float(rng.uniform(-120, 120)), float(rng.uniform(-60, 60))))     # pan: off-axis view
            u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=events)
            lim = (0.01, 100.0, int(rng.choice([64, 128, 256])))
            res.set_limits(lim)
            res.set_uniforms(_ffi.Uniforms.from_buffer_copy(bytes(u)))
            res.set_program(cc, w)
            first = res.draw(W, H)
            second = res.draw(W, H)                  # same shape again: tiles dispatched by the durations the first draw measured
            assert res.info(_ffi.RM_INFO_SPECIALIZED) == 1, res.jit_log()
            assert second.tobytes() == first.tobytes(), "seed %d: the second draw of the same frame differs from the first" % seed
            assert np.array_equal(first[..., 3], np.ones((H, W), np.float32))
            for r0, rows in row_bands(rng):
                ref = oracle.render(u, lim, cc, w, W, H, row0=r0, rows=rows, threads=16)
                got = first[r0:r0 + rows]
                if got.tobytes() != ref.tobytes():
                    bad = np.argwhere((got.view(np.uint32) != ref.view(np.uint32)).any(axis=-1))
                    raise AssertionError("seed %d (%s): rows %d..%d differ from the oracle at %d pixels (first %s, max abs diff %g); "
                                         "decoder: %s; limits %s; events %s; program: cmd_count %d words %s"
                                         % (seed, "blended" if blended else "lattice", r0, r0 + rows, len(bad), bad[:3].tolist(),
                                            float(np.nanmax(np.abs(got.astype(np.float64) - ref))), info, lim, events, cc, [int(x) for x in w]))
    finally:
        res.close()
