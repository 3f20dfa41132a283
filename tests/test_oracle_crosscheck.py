"""The two independently written restatements (C and numpy) must agree bit for bit."""
import numpy as np
import pytest

import scenes
from oracle import rm_oracle_np as onp


def _udict(u):
    return {"viewport_extent": list(u.viewport_extent), "inv_proj": list(u.inv_proj), "inv_view": list(u.inv_view)}


@pytest.mark.parametrize("name,W,H", [("g1", 48, 40), ("g8", 40, 32), ("g32", 32, 24), ("g32_balanced", 24, 16)])
def test_c_vs_numpy_render(oracle, name, W, H):
    cc, w = oracle.serialize(*scenes.SCENES[name]())
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    lim = scenes.LIMITS[name]
    a = oracle.render(u, lim, cc, w, W, H)
    b = onp.render(_udict(u), lim, cc, w, W, H)
    assert a.tobytes() == b.tobytes()


def test_c_vs_numpy_empty_and_band(oracle):
    W, H = 40, 30
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    a = oracle.render(u, (0.01, 100.0, 9), 0, [], W, H, row0=7, rows=11)
    b = onp.render(_udict(u), (0.01, 100.0, 9), 0, [], W, H, row0=7, rows=11)
    assert a.tobytes() == b.tobytes()
    full = oracle.render(u, (0.01, 100.0, 9), 0, [], W, H)
    assert full[7:18].tobytes() == a.tobytes()


def test_c_vs_numpy_map_scene_random_points(oracle):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-3, 3, size=(500, 3)).astype(np.float32)
    for name in ("g8", "g32", "g64", "g32_balanced"):
        cc, w = oracle.serialize(*scenes.SCENES[name]())
        got = onp.map_scene(cc, w, 100.0, pts[:, 0], pts[:, 1], pts[:, 2])
        exp = np.array([oracle.map_scene(cc, w, p) for p in pts], dtype=np.float32)
        assert got.tobytes() == exp.tobytes()


def test_minmax_signed_zero_and_nan():
    z, nz, nan = np.float32(0.0), np.float32(-0.0), np.float32(np.nan)
    assert np.signbit(onp.fmin(z, nz)) and np.signbit(onp.fmin(nz, z))
    assert not np.signbit(onp.fmax(z, nz)) and not np.signbit(onp.fmax(nz, z))
    assert onp.fmin(nan, np.float32(1)) == 1 and onp.fmax(np.float32(1), nan) == 1


def test_mt_equals_single_thread(oracle):
    cc, w = oracle.serialize(*scenes.g8())
    u, *_ = oracle.orbit_uniforms((64.0, 48.0), events=scenes.STILL_CAMERA_EVENTS)
    a, ca = oracle.render(u, (0.01, 100.0, 128), cc, w, 64, 48, want_counters=True)
    b, cb = oracle.render(u, (0.01, 100.0, 128), cc, w, 64, 48, threads=4, want_counters=True)
    assert a.tobytes() == b.tobytes() and ca == cb
    assert ca["rays"] == 64 * 48 * 16 and ca["normal_taps"] == 4 * ca["hits"]
    assert ca["hits"] + ca["floor_hits"] + ca["sky"] == ca["rays"]


def test_random_programs_c_equals_numpy(oracle):
    """The C oracle and the independent numpy restatement on seeded random trees of every node type (generator shared
    with tests/test_gpu_fuzz.py): map_scene bit for bit at random points."""
    import numpy as np
    from oracle import rm_oracle_np as onp
    import scenes
    from test_gpu_fuzz import random_tree
    rng = np.random.default_rng(77)
    checked = 0
    for _ in range(60):
        t = scenes._Tab()
        root = random_tree(rng, t, int(rng.integers(1, 5)), allow_plane=bool(rng.random() < 0.3))
        cc, w = oracle.serialize(t.nodes, root)
        if oracle.validate(cc, w)[0] != 0:
            continue
        P = rng.uniform(-3, 3, size=(40, 3)).astype(np.float32)
        with np.errstate(all="ignore"):
            got = onp.map_scene(cc, w, 100.0, P[:, 0].copy(), P[:, 1].copy(), P[:, 2].copy())
        for i in range(len(P)):
            want = np.float32(oracle.map_scene(cc, w, [float(P[i, 0]), float(P[i, 1]), float(P[i, 2])]))
            a, b = np.float32(got[i]), want
            assert (np.isnan(a) and np.isnan(b)) or a.tobytes() == b.tobytes(), (cc, list(map(int, w)), P[i].tolist(), a, b)
        checked += 1
    assert checked >= 40
