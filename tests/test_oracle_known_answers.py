"""Pins the C oracle to the hand-derivable known answers of SURVEY.md section 8(c).
The reference ships no tests or golden vectors, so these are the only pins there are
("parity unpinned" against the real wgpu render)."""
import math
import struct

import numpy as np
import pytest

import scenes

F = np.float32


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_blob_sizes(oracle):
    assert oracle.lib().rmo_sizeof_uniforms() == 144   # renderer.rs:29-34 via encase
    assert oracle.lib().rmo_sizeof_limits() == 12      # renderer.rs:36-41


def test_serializer_sphere(oracle):
    cc, w = oracle.serialize(*scenes.g1())
    assert cc == 1
    assert list(w) == [0, 0, 0, 0, 0x3F800000]


def test_serializer_postorder(oracle):
    t = scenes._Tab()
    s = t.sphere((0, 0, 0), 1.0)
    b = t.box((0, 0, 0), (1, 1, 1))
    s2 = t.sphere((1, 0, 0), 0.5)
    root = t.op(scenes.SUBTRACTION, t.op(scenes.UNION, s, b), s2)
    cc, w = oracle.serialize(t.nodes, root)
    assert cc == 5 and len(w) == 19
    assert [int(w[0]), int(w[5]), int(w[12]), int(w[13]), int(w[18])] == [0, 1, 100, 0, 101]


@pytest.mark.parametrize("name,n,nwords", [("g1", 1, 5), ("g8", 7, 27), ("g32", 31, 111), ("g64", 63, 223),
                                            ("g32_balanced", 31, 111)])
def test_scene_sizes(oracle, name, n, nwords):
    cc, w = oracle.serialize(*scenes.SCENES[name]())
    assert (cc, len(w)) == (n, nwords)
    rc, depth = oracle.validate(cc, w, strict=True)
    assert rc == 0
    assert depth == {"g1": 1, "g8": 2, "g32": 2, "g64": 2, "g32_balanced": 5}[name]


def test_sdf_known_values(oracle):
    cc, w = oracle.serialize(*scenes.g1())
    assert oracle.map_scene(cc, w, [0, 0, 5]) == 4.0
    t = scenes._Tab()
    cc, w = oracle.serialize(t.nodes, t.box((0, 0, 0), (1, 1, 1)))
    assert oracle.map_scene(cc, w, [2, 0, 0]) == 1.0
    assert oracle.map_scene(cc, w, [0, 0, 0]) == -1.0
    assert oracle.map_scene(cc, w, [2, 2, 0]) == float(np.sqrt(F(2)))


def test_union_subtract(oracle):
    # union(4, 1) = 1 : sphere r=1 at origin seen from (0,0,5) is 4, box seen from (2,0,0)... use two spheres
    t = scenes._Tab()
    a = t.sphere((0, 0, 0), 1.0)       # at (0,0,5): 4
    b = t.sphere((0, 0, 3), 1.0)       # at (0,0,5): 1
    cc, w = oracle.serialize(t.nodes, t.op(scenes.UNION, a, b))
    assert oracle.map_scene(cc, w, [0, 0, 5]) == 1.0
    # subtract(a=-1, b=-0.5) = max(-1, 0.5) = 0.5: at the origin, box(1,1,1) = -1, sphere r=.5 = -.5
    t = scenes._Tab()
    a = t.box((0, 0, 0), (1, 1, 1))
    b = t.sphere((0, 0, 0), 0.5)
    cc, w = oracle.serialize(t.nodes, t.op(scenes.SUBTRACTION, a, b))
    assert oracle.map_scene(cc, w, [0, 0, 0]) == 0.5


def test_unknown_opcode_pushes_zero(oracle):
    # wgsl:223-225 + :199: default -> 0.0 is pushed and consumes no parameters
    assert oracle.map_scene(1, [7], [1, 2, 3]) == 0.0
    rc, _ = oracle.validate(1, [7], strict=True)
    assert rc == -6
    # union(sphere=4, unknown=0) = 0
    cc, w = oracle.serialize(*scenes.g1())
    w2 = list(w) + [55, 100]
    assert oracle.map_scene(3, w2, [0, 0, 5]) == 0.0


def test_validation_errors(oracle):
    assert oracle.validate(1, [0, 0, 0])[0] == -2            # truncated sphere
    assert oracle.validate(1, [100])[0] == -3                # operator on empty stack
    cc, w = oracle.serialize(*scenes.right_deep(33))
    assert oracle.validate(cc, w)[0] == -4                   # 33 deep > 32 (wgsl:173)
    cc, w = oracle.serialize(*scenes.right_deep(32))
    assert oracle.validate(cc, w) == (0, 32)


def test_march_known_ray(oracle):
    cc, w = oracle.serialize(*scenes.g1())
    c = oracle.ray_march(cc, w, [0, 0, 5], [0, 0, -1])
    # step0 s=4 -> dist 4; step1 pos=(0,0,1) s=0 < 0.01 hit; n=(0,0,1); L=(-2,5,-2)/sqrt(33); dot<0 -> 0.02
    assert np.array_equal(c, np.array([F(0.4) * F(0.02), F(0.7) * F(0.02), F(0.1) * F(0.02)], dtype=F))
    g = np.sqrt(c)
    assert np.allclose(g, [0.08944, 0.11832, 0.04472], atol=1e-5)


def test_empty_scene(oracle):
    # cmd_count = 0: map_scene == max_dist (wgsl:189-191), runs max_iter steps, then floor / black
    assert oracle.map_scene(0, [], [1, 2, 3], limits=(0.01, 100.0, 7)) == 100.0
    up = oracle.ray_march(0, [], [0, 0, 5], [0, 0.6, -0.8])
    assert np.array_equal(up, np.zeros(3, dtype=F))
    level = oracle.ray_march(0, [], [0, 0, 5], [0, 0.0, -1.0])   # t = -1.5/0 = -inf -> not > 0
    assert np.array_equal(level, np.zeros(3, dtype=F))
    down = oracle.ray_march(0, [], [0, 0, 5], [0, -0.6, -0.8])
    # t = -1.5/-0.6 = 2.5 ; p = (0, ., 5-2) = (0,3) ; ipos = round(.5)=0 (ties-to-even), round(3.5)=4 -> (0^4)&1 = 0
    t = F(-1.5) / F(-0.6)
    pz = F(5) + F(-0.8) * t
    ix, iz = int(np.rint(F(0) + F(0.5))), int(np.rint(pz + F(0.5)))
    col = F((ix ^ iz) & 1)
    exp = np.array([F(0.1) + F(0.2) * col, F(0.1) + F(0.2) * col, F(0.2) + F(0.2) * col], dtype=F)
    assert np.array_equal(down, exp)


def test_round_half_even_checker(oracle):
    # a ray landing at x = 1.0 exactly: round(1.5) = 2 under ties-to-even (roundf would also give 2),
    # x = 2.0: round(2.5) = 2 (roundf would give 3): this distinguishes rintf from roundf.
    o = [2.0, 0.0, 0.0]
    d = [0.0, -1.0, 0.0]           # straight down: t = 1.5, p.xz = (2, 0)
    c = oracle.ray_march(0, [], o, d)
    ix, iz = 2, 0                  # rint(2.5) = 2, rint(0.5) = 0
    col = F((ix ^ iz) & 1)
    assert c[0] == F(0.1) + F(0.2) * col and col == 0.0


def test_camera_default(oracle):
    u, pos, q, orb = oracle.orbit_uniforms((256.0, 256.0))
    assert list(pos) == [0.0, 0.0, 5.0]
    iv = np.array(list(u.inv_view), dtype=F).reshape(4, 4).T    # column-major -> [r][c]
    assert np.array_equal(iv, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 5], [0, 0, 0, 1]], dtype=F))


def test_camera_orbit_pitch_clamp_dolly(oracle):
    u, pos, q, orb = oracle.orbit_uniforms((256.0, 256.0), events=[(1, 100.0, 0.0)])
    assert orb.yaw == F(1.0)
    # R = Ry(-yaw): position = (-5 sin 1, 0, 5 cos 1)
    assert np.allclose(pos, [-5 * math.sin(1.0), 0.0, 5 * math.cos(1.0)], atol=1e-5)
    _, _, _, orb = oracle.orbit_uniforms((256.0, 256.0), events=[(1, 0.0, 1000.0)])
    assert orb.pitch == F(1.5)
    _, _, _, orb = oracle.orbit_uniforms((256.0, 256.0), events=[(1, 0.0, -1000.0)])
    assert orb.pitch == F(-1.5)
    _, _, _, orb = oracle.orbit_uniforms((256.0, 256.0), events=[(2, 10.0, 0.0)])
    assert orb.radius == F(5.0) + F(10.0) * F(0.01) * F(5.0)
    _, _, _, orb = oracle.orbit_uniforms((256.0, 256.0), events=[(2, -1000.0, 0.0)])
    assert orb.radius == F(0.1)


def test_inv_view_is_rotation_and_position(oracle):
    u, pos, q, orb = oracle.orbit_uniforms((1920.0, 1080.0), events=scenes.STILL_CAMERA_EVENTS)
    iv = np.array(list(u.inv_view), dtype=np.float64).reshape(4, 4).T
    assert np.allclose(iv[:3, 3], pos, atol=1e-5)
    assert np.allclose(iv[:3, :3] @ iv[:3, :3].T, np.eye(3), atol=1e-5)
    assert np.allclose(iv[3], [0, 0, 0, 1], atol=1e-6)


def test_projection_inverse(oracle):
    # aspect 1 -> pt_view = (0.41421357 x, 0.41421357 y, -1, w=1 exactly): no perspective divide needed
    out = np.zeros(16, dtype=F)
    import ctypes as C
    oracle.lib().rmo_perspective_inverse(1.0, math.pi / 4, 1.0, 10000.0, out.ctypes.data_as(C.POINTER(C.c_float)))
    m = out.reshape(4, 4).T
    assert abs(m[0, 0] - 0.41421357) < 1e-7 and m[0, 0] == m[1, 1]
    assert m[2, 3] == -1.0 and m[2, 2] == 0.0
    v = np.array([0.3, -0.7, -1.0, 1.0], dtype=F)
    w = ((m[3, 0] * v[0] + m[3, 1] * v[1]) + m[3, 2] * v[2]) + m[3, 3] * v[3]
    assert w == F(1.0)


def test_aa_offsets():
    offs = [(F(i) + F(0.5)) / F(4) - F(0.5) for i in range(4)]
    assert offs == [-0.375, -0.125, 0.125, 0.375]     # wgsl:50


def test_render_analytic_sphere_mask(oracle):
    """Hit mask of the single-sphere scene vs closed-form ray/sphere intersection."""
    W = H = 96
    u, pos, q, _ = oracle.orbit_uniforms((float(W), float(H)))
    cc, w = oracle.serialize(*scenes.g1())
    img = oracle.render(u, (0.01, 100.0, 64), cc, w, W, H)
    # hit pixels are green-dominant (0.4,0.7,0.1)*k ; floor/black are not
    hit = img[..., 1] > img[..., 2] * 1.2
    tanh = math.tan(math.pi / 8)
    ys, xs = np.mgrid[0:H, 0:W]
    sx = -1 + 2 * (xs + 0.5) / W
    sy = 1 - 2 * (ys + 0.5) / H
    d = np.stack([sx * tanh, sy * tanh, -np.ones_like(sx)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.array([0, 0, 5.0])
    b = (d * o).sum(-1)
    disc = b * b - (o @ o - 1.0)
    analytic = disc > 0
    # disagreement only within ~1 px of the silhouette (AA + min_dist band)
    diff = hit != analytic
    assert diff.mean() < 0.03
    inner = disc > 0.08
    assert hit[inner].all()
    outer = disc < -0.08
    assert (~hit[outer]).all()
    assert np.array_equal(img[..., 3], np.ones((H, W), dtype=F))


def test_unorm8_quantisation_known_answers(oracle):
    """Output stage (extension): clamp, * 255, round to nearest EVEN; NaN -> 0; alpha 1.0 -> 255; channel order."""
    import numpy as np
    px = np.array([[0.0, 1.0, 0.5, 1.0],                      # 0.5 * 255 = 127.5 -> 128 (even)
                   [2.5 / 255.0, 3.5 / 255.0, -0.25, 1.0],     # ties: 2.5 -> 2, 3.5 -> 4; negative -> 0
                   [np.nan, np.inf, 1.5, 1.0],                 # NaN -> 0, inf and 1.5 -> 255
                   [0.08944272, 0.11832160, 0.04472136, 1.0]], dtype=np.float32)   # the hand-derived hit colour of 8(c)
    q = oracle.quantize_unorm8(px)
    assert q.tolist() == [[0, 255, 128, 255], [2, 4, 0, 255], [0, 255, 255, 255], [23, 30, 11, 255]]
    assert oracle.quantize_unorm8(px, bgra=True)[:, [2, 1, 0, 3]].tolist() == q.tolist()
