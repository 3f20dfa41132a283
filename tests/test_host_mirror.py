"""Host-side mirror (librm_host.so: C++ CSGNode / builder / camera) against the oracle's
restatement of the same reference code, bit for bit.  CPU only."""
import numpy as np
import pytest

import scenes
from ray_marching_amd import camera, csg, renderer


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_named_scenes_serialize_like_oracle(oracle, name):
    cc, words = csg.serialize(csg.scene(name))
    occ, owords = oracle.serialize(*scenes.SCENES[name]())
    assert cc == occ
    assert words.tobytes() == owords.tobytes()


def test_builder_api_matches_reference_layout():
    b = csg.CSGCommandBufferBuilder()
    b.push_command(csg.CSGCommandType.Sphere).push_param_vec3([0, 0, 0]).push_param_float(1.0)
    assert b.cmd_count == 1
    assert list(b.buffer) == [0, 0, 0, 0, 0x3F800000]            # SURVEY 8(c) serializer pin
    b.push_command(csg.CSGCommandType.Box).push_param_vec3([1, 2, 3]).push_param_vec3([4, 5, 6])
    b.push_command(csg.CSGCommandType.Union)
    assert b.cmd_count == 3 and len(b.buffer) == 13 and b.buffer[-1] == 100 and b.buffer[5] == 1


def test_tree_postorder_and_deep_copy():
    s, bx, s2 = csg.Sphere((0, 0, 0), 1.0), csg.Box((0, 0, 0), (1, 1, 1)), csg.Sphere((1, 0, 0), 0.5)
    u = csg.Union(s, bx)
    root = csg.Subtraction(u, s2)
    del s, bx, u                                    # children were deep-copied (Box::new(x.clone()))
    cc, w = csg.serialize(root)
    assert cc == 5 and len(w) == 19
    assert [int(w[0]), int(w[5]), int(w[12]), int(w[13]), int(w[18])] == [0, 1, 100, 0, 101]
    cc2, w2 = csg.serialize(root.clone())
    assert cc2 == cc and w2.tobytes() == w.tobytes()


def test_none_scene_is_empty_program():
    assert csg.serialize(None)[0] == 0 and len(csg.serialize(None)[1]) == 0   # renderer.rs:224-227


EVENT_SEQS = [
    [],
    [(1, 35.0, -25.0)],
    [(1, 100.0, 0.0)],
    [(1, 0.0, 1000.0)],
    [(1, 0.0, -1000.0)],
    [(2, 10.0, 0.0)],
    [(2, -1000.0, 0.0)],
    [(0, 12.0, -7.0), (1, 20.0, 10.0), (0, -3.0, 4.0), (2, 2.5, 0.0), (1, -60.0, 33.0)],
]


@pytest.mark.parametrize("events", EVENT_SEQS)
def test_orbit_controller_matches_oracle(oracle, events):
    c = camera.OrbitCameraController.new([0.0, 0.0, 0.0], 5.0)
    for ev, dx, dy in events:
        c.update({0: camera.Pan([dx, dy]), 1: camera.Orbit([dx, dy]), 2: camera.Dolly(dx)}[ev])
    cam = c.camera()
    u = renderer.prepare_uniforms((1920.0, 1080.0), cam)
    ou, opos, oq, oorb = oracle.orbit_uniforms((1920.0, 1080.0), events=events)
    assert (c.pitch, c.yaw, c.radius) == (oorb.pitch, oorb.yaw, oorb.radius)
    assert cam.position.tobytes() == opos.tobytes()
    assert cam.rotation.tobytes() == oq.tobytes()
    assert bytes(u) == bytes(ou)                    # the 144-byte blob, bit for bit


def test_camera_known_answers():
    c = camera.OrbitCameraController.new([0, 0, 0], 5.0)        # main.rs:38
    assert list(c.camera().position) == [0.0, 0.0, 5.0]
    v = c.camera().view()
    assert np.array_equal(v, np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -5], [0, 0, 0, 1]], dtype=np.float32))
    c.update(camera.Orbit([100.0, 0.0]))
    assert c.yaw == np.float32(1.0)
    assert np.allclose(c.camera().position, [-5 * np.sin(1.0), 0, 5 * np.cos(1.0)], atol=1e-5)
    c.update(camera.Orbit([0.0, 1e6]))
    assert c.pitch == np.float32(1.5)
    c.update(camera.Dolly(-1e6))
    assert c.radius == np.float32(0.1)


def test_uniform_blob_layout():
    c = camera.OrbitCameraController.new([0, 0, 0], 5.0)
    u = renderer.prepare_uniforms((640.0, 480.0), c.camera())
    raw = np.frombuffer(bytes(u), dtype=np.float32)
    assert raw.size == 36
    assert raw[0] == 640.0 and raw[1] == 480.0 and raw[2] == 0 and raw[3] == 0   # vec2 + pad to 16
    inv_proj = raw[4:20].reshape(4, 4).T
    inv_view = raw[20:36].reshape(4, 4).T
    assert inv_proj[2, 3] == -1.0 and inv_proj[2, 2] == 0.0
    assert np.array_equal(inv_view[:, 3], np.array([0, 0, 5, 1], dtype=np.float32))
