"""The C++ RayMarchingCallback mirror (csrc/host/renderer.hpp) driven from a C++ program linked
against librm_hip.so, compared with the oracle.  One short child process."""
import os
import subprocess

import numpy as np
import pytest

import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_dropin_program(oracle, tmp_path):
    from ray_marching_amd import build
    exe = build.build_cpp_dropin()
    out = tmp_path / "img.bin"
    out_tagged = tmp_path / "tagged.bin"
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "ray-marching_amd") + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(out), str(out_tagged)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "dropin ok" in r.stdout
    assert "specialised 1" in r.stdout, r.stdout      # hipRTC of the system ROCm compiled the scene's kernel in a pure C++ process
    W, H = 96, 64
    img = np.fromfile(out, dtype=np.float32).reshape(H, W, 4)
    cc, w = oracle.serialize(*scenes.g8())
    u, *_ = oracle.orbit_uniforms((float(W), float(H)), events=scenes.STILL_CAMERA_EVENTS)
    ref = oracle.render(u, (0.01, 100.0, 100), cc, w, W, H, threads=4)     # reference default limits
    assert img.tobytes() == ref.tobytes()
    # extension: the tagged scene of the C++ mirror with a material table
    cc, w = oracle.serialize(*scenes.mat_mix())
    ref = oracle.render(u, (0.01, 100.0, 100), cc, w, W, H, threads=4, materials=scenes.MATERIAL_TABLE)
    assert np.fromfile(out_tagged, dtype=np.float32).reshape(H, W, 4).tobytes() == ref.tobytes()
