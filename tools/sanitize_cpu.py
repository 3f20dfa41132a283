# Runs the host-mirror and oracle code paths under ASan/UBSan (CPU builds only).
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from oracle import cbind
cbind._SO = os.path.join(ROOT, 'oracle', 'librm_oracle_asan.so')
from ray_marching_amd import _ffi
_ffi.HOST_SO = '/tmp/librm_host_asan.so'
from ray_marching_amd import csg, camera, renderer
import scenes
for name in list(scenes.SCENES) + list(scenes.EXT_SCENES):
    allsc = {**scenes.SCENES, **scenes.EXT_SCENES}
    cc, w = cbind.serialize(*allsc[name]())
    hcc, hw = csg.serialize(csg.scene(name))
    assert hcc == cc and hw.tobytes() == w.tobytes()
    u, *_ = cbind.orbit_uniforms((24.0, 16.0), events=scenes.STILL_CAMERA_EVENTS)
    cbind.render(u, (0.01, 100.0, 48), cc, w, 24, 16, threads=3)
c = camera.OrbitCameraController.new([0, 0, 0], 5.0)
for ev in (camera.Pan([3, 4]), camera.Orbit([50, -20]), camera.Dolly(-3.0)):
    c.update(ev)
renderer.prepare_uniforms((640, 480), c.camera())
rng = np.random.default_rng(3)
for _ in range(300):   # malformed programs through the oracle validator
    n = int(rng.integers(0, 10)); words = [int(x) for x in rng.choice([0, 1, 2, 10, 100, 101, 102, 110, 7, 0x3F800000], size=int(rng.integers(0, 30)))]
    cbind.validate(n, words, strict=bool(rng.integers(0, 2)))
print("sanitizer run ok")
