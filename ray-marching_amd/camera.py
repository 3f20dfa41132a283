"""Python face of csrc/host/camera.hpp: Camera and OrbitCameraController of src/camera.rs."""
import ctypes as C

import numpy as np

from . import _ffi


class Camera:  # camera.rs:3-13
    def __init__(self, state):
        self._s = state

    @property
    def position(self):
        return np.array(list(self._s.position), dtype=np.float32)

    @property
    def rotation(self):
        """unit quaternion (w, i, j, k)"""
        return np.array(list(self._s.rotation), dtype=np.float32)

    def view(self):
        """World-to-view matrix, 4x4 numpy array indexed [row][col] (camera.rs:10-12)."""
        out = (C.c_float * 16)()
        _ffi.host_lib().rmh_camera_view(C.byref(self._s), out)
        return np.array(list(out), dtype=np.float32).reshape(4, 4).T


class Pan:  # OrbitCameraControllerEvent::Pan([dx, dy]) camera.rs:16
    def __init__(self, delta):
        self.event, self.dx, self.dy = 0, float(delta[0]), float(delta[1])


class Orbit:  # camera.rs:17
    def __init__(self, delta):
        self.event, self.dx, self.dy = 1, float(delta[0]), float(delta[1])


class Dolly:  # camera.rs:18
    def __init__(self, delta):
        self.event, self.dx, self.dy = 2, float(delta), 0.0


class OrbitCameraController:  # camera.rs:21-85
    def __init__(self, target=(0.0, 0.0, 0.0), radius=5.0):
        self._L = _ffi.host_lib()
        self._s = _ffi.OrbitState()
        self._L.rmh_orbit_new(C.byref(self._s), (C.c_float * 3)(*[float(t) for t in target]), float(radius))

    @classmethod
    def new(cls, target, radius):
        return cls(target, radius)

    def update(self, event):
        self._L.rmh_orbit_update(C.byref(self._s), event.event, event.dx, event.dy)

    def camera(self):
        cam = _ffi.CameraState()
        self._L.rmh_orbit_camera(C.byref(self._s), C.byref(cam))
        return Camera(cam)

    pitch = property(lambda self: float(self._s.pitch))
    yaw = property(lambda self: float(self._s.yaw))
    radius = property(lambda self: float(self._s.radius))
    target = property(lambda self: np.array(list(self._s.target), dtype=np.float32))

    def set_angles(self, yaw, pitch, radius=None):
        """Direct state write used by the orbit-batch generator (frame f: yaw = 2*pi*f/N)."""
        self._s.yaw, self._s.pitch = float(yaw), float(pitch)
        if radius is not None:
            self._s.radius = float(radius)
