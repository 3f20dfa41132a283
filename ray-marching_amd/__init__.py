"""ray-marching_amd: MI355X-native drop-in for the `src/ray_marching` wgpu pipeline of
Mesoptier/ray-marching.  The product is csrc/ (hand-written HIP kernels for gfx950 behind the
C ABI of include/rm_abi.h, plus the C++ host mirror of the reference's CSGNode / camera API);
the Python modules are thin ctypes bindings used by tests and bench.py."""
__version__ = "0.1.0"
