"""ctypes bindings of the two native libraries.  There is NO fallback: if librm_hip.so is
missing or no GPU is visible, the render path raises (the product never runs on the CPU)."""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
HIP_SO = os.environ.get("RM_HIP_SO") or os.path.join(PKG, "librm_hip.so")   # RM_HIP_SO: A/B builds
HOST_SO = os.path.join(PKG, "librm_host.so")


class RmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("rm status %d: %s" % (status, message))
        self.status = status


class Uniforms(C.Structure):
    """rm_uniforms, 144 bytes (include/rm_abi.h)."""
    _fields_ = [("viewport_extent", C.c_float * 2), ("_pad", C.c_float * 2),
                ("inv_proj", C.c_float * 16), ("inv_view", C.c_float * 16)]


class Limits(C.Structure):
    """rm_limits, 12 bytes."""
    _fields_ = [("min_dist", C.c_float), ("max_dist", C.c_float), ("max_iter", C.c_uint32)]


# status codes (include/rm_abi.h enum rm_status)
RM_OK, RM_ERR_NULL, RM_ERR_TRUNCATED, RM_ERR_STACK_UNDERFLOW, RM_ERR_STACK_OVERFLOW = 0, -1, -2, -3, -4
RM_ERR_EMPTY_RESULT, RM_ERR_OPCODE, RM_ERR_TOO_LARGE, RM_ERR_RANGE, RM_ERR_DEVICE = -5, -6, -7, -8, -9
RM_ERR_NO_DEVICE, RM_ERR_ARG, RM_ERR_TRANSFORM, RM_ERR_MATERIAL = -10, -11, -12, -13
RM_BUF_LIMITS, RM_BUF_COMMANDS, RM_BUF_UNIFORMS = 0, 1, 2
RM_OPT_KERNEL, RM_OPT_TIMING, RM_OPT_STRICT_CAP, RM_OPT_REFILL_MIN, RM_OPT_CULL = 0, 1, 2, 3, 4
RM_OPT_BALANCE, RM_OPT_WAVE_STATS, RM_OPT_WAVES_PER_TILE, RM_OPT_SPECIALIZE, RM_OPT_PRUNE = 5, 6, 7, 8, 9
RM_KERNEL_DEFAULT, RM_KERNEL_PIXEL = 0, 1      # 2..11: the retired v2-v4 variants of ABI version 1
RM_KERNEL_V5, RM_KERNEL_V5_LDS = 12, 13
RM_JIT_PRUNE = 0x100
RM_STREAM_OWN = (1 << 64) - 1   # (void*)-1: the context's own stream
RM_OPT_OUTPUT_FORMAT = 10
RM_FORMAT_RGBA32F, RM_FORMAT_RGBA8_UNORM, RM_FORMAT_BGRA8_UNORM = 0, 1, 2
RM_INFO_KERNEL_MS, RM_INFO_PROGRAM_COMMANDS, RM_INFO_PROGRAM_WORDS, RM_INFO_PROGRAM_DEPTH = 0, 1, 2, 3
RM_INFO_DEVICE, RM_INFO_CU_COUNT, RM_INFO_SPECIALIZED, RM_INFO_JIT_STATE, RM_INFO_JIT_COMPILE_MS = 4, 5, 6, 7, 8
RM_INFO_PRUNED, RM_INFO_INTERPRETER_LOOP, RM_INFO_JIT_FROM_CACHE = 9, 10, 11

_hip = None
_host = None


def _preload_hip_runtime():
    """One HIP runtime per process.  The torch wheel bundles its own libamdhip64.so (SONAME
    libamdhip64.so.7, requested by libtorch_hip.so as plain "libamdhip64.so"); if librm_hip.so
    pulled in /opt/rocm's copy first, a later `import torch` would load a SECOND runtime, which
    then sees no GPU, and torch stream handles would be meaningless to us.  So when torch is
    installed its copy is loaded first and librm_hip.so binds to it by SONAME.  Hosts without
    torch (C++, Rust) simply use the system ROCm runtime.  RM_HIP_RUNTIME=system skips this."""
    if os.environ.get("RM_HIP_RUNTIME", "") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
        cand = os.path.join(libdir, "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            # ... and the run-time compiler of the same ROCm release (csrc/rm_jit.h asks for it by SONAME)
            rtc = os.path.join(libdir, "libhiprtc.so")
            if os.path.exists(rtc) and not os.environ.get("RM_HIPRTC_SO"):
                C.CDLL(rtc, mode=C.RTLD_GLOBAL)


def hip_lib():
    """librm_hip.so, loaded once.  Raises if it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_SO):
            raise ImportError("%s is missing: run `python -m ray_marching_amd.build` (there is no CPU fallback)"
                              % HIP_SO)
        _preload_hip_runtime()
        L = C.CDLL(HIP_SO)
        vp, u32, u64, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64
        L.rm_abi_version.restype = C.c_int
        L.rm_device_count.restype = C.c_int
        L.rm_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.rm_destroy.argtypes = [vp]
        L.rm_destroy.restype = None
        L.rm_write_buffer.argtypes = [vp, C.c_int, u64, vp, u64]
        L.rm_set_uniforms.argtypes = [vp, C.POINTER(Uniforms)]
        L.rm_set_limits.argtypes = [vp, C.POINTER(Limits)]
        L.rm_set_program.argtypes = [vp, u32, C.POINTER(u32), u32]
        L.rm_set_materials.argtypes = [vp, u32, C.POINTER(C.c_float)]
        L.rm_resize_command_buffer.argtypes = [vp, u64]
        L.rm_validate.argtypes = [vp]
        L.rm_validate_program.argtypes = [u32, C.POINTER(u32), u32, C.POINTER(u32)]
        L.rm_validate_program.restype = C.c_int
        L.rm_program_info.argtypes = [u32, C.POINTER(u32), u32, C.POINTER(u32), u32]
        L.rm_program_info.restype = C.c_int
        L.rm_draw.argtypes = [vp, u32, u32, u32, u32, vp, C.c_int, vp]
        L.rm_draw_strips.argtypes = [vp, u32, u32, u32, u32, u32, vp, C.c_int, vp, C.POINTER(u32)]
        L.rm_draw_strips.restype = C.c_int
        L.rm_draw_batch.argtypes = [vp, C.POINTER(Uniforms), u32, u32, u32, vp, C.c_int, vp]
        L.rm_gather_strips.argtypes = [vp, u32, u32, u32, u32, u32, vp, vp, vp]
        L.rm_gather_strips.restype = C.c_int
        L.rm_host_register.argtypes = [vp, u64]
        L.rm_host_register.restype = C.c_int
        L.rm_host_unregister.argtypes = [vp]
        L.rm_host_unregister.restype = C.c_int
        L.rm_sync.argtypes = [vp]
        L.rm_sync_context.argtypes = [vp]
        L.rm_sync_context.restype = C.c_int
        L.rm_set_option.argtypes = [vp, C.c_int, i64]
        L.rm_get_info.argtypes = [vp, C.c_int, C.POINTER(C.c_double)]
        L.rm_measure_write_bandwidth.argtypes = [vp, u64, C.c_int, C.POINTER(C.c_double)]
        L.rm_selftest_sqrt.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
        L.rm_selftest_sqrt.restype = C.c_int
        L.rm_selftest_ops.argtypes = [vp, vp, vp, vp, u32]
        L.rm_selftest_ops.restype = C.c_int
        L.rm_selftest_wave.argtypes = [vp, vp, u32, vp]
        L.rm_selftest_wave.restype = C.c_int
        L.rm_read_wave_stats.argtypes = [vp, vp, u64, C.POINTER(u64)]
        L.rm_read_wave_stats.restype = C.c_int
        sz = C.c_size_t
        L.rm_jit_source.argtypes = [u32, C.POINTER(u32), u32, C.c_int, C.c_char_p, sz, C.POINTER(sz)]
        L.rm_jit_source.restype = C.c_int
        L.rm_jit_compile.argtypes = [u32, C.POINTER(u32), u32, C.c_int, C.POINTER(C.c_double), C.POINTER(sz),
                                     C.c_char_p, sz]
        L.rm_jit_compile.restype = C.c_int
        L.rm_jit_log.argtypes = [vp, C.c_char_p, sz]
        L.rm_jit_log.restype = C.c_int
        L.rm_last_error.argtypes = [vp]
        L.rm_last_error.restype = C.c_char_p
        L.rm_status_string.argtypes = [C.c_int]
        L.rm_status_string.restype = C.c_char_p
        for name in ("rm_create", "rm_write_buffer", "rm_set_uniforms", "rm_set_limits", "rm_set_program",
                     "rm_resize_command_buffer", "rm_validate", "rm_draw", "rm_draw_batch", "rm_sync",
                     "rm_set_option", "rm_get_info", "rm_measure_write_bandwidth", "rm_set_materials"):
            getattr(L, name).restype = C.c_int
        _hip = L
    return _hip


def check(ctx, status):
    if status != RM_OK:
        L = hip_lib()
        msg = L.rm_last_error(ctx) or b""
        raise RmError(status, msg.decode() or L.rm_status_string(status).decode())
    return status


class OrbitState(C.Structure):
    """rmh_orbit (include/rm_host.h)."""
    _fields_ = [("target", C.c_float * 3), ("pitch", C.c_float), ("yaw", C.c_float), ("radius", C.c_float),
                ("pan_speed", C.c_float), ("yaw_speed", C.c_float), ("pitch_speed", C.c_float),
                ("dolly_speed", C.c_float)]


class CameraState(C.Structure):
    """rmh_camera: position + unit quaternion (w,i,j,k)."""
    _fields_ = [("position", C.c_float * 3), ("rotation", C.c_float * 4)]


def host_lib():
    """librm_host.so (pure host code: scene model, serializer, camera)."""
    global _host
    if _host is None:
        if not os.path.exists(HOST_SO):
            raise ImportError("%s is missing: run `python -m ray_marching_amd.build`" % HOST_SO)
        L = C.CDLL(HOST_SO)
        vp, u32, f32p = C.c_void_p, C.c_uint32, C.POINTER(C.c_float)
        L.rmh_sphere.argtypes = [f32p, C.c_float]
        L.rmh_box.argtypes = [f32p, f32p]
        L.rmh_union.argtypes = [vp, vp]
        L.rmh_subtraction.argtypes = [vp, vp]
        L.rmh_plane.argtypes = [f32p, C.c_float]
        L.rmh_cylinder.argtypes = [f32p, C.c_float, C.c_float]
        L.rmh_intersection.argtypes = [vp, vp]
        L.rmh_smooth_union.argtypes = [vp, vp, C.c_float]
        L.rmh_translation.argtypes = [vp, f32p]
        L.rmh_rotation.argtypes = [vp, f32p]
        L.rmh_scale.argtypes = [vp, C.c_float]
        L.rmh_material.argtypes = [vp, u32]
        L.rmh_node_clone.argtypes = [vp]
        L.rmh_scene.argtypes = [C.c_char_p]
        for n in ("rmh_sphere", "rmh_box", "rmh_union", "rmh_subtraction", "rmh_node_clone", "rmh_scene",
                  "rmh_builder_new", "rmh_plane", "rmh_cylinder", "rmh_intersection", "rmh_smooth_union",
                  "rmh_translation", "rmh_rotation", "rmh_scale", "rmh_material"):
            getattr(L, n).restype = vp
        L.rmh_node_free.argtypes = [vp]
        L.rmh_node_free.restype = None
        L.rmh_builder_free.argtypes = [vp]
        L.rmh_builder_free.restype = None
        L.rmh_builder_push_command.argtypes = [vp, u32]
        L.rmh_builder_push_command.restype = None
        L.rmh_builder_push_param_vec3.argtypes = [vp, f32p]
        L.rmh_builder_push_param_vec3.restype = None
        L.rmh_builder_push_param_float.argtypes = [vp, C.c_float]
        L.rmh_builder_push_param_float.restype = None
        L.rmh_builder_cmd_count.argtypes = [vp]
        L.rmh_builder_cmd_count.restype = u32
        L.rmh_builder_len.argtypes = [vp]
        L.rmh_builder_len.restype = u32
        L.rmh_builder_buffer.argtypes = [vp]
        L.rmh_builder_buffer.restype = C.POINTER(u32)
        L.rmh_build_commands.argtypes = [vp, vp]
        L.rmh_build_commands.restype = None
        L.rmh_orbit_new.argtypes = [C.POINTER(OrbitState), f32p, C.c_float]
        L.rmh_orbit_new.restype = None
        L.rmh_orbit_update.argtypes = [C.POINTER(OrbitState), C.c_int, C.c_float, C.c_float]
        L.rmh_orbit_update.restype = None
        L.rmh_orbit_camera.argtypes = [C.POINTER(OrbitState), C.POINTER(CameraState)]
        L.rmh_orbit_camera.restype = None
        L.rmh_camera_view.argtypes = [C.POINTER(CameraState), f32p]
        L.rmh_camera_view.restype = None
        L.rmh_prepare_uniforms.argtypes = [f32p, C.POINTER(CameraState), vp]
        L.rmh_prepare_uniforms.restype = None
        _host = L
    return _host
