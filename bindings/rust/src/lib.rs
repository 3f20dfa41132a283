//! rm_hip -- Rust binding of include/rm_abi.h (librm_hip.so, C ABI version 2).
//!
//! NOT COMPILED IN THE BUILD IMAGE: it has no rustc/cargo.  What is checked there instead
//! (tests/test_rust_binding.py): every function include/rm_abi.h declares appears in the `extern "C"`
//! block below with the same name, the same number of parameters and matching parameter / return
//! types, every constant has the header's value, and attributes sit on items that accept them; the
//! same test runs `cargo check --offline` when a toolchain is on PATH.  The verified consumers of
//! the ABI are the ctypes bindings (ray-marching_amd/_ffi.py) and the C++ mirror
//! (ray-marching_amd/csrc/host/renderer.hpp).
//!
//! The crate shows what a maintainer of Mesoptier/ray-marching would add to swap the wgpu objects
//! of `src/ray_marching/renderer.rs` for the HIP path while keeping `CSGNode`, `BuildCommands`,
//! `CSGCommandBufferBuilder`, `Camera` and `RayMarchingCallback::new(time, csg_node, viewport,
//! camera)` untouched.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};
use std::fmt;

/// Version of the C ABI these declarations were written against (`RM_ABI_VERSION`).
pub const RM_ABI_VERSION: c_int = 2;

#[repr(C)]
pub struct rm_ctx {
    _private: [u8; 0],
}

/// `Uniforms` (renderer.rs:29-34), 144 bytes: exactly `Uniforms::as_shader_bytes()`.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rm_uniforms {
    pub viewport_extent: [f32; 2],
    pub _pad: [f32; 2],
    pub inv_proj: [f32; 16],
    pub inv_view: [f32; 16],
}

/// `RayMarchLimits` (renderer.rs:36-41), 12 bytes.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct rm_limits {
    pub min_dist: f32,
    pub max_dist: f32,
    pub max_iter: u32,
}

// enum rm_buffer: binding numbers of the reference's bind group (renderer.rs:60-94)
pub const RM_BUF_LIMITS: c_int = 0;
pub const RM_BUF_COMMANDS: c_int = 1;
pub const RM_BUF_UNIFORMS: c_int = 2;

// enum rm_status
pub const RM_OK: c_int = 0;
pub const RM_ERR_NULL: c_int = -1;
pub const RM_ERR_TRUNCATED: c_int = -2;
pub const RM_ERR_STACK_UNDERFLOW: c_int = -3;
pub const RM_ERR_STACK_OVERFLOW: c_int = -4;
pub const RM_ERR_EMPTY_RESULT: c_int = -5;
pub const RM_ERR_OPCODE: c_int = -6;
pub const RM_ERR_TOO_LARGE: c_int = -7;
pub const RM_ERR_RANGE: c_int = -8;
pub const RM_ERR_DEVICE: c_int = -9;
pub const RM_ERR_NO_DEVICE: c_int = -10;
pub const RM_ERR_ARG: c_int = -11;
pub const RM_ERR_TRANSFORM: c_int = -12;
pub const RM_ERR_MATERIAL: c_int = -13;

// enum rm_option (rm_set_option keys)
pub const RM_OPT_KERNEL: c_int = 0;
pub const RM_OPT_TIMING: c_int = 1;
pub const RM_OPT_STRICT_CAP: c_int = 2;
pub const RM_OPT_REFILL_MIN: c_int = 3;
pub const RM_OPT_CULL: c_int = 4;
pub const RM_OPT_BALANCE: c_int = 5;
pub const RM_OPT_WAVE_STATS: c_int = 6;
pub const RM_OPT_WAVES_PER_TILE: c_int = 7;
pub const RM_OPT_SPECIALIZE: c_int = 8; // 0 interpreter kernel only, 1 (default) compile per scene structure in the background, 2 blocking
pub const RM_OPT_PRUNE: c_int = 9;
pub const RM_OPT_OUTPUT_FORMAT: c_int = 10; // RM_FORMAT_*

// enum rm_format
pub const RM_FORMAT_RGBA32F: c_int = 0;
pub const RM_FORMAT_RGBA8_UNORM: c_int = 1;
pub const RM_FORMAT_BGRA8_UNORM: c_int = 2; // what an egui/wgpu surface usually is (renderer.rs:113 `target_format`)

// enum rm_kernel
pub const RM_KERNEL_DEFAULT: c_int = 0;
pub const RM_KERNEL_PIXEL: c_int = 1;
pub const RM_KERNEL_V5: c_int = 12;
pub const RM_KERNEL_V5_LDS: c_int = 13;

// enum rm_info (rm_get_info keys)
pub const RM_INFO_KERNEL_MS: c_int = 0;
pub const RM_INFO_PROGRAM_COMMANDS: c_int = 1;
pub const RM_INFO_PROGRAM_WORDS: c_int = 2;
pub const RM_INFO_PROGRAM_DEPTH: c_int = 3;
pub const RM_INFO_DEVICE: c_int = 4;
pub const RM_INFO_CU_COUNT: c_int = 5;
pub const RM_INFO_SPECIALIZED: c_int = 6;
pub const RM_INFO_JIT_STATE: c_int = 7;
pub const RM_INFO_JIT_COMPILE_MS: c_int = 8;
pub const RM_INFO_PRUNED: c_int = 9;
pub const RM_INFO_INTERPRETER_LOOP: c_int = 10;
pub const RM_INFO_JIT_FROM_CACHE: c_int = 11;

// enum rm_program_fact: indices into the array rm_program_info fills
pub const RM_PROGRAM_RECORDS: usize = 0;
pub const RM_PROGRAM_CONES: usize = 1;
pub const RM_PROGRAM_SLABS: usize = 2;
pub const RM_PROGRAM_SUBTRACTED_LEAVES: usize = 3;
pub const RM_PROGRAM_GROUPS: usize = 4;
pub const RM_PROGRAM_SPILL_DEPTH: usize = 5;
pub const RM_PROGRAM_IS_CHAIN: usize = 6;
pub const RM_PROGRAM_PRUNABLE: usize = 7;
pub const RM_PROGRAM_BOUND_WALK: usize = 8;
pub const RM_PROGRAM_HAS_XFORMS: usize = 9;
pub const RM_PROGRAM_LEAVES: usize = 10;
pub const RM_PROGRAM_AUTO_PRUNED: usize = 11;
pub const RM_PROGRAM_FACTS: usize = 12;

/// `RM_JIT_PRUNE`: OR into `waves_per_tile` of rm_jit_source / rm_jit_compile.
pub const RM_JIT_PRUNE: c_int = 0x100;

/// `RM_STREAM_OWN`: pass as `stream` of a device-destination draw to use the context's own stream.
pub const RM_STREAM_OWN: *mut c_void = usize::MAX as *mut c_void;

// Opcodes of the node types the reference only names in comments (builder.rs:8,14,16-23) and that the device
// path implements as extensions: a CSGCommandType that gains these variants serialises them unchanged.
//   Plane = 2, Intersection = 102, TranslationPush = 200, TranslationPop, RotationPush, RotationPop, ScalePush, ScalePop
//   (Cylinder = 10, SmoothUnion = 110 and Material = 300 are this repo's own numbers)

#[link(name = "rm_hip")]
extern "C" {
    pub fn rm_abi_version() -> c_int;
    pub fn rm_device_count() -> c_int;
    pub fn rm_create(device: c_int, out: *mut *mut rm_ctx) -> c_int;
    pub fn rm_destroy(ctx: *mut rm_ctx);
    pub fn rm_write_buffer(ctx: *mut rm_ctx, buffer: c_int, offset: u64, data: *const c_void, size: u64) -> c_int;
    pub fn rm_set_uniforms(ctx: *mut rm_ctx, u: *const rm_uniforms) -> c_int;
    pub fn rm_set_limits(ctx: *mut rm_ctx, l: *const rm_limits) -> c_int;
    pub fn rm_set_program(ctx: *mut rm_ctx, cmd_count: u32, words: *const u32, n_words: u32) -> c_int;
    pub fn rm_set_materials(ctx: *mut rm_ctx, count: u32, rgb: *const f32) -> c_int;
    pub fn rm_resize_command_buffer(ctx: *mut rm_ctx, bytes: u64) -> c_int;
    pub fn rm_validate(ctx: *mut rm_ctx) -> c_int;
    pub fn rm_validate_program(cmd_count: u32, words: *const u32, n_words: u32, out_max_depth: *mut u32) -> c_int;
    pub fn rm_program_info(cmd_count: u32, words: *const u32, n_words: u32, out: *mut u32, n_out: u32) -> c_int;
    pub fn rm_draw(ctx: *mut rm_ctx, w: u32, h: u32, row0: u32, rows: u32, out_rgba: *mut f32, out_is_device: c_int,
                   stream: *mut c_void) -> c_int;
    pub fn rm_draw_strips(ctx: *mut rm_ctx, w: u32, h: u32, strip_rows: u32, first: u32, stride: u32, out_rgba: *mut f32,
                          out_is_device: c_int, stream: *mut c_void, out_rows: *mut u32) -> c_int;
    pub fn rm_gather_strips(ctx: *mut rm_ctx, w: u32, h: u32, strip_rows: u32, first: u32, stride: u32,
                            strips_device: *const c_void, host_image: *mut c_void, stream: *mut c_void) -> c_int;
    pub fn rm_host_register(ptr: *mut c_void, bytes: u64) -> c_int;
    pub fn rm_host_unregister(ptr: *mut c_void) -> c_int;
    pub fn rm_draw_batch(ctx: *mut rm_ctx, frames: *const rm_uniforms, n_frames: u32, w: u32, h: u32, out_rgba: *mut f32,
                         out_is_device: c_int, stream: *mut c_void) -> c_int;
    pub fn rm_sync(ctx: *mut rm_ctx) -> c_int;
    pub fn rm_sync_context(ctx: *mut rm_ctx) -> c_int;
    pub fn rm_set_option(ctx: *mut rm_ctx, key: c_int, value: i64) -> c_int;
    pub fn rm_get_info(ctx: *mut rm_ctx, key: c_int, out: *mut f64) -> c_int;
    pub fn rm_measure_write_bandwidth(ctx: *mut rm_ctx, bytes: u64, iters: c_int, out_gbps: *mut f64) -> c_int;
    pub fn rm_selftest_sqrt(ctx: *mut rm_ctx, out_mismatches: *mut u64, out_first_bad_bits: *mut u32) -> c_int;
    pub fn rm_selftest_ops(ctx: *mut rm_ctx, a: *const f32, b: *const f32, out: *mut f32, n: u32) -> c_int;
    pub fn rm_selftest_wave(ctx: *mut rm_ctx, input: *const f32, n_waves: u32, out: *mut f32) -> c_int;
    pub fn rm_read_wave_stats(ctx: *mut rm_ctx, dst: *mut c_void, cap_bytes: u64, out_bytes: *mut u64) -> c_int;
    pub fn rm_jit_source(cmd_count: u32, words: *const u32, n_words: u32, waves_per_tile: c_int, buf: *mut c_char,
                         cap: usize, needed: *mut usize) -> c_int;
    pub fn rm_jit_compile(cmd_count: u32, words: *const u32, n_words: u32, waves_per_tile: c_int, compile_ms: *mut f64,
                          code_bytes: *mut usize, log: *mut c_char, log_cap: usize) -> c_int;
    pub fn rm_jit_log(ctx: *mut rm_ctx, buf: *mut c_char, cap: usize) -> c_int;
    pub fn rm_last_error(ctx: *mut rm_ctx) -> *const c_char;
    pub fn rm_status_string(status: c_int) -> *const c_char;
}

/// What the reference `unwrap()`s away (renderer.rs:24, 203, 250): a status code of `enum rm_status`
/// plus the library's message for it.
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct RmError {
    pub status: i32,
    pub message: String,
}

impl fmt::Display for RmError {
    fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
        write!(f, "rm_hip error {}: {}", self.status, self.message)
    }
}

impl std::error::Error for RmError {}

/// Replaces `RayMarchingResources` (renderer.rs:43-49): owns the GPU state of one device.
pub struct RayMarchingResources {
    ctx: *mut rm_ctx,
}

unsafe impl Send for RayMarchingResources {} // a context may move between threads; it is not Sync

impl RayMarchingResources {
    /// `RayMarchingResources::new` (renderer.rs:51-175) without a wgpu `RenderState`.
    pub fn new(device: i32) -> Result<Self, RmError> {
        let mut ctx: *mut rm_ctx = std::ptr::null_mut();
        let rc = unsafe { rm_create(device, &mut ctx) };
        if rc != RM_OK {
            let msg = unsafe { CStr::from_ptr(rm_last_error(std::ptr::null_mut())) };
            return Err(RmError { status: rc, message: msg.to_string_lossy().into_owned() });
        }
        Ok(Self { ctx })
    }

    fn check(&self, rc: c_int) -> Result<(), RmError> {
        if rc == RM_OK {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(rm_last_error(self.ctx)) };
        Err(RmError { status: rc, message: msg.to_string_lossy().into_owned() })
    }

    /// `queue.write_buffer(buffer, offset, data)` (renderer.rs:213, 230, 235).
    pub fn write_buffer(&self, buffer: c_int, offset: u64, data: &[u8]) -> Result<(), RmError> {
        self.check(unsafe { rm_write_buffer(self.ctx, buffer, offset, data.as_ptr() as *const c_void, data.len() as u64) })
    }

    /// The reference's TODO (renderer.rs:229): a command buffer larger than 1024 bytes.
    pub fn resize_command_buffer(&self, bytes: u64) -> Result<(), RmError> {
        self.check(unsafe { rm_resize_command_buffer(self.ctx, bytes) })
    }

    /// `RayMarchLimits` (renderer.rs:130-140): the reference writes them once at start-up.
    pub fn set_limits(&self, limits: &rm_limits) -> Result<(), RmError> {
        self.check(unsafe { rm_set_limits(self.ctx, limits) })
    }

    pub fn set_option(&self, key: c_int, value: i64) -> Result<(), RmError> {
        self.check(unsafe { rm_set_option(self.ctx, key, value) })
    }

    /// `render_pass.draw(0..4, 0..2)` (renderer.rs:254) into a host RGBA32F image (top row first).
    pub fn draw(&self, width: u32, height: u32, out_rgba: &mut [f32]) -> Result<(), RmError> {
        assert!(out_rgba.len() >= (width as usize) * (height as usize) * 4);
        self.check(unsafe { rm_draw(self.ctx, width, height, 0, height, out_rgba.as_mut_ptr(), 0, std::ptr::null_mut()) })
    }

    /// This GPU's interleaved strips of a frame tiled over `stride` GPUs (north-star layout), host destination.
    /// Returns the number of rows written.
    pub fn draw_strips(&self, width: u32, height: u32, strip_rows: u32, first: u32, stride: u32,
                       out_rgba: &mut [f32]) -> Result<u32, RmError> {
        // The library writes every row of this GPU's strips: the slice must hold them, or a safe caller could make it
        // write past the end.  (Arguments the library rejects -- strip_rows 0, first >= stride -- write nothing.)
        let need = strip_row_count(height, strip_rows, first, stride) as usize * width as usize * 4;
        assert!(out_rgba.len() >= need, "draw_strips: out_rgba holds {} floats, this GPU's strips need {}", out_rgba.len(), need);
        let mut rows: u32 = 0;
        self.check(unsafe {
            rm_draw_strips(self.ctx, width, height, strip_rows, first, stride, out_rgba.as_mut_ptr(), 0, std::ptr::null_mut(),
                           &mut rows)
        })?;
        Ok(rows)
    }
}

/// Output rows of strips `first`, `first + stride`, ... (`strip_rows` rows each; the frame's last strip may be ragged) of an
/// image `height` rows high: what `rm_draw_strips` renders for one GPU.  0 for arguments the library rejects.
pub fn strip_row_count(height: u32, strip_rows: u32, first: u32, stride: u32) -> u32 {
    if strip_rows == 0 || stride == 0 || first >= stride {
        return 0;
    }
    let n_strips = (height + strip_rows - 1) / strip_rows;
    let mut rows = 0u32;
    let mut s = first;
    while s < n_strips {
        let r0 = s * strip_rows;
        rows += (height - r0).min(strip_rows);
        s += stride;
    }
    rows
}

impl Drop for RayMarchingResources {
    fn drop(&mut self) {
        unsafe { rm_destroy(self.ctx) }
    }
}

// ---------------------------------------------------------------------------------------------
// What `impl CallbackTrait for RayMarchingCallback` (renderer.rs:195-256) becomes.  `Uniforms`,
// `AsShaderBytes`, `CSGCommandBufferBuilder`, `BuildCommands`, `CSGNode` and `Camera` are the
// reference's own items, unchanged:
//
//     fn prepare(&self, resources: &RayMarchingResources) -> Result<(), RmError> {
//         let projection = Perspective3::new(self.viewport[0] / self.viewport[1], FRAC_PI_4, 1.0, 10000.0);
//         let uniforms = Uniforms {
//             viewport_extent: Vector2::new(self.viewport[0], self.viewport[1]),
//             inv_proj: projection.inverse(),
//             inv_view: self.camera.view().inverse().to_homogeneous(),
//         };
//         resources.write_buffer(RM_BUF_UNIFORMS, 0, &uniforms.as_shader_bytes())?;       // renderer.rs:213-222
//         let mut builder = CSGCommandBufferBuilder::new();
//         if let Some(csg_node) = &self.csg_node { csg_node.build_commands(&mut builder); }
//         resources.write_buffer(RM_BUF_COMMANDS, 0, bytemuck::cast_slice(&[builder.cmd_count]))?;  // :230-234
//         resources.write_buffer(RM_BUF_COMMANDS, 4, bytemuck::cast_slice(&builder.buffer))         // :235-239
//     }
//     fn paint(&self, resources: &RayMarchingResources, out: &mut [f32]) -> Result<(), RmError> {
//         resources.draw(self.viewport[0] as u32, self.viewport[1] as u32, out)           // renderer.rs:252-254
//     }
// ---------------------------------------------------------------------------------------------

#[cfg(test)]
mod tests {
    use super::*;

    #[test]
    fn blob_layouts_match_the_reference() {
        assert_eq!(std::mem::size_of::<rm_uniforms>(), 144); // renderer.rs:29-34 through encase
        assert_eq!(std::mem::size_of::<rm_limits>(), 12); // renderer.rs:36-41
    }

    #[test]
    fn error_is_a_std_error() {
        let e: Box<dyn std::error::Error> = Box::new(RmError { status: RM_ERR_ARG, message: "x".into() });
        assert!(e.to_string().contains("-11"));
    }
}
