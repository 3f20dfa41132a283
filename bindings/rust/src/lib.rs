//! rm_hip -- Rust binding of include/rm_abi.h (librm_hip.so).
//!
//! UNVERIFIED SOURCE: the build image has no rustc/cargo, so this file has never been compiled.
//! The verified consumers of the ABI are the ctypes bindings (ray-marching_amd/_ffi.py) and the
//! C++ mirror (ray-marching_amd/csrc/host/renderer.hpp).  It shows what a maintainer of
//! Mesoptier/ray-marching would add to swap `src/ray_marching/renderer.rs`'s wgpu objects for the
//! HIP path while keeping `CSGNode`, `BuildCommands`, `CSGCommandBufferBuilder`, `Camera` and
//! `RayMarchingCallback::new(time, csg_node, viewport, camera)` untouched.
#![allow(non_camel_case_types)]
use std::ffi::{c_char, c_int, c_void, CStr};

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

pub const RM_BUF_LIMITS: c_int = 0; // binding 0
pub const RM_BUF_COMMANDS: c_int = 1; // binding 1
pub const RM_BUF_UNIFORMS: c_int = 2; // binding 2

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
    pub fn rm_set_materials(ctx: *mut rm_ctx, count: u32, rgb: *const f32) -> c_int; // extension: count x 3 floats
    pub fn rm_resize_command_buffer(ctx: *mut rm_ctx, bytes: u64) -> c_int;
    pub fn rm_validate(ctx: *mut rm_ctx) -> c_int;
    pub fn rm_validate_program(cmd_count: u32, words: *const u32, n_words: u32, out_max_depth: *mut u32) -> c_int;
    pub fn rm_draw(ctx: *mut rm_ctx, w: u32, h: u32, row0: u32, rows: u32, out_rgba: *mut f32,
                   out_is_device: c_int, stream: *mut c_void) -> c_int;
    pub fn rm_draw_strips(ctx: *mut rm_ctx, w: u32, h: u32, strip_rows: u32, first: u32, stride: u32,
                          out_rgba: *mut f32, out_is_device: c_int, stream: *mut c_void, out_rows: *mut u32) -> c_int;
    pub fn rm_draw_batch(ctx: *mut rm_ctx, frames: *const rm_uniforms, n_frames: u32, w: u32, h: u32,
                         out_rgba: *mut f32, out_is_device: c_int, stream: *mut c_void) -> c_int;
    pub fn rm_sync(ctx: *mut rm_ctx) -> c_int;
    pub fn rm_set_option(ctx: *mut rm_ctx, key: c_int, value: i64) -> c_int;
    pub fn rm_get_info(ctx: *mut rm_ctx, key: c_int, out: *mut f64) -> c_int;
    pub fn rm_jit_log(ctx: *mut rm_ctx, buf: *mut c_char, cap: usize) -> c_int;
    pub fn rm_last_error(ctx: *mut rm_ctx) -> *const c_char;
    pub fn rm_status_string(status: c_int) -> *const c_char;
}

#[derive(Debug)]
// rm_set_option keys a host is likely to touch (include/rm_abi.h enum rm_option)
pub const RM_OPT_SPECIALIZE: c_int = 8; // 0 interpreter kernel only, 1 (default) compile per scene structure in the background, 2 blocking
pub const RM_OPT_OUTPUT_FORMAT: c_int = 10; // RM_FORMAT_*
pub const RM_FORMAT_RGBA32F: i64 = 0;
pub const RM_FORMAT_RGBA8_UNORM: i64 = 1;
pub const RM_FORMAT_BGRA8_UNORM: i64 = 2; // what an egui/wgpu surface usually is (renderer.rs:113 `target_format`)

// Opcodes of the node types the reference only names in comments (builder.rs:8,14,16-23) and that the device
// path implements as extensions: a CSGCommandType that gains these variants serialises them unchanged.
//   Plane = 2, Intersection = 102, TranslationPush = 200, TranslationPop, RotationPush, RotationPop, ScalePush, ScalePop
//   (Cylinder = 10 and SmoothUnion = 110 are this repo's own numbers for BASELINE configs 2-3)

pub struct RmError {
    pub status: i32,
    pub message: String,
}

/// Replaces `RayMarchingResources` (renderer.rs:43-49): owns the GPU state of one device.
pub struct RayMarchingResources {
    ctx: *mut rm_ctx,
}

unsafe impl Send for RayMarchingResources {} // a context may move between threads; it is not Sync

impl RayMarchingResources {
    /// `RayMarchingResources::new` (renderer.rs:51-175) without a wgpu `RenderState`.
    pub fn new(device: i32) -> Result<Self, RmError> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { rm_create(device, &mut ctx) };
        if rc != 0 {
            let msg = unsafe { CStr::from_ptr(rm_last_error(std::ptr::null_mut())) };
            return Err(RmError { status: rc, message: msg.to_string_lossy().into_owned() });
        }
        Ok(Self { ctx })
    }

    fn check(&self, rc: c_int) -> Result<(), RmError> {
        if rc == 0 {
            return Ok(());
        }
        let msg = unsafe { CStr::from_ptr(rm_last_error(self.ctx)) };
        Err(RmError { status: rc, message: msg.to_string_lossy().into_owned() })
    }

    /// `queue.write_buffer(buffer, offset, data)` (renderer.rs:213, 230, 235).
    pub fn write_buffer(&self, buffer: c_int, offset: u64, data: &[u8]) -> Result<(), RmError> {
        self.check(unsafe { rm_write_buffer(self.ctx, buffer, offset, data.as_ptr() as *const c_void, data.len() as u64) })
    }

    /// `render_pass.draw(0..4, 0..2)` (renderer.rs:254) into a host RGBA32F image (top row first).
    pub fn draw(&self, width: u32, height: u32, out_rgba: &mut [f32]) -> Result<(), RmError> {
        assert!(out_rgba.len() >= (width as usize) * (height as usize) * 4);
        self.check(unsafe { rm_draw(self.ctx, width, height, 0, height, out_rgba.as_mut_ptr(), 0, std::ptr::null_mut()) })
    }
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
