/*
 * rm_abi.h -- C ABI of the MI355X-native SDF ray-marching render path (librm_hip.so).
 *
 * This is the drop-in boundary for the `src/ray_marching` wgpu pipeline of
 * Mesoptier/ray-marching.  Everything the reference's `RayMarchingCallback::prepare`
 * hands to the GPU is three flat byte blobs (limits, CSG command buffer, uniforms) and
 * one draw; this header exposes exactly that.  All citations are file:line relative to
 * the reference root.
 *
 *   reference (Rust/wgpu)                                   this ABI
 *   ------------------------------------------------------  ---------------------------
 *   RayMarchingResources::new          renderer.rs:51-175   rm_create
 *   (drop of RayMarchingResources)                          rm_destroy
 *   queue.write_buffer(uniforms, 0,..) renderer.rs:213-222  rm_write_buffer(RM_BUF_UNIFORMS) / rm_set_uniforms
 *   queue.write_buffer(cmd_buffer,0,..) renderer.rs:230-234 rm_write_buffer(RM_BUF_COMMANDS, 0, &cmd_count, 4)
 *   queue.write_buffer(cmd_buffer,4,..) renderer.rs:235-239 rm_write_buffer(RM_BUF_COMMANDS, 4, words, 4*n) / rm_set_program
 *   create_buffer_init(limits)         renderer.rs:130-140  rm_write_buffer(RM_BUF_LIMITS) / rm_set_limits
 *   render_pass.draw(0..4, 0..2)       renderer.rs:252-254  rm_draw   (the image is shaded once, not twice)
 *   TODO "Recreate the buffers if too small" renderer.rs:229 rm_resize_command_buffer
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function returning
 * `int` returns RM_OK (0) or a negative rm_status; nothing throws or unwinds across the
 * ABI.  The caller owns every pointer it passes in; the library copies before returning.
 * An rm_ctx is bound to one GPU and is externally synchronised (one thread at a time);
 * different contexts may be used concurrently from different threads/processes.  All draws of ONE
 * context share its scratch buffers and therefore execute in the order they were issued: a draw issued
 * on another stream than the context's previous draw (a host-destination draw runs on the context's own
 * stream) first waits for everything queued on that previous stream.  Frames that should overlap use
 * one context each (RM_STREAM_OWN below).  Buffer writes are ordered with the
 * draws like queue.write_buffer is in wgpu (renderer.rs:213-239): limits and uniforms travel with each
 * draw as kernel arguments; a changed program (and the cameras of rm_draw_batch) is copied to the GPU on
 * the stream of the next draw, behind the draws already queued there, so a frame that is still in flight
 * keeps the program it was issued with.  Rewriting the command buffer with the bytes it already holds
 * (the reference does so every frame, renderer.rs:224-239) costs a memcmp and no decode or copy.
 *
 * Output image: RGBA32F, 16 bytes per pixel, row-major, top row first (framebuffer
 * orientation); pixel (px,py) has pt_screen = (-1 + 2(px+.5)/W, 1 - 2(py+.5)/H), the
 * mapping vs_main + the rasteriser imply (ray_marching.wgsl:7-20).
 */
#ifndef RM_ABI_H
#define RM_ABI_H

#if !defined(__HIPCC_RTC__) /* hipRTC provides the fixed-width types itself */
#include <stddef.h>
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define RM_ABI_VERSION 2 /* 2: kernel variants 2..11 (v2-v4) retired; rm_gather_strips, rm_host_register added */

typedef struct rm_ctx rm_ctx;

/* binding 2: `Uniforms` (ray_marching.wgsl:22-31 <-> renderer.rs:29-34), 144 bytes,
 * matrices column-major, as produced by encase's UniformBuffer::write. */
typedef struct rm_uniforms {
    float viewport_extent[2]; /* byte 0  */
    float _pad[2];            /* byte 8  */
    float inv_proj[16];       /* byte 16 */
    float inv_view[16];       /* byte 80 */
} rm_uniforms;

/* binding 0: `RayMarchLimits` (ray_marching.wgsl:78-85 <-> renderer.rs:36-41), 12 bytes. */
typedef struct rm_limits {
    float min_dist;
    float max_dist;
    uint32_t max_iter;
} rm_limits;

/* Command stream (csg/builder.rs:1-62): u32 opcode followed by f32::to_bits parameters, post-order.
 *   reference:  Sphere 0 (c.xyz, r)   Box 1 (c.xyz, half-extents.xyz)   Union 100   Subtraction 101
 *   extensions (NOT implemented by the reference, semantics in DESIGN.md section 8; only the v5 kernels):
 *               Plane 2 (n.xyz, h)   Cylinder 10 (c.xyz, r, half_h)   Intersection 102   SmoothUnion 110 (k)
 *               space transformations, the slots the reference reserves by comment (builder.rs:16-23), written
 *               Push(params), <one child>, Pop:  TranslationPush 200 (t.xyz) / Pop 201   RotationPush 202 (unit
 *               quaternion w,i,j,k) / Pop 203   ScalePush 204 (uniform s) / Pop 205
 * Any other opcode is rejected with RM_ERR_OPCODE. */

/* binding numbers of the reference's bind group (renderer.rs:60-94, 149-166) */
enum rm_buffer {
    RM_BUF_LIMITS = 0,   /* 12 B,  initial {0.01, 100.0, 100}      renderer.rs:130-140 */
    RM_BUF_COMMANDS = 1, /* 1024 B, u32 cmd_count @0, u32 words @4 renderer.rs:142-147, wgsl:146-151 */
    RM_BUF_UNIFORMS = 2  /* 144 B, initial all-zero                 renderer.rs:124-128 */
};

enum rm_status {
    RM_OK = 0,
    RM_ERR_NULL = -1,            /* required pointer is NULL */
    RM_ERR_TRUNCATED = -2,       /* a command reads past the end of the command buffer */
    RM_ERR_STACK_UNDERFLOW = -3, /* binary operator with fewer than two operands (UB in wgsl:177-180) */
    RM_ERR_STACK_OVERFLOW = -4,  /* value stack deeper than 32 (wgsl:173) */
    RM_ERR_EMPTY_RESULT = -5,    /* program leaves nothing on the stack */
    RM_ERR_OPCODE = -6,          /* opcode the reference does not define (wgsl:223-225 would yield 0.0) */
    RM_ERR_TOO_LARGE = -7,       /* write past the end of a buffer / program larger than the buffer */
    RM_ERR_RANGE = -8,           /* row band / image size out of range, or max_iter > 65536 */
    RM_ERR_DEVICE = -9,          /* a HIP call failed; see rm_last_error */
    RM_ERR_NO_DEVICE = -10,      /* no usable GPU */
    RM_ERR_ARG = -11,            /* invalid enum / option value */
    RM_ERR_TRANSFORM = -12,      /* transform push / pop (extension opcodes 200-205) not nested properly, deeper than 8,
                                    or not around exactly one value */
    RM_ERR_MATERIAL = -13        /* a Material tag (extension opcode 300) names an index >= 256, or (at draw time) one
                                    the material table does not have */
};

/* rm_set_option / rm_get_info keys */
enum rm_option {
    RM_OPT_KERNEL = 0,     /* which kernel rm_draw launches; see enum rm_kernel */
    RM_OPT_TIMING = 1,     /* 1: bracket every launch of the dominant (march) kernel with HIP events on its stream,
                              without synchronising; read with rm_get_info(RM_INFO_KERNEL_MS) */
    RM_OPT_STRICT_CAP = 2, /* reserved */
    RM_OPT_REFILL_MIN = 3, /* idle lanes of a wave that trigger a refill from the tile's ray pool, 1..64; 0 (default): 64 -- a wave
                              marches its 64 rays in step -- when the kernel prunes far primitives, else 1 */
    RM_OPT_CULL = 4,       /* 1 (default) = shade rays that provably miss the scene without marching (exact) */
    RM_OPT_BALANCE = 5,    /* dispatch order of the tiles that need marching (it never changes a pixel): 0 raster order;
                              1 most pending pixels first; 2 partially covered tiles first; 3 (default) the tiles that
                              took longest in the context's previous draw of the same shape first -- consecutive frames
                              of a view look alike, and a kernel that ends on its shortest tiles has no tail
                              (one frame at a time: +6-9 %); the first draw of a shape falls back to 1 */
    RM_OPT_WAVES_PER_TILE = 7, /* waves (1, 2, 4, 8) sharing one tile's ray pool; 0 (default): 4, or 8 for a launch of at most
                                  6000 tiles, whose duration is the latency of its heaviest tile */
    RM_OPT_WAVE_STATS = 6, /* diagnostics: the v5 kernels record per-wave timing/loop statistics (rm_read_wave_stats) */
    RM_OPT_OUTPUT_FORMAT = 10, /* enum rm_format: what rm_draw / rm_draw_strips / rm_draw_batch write (default RM_FORMAT_RGBA32F).
                                  The 8-bit formats are the output stage of SURVEY 8(f)-3: the reference's own colour target is
                                  the 8-bit egui surface (renderer.rs:113).  Only the default (v5) kernels implement them. */
    RM_OPT_PRUNE = 9,      /* specialised kernels: skip, per wave and per march step, primitives that provably cannot
                              influence the scene value there (exact; DESIGN.md "Pruning"): 0 never, 1 always, 2 (default)
                              for programs with at least 12 spheres + boxes -- on MI355X the wave-uniform tests cost more
                              than they save at 4 primitives (-6 %) and pay from there on (+6 % at 16, +14 % at 32) */
    RM_OPT_SPECIALIZE = 8  /* structure specialisation of the default (v5) march kernel: the command sequence is compiled
                              into straight-line code with hipRTC, once per program STRUCTURE (parameters stay data);
                              results are bit-identical to the interpreter kernel.
                              0 = never; 1 (default) = compile on a background thread, draw with the interpreter kernel
                              until the compiled one is ready; 2 = the first draw of a new structure waits for the compiler.
                              Without libhiprtc, or if compilation fails, the interpreter kernel keeps drawing. */
};
/* Pixel formats of the output image (row-major, top row first).  The 8-bit formats quantise the RGBA32F value the
 * shader returns (wgsl:73-75) the way a UNORM colour target does: clamp to [0,1] (NaN -> 0), times 255, round to nearest
 * even; alpha is 255.  "out_rgba" pointers then address 4 bytes per pixel instead of 16. */
enum rm_format {
    RM_FORMAT_RGBA32F = 0,      /* 16 B/pixel: r, g, b, a as binary32 */
    RM_FORMAT_RGBA8_UNORM = 1,  /* 4 B/pixel: bytes r, g, b, a */
    RM_FORMAT_BGRA8_UNORM = 2   /* 4 B/pixel: bytes b, g, r, a (wgpu's usual surface format) */
};
enum rm_kernel {
    RM_KERNEL_DEFAULT = 0,   /* the tuned kernel: v5 with the program staged in LDS, its march kernel compiled per program
                                structure (RM_OPT_SPECIALIZE) */
    RM_KERNEL_PIXEL = 1,     /* v1: north_star's literal design -- one thread per pixel, program staged in LDS, lock-step AA
                                and march loops (reference node types only); kept as the A/B reference point */
    /* 2..11 were the v2-v4 ray-pool / queue experiments of ABI version 1; retired (DESIGN.md 5 keeps their measurements) */
    /* v5: ray pool shared by the waves of a tile, full-width ray production / shading through per-wave LDS buffers, exact
       miss tests, persistent workgroups over a sorted work list; the interpreter form (RM_OPT_SPECIALIZE = 0) of the default */
    RM_KERNEL_V5 = 12,       /* program read through the scalar cache (what programs too long for LDS fall back to) */
    RM_KERNEL_V5_LDS = 13    /* program staged in LDS once per workgroup */
};
enum rm_info {
    RM_INFO_KERNEL_MS = 0,       /* mean duration (ms) of the march kernel over the launches timed since the last query
                                    (synchronises on them and resets the set) */
    RM_INFO_PROGRAM_COMMANDS = 1,
    RM_INFO_PROGRAM_WORDS = 2,
    RM_INFO_PROGRAM_DEPTH = 3,   /* maximum value-stack depth of the current program */
    RM_INFO_DEVICE = 4,
    RM_INFO_CU_COUNT = 5,
    RM_INFO_SPECIALIZED = 6,     /* 1 if the last march launch ran a structure-specialised kernel, else 0 */
    RM_INFO_JIT_STATE = 7,       /* specialisation of the current program: 0 none requested, 1 compiling, 2 ready, 3 failed
                                    (rm_jit_log has the reason) */
    RM_INFO_JIT_COMPILE_MS = 8,  /* wall time hipRTC took for the current program's kernel -- or the read from the disk cache
                                  * (RM_INFO_JIT_FROM_CACHE) --; 0 until it is ready */
    RM_INFO_PRUNED = 9,          /* which skipping rule the kernel requested for the current program carries (RM_OPT_PRUNE): 0 none,
                                  * 1 far-primitive pruning on a threshold (min / max programs), 2 the rules of programs that blend
                                  * with SmoothUnion along a top-level chain; both through wave-level culling */
    RM_INFO_INTERPRETER_LOOP = 10, /* record loop the interpreter kernels ran the program of the last march launch with: 0 the
                                    general one (value stack, every node type; also reported when a specialised kernel ran), 1 the chain loop ("a op b op c ...": no
                                    stack), 2 the chain loop over the records wave-level culling names (exact; the default for chains of a dozen leaves or more staged in
                                    LDS), 3 the tree loop (reference node types in any arrangement: one dispatch per record), 4 the tree loop over the
                                    records wave-level culling leaves (exact; trees of a dozen leaves or more and at most 128 records, staged in LDS), 5 the
                                    general record machine over the units of a blending chain that wave-level culling names (exact; eight leaves or more) */
    RM_INFO_JIT_FROM_CACHE = 11  /* 1 when the current program's kernel was read from the disk cache of compiled structures
                                  * (RM_JIT_CACHE_DIR; default: jit_cache next to the library) instead of being compiled */
};

int rm_abi_version(void);
int rm_device_count(void);

/* RayMarchingResources::new (renderer.rs:51-175).  device = HIP ordinal. */
int rm_create(int device, rm_ctx** out);
void rm_destroy(rm_ctx* ctx);

/* Queue::write_buffer(buffer, offset, data) (renderer.rs:213,230,235).  offset and size
 * must be multiples of 4 (wgpu COPY_BUFFER_ALIGNMENT) and stay inside the buffer. */
int rm_write_buffer(rm_ctx* ctx, int buffer, uint64_t offset, const void* data, uint64_t size);

/* Typed forms of the same writes. */
int rm_set_uniforms(rm_ctx* ctx, const rm_uniforms* u);
int rm_set_limits(rm_ctx* ctx, const rm_limits* l);
/* = write_buffer(cmd,0,&cmd_count) + write_buffer(cmd,4,words) after validating the program;
 * on error the command buffer is left unchanged.  cmd_count = 0 is the `csg_node == None`
 * case (renderer.rs:224-227). */
int rm_set_program(rm_ctx* ctx, uint32_t cmd_count, const uint32_t* words, uint32_t n_words);

/* Material table (extension; the reference has no materials -- README.md:11 lists them as future work -- and shades
 * every hit with (0.4, 0.7, 0.1) * diffuse, wgsl:105).  `rgb` = count x 3 floats, count in [1, 256]; entry i is the
 * albedo of the surfaces tagged by the command [300, i] (a unary postfix tag on the value on top of the stack; a
 * binary operator keeps the tag of the operand that decides its result, primitives start with tag 0).  The default
 * table is the single entry (0.4, 0.7, 0.1): programs without tags render exactly as the reference does whatever the
 * table holds in its other entries.  A draw of a program that names an index >= count fails with RM_ERR_MATERIAL.
 * Like every buffer write the table is ordered with the draws of the stream. */
int rm_set_materials(rm_ctx* ctx, uint32_t count, const float* rgb);

/* The reference's TODO (renderer.rs:229): grow the command buffer beyond 1024 bytes.
 * Contents are preserved.  bytes in [1024, 65536], multiple of 4. */
int rm_resize_command_buffer(rm_ctx* ctx, uint64_t bytes);

/* Validate the current command-buffer contents without drawing. */
int rm_validate(rm_ctx* ctx);

/* Context-free validation of a program in the reference wire format (pure host code, usable
 * without a GPU): the checks the reference leaves as UB (wgsl:177-185) or as a wgpu panic.
 * out_max_depth (nullable) receives the deepest value-stack use of the reference machine. */
int rm_validate_program(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, uint32_t* out_max_depth);

/* Diagnostics (pure host code): what the upload-time decoder makes of a command stream.  out[0..n_out) receives, in this order
 * (indices RM_PROGRAM_*): records, cone entries and slab entries of the miss-test tables, leaves left out of those tables
 * because they sit in the right operand of a Subtraction, units of wave-level culling (RM_PROGRAM_GROUPS: the name is round 2's, when
 * they were pairs of leaves; 0 when the program has none), value-stack slots the accumulator machine spills, then four 0/1 facts -- chain program (stack-free interpreter loop), prunable (far-primitive pruning applies),
 * miss test on lower bounds applies, program has space transformations --, the sphere + box leaves the program evaluates
 * (subtracted ones included), and what the automatic pruning decision (RM_OPT_PRUNE = 2) gives this program: 0 the plain kernel,
 * 1 / 2 as RM_INFO_PRUNED.
 * Same status codes as rm_validate_program. */
enum rm_program_fact {
    RM_PROGRAM_RECORDS = 0, RM_PROGRAM_CONES = 1, RM_PROGRAM_SLABS = 2, RM_PROGRAM_SUBTRACTED_LEAVES = 3, RM_PROGRAM_GROUPS = 4,
    RM_PROGRAM_SPILL_DEPTH = 5, RM_PROGRAM_IS_CHAIN = 6, RM_PROGRAM_PRUNABLE = 7, RM_PROGRAM_BOUND_WALK = 8, RM_PROGRAM_HAS_XFORMS = 9,
    RM_PROGRAM_LEAVES = 10, RM_PROGRAM_AUTO_PRUNED = 11, RM_PROGRAM_FACTS = 12
};
int rm_program_info(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, uint32_t* out, uint32_t n_out);

/* paint (renderer.rs:244-255) restricted to rows [row0,row0+rows) of a W x H target.
 * out_rgba receives rows*W*4 floats.  out_is_device = 0: host memory, filled on return.
 * out_is_device = 1: device memory on ctx's GPU; the launch is asynchronous on `stream`, a
 * hipStream_t of the caller (NULL = HIP's null stream, as everywhere in HIP); the caller
 * synchronises that stream, or calls rm_sync.  `stream` is ignored for host output. */
int rm_draw(rm_ctx* ctx, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows, float* out_rgba,
            int out_is_device, void* stream);

/* One GPU's share of an image tiled over `stride` GPUs (north-star: "the image tiles across the
 * GPUs of one node with a final host-side gather"): renders the strips first, first+stride,
 * first+2*stride, ... of strip_rows rows each (strip_rows a multiple of 8: the kernels work on 8x8-pixel tiles) in ONE launch.
 * out_rgba receives them back to back; *out_rows = number of rows written (0 if this GPU has no
 * strip).  Interleaving balances the load: the costly part of a frame is usually its centre. */
int rm_draw_strips(rm_ctx* ctx, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t first, uint32_t stride,
                   float* out_rgba, int out_is_device, void* stream, uint32_t* out_rows);

/* The final host-side gather of a tiled frame (north-star: "a final host-side gather (no RCCL collectives)"):
 * copies the strips rm_draw_strips(first, stride) wrote back to back into `strips_device` to their rows of the full
 * W x H host image `host_image` (pixel size per RM_OPT_OUTPUT_FORMAT), asynchronously on `stream` (as rm_draw's stream).
 * Every GPU of the node targets the same image -- e.g. one POSIX shared-memory segment mapped by all rank processes and
 * registered with rm_host_register -- and writes only its own rows: no rank relays another rank's pixels and the only
 * inter-process traffic is a "frame complete" flag.  With unregistered (pageable) memory the copies are synchronous. */
int rm_gather_strips(rm_ctx* ctx, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t first, uint32_t stride,
                     const void* strips_device, void* host_image, void* stream);
/* Page-locks / unlocks caller-owned host memory (hipHostRegister) so that copies into it run asynchronously at PCIe rate. */
int rm_host_register(void* ptr, uint64_t bytes);
int rm_host_unregister(void* ptr);

/* n_frames draws that differ only in their uniforms (camera-orbit batch); frame f is
 * written at out_rgba + f*W*H*4. */
int rm_draw_batch(rm_ctx* ctx, const rm_uniforms* frames, uint32_t n_frames, uint32_t W, uint32_t H,
                  float* out_rgba, int out_is_device, void* stream);

/* Waits for all work on the context's GPU (hipDeviceSynchronize). */
int rm_sync(rm_ctx* ctx);
/* HIP graphs: after its first draw of a given size and program (which allocates scratch buffers and compiles /
 * loads the scene's kernel), a device-destination rm_draw issues nothing but kernel launches on `stream` -- no
 * allocation, copy, event or synchronisation (RM_OPT_TIMING off) -- and can be stream-captured and replayed
 * (tests: test_draw_is_stream_capturable).  The uniforms travel as kernel arguments, i.e. are baked into the graph. */
/* Frames in flight without creating HIP streams in the host language: pass RM_STREAM_OWN as `stream` of a
 * device-destination draw and the launches go to the context's own stream; rm_sync_context waits for that stream
 * only.  A host that alternates two or three contexts this way (frame f -> context f % F, each with its own output
 * buffer) overlaps the tail of one frame with the start of the next: +15-20 % frames per second. */
#define RM_STREAM_OWN ((void*)(intptr_t)-1)
int rm_sync_context(rm_ctx* ctx);

int rm_set_option(rm_ctx* ctx, int key, int64_t value);
int rm_get_info(rm_ctx* ctx, int key, double* out);

/* Stream-write calibration: a fill kernel writes `bytes` of device memory `iters` times
 * with 16 B/lane stores; reports the achieved GB/s (the measured HBM-write ceiling). */
int rm_measure_write_bandwidth(rm_ctx* ctx, uint64_t bytes, int iters, double* out_gbps);

/* Self-tests of the arithmetic building blocks (used by tests/test_gpu_arithmetic.py).
 * rm_selftest_sqrt: runs the kernels' short correctly-rounded sqrt (v_rsq_f32 + one FMA-form Newton step, with its range
 * guard) against the generic one on EVERY non-negative binary32 bit pattern: wherever the guard lets the short form
 * through, the result must be the correctly rounded root, and the guard must not reject anything in [2^-96, FLT_MAX];
 * *out_mismatches must come back 0.
 * rm_selftest_ops: out[k*n + i], k = 0..7: min(a,b), max(a,b), v_min_f32(a,b), v_max_f32(a,-b), short sqrt(a),
 * generic sqrt(a), a / b, (float) i32(round(a)) -- compared on the host with the oracle's definitions. */
int rm_selftest_sqrt(rm_ctx* ctx, uint64_t* out_mismatches, uint32_t* out_first_bad_bits);
int rm_selftest_ops(rm_ctx* ctx, const float* a, const float* b, float* out, uint32_t n);
/* The two cross-lane primitives of wave-level culling (DPP row operations), one wave per 64 inputs (non-negative floats, +inf,
 * NaN): out[i] = the largest value of input i's wave (by bit pattern: a NaN wins), out[64 n_waves + i] = the minimum over the
 * inputs of the LOWER lanes of its wave (+inf for lane 0; a NaN is skipped).  tests/test_gpu_arithmetic.py compares with numpy. */
int rm_selftest_wave(rm_ctx* ctx, const float* in, uint32_t n_waves, float* out);

/* Diagnostics: per-wave records of the last draw made with RM_OPT_WAVE_STATS = 1, four u64 per
 * wave in dispatch order: [0] start, [1] end (100 MHz s_memrealtime ticks), [2] tile id << 32 |
 * map_scene iterations, [3] refills << 32 | sum over iterations of live lanes. */
int rm_read_wave_stats(rm_ctx* ctx, void* dst, uint64_t cap_bytes, uint64_t* out_bytes);

/* Message for the last error on this context (ctx may be NULL: last rm_create error). */
/* Structure specialisation (RM_OPT_SPECIALIZE), inspection entry points.  None of them needs a GPU.
 * rm_jit_source: the HIP source generated for a command stream (NUL-terminated, truncated to cap; *needed = full length + 1).
 * rm_jit_compile: generate and compile it for gfx950 with hipRTC; RM_ERR_DEVICE if libhiprtc is missing or the
 *                 compilation fails (log receives the reason).  Nothing is cached or loaded.
 * rm_jit_log: compiler / loader messages for the context's current program (empty string if none). */
#define RM_JIT_PRUNE 0x100 /* OR into waves_per_tile: generate the RM_OPT_PRUNE = 1 form of the kernel */
int rm_jit_source(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, int waves_per_tile, char* buf, size_t cap,
                  size_t* needed);
int rm_jit_compile(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, int waves_per_tile, double* compile_ms,
                   size_t* code_bytes, char* log, size_t log_cap);
int rm_jit_log(rm_ctx* ctx, char* buf, size_t cap);

const char* rm_last_error(rm_ctx* ctx);
const char* rm_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* RM_ABI_H */
