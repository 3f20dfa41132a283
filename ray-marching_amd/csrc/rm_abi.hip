// rm_abi.hip -- host side of librm_hip.so: the C ABI declared in include/rm_abi.h.
// Mirrors RayMarchingResources / RayMarchingCallback::{prepare,paint}
// (src/ray_marching/renderer.rs:43-49, 51-175, 195-256 of the reference).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rm_abi.h"
#include "rm_decode.h"
#include "rm_jit.h"
#include "rm_device.h"
#include "rm_kernels.h"
#include "rm_interp.h"
#include "rm_kernel_v5.h"

#define RM_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

constexpr uint64_t kRefCmdBufferBytes = 1024;  // renderer.rs:142-147
constexpr uint64_t kMaxCmdBufferBytes = 65536;
constexpr uint32_t kMaxDim = 1u << 16;
constexpr uint32_t kMaxIter = 1u << 16;
constexpr uint32_t kPruneLeaves = 12;  // RM_OPT_PRUNE = 2: programs that EVALUATE this many spheres + boxes (RmDecoded::n_leaves: subtracted
                                        // ones included, they have no miss-test slot but cost the same) get the pruned kernel

constexpr uint32_t kBlendPruneLeaves = 8;  // ... and programs that blend, the rules of their chains (rm_units.h), from this many
thread_local std::string g_create_error;

}  // namespace

struct rm_ctx {
    int device = -1;
    int cu_count = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Draws of one context share its scratch (program copy, tile cost / order / counters / measurements).  They are
    // ordered by their stream; when a draw arrives on ANOTHER stream than the previous one (a host-destination draw
    // runs on the context's own stream, device-destination draws on the caller's), the new stream first waits for
    // everything queued on the previous one: order_with_previous().
    hipStream_t last_stream = nullptr;
    bool last_stream_valid = false;
    hipEvent_t ev_order = nullptr;
    // RM_OPT_TIMING: one event pair per timed launch of the dominant kernel, read back (and
    // reset) by rm_get_info(RM_INFO_KERNEL_MS): no synchronisation inside the launch path.
    std::vector<std::pair<hipEvent_t, hipEvent_t>> tev;
    size_t tev_used = 0;
    // host shadows of the three buffers of the reference's bind group
    rm_limits limits{0.01f, 100.0f, 100u};  // renderer.rs:133-137
    rm_uniforms uniforms{};                 // Uniforms::default(), renderer.rs:126
    std::vector<uint32_t> cmd;              // [0] = cmd_count, [1..] = words; zero-initialised like a wgpu buffer
    bool cmd_dirty = true;
    int cmd_status = RM_OK;
    // decoded program, device copy
    RmDecoded decoded;
    RmRecord* d_prog = nullptr;
    size_t d_prog_cap = 0;
    float4* d_bounds = nullptr;  // world-space bounding spheres of a program with transforms (RmDecoded::bounds)
    size_t d_bounds_cap = 0;
    // materials (extension): the tagged decoding of the program and the albedo table
    RmRecord* d_mprog = nullptr;
    size_t d_mprog_cap = 0;
    std::vector<float4> materials{make_float4(0.4f, 0.7f, 0.1f, 0.0f)};  // wgsl:105
    float4* d_materials = nullptr;  // RM_MAX_MATERIALS entries
    bool materials_dirty = true;
    // scratch for host-destination draws and batch uniforms
    float* d_out = nullptr;
    size_t d_out_bytes = 0;
    rm_uniforms* d_frames = nullptr;
    size_t d_frames_cap = 0;
    // options / info
    int kernel = RM_KERNEL_DEFAULT;
    uint32_t refill_min_v5 = 0;  // 0: by what the kernel does with coherent rays (launch_v5_w)
    bool cull = true;
    int balance = 3;  // RM_OPT_BALANCE: 0 raster order, 1 most pending pixels first, 2 partially covered tiles first,
                      // 3 (default) longest tiles of the previous draw of the same shape first
    int waves_per_tile = 0;  // 0: by the size of the launch (launch_v5)
    bool wave_stats = false;
    unsigned long long* d_stats = nullptr;
    size_t d_stats_bytes = 0, stats_valid_bytes = 0;
    uint32_t* d_cost = nullptr;   // per-tile cost estimates / dispatch order of the balance pre-pass
    uint32_t* d_order = nullptr;
    uint32_t* d_counters = nullptr;  // v5: {work-list length, cursor} per frame
    uint32_t* d_measured = nullptr;  // v5, RM_OPT_BALANCE = 3: per-tile durations of the previous draw of the same shape
    size_t d_measured_cap = 0;
    uint64_t measured_shape = 0;     // hash of (W, rows, strips, frames) the measurements belong to; 0 = none yet
    size_t d_counters_cap = 0;
    size_t d_tiles_cap = 0;
    bool timing = false;
    double last_kernel_ms = 0.0;
    // structure specialisation (rm_jit.h): 0 off, 1 compile in the background and switch over when
    // ready, 2 wait for the compiler at the first draw of a new structure
    int specialize = 1;
    int out_format = RM_FORMAT_RGBA32F;  // RM_OPT_OUTPUT_FORMAT
    int prune = 2;  // RM_OPT_PRUNE: far-primitive pruning in specialised kernels: 0 off, 1 on, 2 (default) on for programs
                    // with at least kPruneLeaves spheres + boxes (measured: 4 leaves -6 %, 16 leaves +6 %, 32 leaves +14 %)
    uint64_t prog_gen = 0;  // bumped whenever the decoded program changes
    std::shared_ptr<rmjit::Entry> spec;
    uint64_t spec_gen = ~0ull;
    int spec_wpt = 0;
    int spec_pruned = 0;            // rmjit::PRUNE_* of the requested kernel
    bool spec_stats = false;        // ... and whether it keeps the counters of RM_OPT_WAVE_STATS
    bool last_specialized = false;  // the last march launch ran a specialised kernel
    int last_loop = 0;              // RM_INFO_INTERPRETER_LOOP of the last march launch
    // stream-ordered uploads (program records, bounds, batch uniforms): four pinned staging buffers, see upload()
    struct Staging { void* host = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool pending = false; };
    Staging staging[4];
    unsigned staging_next = 0;
    std::string err;
};

namespace {

int fail(rm_ctx* c, int status, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return status;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(ctx, RM_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                       \
    } while (0)

// Host data -> device scratch of the context, ordered on the stream the next draw is issued on (what
// queue.write_buffer is to wgpu, renderer.rs:213-239): a draw of the previous program that is still queued or
// running on `s` keeps reading the previous contents, and the caller's memory is free again when this returns.
// The bytes wait in one of four pinned staging buffers; only a fifth upload with the first still pending blocks.
int upload(rm_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t s) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) (void)hipGetLastError();
    else if (capturing != hipStreamCaptureStatusNone)
        return fail(c, RM_ERR_ARG, "the program or batch changed during stream capture: draw once before capturing");
    rm_ctx::Staging& st = c->staging[c->staging_next++ & 3u];
    if (st.pending) {
        HIP_TRY(c, hipEventSynchronize(st.done));
        st.pending = false;
    }
    if (bytes > st.cap) {
        if (st.host) (void)hipHostFree(st.host);
        st.host = nullptr;
        st.cap = 0;
        const size_t cap = std::max<size_t>(4096, bytes + bytes / 2);
        HIP_TRY(c, hipHostMalloc(&st.host, cap, hipHostMallocDefault));
        st.cap = cap;
    }
    if (!st.done) HIP_TRY(c, hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
    std::memcpy(st.host, src, bytes);
    HIP_TRY(c, hipMemcpyAsync(dst, st.host, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(st.done, s));
    st.pending = true;
    return RM_OK;
}

int ensure_program(rm_ctx* c, hipStream_t s) {
    if (!c->cmd_dirty) return c->cmd_status;
    c->cmd_dirty = false;
    const uint32_t cap_words = (uint32_t)c->cmd.size() - 1u;
    RmDecoded d;
    int rc = rm_decode_program(c->cmd[0], c->cmd.data() + 1, cap_words, &d);
    if (rc != RM_OK) {
        c->cmd_status = rc;
        return fail(c, rc, "invalid CSG program in command buffer: %s", rm_status_string(rc));
    }
    // the unit records of wave-level culling (RmDecoded::units) follow the program's records in the same buffer
    std::vector<RmRecord> image = d.rec;
    image.insert(image.end(), d.units.begin(), d.units.end());
    image.insert(image.end(), d.tree.begin(), d.tree.end());  // ... and the operand masks of a tree program's records (RmDecoded::tree)
    if (image.size() > c->d_prog_cap) {
        if (c->d_prog) (void)hipFree(c->d_prog);
        c->d_prog = nullptr;
        c->d_prog_cap = 0;
        size_t cap = std::max<size_t>(64, image.size() * 2);
        hipError_t e = hipMalloc(&c->d_prog, cap * sizeof(RmRecord));
        if (e != hipSuccess) {
            c->cmd_dirty = true;
            return fail(c, RM_ERR_DEVICE, "hipMalloc(program) failed: %s", hipGetErrorString(e));
        }
        c->d_prog_cap = cap;
    }
    if (!image.empty()) {
        if (int urc = upload(c, c->d_prog, image.data(), image.size() * sizeof(RmRecord), s)) {
            c->cmd_dirty = true;
            return urc;
        }
    }
    if (!d.bounds.empty()) {
        const size_t n = d.bounds.size() / 4u;
        if (n > c->d_bounds_cap) {
            if (c->d_bounds) (void)hipFree(c->d_bounds);
            c->d_bounds = nullptr;
            c->d_bounds_cap = 0;
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_bounds), std::max<size_t>(64, 2 * n) * sizeof(float4));
            if (e != hipSuccess) {
                c->cmd_dirty = true;
                return fail(c, RM_ERR_DEVICE, "hipMalloc(bounds) failed: %s", hipGetErrorString(e));
            }
            c->d_bounds_cap = std::max<size_t>(64, 2 * n);
        }
        if (int urc = upload(c, c->d_bounds, d.bounds.data(), d.bounds.size() * sizeof(float), s)) {
            c->cmd_dirty = true;
            return urc;
        }
    }
    if (!d.mrec.empty()) {
        if (d.mrec.size() > c->d_mprog_cap) {
            if (c->d_mprog) (void)hipFree(c->d_mprog);
            c->d_mprog = nullptr;
            c->d_mprog_cap = 0;
            const size_t cap = std::max<size_t>(64, d.mrec.size() * 2);
            hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->d_mprog), cap * sizeof(RmRecord));
            if (e != hipSuccess) {
                c->cmd_dirty = true;
                return fail(c, RM_ERR_DEVICE, "hipMalloc(material program) failed: %s", hipGetErrorString(e));
            }
            c->d_mprog_cap = cap;
        }
        if (int urc = upload(c, c->d_mprog, d.mrec.data(), d.mrec.size() * sizeof(RmRecord), s)) {
            c->cmd_dirty = true;
            return urc;
        }
    }
    c->decoded = std::move(d);
    c->prog_gen++;
    c->cmd_status = RM_OK;
    return RM_OK;
}

// The material table of a tagged program, on the stream of the draw that needs it.
int ensure_materials(rm_ctx* c, hipStream_t s) {
    if (!c->decoded.has_materials) return RM_OK;
    if (c->decoded.max_material >= c->materials.size())
        return fail(c, RM_ERR_MATERIAL, "the program tags a surface with material %u, the material table has %zu entries "
                    "(rm_set_materials)", c->decoded.max_material, c->materials.size());
    if (!c->d_materials) HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(&c->d_materials), RM_MAX_MATERIALS * sizeof(float4)));
    if (c->materials_dirty) {
        if (int rc = upload(c, c->d_materials, c->materials.data(), c->materials.size() * sizeof(float4), s)) return rc;
        c->materials_dirty = false;
    }
    return RM_OK;
}

// Which skipping rule the specialised kernel of a program gets (rmjit::PRUNE_*): RM_OPT_PRUNE 0 none, 1 whichever applies, 2
// (default) whichever applies if the program evaluates enough leaves for the tests to pay.
int prune_kind(const RmDecoded& d, int option) {
    if (option == 0) return rmjit::PRUNE_NONE;
    static const uint32_t blend_leaves = std::getenv("RM_BLEND_PRUNE_LEAVES") ? (uint32_t)std::atoi(std::getenv("RM_BLEND_PRUNE_LEAVES")) : kBlendPruneLeaves;
    if (d.unit_mode == RM_UNITS_LATTICE && (option == 1 || d.n_leaves >= kPruneLeaves)) return rmjit::PRUNE_LATTICE;
    if (d.unit_mode == RM_UNITS_BLEND && (option == 1 || d.n_leaves >= blend_leaves)) return rmjit::PRUNE_BLEND;
    return rmjit::PRUNE_NONE;
}

// The specialised march kernel for the current program and WPT waves per tile on this device, or
// nullptr: specialisation off, not possible, still compiling (mode 1) or failed -- the caller then
// launches the interpreter kernel.  Never an error.
hipFunction_t specialised_kernel(rm_ctx* c, int wpt) {
    if (!c->specialize || !rmjit::can_specialise(c->decoded.rec)) return nullptr;
    if (c->spec_gen != c->prog_gen || c->spec_wpt != wpt || c->spec_stats != c->wave_stats) {
        // same structure as before (parameters moved): the key lookup finds the same entry
        const int prune = prune_kind(c->decoded, c->prune);
        // (the per-wave counters of RM_OPT_WAVE_STATS are compiled into a kernel of their own: the default one does without)
        c->spec = rmjit::Cache::get().request(c->decoded.rec, c->decoded.mrec, wpt, prune | (c->wave_stats ? rmjit::KERNEL_WITH_STATS : 0));
        c->spec_pruned = prune;
        c->spec_stats = c->wave_stats;
        c->spec_gen = c->prog_gen;
        c->spec_wpt = wpt;
    }
    rmjit::Entry* e = c->spec.get();
    if (!e) return nullptr;
    const rmjit::Entry::State st = c->specialize >= 2 ? e->wait() : e->peek();
    if (st != rmjit::Entry::READY) return nullptr;
    std::lock_guard<std::mutex> lk(e->m);
    rmjit::Entry::Loaded& l = e->loaded[c->device];
    if (!l.function && !l.module && !e->code.empty()) {
        auto load = [&]() -> bool {
            hipModule_t mod = nullptr;
            hipFunction_t fn = nullptr;
            if (hipModuleLoadData(&mod, e->code.data()) == hipSuccess &&
                hipModuleGetFunction(&fn, mod, rmjit::kernel_name()) == hipSuccess) {
                l.module = mod;
                l.function = fn;
                return true;
            }
            (void)hipGetLastError();
            if (mod) (void)hipModuleUnload(mod);
            return false;
        };
        bool ok = load();
        if (!ok && !e->cached_source.empty()) {
            // the code object came from the disk cache and the loader rejects it (written by another driver stack, or
            // damaged in a way the checksum cannot see): drop the file and compile the source afresh, once
            const std::string src = std::move(e->cached_source);
            e->cached_source.clear();
            const std::string file = rmjit::cache_path(src);
            if (!file.empty()) std::remove(file.c_str());
            std::string log;
            double ms = 0.0;
            e->code.clear();
            if (rmjit::compile(src, &e->code, &log, &ms)) {
                e->compile_ms = ms;
                e->log += "\ncached code object rejected by the loader; recompiled";
                ok = load();
            } else {
                e->log += "\n" + log;
            }
        }
        if (ok) {
            // an evicted entry (more than 256 structures in one process) may still have launches in flight on a
            // context that moved on to another program: wait for the device before the code object goes away
            e->unload = [](void* m, int device) {
                int current = -1;
                if (hipGetDevice(&current) != hipSuccess) current = -1;
                if (hipSetDevice(device) == hipSuccess) (void)hipDeviceSynchronize();
                (void)hipModuleUnload(static_cast<hipModule_t>(m));
                if (current >= 0) (void)hipSetDevice(current);
                (void)hipGetLastError();
            };
        } else {
            e->log += "\nloading the compiled module failed";
            e->code.clear();  // do not try again
        }
    }
    return static_cast<hipFunction_t>(l.function);
}

// Device scratch that only ever grows (allocated outside of any timed or captured region the first
// time a size is seen; later draws of the same size allocate nothing).
template <class T>
int grow_device(rm_ctx* c, T** buf, size_t* cap, size_t need_elems) {
    if (need_elems <= *cap) return RM_OK;
    if (*buf) (void)hipFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void**>(buf), need_elems * sizeof(T)));
    *cap = need_elems;
    return RM_OK;
}
int ensure_tile_buffers(rm_ctx* c, size_t n_tiles_total);
int ensure_stats(rm_ctx* c, RmLaunch& L, size_t n_waves);

int finish_launch(rm_ctx* c, hipStream_t s);
int time_begin(rm_ctx* c, hipStream_t s);
int time_end(rm_ctx* c, hipStream_t s);

// LDS behind the program copy of a march workgroup: {pool cursor, tile slot, veto, pad} + the 16 AA sample offsets (+ pad)
constexpr size_t kV5TailBytes = 16u + 144u;

template <int WPT>
int launch_v5_w(rm_ctx* c, const RmLaunch& L_in, bool lds, uint32_t n_frames, hipStream_t s) {
    RmLaunch L = L_in;
    bool cull = c->cull && L.n_rec <= 256u && !c->decoded.cull_veto;
    if (L.n_rec == 0u && L.max_dist < L.min_dist) cull = false;  // see launch_multi_w
    L.n_cull = cull ? L.n_rec : 0u;
    // interpreter kernels: which map_scene loop a chain program takes (RmLaunch::flags).  RM_CHAIN_MODE (diagnostics):
    // 0 the general record loop, 1 the chain / tree loops over every record, 2 (default) over the records the wave's unit mask names / leaves
    static const int chain_mode = std::getenv("RM_CHAIN_MODE") ? std::atoi(std::getenv("RM_CHAIN_MODE")) : 2;
    const bool chain = c->decoded.is_chain && chain_mode > 0;
    // ... and whether the interpreter uses the wave-level culling mask (bit 3): 2 (default) yes, wherever the program has units
    // Measured (profiles/r03_interpreter_loops.txt): over a chain the mask names the records to fetch at all -- 64-node scene at 4K
    // 15.5 -> 5.3 ms, metric scene 1.23 -> 0.71 ms --, and over a tree the records that are left once operands without a needed
    // leaf are dropped with their operators (map_scene_tree_masked); but it has a fixed price per evaluation (~250 cycles) that four
    // leaves do not repay (8-node scene 0.44 -> 0.66 ms), and in the general loop, where a skipped record is still fetched and
    // decoded, it loses (the blended scene 4.0 -> 4.4): chains and trees of a dozen leaves or more.
    // bit 4: tree program (every record one of the eight fast shapes): the interpreter's one-dispatch-per-record loop.  RM_CHAIN_MODE=0
    // keeps the general loop for everything (diagnostics)
    const bool tree = c->decoded.is_tree && chain_mode > 0;
    static const bool tree_masks = !(std::getenv("RM_TREE_MASKS") && std::atoi(std::getenv("RM_TREE_MASKS")) == 0);  // A/B
    const bool tree_units = tree && !chain && !c->decoded.has_extensions && !c->decoded.tree.empty() && tree_masks && lds;
    // ... and a chain of blends (the wider interpreter: SmoothUnion is an extension): the record machine over the units the mask names
    static const uint32_t blend_leaves = std::getenv("RM_BLEND_PRUNE_LEAVES") ? (uint32_t)std::atoi(std::getenv("RM_BLEND_PRUNE_LEAVES")) : kBlendPruneLeaves;
    const bool blend_units = c->decoded.unit_mode == RM_UNITS_BLEND && lds && !chain && chain_mode >= 2 && c->decoded.n_leaves >= blend_leaves;
    const bool units = blend_units ||
                       (c->decoded.unit_mode == RM_UNITS_LATTICE && (chain || tree_units) && chain_mode >= 2 && c->decoded.n_leaves >= kPruneLeaves);
    L.n_tree = units && tree_units ? (uint32_t)c->decoded.tree.size() : 0u;
    if (L.n_tree != 0u) L.spill_depth += 1u;  // (map_scene_tree_masked spills at every push)
    L.flags = (cull ? 1u : 0u) | (chain ? 4u : 0u) | (units ? 8u : 0u) | (tree ? 16u : 0u);
    // (the scalar-cache variant has no unit records at hand)
    c->last_loop = (L.flags & 4u) ? (((L.flags & 8u) && lds) ? 2 : 1) : (tree && !c->decoded.has_extensions) ? (L.n_tree != 0u ? 4 : 3) : blend_units ? 5 : 0;
    // programs that blend: a ray the plain miss tests cannot clear (every bound is inflated by the blend radius) gets the
    // program run on lower bounds of its leaves along the ray.  RM_BOUND_WALK=0 (diagnostics) keeps the plain tests only.
    static const bool bound_walk_on = !(std::getenv("RM_BOUND_WALK") && std::atoi(std::getenv("RM_BOUND_WALK")) == 0);
    if (cull && c->decoded.bound_walk && bound_walk_on) L.flags |= 32u;
    // diagnostics: RM_PRE_NEED_MAX=n, the largest number of pixels of a clear tile the pre-pass finishes sample-parallel (default 24;
    // 0: only tiles that need none, 64: always)
    static const int pre_need_max = std::getenv("RM_PRE_NEED_MAX") ? std::atoi(std::getenv("RM_PRE_NEED_MAX")) : -1;
    if (pre_need_max >= 0 && pre_need_max <= 64) L.flags |= (uint32_t)(pre_need_max + 1) << 8;
    const uint32_t n_tiles = ((L.W + 7u) / 8u) * ((L.rows + 7u) / 8u);
    if (!cull) L.n_cone = L.n_slab = 0u;
    const size_t cull_bytes = (size_t)L.n_cone * 16u + (size_t)L.n_slab * 48u;
    // structure-specialised kernel (values live in registers: no LDS spill stack)
    hipFunction_t spec_fn = lds ? specialised_kernel(c, WPT) : nullptr;
    if (spec_fn) { L.spill_depth = 0u; L.n_tree = 0u; }
    // the material evaluation of a tagged program borrows the spill area: (distance, index) pairs + saved positions
    // (a specialised kernel with the generated material walk keeps those pairs in registers too)
    if (L.n_mrec != 0u && !(spec_fn && c->spec && c->spec->material_walk))
        L.spill_depth = std::max(L.spill_depth, 2u * c->decoded.mat_spill_depth + 3u * c->decoded.mat_xform_depth);
    L.wave_dwords = rmk::V5_WAVE_DWORDS - ((spec_fn && c->spec && c->spec->taps4 && L.n_mrec == 0u) ? rmk::V5_TN_DWORDS : 0u);
    const size_t shmem = (size_t)(1024u + WPT * L.wave_dwords) * 4u +
                         (size_t)L.spill_depth * 64u * WPT * 4u + cull_bytes +
                         (lds ? (size_t)(L.n_rec + L.n_grp + L.n_tree) * sizeof(RmRecord) : 0u) + kV5TailBytes +
                         (L.n_mrec != 0u ? 1024u : 0u);
    if (shmem > 64u * 1024u) return fail(c, RM_ERR_TOO_LARGE, "program needs %zu bytes of LDS per tile", shmem);
    // pre-pass buffers: cost + work list per tile, {count, cursor} per frame
    if (int rc = ensure_tile_buffers(c, (size_t)n_tiles * n_frames)) return rc;
    if (int rc = grow_device(c, &c->d_counters, &c->d_counters_cap, (size_t)n_frames * 4u)) return rc;
    const uint32_t pre_tiles_per_group = rmk::V5_PRE_TILES * rmk::V5_PRE_TILES_PER_WAVE;
    hipLaunchKernelGGL(rmk::rm_tile_pre_v5, dim3((n_tiles + pre_tiles_per_group - 1u) / pre_tiles_per_group, 1, n_frames),
                       dim3(64u * rmk::V5_PRE_TILES), 16u + cull_bytes + (size_t)(L.n_cone + L.n_slab) * 8u, s, L, c->d_cost, n_tiles);
    // RM_OPT_BALANCE = 3: the march kernel records how long every tile took; the next draw of the same shape
    // dispatches the longest first (consecutive frames of an interactive view or an orbit look alike)
    const uint32_t* prev = nullptr;
    uint32_t* measured = nullptr;
    if (c->balance == 3) {
        const size_t need = (size_t)n_tiles * n_frames;
        if (need > c->d_measured_cap) {
            if (int rc = grow_device(c, &c->d_measured, &c->d_measured_cap, need)) return rc;
            c->measured_shape = 0;
        }
        const uint64_t shape = ((uint64_t)L.W << 40) ^ ((uint64_t)L.rows << 20) ^ ((uint64_t)L.strip_rows << 12) ^
                               ((uint64_t)L.strip_first << 6) ^ (uint64_t)L.strip_stride ^ ((uint64_t)n_frames << 52) ^ 1ull;
        if (c->measured_shape == shape) prev = c->d_measured;
        else HIP_TRY(c, hipMemsetAsync(c->d_measured, 0, need * sizeof(uint32_t), s));
        c->measured_shape = shape;
        measured = c->d_measured;
    }
    hipLaunchKernelGGL(rmk::rm_tile_sort_v5, dim3(n_frames), dim3(1024), 0, s, L, c->d_cost, c->d_order, c->d_counters,
                       n_tiles, (uint32_t)c->balance, prev);
    rmk::V5Work work{c->d_order, c->d_counters, measured};
    // persistent grid: about as many workgroups as fit the chip (LDS, 32 waves per CU), never more than tiles
    uint32_t per_cu = (uint32_t)std::min<size_t>(32u / WPT, (160u * 1024u) / shmem);
    if (per_cu < 1u) per_cu = 1u;
    // ... but never more than 24 waves of ONE launch on a CU: where a seventh workgroup would fit (kernels of at most 72 vector
    // registers and 22.8 KB of LDS), a frame drawn alone is slower with it (a tile's four waves get a seventh of the CU instead of a
    // sixth, and the launch ends with its last tiles: -4 %), while the free slot lets the next frame's launch start on the same CU
    // (frames in flight +6 %).  RM_WG_PER_CU_CAP (A/B): another cap, 0 none.
    static const int per_cu_cap = std::getenv("RM_WG_PER_CU_CAP") ? std::atoi(std::getenv("RM_WG_PER_CU_CAP")) : 24 / WPT;
    if (per_cu_cap > 0 && per_cu > (uint32_t)per_cu_cap) per_cu = (uint32_t)per_cu_cap;
    const uint32_t n_wg = std::min<uint32_t>(n_tiles, (uint32_t)std::max(1, c->cu_count) * per_cu);
    dim3 grid(n_wg, 1, n_frames);
    if (int rc = ensure_stats(c, L, (size_t)n_wg * n_frames * WPT)) return rc;
    if (int rc = time_begin(c, s)) return rc;
    // reference-only programs run the lean interpreter (chain and tree loops only); extension node types select the wider one,
    // which has the general record loop (RM_CHAIN_MODE=0, diagnostics: everything takes that one)
    const bool ext = c->decoded.has_extensions || !c->decoded.is_tree || chain_mode == 0;
    c->last_specialized = spec_fn != nullptr;
    // When a wave takes new rays.  With far-primitive pruning every evaluation tests which primitives are near for ANY lane:
    // lanes at unrelated march depths (a lane refilled the moment it retires) keep most of them near, while 64 rays started
    // together stay in step -- far from everything at first, then close to one or two primitives each -- and the union shrinks
    // with them.  That is worth more than the lanes that idle until the last ray of the batch is done (metric frame 0.505 ->
    // 0.482 ms, 64-node scene +7 %; profiles/r02_refill_threshold_ab.txt); without pruning nothing is gained and the idle
    // lanes cost (8-node scene -17 %, the blended scene -2 %), so those kernels keep refilling lane by lane.
    static const bool blend_in_step = !(std::getenv("RM_BLEND_IN_STEP") && std::atoi(std::getenv("RM_BLEND_IN_STEP")) == 0);  // A/B
    const bool pruning = spec_fn ? (c->spec && (c->spec_pruned == rmjit::PRUNE_LATTICE || (c->spec_pruned == rmjit::PRUNE_BLEND && blend_in_step)))
                                 : ((L.flags & 8u) != 0u && lds);
    const uint32_t refill_auto = c->refill_min_v5 != 0u ? c->refill_min_v5 : (pruning ? 64u : 1u);
    if (spec_fn) {
        uint32_t n_tiles_arg = n_tiles, refill = refill_auto;
        void* args[] = {&L, &work, &n_tiles_arg, &refill};
        hipError_t e = hipModuleLaunchKernel(spec_fn, grid.x, grid.y, grid.z, 64u * WPT, 1, 1, (unsigned)shmem, s, args, nullptr);
        if (e != hipSuccess) return fail(c, RM_ERR_DEVICE, "launch of the specialised kernel failed: %s", hipGetErrorString(e));
    } else if (lds && !ext) {  // one lean kernel per record loop (c->last_loop: 1 .. 4 here)
        switch (c->last_loop) {
        case 1: hipLaunchKernelGGL((rmk::rm_render_v5_lean<WPT, 1>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto); break;
        case 2: hipLaunchKernelGGL((rmk::rm_render_v5_lean<WPT, 2>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto); break;
        case 3: hipLaunchKernelGGL((rmk::rm_render_v5_lean<WPT, 3>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto); break;
        default: hipLaunchKernelGGL((rmk::rm_render_v5_lean<WPT, 4>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto); break;
        }
    }
    else if (lds)
        hipLaunchKernelGGL((rmk::rm_render_v5<rmk::ProgLds, true, WPT, true>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto);
    else if (!ext)
        hipLaunchKernelGGL((rmk::rm_render_v5<rmk::ProgSmem, false, WPT, false>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto);
    else
        hipLaunchKernelGGL((rmk::rm_render_v5<rmk::ProgSmem, false, WPT, true>), grid, dim3(64 * WPT), shmem, s, L, work, n_tiles, refill_auto);
    if (int rc = time_end(c, s)) return rc;
    return finish_launch(c, s);
}

int launch_v5(rm_ctx* c, const RmLaunch& L, bool lds, uint32_t n_frames, hipStream_t s) {
    // Waves per tile.  A march launch lasts at least as long as its heaviest tile (1024 rays through one workgroup); four waves
    // per tile give the best throughput when there are tiles to fill the chip with, eight halve that latency and win once a
    // launch has fewer tiles than the chip has room for -- one GPU's share of a frame tiled over eight (DESIGN.md section 7:
    // 144 rows of 1080p: 0.095 -> 0.084 ms per frame; the whole frame: 0.546 -> 0.613).
    const size_t launch_tiles = (size_t)((L.W + 7u) / 8u) * ((L.rows + 7u) / 8u) * n_frames;
    int wpt = c->waves_per_tile != 0 ? c->waves_per_tile : (launch_tiles <= 6000u ? 8 : 4);
    const size_t cull_bytes = L.n_rec <= 256u ? (size_t)L.n_cone * 16u + (size_t)L.n_slab * 48u : 0u;  // tables exist up to 256 records
    const size_t prog_bytes = (size_t)(L.n_rec + L.n_grp + c->decoded.tree.size()) * sizeof(RmRecord);  // (the tree table: at most 4 KB, when the interpreter uses it)
    const size_t depth = std::max<size_t>(L.spill_depth + (c->decoded.tree.empty() ? 0u : 1u), L.n_mrec != 0u ? 2u * c->decoded.mat_spill_depth + 3u * c->decoded.mat_xform_depth : 0u);
    const size_t per_wave = rmk::V5_WAVE_DWORDS * 4u + depth * 256u;
    const size_t fixed = 4096u + cull_bytes + kV5TailBytes + (L.n_mrec != 0u ? 1024u : 0u);
    // A long program (rm_resize_command_buffer admits 64 KB of commands, ~2 700 leaves = 85 KB of records) does not fit
    // a workgroup's LDS next to the ray buffers: it is then read through the scalar cache instead (ProgSmem).
    if (lds && fixed + prog_bytes + per_wave > 60u * 1024u) lds = false;
    const size_t fixed_all = fixed + (lds ? prog_bytes : 0u);
    while (wpt > 1 && fixed_all + (size_t)wpt * per_wave > 48u * 1024u) wpt /= 2;
    switch (wpt) {
    case 1: return launch_v5_w<1>(c, L, lds, n_frames, s);
    case 2: return launch_v5_w<2>(c, L, lds, n_frames, s);
    case 8: return launch_v5_w<8>(c, L, lds, n_frames, s);
    default: return launch_v5_w<4>(c, L, lds, n_frames, s);
    }
}

struct StripSpec { uint32_t rows = 0, first = 0, stride = 0; };

int launch(rm_ctx* c, const rm_uniforms* frames_dev, uint32_t n_frames, uint32_t W, uint32_t H, uint32_t row0,
           uint32_t rows, float* d_out, hipStream_t s, StripSpec strips = StripSpec()) {
    RmLaunch L;
    L.strip_rows = strips.rows; L.strip_first = strips.first; L.strip_stride = strips.stride;
    L.prog = c->d_prog;
    L.n_rec = (uint32_t)c->decoded.rec.size();
    L.n_grp = (uint32_t)c->decoded.units.size();
    L.n_tree = 0u;  // (set by launch_v5_w when the interpreter runs the masked tree loop)
    L.unit_mode = c->decoded.unit_mode;
    L.unit_kmax = c->decoded.unit_kmax;
    L.value_spill_depth = c->decoded.spill_depth;
    L.spill_depth = c->decoded.spill_depth + 3u * c->decoded.xform_depth;  // saved positions follow the value stack
    L.bounds = c->decoded.has_xforms ? c->d_bounds : nullptr;
    L.mprog = c->d_mprog;
    L.n_mrec = (uint32_t)c->decoded.mrec.size();
    L.mat_value_depth = c->decoded.mat_spill_depth;
    L.materials = c->d_materials;
    L.n_cull = 0;
    L.flags = 0;
    L.n_cone = c->decoded.n_sphere;
    L.n_slab = c->decoded.n_box;
    L.smooth_slack = (float)c->decoded.smooth_slack;
    L.scene_scale = c->decoded.scene_scale;
    L.min_dist = c->limits.min_dist;
    L.max_dist = c->limits.max_dist;
    L.max_iter = c->limits.max_iter;
    L.W = W; L.H = H; L.row0 = row0; L.rows = rows;
    L.out = d_out;
    L.out_format = (uint32_t)c->out_format;
    if (c->out_format != RM_FORMAT_RGBA32F && !(c->kernel == RM_KERNEL_DEFAULT || c->kernel == RM_KERNEL_V5 || c->kernel == RM_KERNEL_V5_LDS))
        return fail(c, RM_ERR_ARG, "kernel variant %d writes RGBA32F only (8-bit output formats need the default kernels)", c->kernel);
    L.frames = frames_dev;
    L.order = nullptr;
    L.stats = nullptr;
    L.u = c->uniforms;
    int kernel = c->kernel == RM_KERNEL_DEFAULT ? RM_KERNEL_V5_LDS : c->kernel;
    if (kernel == RM_KERNEL_V5 || kernel == RM_KERNEL_V5_LDS) return launch_v5(c, L, kernel == RM_KERNEL_V5_LDS, n_frames, s);
    if (kernel != RM_KERNEL_PIXEL) return fail(c, RM_ERR_ARG, "kernel variant %d is not available", kernel);
    // v1: north_star's literal design (one thread per pixel, program staged in LDS, lock-step loops); reference node types only
    if (c->decoded.has_extensions)
        return fail(c, RM_ERR_ARG, "kernel variant %d renders reference node types only (program uses extension nodes)", kernel);
    if (int rc = time_begin(c, s)) return rc;
    {
        dim3 grid((W + 15u) / 16u, (rows + 15u) / 16u, n_frames);
        size_t shmem = (size_t)L.n_rec * sizeof(RmRecord) + (size_t)L.spill_depth * 256u * sizeof(float);
        if (shmem > 64u * 1024u) return fail(c, RM_ERR_TOO_LARGE, "program needs %zu bytes of LDS per workgroup", shmem);
        hipLaunchKernelGGL(rmk::rm_render_pixel, grid, dim3(256), shmem, s, L);
    }
    if (int rc = time_end(c, s)) return rc;
    return finish_launch(c, s);
}

int ensure_tile_buffers(rm_ctx* c, size_t n_tiles_total) {
    size_t cap = c->d_tiles_cap;
    if (int rc = grow_device(c, &c->d_cost, &cap, n_tiles_total)) return rc;
    if (int rc = grow_device(c, &c->d_order, &c->d_tiles_cap, n_tiles_total)) return rc;
    return RM_OK;
}
int ensure_stats(rm_ctx* c, RmLaunch& L, size_t n_waves) {
    if (!c->wave_stats) return RM_OK;
    size_t cap = c->d_stats_bytes / sizeof(unsigned long long);  // in u64 elements
    if (int rc = grow_device(c, &c->d_stats, &cap, n_waves * 4u)) return rc;
    c->d_stats_bytes = cap * sizeof(unsigned long long);
    c->stats_valid_bytes = n_waves * 4u * sizeof(unsigned long long);
    L.stats = c->d_stats;
    return RM_OK;
}

int time_begin(rm_ctx* c, hipStream_t s) {
    if (!c->timing) return RM_OK;
    if (c->tev_used == c->tev.size()) {
        if (c->tev.size() >= 4096) return RM_OK;  // stop recording, keep running
        hipEvent_t a, b;
        HIP_TRY(c, hipEventCreate(&a));
        HIP_TRY(c, hipEventCreate(&b));
        c->tev.emplace_back(a, b);
    }
    HIP_TRY(c, hipEventRecord(c->tev[c->tev_used].first, s));
    return RM_OK;
}
int time_end(rm_ctx* c, hipStream_t s) {
    if (!c->timing || c->tev_used >= c->tev.size()) return RM_OK;
    HIP_TRY(c, hipEventRecord(c->tev[c->tev_used].second, s));
    c->tev_used++;
    return RM_OK;
}

int finish_launch(rm_ctx* c, hipStream_t s) {
    HIP_TRY(c, hipGetLastError());
    return RM_OK;
}

int check_dims(rm_ctx* c, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows) {
    if (W == 0 || H == 0 || W > kMaxDim || H > kMaxDim) return fail(c, RM_ERR_RANGE, "image size %ux%u out of range", W, H);
    if (rows == 0 || row0 >= H || rows > H - row0)
        return fail(c, RM_ERR_RANGE, "row band [%u,+%u) outside image height %u", row0, rows, H);
    return RM_OK;
}

// A draw with an astronomically large max_iter would never finish (an empty scene marches all
// max_iter steps, wgsl:189-191 + :109); refuse it instead of hanging the GPU.
int check_limits(rm_ctx* c) {
    if (c->limits.max_iter > kMaxIter)
        return fail(c, RM_ERR_RANGE, "max_iter %u exceeds the supported maximum %u", c->limits.max_iter, kMaxIter);
    return RM_OK;
}

hipStream_t user_stream(const rm_ctx* c, void* stream) {
    return stream == RM_STREAM_OWN ? c->stream : static_cast<hipStream_t>(stream);
}

// See rm_ctx::last_stream.  Nothing is issued while consecutive draws stay on one stream (the steady state, and the
// only state during stream capture).  A previous stream that no longer exists, or that is still being captured (its
// draws have not run and cannot race), makes the record fail: there is nothing to wait for then.
void order_with_previous(rm_ctx* c, hipStream_t s) {
    if (c->last_stream_valid && c->last_stream != s) {
        // a stream that is being captured cannot wait for work outside its graph (the attempt would invalidate the capture):
        // capture a draw on the stream the context drew on last
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &capturing) != hipSuccess) (void)hipGetLastError();
        if (capturing != hipStreamCaptureStatusNone) {
            c->last_stream = s;
            return;
        }
        // The PREVIOUS stream may be the one that is capturing (capture begun on A, a draw captured there, and now -- before
        // the capture ends -- a draw on another stream B): an event recorded on A would become a node of A's graph, and B's
        // wait on it would pull B into that capture (or fail with a capture-isolation error, invalidating it).  Captured work
        // has not run and cannot race with this draw: nothing to wait for.
        hipStreamCaptureStatus prev_capturing = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->last_stream, &prev_capturing) != hipSuccess) {
            (void)hipGetLastError();  // the previous stream no longer exists
            prev_capturing = hipStreamCaptureStatusActive;
        }
        if (prev_capturing == hipStreamCaptureStatusNone) {
            if (hipEventRecord(c->ev_order, c->last_stream) == hipSuccess) {
                if (hipStreamWaitEvent(s, c->ev_order, 0) != hipSuccess) (void)hipGetLastError();
            } else {
                (void)hipGetLastError();
            }
        }
    }
    c->last_stream = s;
    c->last_stream_valid = true;
}

size_t pixel_bytes(const rm_ctx* c) { return c->out_format == RM_FORMAT_RGBA32F ? 16u : 4u; }

int ensure_out(rm_ctx* c, size_t bytes) {
    if (bytes <= c->d_out_bytes) return RM_OK;
    if (c->d_out) (void)hipFree(c->d_out);
    c->d_out = nullptr;
    c->d_out_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->d_out, bytes));
    c->d_out_bytes = bytes;
    return RM_OK;
}

}  // namespace

RM_EXPORT int rm_abi_version(void) { return RM_ABI_VERSION; }

RM_EXPORT int rm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

RM_EXPORT int rm_create(int device, rm_ctx** out) {
    if (!out) return fail(nullptr, RM_ERR_NULL, "rm_create: out is NULL");
    *out = nullptr;
    int n = rm_device_count();
    if (n <= 0) return fail(nullptr, RM_ERR_NO_DEVICE, "rm_create: no HIP device is visible");
    if (device < 0 || device >= n) return fail(nullptr, RM_ERR_ARG, "rm_create: device %d not in [0,%d)", device, n);
    rm_ctx* c = new (std::nothrow) rm_ctx();
    if (!c) return fail(nullptr, RM_ERR_DEVICE, "rm_create: out of host memory");
    c->device = device;
    c->cmd.assign(kRefCmdBufferBytes / 4, 0u);
    hipError_t e = hipSetDevice(device);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
    if (e == hipSuccess) {
        c->cu_count = prop.multiProcessorCount;
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreate(&c->ev0);
    if (e == hipSuccess) e = hipEventCreate(&c->ev1);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming);
    if (e != hipSuccess) {
        int rc = fail(nullptr, RM_ERR_DEVICE, "rm_create: HIP initialisation failed: %s", hipGetErrorString(e));
        rm_destroy(c);
        return rc;
    }
    *out = c;
    return RM_OK;
}

RM_EXPORT void rm_destroy(rm_ctx* c) {
    if (!c) return;
    if (c->device >= 0) (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_prog) (void)hipFree(c->d_prog);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_frames) (void)hipFree(c->d_frames);
    if (c->d_stats) (void)hipFree(c->d_stats);
    if (c->d_cost) (void)hipFree(c->d_cost);
    if (c->d_order) (void)hipFree(c->d_order);
    if (c->d_counters) (void)hipFree(c->d_counters);
    if (c->d_measured) (void)hipFree(c->d_measured);
    if (c->d_bounds) (void)hipFree(c->d_bounds);
    if (c->d_mprog) (void)hipFree(c->d_mprog);
    if (c->d_materials) (void)hipFree(c->d_materials);
    for (auto& st : c->staging) {
        if (st.host) (void)hipHostFree(st.host);
        if (st.done) (void)hipEventDestroy(st.done);
    }
    for (auto& e : c->tev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

RM_EXPORT int rm_write_buffer(rm_ctx* c, int buffer, uint64_t offset, const void* data, uint64_t size) {
    if (!c) return RM_ERR_NULL;
    if (!data && size) return fail(c, RM_ERR_NULL, "rm_write_buffer: data is NULL");
    if ((offset & 3u) || (size & 3u))
        return fail(c, RM_ERR_ARG, "rm_write_buffer: offset %llu / size %llu not multiples of 4",
                    (unsigned long long)offset, (unsigned long long)size);
    uint8_t* dst = nullptr;
    uint64_t cap = 0;
    switch (buffer) {
    case RM_BUF_LIMITS: dst = reinterpret_cast<uint8_t*>(&c->limits); cap = sizeof(rm_limits); break;
    case RM_BUF_COMMANDS: dst = reinterpret_cast<uint8_t*>(c->cmd.data()); cap = c->cmd.size() * 4; break;
    case RM_BUF_UNIFORMS: dst = reinterpret_cast<uint8_t*>(&c->uniforms); cap = sizeof(rm_uniforms); break;
    default: return fail(c, RM_ERR_ARG, "rm_write_buffer: unknown buffer %d", buffer);
    }
    if (offset > cap || size > cap - offset)
        return fail(c, RM_ERR_TOO_LARGE, "rm_write_buffer: [%llu,+%llu) exceeds the %llu-byte buffer %d",
                    (unsigned long long)offset, (unsigned long long)size, (unsigned long long)cap, buffer);
    if (buffer == RM_BUF_COMMANDS && size && std::memcmp(dst + offset, data, size) != 0) c->cmd_dirty = true;
    if (size) std::memcpy(dst + offset, data, size);
    return RM_OK;
}

RM_EXPORT int rm_set_uniforms(rm_ctx* c, const rm_uniforms* u) {
    if (!c) return RM_ERR_NULL;
    if (!u) return fail(c, RM_ERR_NULL, "rm_set_uniforms: u is NULL");
    return rm_write_buffer(c, RM_BUF_UNIFORMS, 0, u, sizeof *u);
}

RM_EXPORT int rm_set_limits(rm_ctx* c, const rm_limits* l) {
    if (!c) return RM_ERR_NULL;
    if (!l) return fail(c, RM_ERR_NULL, "rm_set_limits: l is NULL");
    return rm_write_buffer(c, RM_BUF_LIMITS, 0, l, sizeof *l);
}

RM_EXPORT int rm_set_program(rm_ctx* c, uint32_t cmd_count, const uint32_t* words, uint32_t n_words) {
    if (!c) return RM_ERR_NULL;
    if (n_words && !words) return fail(c, RM_ERR_NULL, "rm_set_program: words is NULL");
    const uint64_t cap_words = c->cmd.size() - 1;
    if (n_words > cap_words)
        return fail(c, RM_ERR_TOO_LARGE, "rm_set_program: %u words do not fit the %llu-byte command buffer "
                    "(rm_resize_command_buffer lifts the reference's 1024-byte limit)",
                    n_words, (unsigned long long)(c->cmd.size() * 4));
    // The reference rebuilds and rewrites the command buffer every frame (renderer.rs:224-239); a rewrite that leaves
    // the buffer as it is keeps the decoded program and its device copy.
    if (!c->cmd_dirty && c->cmd_status == RM_OK && c->cmd[0] == cmd_count &&
        (n_words == 0u || std::memcmp(c->cmd.data() + 1, words, (size_t)n_words * 4) == 0))
        return RM_OK;
    RmDecoded d;
    int rc = rm_decode_program(cmd_count, words, n_words, &d);
    if (rc != RM_OK) return fail(c, rc, "rm_set_program: %s", rm_status_string(rc));
    c->cmd[0] = cmd_count;                                               // renderer.rs:230-234
    if (n_words) std::memcpy(c->cmd.data() + 1, words, (size_t)n_words * 4);  // renderer.rs:235-239
    c->cmd_dirty = true;
    return RM_OK;
}

RM_EXPORT int rm_set_materials(rm_ctx* c, uint32_t count, const float* rgb) {
    if (!c) return RM_ERR_NULL;
    if (!rgb) return fail(c, RM_ERR_NULL, "rm_set_materials: rgb is NULL");
    if (count < 1u || count > RM_MAX_MATERIALS)
        return fail(c, RM_ERR_MATERIAL, "rm_set_materials: %u entries, the table holds 1 to %u", count, (unsigned)RM_MAX_MATERIALS);
    c->materials.resize(count);
    for (uint32_t i = 0; i < count; i++) c->materials[i] = make_float4(rgb[3u * i], rgb[3u * i + 1u], rgb[3u * i + 2u], 0.0f);
    c->materials_dirty = true;
    return RM_OK;
}

RM_EXPORT int rm_resize_command_buffer(rm_ctx* c, uint64_t bytes) {
    if (!c) return RM_ERR_NULL;
    if (bytes < kRefCmdBufferBytes || bytes > kMaxCmdBufferBytes || (bytes & 3u))
        return fail(c, RM_ERR_ARG, "rm_resize_command_buffer: %llu not a multiple of 4 in [1024, 65536]",
                    (unsigned long long)bytes);
    c->cmd.resize(bytes / 4, 0u);
    c->cmd_dirty = true;
    return RM_OK;
}

RM_EXPORT int rm_validate(rm_ctx* c) {
    if (!c) return RM_ERR_NULL;
    RmDecoded d;
    int rc = rm_decode_program(c->cmd[0], c->cmd.data() + 1, (uint32_t)c->cmd.size() - 1u, &d);
    if (rc != RM_OK) return fail(c, rc, "invalid CSG program in command buffer: %s", rm_status_string(rc));
    return RM_OK;
}

RM_EXPORT int rm_validate_program(uint32_t cmd_count, const uint32_t* words, uint32_t n_words,
                                  uint32_t* out_max_depth) {
    RmDecoded d;
    int rc = rm_decode_program(cmd_count, words, n_words, &d);
    if (rc == RM_OK && out_max_depth) *out_max_depth = d.max_depth;
    return rc;
}

RM_EXPORT int rm_program_info(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, uint32_t* out, uint32_t n_out) {
    RmDecoded d;
    int rc = rm_decode_program(cmd_count, words, n_words, &d);
    if (rc != RM_OK) return rc;
    uint32_t subtracted = 0u;
    for (const RmRecord& r : d.rec) subtracted += (r.op & RM_OP_NOCULL) != 0u;
    const uint32_t facts[RM_PROGRAM_FACTS] = {(uint32_t)d.rec.size(), d.n_sphere, d.n_box, subtracted, (uint32_t)d.units.size(), d.spill_depth,
                                              d.is_chain ? 1u : 0u, d.prunable ? 1u : 0u, d.bound_walk ? 1u : 0u, d.has_xforms ? 1u : 0u,
                                              d.n_leaves, (uint32_t)prune_kind(d, 2)};
    for (uint32_t i = 0; out && i < n_out && i < (uint32_t)RM_PROGRAM_FACTS; i++) out[i] = facts[i];
    return RM_OK;
}

RM_EXPORT int rm_draw(rm_ctx* c, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows, float* out_rgba,
                      int out_is_device, void* stream) {
    if (!c) return RM_ERR_NULL;
    if (!out_rgba) return fail(c, RM_ERR_NULL, "rm_draw: out_rgba is NULL");
    int rc = check_dims(c, W, H, row0, rows);
    if (rc != RM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = out_is_device ? user_stream(c, stream) : c->stream;
    order_with_previous(c, s);
    rc = ensure_program(c, s);
    if (rc == RM_OK) rc = ensure_materials(c, s);
    if (rc != RM_OK) return rc;
    rc = check_limits(c);
    if (rc != RM_OK) return rc;
    const size_t bytes = (size_t)rows * W * pixel_bytes(c);
    if (out_is_device) return launch(c, nullptr, 1, W, H, row0, rows, out_rgba, s);
    rc = ensure_out(c, bytes);
    if (rc != RM_OK) return rc;
    rc = launch(c, nullptr, 1, W, H, row0, rows, c->d_out, s);
    if (rc != RM_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_rgba, c->d_out, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return RM_OK;
}

// Output rows of the strips first, first+stride, ... (strip_rows rows each) of an H-row image.
static uint32_t strip_row_count(uint32_t H, uint32_t strip_rows, uint32_t first, uint32_t stride) {
    const uint32_t n_strips = (H + strip_rows - 1u) / strip_rows;
    uint32_t rows = 0;
    for (uint32_t sidx = first; sidx < n_strips; sidx += stride) {
        const uint32_t r0 = sidx * strip_rows;
        rows += H - r0 < strip_rows ? H - r0 : strip_rows;
    }
    return rows;
}

RM_EXPORT int rm_draw_strips(rm_ctx* c, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t first, uint32_t stride,
                             float* out_rgba, int out_is_device, void* stream, uint32_t* out_rows) {
    if (!c) return RM_ERR_NULL;
    if (!out_rows) return fail(c, RM_ERR_NULL, "rm_draw_strips: out_rows is NULL");
    int rc = check_dims(c, W, H, 0, H);
    if (rc != RM_OK) return rc;
    if (strip_rows == 0u || (strip_rows % 8u) != 0u || stride == 0u || first >= stride)
        return fail(c, RM_ERR_ARG, "rm_draw_strips: strip_rows %u must be a positive multiple of 8, first %u < stride %u",
                    strip_rows, first, stride);
    const uint32_t rows = strip_row_count(H, strip_rows, first, stride);
    *out_rows = rows;
    if (rows == 0u) return RM_OK;  // more ranks than strips: nothing to do for this one
    if (!out_rgba) return fail(c, RM_ERR_NULL, "rm_draw_strips: out_rgba is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    order_with_previous(c, out_is_device ? user_stream(c, stream) : c->stream);
    rc = ensure_program(c, out_is_device ? user_stream(c, stream) : c->stream);
    if (rc == RM_OK) rc = ensure_materials(c, out_is_device ? user_stream(c, stream) : c->stream);
    if (rc != RM_OK) return rc;
    rc = check_limits(c);
    if (rc != RM_OK) return rc;
    StripSpec sp;
    sp.rows = strip_rows; sp.first = first; sp.stride = stride;
    const size_t bytes = (size_t)rows * W * pixel_bytes(c);
    if (out_is_device) return launch(c, nullptr, 1, W, H, 0, rows, out_rgba, user_stream(c, stream), sp);
    rc = ensure_out(c, bytes);
    if (rc != RM_OK) return rc;
    rc = launch(c, nullptr, 1, W, H, 0, rows, c->d_out, c->stream, sp);
    if (rc != RM_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_rgba, c->d_out, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RM_OK;
}

// The final host-side gather of a tiled frame (north-star): this GPU's strips go from the compact device
// buffer rm_draw_strips filled to their rows of the full H-row host image, one D2H copy per strip on `stream`.
RM_EXPORT int rm_gather_strips(rm_ctx* c, uint32_t W, uint32_t H, uint32_t strip_rows, uint32_t first, uint32_t stride,
                               const void* strips_device, void* host_image, void* stream) {
    if (!c) return RM_ERR_NULL;
    int rc = check_dims(c, W, H, 0, H);
    if (rc != RM_OK) return rc;
    if (strip_rows == 0u || (strip_rows % 8u) != 0u || stride == 0u || first >= stride)
        return fail(c, RM_ERR_ARG, "rm_gather_strips: strip_rows %u must be a positive multiple of 8, first %u < stride %u",
                    strip_rows, first, stride);
    const uint32_t n_strips = (H + strip_rows - 1u) / strip_rows;
    if (first >= n_strips) return RM_OK;  // this GPU has no strip
    if (!strips_device || !host_image) return fail(c, RM_ERR_NULL, "rm_gather_strips: NULL buffer");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = user_stream(c, stream);
    const size_t row_bytes = (size_t)W * pixel_bytes(c);
    const uint8_t* src = static_cast<const uint8_t*>(strips_device);
    uint8_t* dst = static_cast<uint8_t*>(host_image);
    if (stride == 1u) {  // one GPU: the compact buffer IS the frame
        HIP_TRY(c, hipMemcpyAsync(dst, src, row_bytes * H, hipMemcpyDeviceToHost, s));
        return RM_OK;
    }
    // This GPU's strips are equally spaced in the frame: all the full ones go in ONE pitched copy (a row of the copy = one
    // strip, destination pitch = `stride` strips), a ragged last strip in a second one.  RM_GATHER_PER_STRIP=1: one copy per
    // strip (A/B).
    static const bool per_strip = std::getenv("RM_GATHER_PER_STRIP") != nullptr;
    const size_t strip_bytes = row_bytes * strip_rows;
    uint32_t n_mine = 0, n_full = 0;
    for (uint32_t sidx = first; sidx < n_strips; sidx += stride) {
        n_mine++;
        if ((sidx + 1u) * strip_rows <= H) n_full++;
    }
    if (!per_strip && n_full >= 2u) {
        HIP_TRY(c, hipMemcpy2DAsync(dst + strip_bytes * first, strip_bytes * stride, src, strip_bytes, strip_bytes, n_full,
                                    hipMemcpyDeviceToHost, s));
        if (n_mine > n_full) {  // the frame's last strip is this GPU's and is shorter
            const uint32_t sidx = first + n_full * stride, r0 = sidx * strip_rows;
            HIP_TRY(c, hipMemcpyAsync(dst + row_bytes * r0, src + strip_bytes * n_full, row_bytes * (H - r0), hipMemcpyDeviceToHost, s));
        }
        return RM_OK;
    }
    for (uint32_t sidx = first; sidx < n_strips; sidx += stride) {
        const uint32_t r0 = sidx * strip_rows, rows = H - r0 < strip_rows ? H - r0 : strip_rows;
        HIP_TRY(c, hipMemcpyAsync(dst + row_bytes * r0, src, row_bytes * rows, hipMemcpyDeviceToHost, s));
        src += row_bytes * rows;
    }
    return RM_OK;
}

RM_EXPORT int rm_host_register(void* ptr, uint64_t bytes) {
    if (!ptr || !bytes) return RM_ERR_NULL;
    if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return RM_ERR_DEVICE;
    }
    return RM_OK;
}

RM_EXPORT int rm_host_unregister(void* ptr) {
    if (!ptr) return RM_ERR_NULL;
    if (hipHostUnregister(ptr) != hipSuccess) {
        (void)hipGetLastError();
        return RM_ERR_DEVICE;
    }
    return RM_OK;
}

RM_EXPORT int rm_draw_batch(rm_ctx* c, const rm_uniforms* frames, uint32_t n_frames, uint32_t W, uint32_t H,
                            float* out_rgba, int out_is_device, void* stream) {
    if (!c) return RM_ERR_NULL;
    if (!out_rgba || !frames) return fail(c, RM_ERR_NULL, "rm_draw_batch: NULL argument");
    if (n_frames == 0 || n_frames > 65535u) return fail(c, RM_ERR_RANGE, "rm_draw_batch: n_frames %u not in [1,65535]", n_frames);
    int rc = check_dims(c, W, H, 0, H);
    if (rc != RM_OK) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = out_is_device ? user_stream(c, stream) : c->stream;
    order_with_previous(c, s);
    rc = ensure_program(c, s);
    if (rc == RM_OK) rc = ensure_materials(c, s);
    if (rc != RM_OK) return rc;
    rc = check_limits(c);
    if (rc != RM_OK) return rc;
    if (n_frames > c->d_frames_cap) {
        if (c->d_frames) (void)hipFree(c->d_frames);
        c->d_frames = nullptr;
        c->d_frames_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_frames, (size_t)n_frames * sizeof(rm_uniforms)));
        c->d_frames_cap = n_frames;
    }
    rc = upload(c, c->d_frames, frames, (size_t)n_frames * sizeof(rm_uniforms), s);
    if (rc != RM_OK) return rc;
    const size_t bytes = (size_t)n_frames * H * W * pixel_bytes(c);
    if (out_is_device) return launch(c, c->d_frames, n_frames, W, H, 0, H, out_rgba, s);
    rc = ensure_out(c, bytes);
    if (rc != RM_OK) return rc;
    rc = launch(c, c->d_frames, n_frames, W, H, 0, H, c->d_out, s);
    if (rc != RM_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(out_rgba, c->d_out, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return RM_OK;
}

RM_EXPORT int rm_sync_context(rm_ctx* c) {
    if (!c) return RM_ERR_NULL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RM_OK;
}

RM_EXPORT int rm_sync(rm_ctx* c) {
    if (!c) return RM_ERR_NULL;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    return RM_OK;
}

RM_EXPORT int rm_set_option(rm_ctx* c, int key, int64_t value) {
    if (!c) return RM_ERR_NULL;
    switch (key) {
    case RM_OPT_KERNEL:
        if (value != RM_KERNEL_DEFAULT && value != RM_KERNEL_PIXEL && value != RM_KERNEL_V5 && value != RM_KERNEL_V5_LDS)
            return fail(c, RM_ERR_ARG, "unknown kernel %lld (the v2-v4 variants 2..11 of ABI version 1 are retired)", (long long)value);
        c->kernel = (int)value;
        return RM_OK;
    case RM_OPT_TIMING: c->timing = value != 0; c->tev_used = 0; return RM_OK;
    case RM_OPT_STRICT_CAP: return RM_OK;
    case RM_OPT_CULL: c->cull = value != 0; return RM_OK;
    case RM_OPT_BALANCE: c->balance = value < 0 ? 0 : value > 3 ? 3 : (int)value; c->measured_shape = 0; return RM_OK;
    case RM_OPT_WAVE_STATS: c->wave_stats = value != 0; return RM_OK;
    case RM_OPT_PRUNE:
        if (value < 0 || value > 2) return fail(c, RM_ERR_ARG, "RM_OPT_PRUNE: %lld is not 0, 1 or 2", (long long)value);
        c->prune = (int)value;
        c->spec_gen = ~0ull;
        return RM_OK;
    case RM_OPT_OUTPUT_FORMAT:
        if (value < RM_FORMAT_RGBA32F || value > RM_FORMAT_BGRA8_UNORM) return fail(c, RM_ERR_ARG, "unknown output format %lld", (long long)value);
        c->out_format = (int)value;
        return RM_OK;
    case RM_OPT_SPECIALIZE:
        if (value < 0 || value > 2) return fail(c, RM_ERR_ARG, "RM_OPT_SPECIALIZE: %lld is not 0, 1 or 2", (long long)value);
        c->specialize = (int)value;
        return RM_OK;
    case RM_OPT_WAVES_PER_TILE:
        if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(c, RM_ERR_ARG, "waves_per_tile must be 0 (automatic), 1, 2, 4 or 8");
        c->waves_per_tile = (int)value;
        return RM_OK;
    case RM_OPT_REFILL_MIN:
        if (value < 0 || value > 64) return fail(c, RM_ERR_ARG, "refill_min %lld not in [0,64]", (long long)value);
        c->refill_min_v5 = (uint32_t)value;
        return RM_OK;
    default: return fail(c, RM_ERR_ARG, "unknown option %d", key);
    }
}

RM_EXPORT int rm_get_info(rm_ctx* c, int key, double* out) {
    if (!c) return RM_ERR_NULL;
    if (!out) return fail(c, RM_ERR_NULL, "rm_get_info: out is NULL");
    switch (key) {
    case RM_INFO_KERNEL_MS: {  // mean duration of the dominant kernel over the launches timed since the last query
        double sum = 0.0;
        for (size_t i = 0; i < c->tev_used; i++) {
            HIP_TRY(c, hipEventSynchronize(c->tev[i].second));
            float ms = 0.f;
            HIP_TRY(c, hipEventElapsedTime(&ms, c->tev[i].first, c->tev[i].second));
            sum += ms;
        }
        if (c->tev_used) c->last_kernel_ms = sum / (double)c->tev_used;
        c->tev_used = 0;
        *out = c->last_kernel_ms;
        return RM_OK;
    }
    case RM_INFO_DEVICE: *out = c->device; return RM_OK;
    case RM_INFO_CU_COUNT: *out = c->cu_count; return RM_OK;
    case RM_INFO_SPECIALIZED: *out = c->last_specialized ? 1.0 : 0.0; return RM_OK;
    case RM_INFO_INTERPRETER_LOOP: *out = c->last_specialized ? 0.0 : (double)c->last_loop; return RM_OK;
    case RM_INFO_PRUNED: *out = c->spec && c->spec_gen == c->prog_gen && !c->cmd_dirty ? (double)c->spec_pruned : 0.0; return RM_OK;
    case RM_INFO_JIT_STATE:
    case RM_INFO_JIT_FROM_CACHE:
    case RM_INFO_JIT_COMPILE_MS: {
        *out = 0.0;
        if (c->spec && c->spec_gen == c->prog_gen && !c->cmd_dirty) {
            std::lock_guard<std::mutex> lk(c->spec->m);
            if (key == RM_INFO_JIT_STATE) *out = 1.0 + (double)c->spec->state;
            else if (key == RM_INFO_JIT_FROM_CACHE) *out = c->spec->from_cache ? 1.0 : 0.0;
            else *out = c->spec->compile_ms;
        }
        return RM_OK;
    }
    case RM_INFO_PROGRAM_COMMANDS:
    case RM_INFO_PROGRAM_WORDS:
    case RM_INFO_PROGRAM_DEPTH: {
        RmDecoded d;
        int rc = rm_decode_program(c->cmd[0], c->cmd.data() + 1, (uint32_t)c->cmd.size() - 1u, &d);
        if (rc != RM_OK) return fail(c, rc, "invalid CSG program in command buffer: %s", rm_status_string(rc));
        *out = key == RM_INFO_PROGRAM_COMMANDS ? c->cmd[0] : key == RM_INFO_PROGRAM_WORDS ? d.n_words : d.max_depth;
        return RM_OK;
    }
    default: return fail(c, RM_ERR_ARG, "unknown info key %d", key);
    }
}

RM_EXPORT int rm_read_wave_stats(rm_ctx* c, void* dst, uint64_t cap_bytes, uint64_t* out_bytes) {
    if (!c) return RM_ERR_NULL;
    if (!dst || !out_bytes) return fail(c, RM_ERR_NULL, "rm_read_wave_stats: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    const uint64_t n = c->stats_valid_bytes < cap_bytes ? c->stats_valid_bytes : cap_bytes;
    if (n) HIP_TRY(c, hipMemcpy(dst, c->d_stats, n, hipMemcpyDeviceToHost));
    *out_bytes = n;
    return RM_OK;
}

RM_EXPORT int rm_selftest_sqrt(rm_ctx* c, uint64_t* out_mismatches, uint32_t* out_first_bad_bits) {
    if (!c) return RM_ERR_NULL;
    if (!out_mismatches || !out_first_bad_bits) return fail(c, RM_ERR_NULL, "rm_selftest_sqrt: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long* d_bad = nullptr;
    uint32_t* d_first = nullptr;
    HIP_TRY(c, hipMalloc(&d_bad, 8));
    HIP_TRY(c, hipMalloc(&d_first, 4));
    HIP_TRY(c, hipMemset(d_bad, 0, 8));
    HIP_TRY(c, hipMemset(d_first, 0xFF, 4));
    hipLaunchKernelGGL(rmk::rm_selftest_sqrt_kernel, dim3(std::max(1, c->cu_count) * 16), dim3(256), 0, c->stream, 0u,
                       (uint64_t)1 << 32, d_bad, d_first);
    hipError_t e = hipStreamSynchronize(c->stream);
    unsigned long long bad = 0;
    if (e == hipSuccess) e = hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(out_first_bad_bits, d_first, 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    (void)hipFree(d_first);
    if (e != hipSuccess) return fail(c, RM_ERR_DEVICE, "rm_selftest_sqrt: %s", hipGetErrorString(e));
    *out_mismatches = bad;
    return RM_OK;
}

RM_EXPORT int rm_selftest_ops(rm_ctx* c, const float* a, const float* b, float* out, uint32_t n) {
    if (!c) return RM_ERR_NULL;
    if (!a || !b || !out || n == 0u) return fail(c, RM_ERR_NULL, "rm_selftest_ops: NULL argument");
    HIP_TRY(c, hipSetDevice(c->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    HIP_TRY(c, hipMalloc(&da, (size_t)n * 4));
    HIP_TRY(c, hipMalloc(&db, (size_t)n * 4));
    HIP_TRY(c, hipMalloc(&dout, (size_t)n * 32));
    hipError_t e = hipMemcpy(da, a, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, b, (size_t)n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rmk::rm_selftest_ops_kernel, dim3((n + 255u) / 256u), dim3(256), 0, c->stream, da, db, dout, n);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * 32, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, RM_ERR_DEVICE, "rm_selftest_ops: %s", hipGetErrorString(e));
    return RM_OK;
}

RM_EXPORT int rm_selftest_wave(rm_ctx* c, const float* in, uint32_t n_waves, float* out) {
    if (!c) return RM_ERR_NULL;
    if (!in || !out || n_waves == 0u || n_waves > 65535u) return fail(c, RM_ERR_NULL, "rm_selftest_wave: bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n = (size_t)n_waves * 64u;
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(c, hipMalloc(&din, n * 4));
    HIP_TRY(c, hipMalloc(&dout, n * 8));
    hipError_t e = hipMemcpy(din, in, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rmk::rm_selftest_wave_kernel, dim3(n_waves), dim3(64), 0, c->stream, din, dout, n_waves);
        e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(din); (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, RM_ERR_DEVICE, "rm_selftest_wave: %s", hipGetErrorString(e));
    return RM_OK;
}

RM_EXPORT int rm_measure_write_bandwidth(rm_ctx* c, uint64_t bytes, int iters, double* out_gbps) {
    if (!c) return RM_ERR_NULL;
    if (!out_gbps) return fail(c, RM_ERR_NULL, "rm_measure_write_bandwidth: out is NULL");
    if (bytes < 4096 || (bytes & 15u) || iters < 1 || iters > 1000) return fail(c, RM_ERR_ARG, "bad bytes/iters");
    HIP_TRY(c, hipSetDevice(c->device));
    float4* buf = nullptr;
    HIP_TRY(c, hipMalloc(&buf, bytes));
    const size_t n_vec = bytes / 16u;
    const int grid = std::max(1, c->cu_count) * 8;
    hipLaunchKernelGGL(rmk::rm_fill, dim3(grid), dim3(256), 0, c->stream, buf, n_vec, 0.0f);  // warm-up
    hipError_t e = hipEventRecord(c->ev0, c->stream);
    for (int i = 0; i < iters && e == hipSuccess; i++)
        hipLaunchKernelGGL(rmk::rm_fill, dim3(grid), dim3(256), 0, c->stream, buf, n_vec, (float)i);
    if (e == hipSuccess) e = hipEventRecord(c->ev1, c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(c->ev1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->ev0, c->ev1);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(c, RM_ERR_DEVICE, "write-bandwidth calibration failed: %s", hipGetErrorString(e));
    *out_gbps = (double)bytes * iters / (ms * 1e-3) / 1e9;
    return RM_OK;
}

namespace {
int jit_decode(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, int wpt, bool prune, std::string* src, std::string* capped = nullptr) {
    if (wpt != 1 && wpt != 2 && wpt != 4 && wpt != 8) return RM_ERR_ARG;
    RmDecoded d;
    int rc = rm_decode_program(cmd_count, words, n_words, &d);
    if (rc != RM_OK) return rc;
    const int kind = !prune ? rmjit::PRUNE_NONE : d.unit_mode == RM_UNITS_LATTICE ? rmjit::PRUNE_LATTICE : d.unit_mode == RM_UNITS_BLEND ? rmjit::PRUNE_BLEND : rmjit::PRUNE_NONE;
    if (!rmjit::can_specialise(d.rec) || !rmjit::generate_source(d.rec, d.mrec, wpt, kind, src, nullptr, nullptr, capped)) return RM_ERR_ARG;
    return RM_OK;
}
void copy_out(const std::string& s, char* buf, size_t cap) {
    if (!buf || !cap) return;
    const size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
}
}  // namespace

RM_EXPORT int rm_jit_source(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, int waves_per_tile, char* buf,
                            size_t cap, size_t* needed) {
    std::string src;
    int rc = jit_decode(cmd_count, words, n_words, waves_per_tile & 0xFF, (waves_per_tile & RM_JIT_PRUNE) != 0, &src);
    if (rc != RM_OK) return rc;
    if (needed) *needed = src.size() + 1;
    copy_out(src, buf, cap);
    return RM_OK;
}

RM_EXPORT int rm_jit_compile(uint32_t cmd_count, const uint32_t* words, uint32_t n_words, int waves_per_tile,
                             double* compile_ms, size_t* code_bytes, char* log, size_t log_cap) {
    std::string src, capped, msg;
    int rc = jit_decode(cmd_count, words, n_words, waves_per_tile & 0xFF, (waves_per_tile & RM_JIT_PRUNE) != 0, &src, &capped);
    if (rc != RM_OK) return rc;
    std::vector<char> code;
    double ms = 0.0;
    const bool ok = rmjit::compile_best(src, capped, &code, &msg, &ms);  // (what a draw would get: the disk cache is warmed through here)
    if (compile_ms) *compile_ms = ms;
    if (code_bytes) *code_bytes = code.size();
    copy_out(msg, log, log_cap);
    return ok ? RM_OK : RM_ERR_DEVICE;
}

RM_EXPORT int rm_jit_log(rm_ctx* c, char* buf, size_t cap) {
    if (!c) return RM_ERR_NULL;
    std::string msg;
    if (c->spec) {
        std::lock_guard<std::mutex> lk(c->spec->m);
        msg = c->spec->log;
    }
    copy_out(msg, buf, cap);
    return RM_OK;
}

RM_EXPORT const char* rm_last_error(rm_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

RM_EXPORT const char* rm_status_string(int status) {
    switch (status) {
    case RM_OK: return "ok";
    case RM_ERR_NULL: return "null pointer";
    case RM_ERR_TRUNCATED: return "command reads past the end of the command buffer";
    case RM_ERR_STACK_UNDERFLOW: return "binary operator with fewer than two operands on the value stack";
    case RM_ERR_STACK_OVERFLOW: return "value stack deeper than 32";
    case RM_ERR_EMPTY_RESULT: return "program leaves no value on the stack";
    case RM_ERR_OPCODE: return "unknown opcode";
    case RM_ERR_TOO_LARGE: return "write exceeds buffer size";
    case RM_ERR_RANGE: return "image size or row band out of range";
    case RM_ERR_DEVICE: return "HIP runtime error";
    case RM_ERR_NO_DEVICE: return "no GPU available";
    case RM_ERR_ARG: return "invalid argument";
    case RM_ERR_TRANSFORM: return "transform push/pop commands are not properly nested around one value";
    case RM_ERR_MATERIAL: return "material index outside the material table";
    default: return "unknown status";
    }
}
