// rm_jit.h -- structure specialisation of the march kernel with hipRTC.  Host only.
//
// The reference evaluates the scene with a stack machine that re-decodes the same command
// sequence at every march step of every ray (ray_marching.wgsl:187-203: a loop over cmd_count with
// a switch per command).  A scene's STRUCTURE (which node types, in which order) changes only when
// the application edits the CSG tree (csg/builder.rs:26-62 is re-run); its PARAMETERS (centres,
// radii) may change every frame.  This file turns the structure into straight-line device code:
// one call per leaf with its parameters read from the LDS copy of the decoded program at constant
// offsets, operands held in registers instead of the LDS spill stack, no opcode decode, no loop.
// The arithmetic applied to every value is the interpreter's, operation for operation (the same
// sdf_*_t / vmin / vmax_negb functions are called), so the result is bit-identical -- only the
// bookkeeping around it disappears.  Parameters are NOT baked in: an animation that moves
// primitives keeps its compiled kernel.
//
// The generated translation unit includes the library's own kernel headers (embedded as strings by
// build.py -> generated/rm_jit_sources.inc), defines rmk::map_scene_spec<FAST>() and instantiates
// rm_render_v5_body<..., SPEC = true> behind an extern "C" kernel.  It is compiled for gfx950 by
// libhiprtc (dlopen'ed: the library has no link-time dependency on it) on a worker thread and
// cached per (structure, waves per tile).  Everything falls back to the interpreter kernel:
// hipRTC missing, a compile error, an empty or very long program.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <unistd.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rm_device.h"
#include "rm_groups.h"

namespace rmjit {

#include "generated/rm_jit_sources.inc"  // kHeaderNames[], kHeaderSources[], kNumHeaders

constexpr uint32_t kMaxRecords = 255u;  // the reference's 1024-byte command buffer holds at most 255 words

// ---- hipRTC, loaded on first use -----------------------------------------------------------------
struct Rtc {
    typedef struct _hiprtcProgram* Program;
    int (*CreateProgram)(Program*, const char*, const char*, int, const char* const*, const char* const*) = nullptr;
    int (*CompileProgram)(Program, int, const char* const*) = nullptr;
    int (*GetProgramLogSize)(Program, size_t*) = nullptr;
    int (*GetProgramLog)(Program, char*) = nullptr;
    int (*GetCodeSize)(Program, size_t*) = nullptr;
    int (*GetCode)(Program, char*) = nullptr;
    int (*DestroyProgram)(Program*) = nullptr;
    int (*Version)(int*, int*) = nullptr;
    void* handle = nullptr;
    std::string error;
    std::string identity;  // which compiler this is: library file + version (part of the disk-cache key)

    static Rtc& get() {
        static Rtc r;
        static std::once_flag once;
        std::call_once(once, [] { r.load(); });
        return r;
    }
    bool ok() const { return handle != nullptr; }

private:
    void load() {
        // By SONAME first: a process that already holds a hipRTC (PyTorch bundles one next to its own
        // libamdhip64 and comgr) keeps using that one; otherwise the ROCm installation on this library's
        // RUNPATH.  Mixing the run-time compiler of one ROCm release with the comgr of another is avoided.
        const char* names[] = {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"};
        if (const char* env = std::getenv("RM_HIPRTC_SO")) handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
        for (size_t i = 0; !handle && i < sizeof names / sizeof *names; i++) handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!handle) {
            const char* e = dlerror();
            error = std::string("libhiprtc.so could not be loaded: ") + (e ? e : "?");
            return;
        }
        bool all = true;
        auto sym = [&](const char* n) {
            void* p = dlsym(handle, n);
            if (!p) { all = false; error = std::string("libhiprtc.so lacks ") + n; }
            return p;
        };
        CreateProgram = reinterpret_cast<decltype(CreateProgram)>(sym("hiprtcCreateProgram"));
        CompileProgram = reinterpret_cast<decltype(CompileProgram)>(sym("hiprtcCompileProgram"));
        GetProgramLogSize = reinterpret_cast<decltype(GetProgramLogSize)>(sym("hiprtcGetProgramLogSize"));
        GetProgramLog = reinterpret_cast<decltype(GetProgramLog)>(sym("hiprtcGetProgramLog"));
        GetCodeSize = reinterpret_cast<decltype(GetCodeSize)>(sym("hiprtcGetCodeSize"));
        GetCode = reinterpret_cast<decltype(GetCode)>(sym("hiprtcGetCode"));
        DestroyProgram = reinterpret_cast<decltype(DestroyProgram)>(sym("hiprtcDestroyProgram"));
        if (!all) { handle = nullptr; return; }  // the library stays mapped; it is simply not used
        Version = reinterpret_cast<decltype(Version)>(dlsym(handle, "hiprtcVersion"));
        int major = 0, minor = 0;
        if (Version) (void)Version(&major, &minor);
        Dl_info info;
        std::memset(&info, 0, sizeof info);
        (void)dladdr(reinterpret_cast<void*>(CreateProgram), &info);
        identity = std::string(info.dli_fname ? info.dli_fname : "?") + ":" + std::to_string(major) + "." + std::to_string(minor);
    }
};

// ---- source generation ---------------------------------------------------------------------------
// The structure of a decoded program: one character pair per record.  Two programs with the same
// key run the same generated code (their parameters differ, and those stay in LDS).
inline std::string structure_key(const std::vector<RmRecord>& rec) {
    std::string k;
    k.reserve(rec.size() * 2);
    for (const RmRecord& r : rec) {
        k.push_back("PSBCLXM?"[RM_OP_KIND(r.op)]);
        k.push_back("pusixy??"[RM_OP_MODE(r.op)]);
        // a SmoothUnion whose k is NaN or infinite takes its leaf out of the local skipping rule, which changes how leaves pair
        // up (rm_groups.h): part of the structure
        if (RM_OP_KIND(r.op) == RM_KIND_POP && RM_OP_MODE(r.op) == RM_MODE_SMOOTH && !(r.p[0] == r.p[0] && std::fabs(r.p[0]) < 1.0e30f)) k.push_back('!');
    }
    return k;
}

inline bool can_specialise(const std::vector<RmRecord>& rec) { return !rec.empty() && rec.size() <= kMaxRecords; }

// Straight-line map_scene for `rec`, mirroring exec_command (rm_interp.h) record by record
// with the value stack resolved at generation time: the accumulator and every spilled value become
// named values.  With `prune`, sphere and box leaves are wrapped in the wave-uniform far test of
// rm_kernel_v5.h ("Pruning").  Returns false if the records do not form a valid program (cannot
// happen for the output of rm_decode_program).
// A/B knobs of the generated code (environment, read when a structure is generated; defaults are the measured best)
inline int jit_knob(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

inline bool generate_map_scene(const std::vector<RmRecord>& rec, bool prune, std::string* out) {
    // diagnostics (tools/wave_stats.py): 1 counts evaluated leaves, 2 leaf tests executed (leaves of near groups), 3 near groups,
    // 4 (group, lane) pairs that are near
    const int count_mode = prune ? jit_knob("RM_JIT_PRUNE_STATS", 0) : 0;
    const bool count = count_mode == 1;
    std::string s;
    char line[512];
    s += "namespace rmk {\n";
    s += "template <bool FAST>\n";
    s += "RM_DEV float map_scene_spec(LdsF lp, float qx, float qy, float qz, float thr, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval) {\n";
    if (prune) {
        s += "    const float thrk = thr * 1.000005f;\n";          // sphere test: ((thr + r) k)^2
        s += "    const float thr2k = (thr * thr) * 1.00001f;\n";  // box test
        s += "    const float inf = __uint_as_float(0x7F800000u);\n";
    }
    // A scheduling barrier after every 4 leaves: left alone the compiler hoists the parameter loads of the whole
    // program to the top of the straight-line code (88 VGPRs for 16 leaves, 120-139 for 32: 3 waves per SIMD);
    // with the barriers 61-62 VGPRs whatever the length.  64-node scene at 4K 636 -> 682 Mpx/s, metric frame +2 %.
    // (RM_JIT_SCHED_BARRIER=N overrides, 0 disables.)
    const int sched_every = std::getenv("RM_JIT_SCHED_BARRIER") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER")) : 4;
    int leaves = 0;
    // Grouped far tests: consecutive sphere / box leaves pair up (RmDecoded::groups: pair g = pruned leaves 2g, 2g + 1;
    // its bounding sphere is record n_rec + g of the LDS copy); one test clears both members.
    int n_pruned_total = 0, n_pruned = 0;
    // Members of a near pair: a box keeps a far test of its own behind the pair's (one compare on a value it computes
    // anyway, saves the square root and the inside term), a sphere does not (its test is three more vector instructions
    // and a branch to save five).  Measured (RM_JIT_LEAF_TESTS: 1 both / 0 neither / 2 boxes = default / 3 spheres), march
    // kernel of the metric frame 0.780 / 0.780 / 0.761 / 0.784 ms, G64 at 4K 7.38 / 7.46 / 7.31 / 7.47 ms.
    const int leaf_tests = jit_knob("RM_JIT_LEAF_TESTS", 2);
    const bool sub_tests = jit_knob("RM_JIT_SUB_TESTS", 1) != 0;  // A/B: the local test of subtracted leaves (below)
    for (const RmRecord& r : rec) n_pruned_total += prune && (RM_OP_KIND(r.op) == RM_KIND_SPHERE || RM_OP_KIND(r.op) == RM_KIND_BOX);
    std::vector<int> stack;  // value numbers; back() is the accumulator
    std::vector<int> pos;    // position numbers of the open transform scopes; back() is the current one (0 = qx, qy, qz)
    int nv = 0, np = 0;
    s += "    const float x0 = qx, y0 = qy, z0 = qz;\n";
    pos.push_back(0);
    // all group tests up front: their LDS reads go out together instead of one stalling in front of every pair
    for (int g = 0; 2 * g + 1 < n_pruned_total; g++) {
        const unsigned goff = (unsigned)(rec.size() + (size_t)g) * 8u;
        std::snprintf(line, sizeof line, "    const bool g%d = spec_group_near(live, lp + %u, x0, y0, z0, thrk);\n", g, goff);
        s += line;
        if (count_mode == 3) { std::snprintf(line, sizeof line, "    if (g%d) n_eval += 1u;\n", g); s += line; }
        if (count_mode == 4) {  // lanes for which the group is near (what a perfectly coherent wave would pay for)
            std::snprintf(line, sizeof line, "    n_eval += spec_group_near_lanes(live, lp + %u, x0, y0, z0, thrk);\n", goff);
            s += line;
        }
    }
    for (size_t i = 0; i < rec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        const unsigned off = (unsigned)i * 8u;  // first parameter of record i, in dwords (records are staged rotated: lds_load4)
        if (kind == RM_KIND_XFORM) {  // space transformation: a new position value (push) or back to the enclosing one (pop)
            const int c = pos.back();
            if ((mode & 1u) == 0u) {
                const int n = ++np;
                if (mode == RM_XF_T_PUSH)
                    std::snprintf(line, sizeof line, "    const float x%d = x%d - lp[%u], y%d = y%d - lp[%u], z%d = z%d - lp[%u];\n",
                                  n, c, off, n, c, off + 1u, n, c, off + 2u);
                else if (mode == RM_XF_R_PUSH)
                    std::snprintf(line, sizeof line, "    float x%d = x%d, y%d = y%d, z%d = z%d; xf_rotate_conj(lp[%u], lp[%u], lp[%u], lp[%u], x%d, y%d, z%d);\n",
                                  n, c, n, c, n, c, off, off + 1u, off + 2u, off + 3u, n, n, n);
                else
                    std::snprintf(line, sizeof line, "    const float x%d = x%d / lp[%u], y%d = y%d / lp[%u], z%d = z%d / lp[%u];\n",
                                  n, c, off, n, c, off, n, c, off);
                s += line;
                pos.push_back(n);
            } else {
                if (pos.size() < 2 || stack.empty()) return false;
                pos.pop_back();
                if (mode == RM_XF_S_POP) {
                    const int a = stack.back(); stack.pop_back();
                    const int w = nv++;
                    std::snprintf(line, sizeof line, "    const float v%d = v%d * lp[%u];\n", w, a, off);
                    s += line;
                    stack.push_back(w);
                }
            }
            continue;
        }
        char P[64];  // "xN, yN, zN": the position this record's leaf is evaluated at
        std::snprintf(P, sizeof P, "x%d, y%d, z%d", pos.back(), pos.back(), pos.back());
        const char* op = mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr;
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
            const int b = stack.back(); stack.pop_back();
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            if (mode == RM_MODE_SMOOTH) std::snprintf(line, sizeof line, "    const float v%d = spec_smooth_union(lp + %u, v%d, v%d, live);\n", w, off, a, b);
            else if (op) std::snprintf(line, sizeof line, "    const float v%d = %s(v%d, v%d);\n", w, op, a, b);
            else return false;
            s += line;
            stack.push_back(w);
            continue;
        }
        if (mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;  // the decoder never fuses an operator with a parameter
        int a = -1;
        if (mode != RM_MODE_PUSH) {
            if (stack.empty()) return false;
            a = stack.back(); stack.pop_back();
        }
        const int w = nv++;  // the record's result: the leaf (PUSH) or op(acc, leaf)
        const bool pruned = prune && (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX);
        if (pruned) {
            // result if the leaf is far for every live lane: +inf (PUSH), acc (UNION: min(acc, +inf); SUB: max(acc, -inf));
            // Intersection max(acc, +inf) = +inf
            if (mode == RM_MODE_PUSH || mode == RM_MODE_INTER) std::snprintf(line, sizeof line, "    float v%d = inf;\n", w);
            else std::snprintf(line, sizeof line, "    float v%d = v%d;\n", w, a);
            s += line;
            char leaf[128];
            const int ordinal = n_pruned++, grp = ordinal / 2;
            if ((ordinal | 1) < n_pruned_total) {  // this leaf has a partner
                std::snprintf(line, sizeof line, "    if (g%d)\n", grp);
                s += line;
            }
            const char* tested = count_mode == 2 ? "n_eval += 1u; " : "";
            const bool paired = (ordinal | 1) < n_pruned_total;
            bool own_test = !paired || leaf_tests == 1 || (leaf_tests == 2 && kind == RM_KIND_BOX) || (leaf_tests == 3 && kind == RM_KIND_SPHERE);
            // A SUBTRACTED leaf changes max(acc, -v) only where -v > acc: at a position outside it (v > 0) with acc >= 0 -- a ray
            // that is not inside anything -- never.  That is a test on values at hand (the squared distance, the accumulator),
            // sharper than the threshold test it replaces wherever acc >= 0, and exact without any Lipschitz argument.
            const bool local_sub = mode == RM_MODE_SUB && sub_tests;
            if (local_sub) {
                own_test = true;
                if (kind == RM_KIND_SPHERE) {
                    std::snprintf(line, sizeof line, "    { %sconst float a = spec_sphere_a(lp + %u, %s);\n      if (spec_sub_sphere_near(live, lp + %u, a, v%d)) ", tested, off, P, off, a);
                    std::snprintf(leaf, sizeof leaf, "spec_sphere_v<FAST>(lp + %u, a, tiny)", off);
                } else {
                    std::snprintf(line, sizeof line, "    { %sconst SpecBox b = spec_box_a(lp + %u, %s);\n      if (spec_sub_box_near(live, b.a, v%d)) ", tested, off, P, a);
                    std::snprintf(leaf, sizeof leaf, "spec_box_v<FAST>(b, tiny)");
                }
            } else if (!own_test) {  // a member of a near pair is evaluated without a test of its own
                std::snprintf(leaf, sizeof leaf, "%s<FAST>(lp + %u, %s, tiny)", kind == RM_KIND_SPHERE ? "spec_sphere" : "spec_box", off, P);
                std::snprintf(line, sizeof line, "    %s", tested);
            } else if (kind == RM_KIND_SPHERE) {
                std::snprintf(line, sizeof line, "    { %sconst float a = spec_sphere_a(lp + %u, %s);\n      if (spec_any_near(live, spec_sphere_far(lp + %u, a, thrk))) ", tested, off, P, off);
                std::snprintf(leaf, sizeof leaf, "spec_sphere_v<FAST>(lp + %u, a, tiny)", off);
            } else {
                std::snprintf(line, sizeof line, "    { %sconst SpecBox b = spec_box_a(lp + %u, %s);\n      if (spec_any_near(live, b.a > thr2k)) ", tested, off, P);
                std::snprintf(leaf, sizeof leaf, "spec_box_v<FAST>(b, tiny)");
            }
            s += line;
            const char* close = own_test ? " }" : "";
            if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "{ v%d = %s; %s}%s\n", w, leaf, count ? "n_eval += 1u; " : "", close);
            else std::snprintf(line, sizeof line, "{ v%d = %s(v%d, %s); %s}%s\n", w, op, a, leaf, count ? "n_eval += 1u; " : "", close);
            s += line;
        } else {
            const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>"
                           : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>" : kind == RM_KIND_PLANE ? "spec_plane" : nullptr;
            if (!fn) return false;
            char leaf[128];
            if (kind == RM_KIND_PLANE) std::snprintf(leaf, sizeof leaf, "%s(lp + %u, %s)", fn, off, P);
            else std::snprintf(leaf, sizeof leaf, "%s(lp + %u, %s, tiny)", fn, off, P);
            if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "    const float v%d = %s;\n", w, leaf);
            else std::snprintf(line, sizeof line, "    const float v%d = %s(v%d, %s);\n", w, op, a, leaf);
            s += line;
        }
        stack.push_back(w);
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    }
    if (stack.empty()) return false;
    std::snprintf(line, sizeof line, "    return v%d;\n}\n}  // namespace rmk\n", stack.back());
    s += line;
    *out = std::move(s);
    return true;
}

// The four normal taps of a hit (wgsl:135-144) in one pass over the program: every record is applied to the four
// positions c + k_t eps with the leaf functions and operators map_scene_spec uses, so each f[t] goes through the same
// operations as a separate evaluation would -- but the compiler sees that the twelve coordinates take only six
// different values, and one loop iteration replaces four.  With `prune` (bounded 1-Lipschitz leaves, min / max
// operators), the far test of a sphere / box runs ONCE, at the hit position c, against a threshold the caller widened
// by the tap offset eps sqrt(3): a leaf that passes it is far from all four taps and is skipped for all of them.
// Returns false when the program contains a SmoothUnion (see below) or the records do not form a valid program: the
// kernel then taps one position at a time through map_scene_spec.
inline bool generate_map_scene_taps(const std::vector<RmRecord>& rec, bool prune, std::string* out) {
    std::string s;
    char line[768];
    s += "namespace rmk {\n";
    s += "template <bool FAST>\n";
    s += "RM_DEV void map_scene_taps(LdsF lp, float cx, float cy, float cz, float thr, unsigned long long live, SqrtGuard& tiny, float (&f)[4]) {\n";
    if (prune) {
        s += "    const float thrk = thr * 1.000005f;\n";
        s += "    const float thr2k = (thr * thr) * 1.00001f;\n";
        s += "    const float inf = __uint_as_float(0x7F800000u);\n";
    }
    const bool fence = jit_knob("RM_JIT_GUARD_FENCE", 1) != 0;  // A/B: see guard_fence (rm_kernel_v5.h)
    s += "    const float e = 0.0001f;\n";  // wgsl:136; k = (1,-1): taps (+,-,-), (-,-,+), (-,+,-), (+,+,+) (wgsl:138-141)
    s += "    const float x0_0 = cx + e, x0_1 = cx - e, x0_2 = cx - e, x0_3 = cx + e;\n";
    s += "    const float y0_0 = cy - e, y0_1 = cy - e, y0_2 = cy + e, y0_3 = cy + e;\n";
    s += "    const float z0_0 = cz - e, z0_1 = cz + e, z0_2 = cz - e, z0_3 = cz + e;\n";
    if (prune) {  // group tests up front, as in generate_map_scene
        int total = 0;
        for (const RmRecord& r : rec) total += RM_OP_KIND(r.op) == RM_KIND_SPHERE || RM_OP_KIND(r.op) == RM_KIND_BOX;
        for (int g = 0; 2 * g + 1 < total; g++) {
            const unsigned goff = (unsigned)(rec.size() + (size_t)g) * 8u;
            std::snprintf(line, sizeof line, "    const bool g%d = spec_group_near(live, lp + %u, cx, cy, cz, thrk);\n", g, goff);
            s += line;
        }
    }
    const int sched_every = std::getenv("RM_JIT_SCHED_BARRIER_TAPS") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER_TAPS")) : 2;
    int leaves = 0, nv = 0, np = 0;
    int n_pruned_total = 0, n_pruned = 0;  // grouped far tests, as in generate_map_scene
    for (const RmRecord& r : rec) n_pruned_total += prune && (RM_OP_KIND(r.op) == RM_KIND_SPHERE || RM_OP_KIND(r.op) == RM_KIND_BOX);
    std::vector<int> stack;
    std::vector<int> pos;  // open transform scopes, as in generate_map_scene
    pos.push_back(0);
    for (size_t i = 0; i < rec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        const unsigned off = (unsigned)i * 8u;
        if (kind == RM_KIND_XFORM) {
            if (prune) return false;  // pruned programs have no transforms
            const int c = pos.back();
            if ((mode & 1u) == 0u) {
                const int n = ++np;
                for (int t = 0; t < 4; t++) {
                    if (mode == RM_XF_T_PUSH)
                        std::snprintf(line, sizeof line, "    const float x%d_%d = x%d_%d - lp[%u], y%d_%d = y%d_%d - lp[%u], z%d_%d = z%d_%d - lp[%u];\n",
                                      n, t, c, t, off, n, t, c, t, off + 1u, n, t, c, t, off + 2u);
                    else if (mode == RM_XF_R_PUSH)
                        std::snprintf(line, sizeof line, "    float x%d_%d = x%d_%d, y%d_%d = y%d_%d, z%d_%d = z%d_%d; xf_rotate_conj(lp[%u], lp[%u], lp[%u], lp[%u], x%d_%d, y%d_%d, z%d_%d);\n",
                                      n, t, c, t, n, t, c, t, n, t, c, t, off, off + 1u, off + 2u, off + 3u, n, t, n, t, n, t);
                    else
                        std::snprintf(line, sizeof line, "    const float x%d_%d = x%d_%d / lp[%u], y%d_%d = y%d_%d / lp[%u], z%d_%d = z%d_%d / lp[%u];\n",
                                      n, t, c, t, off, n, t, c, t, off, n, t, c, t, off);
                    s += line;
                }
                pos.push_back(n);
            } else {
                if (pos.size() < 2 || stack.empty()) return false;
                pos.pop_back();
                if (mode == RM_XF_S_POP) {
                    const int a = stack.back(); stack.pop_back();
                    const int w = nv++;
                    for (int t = 0; t < 4; t++) {
                        std::snprintf(line, sizeof line, "    const float v%d_%d = v%d_%d * lp[%u];\n", w, t, a, t, off);
                        s += line;
                    }
                    stack.push_back(w);
                }
            }
            continue;
        }
        const char* op = mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr;
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
            const int b = stack.back(); stack.pop_back();
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            if (mode == RM_MODE_SMOOTH) {
                // SmoothUnion: left to itself the compiler interleaves the four copies of the correctly rounded division
                // (~130 VGPRs, 3 waves per SIMD: 40 % slower than tapping one position at a time); spec_smooth_union4
                // tests the blend zone once for the four taps and keeps one division in flight.
                // RM_JIT_TAPS4_SMOOTH=0 restores the one-position taps.
                const char* knob = std::getenv("RM_JIT_TAPS4_SMOOTH");
                if (knob && std::atoi(knob) == 0) return false;
                std::snprintf(line, sizeof line,
                              "    float v%d_0, v%d_1, v%d_2, v%d_3;\n"
                              "    { const float sa[4] = {v%d_0, v%d_1, v%d_2, v%d_3}, sb[4] = {v%d_0, v%d_1, v%d_2, v%d_3}; float so[4];\n"
                              "      spec_smooth_union4(lp + %u, sa, sb, live, so); v%d_0 = so[0]; v%d_1 = so[1]; v%d_2 = so[2]; v%d_3 = so[3]; }\n",
                              w, w, w, w, a, a, a, a, b, b, b, b, off, w, w, w, w);
                s += line;
                stack.push_back(w);
                continue;
            }
            if (!op) return false;
            for (int t = 0; t < 4; t++) {
                std::snprintf(line, sizeof line, "    const float v%d_%d = %s(v%d_%d, v%d_%d);\n", w, t, op, a, t, b, t);
                s += line;
            }
            stack.push_back(w);
            continue;
        }
        const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>"
                       : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>" : kind == RM_KIND_PLANE ? "spec_plane" : nullptr;
        if (!fn || mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;
        if (prune && kind == RM_KIND_PLANE) return false;
        int a = -1;
        if (mode != RM_MODE_PUSH) {
            if (stack.empty()) return false;
            a = stack.back(); stack.pop_back();
        }
        const int w = nv++;
        const int c = pos.back();
        const bool pruned = prune && (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX);
        if (pruned) {
            for (int t = 0; t < 4; t++) {  // the value if the leaf is far: see generate_map_scene
                if (mode == RM_MODE_PUSH || mode == RM_MODE_INTER) std::snprintf(line, sizeof line, "    float v%d_%d = inf;\n", w, t);
                else std::snprintf(line, sizeof line, "    float v%d_%d = v%d_%d;\n", w, t, a, t);
                s += line;
            }
            const int ordinal = n_pruned++, grp = ordinal / 2;
            const char* guard = "";
            char gname[32];
            if ((ordinal | 1) < n_pruned_total) {
                std::snprintf(gname, sizeof gname, "g%d && ", grp);
                guard = gname;
            }
            if (kind == RM_KIND_SPHERE)
                std::snprintf(line, sizeof line, "    if (%sspec_any_near(live, spec_sphere_far(lp + %u, spec_sphere_a(lp + %u, cx, cy, cz), thrk))) {\n", guard, off, off);
            else
                std::snprintf(line, sizeof line, "    if (%sspec_any_near(live, spec_box_a(lp + %u, cx, cy, cz).a > thr2k)) {\n", guard, off);
            s += line;
        }
        for (int t = 0; t < 4; t++) {
            char leaf[192];
            if (kind == RM_KIND_PLANE) std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x%d_%d, y%d_%d, z%d_%d)", fn, off, c, t, c, t, c, t);
            else std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x%d_%d, y%d_%d, z%d_%d, tiny)", fn, off, c, t, c, t, c, t);
            const char* decl = pruned ? "    " : "const float ";
            if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "    %sv%d_%d = %s;\n", decl, w, t, leaf);
            else std::snprintf(line, sizeof line, "    %sv%d_%d = %s(v%d_%d, %s);\n", decl, w, t, op, a, t, leaf);
            s += line;
        }
        if (pruned) s += "    }\n";
        if (fence && kind != RM_KIND_PLANE) s += "    guard_fence(tiny);\n";  // see guard_fence (rm_kernel_v5.h)
        stack.push_back(w);
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    }
    if (stack.empty()) return false;
    for (int t = 0; t < 4; t++) {
        std::snprintf(line, sizeof line, "    f[%d] = v%d_%d;\n", t, stack.back(), t);
        s += line;
    }
    s += "}\n}  // namespace rmk\n";
    *out = std::move(s);
    return true;
}


// ---- programs that blend: the LOCAL skipping rule (rm_groups.h, rm_kernel_v5.h spec_local_near) -----------------------
// map_scene_spec for a program with SmoothUnion operators.  Everything is evaluated as generate_map_scene(prune = false)
// would, except the leaves the local rule applies to (right operand of a Union, or of the SmoothUnion that follows): each
// such leaf -- or a PAIR of them, blended into the same accumulator one after the other -- sits behind the wave-uniform test
// "is its lower bound less than k above the accumulator for any live lane"; when it is not, leaf and operator are skipped
// and the result is the accumulator, the very bits the evaluation would have produced.  Subtracted leaves keep the local
// test of the lattice form (spec_sub_*_near), which needs no threshold either.  `thr` carries only the float margin m.
inline bool generate_map_scene_blend(const std::vector<RmRecord>& rec, std::string* out) {
    const int count_mode = jit_knob("RM_JIT_PRUNE_STATS", 0);  // 1 leaves evaluated, 3 near pairs
    const char* counted = count_mode == 1 ? "n_eval += 1u; " : "";
    // members of a near pair: 0 untested (default; measured, config 3 at 4K: 4.86 ms against 5.01 with a test each: a box's test
    // needs 14 of the box's 26 vector instructions first), 1 a test each, 2 boxes only, 3 spheres only; leaves without a partner
    // always have one
    const int leaf_tests = jit_knob("RM_JIT_BLEND_LEAF_TESTS", 0);
    const bool upfront = jit_knob("RM_JIT_BLEND_UPFRONT", 1) != 0;  // the pairs' squared distances at the top (their LDS reads go out together)
    const bool sub_tests = jit_knob("RM_JIT_SUB_TESTS", 1) != 0;
    const int sched_every = std::getenv("RM_JIT_SCHED_BARRIER") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER")) : 4;
    const std::vector<std::pair<int, int>> pairs = rm_blend_pairs(rec);
    std::map<int, int> pair_of_first;
    for (size_t g = 0; g < pairs.size(); g++) pair_of_first[pairs[g].first] = (int)g;
    std::string s;
    char line[768];
    s += "namespace rmk {\n";
    s += "template <bool FAST>\n";
    s += "RM_DEV float map_scene_spec(LdsF lp, float qx, float qy, float qz, float thr, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval) {\n";
    s += "    const float m = thr;\n";  // the float margin of the local rule; a blend kernel's body passes no threshold
    s += "    const float x0 = qx, y0 = qy, z0 = qz;\n";
    auto pair_distance = [&](int g) {
        std::snprintf(line, sizeof line, "    float kr%d; const float pa%d = spec_pair_a(lp + %u, x0, y0, z0, kr%d);\n", g, g,
                      (unsigned)(rec.size() + (size_t)g) * 8u, g);
        s += line;
    };
    if (upfront)
        for (size_t g = 0; g < pairs.size(); g++) pair_distance((int)g);
    std::vector<int> stack;
    int nv = 0, leaves = 0;
    auto barrier = [&]() {
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    };
    // one local leaf and its operator: v<w> (declared by the caller, = v<a>) becomes op(v<a>, leaf) unless the leaf is skipped
    auto emit_member = [&](size_t i, const RmLeafUse& use, int a, int w, bool test) {
        const uint32_t kind = RM_OP_KIND(rec[i].op);
        const unsigned off = (unsigned)i * 8u, koff = use.k_rec >= 0 ? (unsigned)use.k_rec * 8u : 0u;
        char kexpr[64], open_apply[96];
        if (use.k_rec >= 0) {
            std::snprintf(kexpr, sizeof kexpr, "spec_local_k(lp + %u)", koff);
            std::snprintf(open_apply, sizeof open_apply, "spec_smooth_union(lp + %u, v%d, ", koff, a);
        } else {
            std::snprintf(kexpr, sizeof kexpr, "0.0f");
            std::snprintf(open_apply, sizeof open_apply, "vmin(v%d, ", a);
        }
        const char* close_apply = use.k_rec >= 0 ? ", live)" : ")";
        if (kind == RM_KIND_SPHERE) {
            if (test)
                std::snprintf(line, sizeof line,
                              "    { const float a = spec_sphere_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_local_near(live, a, ((v%d + m) + %s) + lp[%u])) { v%d = %sspec_sphere_v<FAST>(lp + %u, a, tiny)%s; %s} }\n",
                              off, a, kexpr, off + 3u, w, open_apply, off, close_apply, counted);
            else
                std::snprintf(line, sizeof line, "    { v%d = %sspec_sphere<FAST>(lp + %u, x0, y0, z0, tiny)%s; %s}\n", w, open_apply, off, close_apply, counted);
        } else {
            if (test)
                std::snprintf(line, sizeof line,
                              "    { const SpecBox b = spec_box_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_local_box_near(live, b.a, (v%d + m) + %s)) { v%d = %sspec_box_v<FAST>(b, tiny)%s; %s} }\n",
                              off, a, kexpr, w, open_apply, close_apply, counted);
            else
                std::snprintf(line, sizeof line, "    { v%d = %sspec_box<FAST>(lp + %u, x0, y0, z0, tiny)%s; %s}\n", w, open_apply, off, close_apply, counted);
        }
        s += line;
    };
    auto wants_test = [&](uint32_t kind) {
        return leaf_tests == 1 || (leaf_tests == 2 && kind == RM_KIND_BOX) || (leaf_tests == 3 && kind == RM_KIND_SPHERE);
    };
    for (size_t i = 0; i < rec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        const unsigned off = (unsigned)i * 8u;
        if (kind == RM_KIND_XFORM || kind == RM_KIND_MATERIAL) return false;  // the local rule is generated for world-space leaves only
        const char* op = mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr;
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
            const int b = stack.back(); stack.pop_back();
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            if (mode == RM_MODE_SMOOTH) std::snprintf(line, sizeof line, "    const float v%d = spec_smooth_union(lp + %u, v%d, v%d, live);\n", w, off, a, b);
            else if (op) std::snprintf(line, sizeof line, "    const float v%d = %s(v%d, v%d);\n", w, op, a, b);
            else return false;
            s += line;
            stack.push_back(w);
            continue;
        }
        const RmLeafUse use = rm_leaf_use(rec, i);
        if (use.local) {
            if (stack.empty()) return false;
            const int a = stack.back(); stack.pop_back();
            auto pf = pair_of_first.find((int)i);
            if (pf != pair_of_first.end()) {
                const int g = pf->second;
                const size_t j = (size_t)pairs[(size_t)g].second;
                const RmLeafUse use2 = rm_leaf_use(rec, j);
                const int w1 = nv++, w2 = nv++;
                std::snprintf(line, sizeof line, "    float v%d = v%d, v%d = v%d;\n", w1, a, w2, a);
                s += line;
                if (!upfront) pair_distance(g);
                std::snprintf(line, sizeof line, "    if (spec_local_near(live, pa%d, (v%d + m) + kr%d)) {\n%s", g, a, g, count_mode == 3 ? "    n_eval += 1u;\n" : "");
                s += line;
                emit_member(i, use, a, w1, wants_test(kind));
                std::snprintf(line, sizeof line, "    v%d = v%d;\n", w2, w1);
                s += line;
                emit_member(j, use2, w1, w2, wants_test(RM_OP_KIND(rec[j].op)));
                s += "    }\n";
                stack.push_back(w2);
                barrier();
                barrier();
                i = (size_t)use2.next - 1u;
            } else {
                const int w = nv++;
                std::snprintf(line, sizeof line, "    float v%d = v%d;\n", w, a);
                s += line;
                emit_member(i, use, a, w, true);
                stack.push_back(w);
                barrier();
                i = (size_t)use.next - 1u;
            }
            continue;
        }
        if (mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;
        int a = -1;
        if (mode != RM_MODE_PUSH) {
            if (stack.empty()) return false;
            a = stack.back(); stack.pop_back();
        }
        const int w = nv++;
        if (mode == RM_MODE_SUB && sub_tests && (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX)) {  // see generate_map_scene
            std::snprintf(line, sizeof line, "    float v%d = v%d;\n", w, a);
            s += line;
            if (kind == RM_KIND_SPHERE)
                std::snprintf(line, sizeof line,
                              "    { const float a = spec_sphere_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_sub_sphere_near(live, lp + %u, a, v%d)) { v%d = vmax_negb(v%d, spec_sphere_v<FAST>(lp + %u, a, tiny)); %s} }\n",
                              off, off, a, w, a, off, counted);
            else
                std::snprintf(line, sizeof line,
                              "    { const SpecBox b = spec_box_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_sub_box_near(live, b.a, v%d)) { v%d = vmax_negb(v%d, spec_box_v<FAST>(b, tiny)); %s} }\n",
                              off, a, w, a, counted);
            s += line;
        } else {
            const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>"
                           : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>" : kind == RM_KIND_PLANE ? "spec_plane" : nullptr;
            if (!fn) return false;
            char leaf[128];
            if (kind == RM_KIND_PLANE) std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x0, y0, z0)", fn, off);
            else std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x0, y0, z0, tiny)", fn, off);
            if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "    const float v%d = %s; %s\n", w, leaf, counted);
            else std::snprintf(line, sizeof line, "    const float v%d = %s(v%d, %s); %s\n", w, op, a, leaf, counted);
            s += line;
        }
        stack.push_back(w);
        barrier();
    }
    if (stack.empty()) return false;
    std::snprintf(line, sizeof line, "    return v%d;\n}\n}  // namespace rmk\n", stack.back());
    s += line;
    *out = std::move(s);
    return true;
}

// The four taps of a hit for a program that blends (see generate_map_scene_taps): the local rule is tested ONCE per leaf or
// pair, with the bound taken at the hit position c -- eps sqrt(3) from every tap: the caller's margin `thr` carries it --
// against each tap's own accumulator; a leaf is skipped only when it is far for all four taps of every live lane.
inline bool generate_map_scene_taps_blend(const std::vector<RmRecord>& rec, std::string* out) {
    const int leaf_tests = jit_knob("RM_JIT_BLEND_LEAF_TESTS", 0);
    const bool fence = jit_knob("RM_JIT_GUARD_FENCE", 1) != 0;
    const int sched_every = std::getenv("RM_JIT_SCHED_BARRIER_TAPS") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER_TAPS")) : 2;
    const char* knob = std::getenv("RM_JIT_TAPS4_SMOOTH");
    if (knob && std::atoi(knob) == 0) return false;
    const std::vector<std::pair<int, int>> pairs = rm_blend_pairs(rec);
    std::map<int, int> pair_of_first;
    for (size_t g = 0; g < pairs.size(); g++) pair_of_first[pairs[g].first] = (int)g;
    std::string s;
    char line[1024];
    s += "namespace rmk {\n";
    s += "template <bool FAST>\n";
    s += "RM_DEV void map_scene_taps(LdsF lp, float cx, float cy, float cz, float thr, unsigned long long live, SqrtGuard& tiny, float (&f)[4]) {\n";
    s += "    const float m = thr;\n";
    s += "    const float e = 0.0001f;\n";
    s += "    const float x0_0 = cx + e, x0_1 = cx - e, x0_2 = cx - e, x0_3 = cx + e;\n";
    s += "    const float y0_0 = cy - e, y0_1 = cy - e, y0_2 = cy + e, y0_3 = cy + e;\n";
    s += "    const float z0_0 = cz - e, z0_1 = cz + e, z0_2 = cz - e, z0_3 = cz + e;\n";
    for (size_t g = 0; g < pairs.size(); g++) {
        std::snprintf(line, sizeof line, "    float kr%d; const float pa%d = spec_pair_a(lp + %u, cx, cy, cz, kr%d);\n", (int)g, (int)g,
                      (unsigned)(rec.size() + g) * 8u, (int)g);
        s += line;
    }
    std::vector<int> stack;
    int nv = 0, leaves = 0;
    auto barrier = [&]() {
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    };
    auto apply4 = [&](const RmLeafUse& use, int a, int w, const char* leaf_fmt_fn, unsigned off) {
        // v<w>_t = op(v<a>_t, leaf(tap t)) for the four taps
        if (use.k_rec >= 0) {
            std::snprintf(line, sizeof line,
                          "      { const float sa[4] = {v%d_0, v%d_1, v%d_2, v%d_3};\n"
                          "        const float sb[4] = {%s(lp + %u, x0_0, y0_0, z0_0, tiny), %s(lp + %u, x0_1, y0_1, z0_1, tiny), %s(lp + %u, x0_2, y0_2, z0_2, tiny), %s(lp + %u, x0_3, y0_3, z0_3, tiny)};\n"
                          "        float so[4]; spec_smooth_union4(lp + %u, sa, sb, live, so); v%d_0 = so[0]; v%d_1 = so[1]; v%d_2 = so[2]; v%d_3 = so[3]; }\n",
                          a, a, a, a, leaf_fmt_fn, off, leaf_fmt_fn, off, leaf_fmt_fn, off, leaf_fmt_fn, off, (unsigned)use.k_rec * 8u, w, w, w, w);
            s += line;
        } else {
            for (int t = 0; t < 4; t++) {
                std::snprintf(line, sizeof line, "      v%d_%d = vmin(v%d_%d, %s(lp + %u, x0_%d, y0_%d, z0_%d, tiny));\n", w, t, a, t, leaf_fmt_fn, off, t, t, t);
                s += line;
            }
        }
    };
    auto emit_member = [&](size_t i, const RmLeafUse& use, int a, int w, bool test) {
        const uint32_t kind = RM_OP_KIND(rec[i].op);
        const unsigned off = (unsigned)i * 8u;
        char kexpr[64];
        if (use.k_rec >= 0) std::snprintf(kexpr, sizeof kexpr, "spec_local_k(lp + %u)", (unsigned)use.k_rec * 8u);
        else std::snprintf(kexpr, sizeof kexpr, "0.0f");
        const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : "spec_box<FAST>";
        if (test) {
            if (kind == RM_KIND_SPHERE)
                std::snprintf(line, sizeof line,
                              "    { const float kq = %s + lp[%u]; const float rh[4] = {(v%d_0 + m) + kq, (v%d_1 + m) + kq, (v%d_2 + m) + kq, (v%d_3 + m) + kq};\n"
                              "      if (spec_local_near4(live, spec_sphere_a(lp + %u, cx, cy, cz), rh, true)) {\n", kexpr, off + 3u, a, a, a, a, off);
            else
                std::snprintf(line, sizeof line,
                              "    { const float kq = %s; const float rh[4] = {(v%d_0 + m) + kq, (v%d_1 + m) + kq, (v%d_2 + m) + kq, (v%d_3 + m) + kq};\n"
                              "      if (spec_local_near4(live, spec_box_a(lp + %u, cx, cy, cz).a, rh, false)) {\n", kexpr, a, a, a, a, off);
            s += line;
            apply4(use, a, w, fn, off);
            s += "      } }\n";
        } else {
            s += "    {\n";
            apply4(use, a, w, fn, off);
            s += "    }\n";
        }
        if (fence) s += "    guard_fence(tiny);\n";
    };
    auto wants_test = [&](uint32_t kind) {
        return leaf_tests == 1 || (leaf_tests == 2 && kind == RM_KIND_BOX) || (leaf_tests == 3 && kind == RM_KIND_SPHERE);
    };
    for (size_t i = 0; i < rec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        const unsigned off = (unsigned)i * 8u;
        if (kind == RM_KIND_XFORM || kind == RM_KIND_MATERIAL) return false;
        const char* op = mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr;
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
            const int b = stack.back(); stack.pop_back();
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            if (mode == RM_MODE_SMOOTH) {
                std::snprintf(line, sizeof line,
                              "    float v%d_0, v%d_1, v%d_2, v%d_3;\n"
                              "    { const float sa[4] = {v%d_0, v%d_1, v%d_2, v%d_3}, sb[4] = {v%d_0, v%d_1, v%d_2, v%d_3}; float so[4];\n"
                              "      spec_smooth_union4(lp + %u, sa, sb, live, so); v%d_0 = so[0]; v%d_1 = so[1]; v%d_2 = so[2]; v%d_3 = so[3]; }\n",
                              w, w, w, w, a, a, a, a, b, b, b, b, off, w, w, w, w);
                s += line;
            } else {
                if (!op) return false;
                for (int t = 0; t < 4; t++) {
                    std::snprintf(line, sizeof line, "    const float v%d_%d = %s(v%d_%d, v%d_%d);\n", w, t, op, a, t, b, t);
                    s += line;
                }
            }
            stack.push_back(w);
            continue;
        }
        const RmLeafUse use = rm_leaf_use(rec, i);
        if (use.local) {
            if (stack.empty()) return false;
            const int a = stack.back(); stack.pop_back();
            auto pf = pair_of_first.find((int)i);
            if (pf != pair_of_first.end()) {
                const int g = pf->second;
                const size_t j = (size_t)pairs[(size_t)g].second;
                const RmLeafUse use2 = rm_leaf_use(rec, j);
                const int w1 = nv++, w2 = nv++;
                for (int t = 0; t < 4; t++) {
                    std::snprintf(line, sizeof line, "    float v%d_%d = v%d_%d, v%d_%d = v%d_%d;\n", w1, t, a, t, w2, t, a, t);
                    s += line;
                }
                std::snprintf(line, sizeof line,
                              "    { const float rh[4] = {(v%d_0 + m) + kr%d, (v%d_1 + m) + kr%d, (v%d_2 + m) + kr%d, (v%d_3 + m) + kr%d};\n"
                              "    if (spec_local_near4(live, pa%d, rh, true)) {\n", a, g, a, g, a, g, a, g, g);
                s += line;
                emit_member(i, use, a, w1, wants_test(kind));
                for (int t = 0; t < 4; t++) {
                    std::snprintf(line, sizeof line, "    v%d_%d = v%d_%d;\n", w2, t, w1, t);
                    s += line;
                }
                emit_member(j, use2, w1, w2, wants_test(RM_OP_KIND(rec[j].op)));
                s += "    } }\n";
                stack.push_back(w2);
                barrier();
                barrier();
                i = (size_t)use2.next - 1u;
            } else {
                const int w = nv++;
                for (int t = 0; t < 4; t++) {
                    std::snprintf(line, sizeof line, "    float v%d_%d = v%d_%d;\n", w, t, a, t);
                    s += line;
                }
                emit_member(i, use, a, w, true);
                stack.push_back(w);
                barrier();
                i = (size_t)use.next - 1u;
            }
            continue;
        }
        const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>"
                       : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>" : kind == RM_KIND_PLANE ? "spec_plane" : nullptr;
        if (!fn || mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;
        int a = -1;
        if (mode != RM_MODE_PUSH) {
            if (stack.empty()) return false;
            a = stack.back(); stack.pop_back();
        }
        const int w = nv++;
        for (int t = 0; t < 4; t++) {
            char leaf[192];
            if (kind == RM_KIND_PLANE) std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x0_%d, y0_%d, z0_%d)", fn, off, t, t, t);
            else std::snprintf(leaf, sizeof leaf, "%s(lp + %u, x0_%d, y0_%d, z0_%d, tiny)", fn, off, t, t, t);
            if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "    const float v%d_%d = %s;\n", w, t, leaf);
            else std::snprintf(line, sizeof line, "    const float v%d_%d = %s(v%d_%d, %s);\n", w, t, op, a, t, leaf);
            s += line;
        }
        if (fence && kind != RM_KIND_PLANE) s += "    guard_fence(tiny);\n";
        stack.push_back(w);
        barrier();
    }
    if (stack.empty()) return false;
    for (int t = 0; t < 4; t++) {
        std::snprintf(line, sizeof line, "    f[%d] = v%d_%d;\n", t, stack.back(), t);
        s += line;
    }
    s += "}\n}  // namespace rmk\n";
    *out = std::move(s);
    return true;
}


// ---- programs that blend, skip sets carried along the ray (rm_kernel_v5.h "Skip sets carried ALONG a ray") ---------------
// The top level of the program as a chain of UNITS, each starting and ending with one value (the accumulator) on the stack:
enum : int { BU_START = 0,    // record 0: the leaf that starts the chain
             BU_SINGLE = 1,   // a local leaf (rm_groups.h) and its Union / SmoothUnion
             BU_PAIR = 2,     // two of them, tested together
             BU_SUB = 3,      // a sphere / box fused with a Subtraction: the local test of the lattice form
             BU_GENERIC = 4 };// anything else that takes the accumulator to its next value: a fused Intersection / cylinder / plane,
                              // or a sub-tree (pushed, built, popped into the chain by its operator)
struct BlendUnit { int kind, first, last, second; };  // records [first, last]; second: the pair's second leaf record
inline bool blend_units(const std::vector<RmRecord>& rec, std::vector<BlendUnit>* out) {
    if (rec.empty() || RM_OP_MODE(rec[0].op) != RM_MODE_PUSH || (rec[0].op & RM_OP_SPILL)) return false;
    const uint32_t k0 = RM_OP_KIND(rec[0].op);
    if (k0 != RM_KIND_SPHERE && k0 != RM_KIND_BOX && k0 != RM_KIND_CYLINDER) return false;
    std::vector<BlendUnit> u;
    u.push_back({BU_START, 0, 0, -1});
    std::map<int, int> second_of;
    for (const std::pair<int, int>& pr : rm_blend_pairs(rec)) second_of[pr.first] = pr.second;
    for (size_t i = 1; i < rec.size();) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        if (kind == RM_KIND_XFORM || kind == RM_KIND_MATERIAL || kind == RM_KIND_PLANE || kind == RM_KIND_POP) return false;  // (a POP at depth 1 cannot be)
        const RmLeafUse use = rm_leaf_use(rec, i);
        if (use.local) {
            auto it = second_of.find((int)i);
            if (it != second_of.end()) {
                const RmLeafUse use2 = rm_leaf_use(rec, (size_t)it->second);
                u.push_back({BU_PAIR, (int)i, use2.next - 1, it->second});
                i = (size_t)use2.next;
            } else {
                u.push_back({BU_SINGLE, (int)i, use.next - 1, -1});
                i = (size_t)use.next;
            }
            continue;
        }
        if (mode != RM_MODE_PUSH) {  // a leaf fused with its operator
            u.push_back({mode == RM_MODE_SUB && (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX) ? BU_SUB : BU_GENERIC, (int)i, (int)i, -1});
            i++;
            continue;
        }
        // a pushed leaf that is not a local one: the start of a sub-tree; the unit ends with the operator that pops the chain's
        // accumulator back (stack depth, counted from the accumulator = 1, returns to 1)
        int depth = 1;
        size_t j = i;
        for (; j < rec.size(); j++) {
            const uint32_t kj = RM_OP_KIND(rec[j].op), mj = RM_OP_MODE(rec[j].op);
            if (kj == RM_KIND_XFORM || kj == RM_KIND_MATERIAL || kj == RM_KIND_PLANE) return false;
            if (kj == RM_KIND_POP) depth--;
            else if (mj == RM_MODE_PUSH) depth++;
            if (depth == 1) break;
        }
        if (j == rec.size()) return false;  // the program ends with more than one value: not a chain
        u.push_back({BU_GENERIC, (int)i, (int)j, -1});
        i = j + 1;
    }
    // worth it from a handful of units on; one bit per unit
    const int min_units = jit_knob("RM_JIT_CACHED_MIN_UNITS", 4);
    if ((int)u.size() < min_units || u.size() > 64) return false;
    *out = std::move(u);
    return true;
}

// Plain code for records [first, last] of `rec` on a stack of named values (as generate_map_scene without pruning); used for
// the generic units.  `pfx` prefixes the value names so that two units never collide.
inline bool emit_plain_records(const std::vector<RmRecord>& rec, int first, int last, std::vector<std::string>& stack, const char* pfx,
                               const char* counted, std::string& s) {
    char line[768];
    int nv = 0;
    for (int i = first; i <= last; i++) {
        const uint32_t kind = RM_OP_KIND(rec[(size_t)i].op), mode = RM_OP_MODE(rec[(size_t)i].op);
        const unsigned off = (unsigned)i * 8u;
        const char* op = mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr;
        char w[48];
        std::snprintf(w, sizeof w, "%s%d", pfx, nv++);
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
            const std::string b = stack.back(); stack.pop_back();
            const std::string a = stack.back(); stack.pop_back();
            if (mode == RM_MODE_SMOOTH) std::snprintf(line, sizeof line, "    const float %s = spec_smooth_union(lp + %u, %s, %s, live);\n", w, off, a.c_str(), b.c_str());
            else if (op) std::snprintf(line, sizeof line, "    const float %s = %s(%s, %s);\n", w, op, a.c_str(), b.c_str());
            else return false;
            s += line;
            stack.push_back(w);
            continue;
        }
        const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>" : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>" : nullptr;
        if (!fn || mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;
        if (mode == RM_MODE_PUSH) {
            std::snprintf(line, sizeof line, "    const float %s = %s(lp + %u, x0, y0, z0, tiny); %s\n", w, fn, off, counted);
        } else {
            if (stack.empty()) return false;
            const std::string a = stack.back(); stack.pop_back();
            std::snprintf(line, sizeof line, "    const float %s = %s(%s, %s(lp + %u, x0, y0, z0, tiny)); %s\n", w, op, a.c_str(), fn, off, counted);
        }
        s += line;
        stack.push_back(w);
    }
    return true;
}

inline bool generate_blend_cached(const std::vector<RmRecord>& rec, std::string* out) {
    std::vector<BlendUnit> units;
    if (!blend_units(rec, &units)) return false;
    const int count_mode = jit_knob("RM_JIT_PRUNE_STATS", 0);  // 1 leaves evaluated
    const char* counted = count_mode == 1 ? "n_eval += 1u; " : "";
    const int leaf_tests = jit_knob("RM_JIT_BLEND_LEAF_TESTS", 0);  // members of a near pair in a REFRESH: 0 untested (default), 1 tested
    const int sched_every = std::getenv("RM_JIT_SCHED_BARRIER") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER")) : 4;
    std::map<int, int> pair_index;  // first leaf record -> index of the pair's group record
    {
        const std::vector<std::pair<int, int>> pairs = rm_blend_pairs(rec);
        for (size_t g = 0; g < pairs.size(); g++) pair_index[pairs[g].first] = (int)g;
    }
    char line[1024];
    std::string r, c;  // refresh, cached
    r += "namespace rmk {\ntemplate <bool FAST>\n"
         "RM_DEV float map_scene_refresh(LdsF lp, float qx, float qy, float qz, float thr, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval, SpecCache& cache) {\n"
         "    const float m = thr;\n    const float x0 = qx, y0 = qy, z0 = qz;\n    float rbud = __uint_as_float(0x7F800000u);\n";
    c += "namespace rmk {\ntemplate <bool FAST>\n"
         "RM_DEV float map_scene_cached(LdsF lp, float qx, float qy, float qz, unsigned long long skip, uint32_t jstar, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval) {\n"
         "    const float x0 = qx, y0 = qy, z0 = qz;\n";
    for (auto& kv : pair_index) {
        std::snprintf(line, sizeof line, "    float kr%d; const float pa%d = spec_pair_a(lp + %u, x0, y0, z0, kr%d);\n", kv.second, kv.second,
                      (unsigned)(rec.size() + (size_t)kv.second) * 8u, kv.second);
        r += line;
    }
    int leaves = 0;
    auto barrier = [&](std::string& s) {
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    };
    // a local member (leaf record i) in the refresh: vW (declared by the caller, = vA) becomes op(vA, leaf) unless its own test
    // skips it; an evaluated member is a restart candidate
    auto refresh_member = [&](size_t i, int a, int w, bool test, unsigned long long bit_if_single) {
        const RmLeafUse use = rm_leaf_use(rec, i);
        const uint32_t kind = RM_OP_KIND(rec[i].op);
        const unsigned off = (unsigned)i * 8u, koff = use.k_rec >= 0 ? (unsigned)use.k_rec * 8u : 0u;
        char kexpr[64], apply_open[96];
        if (use.k_rec >= 0) {
            std::snprintf(kexpr, sizeof kexpr, "spec_local_k(lp + %u)", koff);
            std::snprintf(apply_open, sizeof apply_open, "spec_smooth_union(lp + %u, v%d, ", koff, a);
        } else {
            std::snprintf(kexpr, sizeof kexpr, "0.0f");
            std::snprintf(apply_open, sizeof apply_open, "vmin(v%d, ", a);
        }
        const char* apply_close = use.k_rec >= 0 ? ", live)" : ")";
        std::snprintf(line, sizeof line, "    { const float kk = %s;\n", kexpr);
        r += line;
        if (kind == RM_KIND_SPHERE) {
            if (test) {
                std::snprintf(line, sizeof line,
                              "      const float a = spec_sphere_a(lp + %u, x0, y0, z0); const float rhs = ((v%d + m) + kk) + lp[%u];\n"
                              "      if (spec_local_near(live, a, rhs)) { const float t = spec_sphere_v<FAST>(lp + %u, a, tiny); %s\n"
                              "        spec_cache_restart(cache, %uu, v%d, kk, t, rbud); v%d = %st%s; }\n",
                              off, a, off + 3u, off, counted, (unsigned)i, a, w, apply_open, apply_close);
                r += line;
                if (bit_if_single) {
                    std::snprintf(line, sizeof line, "      else spec_cache_far(cache, live, 0x%llxull, (sqrt_lo(a) - rhs) * 0.5f);\n", bit_if_single);
                    r += line;
                }
            } else {
                std::snprintf(line, sizeof line, "      const float t = spec_sphere<FAST>(lp + %u, x0, y0, z0, tiny); %s\n"
                              "      spec_cache_restart(cache, %uu, v%d, kk, t, rbud); v%d = %st%s;\n", off, counted, (unsigned)i, a, w, apply_open, apply_close);
                r += line;
            }
        } else {
            if (test) {
                std::snprintf(line, sizeof line,
                              "      const SpecBox b = spec_box_a(lp + %u, x0, y0, z0); const float rhs = (v%d + m) + kk;\n"
                              "      if (spec_local_box_near(live, b.a, rhs)) { const float t = spec_box_v<FAST>(b, tiny); %s\n"
                              "        spec_cache_restart(cache, %uu, v%d, kk, t, rbud); v%d = %st%s; }\n",
                              off, a, counted, (unsigned)i, a, w, apply_open, apply_close);
                r += line;
                if (bit_if_single) {
                    std::snprintf(line, sizeof line, "      else spec_cache_far(cache, live, 0x%llxull, (sqrt_lo(b.a) - rhs) * 0.5f);\n", bit_if_single);
                    r += line;
                }
            } else {
                std::snprintf(line, sizeof line, "      const float t = spec_box<FAST>(lp + %u, x0, y0, z0, tiny); %s\n"
                              "      spec_cache_restart(cache, %uu, v%d, kk, t, rbud); v%d = %st%s;\n", off, counted, (unsigned)i, a, w, apply_open, apply_close);
                r += line;
            }
        }
        r += "    }\n";
    };
    // the same member in a cached evaluation: no test; a lane whose restart is this leaf starts its accumulator here
    auto cached_member = [&](size_t i, int a, int w) {
        const RmLeafUse use = rm_leaf_use(rec, i);
        const uint32_t kind = RM_OP_KIND(rec[i].op);
        const unsigned off = (unsigned)i * 8u;
        char applied[160];
        if (use.k_rec >= 0) std::snprintf(applied, sizeof applied, "spec_smooth_union(lp + %u, v%d, t, live)", (unsigned)use.k_rec * 8u, a);
        else std::snprintf(applied, sizeof applied, "vmin(v%d, t)", a);
        std::snprintf(line, sizeof line, "      { const float t = %s<FAST>(lp + %u, x0, y0, z0, tiny); %sv%d = jstar == %uu ? t : %s; }\n",
                      kind == RM_KIND_SPHERE ? "spec_sphere" : "spec_box", off, counted, w, (unsigned)i, applied);
        c += line;
    };
    int nv = 0, acc = -1;
    for (size_t ui = 0; ui < units.size(); ui++) {
        const BlendUnit& u = units[ui];
        const unsigned long long bit = 1ull << ui;
        const unsigned off = (unsigned)u.first * 8u;
        const uint32_t kind = RM_OP_KIND(rec[(size_t)u.first].op);
        if (u.kind == BU_START) {
            const char* fn = kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>" : "spec_cylinder<FAST>";
            acc = nv++;
            std::snprintf(line, sizeof line, "    const float v%d = %s(lp + %u, x0, y0, z0, tiny); %s\n", acc, fn, off, counted);
            r += line;
            std::snprintf(line, sizeof line, "    float v%d = __uint_as_float(0x7F800000u);\n    if (!(skip & 0x%llxull)) { v%d = %s(lp + %u, x0, y0, z0, tiny); %s}\n",
                          acc, bit, acc, fn, off, counted);
            c += line;
        } else if (u.kind == BU_SINGLE) {
            const int w = nv++;
            std::snprintf(line, sizeof line, "    float v%d = v%d;\n", w, acc);
            r += line;
            refresh_member((size_t)u.first, acc, w, true, bit);
            std::snprintf(line, sizeof line, "    float v%d = v%d;\n    if (!(skip & 0x%llxull))\n", w, acc, bit);
            c += line;
            cached_member((size_t)u.first, acc, w);
            acc = w;
            barrier(r); barrier(c);
        } else if (u.kind == BU_PAIR) {
            const int g = pair_index[u.first];
            const int w1 = nv++, w2 = nv++;
            std::snprintf(line, sizeof line, "    float v%d = v%d, v%d = v%d;\n    { const float rhs = (v%d + m) + kr%d;\n    if (spec_local_near(live, pa%d, rhs)) {\n",
                          w1, acc, w2, acc, acc, g, g);
            r += line;
            refresh_member((size_t)u.first, acc, w1, leaf_tests == 1, 0ull);
            std::snprintf(line, sizeof line, "    v%d = v%d;\n", w2, w1);
            r += line;
            refresh_member((size_t)u.second, w1, w2, leaf_tests == 1, 0ull);
            std::snprintf(line, sizeof line, "    } else spec_cache_far(cache, live, 0x%llxull, (sqrt_lo(pa%d) - rhs) * 0.5f); }\n", bit, g);
            r += line;
            std::snprintf(line, sizeof line, "    float v%d = v%d, v%d = v%d;\n    if (!(skip & 0x%llxull)) {\n", w1, acc, w2, acc, bit);
            c += line;
            cached_member((size_t)u.first, acc, w1);
            cached_member((size_t)u.second, w1, w2);
            c += "    }\n";
            acc = w2;
            barrier(r); barrier(r); barrier(c); barrier(c);
        } else if (u.kind == BU_SUB) {
            // max(acc, -v) = acc while v + acc >= 0: outside the leaf (v > 0) with acc >= 0 is what the test establishes; the
            // slack of that is the smaller of the two
            const int w = nv++;
            std::snprintf(line, sizeof line, "    float v%d = v%d;\n", w, acc);
            r += line;
            if (kind == RM_KIND_SPHERE)
                std::snprintf(line, sizeof line,
                              "    { const float a = spec_sphere_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_sub_sphere_near(live, lp + %u, a, v%d)) { v%d = vmax_negb(v%d, spec_sphere_v<FAST>(lp + %u, a, tiny)); %s}\n"
                              "      else spec_cache_far(cache, live, 0x%llxull, fmin_(sqrt_lo(a) - lp[%u], v%d) - m); }\n",
                              off, off, acc, w, acc, off, counted, bit, off + 4u, acc);
            else
                std::snprintf(line, sizeof line,
                              "    { const SpecBox b = spec_box_a(lp + %u, x0, y0, z0);\n"
                              "      if (spec_sub_box_near(live, b.a, v%d)) { v%d = vmax_negb(v%d, spec_box_v<FAST>(b, tiny)); %s}\n"
                              "      else spec_cache_far(cache, live, 0x%llxull, fmin_(sqrt_lo(b.a), v%d) - m); }\n",
                              off, acc, w, acc, counted, bit, acc);
            r += line;
            std::snprintf(line, sizeof line, "    float v%d = v%d;\n    if (!(skip & 0x%llxull)) { v%d = vmax_negb(v%d, %s<FAST>(lp + %u, x0, y0, z0, tiny)); %s}\n",
                          w, acc, bit, w, acc, kind == RM_KIND_SPHERE ? "spec_sphere" : "spec_box", off, counted);
            c += line;
            acc = w;
            barrier(r); barrier(c);
        } else {  // BU_GENERIC: evaluated as it stands; skipped in a cached evaluation only when it is dead (in front of every restart)
            char pfx[32], accname[32];
            std::snprintf(pfx, sizeof pfx, "g%d_", (int)ui);
            std::snprintf(accname, sizeof accname, "v%d", acc);
            const int w = nv++;
            for (int which = 0; which < 2; which++) {
                std::string& s = which ? c : r;
                std::vector<std::string> st;
                st.push_back(accname);
                std::snprintf(line, sizeof line, which ? "    float v%d = v%d;\n    if (!(skip & 0x%llxull)) {\n" : "    float v%d = v%d;\n    {\n", w, acc, bit);
                s += line;
                if (!emit_plain_records(rec, u.first, u.last, st, pfx, counted, s) || st.size() != 1) return false;
                std::snprintf(line, sizeof line, "    v%d = %s;\n    }\n", w, st.back().c_str());
                s += line;
            }
            acc = w;
            barrier(r); barrier(c);
        }
    }
    // units in front of every live lane's restart are dead (one comparison each, in order, until one is alive)
    r += "    cache.budget = fmin_(cache.budget, rbud);\n";
    std::string tail;
    int opened = 0;
    for (size_t ui = 0; ui + 1 < units.size(); ui++) {  // the last unit holds the last possible restart: never dead
        std::snprintf(line, sizeof line, "    if ((__builtin_amdgcn_ballot_w64(cache.jstar <= %uu) & live) == 0ull) { cache.skip |= 0x%llxull;\n", (unsigned)units[ui].last, 1ull << ui);
        tail += line;
        opened++;
    }
    for (int k = 0; k < opened; k++) tail += "}";
    r += tail + "\n";
    std::snprintf(line, sizeof line, "    return v%d;\n}\n", acc);
    r += line;
    c += line;
    // map_scene_spec (one-position taps, should the four-tap function not be generated): the refresh without its book-keeping
    r += "template <bool FAST>\nRM_DEV float map_scene_spec(LdsF lp, float qx, float qy, float qz, float thr, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval) {\n"
         "    SpecCache unused;\n    return map_scene_refresh<FAST>(lp, qx, qy, qz, thr, live, tiny, n_eval, unused);\n}\n}  // namespace rmk\n";
    c += "}  // namespace rmk\n";
    *out = r + c;
    return true;
}


// The material walk of a tagged program (rm_interp.h map_scene_material) as straight-line code: one evaluation of the
// program WITH its Material tags at the position of a hit, every value a (distance, index) pair of named variables --
// no stack in LDS, no decode.  Distances go through the operations map_scene_material applies (the interpreter's leaf
// functions and operators; the short exact sqrt with its guard, the generic one when the guard objects), indices through its selection rules: primitives carry 0, a tag overwrites
// the index of the value on top, an operator keeps the index of the operand that decides its result (Union / SmoothUnion
// b < a, Subtraction -b > a, Intersection b > a take b's; ties and NaN a's).  Parameters -- tag indices included -- are
// read from the device copy of the tagged records (uniform addresses: scalar loads), so they stay data.
inline bool generate_material_walk(const std::vector<RmRecord>& mrec, std::string* out) {
    std::string s;
    char line[768];
    s += "namespace rmk {\n";
    s += "template <bool FAST>\n";
    s += "RM_DEV uint32_t map_scene_material_spec(const RmRecord* __restrict__ mp, float qx, float qy, float qz, SqrtGuard& tiny) {\n";
    s += "    const float x0 = qx, y0 = qy, z0 = qz;\n";
    std::vector<int> stack, pos;
    pos.push_back(0);
    int nv = 0, np = 0, leaves = 0;
    for (size_t i = 0; i < mrec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(mrec[i].op), mode = RM_OP_MODE(mrec[i].op);
        const unsigned r = (unsigned)i;
        if (kind == RM_KIND_MATERIAL) {  // tags the value on top
            if (stack.empty()) return false;
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            std::snprintf(line, sizeof line, "    const float v%d = v%d; const uint32_t m%d = __float_as_uint(mp[%u].p[0]);\n", w, a, w, r);
            s += line;
            stack.push_back(w);
            continue;
        }
        if (kind == RM_KIND_XFORM) {
            const int c = pos.back();
            if ((mode & 1u) == 0u) {
                const int n = ++np;
                if (mode == RM_XF_T_PUSH)
                    std::snprintf(line, sizeof line, "    const float x%d = x%d - mp[%u].p[0], y%d = y%d - mp[%u].p[1], z%d = z%d - mp[%u].p[2];\n", n, c, r, n, c, r, n, c, r);
                else if (mode == RM_XF_R_PUSH)
                    std::snprintf(line, sizeof line, "    float x%d = x%d, y%d = y%d, z%d = z%d; xf_rotate_conj(mp[%u].p[0], mp[%u].p[1], mp[%u].p[2], mp[%u].p[3], x%d, y%d, z%d);\n",
                                  n, c, n, c, n, c, r, r, r, r, n, n, n);
                else
                    std::snprintf(line, sizeof line, "    const float x%d = x%d / mp[%u].p[0], y%d = y%d / mp[%u].p[0], z%d = z%d / mp[%u].p[0];\n", n, c, r, n, c, r, n, c, r);
                s += line;
                pos.push_back(n);
            } else {
                if (pos.size() < 2 || stack.empty()) return false;
                pos.pop_back();
                if (mode == RM_XF_S_POP) {
                    const int a = stack.back(); stack.pop_back();
                    const int w = nv++;
                    std::snprintf(line, sizeof line, "    const float v%d = v%d * mp[%u].p[0]; const uint32_t m%d = m%d;\n", w, a, r, w, a);
                    s += line;
                    stack.push_back(w);
                }
            }
            continue;
        }
        // binary operator on (a, b): value expression and the "b decides" predicate, as map_scene_material
        auto combine = [&](int w, const char* a, const char* am, const char* b, const char* bm) -> bool {
            if (mode == RM_MODE_UNION)
                std::snprintf(line, sizeof line, "    const float v%d = vmin(%s, %s); const uint32_t m%d = %s < %s ? %s : %s;\n", w, a, b, w, b, a, bm, am);
            else if (mode == RM_MODE_SUB)
                std::snprintf(line, sizeof line, "    const float v%d = vmax_negb(%s, %s); const uint32_t m%d = -%s > %s ? %s : %s;\n", w, a, b, w, b, a, bm, am);
            else if (mode == RM_MODE_INTER)
                std::snprintf(line, sizeof line, "    const float v%d = fmax_(%s, %s); const uint32_t m%d = %s > %s ? %s : %s;\n", w, a, b, w, b, a, bm, am);
            else if (mode == RM_MODE_SMOOTH)
                std::snprintf(line, sizeof line, "    const float v%d = material_smooth_union(mp[%u].p[0], %s, %s); const uint32_t m%d = %s < %s ? %s : %s;\n",
                              w, r, a, b, w, b, a, bm, am);
            else
                return false;
            s += line;
            return true;
        };
        if (kind == RM_KIND_POP) {
            if (stack.size() < 2) return false;
            const int b = stack.back(); stack.pop_back();
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            char an[16], am[16], bn[16], bm[16];
            std::snprintf(an, sizeof an, "v%d", a); std::snprintf(am, sizeof am, "m%d", a);
            std::snprintf(bn, sizeof bn, "v%d", b); std::snprintf(bm, sizeof bm, "m%d", b);
            if (!combine(w, an, am, bn, bm)) return false;
            stack.push_back(w);
            continue;
        }
        const char* fn = kind == RM_KIND_SPHERE ? "sdf_sphere_t<FAST>" : kind == RM_KIND_BOX ? "sdf_box_t<FAST>"
                       : kind == RM_KIND_CYLINDER ? "sdf_cylinder_t<FAST>" : nullptr;
        const int c = pos.back();
        const int leaf = nv++;  // the leaf's own value; its index is 0
        if (fn) std::snprintf(line, sizeof line, "    const float v%d = %s(x%d, y%d, z%d, mp[%u].p, tiny);\n", leaf, fn, c, c, c, r);
        else if (kind == RM_KIND_PLANE)
            std::snprintf(line, sizeof line, "    const float v%d = ((x%d * mp[%u].p[0] + y%d * mp[%u].p[1]) + z%d * mp[%u].p[2]) + mp[%u].p[3];\n", leaf, c, r, c, r, c, r, r);
        else return false;
        s += line;
        if (fn) s += "    guard_fence(tiny);\n";
        if (++leaves % 2 == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";  // as in map_scene_spec: bounds the compiler's hoisting
        if (mode == RM_MODE_PUSH) {
            std::snprintf(line, sizeof line, "    const uint32_t m%d = 0u;\n", leaf);
            s += line;
            stack.push_back(leaf);
        } else {  // leaf fused with the operator that consumes it: a = accumulator, b = leaf (index 0)
            if (stack.empty() || mode == RM_MODE_SMOOTH) return false;
            const int a = stack.back(); stack.pop_back();
            const int w = nv++;
            char an[16], am[16], bn[16];
            std::snprintf(an, sizeof an, "v%d", a); std::snprintf(am, sizeof am, "m%d", a); std::snprintf(bn, sizeof bn, "v%d", leaf);
            if (!combine(w, an, am, bn, "0u")) return false;
            stack.push_back(w);
        }
    }
    if (stack.empty()) return false;
    std::snprintf(line, sizeof line, "    return m%d;\n}\n}  // namespace rmk\n", stack.back());
    s += line;
    *out = std::move(s);
    return true;
}

inline const char* kernel_name() { return "rm_render_v5_spec"; }

// mrec: the program decoded with its Material tags (empty for an untagged program): the kernel then gets the material
// phase, with the walk generated as code when jit_knob RM_JIT_MATERIAL_WALK allows (default) and possible.
// Whether a program of this STRUCTURE can meet the miss test on lower bounds (RmDecoded::bound_walk, which also looks at
// the parameters): only then is the test compiled into its kernel -- it costs two or three registers the others need.
inline bool structure_allows_bound_walk(const std::vector<RmRecord>& rec) {
    bool smooth = false;
    int spilled = 0, depth = 0;
    for (const RmRecord& r : rec) {
        const uint32_t kind = RM_OP_KIND(r.op);
        if (kind == RM_KIND_XFORM || kind == RM_KIND_MATERIAL) return false;
        smooth = smooth || RM_OP_MODE(r.op) == RM_MODE_SMOOTH || RM_OP_MODE(r.op) == RM_MODE_INTER || kind == RM_KIND_PLANE;
        if (kind == RM_KIND_POP) spilled--;
        else if (r.op & RM_OP_SPILL) spilled++;
        depth = spilled > depth ? spilled : depth;
    }
    return smooth && depth <= 1;
}

// prune: 0 every leaf is evaluated, 1 far-primitive pruning on a threshold (lattice programs: rm_kernel_v5.h "Pruning"), 2 the
// local skipping rule of programs that blend (rm_groups.h)
enum : int { PRUNE_NONE = 0, PRUNE_LATTICE = 1, PRUNE_BLEND = 2 };
inline bool generate_source(const std::vector<RmRecord>& rec, const std::vector<RmRecord>& mrec, int wpt, int prune_kind, std::string* out,
                            bool* walk_generated = nullptr) {
    const bool materials = !mrec.empty();
    const bool prune = prune_kind == PRUNE_LATTICE, blend = prune_kind == PRUNE_BLEND;
    std::string body, taps, walk;
    // programs that blend: skip sets carried along the ray where the top level is a chain (generate_blend_cached), else the
    // local rule tested at every evaluation
    const bool cached = blend && jit_knob("RM_JIT_CACHED", 0) != 0 && generate_blend_cached(rec, &body);  // measured slower: off
    if (!cached && !(blend ? generate_map_scene_blend(rec, &body) : generate_map_scene(rec, prune, &body))) return false;
    const bool walk_spec = materials && jit_knob("RM_JIT_MATERIAL_WALK", 1) != 0 && mrec.size() <= kMaxRecords && generate_material_walk(mrec, &walk);
    if (walk_generated) *walk_generated = walk_spec;
    const char* taps_knob = std::getenv("RM_JIT_TAPS4");  // A/B: RM_JIT_TAPS4=0 keeps the taps on map_scene_spec
    const bool taps4 = !(taps_knob && std::atoi(taps_knob) == 0) &&
                       (blend ? generate_map_scene_taps_blend(rec, &taps) : generate_map_scene_taps(rec, prune, &taps));
    std::string s;
    // hipRTC's built-in runtime header keeps the fixed-width integer types in a namespace of its own
    s += "typedef unsigned char rm_rtc_u8;\ntypedef unsigned short rm_rtc_u16;\ntypedef unsigned int rm_rtc_u32;\n"
         "typedef unsigned long long rm_rtc_u64;\ntypedef int rm_rtc_i32;\ntypedef long long rm_rtc_i64;\n"
         "#define uint8_t rm_rtc_u8\n#define uint16_t rm_rtc_u16\n#define uint32_t rm_rtc_u32\n#define uint64_t rm_rtc_u64\n"
         "#define int32_t rm_rtc_i32\n#define int64_t rm_rtc_i64\n";
    s += "#define RM_JIT_TU 1\n";
    if (prune) s += "#define RM_JIT_PRUNE_ON 1\n";
    if (blend) s += "#define RM_JIT_BLEND_PRUNE 1\n";
    if (cached) s += "#define RM_JIT_CACHED 1\n";
    if (cached && jit_knob("RM_JIT_PRUNE_STATS", 0) == 5) s += "#define RM_JIT_COUNT_REFRESH 1\n";
    if (taps4) s += "#define RM_JIT_TAPS4 1\n";
    if (walk_spec) s += "#define RM_JIT_MATERIAL_WALK 1\n";
    if (structure_allows_bound_walk(rec)) s += "#define RM_JIT_BOUND_WALK 1\n";
    if (const char* pr = std::getenv("RM_JIT_PRIO_LONG_RAYS")) {  // experiment knob
        s += "#define RM_PRIO_LONG_RAYS ";
        s += std::to_string(std::atoi(pr));
        s += "u\n";
    }
    s += "#include \"rm_kernel_v5.h\"\n";
    s += body;
    if (taps4) s += taps;
    if (walk_spec) s += walk;
    char line[512];
    if (const char* w = std::getenv("RM_JIT_WAVES_PER_EU")) {  // experiment knob: cap the VGPR budget
        std::snprintf(line, sizeof line, "__attribute__((amdgpu_waves_per_eu(%d, %d)))\n", std::atoi(w), std::atoi(w));
        s += line;
    }
    std::snprintf(line, sizeof line,
                  "extern \"C\" __global__ __launch_bounds__(%d) void %s(RmLaunch L, rmk::V5Work work, uint32_t n_tiles, "
                  "uint32_t refill_min) {\n    rmk::rm_render_v5_body<rmk::ProgLds, true, %d, false, true, %s>(L, work, n_tiles, refill_min);\n}\n",
                  64 * wpt, kernel_name(), wpt, materials ? "true" : "false");
    s += line;
    *out = std::move(s);
    return true;
}

// Disk-cache file (RM_JIT_CACHE_DIR): a 32-byte header in front of the code object, so that a truncated or corrupt
// file is recognised when it is read and not when hipModuleLoadData chokes on it.
struct CacheHeader {
    char magic[8];       // "RMJITCO\1"
    uint64_t bytes;      // length of the code object that follows
    uint64_t checksum;   // FNV-1a of those bytes
    uint64_t reserved;
};
inline uint64_t fnv1a(const char* p, size_t n, uint64_t h = 1469598103934665603ull) {
    for (size_t i = 0; i < n; i++) { h ^= (unsigned char)p[i]; h *= 1099511628211ull; }
    return h;
}
inline const char* const* compile_options(int* n) {
    // the flags of the offline build (build.py HIP_FLAGS) that affect code generation.  RM_JIT_OPT_LEVEL (diagnostics, read once):
    // another -O level
    static const char* level = [] {
        const char* v = std::getenv("RM_JIT_OPT_LEVEL");
        return v && std::strlen(v) == 1 && std::strchr("0123s", v[0]) ? (v[0] == '0' ? "-O0" : v[0] == '1' ? "-O1" : v[0] == '2' ? "-O2" : v[0] == 's' ? "-Os" : "-O3") : "-O3";
    }();
    static const char* const opts[] = {"--offload-arch=gfx950", level, "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize"};
    *n = (int)(sizeof opts / sizeof *opts);
    return opts;
}
// Where the code object of `src` lives in the disk cache ("" when the cache is off): keyed by everything that
// determines it -- the generated source, the embedded headers, the compile options, the compiler (file + version).
inline std::string cache_path(const std::string& src) {
    const char* dir = std::getenv("RM_JIT_CACHE_DIR");
    Rtc& rtc = Rtc::get();
    if (!dir || !rtc.ok()) return std::string();
    uint64_t h = fnv1a(src.data(), src.size());
    for (int i = 0; i < kNumHeaders; i++) h = fnv1a(kHeaderSources[i], std::strlen(kHeaderSources[i]), h);
    int n_opts = 0;
    const char* const* opts = compile_options(&n_opts);
    for (int i = 0; i < n_opts; i++) h = fnv1a(opts[i], std::strlen(opts[i]) + 1, h);
    h = fnv1a(rtc.identity.data(), rtc.identity.size(), h);
    char name[64];
    std::snprintf(name, sizeof name, "/rm_%016llx.co", (unsigned long long)h);
    return std::string(dir) + name;
}

// Compile `src` for gfx950.  No HIP runtime call is made: this runs on worker threads and on
// machines without a GPU (the build check).  *from_cache (nullable) says whether the code came from the disk cache.
inline bool compile(const std::string& src, std::vector<char>* code, std::string* log, double* ms, bool* from_cache = nullptr) {
    Rtc& rtc = Rtc::get();
    if (from_cache) *from_cache = false;
    if (!rtc.ok()) { *log = rtc.error; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    // Optional disk cache (RM_JIT_CACHE_DIR): a second process starts warm.
    const std::string cache_file = cache_path(src);
    if (!cache_file.empty()) {
        if (FILE* f = std::fopen(cache_file.c_str(), "rb")) {
            CacheHeader hd;
            bool ok = std::fread(&hd, 1, sizeof hd, f) == sizeof hd && std::memcmp(hd.magic, "RMJITCO\1", 8) == 0 &&
                      hd.bytes > 64 && hd.bytes < (1ull << 30);
            if (ok) {
                code->resize((size_t)hd.bytes);
                ok = std::fread(code->data(), 1, code->size(), f) == code->size() && std::fgetc(f) == EOF &&
                     std::memcmp(code->data(), "\177ELF", 4) == 0 && fnv1a(code->data(), code->size()) == hd.checksum;
            }
            std::fclose(f);
            if (ok) {
                *log = "loaded from " + cache_file;
                if (from_cache) *from_cache = true;
                if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
                return true;
            }
            code->clear();
            std::remove(cache_file.c_str());  // truncated or corrupt: it is rewritten below
        }
    }
    Rtc::Program prog = nullptr;
    int rc = rtc.CreateProgram(&prog, src.c_str(), "rm_spec.hip", (int)kNumHeaders, kHeaderSources, kHeaderNames);
    if (rc != 0) { *log = "hiprtcCreateProgram failed: " + std::to_string(rc); return false; }
    int n_opts = 0;
    const char* const* opts = compile_options(&n_opts);
    rc = rtc.CompileProgram(prog, n_opts, opts);
    size_t n = 0;
    if (rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) {
        log->resize(n);
        rtc.GetProgramLog(prog, &(*log)[0]);
    }
    bool ok = rc == 0;
    if (ok) {
        n = 0;
        ok = rtc.GetCodeSize(prog, &n) == 0 && n > 0;
        if (ok) {
            code->resize(n);
            ok = rtc.GetCode(prog, code->data()) == 0;
        }
    } else if (log->empty()) {
        *log = "hiprtcCompileProgram failed: " + std::to_string(rc);
    }
    rtc.DestroyProgram(&prog);
    if (ok && !cache_file.empty()) {  // write-then-rename: another process never sees half a file
        const std::string tmp = cache_file + ".part" + std::to_string((long)getpid());
        if (FILE* f = std::fopen(tmp.c_str(), "wb")) {
            CacheHeader hd;
            std::memset(&hd, 0, sizeof hd);
            std::memcpy(hd.magic, "RMJITCO\1", 8);
            hd.bytes = code->size();
            hd.checksum = fnv1a(code->data(), code->size());
            const bool w = std::fwrite(&hd, 1, sizeof hd, f) == sizeof hd && std::fwrite(code->data(), 1, code->size(), f) == code->size();
            const bool closed = std::fclose(f) == 0;
            if (!w || !closed || std::rename(tmp.c_str(), cache_file.c_str()) != 0) std::remove(tmp.c_str());
        }
    }
    if (const char* dir = std::getenv("RM_JIT_DUMP_DIR")) {  // diagnostics: keep what was compiled
        static std::atomic<int> serial{0};
        const std::string base = std::string(dir) + "/rm_spec_" + std::to_string(serial++);
        if (FILE* f = std::fopen((base + ".hip").c_str(), "wb")) { std::fwrite(src.data(), 1, src.size(), f); std::fclose(f); }
        if (ok)
            if (FILE* f = std::fopen((base + ".co").c_str(), "wb")) { std::fwrite(code->data(), 1, code->size(), f); std::fclose(f); }
    }
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return ok;
}

// ---- cache ------------------------------------------------------------------------------------------
struct Entry {
    enum State : int { COMPILING = 0, READY = 1, FAILED = 2 };
    std::mutex m;
    std::condition_variable cv;
    State state = COMPILING;
    std::vector<char> code;  // gfx950 code object
    std::string log;
    double compile_ms = 0.0;
    bool material_walk = false;  // the kernel carries the generated material walk (it needs no LDS stack for the material phase)
    std::string cached_source;  // non-empty iff `code` came from the disk cache: if the loader rejects it, the file is
                                // dropped and this source compiled afresh, once (rm_abi.hip specialised_kernel)
    // Filled by the caller's (HIP) thread under `m`: device ordinal -> {hipModule_t, hipFunction_t}.
    struct Loaded { void* module = nullptr; void* function = nullptr; };
    std::map<int, Loaded> loaded;
    void (*unload)(void* module, int device) = nullptr;  // waits for the device, then hipModuleUnload; set by whoever loads

    State wait() {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return state != COMPILING; });
        return state;
    }
    State peek() {
        std::lock_guard<std::mutex> lk(m);
        return state;
    }
    ~Entry() {
        if (unload)
            for (auto& kv : loaded)
                if (kv.second.module) unload(kv.second.module, kv.first);
    }
};

class Cache {
public:
    static constexpr size_t kMaxEntries = 256;  // beyond this, entries no context holds are dropped

    static Cache& get() {
        static Cache* c = new Cache;  // never destroyed: see shutdown()
        return *c;
    }
    // The entry for (rec structure, wpt); queues its compilation for the worker thread the first time.
    // materials: the program carries Material tags (the kernel gets the material phase; the structure is that of
    // the untagged program)
    std::shared_ptr<Entry> request(const std::vector<RmRecord>& rec, const std::vector<RmRecord>& mrec, int wpt, int prune) {
        // a tagged program's kernel also depends on where its tags sit (the material walk is generated from mrec), and on the
        // A/B knobs of the generator as the environment holds them now (so that a process may compare two settings)
        static const char* const knobs[] = {"RM_JIT_BLEND_LEAF_TESTS", "RM_JIT_BLEND_UPFRONT", "RM_JIT_CACHED", "RM_JIT_CACHED_MIN_UNITS", "RM_JIT_GUARD_FENCE", "RM_JIT_LEAF_TESTS", "RM_JIT_MATERIAL_WALK",
                                            "RM_JIT_PRIO_LONG_RAYS", "RM_JIT_PRUNE_STATS", "RM_JIT_SCHED_BARRIER", "RM_JIT_SCHED_BARRIER_TAPS",
                                            "RM_JIT_SUB_TESTS", "RM_JIT_TAPS4", "RM_JIT_TAPS4_SMOOTH", "RM_JIT_WAVES_PER_EU"};
        std::string knob_key;
        for (const char* name : knobs)
            if (const char* v = std::getenv(name)) knob_key += std::string("|") + name + "=" + v;
        const std::string key = std::to_string(wpt) + (prune == PRUNE_LATTICE ? "p" : prune == PRUNE_BLEND ? "b" : "") + ":" + structure_key(rec) +
                                (mrec.empty() ? "" : "|m:" + structure_key(mrec)) + knob_key;
        std::unique_lock<std::mutex> lk(m_);
        auto it = entries_.find(key);
        if (it != entries_.end()) return it->second;
        if (entries_.size() >= kMaxEntries) {
            for (auto j = entries_.begin(); j != entries_.end();)
                j = (j->second.use_count() == 1 && j->second->peek() != Entry::COMPILING) ? entries_.erase(j) : ++j;
        }
        auto e = std::make_shared<Entry>();
        entries_[key] = e;
        Job job;
        job.entry = e;
        if (!generate_source(rec, mrec, wpt, prune, &job.source, &e->material_walk)) {
            e->state = Entry::FAILED;
            e->log = "program structure could not be turned into code";
            return e;
        }
        queue_.push_back(std::move(job));
        if (!worker_.joinable()) {
            worker_ = std::thread([this] { run(); });
            std::atexit([] { Cache::get().shutdown(); });
        }
        lk.unlock();
        cv_.notify_one();
        return e;
    }
    size_t size() {
        std::lock_guard<std::mutex> lk(m_);
        return entries_.size();
    }
    // A process must not run its static destructors underneath a compiling thread: finish the job in
    // flight, drop the rest.  Modules stay loaded; the HIP runtime may already be gone.
    void shutdown() {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
    }

private:
    struct Job { std::shared_ptr<Entry> entry; std::string source; };
    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !queue_.empty(); });
                if (stop_) break;
                job = std::move(queue_.front());
                queue_.erase(queue_.begin());
            }
            std::vector<char> code;
            std::string log;
            double ms = 0.0;
            bool from_cache = false;
            const bool ok = compile(job.source, &code, &log, &ms, &from_cache);
            {
                std::lock_guard<std::mutex> g(job.entry->m);
                if (ok && from_cache) job.entry->cached_source = std::move(job.source);
                job.entry->code = std::move(code);
                job.entry->log = std::move(log);
                job.entry->compile_ms = ms;
                job.entry->state = ok ? Entry::READY : Entry::FAILED;
            }
            job.entry->cv.notify_all();
        }
        // anything still queued will never be compiled: release the waiters
        std::lock_guard<std::mutex> lk(m_);
        for (Job& j : queue_) {
            {
                std::lock_guard<std::mutex> g(j.entry->m);
                j.entry->state = Entry::FAILED;
                j.entry->log = "process is shutting down";
            }
            j.entry->cv.notify_all();
        }
        queue_.clear();
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::map<std::string, std::shared_ptr<Entry>> entries_;
    std::vector<Job> queue_;
    std::thread worker_;
    bool stop_ = false;
};

}  // namespace rmjit
