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
#include <sys/stat.h>
#include <unistd.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "rm_device.h"
#include "rm_units.h"

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

// A/B knobs of the generated code (environment, read when a structure is generated; defaults are the measured best)
inline int jit_knob(const char* name, int dflt) {
    const char* v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

// prune: 0 every leaf is evaluated, 1 a lattice program (min / max over bounded leaves) whose leaves sit behind a bit of the
// mask wave-level culling computes (threshold rule), 2 a program that blends whose top-level chain's units do (rm_units.h,
// rm_kernel_v5.h "Wave-level culling")
enum : int { PRUNE_NONE = 0, PRUNE_LATTICE = 1, PRUNE_BLEND = 2 };
// ... | KERNEL_WITH_STATS: the kernel keeps the per-wave counters of RM_OPT_WAVE_STATS (iterations, live lanes).  A kernel without
// them -- the default -- saves five scalar instructions per march step (rm_kernel_v5.h, RM_NO_WAVE_STATS)
enum : int { KERNEL_WITH_STATS = 0x100, PRUNE_KIND_MASK = 0xFF };

// Straight-line code for `rec`, mirroring exec_command (rm_interp.h) record by record with the value stack resolved at
// generation time: the accumulator and every spilled value become named values, no opcode decode, no loop; parameters are
// read from the LDS copy of the program at constant offsets.  One generator for both functions of a translation unit:
//   T = 1   map_scene_spec<FAST>(lp, qx, qy, qz, need, live, tiny, n_eval): the value at one position per lane
//   T = 4   map_scene_taps<FAST>(lp, cx, cy, cz, need, live, tiny, f): the four normal taps of a hit at c (wgsl:135-144) in one
//           pass -- every record applied to the four positions c + k_t eps with the same leaf functions and operators, so
//           each f[t] goes through the operations a separate evaluation would; the compiler sees that the twelve
//           coordinates take only six different values, and one loop trip replaces four
// With pruning, a UNIT (rm_units.h) is code behind one bit of `need`, which the caller computed for the whole wave:
//   lattice programs   every bounded leaf; skipped, it is +inf where it is pushed or intersected ("acc = min(acc, +inf)" and
//                      "acc = max(acc, -inf)" leave the accumulator alone)
//   blending chains    every step of the top-level chain -- the leaf that starts it (skipped: +inf, which the first unit
//                      that is needed, a restart, turns into its own leaf's value: smin(+inf, v) = v), a leaf with its Union /
//                      SmoothUnion / Subtraction / Intersection (skipped: the accumulator stays), an opaque stretch of records
// Returns false if the records do not form a valid program (cannot happen for the output of rm_decode_program) or the
// program has no units of the kind asked for.
inline bool generate_scene_code(const std::vector<RmRecord>& rec, int prune, int T, std::string* out) {
    const int count_mode = (prune && T == 1) ? jit_knob("RM_JIT_PRUNE_STATS", 0) : 0;  // 1: leaves evaluated (tools/wave_stats.py)
    const char* counted = count_mode == 1 ? " n_eval += 1u;" : "";
    // A scheduling barrier after every few leaves: left alone the compiler hoists the parameter loads of the whole
    // program to the top of the straight-line code (88 VGPRs for 16 leaves, 120-139 for 32: 3 waves per SIMD);
    // with the barriers 61-62 VGPRs whatever the length.  64-node scene at 4K 636 -> 682 Mpx/s, metric frame +2 %.
    const int sched_every = T == 1 ? (std::getenv("RM_JIT_SCHED_BARRIER") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER")) : 4)
                                   : (std::getenv("RM_JIT_SCHED_BARRIER_TAPS") ? std::atoi(std::getenv("RM_JIT_SCHED_BARRIER_TAPS")) : 2);
    const bool sub_tests = jit_knob("RM_JIT_SUB_TESTS", 1) != 0 && T == 1;  // the local test of subtracted leaves (below)
    const bool fence = T == 4 && jit_knob("RM_JIT_GUARD_FENCE", 1) != 0;    // see guard_fence (rm_kernel_v5.h)
    std::vector<RmUnit> units;
    if (prune == PRUNE_LATTICE && !rm_lattice_units(rec, &units)) return false;
    if (prune == PRUNE_BLEND && !rm_blend_units(rec, &units)) return false;
    std::vector<int> unit_at(rec.size(), -1);
    for (size_t u = 0; u < units.size(); u++) unit_at[(size_t)units[u].first] = (int)u;

    std::string s;
    char line[1024];
    s += "namespace rmk {\ntemplate <bool FAST>\n";
    if (T == 1) {
        s += "RM_DEV float map_scene_spec(LdsF lp, float qx, float qy, float qz, unsigned long long need, unsigned long long live, SqrtGuard& tiny, uint32_t& n_eval) {\n";
        if (prune) s += "    const float inf = __uint_as_float(0x7F800000u);\n";
        s += "    const float x0 = qx, y0 = qy, z0 = qz;\n";
    } else {
        s += "RM_DEV void map_scene_taps(LdsF lp, float cx, float cy, float cz, unsigned long long need, unsigned long long live, SqrtGuard& tiny, float (&f)[4]) {\n";
        if (prune) s += "    const float inf = __uint_as_float(0x7F800000u);\n";
        s += "    const float e = 0.0001f;\n";  // wgsl:136; k = (1,-1): taps (+,-,-), (-,-,+), (-,+,-), (+,+,+) (wgsl:138-141)
        s += "    const float x0_0 = cx + e, x0_1 = cx - e, x0_2 = cx - e, x0_3 = cx + e;\n";
        s += "    const float y0_0 = cy - e, y0_1 = cy - e, y0_2 = cy + e, y0_3 = cy + e;\n";
        s += "    const float z0_0 = cz - e, z0_1 = cz + e, z0_2 = cz - e, z0_3 = cz + e;\n";
    }
    // names: value w at tap t, position scope c at tap t
    auto V = [&](int w, int t) { char b[32]; if (T == 1) std::snprintf(b, sizeof b, "v%d", w); else std::snprintf(b, sizeof b, "v%d_%d", w, t); return std::string(b); };
    auto P = [&](int c, int t) {
        char b[96];
        if (T == 1) std::snprintf(b, sizeof b, "x%d, y%d, z%d", c, c, c);
        else std::snprintf(b, sizeof b, "x%d_%d, y%d_%d, z%d_%d", c, t, c, t, c, t);
        return std::string(b);
    };
    auto leaf_fn = [](uint32_t kind) -> const char* {
        return kind == RM_KIND_SPHERE ? "spec_sphere<FAST>" : kind == RM_KIND_BOX ? "spec_box<FAST>" : kind == RM_KIND_CYLINDER ? "spec_cylinder<FAST>"
             : kind == RM_KIND_PLANE ? "spec_plane" : nullptr;
    };
    auto leaf_expr = [&](size_t i, int c, int t) {
        const uint32_t kind = RM_OP_KIND(rec[i].op);
        char b[256];
        if (kind == RM_KIND_PLANE) std::snprintf(b, sizeof b, "%s(lp + %u, %s)", leaf_fn(kind), (unsigned)i * 8u, P(c, t).c_str());
        else std::snprintf(b, sizeof b, "%s(lp + %u, %s, tiny)", leaf_fn(kind), (unsigned)i * 8u, P(c, t).c_str());
        return std::string(b);
    };
    auto op_name = [](uint32_t mode) -> const char* { return mode == RM_MODE_UNION ? "vmin" : mode == RM_MODE_SUB ? "vmax_negb" : mode == RM_MODE_INTER ? "fmax_" : nullptr; };
    // w = smooth_union(record koff; a, b): the T values at once (T = 4: ONE blend-zone test for the four taps, spec_smooth_union4)
    auto emit_smooth = [&](const char* decl, int w, unsigned koff, const std::vector<std::string>& a, const std::vector<std::string>& b, const char* ind) {
        if (T == 1) {
            std::snprintf(line, sizeof line, "%s%s%s = spec_smooth_union(lp + %u, %s, %s, live);\n", ind, decl, V(w, 0).c_str(), koff, a[0].c_str(), b[0].c_str());
            s += line;
            return;
        }
        if (decl[0]) {
            std::snprintf(line, sizeof line, "%sfloat %s, %s, %s, %s;\n", ind, V(w, 0).c_str(), V(w, 1).c_str(), V(w, 2).c_str(), V(w, 3).c_str());
            s += line;
        }
        std::snprintf(line, sizeof line,
                      "%s{ const float sa[4] = {%s, %s, %s, %s}, sb[4] = {%s, %s, %s, %s}; float so[4];\n"
                      "%s  spec_smooth_union4(lp + %u, sa, sb, live, so); %s = so[0]; %s = so[1]; %s = so[2]; %s = so[3]; }\n",
                      ind, a[0].c_str(), a[1].c_str(), a[2].c_str(), a[3].c_str(), b[0].c_str(), b[1].c_str(), b[2].c_str(), b[3].c_str(),
                      ind, koff, V(w, 0).c_str(), V(w, 1).c_str(), V(w, 2).c_str(), V(w, 3).c_str());
        s += line;
    };
    auto names = [&](int w) { std::vector<std::string> n; for (int t = 0; t < T; t++) n.push_back(V(w, t)); return n; };

    std::vector<int> stack;  // value numbers; back() is the accumulator
    std::vector<int> pos;    // position numbers of the open transform scopes; back() is the current one
    pos.push_back(0);
    int nv = 0, np = 0, leaves = 0;
    auto after_leaf = [&](bool is_plane) {
        if (fence && !is_plane) s += "    guard_fence(tiny);\n";
        if (sched_every > 0 && ++leaves % sched_every == 0) s += "    __builtin_amdgcn_sched_barrier(0);\n";
    };
    // records [first, last] as they stand (no unit inside): operators on the stack, leaves evaluated
    auto emit_plain = [&](size_t first, size_t last, const char* ind) -> bool {
        for (size_t i = first; i <= last; i++) {
            const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
            const unsigned off = (unsigned)i * 8u;
            if (kind == RM_KIND_MATERIAL) return false;  // (`rec` never holds tags)
            if (kind == RM_KIND_XFORM) {  // space transformation: a new position value (push) or back to the enclosing one (pop)
                const int c = pos.back();
                if ((mode & 1u) == 0u) {
                    const int n = ++np;
                    for (int t = 0; t < T; t++) {
                        char cx[24], cy[24], cz[24], nx[24], ny[24], nz[24];
                        auto nm = [&](char* b, char a, int q) { if (T == 1) std::snprintf(b, 24, "%c%d", a, q); else std::snprintf(b, 24, "%c%d_%d", a, q, t); };
                        nm(cx, 'x', c); nm(cy, 'y', c); nm(cz, 'z', c); nm(nx, 'x', n); nm(ny, 'y', n); nm(nz, 'z', n);
                        if (mode == RM_XF_T_PUSH)
                            std::snprintf(line, sizeof line, "%sconst float %s = %s - lp[%u], %s = %s - lp[%u], %s = %s - lp[%u];\n", ind, nx, cx, off, ny, cy, off + 1u, nz, cz, off + 2u);
                        else if (mode == RM_XF_R_PUSH)
                            std::snprintf(line, sizeof line, "%sfloat %s = %s, %s = %s, %s = %s; xf_rotate_conj(lp[%u], lp[%u], lp[%u], lp[%u], %s, %s, %s);\n",
                                          ind, nx, cx, ny, cy, nz, cz, off, off + 1u, off + 2u, off + 3u, nx, ny, nz);
                        else
                            std::snprintf(line, sizeof line, "%sconst float %s = %s / lp[%u], %s = %s / lp[%u], %s = %s / lp[%u];\n", ind, nx, cx, off, ny, cy, off, nz, cz, off);
                        s += line;
                    }
                    pos.push_back(n);
                } else {
                    if (pos.size() < 2 || stack.empty()) return false;
                    pos.pop_back();
                    if (mode == RM_XF_S_POP) {
                        const int a = stack.back(); stack.pop_back();
                        const int w = nv++;
                        for (int t = 0; t < T; t++) {
                            std::snprintf(line, sizeof line, "%sconst float %s = %s * lp[%u];\n", ind, V(w, t).c_str(), V(a, t).c_str(), off);
                            s += line;
                        }
                        stack.push_back(w);
                    }
                }
                continue;
            }
            const char* op = op_name(mode);
            if (kind == RM_KIND_POP) {
                if (stack.size() < 2 || mode == RM_MODE_PUSH) return false;
                const int b = stack.back(); stack.pop_back();
                const int a = stack.back(); stack.pop_back();
                const int w = nv++;
                if (mode == RM_MODE_SMOOTH) {
                    emit_smooth("const float ", w, off, names(a), names(b), ind);
                } else {
                    if (!op) return false;
                    for (int t = 0; t < T; t++) {
                        std::snprintf(line, sizeof line, "%sconst float %s = %s(%s, %s);\n", ind, V(w, t).c_str(), op, V(a, t).c_str(), V(b, t).c_str());
                        s += line;
                    }
                }
                stack.push_back(w);
                continue;
            }
            if (!leaf_fn(kind) || mode == RM_MODE_SMOOTH || (mode != RM_MODE_PUSH && !op)) return false;  // the decoder never fuses an operator with a parameter
            int a = -1;
            if (mode != RM_MODE_PUSH) {
                if (stack.empty()) return false;
                a = stack.back(); stack.pop_back();
            }
            const int w = nv++;
            for (int t = 0; t < T; t++) {
                if (mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "%sconst float %s = %s;%s\n", ind, V(w, t).c_str(), leaf_expr(i, pos.back(), t).c_str(), counted);
                else std::snprintf(line, sizeof line, "%sconst float %s = %s(%s, %s);%s\n", ind, V(w, t).c_str(), op, V(a, t).c_str(), leaf_expr(i, pos.back(), t).c_str(), counted);
                s += line;
            }
            stack.push_back(w);
            after_leaf(kind == RM_KIND_PLANE);
        }
        return true;
    };

    // GROUPS of units: up to four consecutive units that each take the accumulator to its next value (a leaf fused with its
    // operator, "leaf; SmoothUnion") sit behind ONE more test -- is any of them needed -- on a 32-bit word of the mask that the
    // tests inside then share.  The rays of a wave are near one or two primitives, so most groups are skipped whole: two scalar
    // instructions and a branch instead of eight and four, in kernels whose scalar side is their busiest (rm_kernel_v5.h,
    // profiles/r03_ubench_scalar_issue_cycles.txt).  group_first[u]: units in the group that starts at unit u (0: none starts there).
    std::vector<int> group_first(units.size(), 0);
    if (jit_knob("RM_JIT_UNIT_GROUPS", 1) != 0) {
        auto flows = [&](size_t u) {  // the unit consumes the accumulator and leaves the next one, whatever happens inside
            const RmUnit& q = units[u];
            if (q.kind == RM_UNIT_OPAQUE || q.leaf < 0) return false;
            const uint32_t m = q.k_rec >= 0 ? (uint32_t)RM_MODE_SMOOTH : RM_OP_MODE(rec[(size_t)q.first].op);
            if (m == RM_MODE_PUSH || (m == RM_MODE_INTER && q.kind == RM_UNIT_LEAF)) return false;  // (skipped, these are +inf)
            return true;
        };
        // ... or starts a value of its own: a pushed leaf (skipped: +inf), which the units that flow behind it take on -- "leaf; leaf op"
        // pairs are what a balanced tree is made of, and the leaf that starts a chain joins the chain's first group
        auto pushes = [&](size_t u) {
            const RmUnit& q = units[u];
            if (q.kind == RM_UNIT_OPAQUE || q.leaf < 0 || q.k_rec >= 0 || q.first != q.last) return false;
            return RM_OP_MODE(rec[(size_t)q.first].op) == RM_MODE_PUSH;
        };
        for (size_t u = 0; u < units.size();) {
            if (!flows(u) && !(pushes(u) && u + 1 < units.size() && flows(u + 1) && units[u + 1].first == units[u].last + 1)) { u++; continue; }
            size_t e = u;  // the run [u, e] of consecutive units, consecutive in the records as well
            while (e + 1 < units.size() && flows(e + 1) && units[e + 1].first == units[e].last + 1) e++;
            for (size_t g = u; g <= e;) {
                size_t n = e - g + 1 < 4 ? e - g + 1 : 4;
                if (n == 4 && e - g + 1 == 5) n = 3;           // (3 + 2 rather than 4 + 1)
                if ((g >> 5) != ((g + n - 1) >> 5)) n = 32 - (g & 31);  // a group stays within one word of the mask
                if (n >= 2) group_first[g] = (int)n;
                g += n;
            }
            u = e + 1;
        }
    }
    int group_left = 0, group_value = -1;  // units still to come in the open group; the value that leaves it

    for (size_t i = 0; i < rec.size();) {
        const int ui = unit_at[i];
        if (ui < 0) {
            if (!emit_plain(i, i, "    ")) return false;
            i++;
            continue;
        }
        const RmUnit& u = units[(size_t)ui];
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        const int c = pos.back();
        if (group_left == 0 && group_first[(size_t)ui] != 0) {  // a group starts: its value is the accumulator -- or +inf, if the group starts with
                                                                // a pushed leaf -- unless something inside is needed
            const bool fresh = mode == RM_MODE_PUSH && u.k_rec < 0;
            if (!fresh && stack.empty()) return false;
            group_left = group_first[(size_t)ui];
            group_value = nv++;
            for (int t = 0; t < T; t++) {
                std::snprintf(line, sizeof line, "    float %s = %s;\n", V(group_value, t).c_str(), fresh ? "inf" : V(stack.back(), t).c_str());
                s += line;
            }
            uint32_t mask = 0u;
            for (int k = 0; k < group_left; k++) mask |= 1u << ((ui + k) & 31);
            std::snprintf(line, sizeof line, "    { const uint32_t wg = unit_word(need, %du);\n    if ((wg & 0x%xu) != 0u) {\n", ui, mask);
            s += line;
        }
        char guard[64];
        if (group_left != 0) std::snprintf(guard, sizeof guard, "unit_in_word(wg, %du)", ui);
        else std::snprintf(guard, sizeof guard, "unit_needed(need, %du)", ui);
        auto close_group = [&]() {  // after the unit's code: the last unit of a group hands its value out
            if (group_left == 0 || --group_left != 0) return;
            for (int t = 0; t < T; t++) { std::snprintf(line, sizeof line, "        %s = %s;\n", V(group_value, t).c_str(), V(stack.back(), t).c_str()); s += line; }
            s += "    } }\n";
            stack.back() = group_value;
        };
        if (u.kind == RM_UNIT_OPAQUE) {
            // an opaque stretch of the chain: from the accumulator to its next value; evaluated unless it is dead
            if (stack.empty()) return false;
            const int a = stack.back();
            const int w = nv++;
            for (int t = 0; t < T; t++) { std::snprintf(line, sizeof line, "    float %s = %s;\n", V(w, t).c_str(), V(a, t).c_str()); s += line; }
            std::snprintf(line, sizeof line, "    if (unit_needed(need, %du)) {\n", ui);
            s += line;
            const size_t depth0 = stack.size();
            if (!emit_plain((size_t)u.first, (size_t)u.last, "        ") || stack.size() != depth0) return false;
            for (int t = 0; t < T; t++) { std::snprintf(line, sizeof line, "        %s = %s;\n", V(w, t).c_str(), V(stack.back(), t).c_str()); s += line; }
            s += "    }\n";
            stack.back() = w;
            i = (size_t)u.last + 1u;
            continue;
        }
        // a unit that is one leaf: skipped, the result is +inf where the leaf would be pushed (or intersected with, in a lattice
        // program: max(acc, +inf)), else the accumulator
        const bool with_smooth = u.k_rec >= 0;  // "leaf; SmoothUnion": two records
        const uint32_t eff_mode = with_smooth ? (uint32_t)RM_MODE_SMOOTH : mode;
        int a = -1;
        if (eff_mode != RM_MODE_PUSH) {
            if (stack.empty()) return false;
            a = stack.back(); stack.pop_back();
        }
        const int w = nv++;
        const bool to_inf = eff_mode == RM_MODE_PUSH || (eff_mode == RM_MODE_INTER && u.kind == RM_UNIT_LEAF);
        for (int t = 0; t < T; t++) {
            if (to_inf) std::snprintf(line, sizeof line, "    float %s = inf;\n", V(w, t).c_str());
            else std::snprintf(line, sizeof line, "    float %s = %s;\n", V(w, t).c_str(), V(a, t).c_str());
            s += line;
        }
        std::snprintf(line, sizeof line, "    if (%s) {%s\n", guard, counted);
        s += line;
        const unsigned off = (unsigned)i * 8u;
        if (with_smooth) {
            std::vector<std::string> lv;
            for (int t = 0; t < T; t++) lv.push_back(leaf_expr(i, c, t));
            emit_smooth("", w, (unsigned)u.k_rec * 8u, names(a), lv, "        ");
        } else if (eff_mode == RM_MODE_SUB && sub_tests && (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX)) {
            // A SUBTRACTED leaf changes max(acc, -v) only where -v > acc: at a position outside it (v > 0) with acc >= 0 -- a ray
            // that is not inside anything -- never.  A test on values at hand (the squared distance, the accumulator), exact
            // without any Lipschitz argument; the leaf's square root is taken only if some live lane fails it.
            if (kind == RM_KIND_SPHERE)
                std::snprintf(line, sizeof line, "        const float a = spec_sphere_a(lp + %u, %s);\n        if (spec_sub_sphere_near(live, lp + %u, a, %s)) %s = vmax_negb(%s, spec_sphere_v<FAST>(lp + %u, a, tiny));\n",
                              off, P(c, 0).c_str(), off, V(a, 0).c_str(), V(w, 0).c_str(), V(a, 0).c_str(), off);
            else
                std::snprintf(line, sizeof line, "        const SpecBox b = spec_box_a(lp + %u, %s);\n        if (spec_sub_box_near(live, b.a, %s)) %s = vmax_negb(%s, spec_box_v<FAST>(b, tiny));\n",
                              off, P(c, 0).c_str(), V(a, 0).c_str(), V(w, 0).c_str(), V(a, 0).c_str());
            s += line;
        } else {
            const char* op = op_name(eff_mode);
            if (eff_mode != RM_MODE_PUSH && !op) return false;
            for (int t = 0; t < T; t++) {
                if (eff_mode == RM_MODE_PUSH) std::snprintf(line, sizeof line, "        %s = %s;\n", V(w, t).c_str(), leaf_expr(i, c, t).c_str());
                else std::snprintf(line, sizeof line, "        %s = %s(%s, %s);\n", V(w, t).c_str(), op, V(a, t).c_str(), leaf_expr(i, c, t).c_str());
                s += line;
            }
        }
        s += "    }\n";
        stack.push_back(w);
        after_leaf(false);
        close_group();
        i = (size_t)u.last + 1u;
    }
    if (stack.empty() || group_left != 0) return false;
    if (T == 1) {
        std::snprintf(line, sizeof line, "    return %s;\n}\n}  // namespace rmk\n", V(stack.back(), 0).c_str());
        s += line;
    } else {
        for (int t = 0; t < 4; t++) {
            std::snprintf(line, sizeof line, "    f[%d] = %s;\n", t, V(stack.back(), t).c_str());
            s += line;
        }
        s += "}\n}  // namespace rmk\n";
    }
    *out = std::move(s);
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

// capped_out (nullable): a second source of the same kernel capped at 80 vector registers (6 waves per SIMD), or left empty -- for the
// kernels that may sit a few registers above 80 and are faster with one or two of them spilled: the trees of a lattice program (the
// balanced tree of the metric scene's leaves: 84 registers, 0.40 -> 0.38 ms capped).  Whether the cap is cheap shows only in the compiled
// code: compile_best compiles the capped source first and keeps it iff it spills at most eight registers.
inline bool generate_source(const std::vector<RmRecord>& rec, const std::vector<RmRecord>& mrec, int wpt, int prune_kind, std::string* out,
                            bool* walk_generated = nullptr, bool* taps4_generated = nullptr, std::string* capped_out = nullptr) {
    const bool materials = !mrec.empty();
    const bool with_stats = (prune_kind & KERNEL_WITH_STATS) != 0;
    prune_kind &= PRUNE_KIND_MASK;
    std::string body, taps, walk;
    if (!generate_scene_code(rec, prune_kind, 1, &body)) return false;
    const bool walk_spec = materials && jit_knob("RM_JIT_MATERIAL_WALK", 1) != 0 && mrec.size() <= kMaxRecords && generate_material_walk(mrec, &walk);
    if (walk_generated) *walk_generated = walk_spec;
    const char* taps_knob = std::getenv("RM_JIT_TAPS4");  // A/B: RM_JIT_TAPS4=0 keeps the taps on map_scene_spec
    // (RM_JIT_TAPS4_SMOOTH=0: programs with a SmoothUnion keep the one-position taps)
    bool taps4 = !(taps_knob && std::atoi(taps_knob) == 0);
    if (taps4 && rm_has_blend(rec) && jit_knob("RM_JIT_TAPS4_SMOOTH", 1) == 0) taps4 = false;
    taps4 = taps4 && generate_scene_code(rec, prune_kind, 4, &taps);
    if (taps4_generated) *taps4_generated = taps4;
    std::string s;
    // hipRTC's built-in runtime header keeps the fixed-width integer types in a namespace of its own
    s += "typedef unsigned char rm_rtc_u8;\ntypedef unsigned short rm_rtc_u16;\ntypedef unsigned int rm_rtc_u32;\n"
         "typedef unsigned long long rm_rtc_u64;\ntypedef int rm_rtc_i32;\ntypedef long long rm_rtc_i64;\n"
         "#define uint8_t rm_rtc_u8\n#define uint16_t rm_rtc_u16\n#define uint32_t rm_rtc_u32\n#define uint64_t rm_rtc_u64\n"
         "#define int32_t rm_rtc_i32\n#define int64_t rm_rtc_i64\n";
    s += "#define RM_JIT_TU 1\n";
    if (prune_kind == PRUNE_LATTICE) s += "#define RM_JIT_PRUNE_ON 1\n";
    if (prune_kind == PRUNE_BLEND) s += "#define RM_JIT_BLEND_PRUNE 1\n";
    if (taps4) s += "#define RM_JIT_TAPS4 1\n";
    if (walk_spec) s += "#define RM_JIT_MATERIAL_WALK 1\n";
    if (structure_allows_bound_walk(rec)) s += "#define RM_JIT_BOUND_WALK 1\n";
    if (const char* f = std::getenv("RM_JIT_UNIT_TEST")) s += "#define RM_UNIT_TEST_FORM " + std::to_string(std::atoi(f)) + "\n";  // A/B: unit_needed
    if (!with_stats) s += "#define RM_NO_WAVE_STATS 1\n";
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
    // The kernel of a blending chain needs 86 vector registers (the mask's prefix scans on top of the four-tap function): 5 waves per
    // SIMD.  Capped at 80 it spills nine of them and is 5 % FASTER (config 3 at 4K: 3.44 -> 3.27 ms); every other kind of kernel that
    // sits above 80 loses by the same cap (balanced tree +7 %, materials +3 %, transforms +1 %: profiles/r03_refill_threshold_and_forced_occupancy.txt).
    // RM_JIT_WAVES_PER_EU (A/B): n forces n waves per SIMD for every kernel, 0 none.
    int waves = prune_kind == PRUNE_BLEND ? 6 : 0;
    // A CHAIN ("a op b op c ...": every record after the first a leaf fused with its operator) without blends or materials and
    // with the four-tap function takes 73 vector registers -- one more than 7 waves per SIMD allow -- and 22.5 KB of LDS per
    // workgroup (no partial normals, RmLaunch::wave_dwords): room for 7 workgroups on a CU.  Capped at 72 it spills one register;
    // a launch still takes only 6 workgroups per CU (rm_abi.hip launch_v5_w), so a frame drawn alone runs as before (-0.5 .. +0.8 %)
    // and the seventh slot goes to the NEXT frame's launch: two to eight frames in flight +4 .. +7 % (profiles/r03_seven_waves_per_simd_ab.txt).
    bool chain = !materials && taps4 && prune_kind != PRUNE_BLEND && !rec.empty();
    for (size_t i = 1; chain && i < rec.size(); i++) {
        const uint32_t kind = RM_OP_KIND(rec[i].op), mode = RM_OP_MODE(rec[i].op);
        chain = (kind == RM_KIND_SPHERE || kind == RM_KIND_BOX || kind == RM_KIND_CYLINDER || kind == RM_KIND_PLANE) && mode != RM_MODE_PUSH;
    }
    if (chain) waves = 7;
    const bool knob = std::getenv("RM_JIT_WAVES_PER_EU") != nullptr;
    if (knob) waves = std::atoi(std::getenv("RM_JIT_WAVES_PER_EU"));
    const bool probe = capped_out && !knob && waves == 0 && prune_kind == PRUNE_LATTICE && !materials && jit_knob("RM_JIT_PROBE_CAP", 1) != 0;
    auto kernel_text = [&](int w) {
        std::string k;
        if (w > 0) {
            std::snprintf(line, sizeof line, "__attribute__((amdgpu_waves_per_eu(%d, %d)))\n", w, w);
            k += line;
        }
        std::snprintf(line, sizeof line,
                      "extern \"C\" __global__ __launch_bounds__(%d) void %s(RmLaunch L, rmk::V5Work work, uint32_t n_tiles, "
                      "uint32_t refill_min) {\n    rmk::rm_render_v5_body<rmk::ProgLds, true, %d, false, true, %s>(L, work, n_tiles, refill_min);\n}\n",
                      64 * wpt, kernel_name(), wpt, materials ? "true" : "false");
        k += line;
        return k;
    };
    if (capped_out) *capped_out = probe ? s + kernel_text(6) : std::string();
    s += kernel_text(waves);
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
// Directory of the disk cache: RM_JIT_CACHE_DIR if set ("" or "off" switches the cache off), else `jit_cache` next to this library
// (created on first use; a directory that cannot be written simply leaves the cache cold).  The interactive host this replaces
// (an editor whose every structural edit is a new program) pays the compiler once per structure EVER, not once per process:
// the analogue of a driver's on-disk shader cache.  build() warms it for the structures of the BASELINE scenes.
inline std::string cache_dir() {
    if (const char* env = std::getenv("RM_JIT_CACHE_DIR")) return (env[0] == 0 || std::strcmp(env, "off") == 0) ? std::string() : std::string(env);
    Dl_info info;
    std::memset(&info, 0, sizeof info);
    if (!dladdr(reinterpret_cast<void*>(&cache_dir), &info) || !info.dli_fname) return std::string();
    std::string lib(info.dli_fname);
    const size_t slash = lib.rfind('/');
    if (slash == std::string::npos) return std::string();
    return lib.substr(0, slash) + "/jit_cache";
}
inline std::string cache_path(const std::string& src) {
    const std::string dir_s = cache_dir();
    const char* dir = dir_s.empty() ? nullptr : dir_s.c_str();
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
    // Disk cache (cache_dir()): a second process starts warm.
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
        {
            const size_t slash = cache_file.rfind('/');
            if (slash != std::string::npos && slash > 0) (void)::mkdir(cache_file.substr(0, slash).c_str(), 0777);  // (exists already: fine)
        }
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

// Scratch bytes per lane of `kernel` in a code object (the `private_segment_fixed_size` of its kernel descriptor, symbol
// "<kernel>.kd": 4 bytes per spilled vector register), or UINT32_MAX when the file does not parse.  ELF64, little endian.
inline uint32_t code_object_scratch_bytes(const std::vector<char>& co, const char* kernel) {
    auto rd = [&](uint64_t off, void* dst, size_t n) { if (off > co.size() || n > co.size() - off) return false; std::memcpy(dst, co.data() + off, n); return true; };
    uint64_t shoff = 0;
    uint16_t shentsize = 0, shnum = 0;
    if (co.size() < 64 || std::memcmp(co.data(), "\177ELF\2\1", 6) != 0 || !rd(0x28, &shoff, 8) || !rd(0x3A, &shentsize, 2) || !rd(0x3C, &shnum, 2) ||
        shentsize < 64) return UINT32_MAX;
    struct Sec { uint32_t type, link; uint64_t addr, offset, size; };
    auto section = [&](uint32_t i, Sec* o) {
        const uint64_t b = shoff + (uint64_t)i * shentsize;
        return i < shnum && rd(b + 4, &o->type, 4) && rd(b + 0x10, &o->addr, 8) && rd(b + 0x18, &o->offset, 8) && rd(b + 0x20, &o->size, 8) && rd(b + 0x28, &o->link, 4);
    };
    const std::string want = std::string(kernel) + ".kd";
    for (uint32_t i = 0; i < shnum; i++) {
        Sec sym, str;
        if (!section(i, &sym) || sym.type != 2u /* SHT_SYMTAB */ || !section(sym.link, &str)) continue;
        for (uint64_t k = 0; k + 24 <= sym.size; k += 24) {
            uint32_t name = 0;
            uint16_t shndx = 0;
            uint64_t value = 0;
            if (!rd(sym.offset + k, &name, 4) || !rd(sym.offset + k + 6, &shndx, 2) || !rd(sym.offset + k + 8, &value, 8)) return UINT32_MAX;
            if (name >= str.size || str.size - name < want.size() + 1) continue;
            std::string got(want.size() + 1, 0);
            if (!rd(str.offset + name, &got[0], got.size()) || got.compare(0, want.size(), want) != 0 || got[want.size()] != 0) continue;
            Sec home;
            uint32_t scratch = 0;
            if (!section(shndx, &home) || value < home.addr || !rd(value - home.addr + home.offset + 4, &scratch, 4)) return UINT32_MAX;
            return scratch;
        }
    }
    return UINT32_MAX;
}

// `src`, or -- when generate_source offered one -- the capped form of the same kernel if it spills at most eight registers (measured: the
// balanced tree with four spilled 0.403 -> 0.381 ms, the blending chain with nine 3.44 -> 3.27; a kernel that needs many more than 80 loses).
constexpr uint32_t kCapScratchBytes = 32u;
// *used (nullable): the source whose code object `code` holds.
inline bool compile_best(const std::string& src, const std::string& capped, std::vector<char>* code, std::string* log, double* ms,
                         bool* from_cache = nullptr, std::string* used = nullptr) {
    double ms_cap = 0.0;
    if (!capped.empty()) {
        bool cached = false;
        if (compile(capped, code, log, &ms_cap, &cached) && code_object_scratch_bytes(*code, kernel_name()) <= kCapScratchBytes) {
            *log += "\nkernel capped at 80 vector registers (6 waves per SIMD): " + std::to_string(code_object_scratch_bytes(*code, kernel_name())) + " bytes of scratch per lane";
            if (ms) *ms = ms_cap;
            if (from_cache) *from_cache = cached;
            if (used) *used = capped;
            return true;
        }
        code->clear();
    }
    const bool ok = compile(src, code, log, ms, from_cache);
    if (ms) *ms += ms_cap;
    if (used) *used = src;
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
    bool taps4 = false;          // ... the four-tap function (without materials it needs no partial normals in LDS: RmLaunch::wave_dwords)
    bool from_cache = false;     // the code object was read from the disk cache (compile_ms is then the time of that read)
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
        static const char* const knobs[] = {"RM_JIT_GUARD_FENCE", "RM_JIT_MATERIAL_WALK",
                                            "RM_JIT_PRIO_LONG_RAYS", "RM_JIT_PRUNE_STATS", "RM_JIT_SCHED_BARRIER", "RM_JIT_SCHED_BARRIER_TAPS",
                                            "RM_JIT_SUB_TESTS", "RM_JIT_TAPS4", "RM_JIT_TAPS4_SMOOTH", "RM_JIT_WAVES_PER_EU", "RM_JIT_UNIT_TEST", "RM_JIT_UNIT_GROUPS", "RM_JIT_PROBE_CAP"};
        std::string knob_key;
        for (const char* name : knobs)
            if (const char* v = std::getenv(name)) knob_key += std::string("|") + name + "=" + v;
        const std::string key = std::to_string(wpt) + ((prune & PRUNE_KIND_MASK) == PRUNE_LATTICE ? "p" : (prune & PRUNE_KIND_MASK) == PRUNE_BLEND ? "b" : "") +
                                ((prune & KERNEL_WITH_STATS) ? "s" : "") + ":" + structure_key(rec) +
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
        if (!generate_source(rec, mrec, wpt, prune, &job.source, &e->material_walk, &e->taps4, &job.capped)) {
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
    struct Job { std::shared_ptr<Entry> entry; std::string source, capped; };
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
            std::string used;
            const bool ok = compile_best(job.source, job.capped, &code, &log, &ms, &from_cache, &used);
            {
                std::lock_guard<std::mutex> g(job.entry->m);
                job.entry->from_cache = ok && from_cache;
                if (ok && from_cache) job.entry->cached_source = std::move(used);
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
