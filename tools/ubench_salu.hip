// ubench_salu.hip -- what SCALAR instructions cost a SIMD on gfx950, alone and next to vector instructions of other waves; same
// frame as ubench_valu.hip (clock from s_memtime / s_memrealtime, cycles per instruction per SIMD from the kernel duration).
// A CU has ONE scalar unit for its four SIMDs (MI355X_MICROARCH.md): does scalar work hide behind the vector work of the
// other waves, or does it take issue time of its own?  Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_salu tools/ubench_salu.hip
//
// Round 1 priced the march kernel's instruction mix with wall-clock rates converted at an assumed 2.4 GHz
// ("v_add 2.77 cycles", against the 2 cycles MI355X_MICROARCH.md gives for a wave64 instruction on a SIMD-32).
// A chip-wide loop of back-to-back vector instructions is exactly where DVFS lowers the clock, so this version
// stamps every wave with s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop:
//     clock                                = median over waves of dt_cycles / dt_realtime x 100 MHz
//     cycles per wave-instruction per SIMD = kernel duration (HIP events) x clock / (waves per SIMD x instructions per wave)
// (a wave's own dt_cycles is NOT the kernel's: the SIMD issues oldest-first, so the eight waves of a SIMD finish one
// after the other and each is resident for about a third of the kernel; the first version of this file divided the
// median wave lifetime by the instruction count and reported 0.7 "cycles" for v_add)
// Each wave runs 8 independent dependency chains; a 256-thread block places one wave on each SIMD of a CU, and
// `w` blocks per CU give w waves per SIMD.  Everything stays in registers (the LDS variant reads one broadcast dword).
// Run it under `rocprofv3 --pmc GRBM_GUI_ACTIVE` as well: GRBM_GUI_ACTIVE / 8 / duration is the same clock seen from outside.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

enum Op { S_ADD, S_AND64, S_CSELECT, S_FF1, S_BCNT, V_READLANE, V_READFIRSTLANE, V_ADD, MIX_S_V, MIX_2S_V, MIX_3S_V, MIX_S_2V, BRANCH_NOT_TAKEN, CMP_BRANCH, S_LOAD_FREE, N_OPS };

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    float b = seed * 1.0001f;
    unsigned s[8];
    for (int i = 0; i < 8; i++) s[i] = __builtin_amdgcn_readfirstlane(__float_as_uint(seed) + i);
    unsigned long long m[8];
    for (int i = 0; i < 8; i++) m[i] = ((unsigned long long)s[i] << 20) | 0x10001ull;
    const unsigned sb = (unsigned)iters * 3u + 1u;  // (a kernel argument: scalar)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define SADD(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(sb) : "scc");
#define VADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE(i)                                                                                                          \
    if (OP == S_ADD) SADD(i)                                                                                            \
    if (OP == S_AND64) asm volatile("s_and_b64 %0, %0, %1" : "+s"(m[i]) : "s"(m[(i + 1) & 7]) : "scc");                  \
    if (OP == S_CSELECT) asm volatile("s_cselect_b32 %0, %0, %1" : "+s"(s[i]) : "s"(sb));                                \
    if (OP == S_FF1) asm volatile("s_ff1_i32_b64 %0, %1" : "=s"(s[i]) : "s"(m[i]));                                      \
    if (OP == S_BCNT) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(s[i]) : "s"(m[i]) : "scc");                           \
    if (OP == V_READLANE) asm volatile("v_readlane_b32 %0, %1, 7" : "=s"(s[i]) : "v"(a[i]));                             \
    if (OP == V_READFIRSTLANE) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s[i]) : "v"(a[i]));                      \
    if (OP == V_ADD) VADD(i)                                                                                            \
    if (OP == MIX_S_V) { SADD(i) VADD(i) }                                                                              \
    if (OP == MIX_2S_V) { SADD(i) SADD((i + 1) & 7) VADD(i) }                                                           \
    if (OP == MIX_3S_V) { SADD(i) SADD((i + 1) & 7) SADD((i + 2) & 7) VADD(i) }                                         \
    if (OP == MIX_S_2V) { SADD(i) VADD(i) VADD((i + 1) & 7) }                                                           \
    if (OP == BRANCH_NOT_TAKEN) asm volatile("s_cmp_eq_u32 %0, 0x12345\n\ts_cbranch_scc1 1f\n1:" : : "s"(s[i]) : "scc");  \
    if (OP == CMP_BRANCH) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\ts_cbranch_vccz 2f\n2:" : : "v"(a[i]), "v"(b) : "vcc", "scc");
            REP8(ONE)
#undef ONE
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float sum = 0;
    for (int i = 0; i < 8; i++) sum += a[i] + (float)s[i] + (float)(unsigned)m[i];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* st = stamps + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6));
        st[0] = t0; st[1] = t1; st[2] = r0; st[3] = r1;
    }
}

struct Result { double cycles_per_inst, clock_ghz, wall_ns_per_inst; };

template <int OP>
int run(const char* name, unsigned long long* d_st, float* d_out, int n_cu, int w, int per_slot = 1, const char* note = "") {
    const int blocks = n_cu * w, iters = 4000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_out, 200, 1.0f);  // warm
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_st, d_out, iters, 1.0f);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> st((size_t)blocks * 16);
    CHK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> cyc, clk;
    for (int wv = 0; wv < blocks * 4; wv++) {
        const double dc = (double)(st[4 * wv + 1] - st[4 * wv]), dr = (double)(st[4 * wv + 3] - st[4 * wv + 2]);
        cyc.push_back(dc);
        if (dr > 0) clk.push_back(dc / dr * 0.1);  // cycles per 10 ns -> GHz
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double n_inst = (double)iters * 64 * per_slot;
    const double med_cyc = cyc[cyc.size() / 2], med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double wall_ns = ms * 1e6 / (n_inst * w);
    printf("%-30s waves/SIMD %d  %6.2f cycles per instruction per SIMD  clock %.2f GHz  (wall %.3f ns per instr per SIMD; a wave is resident %.0f %% of the kernel)%s\n",
           name, w, wall_ns * med_clk, med_clk, wall_ns, 100.0 * med_cyc / (ms * 1e6 * med_clk), note);
    return 0;
}

int main(int, char**) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_st;
    float* d_out;
    CHK(hipMalloc(&d_st, (size_t)n_cu * 8 * 16 * 8));
    CHK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
    printf("# %s, %d CUs; cycles = s_memtime ticks, clock = s_memtime / s_memrealtime (100 MHz); median over waves\n", prop.gcnArchName, n_cu);
    for (int w : {8, 6, 2, 1}) {
        run<V_ADD>("v_add_f32", d_st, d_out, n_cu, w);
        run<S_ADD>("s_add_u32", d_st, d_out, n_cu, w);
        run<S_AND64>("s_and_b64", d_st, d_out, n_cu, w);
        run<S_CSELECT>("s_cselect_b32", d_st, d_out, n_cu, w);
        run<S_FF1>("s_ff1_i32_b64", d_st, d_out, n_cu, w);
        run<S_BCNT>("s_bcnt1_i32_b64", d_st, d_out, n_cu, w);
        run<V_READLANE>("v_readlane_b32", d_st, d_out, n_cu, w);
        run<V_READFIRSTLANE>("v_readfirstlane_b32", d_st, d_out, n_cu, w);
        run<MIX_S_V>("mix: s_add, v_add", d_st, d_out, n_cu, w, 2, "  [per instruction of the mix]");
        run<MIX_2S_V>("mix: 2 s_add, v_add", d_st, d_out, n_cu, w, 3, "  [per instruction of the mix]");
        run<MIX_3S_V>("mix: 3 s_add, v_add", d_st, d_out, n_cu, w, 4, "  [per instruction of the mix]");
        run<MIX_S_2V>("mix: s_add, 2 v_add", d_st, d_out, n_cu, w, 3, "  [per instruction of the mix]");
        run<BRANCH_NOT_TAKEN>("s_cmp + s_cbranch (not taken)", d_st, d_out, n_cu, w, 2, "  [per instruction of the pair]");
        run<CMP_BRANCH>("v_cmp + s_and + s_cbranch_vccz (taken, to the next instruction)", d_st, d_out, n_cu, w, 3, "  [per instruction of the three]");
    }
    return 0;
}
