// ubench_valu.hip -- what one VALU instruction class costs a SIMD on gfx950 (wave64), in CYCLES at a MEASURED clock.
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

enum Op { FMA, MUL, ADD, SUB_SGPR, SUB_ABS, MAX, MIN, MAX3, MAX_NEG, CMP_VCC, CMP_SGPR, CNDMASK, RSQ, RCP, SQRT, MOV, ADD_U32, MIN3_U32, XOR,
          MIX_MAX_ADD, MIX_RSQ_3ADD, MIX_CMP_ADD, LDS_ADD, ADD_LITERAL, PK_ADD, PK_MUL, PK_FMA, PK_ADD_BCAST, PK_ADD_NEG, MIX_PK_ADD, MIX_PK_MAX, N_OPS };
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long* stamps, float* out, int iters, float seed) {
    __shared__ float lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = seed + threadIdx.x;
    __syncthreads();
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    float b = seed * 1.0001f, c = seed * 0.5f;
    f2 a2[8], b2 = {b, c}, c2 = {c, b};  // packed f32 (VOP3P): two values per lane in an even-aligned register pair
    for (int i = 0; i < 8; i++) a2[i] = f2{seed + threadIdx.x * 0.001f + i, seed - i};
    const float sb = __builtin_amdgcn_readfirstlane(__float_as_uint(b)) ? b : c;  // stays a VGPR; the "s" operand below is forced
    (void)sb;
    const __attribute__((address_space(3))) float* lp = (const __attribute__((address_space(3))) float*)lds;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define ONE(i)                                                                                                          \
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                                \
    if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    if (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    if (OP == SUB_SGPR) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "s"(seed));                                    \
    if (OP == SUB_ABS) asm volatile("v_sub_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));                                      \
    if (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    if (OP == MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    if (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                              \
    if (OP == MAX_NEG) asm volatile("v_max_f32 %0, %0, -%1" : "+v"(a[i]) : "v"(b));                                       \
    if (OP == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                            \
    if (OP == CMP_SGPR) { unsigned long long m; asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(a[i]), "v"(b)); }  \
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));                               \
    if (OP == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));                                                         \
    if (OP == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));                                                         \
    if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                                       \
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));                                                \
    if (OP == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                        \
    if (OP == MIN3_U32) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                          \
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                            \
    if (OP == MIX_MAX_ADD) { if ((i & 1) == 0) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                \
                             else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }                           \
    if (OP == MIX_RSQ_3ADD) { if ((i & 3) == 0) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));                            \
                              else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }                          \
    if (OP == MIX_CMP_ADD) { if ((i & 1) == 0) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");    \
                             else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }                           \
    if (OP == LDS_ADD) { float p = lp[(i + r) & 63]; asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(p)); }        \
    if (OP == ADD_LITERAL) asm volatile("v_add_f32 %0, 0x3f8ccccd, %0" : "+v"(a[i]));                                      \
    if (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));                                    \
    if (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));                                    \
    if (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a2[i]) : "v"(b2), "v"(c2));                       \
    if (OP == PK_ADD_BCAST) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a2[i]) : "v"(b2));              \
    if (OP == PK_ADD_NEG) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(a2[i]) : "v"(b2));      \
    if (OP == MIX_PK_ADD) { if ((i & 1) == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));            \
                            else asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }                            \
    if (OP == MIX_PK_MAX) { if ((i & 1) == 0) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a2[i]) : "v"(b2));            \
                            else asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); }
            REP8(ONE)
#undef ONE
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + a2[i].x + a2[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* st = stamps + 4ull * (blockIdx.x * 4u + (threadIdx.x >> 6));
        st[0] = t0; st[1] = t1; st[2] = r0; st[3] = r1;
    }
}

struct Result { double cycles_per_inst, clock_ghz, wall_ns_per_inst; };

template <int OP>
int run(const char* name, unsigned long long* d_st, float* d_out, int n_cu, int w, const char* note = "") {
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
    const double n_inst = (double)iters * 64;
    const double med_cyc = cyc[cyc.size() / 2], med_clk = clk.empty() ? 0.0 : clk[clk.size() / 2];
    const double wall_ns = ms * 1e6 / (n_inst * w);
    printf("%-30s waves/SIMD %d  %6.2f cycles per wave-instr per SIMD  clock %.2f GHz  (wall %.3f ns per instr per SIMD; a wave is resident %.0f %% of the kernel)%s\n",
           name, w, wall_ns * med_clk, med_clk, wall_ns, 100.0 * med_cyc / (ms * 1e6 * med_clk), note);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    unsigned long long* d_st;
    float* d_out;
    CHK(hipMalloc(&d_st, (size_t)n_cu * 8 * 16 * 8));
    CHK(hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4));
    printf("# %s, %d CUs; cycles = s_memtime ticks, clock = s_memtime / s_memrealtime (100 MHz); median over waves\n", prop.gcnArchName, n_cu);
    const bool quick = argc > 1 && !strcmp(argv[1], "--quick");
    if (argc > 1 && !strcmp(argv[1], "--packed")) {  // packed f32 (two values per lane per instruction) next to the plain forms
        for (int w : {8, 5, 2}) {
            run<ADD>("v_add_f32", d_st, d_out, n_cu, w);
            run<FMA>("v_fma_f32", d_st, d_out, n_cu, w);
            run<PK_ADD>("v_pk_add_f32", d_st, d_out, n_cu, w, "  [two adds per lane]");
            run<PK_MUL>("v_pk_mul_f32", d_st, d_out, n_cu, w, "  [two multiplies per lane]");
            run<PK_FMA>("v_pk_fma_f32", d_st, d_out, n_cu, w, "  [two fmas per lane]");
            run<PK_ADD_BCAST>("v_pk_add_f32 op_sel_hi:[1,0]", d_st, d_out, n_cu, w, "  [second operand: its low half for both]");
            run<PK_ADD_NEG>("v_pk_add_f32 neg (a - b)", d_st, d_out, n_cu, w);
            run<MIX_PK_ADD>("mix: v_pk_add, v_add alternating", d_st, d_out, n_cu, w, "  [per instruction of the mix]");
            run<MIX_PK_MAX>("mix: v_pk_mul, v_max alternating", d_st, d_out, n_cu, w, "  [per instruction of the mix]");
        }
        return 0;
    }
    for (int w : {8, 5, 4, 2, 1}) run<FMA>("v_fma_f32", d_st, d_out, n_cu, w);
    for (int w : {8, 5}) {
        run<ADD>("v_add_f32", d_st, d_out, n_cu, w);
        run<MUL>("v_mul_f32", d_st, d_out, n_cu, w);
        run<SUB_SGPR>("v_sub_f32 (sgpr operand)", d_st, d_out, n_cu, w);
        run<SUB_ABS>("v_sub_f32 |abs| modifier", d_st, d_out, n_cu, w);
        run<ADD_LITERAL>("v_add_f32 (32-bit literal)", d_st, d_out, n_cu, w);
        run<MAX>("v_max_f32", d_st, d_out, n_cu, w);
        run<MIN>("v_min_f32", d_st, d_out, n_cu, w);
        run<MAX_NEG>("v_max_f32 (neg modifier)", d_st, d_out, n_cu, w);
        run<MAX3>("v_max3_f32", d_st, d_out, n_cu, w);
        run<CMP_VCC>("v_cmp_lt_f32 -> vcc", d_st, d_out, n_cu, w);
        run<CMP_SGPR>("v_cmp_lt_f32 -> sgpr pair", d_st, d_out, n_cu, w);
        run<CNDMASK>("v_cndmask_b32 (vcc)", d_st, d_out, n_cu, w);
        run<MOV>("v_mov_b32", d_st, d_out, n_cu, w);
        run<ADD_U32>("v_add_u32", d_st, d_out, n_cu, w);
        run<MIN3_U32>("v_min3_u32", d_st, d_out, n_cu, w);
        run<XOR>("v_xor_b32", d_st, d_out, n_cu, w);
        run<RSQ>("v_rsq_f32", d_st, d_out, n_cu, w);
        run<RCP>("v_rcp_f32", d_st, d_out, n_cu, w);
        run<SQRT>("v_sqrt_f32", d_st, d_out, n_cu, w);
        run<MIX_MAX_ADD>("mix: v_max, v_add alternating", d_st, d_out, n_cu, w, "  [per instruction of the mix]");
        run<MIX_CMP_ADD>("mix: v_cmp, v_add alternating", d_st, d_out, n_cu, w, "  [per instruction of the mix]");
        run<MIX_RSQ_3ADD>("mix: v_rsq + 3 v_add", d_st, d_out, n_cu, w, "  [per instruction of the mix]");
        run<LDS_ADD>("ds_read_b32 (broadcast) + v_add", d_st, d_out, n_cu, w, "  [per v_add; one LDS read each]");
        if (quick) break;
    }
    return 0;
}
