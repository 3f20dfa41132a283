// ubench_valu.hip -- per-instruction VALU issue throughput on gfx950 (wave64), to price the
// instruction mix of the ray-marching interpreter.  Each kernel runs 8 independent dependency
// chains per wave, 32 waves per CU (8 per SIMD), everything in registers.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define CHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(r_), __LINE__); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    for (int i = 0; i < 8; i++) a[i] = seed + threadIdx.x * 0.001f + i;
    float b = seed * 1.0001f, c = seed * 0.5f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define ONE(i)                                                                                            \
    if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                    \
    if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
    if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
    if (OP == 3) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));                                            \
    if (OP == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
    if (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                   \
    if (OP == 6) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc"); \
    if (OP == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                                \
    if (OP == 8) asm volatile("v_sub_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(b));                              \
    if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");               \
    if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");                   \
    if (OP == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));                                   \
    if (OP == 12) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));                  \
    if (OP == 13) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "s"(b));                               \
    if (OP == 14) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));                               \
    if (OP == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// packed: 4 chains of float2
__global__ __launch_bounds__(256) void k_pk(float* out, int iters, float seed, int op) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; i++) a[i] = f2{seed + threadIdx.x * 0.001f + i, seed + i};
    f2 b{seed * 1.0001f, seed}, c{seed * 0.5f, seed};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#define ONE(i)                                                                                  \
    if (op == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));       \
    else if (op == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));              \
    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            REP8(ONE)
#undef ONE
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
int run(const char* name, float* d, int instr_per_one, int waves_per_simd) {
    int cus = 256;
    int blocks = cus * waves_per_simd;  // 256-thread blocks = 4 waves -> one per SIMD each
    const int iters = 2000;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double instr_per_wave = (double)iters * 64 * instr_per_one;
    double per_simd = instr_per_wave * waves_per_simd;   // wave-instructions per SIMD
    double ns_per_instr = ms * 1e6 / per_simd;
    printf("%-28s waves/SIMD %d  %.3f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)  chip %.1f G wave-instr/s\n",
           name, waves_per_simd, ns_per_instr, ns_per_instr * 2.4, 1024.0 / ns_per_instr);
    return 0;
}

int main() {
    float* d;
    CHK(hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4));
    for (int w : {8, 4, 2, 1}) {
        run<0>("v_fma_f32", d, 1, w);
    }
    run<1>("v_mul_f32", d, 1, 8);
    run<2>("v_add_f32", d, 1, 8);
    run<3>("v_sqrt_f32", d, 1, 8);
    run<15>("v_rcp_f32", d, 1, 8);
    run<4>("v_max_f32", d, 1, 8);
    run<5>("v_max3_f32", d, 1, 8);
    run<6>("v_cmp+v_cndmask pair", d, 2, 8);
    run<10>("v_cmp_lt_f32", d, 1, 8);
    run<9>("v_cndmask_b32", d, 1, 8);
    run<7>("v_add_u32", d, 1, 8);
    run<8>("v_sub_f32 |abs|", d, 1, 8);
    run<11>("v_mov_b32", d, 1, 8);
    run<12>("v_min3_u32", d, 1, 8);
    run<13>("v_sub_f32 sgpr operand", d, 1, 8);
    run<14>("v_xor_b32", d, 1, 8);
    // packed
    for (int op = 0; op < 3; op++) {
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        const int iters = 2000, w = 8;
        hipLaunchKernelGGL(k_pk, dim3(256 * w), dim3(256), 0, 0, d, 10, 1.0f, op);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_pk, dim3(256 * w), dim3(256), 0, 0, d, iters, 1.0f, op);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        double ns = ms * 1e6 / ((double)iters * 64 * w);
        printf("%-28s waves/SIMD %d  %.3f ns per wave-instr per SIMD  (= %.2f cycles at 2.4 GHz)\n",
               op == 0 ? "v_pk_fma_f32" : op == 1 ? "v_pk_mul_f32" : "v_pk_add_f32", w, ns, ns * 2.4);
    }
    return 0;
}
