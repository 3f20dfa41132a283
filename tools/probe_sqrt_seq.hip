// probe: which short instruction sequences give the correctly rounded binary32 square root for EVERY input in
// [2^-96, 2^96]?  (v_sqrt_f32 / v_rsq_f32 are 1-ulp approximations.)
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probe_sqrt_seq.hip -o build/probe_sqrt_seq
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NV 6
__device__ __forceinline__ float variant(int v, float x) {
    switch (v) {
    case 0: {  // A: rsq, one Newton step in FMA form
        const float y = __builtin_amdgcn_rsqf(x), g = x * y, h = 0.5f * y;
        return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
    }
    case 1: {  // B: A + a second step
        const float y = __builtin_amdgcn_rsqf(x), g = x * y, h = 0.5f * y;
        const float s = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
        return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
    }
    case 2: {  // C: v_sqrt start + rsq-based step
        const float s0 = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rsqf(x);
        return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
    }
    case 3: {  // D: v_sqrt start + rcp-based step
        const float s0 = __builtin_amdgcn_sqrtf(x), h = 0.5f * __builtin_amdgcn_rcpf(s0);
        return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
    }
    case 4: {  // E: v_sqrt + upward correction only
        const float s = __builtin_amdgcn_sqrtf(x);
        const float su = __uint_as_float(__float_as_uint(s) + 1u);
        return __builtin_fmaf(-su, s, x) > 0.0f ? su : s;
    }
    default: {  // F: the current sqrt_rn_fast (control: must be 0)
        float s = __builtin_amdgcn_sqrtf(x);
        const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
        const float vp = __builtin_fmaf(-sd, s, x), vs = __builtin_fmaf(-su, s, x);
        s = vp <= 0.0f ? sd : s;
        return vs > 0.0f ? su : s;
    }
    }
}
__global__ void k(unsigned long long* bad, uint32_t* first) {
    const uint64_t lo = 0x0F800000ull, hi = 0x6F800000ull;  // [2^-96, 2^96)
    unsigned long long loc[NV] = {};
    for (uint64_t b = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const uint32_t r = __float_as_uint(__builtin_sqrtf(x));
#pragma unroll
        for (int v = 0; v < NV; v++) {
            if (__float_as_uint(variant(v, x)) != r) {
                if (loc[v]++ == 0) atomicMin(&first[v], (uint32_t)b);
            }
        }
    }
    for (int v = 0; v < NV; v++) if (loc[v]) atomicAdd(&bad[v], loc[v]);
}
int main() {
    unsigned long long *d, h[NV];
    uint32_t *f, hf[NV];
    hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
    hipMalloc(&f, sizeof hf); hipMemset(f, 0xFF, sizeof hf);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, f);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hf, f, sizeof hf, hipMemcpyDeviceToHost);
    const char* names[NV] = {"A rsq + 1 step", "B rsq + 2 steps", "C sqrt + rsq step", "D sqrt + rcp step", "E sqrt + up-correction only", "F sqrt_rn_fast (control)"};
    for (int v = 0; v < NV; v++) printf("%-28s mismatches %llu  (smallest bad input bits 0x%08X)\n", names[v], h[v], hf[v]);
    return 0;
}
