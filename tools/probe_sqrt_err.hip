// probe: error distribution of v_sqrt_f32 against the correctly rounded square root, over EVERY positive
// normal binary32 input.  Decides which of sqrt_rn_fast's two corrections the hardware can actually need.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probe_sqrt_err.hip -o build/probe_sqrt_err
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(unsigned long long* hist) {
    // hist[0..4]: diff <= -2, -1, 0, +1, >= +2 (ulps, v_sqrt minus correctly rounded)
    __shared__ unsigned long long h[5];
    if (threadIdx.x < 5) h[threadIdx.x] = 0;
    __syncthreads();
    unsigned long long loc[5] = {0, 0, 0, 0, 0};
    const uint64_t first = 0x00800000ull, last = 0x7F800000ull;  // [min normal, +inf)
    for (uint64_t b = first + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < last; b += (uint64_t)gridDim.x * blockDim.x) {
        const float x = __uint_as_float((uint32_t)b);
        const float s = __builtin_amdgcn_sqrtf(x);
        const float r = __builtin_sqrtf(x);  // correctly rounded (default -fno-fast-math lowering)
        const int d = (int)__float_as_uint(s) - (int)__float_as_uint(r);
        loc[d <= -2 ? 0 : d >= 2 ? 4 : d + 2]++;
    }
    for (int i = 0; i < 5; i++) atomicAdd(&h[i], loc[i]);
    __syncthreads();
    if (threadIdx.x < 5) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
int main() {
    unsigned long long *d, h[5];
    hipMalloc(&d, 40);
    hipMemset(d, 0, 40);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
    hipMemcpy(h, d, 40, hipMemcpyDeviceToHost);
    printf("v_sqrt_f32 - correctly rounded sqrt over all positive normal inputs (ulps):\n");
    printf("  <= -2: %llu\n  -1: %llu\n   0: %llu\n  +1: %llu\n  >= +2: %llu\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
