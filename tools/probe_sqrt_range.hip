// probe: over ALL non-negative binary32 inputs, where does  y = v_rsq(x); g = x*y; h = y/2; s = fma(fma(-g,g,x), h, g)
// differ from the correctly rounded square root?
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/probe_sqrt_range.hip -o build/probe_sqrt_range
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ float seqA(float x) {
    const float y = __builtin_amdgcn_rsqf(x), g = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
}
__global__ void k(unsigned long long* cnt, uint32_t* lo, uint32_t* hi) {
    // regions: 0 = zero, 1 = denormal, 2 = [2^-126, 2^-96), 3 = [2^-96, 2^96), 4 = [2^96, max], 5 = inf
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7F800000ull; b += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = (uint32_t)b;
        const float x = __uint_as_float(bits);
        const uint32_t want = __float_as_uint(__builtin_sqrtf(x)), got = __float_as_uint(seqA(x));
        if (want != got) {
            const int r = bits == 0 ? 0 : bits < 0x00800000u ? 1 : bits < 0x0F800000u ? 2 : bits < 0x6F800000u ? 3 : bits < 0x7F800000u ? 4 : 5;
            atomicAdd(&cnt[r], 1ull);
            atomicMin(&lo[r], bits);
            atomicMax(&hi[r], bits);
        }
    }
}
int main() {
    unsigned long long *d, h[6];
    uint32_t *lo, *hi, hl[6], hh[6];
    hipMalloc(&d, sizeof h); hipMemset(d, 0, sizeof h);
    hipMalloc(&lo, sizeof hl); hipMemset(lo, 0xFF, sizeof hl);
    hipMalloc(&hi, sizeof hh); hipMemset(hi, 0, sizeof hh);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d, lo, hi);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(hl, lo, sizeof hl, hipMemcpyDeviceToHost);
    hipMemcpy(hh, hi, sizeof hh, hipMemcpyDeviceToHost);
    const char* names[6] = {"zero", "denormal", "[2^-126,2^-96)", "[2^-96,2^96)", "[2^96,max]", "inf"};
    for (int r = 0; r < 6; r++) printf("%-16s mismatches %10llu  bits in [0x%08X, 0x%08X]\n", names[r], h[r], hl[r], hh[r]);
    return 0;
}
