// probe: does v_sqrt_f32 honour denormal inputs/outputs with the kernel's FP mode (denorm preserve)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
__global__ void k(const float* in, float* out, int n) {
    int i = threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_sqrtf(in[i]);
}
int main() {
    float h[8] = {1e-40f, 1.4e-45f, 1e-38f, 2.0e-38f, 1e-30f, 0.0f, 4.0f, 3.0e-39f};
    float *d, *o, r[8];
    hipMalloc(&d, 32); hipMalloc(&o, 32);
    hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 8);
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) printf("x=%g  v_sqrt=%g  sqrtf=%g\n", h[i], r[i], sqrtf(h[i]));
    return 0;
}
