// micro-benchmark: sustained v_mfma_f32_16x16x4_f32 rate and the clock a short kernel actually runs at
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *clk) {
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.0f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1];
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
int main() {
    float *o; unsigned long long *c, h[2];
    hipMalloc(&o, 2048 * 256 * 4); hipMalloc(&c, 16);
    for (int blocks : {256, 512, 1024}) for (int iters : {16, 64, 1024}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, iters, c);
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, o, iters, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
        double us = ms * 1e3 / 20, mf = (double)iters * 16;            // MFMAs per wave
        double waves_per_simd = blocks * 4.0 / 1024.0;
        printf("blocks %4d iters %4d: %.1f us/launch; per-SIMD MFMA cycles ideal %.0f; in-kernel: %llu clk for %0.f MFMA/wave (%.1f clk each), clock %.2f GHz; TFLOPs %.1f\n",
               blocks, iters, us, mf * 32 * waves_per_simd, h[0], mf, (double)h[0] / mf, (double)h[0] / ((double)h[1] * 10.0) , blocks * 4.0 * mf * 2048 / (us * 1e-6) / 1e12);
    }
    return 0;
}
