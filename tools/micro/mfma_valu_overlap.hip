// micro-benchmark: do VALU instructions of one wave issue in the shadow of another wave's v_mfma_f32_16x16x4_f32 on the
// same SIMD?  A 512-thread workgroup puts two waves on every SIMD (waves w and w+4).  mode 0: both idle except waves
// 0-3 run MFMAs; mode 1: waves 4-7 run dependent-free VALU FMAs only; mode 2: both at once.  If the SIMD overlaps
// them, time(2) ~ max(time(0), time(1)); if the MFMA blocks the VALU issue port, time(2) ~ time(0) + time(1).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int KF>
__device__ __forceinline__ void inter(int iters, float x, float y, v4f &a0, v4f &a1, float &f0, float &f1, float &f2, float &f3,
                                      float &f4, float &f5, float &f6, float &f7) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            if (KF > 0) f0 = fmaf(f0, 1.0001f, 0.5f);
            if (KF > 1) f1 = fmaf(f1, 1.0001f, 0.5f);
            if (KF > 2) f2 = fmaf(f2, 1.0001f, 0.5f);
            if (KF > 3) f3 = fmaf(f3, 1.0001f, 0.5f);
            if (KF > 4) f4 = fmaf(f4, 1.0001f, 0.5f);
            if (KF > 5) f5 = fmaf(f5, 1.0001f, 0.5f);
            if (KF > 6) f6 = fmaf(f6, 1.0001f, 0.5f);
            if (KF > 7) f7 = fmaf(f7, 1.0001f, 0.5f);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            if (KF > 0) f0 = fmaf(f0, 1.0002f, 0.25f);
            if (KF > 1) f1 = fmaf(f1, 1.0002f, 0.25f);
            if (KF > 2) f2 = fmaf(f2, 1.0002f, 0.25f);
            if (KF > 3) f3 = fmaf(f3, 1.0002f, 0.25f);
            if (KF > 4) f4 = fmaf(f4, 1.0002f, 0.25f);
            if (KF > 5) f5 = fmaf(f5, 1.0002f, 0.25f);
            if (KF > 6) f6 = fmaf(f6, 1.0002f, 0.25f);
            if (KF > 7) f7 = fmaf(f7, 1.0002f, 0.25f);
        }
    }
}
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, unsigned long long *clk) {
    const int wave = threadIdx.x >> 6;
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f + 1.0f;
    float f0 = x, f1 = y, f2 = x + 1, f3 = y + 1, f4 = x + 2, f5 = y + 2, f6 = x + 3, f7 = y + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mode >= 10) {
    } else if (wave < 4) {
        if (mode != 1)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                }
            }
    } else {
        if (mode == 3) __builtin_amdgcn_s_setprio(3);      // mode 3 = mode 2 with the VALU waves at top priority
        if (mode != 0)
            for (int i = 0; i < iters; ++i) {      // 16 x 8 = 128 independent FMAs per iteration = 16 MFMA slots' worth (32 clk each)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
                    f4 = fmaf(f4, 1.0001f, 0.5f); f5 = fmaf(f5, 1.0001f, 0.5f); f6 = fmaf(f6, 1.0001f, 0.5f); f7 = fmaf(f7, 1.0001f, 0.5f);
                }
            }
    }
    if (mode >= 10 && wave < 4) {                     // modes 10+k: ONE wave per SIMD interleaves k independent FMAs after each MFMA
        switch (mode - 10) {
        case 0: inter<0>(iters, x, y, a0, a1, f0, f1, f2, f3, f4, f5, f6, f7); break;
        case 2: inter<2>(iters, x, y, a0, a1, f0, f1, f2, f3, f4, f5, f6, f7); break;
        case 4: inter<4>(iters, x, y, a0, a1, f0, f1, f2, f3, f4, f5, f6, f7); break;
        case 6: inter<6>(iters, x, y, a0, a1, f0, f1, f2, f3, f4, f5, f6, f7); break;
        default: inter<8>(iters, x, y, a0, a1, f0, f1, f2, f3, f4, f5, f6, f7); break;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c, h[8];
    hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    for (int mode = 0; mode < 4; ++mode) {
        const int iters = 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, o, iters, mode, c);
        hipEventRecord(e0);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, o, iters, mode, c);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("mode %d (%s): %.1f us/launch; MFMA wave %llu clk (%d MFMAs -> %.1f clk each), VALU wave %llu clk (%d FMAs -> %.2f clk each)\n",
               mode, mode == 0 ? "MFMA only" : mode == 1 ? "VALU only" : mode == 2 ? "both on each SIMD" : "both, VALU waves at s_setprio 3", ms * 1e3 / 20, h[0], iters * 16,
               (double)h[0] / (iters * 16), h[4], iters * 128, (double)h[4] / (iters * 128));
    }
    for (int kf = 0; kf <= 8; kf += 2) {              // intra-wave: k FMAs in the shadow of each MFMA
        const int iters = 256;
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, o, iters, 10 + kf, c);
        hipDeviceSynchronize();
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("one wave per SIMD, %d FMAs after every MFMA: %.1f clk per MFMA slot\n", kf, (double)h[0] / (iters * 16));
    }
    return 0;
}
