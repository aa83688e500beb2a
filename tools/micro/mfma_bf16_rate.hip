// micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 (gfx950) next to v_mfma_f32_16x16x4_f32, and whether VALU
// work of a second wave on the same SIMD proceeds in its shadow.  Groundwork for carrying the f32 dense layers on the
// bf16 pipe with exact 3-way operand splits (6 bf16 MFMAs per 16x16x32 block instead of 8 f32 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, unsigned long long *clk) {
    const int wave = threadIdx.x >> 6;
    v4f a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    v8bf x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(threadIdx.x * 1e-3f + i); y[i] = (__bf16)(blockIdx.x * 1e-3f + 1.0f + i); }
    float f0 = threadIdx.x, f1 = 1, f2 = 2, f3 = 3, f4 = 4, f5 = 5, f6 = 6, f7 = 7;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        if (mode != 1)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
                }
            }
    } else if (mode != 0) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                f0 = fmaf(f0, 1.0001f, 0.5f); f1 = fmaf(f1, 1.0001f, 0.5f); f2 = fmaf(f2, 1.0001f, 0.5f); f3 = fmaf(f3, 1.0001f, 0.5f);
                f4 = fmaf(f4, 1.0001f, 0.5f); f5 = fmaf(f5, 1.0001f, 0.5f); f6 = fmaf(f6, 1.0001f, 0.5f); f7 = fmaf(f7, 1.0001f, 0.5f);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) clk[wave] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c, h[8];
    hipMalloc(&o, 256 * 512 * 4); hipMalloc(&c, 64);
    const int iters = 256;
    for (int mode = 0; mode < 3; ++mode) {
        hipMemset(c, 0, 64);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, o, iters, mode, c);
        hipDeviceSynchronize();
        hipMemcpy(h, c, 64, hipMemcpyDeviceToHost);
        printf("mode %d (%s): bf16 MFMA wave %llu clk (%.1f clk per 16x16x32 MFMA), VALU wave %llu clk\n", mode,
               mode == 0 ? "bf16 MFMA only" : mode == 1 ? "VALU only" : "both on each SIMD", h[0], (double)h[0] / (iters * 16), h[4]);
    }
    return 0;
}
