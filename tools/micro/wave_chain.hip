// micro-benchmark of the building blocks of the wave-owned policy forward (csrc/cm_policy_w_dev.h), ONE wave per SIMD:
// cycles (s_memtime) of  (a) back-to-back v_mfma_f32_16x16x32_f16,  (b) the stage-wise epilogue (join, tanh, split) of 8 values,
// (c) whole layers 128 -> 64 and 64 -> 128 as the kernel runs them (fragments from LDS, one layer ahead), chained.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -I../../com-marl_amd/csrc wave_chain.hip -o wave_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "cm_policy_w_dev.h"
using namespace cm;
using namespace cm::mw;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mfma(unsigned long long *clk, float *sink, int iters) {
    const int lane = threadIdx.x & 63;
    v8h a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (h16)(0.01f * (lane + e)); b[e] = (h16)(0.02f * (lane - e)); }
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4f){ 0.f, 0.f, 0.f, 0.f };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_epi(unsigned long long *clk, float *sink, int iters, int mode) {
    const int lane = threadIdx.x & 63;
    float v[8];
    for (int e = 0; e < 8; ++e) v[e] = 0.01f * (lane + e);
    float accum = 0.0f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = fmaf(v[e], 0.000244140625f, accum);
        if (mode & 1) tanh_stage<8>(w);
        h16 h[8], l[8];
        if (mode & 2) {
            split_stage<8>(w, h, l);
#pragma unroll
            for (int e = 0; e < 8; ++e) accum += (float)h[e] + (float)l[e];          // dependency to the next iteration
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) accum += w[e];
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * 256 + threadIdx.x] = accum;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

// two chained layers 128 -> 64 -> 128 (tanh), fragments from LDS (random f16), repeated
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_layers(unsigned long long *clk, float *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    uint4 *WL = reinterpret_cast<uint4 *>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int NA = frag_u4(128, 64), NB = frag_u4(64, 128);
    for (int i = tid; i < NA + NB; i += 256) { const unsigned x = 0x1c001c00u + (i * 2654435761u & 0x03ff03ffu); WL[i] = make_uint4(x, x ^ 0x80000000u, x + 7u, x ^ 0x00008000u); }
    float *BL = reinterpret_cast<float *>(WL + NA + NB);
    for (int i = tid; i < 256; i += 256) BL[i] = 0.01f * i;
    __syncthreads();
    Act<4> x;
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) { x.hi[q][e] = (h16)(0.01f * (lane % 7 + e)); x.lo[q][e] = (h16)0.5f; }
    Frags<4, 4> fa; Frags<2, 8> fb;
    fa.fetch(WL, lane);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        fb.fetch(WL + NA, lane);
        Act<2> y;
        dense_act<4, 4, true, true>(fa, BL, x, y, nullptr, lane);
        fa.fetch(WL, lane);
        dense_act<2, 8, true, true>(fb, BL + 64, y, x, nullptr, lane);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int q = 0; q < 4; ++q) s += (float)x.hi[q][0] + (float)x.lo[q][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

int main() {
    unsigned long long *d; float *sink;
    hipMalloc(&d, 8); hipMalloc(&sink, 256 * 256 * 4);
    unsigned long long h;
    const int iters = 200;
    hipFuncSetAttribute((const void *)k_layers, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int blocks : { 1, 256 }) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, d, sink, iters);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("blocks %3d  mfma 16x16x32 f16, 8 accumulators round robin : %.1f clk per MFMA\n", blocks, (double)h / (iters * 8));
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_epi, dim3(blocks), dim3(256), 0, 0, d, sink, iters, mode);
            hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
            printf("blocks %3d  epilogue of 8 values (join%s%s)            : %.1f clk per value\n", blocks, (mode & 1) ? " + tanh" : "", (mode & 2) ? " + split" : "", (double)h / (iters * 8));
        }
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k_layers, dim3(blocks), dim3(256), (frag_u4(128, 64) + frag_u4(64, 128)) * 16 + 1024, 0, d, sink, iters);
        hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("blocks %3d  layers 128->64 + 64->128 (96 MFMA, 48 values/lane)  : %.0f clk per pair  (MFMA floor %d, epilogue at the rate above)\n", blocks, (double)h / iters, 96 * 16);
    }
    hipError_t e = hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(e));
    return 0;
}
