// micro-benchmark (timing only, operand layouts not validated): cost of one dense layer step as the policy kernel
// runs it - A operand from LDS, B fragments in registers, MFMA, bias + fast tanh, result back to LDS - in two forms:
//   f32    : 8 x v_mfma_f32_16x16x4_f32 per 16x16x32 block (what the kernel does today)
//   bf16x6 : activations and weights kept as three bf16 planes (exact 3-way split of the f32 value), 6 x
//            v_mfma_f32_16x16x32_bf16 per block, the epilogue re-splits the f32 result into the planes
// Shape: 32 rows x K = 128 -> 64 outputs, one 4-wave workgroup (each wave: 1 column tile, 2 row tiles), `iters` layers
// back to back, two workgroups per CU resident as in the real kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
constexpr int K = 128, SA = K + 4;
__global__ __launch_bounds__(256) void k_f32(float *out, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) float buf[2][32 * SA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 32 * SA; i += 256) (&buf[0][0])[i] = (i % 7) * 0.01f;
    float b[32];
    for (int i = 0; i < 32; ++i) b[i] = 0.001f * (i + lane);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float *in = buf[it & 1];
        float *o = buf[(it + 1) & 1];
        const float4 *pa = reinterpret_cast<const float4 *>(in + c * SA + 4 * g), *pb = reinterpret_cast<const float4 *>(in + (16 + c) * SA + 4 * g);
        float4 a0[8], a1[8];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) { a0[kq] = pa[4 * kq]; a1[kq] = pb[4 * kq]; }
        v4f acc0 = { 0.1f, 0.1f, 0.1f, 0.1f }, acc1 = acc0;
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kq].x, b[4 * kq + 0], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kq].x, b[4 * kq + 0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kq].y, b[4 * kq + 1], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kq].y, b[4 * kq + 1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kq].z, b[4 * kq + 2], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kq].z, b[4 * kq + 2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[kq].w, b[4 * kq + 3], acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[kq].w, b[4 * kq + 3], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[(4 * g + r) * SA + 16 * wave + c] = fast_tanh(acc0[r]);
            o[(16 + 4 * g + r) * SA + 16 * wave + c] = fast_tanh(acc1[r]);
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = buf[0][tid];
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}
// bf16 planes: plane p of row r at pl[p][r * SB + k], 16-byte reads of 8 consecutive k
constexpr int SB = K + 8;
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    const unsigned int ux = __float_as_uint(x), uh = ux & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(uh);
    const unsigned int um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = (unsigned short)(uh >> 16); m = (unsigned short)(um >> 16); l = (unsigned short)(__float_as_uint(r2) >> 16);
}
__global__ __launch_bounds__(256) void k_bf16(float *out, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) unsigned short pl[2][3][32 * SB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < 2 * 3 * 32 * SB; i += 256) (&pl[0][0][0])[i] = (unsigned short)(0x3C00 + (i % 7));
    v8s bh[4], bm[4], bl[4];                           // weights: 3 planes x 4 k-blocks of 32
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) { bh[q][e] = (short)(0x3C00 + e + lane); bm[q][e] = (short)(0x3800 + e); bl[q][e] = (short)(0x3400 + q); }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int src = it & 1, dst = (it + 1) & 1;
        v4f acc0 = { 0.1f, 0.1f, 0.1f, 0.1f }, acc1 = acc0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // 16x16x32 blocks: lane (row c, k group g) reads 8 consecutive k
            v8s ah0 = *reinterpret_cast<const v8s *>(&pl[src][0][c * SB + 32 * q + 8 * g]), am0 = *reinterpret_cast<const v8s *>(&pl[src][1][c * SB + 32 * q + 8 * g]),
                al0 = *reinterpret_cast<const v8s *>(&pl[src][2][c * SB + 32 * q + 8 * g]);
            v8s ah1 = *reinterpret_cast<const v8s *>(&pl[src][0][(16 + c) * SB + 32 * q + 8 * g]), am1 = *reinterpret_cast<const v8s *>(&pl[src][1][(16 + c) * SB + 32 * q + 8 * g]),
                al1 = *reinterpret_cast<const v8s *>(&pl[src][2][(16 + c) * SB + 32 * q + 8 * g]);
#define MF(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, A), __builtin_bit_cast(v8bf, B), ACC, 0, 0, 0)
            MF(ah0, bh[q], acc0); MF(ah1, bh[q], acc1); MF(ah0, bm[q], acc0); MF(ah1, bm[q], acc1); MF(am0, bh[q], acc0); MF(am1, bh[q], acc1);
            MF(ah0, bl[q], acc0); MF(ah1, bl[q], acc1); MF(al0, bh[q], acc0); MF(al1, bh[q], acc1); MF(am0, bm[q], acc0); MF(am1, bm[q], acc1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned short h, m, l;
            split3(fast_tanh(acc0[r]), h, m, l);
            int o = (4 * g + r) * SB + 16 * wave + c;
            pl[dst][0][o] = h; pl[dst][1][o] = m; pl[dst][2][o] = l;
            split3(fast_tanh(acc1[r]), h, m, l);
            o = (16 + 4 * g + r) * SB + 16 * wave + c;
            pl[dst][0][o] = h; pl[dst][1][o] = m; pl[dst][2][o] = l;
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + tid] = (float)pl[0][0][tid];
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c, h;
    hipMalloc(&o, 512 * 256 * 4); hipMalloc(&c, 8);
    const int iters = 400;
    for (int v = 0; v < 2; ++v) {
        for (int w = 0; w < 3; ++w) { if (v == 0) hipLaunchKernelGGL(k_f32, dim3(512), dim3(256), 0, 0, o, iters, c); else hipLaunchKernelGGL(k_bf16, dim3(512), dim3(256), 0, 0, o, iters, c); }
        hipDeviceSynchronize();
        hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
        printf("%s: %.0f clk per layer (32 rows x 128 -> 64, 2 workgroups per CU)\n", v == 0 ? "f32   " : "bf16x6", (double)h / iters);
    }
    return 0;
}
