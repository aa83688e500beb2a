// micro-benchmark + ACCURACY check of one dense layer  out = tanh(X.W^T + b)  (32 rows, K = 128 -> 64 outputs, one 4-wave
// workgroup, two workgroups per CU as in the policy kernel) carried on the gfx950 matrix cores in four ways:
//   f32     : 8 x v_mfma_f32_16x16x4_f32 per 16x16x32 block (round-1 kernel)
//   f16x2/3 : operands as hi = f16(x), lo = f16((x - hi) * 2^12); products hi.hi + 2^-12 (hi.lo + lo.hi): 3 MFMAs per block,
//             cross terms in their own accumulator.  Operand error 2^-22, dropped lo.lo term 2^-24 relative.
//   f16x2/4 : the same plus lo.lo (scaled 2^-24): 4 MFMAs per block
//   bf16x3/6: exact 3-way bf16 split, 6 MFMAs per block
// Transposed formulation for the 16-bit forms: D[feature][row] = sum_k W[feature][k] X[row][k], i.e. weights are the A
// operand (registers), activations the B operand (8 consecutive k of one row = one ds_read_b128), and a lane's four D
// values are four CONSECUTIVE features of one row -> one ds_write_b64 per plane.
// Each variant is checked against an f64 host reference (max abs error of the tanh output), then timed over `iters`
// chained layers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float fast_tanh(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
constexpr int ROWS = 32, K = 128, OUT = 64;
constexpr float LO_SCALE = 4096.0f, LO_INV = 1.0f / 4096.0f;

// ---------------- f32 (current kernel form) ----------------
constexpr int SA = K + 4;
__global__ __launch_bounds__(256) void k_f32(const float *X, const float *Wt /*[K][OUT]*/, const float *bias, float *Y, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) float buf[2][ROWS * SA];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < ROWS * K; i += 256) { buf[0][(i / K) * SA + i % K] = X[i]; buf[1][(i / K) * SA + i % K] = 0.f; }
    float b[32];
    for (int kk = 0; kk < 32; ++kk) { const int k = 16 * (kk >> 2) + 4 * g + (kk & 3); b[kk] = Wt[k * OUT + 16 * wave + c]; }
    const float bv = bias[16 * wave + c];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const float *in = buf[it & 1];
        float *o = buf[(it + 1) & 1];
        const float4 *pa = reinterpret_cast<const float4 *>(in + c * SA + 4 * g), *pb = reinterpret_cast<const float4 *>(in + (16 + c) * SA + 4 * g);
        float4 a0[8], a1[8];
#pragma unroll
        for (int kq = 0; kq < 8; ++kq) { a0[kq] = pa[4 * kq]; a1[kq] = pb[4 * kq]; }
        v4f acc0 = { bv, bv, bv, bv }, acc1 = acc0;
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
    if (iters == 1) for (int i = tid; i < ROWS * OUT; i += 256) Y[i] = buf[1][(i / OUT) * SA + i % OUT];
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

// ---------------- f16 two-way split ----------------
// planes [row][SH] of f16; SH = K + 8 halves -> row stride (K + 8) / 2 = 68 words == 4 (mod 64): the 16 rows of a
// ds_read_b128 quarter-wave land on distinct bank groups
constexpr int SH = K + 8;
__device__ __forceinline__ void split2(float x, _Float16 &h, _Float16 &l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * LO_SCALE);
}
template <int NMF>
__global__ __launch_bounds__(256) void k_f16(const float *X, const float *W /*[OUT][K]*/, const float *bias, float *Y, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) _Float16 ph[2][ROWS * SH], pl[2][ROWS * SH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < ROWS * K; i += 256) {
        _Float16 h, l; split2(X[i], h, l);
        ph[0][(i / K) * SH + i % K] = h; pl[0][(i / K) * SH + i % K] = l; ph[1][(i / K) * SH + i % K] = (_Float16)0.f; pl[1][(i / K) * SH + i % K] = (_Float16)0.f;
    }
    // A operand = weights of this wave's 16 output features: lane (i = c, k = 32q + 8g + e)
    v8h wh[4], wl[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) { _Float16 h, l; split2(W[(16 * wave + c) * K + 32 * q + 8 * g + e], h, l); wh[q][e] = h; wl[q][e] = l; }
    float bv[4];
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wave + 4 * g + r];     // D rows = features 4g + r
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int src = it & 1, dst = (it + 1) & 1;
        v8h xh0[4], xl0[4], xh1[4], xl1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                  // B operand: lane (j = row c, k = 32q + 8g ..): one b128 per plane
            xh0[q] = *reinterpret_cast<const v8h *>(&ph[src][c * SH + 32 * q + 8 * g]); xl0[q] = *reinterpret_cast<const v8h *>(&pl[src][c * SH + 32 * q + 8 * g]);
            xh1[q] = *reinterpret_cast<const v8h *>(&ph[src][(16 + c) * SH + 32 * q + 8 * g]); xl1[q] = *reinterpret_cast<const v8h *>(&pl[src][(16 + c) * SH + 32 * q + 8 * g]);
        }
        v4f hh0 = { bv[0], bv[1], bv[2], bv[3] }, hh1 = hh0, cr0 = { 0, 0, 0, 0 }, cr1 = cr0, ll0 = cr0, ll1 = cr0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#define MF(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)
            MF(wh[q], xh0[q], hh0); MF(wh[q], xh1[q], hh1);
            MF(wh[q], xl0[q], cr0); MF(wh[q], xl1[q], cr1);
            MF(wl[q], xh0[q], cr0); MF(wl[q], xh1[q], cr1);
            if (NMF == 4) { MF(wl[q], xl0[q], ll0); MF(wl[q], xl1[q], ll1); }
        }
        // lane holds features 16*wave + 4g .. +3 of rows c and 16 + c: 4 consecutive halves per plane -> ds_write_b64
        v4h oh0, ol0, oh1, ol1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float y0 = hh0[r] + cr0[r] * LO_INV, y1 = hh1[r] + cr1[r] * LO_INV;
            if (NMF == 4) { y0 += ll0[r] * (LO_INV * LO_INV); y1 += ll1[r] * (LO_INV * LO_INV); }
            _Float16 h, l;
            split2(fast_tanh(y0), h, l); oh0[r] = h; ol0[r] = l;
            split2(fast_tanh(y1), h, l); oh1[r] = h; ol1[r] = l;
        }
        *reinterpret_cast<v4h *>(&ph[dst][c * SH + 16 * wave + 4 * g]) = oh0; *reinterpret_cast<v4h *>(&pl[dst][c * SH + 16 * wave + 4 * g]) = ol0;
        *reinterpret_cast<v4h *>(&ph[dst][(16 + c) * SH + 16 * wave + 4 * g]) = oh1; *reinterpret_cast<v4h *>(&pl[dst][(16 + c) * SH + 16 * wave + 4 * g]) = ol1;
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (iters == 1) for (int i = tid; i < ROWS * OUT; i += 256) Y[i] = (float)ph[1][(i / OUT) * SH + i % OUT] + (float)pl[1][(i / OUT) * SH + i % OUT] * LO_INV;
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

// ---------------- bf16 three-way split (transposed formulation too) ----------------
__device__ __forceinline__ void split3(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    const unsigned int ux = __float_as_uint(x), uh = ux & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(uh);
    const unsigned int um = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(um);
    h = (unsigned short)(uh >> 16); m = (unsigned short)(um >> 16); l = (unsigned short)(__float_as_uint(r2) >> 16);
}
__global__ __launch_bounds__(256) void k_bf16(const float *X, const float *W, const float *bias, float *Y, int iters, unsigned long long *clk) {
    __shared__ __attribute__((aligned(16))) unsigned short p[2][3][ROWS * SH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int i = tid; i < ROWS * K; i += 256) {
        unsigned short h, m, l; split3(X[i], h, m, l);
        const int o = (i / K) * SH + i % K;
        p[0][0][o] = h; p[0][1][o] = m; p[0][2][o] = l; p[1][0][o] = p[1][1][o] = p[1][2][o] = 0;
    }
    v8s wh[4], wm[4], wl[4];
    for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) { unsigned short h, m, l; split3(W[(16 * wave + c) * K + 32 * q + 8 * g + e], h, m, l); wh[q][e] = (short)h; wm[q][e] = (short)m; wl[q][e] = (short)l; }
    float bv[4];
    for (int r = 0; r < 4; ++r) bv[r] = bias[16 * wave + 4 * g + r];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int src = it & 1, dst = (it + 1) & 1;
        v4f acc0 = { bv[0], bv[1], bv[2], bv[3] }, acc1 = acc0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const v8s xh0 = *reinterpret_cast<const v8s *>(&p[src][0][c * SH + 32 * q + 8 * g]), xm0 = *reinterpret_cast<const v8s *>(&p[src][1][c * SH + 32 * q + 8 * g]),
                      xl0 = *reinterpret_cast<const v8s *>(&p[src][2][c * SH + 32 * q + 8 * g]);
            const v8s xh1 = *reinterpret_cast<const v8s *>(&p[src][0][(16 + c) * SH + 32 * q + 8 * g]), xm1 = *reinterpret_cast<const v8s *>(&p[src][1][(16 + c) * SH + 32 * q + 8 * g]),
                      xl1 = *reinterpret_cast<const v8s *>(&p[src][2][(16 + c) * SH + 32 * q + 8 * g]);
#define MB(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, A), __builtin_bit_cast(v8bf, B), ACC, 0, 0, 0)
            MB(wl[q], xh0, acc0); MB(wl[q], xh1, acc1); MB(wh[q], xl0, acc0); MB(wh[q], xl1, acc1); MB(wm[q], xm0, acc0); MB(wm[q], xm1, acc1);
            MB(wm[q], xh0, acc0); MB(wm[q], xh1, acc1); MB(wh[q], xm0, acc0); MB(wh[q], xm1, acc1); MB(wh[q], xh0, acc0); MB(wh[q], xh1, acc1);
        }
        v4s o0[3], o1[3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned short h, m, l;
            split3(fast_tanh(acc0[r]), h, m, l); o0[0][r] = (short)h; o0[1][r] = (short)m; o0[2][r] = (short)l;
            split3(fast_tanh(acc1[r]), h, m, l); o1[0][r] = (short)h; o1[1][r] = (short)m; o1[2][r] = (short)l;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            *reinterpret_cast<v4s *>(&p[dst][s][c * SH + 16 * wave + 4 * g]) = o0[s];
            *reinterpret_cast<v4s *>(&p[dst][s][(16 + c) * SH + 16 * wave + 4 * g]) = o1[s];
        }
        __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (iters == 1)
        for (int i = tid; i < ROWS * OUT; i += 256) {
            const int o = (i / OUT) * SH + i % OUT;
            Y[i] = __uint_as_float((unsigned)p[1][0][o] << 16) + __uint_as_float((unsigned)p[1][1][o] << 16) + __uint_as_float((unsigned)p[1][2][o] << 16);
        }
    if (blockIdx.x == 0 && tid == 0) clk[0] = t1 - t0;
}

int main() {
    std::vector<float> X(ROWS * K), W(OUT * K), Wt(K * OUT), b(OUT), Y(ROWS * OUT);
    srand(7);
    auto U = [] { return (float)rand() / RAND_MAX * 2.0f - 1.0f; };
    for (auto &v : X) v = tanhf(1.5f * U());                      // activations: tanh outputs
    for (int i = 0; i < 8; ++i) X[i] = U() * 1e-4f;              // a few tiny ones
    for (auto &v : W) v = 0.3f * U();
    for (auto &v : b) v = 0.1f * U();
    for (int o = 0; o < OUT; ++o) for (int k = 0; k < K; ++k) Wt[k * OUT + o] = W[o * K + k];
    std::vector<double> ref(ROWS * OUT);
    for (int r = 0; r < ROWS; ++r) for (int o = 0; o < OUT; ++o) { double s = b[o]; for (int k = 0; k < K; ++k) s += (double)X[r * K + k] * (double)W[o * K + k]; ref[r * OUT + o] = tanh(s); }
    float *dX, *dW, *dWt, *db, *dY; unsigned long long *dc, hc;
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dWt, Wt.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dY, Y.size() * 4); hipMalloc(&dc, 8);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dWt, Wt.data(), Wt.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    const char *names[4] = { "f32 (8 x 16x16x4)  ", "f16x2, 3 MFMAs     ", "f16x2, 4 MFMAs     ", "bf16x3, 6 MFMAs    " };
    for (int v = 0; v < 4; ++v) {
        auto launch = [&](int blocks, int iters) {
            if (v == 0) hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, dX, dWt, db, dY, iters, dc);
            if (v == 1) hipLaunchKernelGGL(k_f16<3>, dim3(blocks), dim3(256), 0, 0, dX, dW, db, dY, iters, dc);
            if (v == 2) hipLaunchKernelGGL(k_f16<4>, dim3(blocks), dim3(256), 0, 0, dX, dW, db, dY, iters, dc);
            if (v == 3) hipLaunchKernelGGL(k_bf16, dim3(blocks), dim3(256), 0, 0, dX, dW, db, dY, iters, dc);
        };
        launch(1, 1);
        hipDeviceSynchronize();
        hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, esum = 0;
        for (size_t i = 0; i < Y.size(); ++i) { const double e = fabs((double)Y[i] - ref[i]); emax = e > emax ? e : emax; esum += e; }
        const int iters = 400;
        for (int w = 0; w < 3; ++w) launch(512, iters);
        hipDeviceSynchronize();
        hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
        printf("%s max |err| vs f64 %.3e  mean %.3e   %.0f clk per layer\n", names[v], emax, esum / Y.size(), (double)hc / iters);
    }
    return 0;
}
