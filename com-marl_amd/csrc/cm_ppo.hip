// cm_ppo.hip - PPO-update side kernels (gfx950):
//   * cm_masked_agg_forward/backward : the adjacency-masked GCN aggregation as one fused op for
//     the autograd path  (com_marl/torch/modules/comm_base_net.py:99-105 +
//     graph_conv_module.py:63-70):  A = M*R*C ; A /= rowsum + 1e-12 ; out = tanh(A.(HW) + b)
//   * cm_discount_returns             : garage/misc/tensor_utils.py:7-23 (f64 recurrence)
//   * cm_gae                          : garage/torch/algos/_utils.py:56-113 + the per-path
//                                       normalisation of centralized_ma_ppo.py:422-426
// Samples are [P*T] env states; a 256-thread workgroup owns EPB whole samples so that the
// N x N mask tile and the N x 64 feature tile are read from HBM exactly once.
#include <algorithm>

#include <stdlib.h>

#include "cm_internal.h"

namespace cm {

constexpr int TPB = 256;
constexpr int RC = 4;

__host__ __device__ inline int agg_epb(int N) { return (N % RC == 0) ? (48 / N > 0 ? 48 / N : 1) : 1; }

// LDS: A [rows][NP], HW [rows][SE], (bwd: dP [rows][SE], dA [rows][NP], mask [rows][NP], den [rows])
template <int E>
__global__ __launch_bounds__(TPB) void agg_fwd_kernel(int S, int N, int EPB, const float *__restrict__ attn,
                                                     const float *__restrict__ adj, const float *__restrict__ chan,
                                                     long ch_stride, const float *__restrict__ hw,
                                                     const float *__restrict__ bias, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SE = E + 4;
    const int tid = threadIdx.x, NN = N * N, NP = N | 1, rows_max = EPB * N;
    float *A = lds, *HW = A + (size_t)rows_max * NP;
    for (int s0 = blockIdx.x * EPB; s0 < S; s0 += gridDim.x * EPB) {
        const int envs = min(EPB, S - s0), rows = envs * N;
        for (int k = tid; k < envs * NN; k += TPB) {
            const int e = k / NN, ij = k - e * NN, r = k / N, j = k - r * N;
            float v = attn[(size_t)s0 * NN + k];
            if (adj) v *= adj[(size_t)s0 * NN + k];
            if (chan) v *= chan[(size_t)(s0 + e) * ch_stride + ij];
            A[(size_t)r * NP + j] = v;
        }
        for (int k = tid; k < rows * E; k += TPB) { const int r = k / E, o = k - r * E; HW[(size_t)r * SE + o] = hw[(size_t)s0 * N * E + k]; }
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            float *ar = A + (size_t)r * NP;
            float sum = 0.0f;
            for (int j = 0; j < N; ++j) sum += ar[j];
            const float den = sum + 1e-12f;
            for (int j = 0; j < N; ++j) ar[j] = ar[j] / den;
        }
        __syncthreads();
        const int o = tid % E, rg = tid / E;
        const float bv = bias ? bias[o] : 0.0f;
        for (int r0 = rg * RC; r0 < rows; r0 += (TPB / E) * RC) {
            const int e = r0 / N;
            const float *h = HW + (size_t)e * N * SE + o;
            float acc[RC] = { 0.0f, 0.0f, 0.0f, 0.0f };
            const float *ar[RC];
#pragma unroll
            for (int i = 0; i < RC; ++i) ar[i] = A + (size_t)min(r0 + i, rows - 1) * NP;
            for (int j = 0; j < N; ++j) {
                const float hv = h[(size_t)j * SE];
#pragma unroll
                for (int i = 0; i < RC; ++i) acc[i] = fmaf(ar[i][j], hv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < RC; ++i)
                if (r0 + i < rows && (r0 + i) / N == e) out[((size_t)s0 * N + r0 + i) * E + o] = tanhf(acc[i] + bv);
        }
        __syncthreads();
    }
}

template <int E>
__global__ __launch_bounds__(TPB) void agg_bwd_kernel(int S, int N, int EPB, const float *__restrict__ attn,
                                                     const float *__restrict__ adj, const float *__restrict__ chan,
                                                     long ch_stride, const float *__restrict__ hw,
                                                     const float *__restrict__ outv, const float *__restrict__ out_minus,
                                                     const float *__restrict__ d_out,
                                                     float *__restrict__ d_attn, float *__restrict__ d_hw,
                                                     float *__restrict__ d_bias) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SE = E + 4;
    const int tid = threadIdx.x, NN = N * N, NP = N | 1, rows_max = EPB * N;
    float *A = lds;                                   // normalised A
    float *MK = A + (size_t)rows_max * NP;            // mask product R*C
    float *DA = MK + (size_t)rows_max * NP;           // dL/dA
    float *HW = DA + (size_t)rows_max * NP;           // [rows][SE]
    float *DP = HW + (size_t)rows_max * SE;           // dL/d(pre-activation) [rows][SE]
    float *den = DP + (size_t)rows_max * SE;          // [rows]
    float *dbs = den + rows_max;                      // [TPB/E][E] partial bias grads
    const int o = tid % E, rg = tid / E;
    float dbias_acc = 0.0f;
    for (int s0 = blockIdx.x * EPB; s0 < S; s0 += gridDim.x * EPB) {
        const int envs = min(EPB, S - s0), rows = envs * N;
        for (int k = tid; k < envs * NN; k += TPB) {
            const int e = k / NN, ij = k - e * NN, r = k / N, j = k - r * N;
            float m = 1.0f;
            if (adj) m *= adj[(size_t)s0 * NN + k];
            if (chan) m *= chan[(size_t)(s0 + e) * ch_stride + ij];
            MK[(size_t)r * NP + j] = m;
            A[(size_t)r * NP + j] = attn[(size_t)s0 * NN + k] * m;
        }
        for (int k = tid; k < rows * E; k += TPB) {
            const int r = k / E, c = k - r * E;
            const size_t g = (size_t)s0 * N * E + k;
            HW[(size_t)r * SE + c] = hw[g];
            const float y = out_minus ? outv[g] - out_minus[g] : outv[g];
            const float dp = d_out[g] * (1.0f - y * y);          // tanh'
            DP[(size_t)r * SE + c] = dp;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            float *ar = A + (size_t)r * NP;
            float sum = 0.0f;
            for (int j = 0; j < N; ++j) sum += ar[j];
            const float dn = sum + 1e-12f;
            den[r] = dn;
            for (int j = 0; j < N; ++j) ar[j] = ar[j] / dn;
        }
        // bias grad partial: column o over this thread's row group
        for (int r = rg; r < rows; r += TPB / E) dbias_acc += DP[(size_t)r * SE + o];
        __syncthreads();
        // d_hw[j][o] = sum_i A[i][j] * dP[i][o]   (rows i of the same sample)
        for (int r0 = rg * RC; r0 < rows; r0 += (TPB / E) * RC) {
            const int e = r0 / N;
            float acc[RC] = { 0.0f, 0.0f, 0.0f, 0.0f };
            for (int i = 0; i < N; ++i) {
                const float dpv = DP[(size_t)(e * N + i) * SE + o];
                const float *ai = A + (size_t)(e * N + i) * NP;
#pragma unroll
                for (int q = 0; q < RC; ++q) {
                    const int j = min(r0 + q, rows - 1) - e * N;
                    if (j < N) acc[q] = fmaf(ai[j], dpv, acc[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < RC; ++q)
                if (r0 + q < rows && (r0 + q) / N == e) d_hw[((size_t)s0 * N + r0 + q) * E + o] = acc[q];
        }
        // dA[i][j] = sum_o dP[i][o] * HW[j][o]
        for (int k = tid; k < envs * NN; k += TPB) {
            const int e = k / NN, ij = k - e * NN, i = ij / N, j = ij - i * N;
            const float4 *x = reinterpret_cast<const float4 *>(DP + (size_t)(e * N + i) * SE);
            const float4 *y = reinterpret_cast<const float4 *>(HW + (size_t)(e * N + j) * SE);
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < E / 4; ++c) {
                const float4 u = x[c], v = y[c];
                acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
            }
            DA[(size_t)(e * N + i) * NP + j] = acc;
        }
        __syncthreads();
        // through the renormalisation: dM_ij = mask_ij * (dA_ij - sum_k dA_ik A_ik) / den_i
        for (int r = tid; r < rows; r += TPB) {
            const float *ar = A + (size_t)r * NP, *da = DA + (size_t)r * NP, *mk = MK + (size_t)r * NP;
            float t = 0.0f;
            for (int j = 0; j < N; ++j) t = fmaf(da[j], ar[j], t);
            const float inv = 1.0f / den[r];
            float *dst = d_attn + ((size_t)s0 * N + r) * N;
            for (int j = 0; j < N; ++j) dst[j] = mk[j] * (da[j] - t) * inv;
        }
        __syncthreads();
    }
    if (d_bias) {
        dbs[rg * E + o] = dbias_acc;
        __syncthreads();
        if (rg == 0) {
            float v = 0.0f;
            for (int q = 0; q < TPB / E; ++q) v += dbs[q * E + o];
            atomicAdd(d_bias + o, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// attention scores + softmax for the autograd path (attention_module.py:39-49):
//   M[s,i,:] = softmax_j( Q[s,i,:] . E[s,j,:] )          Q = linear_in(E) comes from a torch GEMM
// torch.matmul lowers this to a batched GEMM with N x N outputs (4 x 4 at config 2): three such GEMMs
// (forward, dQ, dE) were 31 % of the PPO update.  Here a workgroup owns EPB whole samples, the Q / E
// tiles are read from HBM once and everything else happens in LDS.
// ---------------------------------------------------------------------------------------------
template <int E>
__global__ __launch_bounds__(TPB) void attn_fwd_kernel(int S, int N, int EPB, const float *__restrict__ q,
                                                      const float *__restrict__ e, float *__restrict__ m) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SE = E + 4;
    const int tid = threadIdx.x, NN = N * N, NP = N | 1, rows_max = EPB * N;
    float *Q = lds, *K = Q + (size_t)rows_max * SE, *M = K + (size_t)rows_max * SE;
    for (int s0 = blockIdx.x * EPB; s0 < S; s0 += gridDim.x * EPB) {
        const int envs = min(EPB, S - s0), rows = envs * N;
        for (int k = tid; k < rows * (E / 4); k += TPB) {
            const int r = k / (E / 4), c = k - r * (E / 4);
            reinterpret_cast<float4 *>(Q + (size_t)r * SE)[c] = reinterpret_cast<const float4 *>(q + ((size_t)s0 * N + r) * E)[c];
            reinterpret_cast<float4 *>(K + (size_t)r * SE)[c] = reinterpret_cast<const float4 *>(e + ((size_t)s0 * N + r) * E)[c];
        }
        __syncthreads();
        for (int k = tid; k < envs * NN; k += TPB) {
            const int en = k / NN, ij = k - en * NN, i = ij / N, j = ij - i * N;
            const float4 *x = reinterpret_cast<const float4 *>(Q + (size_t)(en * N + i) * SE);
            const float4 *y = reinterpret_cast<const float4 *>(K + (size_t)(en * N + j) * SE);
            float acc = 0.0f;
#pragma unroll
            for (int c = 0; c < E / 4; ++c) {
                const float4 u = x[c], v = y[c];
                acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
            }
            M[(size_t)(en * N + i) * NP + j] = acc;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            float *mr = M + (size_t)r * NP;
            float mx = -INFINITY, sum = 0.0f;
            for (int j = 0; j < N; ++j) mx = fmaxf(mx, mr[j]);
            for (int j = 0; j < N; ++j) { const float ex = expf(mr[j] - mx); mr[j] = ex; sum += ex; }
            float *dst = m + ((size_t)s0 * N + r) * N;
            for (int j = 0; j < N; ++j) dst[j] = mr[j] / sum;
        }
        __syncthreads();
    }
}

// dS = M * (dM - sum_j dM*M) ; dQ = dS . E ; dE = dS^T . Q
template <int E>
__global__ __launch_bounds__(TPB) void attn_bwd_kernel(int S, int N, int EPB, const float *__restrict__ q,
                                                      const float *__restrict__ e, const float *__restrict__ m,
                                                      const float *__restrict__ d_m, const float *__restrict__ add0,
                                                      const float *__restrict__ add1, float *__restrict__ d_q,
                                                      float *__restrict__ d_e) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SE = E + 4;
    const int tid = threadIdx.x, NP = N | 1, rows_max = EPB * N;
    float *Q = lds, *K = Q + (size_t)rows_max * SE, *DS = K + (size_t)rows_max * SE;
    for (int s0 = blockIdx.x * EPB; s0 < S; s0 += gridDim.x * EPB) {
        const int envs = min(EPB, S - s0), rows = envs * N;
        for (int k = tid; k < rows * (E / 4); k += TPB) {
            const int r = k / (E / 4), c = k - r * (E / 4);
            reinterpret_cast<float4 *>(Q + (size_t)r * SE)[c] = reinterpret_cast<const float4 *>(q + ((size_t)s0 * N + r) * E)[c];
            reinterpret_cast<float4 *>(K + (size_t)r * SE)[c] = reinterpret_cast<const float4 *>(e + ((size_t)s0 * N + r) * E)[c];
        }
        for (int r = tid; r < rows; r += TPB) {
            const float *mr = m + ((size_t)s0 * N + r) * N, *dr = d_m + ((size_t)s0 * N + r) * N;
            float t = 0.0f;
            for (int j = 0; j < N; ++j) t = fmaf(dr[j], mr[j], t);
            for (int j = 0; j < N; ++j) DS[(size_t)r * NP + j] = mr[j] * (dr[j] - t);
        }
        __syncthreads();
        const int o = tid % E, rg = tid / E;
        for (int r0 = rg * RC; r0 < rows; r0 += (TPB / E) * RC) {
            const int en = r0 / N;
            float aq[RC] = { 0.f, 0.f, 0.f, 0.f }, ae[RC] = { 0.f, 0.f, 0.f, 0.f };
            for (int j = 0; j < N; ++j) {
                const float ev = K[(size_t)(en * N + j) * SE + o], qv = Q[(size_t)(en * N + j) * SE + o];
#pragma unroll
                for (int i = 0; i < RC; ++i) {
                    const int r = min(r0 + i, rows - 1), li = r - en * N;
                    if (li < N) {
                        aq[i] = fmaf(DS[(size_t)r * NP + j], ev, aq[i]);                  // dQ[i] += dS[i][j] E[j]
                        ae[i] = fmaf(DS[(size_t)(en * N + j) * NP + li], qv, ae[i]);      // dE[i] += dS[j][i] Q[j]
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < RC; ++i)
                if (r0 + i < rows && (r0 + i) / N == en) {
                    const size_t at = ((size_t)s0 * N + r0 + i) * E + o;
                    d_q[at] = aq[i];
                    float v = ae[i];
                    if (add0) v += add0[at];
                    if (add1) v += add1[at];
                    d_e[at] = v;
                }
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Teams of 4 (the headline config): the two backward kernels above as register kernels.  16 lanes own one env, lane c
// the four features 4c .. 4c+3 of each of the env's four agent rows: every global access is a 16-byte load / store in
// 256-byte row segments, ~256 bytes in flight per lane (the generic kernels stage through LDS with 4-byte accesses and
// run at ~2 TB/s once they read more than three streams).  The 4 x 4 coefficient matrices (normalised A, softmax
// gradient) are computed one element per lane and shared through LDS.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 f4_fma(float s, const float4 &v, const float4 &a) {
    return make_float4(fmaf(s, v.x, a.x), fmaf(s, v.y, a.y), fmaf(s, v.z, a.z), fmaf(s, v.w, a.w));
}
__device__ __forceinline__ float quad_sum(float v) {      // sum over the 4 lanes of a quad (lanes 4k .. 4k+3)
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    return v;
}

__global__ __launch_bounds__(256) void attn_bwd4_kernel(int S, const float *__restrict__ q, const float *__restrict__ e,
                                                        const float *__restrict__ m, const float *__restrict__ d_m,
                                                        const float *__restrict__ add0, const float *__restrict__ add1,
                                                        float *__restrict__ d_q, float *__restrict__ d_e) {
    __shared__ __attribute__((aligned(16))) float DSs[16][16];
    const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
    for (int base = blockIdx.x * 16; base < S; base += gridDim.x * 16) {
        const int s = base + grp;
        const bool live = s < S;
        const size_t row0 = (size_t)s * 4 * 64;
        float4 qv[4], ev[4], a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qv[i] = ev[i] = a0[i] = a1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                qv[i] = *reinterpret_cast<const float4 *>(q + row0 + i * 64 + 4 * c);
                ev[i] = *reinterpret_cast<const float4 *>(e + row0 + i * 64 + 4 * c);
                if (add0) a0[i] = *reinterpret_cast<const float4 *>(add0 + row0 + i * 64 + 4 * c);
                if (add1) a1[i] = *reinterpret_cast<const float4 *>(add1 + row0 + i * 64 + 4 * c);
            }
        }
        // softmax backward, element (i = c / 4, j = c % 4): dS = m * (dm - sum_j dm m)
        const float mv = live ? m[(size_t)s * 16 + c] : 0.0f, dv = live ? d_m[(size_t)s * 16 + c] : 0.0f;
        const float t = quad_sum(dv * mv);
        __syncthreads();                                   // previous iteration's readers are done
        DSs[grp][c] = mv * (dv - t);
        __syncthreads();
        float ds[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(&DSs[grp][4 * k]);
            ds[4 * k] = v.x; ds[4 * k + 1] = v.y; ds[4 * k + 2] = v.z; ds[4 * k + 3] = v.w;
        }
        if (live) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 aq = make_float4(0.f, 0.f, 0.f, 0.f), ae = aq;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    aq = f4_fma(ds[4 * i + j], ev[j], aq);                 // dQ[i] += dS[i][j] E[j]
                    ae = f4_fma(ds[4 * j + i], qv[j], ae);                 // dE[i] += dS[j][i] Q[j]
                }
                if (add0) { ae.x += a0[i].x; ae.y += a0[i].y; ae.z += a0[i].z; ae.w += a0[i].w; }
                if (add1) { ae.x += a1[i].x; ae.y += a1[i].y; ae.z += a1[i].z; ae.w += a1[i].w; }
                *reinterpret_cast<float4 *>(d_q + row0 + i * 64 + 4 * c) = aq;
                *reinterpret_cast<float4 *>(d_e + row0 + i * 64 + 4 * c) = ae;
            }
        }
    }
}

__global__ __launch_bounds__(256) void agg_bwd4_kernel(int S, const float *__restrict__ attn, const float *__restrict__ adj,
                                                       const float *__restrict__ chan, long ch_stride, const float *__restrict__ hw,
                                                       const float *__restrict__ outv, const float *__restrict__ out_minus,
                                                       const float *__restrict__ d_out, float *__restrict__ d_attn,
                                                       float *__restrict__ d_hw, float *__restrict__ d_bias, int bias_reps) {
    __shared__ __attribute__((aligned(16))) float As[16][16];
    __shared__ float4 dbs[256];
    const int tid = threadIdx.x, grp = tid >> 4, c = tid & 15;
    float4 db = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = blockIdx.x * 16; base < S; base += gridDim.x * 16) {
        const int s = base + grp;
        const bool live = s < S;
        const size_t row0 = (size_t)s * 4 * 64;
        float4 h[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h[i] = dp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live) {
                h[i] = *reinterpret_cast<const float4 *>(hw + row0 + i * 64 + 4 * c);
                float4 y = *reinterpret_cast<const float4 *>(outv + row0 + i * 64 + 4 * c);
                if (out_minus) {
                    const float4 u = *reinterpret_cast<const float4 *>(out_minus + row0 + i * 64 + 4 * c);
                    y.x -= u.x; y.y -= u.y; y.z -= u.z; y.w -= u.w;
                }
                const float4 d = *reinterpret_cast<const float4 *>(d_out + row0 + i * 64 + 4 * c);
                dp[i] = make_float4(d.x * (1.0f - y.x * y.x), d.y * (1.0f - y.y * y.y), d.z * (1.0f - y.z * y.z), d.w * (1.0f - y.w * y.w));   // tanh'
            }
        }
        // normalised A, element (i = c / 4, j = c % 4)
        float mk = 1.0f, av = 0.0f;
        if (live) {
            if (adj) mk *= adj[(size_t)s * 16 + c];
            if (chan) mk *= chan[(size_t)s * ch_stride + c];
            av = attn[(size_t)s * 16 + c] * mk;
        }
        const float den = quad_sum(av) + 1e-12f;
        const float an = av / den;
        __syncthreads();
        As[grp][c] = an;
        __syncthreads();
        float A[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = *reinterpret_cast<const float4 *>(&As[grp][4 * k]);
            A[4 * k] = v.x; A[4 * k + 1] = v.y; A[4 * k + 2] = v.z; A[4 * k + 3] = v.w;
        }
        // d_hw[j] = sum_i A[i][j] dP[i]
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = f4_fma(A[4 * i + j], dp[i], acc);
                *reinterpret_cast<float4 *>(d_hw + row0 + j * 64 + 4 * c) = acc;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { db.x += dp[i].x; db.y += dp[i].y; db.z += dp[i].z; db.w += dp[i].w; }
        // dA[i][j] = sum_o dP[i][o] HW[j][o]: this lane's four features, then a reduce-scatter over the env's 16 lanes
        float v[16];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[4 * i + j] = fmaf(dp[i].x, h[j].x, fmaf(dp[i].y, h[j].y, fmaf(dp[i].z, h[j].z, dp[i].w * h[j].w)));
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool hi = c & 8;
            const float mine = hi ? v[k + 8] : v[k], other = hi ? v[k] : v[k + 8];
            v[k] = mine + __shfl_xor(other, 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool hi = c & 4;
            const float mine = hi ? v[k + 4] : v[k], other = hi ? v[k] : v[k + 4];
            v[k] = mine + __shfl_xor(other, 4);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool hi = c & 2;
            const float mine = hi ? v[k + 2] : v[k], other = hi ? v[k] : v[k + 2];
            v[k] = mine + __shfl_xor(other, 2);
        }
        {
            const bool hi = c & 1;
            const float mine = hi ? v[1] : v[0], other = hi ? v[0] : v[1];
            v[0] = mine + __shfl_xor(other, 1);
        }
        // through the renormalisation: dM_ij = mask_ij * (dA_ij - sum_k dA_ik A_ik) / den_i
        const float da = v[0];
        const float tt = quad_sum(da * an);
        if (live) d_attn[(size_t)s * 16 + c] = mk * (da - tt) * (1.0f / den);
    }
    if (d_bias) {
        // 64 atomics per workgroup on the same two cache lines: with 2048 workgroups they serialise in the L2 (0.8 ns each: 105 of the
        // kernel's 312 us at 275 k envs) - the caller may hand `bias_reps` copies of the row, workgroup b adds into copy b % reps
        float *dbp = d_bias + (size_t)(blockIdx.x % (unsigned)bias_reps) * 64;
        dbs[tid] = db;
        __syncthreads();
        if (tid < 16) {
            float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int g2 = 0; g2 < 16; ++g2) { const float4 u = dbs[g2 * 16 + tid]; sum.x += u.x; sum.y += u.y; sum.z += u.z; sum.w += u.w; }
            atomicAdd(dbp + 4 * tid + 0, sum.x); atomicAdd(dbp + 4 * tid + 1, sum.y);
            atomicAdd(dbp + 4 * tid + 2, sum.z); atomicAdd(dbp + 4 * tid + 3, sum.w);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of a per-agent dense layer over R = P*T*N rows (R ~ 1e6, P,Q <= 128):
//   C[p][q] += sum_r A[r][p] * B[r][q]         ( nn.Linear: A = dY, B = X -> dW [out][in], db = colsum(dY);
//                                                GCN H.W   : A = H,  B = dZ -> dW [in][out] )
// rocBLAS/hipBLASLt run these "skinny" GEMMs (tiny output, million-deep reduction) at ~10 TFLOP/s; they
// were 42 % of the PPO update.  Here a workgroup streams 64-row chunks of A and B through LDS and keeps
// its share of the P x Q output in f32 MFMA accumulators; partial results are merged with float atomics.
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int WG_ROWS = 64;

// coalesced [WG_ROWS x W] global -> LDS copy (float4 when the row is a power-of-two number of float4s)
__device__ __forceinline__ void stage_rows(float *dst, int stride, const float *__restrict__ src, int W, long r0, int rows, int tid) {
    const int w4 = W >> 2;
    if ((W & 3) == 0 && (w4 & (w4 - 1)) == 0 && w4 <= TPB) {
        const int x = tid & (w4 - 1), rstep = TPB / w4;
        for (int r = tid / w4; r < WG_ROWS; r += rstep) {
            const float4 v = r < rows ? reinterpret_cast<const float4 *>(src + (r0 + r) * W)[x] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(dst + (size_t)r * stride + 4 * x) = v;
        }
    } else {
        for (int k = tid; k < WG_ROWS * W; k += TPB) { const int r = k / W, x = k - r * W; dst[(size_t)r * stride + x] = r < rows ? src[(r0 + r) * W + x] : 0.0f; }
    }
}

template <int MAXT>   // accumulator tiles per wave
__global__ __launch_bounds__(TPB) void wgrad_kernel(long R, int P, int Q, const float *__restrict__ A, const float *__restrict__ B,
                                                   float *__restrict__ C, float *__restrict__ colsum_a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int PT = (P + 15) >> 4, QT = (Q + 15) >> 4, NT = PT * QT;
    const int SP = PT * 16 + 16, SQ = QT * 16 + 16;          // row strides == 16 (mod 32): conflict-free operand reads
    float *As = lds, *Bs = As + (size_t)WG_ROWS * SP;
    v4f acc[MAXT];
    int aoff[MAXT], boff[MAXT];       // per-tile operand offsets, hoisted out of the k loop (no division per MFMA)
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        acc[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        const int tile = wave + 4 * t;
        const int tl = tile < NT ? tile : 0;              // idle slots recompute tile 0 and are never stored
        const int pt = tl / QT, qt = tl - pt * QT;
        aoff[t] = g * SP + c + pt * 16;
        boff[t] = g * SQ + c + qt * 16;
    }
    float csum = 0.0f;
    // zero the padding columns once (they feed MFMAs whose results are never stored)
    for (int k = tid; k < WG_ROWS * SP; k += TPB) As[k] = 0.0f;
    for (int k = tid; k < WG_ROWS * SQ; k += TPB) Bs[k] = 0.0f;
    __syncthreads();
    const long n_chunks = (R + WG_ROWS - 1) / WG_ROWS;
    for (long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const long r0 = ch * WG_ROWS;
        const int rows = (int)min((long)WG_ROWS, R - r0);
        stage_rows(As, SP, A, P, r0, rows, tid);
        stage_rows(Bs, SQ, B, Q, r0, rows, tid);
        __syncthreads();
        if (colsum_a && tid < P) { float sacc = 0.0f; for (int r = 0; r < WG_ROWS; ++r) sacc += As[(size_t)r * SP + tid]; csum += sacc; }
#pragma unroll 2
        for (int kk = 0; kk < WG_ROWS / 4; ++kk) {
            const float *ar = As + (size_t)(4 * kk) * SP, *br = Bs + (size_t)(4 * kk) * SQ;
            float av[MAXT], bw[MAXT];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) { av[t] = ar[aoff[t]]; bw[t] = br[boff[t]]; }
#pragma unroll
            for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bw[t], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        const int tile = wave + 4 * t;
        if (tile < NT) {
            const int pt = tile / QT, qt = tile - pt * QT, q = qt * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pp = pt * 16 + 4 * g + r;
                if (pp < P && q < Q) atomicAdd(C + (size_t)pp * Q + q, acc[t][r]);
            }
        }
    }
    if (colsum_a && tid < P) atomicAdd(colsum_a + tid, csum);
}

__global__ void returns_kernel(int P, int T, const double *__restrict__ rewards, const int32_t *__restrict__ lens,
                               double gamma, float *__restrict__ returns) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const int n = lens ? lens[p] : T;
    double y = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        if (t < n) { y = rewards[(size_t)p * T + t] + gamma * y; returns[(size_t)p * T + t] = (float)y; }
        else returns[(size_t)p * T + t] = 0.0f;
    }
}

// delta_t = r_t + g V_{t+1} - V_t over the PADDED length (V past the end = 0), A_t = delta_t + g*lam*A_{t+1};
// optional per-path normalisation over the first lens[p] steps, biased variance, applied to the whole row.
__global__ void gae_kernel(int P, int T, const float *__restrict__ rewards, const float *__restrict__ baselines,
                           const int32_t *__restrict__ lens, float gamma, float lam, int normalize, float eps,
                           float *__restrict__ adv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float *r = rewards + (size_t)p * T, *v = baselines + (size_t)p * T;
    float *a = adv + (size_t)p * T;
    const double gl = (double)(gamma * lam);
    double acc = 0.0, vnext = 0.0;
    for (int t = T - 1; t >= 0; --t) {
        const float delta = (r[t] + gamma * (float)vnext) - v[t];       // f32 like the reference's tensor expression
        acc = (double)delta + gl * acc;
        a[t] = (float)acc;
        vnext = v[t];
    }
    if (!normalize) return;
    const int n = lens ? lens[p] : T;
    if (n <= 0) return;
    double s = 0.0;
    for (int t = 0; t < n; ++t) s += a[t];
    const double mean = s / n;
    double q = 0.0;
    for (int t = 0; t < n; ++t) { const double dlt = a[t] - mean; q += dlt * dlt; }
    const float m32 = (float)mean, inv = 1.0f / sqrtf((float)(q / n) + eps);
    for (int t = 0; t < T; ++t) a[t] = (a[t] - m32) * inv;
}

// ---------------------------------------------------------------------------------------------
// Multi-tensor Adam (+ gradient-norm clip) for the ~17 + ~15 small parameter tensors of the policy / critic: ONE launch
// for the norm, ONE for the update, instead of ~6 framework kernels per tensor and a host sync per tensor norm.
// The update is the vendored torch-1.9 Adam of the reference (com_marl/torch/algos/my_optimizer/_functional.py:72-98):
//   m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g g;  p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// and the clip is torch.nn.utils.clip_grad_norm_ (centralized_ma_ppo.py:253-255): g *= min(1, max_norm / (|g| + 1e-6)),
// written back so that p.grad holds the clipped gradient as it does in the reference.
// ---------------------------------------------------------------------------------------------
struct TensorTable { float *p[40]; float *g[40]; float *m[40]; float *v[40]; long n[40]; int count; };

// |g|^2 in two deterministic stages (replicas of a data-parallel job must take bit-identical steps from bit-identical
// all-reduced gradients; a float-atomic sum's order - and with it the clip factor's last bit - varies run to run):
// block b writes its partial sum to ws[1 + b]; the update kernel adds the NORM_BLOCKS partials in index order.
constexpr int NORM_BLOCKS = 64;

__global__ __launch_bounds__(256) void multi_norm_kernel(TensorTable t, float *ws) {
    float acc = 0.0f;
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < t.count; ++k)
        for (long i = i0; i < t.n[k]; i += stride) { const float g = t.g[k][i]; acc = fmaf(g, g, acc); }
    __shared__ float red[256];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s]; __syncthreads(); }
    if (threadIdx.x == 0) ws[1 + blockIdx.x] = red[0];
}

// bc_tab / cursor (cm_multi_adam_step_dev): the two bias-correction factors come from a device table indexed by a device step
// counter, so that a captured hipGraph of the optimiser step can be replayed for successive steps
__global__ __launch_bounds__(256) void multi_adam_kernel(TensorTable t, float *norm_ws, float max_norm, float lr,
                                                        float b1, float b2, float eps, float bc1, float sqrt_bc2,
                                                        const float *__restrict__ bc_tab, const int32_t *__restrict__ cursor, int tab_steps) {
    if (bc_tab) {
        int c = *cursor;
        c = c < 0 ? 0 : (c >= tab_steps ? tab_steps - 1 : c);
        bc1 = bc_tab[2 * c]; sqrt_bc2 = bc_tab[2 * c + 1];
    }
    float coef = 1.0f;
    if (norm_ws) {
        float nsq = 0.0f;
        for (int b = 0; b < NORM_BLOCKS; ++b) nsq += norm_ws[1 + b];
        const float c = max_norm / (sqrtf(nsq) + 1e-6f);
        coef = c < 1.0f ? c : 1.0f;
        if (blockIdx.x == 0 && threadIdx.x == 0) norm_ws[0] = nsq;
    }
    const bool norm_sq = norm_ws != nullptr;
    const float step_size = lr / bc1;
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < t.count; ++k)
        for (long i = i0; i < t.n[k]; i += stride) {
            float g = t.g[k][i];
            if (norm_sq) { g *= coef; t.g[k][i] = g; }
            const float m = t.m[k][i] * b1 + (1.0f - b1) * g;
            const float v = t.v[k][i] * b2 + (1.0f - b2) * g * g;
            t.m[k][i] = m; t.v[k][i] = v;
            const float denom = sqrtf(v) / sqrt_bc2 + eps;
            t.p[k][i] = t.p[k][i] - step_size * (m / denom);
        }
}


// ---------------------------------------------------------------------------------------------------------------
// PPO clipped-surrogate loss + entropy bonus and its gradient wrt the logits, one launch (reference:
// centralized_ma_ppo.py:390-438 _compute_loss, :540-589 _compute_objective; the Categorical(probs=...) arithmetic of
// comm_categorical_mlp_policy.py:121-137: probs = softmax renormalised twice, logits = log(clamp(probs, eps, 1 - eps))).
// One thread per env step: N agents x A <= 8 logits in registers; the chain rule is walked link by link (clamp ->
// normalise -> normalise -> softmax) so that the result is the autograd one, term for term.
//   total = - sum_{valid s} [ min(r adv, clamp(r, 1 - c, 1 + c) adv) + ent_coeff * mean_i H_i ],  r = exp(new_ll - old_ll)
// ---------------------------------------------------------------------------------------------------------------
constexpr int PPO_MAX_A = 8;

__global__ __launch_bounds__(256) void ppo_surrogate_kernel(int P, int T, int N, int A, const float *__restrict__ logits,
                                                            const int32_t *__restrict__ actions, const float *__restrict__ old_ll,
                                                            const float *__restrict__ adv, const int32_t *__restrict__ lens,
                                                            float clip, float ent_coeff, int add_entropy,
                                                            double *__restrict__ total, long long *__restrict__ count,
                                                            float *__restrict__ dlogits) {
    const long s = (long)blockIdx.x * 256 + threadIdx.x;
    const long S = (long)P * T;
    double part = 0.0;
    int valid_here = 0;
    if (s < S) {
        const int pth = (int)(s / T), t = (int)(s - (long)pth * T);
        const bool valid = t < lens[pth];
        valid_here = valid ? 1 : 0;
        const float EPS = 1.1920928955078125e-07f;            // torch.finfo(float32).eps: probs_to_logits clamps to [eps, 1 - eps]
        const float *z0 = logits + s * N * A;
        const int32_t *a0 = actions + s * N;
        const float ol = old_ll[s], ad = adv[s];
        // pass 1: new log-likelihood and entropy
        float new_ll = 0.0f, ent = 0.0f;
        for (int i = 0; i < N; ++i) {
            float z[PPO_MAX_A], p[PPO_MAX_A];
            float mx = -INFINITY;
#pragma unroll
            for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { z[b] = z0[i * A + b]; mx = fmaxf(mx, z[b]); }
            float s0 = 0.0f;
#pragma unroll
            for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p[b] = expf(z[b] - mx); s0 += p[b]; }
            float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
            for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p[b] = p[b] / s0; s1 += p[b]; }
#pragma unroll
            for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p[b] = p[b] / s1; s2 += p[b]; }
            const int ai = a0[i];
            float h = 0.0f;
#pragma unroll
            for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) {
                const float p3 = p[b] / s2;
                const float lg = logf(fminf(fmaxf(p3, EPS), 1.0f - EPS));
                h -= lg * p3;
                if (b == ai) new_ll += lg;
            }
            ent += h;
        }
        ent /= (float)N;
        const float r = expf(new_ll - ol);
        const float rc = fminf(fmaxf(r, 1.0f - clip), 1.0f + clip);
        const float sur = r * ad, clp = rc * ad;
        float obj = fminf(sur, clp);
        if (add_entropy) obj += ent_coeff * ent;
        if (valid) part = -(double)obj;
        if (dlogits) {
            // d total / d new_ll: torch.min splits ties evenly; clamp passes the gradient inside [1 - c, 1 + c] (ends included)
            const float wa = sur < clp ? 1.0f : (sur == clp ? 0.5f : 0.0f), wb = sur > clp ? 1.0f : (sur == clp ? 0.5f : 0.0f);
            const float cg = (r >= 1.0f - clip && r <= 1.0f + clip) ? 1.0f : 0.0f;
            const float g_ll = valid ? -(ad * wa + ad * cg * wb) * r : 0.0f;
            const float g_h = (valid && add_entropy) ? -ent_coeff / (float)N : 0.0f;     // d total / d H_i
            float *d0 = dlogits + s * N * A;
            for (int i = 0; i < N; ++i) {
                float z[PPO_MAX_A], p[PPO_MAX_A], p1[PPO_MAX_A], p2[PPO_MAX_A], p3[PPO_MAX_A], d[PPO_MAX_A];
                float mx = -INFINITY;
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { z[b] = z0[i * A + b]; mx = fmaxf(mx, z[b]); }
                float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p[b] = expf(z[b] - mx); s0 += p[b]; }
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p1[b] = p[b] / s0; s1 += p1[b]; }       // softmax
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) { p2[b] = p1[b] / s1; s2 += p2[b]; }      // _probs: probs / probs.sum
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) p3[b] = p2[b] / s2;                        // Categorical: again
                const int ai = a0[i];
                // d wrt p3: through logits = log(clamp(p3)) (log_prob gather + entropy's logits factor) and entropy's probs factor
                float dot = 0.0f;
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) {
                    const float pc = fminf(fmaxf(p3[b], EPS), 1.0f - EPS);
                    const float lg = logf(pc);
                    const bool in = p3[b] >= EPS && p3[b] <= 1.0f - EPS;
                    const float g_lg = (b == ai ? g_ll : 0.0f) - g_h * p3[b];       // H = - sum lg * p3
                    d[b] = (in ? g_lg / pc : 0.0f) - g_h * lg;
                    dot += d[b] * p3[b];
                }
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) d[b] = (d[b] - dot) / s2;                   // p3 = p2 / s2
                dot = 0.0f;
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) dot += d[b] * p2[b];
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) d[b] = (d[b] - dot) / s1;                   // p2 = p1 / s1
                dot = 0.0f;
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) dot += d[b] * p1[b];
#pragma unroll
                for (int b = 0; b < PPO_MAX_A; ++b) if (b < A) d0[i * A + b] = p1[b] * (d[b] - dot);       // softmax
            }
        }
    }
    // block sums: f64 total, valid count
    __shared__ double sh_t[4];
    __shared__ int sh_c[4];
    for (int o = 32; o > 0; o >>= 1) { part += __shfl_down(part, o); valid_here += __shfl_down(valid_here, o); }
    if ((threadIdx.x & 63) == 0) { sh_t[threadIdx.x >> 6] = part; sh_c[threadIdx.x >> 6] = valid_here; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(total, sh_t[0] + sh_t[1] + sh_t[2] + sh_t[3]);
        atomicAdd((unsigned long long *)count, (unsigned long long)(sh_c[0] + sh_c[1] + sh_c[2] + sh_c[3]));
    }
}

// cm_ppo_mfma.hip: teams of 8 .. 128 on the matrix cores (return 1 = shape not covered)
int agg_bwd_mfma(int S, int N, const float *attn, const float *adj, const float *chan, long ch_stride, const float *hw, const float *out,
                 const float *out_minus, const float *d_out, float *d_attn, float *d_hw, float *d_bias, void *stream);
int attn_bwd_mfma(int S, int N, const float *q, const float *e, const float *m, const float *d_m, const float *add0, const float *add1,
                  float *d_q, float *d_e, void *stream);

// COMMARL_QUAD_BWD=0: teams of 4 take the generic aggregation / attention backward kernels (A/B and test hook)
static bool quad_bwd_on() {
    static const bool v = [] { const char *e = getenv("COMMARL_QUAD_BWD"); return !(e && e[0] == '0'); }();
    return v;
}

}  // namespace cm

using namespace cm;

// ---------------------------------------------------------------------------------------------------------------
// The flat weight copy of a net (transposed [in,out] matrices + biases, what the C-ABI weight structs point into) in ONE
// launch: tensor k is a [rows, cols] row-major source written to dst[k] as its TRANSPOSE ([cols, rows]) when transpose[k], else
// copied as it is.  (The framework's per-tensor copy_ cost 17 launches per net and optimiser step.)
// ---------------------------------------------------------------------------------------------------------------
struct CopyTable { const float *src[40]; float *dst[40]; int rows[40], cols[40], transpose[40]; int count; };

__global__ __launch_bounds__(256) void multi_copy_t_kernel(CopyTable t) {
    const long stride = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < t.count; ++k) {
        const int R = t.rows[k], C = t.cols[k];
        const long n = (long)R * C;
        if (t.transpose[k]) {
            for (long i = i0; i < n; i += stride) { const int c = (int)(i / R), r = (int)(i - (long)c * R); t.dst[k][i] = t.src[k][(long)r * C + c]; }   // dst [C][R]
        } else {
            for (long i = i0; i < n; i += stride) t.dst[k][i] = t.src[k][i];
        }
    }
}

extern "C" int cm_multi_copy_t(int32_t n, const float *const *src, float *const *dst, const int32_t *rows, const int32_t *cols,
                               const int32_t *transpose, void *stream) {
    if (n < 0 || n > 40) return set_error(CM_ERR_ARG, "cm_multi_copy_t: at most 40 tensors per call");
    if (n == 0) return CM_OK;
    if (!src || !dst || !rows || !cols || !transpose) return set_error(CM_ERR_ARG, "cm_multi_copy_t: null argument");
    CopyTable t{};
    t.count = n;
    for (int k = 0; k < n; ++k) {
        if (!src[k] || !dst[k] || rows[k] < 1 || cols[k] < 1) return set_error(CM_ERR_ARG, "cm_multi_copy_t: bad tensor");
        t.src[k] = src[k]; t.dst[k] = dst[k]; t.rows[k] = rows[k]; t.cols[k] = cols[k]; t.transpose[k] = transpose[k];
    }
    hipLaunchKernelGGL(multi_copy_t_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, t);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

static int multi_adam(int32_t n, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq, const int64_t *sizes,
                      float *norm_ws, float max_norm, float lr, float beta1, float beta2, float eps, int32_t step, const float *bc_tab,
                      const int32_t *cursor, int32_t tab_steps, void *stream) {
    if (n < 0 || n > 40) return set_error(CM_ERR_ARG, "cm_multi_adam_step: at most 40 tensors per call");
    if (n == 0) return CM_OK;
    if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes) return set_error(CM_ERR_ARG, "cm_multi_adam_step: bad argument");
    if (bc_tab ? (!cursor || tab_steps < 1) : step < 1) return set_error(CM_ERR_ARG, "cm_multi_adam_step: bad step / bias-correction table");
    TensorTable t{};
    t.count = n;
    for (int k = 0; k < n; ++k) { t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avg[k]; t.v[k] = exp_avg_sq[k]; t.n[k] = (long)sizes[k]; }
    const hipStream_t st = (hipStream_t)stream;
    if (norm_ws) hipLaunchKernelGGL(multi_norm_kernel, dim3(NORM_BLOCKS), dim3(256), 0, st, t, norm_ws);
    const int hs = bc_tab ? 1 : step;
    hipLaunchKernelGGL(multi_adam_kernel, dim3(64), dim3(256), 0, st, t, norm_ws, max_norm, lr, beta1, beta2, eps,
                       (float)(1.0 - pow((double)beta1, (double)hs)), (float)sqrt(1.0 - pow((double)beta2, (double)hs)), bc_tab, cursor,
                       (int)tab_steps);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_multi_adam_step(int32_t n, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                                  const int64_t *sizes, float *norm_ws, float max_norm, float lr, float beta1,
                                  float beta2, float eps, int32_t step, void *stream) {
    return multi_adam(n, params, grads, exp_avg, exp_avg_sq, sizes, norm_ws, max_norm, lr, beta1, beta2, eps, step, nullptr, nullptr, 0, stream);
}

extern "C" void cm_adam_bias_corrections(float beta1, float beta2, int32_t first_step, int32_t n_steps, float *table) {
    for (int i = 0; i < n_steps; ++i) {
        const double s = (double)first_step + i;
        table[2 * i] = (float)(1.0 - pow((double)beta1, s));
        table[2 * i + 1] = (float)sqrt(1.0 - pow((double)beta2, s));
    }
}

extern "C" int cm_multi_adam_step_dev(int32_t n, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                                      const int64_t *sizes, float *norm_ws, float max_norm, float lr, float beta1, float beta2, float eps,
                                      const float *bc_table, const int32_t *cursor, int32_t table_steps, void *stream) {
    return multi_adam(n, params, grads, exp_avg, exp_avg_sq, sizes, norm_ws, max_norm, lr, beta1, beta2, eps, 0, bc_table, cursor, table_steps,
                      stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Gaussian negative log-likelihood of the critic (comm_base_critic.py:59-89: -Normal(values, std.mean()).log_prob(returns).mean()
// with values = the per-agent outputs summed over the team (:110-112) and std = exp(clamp(log_std, min)) of
// gaussian_mlp_module.py:62-188) in ONE launch, and its gradient in one more - the framework spells the same arithmetic as
// ~25 elementwise / reduction launches forward and as many backward, which at the reference's batch size is half of an
// optimiser step's launches.  Normal.log_prob's own expression is kept: var = sigma^2, log_scale = log(sigma),
//   loss = mean_s[(r_s - v_s)^2] / (2 var) + log_scale + log(sqrt(2 pi))
// The sum of squares is accumulated in f64 (block sums + one f64 atomic per block); the last block to finish writes the two
// results and clears the workspace for the next launch.
// ---------------------------------------------------------------------------------------------------------------
struct GaussWs { double sum; unsigned int done; unsigned int pad; };

__device__ __forceinline__ float gauss_sigma(const float *log_std, float min_log_std, int has_min) {
    float ls = log_std[0];
    if (has_min) ls = fmaxf(ls, min_log_std);
    return expf(ls);
}

__global__ __launch_bounds__(256) void gauss_nll_fwd_kernel(long S, int N, const float *__restrict__ per_agent, const float *__restrict__ returns,
                                                            const float *__restrict__ log_std, float min_log_std, int has_min,
                                                            float *__restrict__ out, GaussWs *__restrict__ ws) {
    double part = 0.0;
    for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < S; s += (long)gridDim.x * 256) {
        float v = 0.0f;
        for (int i = 0; i < N; ++i) v += per_agent[s * N + i];
        const float d = returns[s] - v;
        part += (double)(d * d);
    }
    __shared__ double red[256];
    __shared__ bool last;
    red[threadIdx.x] = part;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) { if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k]; __syncthreads(); }
    if (threadIdx.x == 0) {
        atomicAdd(&ws->sum, red[0]);
        __threadfence();
        last = atomicAdd(&ws->done, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        const double total = atomicAdd(&ws->sum, 0.0);             // through L2, after every block's contribution
        const float msq = (float)(total / (double)S);
        const float sigma = gauss_sigma(log_std, min_log_std, has_min);
        const float var = sigma * sigma, log_scale = logf(sigma);
        out[0] = msq / (2.0f * var) + log_scale + 0.918938533204672742f;   // log(sqrt(2 pi))
        out[1] = msq;
        ws->sum = 0.0;
        ws->done = 0u;
    }
}

// d loss / d per_agent[s][i] = -(r_s - v_s) / (var S);  d loss / d log_std = 1 - msq / var (zero below the clamp);  both times *g
__global__ __launch_bounds__(256) void gauss_nll_bwd_kernel(long S, int N, const float *__restrict__ per_agent, const float *__restrict__ returns,
                                                            const float *__restrict__ log_std, float min_log_std, int has_min,
                                                            const float *__restrict__ out, const float *__restrict__ g,
                                                            float *__restrict__ d_per_agent, float *__restrict__ d_log_std) {
    const float sigma = gauss_sigma(log_std, min_log_std, has_min), var = sigma * sigma;
    const float gs = g ? g[0] : 1.0f;
    const float k = -gs / (var * (float)S);
    for (long s = (long)blockIdx.x * 256 + threadIdx.x; s < S; s += (long)gridDim.x * 256) {
        float v = 0.0f;
        for (int i = 0; i < N; ++i) v += per_agent[s * N + i];
        const float dv = k * (returns[s] - v);
        for (int i = 0; i < N; ++i) d_per_agent[s * N + i] = dv;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && d_log_std)
        d_log_std[0] = (has_min && log_std[0] < min_log_std) ? 0.0f : gs * (1.0f - out[1] / var);
}

static size_t agg_lds_fwd(int N, int E) { const int epb = agg_epb(N), rows = epb * N; return ((size_t)rows * (N | 1) + (size_t)rows * (E + 4)) * 4; }
static size_t agg_lds_bwd(int N, int E) {
    const int epb = agg_epb(N), rows = epb * N;
    return ((size_t)rows * (N | 1) * 3 + (size_t)rows * (E + 4) * 2 + rows + TPB) * 4;
}

extern "C" int cm_masked_agg_forward(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                                     const float *chan, int64_t ch_stride, const float *hw, const float *bias,
                                     float *out, void *stream) {
    if (!attn || !hw || !out) return set_error(CM_ERR_ARG, "cm_masked_agg_forward: null argument");
    if (E != 64) return set_error(CM_ERR_ARG, "cm_masked_agg_forward: embedding dim 64 only");
    if (S <= 0) return CM_OK;
    const size_t lds = agg_lds_fwd(N, E);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "cm_masked_agg_forward: n_agents too large");
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    const int epb = agg_epb(N);
    const int blocks = (int)std::min<long>((S + epb - 1) / epb, 256 * 8);
    hipLaunchKernelGGL(agg_fwd_kernel<64>, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, S, N, epb, attn, dist_adj, chan, (long)ch_stride, hw, bias, out);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_masked_agg_backward(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                                      const float *chan, int64_t ch_stride, const float *hw, const float *out, const float *out_minus,
                                      const float *d_out, float *d_attn, float *d_hw, float *d_bias, void *stream) {
    return cm_masked_agg_backward_r(S, N, E, attn, dist_adj, chan, ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias, 1, stream);
}

extern "C" int cm_masked_agg_backward_r(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                                        const float *chan, int64_t ch_stride, const float *hw, const float *out, const float *out_minus,
                                        const float *d_out, float *d_attn, float *d_hw, float *d_bias, int32_t bias_replicas, void *stream) {
    if (!attn || !hw || !out || !d_out || !d_attn || !d_hw) return set_error(CM_ERR_ARG, "cm_masked_agg_backward: null argument");
    if (bias_replicas < 1) return set_error(CM_ERR_ARG, "cm_masked_agg_backward: bias_replicas >= 1 required");
    if (E != 64) return set_error(CM_ERR_ARG, "cm_masked_agg_backward: embedding dim 64 only");
    if (S <= 0) return CM_OK;
    if (N == 4 && quad_bwd_on() && !(((uintptr_t)hw | (uintptr_t)out | (uintptr_t)out_minus | (uintptr_t)d_out | (uintptr_t)d_hw) & 15)) {
        // small batches: fewer workgroups looping (the bias atomics again: 2 500 envs take 11.6 us on 64 workgroups, 19 on 157)
        const long chunks = ((long)S + 15) / 16;
        int blocks = (int)(chunks <= 512 ? std::min<long>(chunks, 64) : std::min<long>(chunks, 2048));
        static const int force = [] { const char *e = getenv("COMMARL_AGG4_BLOCKS"); return e ? atoi(e) : 0; }();
        if (force > 0) blocks = std::min(blocks, force);
        hipLaunchKernelGGL(agg_bwd4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, attn, dist_adj, chan, (long)ch_stride, hw, out,
                           out_minus, d_out, d_attn, d_hw, d_bias, (int)bias_replicas);
        CM_HIP(hipGetLastError());
        return CM_OK;
    }
    if (const int rc = agg_bwd_mfma(S, N, attn, dist_adj, chan, (long)ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias, stream); rc != 1)
        return rc;
    const size_t lds = agg_lds_bwd(N, E);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "cm_masked_agg_backward: n_agents too large");
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    const int epb = agg_epb(N);
    const int blocks = (int)std::min<long>((S + epb - 1) / epb, 256 * 4);
    hipLaunchKernelGGL(agg_bwd_kernel<64>, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, S, N, epb, attn, dist_adj, chan, (long)ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_linear_wgrad(int64_t R, int32_t P, int32_t Q, const float *a, const float *b, float *c, float *colsum_a,
                               void *stream) {
    if (!a || !b || !c) return set_error(CM_ERR_ARG, "cm_linear_wgrad: null argument");
    if (P < 1 || Q < 1 || P > 128 || Q > 128) return set_error(CM_ERR_ARG, "cm_linear_wgrad: 1 <= P, Q <= 128 required");
    if (R <= 0) return CM_OK;
    const int PT = (P + 15) / 16, QT = (Q + 15) / 16, NT = PT * QT;
    const size_t lds = ((size_t)WG_ROWS * (PT * 16 + 16) + (size_t)WG_ROWS * (QT * 16 + 16)) * sizeof(float);
    const long chunks = (R + WG_ROWS - 1) / WG_ROWS;
    const int blocks = (int)std::min<long>(chunks, 512);
    const hipStream_t st = (hipStream_t)stream;
    const int per_wave = (NT + 3) / 4;
    static unsigned long long once = 0;
    if (cm::dev_first(once)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&wgrad_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
#define CM_WG(M) hipLaunchKernelGGL(wgrad_kernel<M>, dim3(blocks), dim3(TPB), lds, st, (long)R, P, Q, a, b, c, colsum_a)
    if (per_wave <= 1) CM_WG(1); else if (per_wave <= 2) CM_WG(2); else if (per_wave <= 4) CM_WG(4);
    else if (per_wave <= 8) CM_WG(8); else CM_WG(16);
#undef CM_WG
    CM_HIP(hipGetLastError());
    return CM_OK;
}

static size_t attn_lds(int N, int E) { const int epb = agg_epb(N), rows = epb * N; return ((size_t)rows * (E + 4) * 2 + (size_t)rows * (N | 1)) * 4; }

extern "C" int cm_attention_forward(int32_t S, int32_t N, int32_t E, const float *q, const float *e, float *m, void *stream) {
    if (!q || !e || !m) return set_error(CM_ERR_ARG, "cm_attention_forward: null argument");
    if (E != 64) return set_error(CM_ERR_ARG, "cm_attention_forward: embedding dim 64 only");
    if (S <= 0) return CM_OK;
    const size_t lds = attn_lds(N, E);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "cm_attention_forward: n_agents too large");
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_fwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    const int epb = agg_epb(N);
    const int blocks = (int)std::min<long>((S + epb - 1) / epb, 256 * 8);
    hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, S, N, epb, q, e, m);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_attention_backward(int32_t S, int32_t N, int32_t E, const float *q, const float *e, const float *m,
                                     const float *d_m, const float *d_e_add0, const float *d_e_add1, float *d_q, float *d_e,
                                     void *stream) {
    if (!q || !e || !m || !d_m || !d_q || !d_e) return set_error(CM_ERR_ARG, "cm_attention_backward: null argument");
    if (d_e == d_e_add0 || d_e == d_e_add1) return set_error(CM_ERR_ARG, "cm_attention_backward: d_e must not alias its addends");
    if (E != 64) return set_error(CM_ERR_ARG, "cm_attention_backward: embedding dim 64 only");
    if (S <= 0) return CM_OK;
    if (N == 4 && quad_bwd_on() && !(((uintptr_t)q | (uintptr_t)e | (uintptr_t)d_e_add0 | (uintptr_t)d_e_add1 | (uintptr_t)d_q | (uintptr_t)d_e) & 15)) {
        const int blocks = (int)std::min<long>((S + 15) / 16, 4096);
        hipLaunchKernelGGL(attn_bwd4_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, q, e, m, d_m, d_e_add0, d_e_add1, d_q, d_e);
        CM_HIP(hipGetLastError());
        return CM_OK;
    }
    if (const int rc = attn_bwd_mfma(S, N, q, e, m, d_m, d_e_add0, d_e_add1, d_q, d_e, stream); rc != 1) return rc;
    const size_t lds = attn_lds(N, E);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "cm_attention_backward: n_agents too large");
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    const int epb = agg_epb(N);
    const int blocks = (int)std::min<long>((S + epb - 1) / epb, 256 * 8);
    hipLaunchKernelGGL(attn_bwd_kernel<64>, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, S, N, epb, q, e, m, d_m, d_e_add0, d_e_add1, d_q, d_e);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_discount_returns(int32_t P, int32_t T, const double *rewards, const int32_t *lens, double gamma,
                                   float *returns, void *stream) {
    if (!rewards || !returns) return set_error(CM_ERR_ARG, "cm_discount_returns: null argument");
    if (P <= 0 || T <= 0) return CM_OK;
    hipLaunchKernelGGL(returns_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, T, rewards, lens, gamma, returns);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_gae(int32_t P, int32_t T, const float *rewards, const float *baselines, const int32_t *lens,
                      float gamma, float lam, int32_t normalize, float eps, float *adv, void *stream) {
    if (!rewards || !baselines || !adv) return set_error(CM_ERR_ARG, "cm_gae: null argument");
    if (P <= 0 || T <= 0) return CM_OK;
    hipLaunchKernelGGL(gae_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, P, T, rewards, baselines, lens, gamma, lam, normalize, eps, adv);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_ppo_surrogate(int32_t P, int32_t T, int32_t N, int32_t A, const float *logits, const int32_t *actions,
                                const float *old_ll, const float *adv, const int32_t *lens, float clip, float ent_coeff,
                                int32_t add_entropy, double *total, int64_t *count, float *dlogits, void *stream) {
    if (!total || !count) return set_error(CM_ERR_ARG, "cm_ppo_surrogate: null argument");
    if (A < 1 || A > PPO_MAX_A || N < 1) return set_error(CM_ERR_ARG, "cm_ppo_surrogate: 1 <= n_actions <= 8 and n_agents >= 1 required");
    CM_HIP(hipMemsetAsync(total, 0, sizeof(double), (hipStream_t)stream));      // an empty minibatch still reports (0, 0)
    CM_HIP(hipMemsetAsync(count, 0, sizeof(int64_t), (hipStream_t)stream));
    if (P <= 0 || T <= 0) return CM_OK;
    if (!logits || !actions || !old_ll || !adv || !lens) return set_error(CM_ERR_ARG, "cm_ppo_surrogate: null argument");
    const long S = (long)P * T;
    hipLaunchKernelGGL(ppo_surrogate_kernel, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, T, N, A, logits, actions,
                       old_ll, adv, lens, clip, ent_coeff, add_entropy, total, (long long *)count, dlogits);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_gauss_nll_forward(int64_t S, int32_t N, const float *per_agent, const float *returns, const float *log_std, float min_log_std,
                                    int32_t has_min, float *out, void *ws, void *stream) {
    if (!per_agent || !returns || !log_std || !out || !ws) return set_error(CM_ERR_ARG, "cm_gauss_nll_forward: null argument");
    if (S < 1 || N < 1) return set_error(CM_ERR_ARG, "cm_gauss_nll_forward: S >= 1 and n_agents >= 1 required");
    const int blocks = (int)std::min<long>((S + 255) / 256, 256);
    hipLaunchKernelGGL(gauss_nll_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)S, N, per_agent, returns, log_std, min_log_std,
                       has_min, out, reinterpret_cast<GaussWs *>(ws));
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_gauss_nll_backward(int64_t S, int32_t N, const float *per_agent, const float *returns, const float *log_std, float min_log_std,
                                     int32_t has_min, const float *out, const float *g, float *d_per_agent, float *d_log_std, void *stream) {
    if (!per_agent || !returns || !log_std || !out || !d_per_agent) return set_error(CM_ERR_ARG, "cm_gauss_nll_backward: null argument");
    if (S < 1 || N < 1) return set_error(CM_ERR_ARG, "cm_gauss_nll_backward: S >= 1 and n_agents >= 1 required");
    const int blocks = (int)std::min<long>((S + 255) / 256, 1024);
    hipLaunchKernelGGL(gauss_nll_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)S, N, per_agent, returns, log_std, min_log_std,
                       has_min, out, g, d_per_agent, d_log_std);
    CM_HIP(hipGetLastError());
    return CM_OK;
}
