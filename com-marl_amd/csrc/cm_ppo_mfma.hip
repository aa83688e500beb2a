// cm_ppo_mfma.hip - backward of the masked aggregation and of the attention softmax for teams of 8 .. 128 agents (the
// PPO update's N x N operators; reference: comm_base_net.py:99-105 + graph_conv_module.py:63-70, attention_module.py:38-47
// and their autograd).  Per env these are two skinny GEMM pairs,
//     aggregation:  d_hw = A^T . dP   [N,N]^T [N,64]       dA = dP . hw^T   [N,64] [64,N]
//     attention  :  d_q  = dS . e     [N,N]   [N,64]       d_e = dS^T . q   [N,N]^T [N,64]
// which the first-generation kernels (cm_ppo.hip, kept for N < 8) ran as scalar loops over LDS: 53 ms per launch at
// N = 72 (4.9 M agent rows), 55 % of that config's update.  Here a workgroup owns one env at a time, the N x N matrix and
// the two [N,64] tiles sit in LDS once, and all four products run on v_mfma_f32_16x16x4_f32; the softmax / renormalisation
// gradients are finished in the accumulator registers (16-lane row reductions), so the N x N gradient goes straight to HBM.
// The kernels are HBM-bound by construction (256 N^2 FLOP against ~(1024 N + 16 N^2) bytes per env).
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "cm_internal.h"

namespace cm {
namespace pm {

constexpr int TPB = 256, E = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

__host__ __device__ inline int np_of(int N) { return (N + 15) & ~15; }
// row stride of the N x N tile: == 16 (mod 32) words, so that a k-step's four rows (lanes g) fall in four different
// 16-bank groups when the 16 lanes c read along a row
__host__ __device__ inline int sa_of(int NP) { return (NP % 32 == 16) ? NP : NP + 16; }
constexpr int SP = 68;                      // [N,64] tile read with lanes c along ROWS (4 * odd: conflict-free), aggregation
constexpr int SR = 80;                      // [N,64] tile read with lanes c along a row (== 16 mod 32), attention

__device__ __forceinline__ float row_sum16(float v) {       // sum over the 16 lanes c of a lane group (same g)
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ v4f mfma4(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------------------------------------------------------
// masked aggregation backward.  MAXNT = row tiles the accumulator arrays are sized for (N <= 16 MAXNT).
// ---------------------------------------------------------------------------------------------------------------
template <int MAXNT>
__global__ __launch_bounds__(TPB) void agg_bwd_kernel(int S, int N, const float *__restrict__ attn, const float *__restrict__ adj,
                                                     const float *__restrict__ chan, long ch_stride, const float *__restrict__ hw,
                                                     const float *__restrict__ outv, const float *__restrict__ out_minus,
                                                     const float *__restrict__ d_out, float *__restrict__ d_attn,
                                                     float *__restrict__ d_hw, float *__restrict__ d_bias, int stop) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NP = np_of(N), NT = NP >> 4, SA = sa_of(NP), NN = N * N;
    float *A = lds;                                       // [NP][SA] normalised A; rows / columns >= N stay zero
    float *HW = A + (size_t)NP * SA;                      // [NP][SP]
    float *DP = HW + (size_t)NP * SP;                     // [NP][SP] dL/d(pre-activation)
    float *den = DP + (size_t)NP * SP;                    // [NP]
    float *dbs = den + NP;                                // [TPB / 64][64] bias partials
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int o = tid & 63, rg = tid >> 6;
    const float rcpN = 1.0f / (float)N;
    float dbias_acc = 0.0f;
    for (int k = tid; k < NP * SA + 2 * NP * SP; k += TPB) lds[k] = 0.0f;
    __syncthreads();
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const size_t base = (size_t)s * NN, row0 = (size_t)s * N * E;
        // every load of a batch is issued before the first LDS write: one HBM round trip per batch, not per element
        // (the straightforward loop waited ~2 us twenty times per env at N = 72)
        {
            const float4 *h4 = reinterpret_cast<const float4 *>(hw + row0), *y4 = reinterpret_cast<const float4 *>(outv + row0);
            const float4 *u4 = out_minus ? reinterpret_cast<const float4 *>(out_minus + row0) : nullptr;
            const float4 *d4 = reinterpret_cast<const float4 *>(d_out + row0);
            const int n4 = N * (E / 4);                                            // <= 256 MAXNT float4 per operand
            constexpr int U4 = MAXNT <= 5 ? MAXNT : 4;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = tid; k0 < n4; k0 += U4 * TPB) {
                float4 hv[U4], yv[U4], uv[U4], dv[U4];
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int k = k0 + u * TPB;
                    const bool in = k < n4;
                    hv[u] = in ? h4[k] : z4; yv[u] = in ? y4[k] : z4; dv[u] = in ? d4[k] : z4;
                    uv[u] = (in && u4) ? u4[k] : z4;
                }
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int k = k0 + u * TPB;
                    if (k < n4) {
                        const int r = k >> 4, q = k & 15;
                        const float4 y = make_float4(yv[u].x - uv[u].x, yv[u].y - uv[u].y, yv[u].z - uv[u].z, yv[u].w - uv[u].w);
                        const float4 d = dv[u];
                        *reinterpret_cast<float4 *>(HW + (size_t)r * SP + 4 * q) = hv[u];
                        *reinterpret_cast<float4 *>(DP + (size_t)r * SP + 4 * q) =
                            make_float4(d.x * (1.0f - y.x * y.x), d.y * (1.0f - y.y * y.y), d.z * (1.0f - y.z * y.z), d.w * (1.0f - y.w * y.w));   // tanh'
                    }
                }
            }
            constexpr int UA = MAXNT <= 5 ? MAXNT * MAXNT : 16;                   // N^2 / 256 <= MAXNT^2: one batch up to N = 80
            for (int k0 = tid; k0 < NN; k0 += UA * TPB) {
                float av[UA], mv[UA];
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    const int k = k0 + u * TPB;
                    const bool in = k < NN;
                    av[u] = in ? attn[base + k] : 0.0f;
                    mv[u] = (in && adj) ? adj[base + k] : 1.0f;
                    if (in && chan) mv[u] *= chan[(size_t)s * ch_stride + k];
                }
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    const int k = k0 + u * TPB;
                    if (k < NN) {
                        const int i = (int)(((float)k + 0.5f) * rcpN), j = k - i * N;  // exact for k < 2^22
                        A[(size_t)i * SA + j] = av[u] * mv[u];
                    }
                }
            }
        }
        __syncthreads();
        if (stop == 1) { __syncthreads(); continue; }                            // (diagnostic: COMMARL_NXN_STOP)
        for (int r = tid >> 4; r < N; r += TPB / 16) {                           // 16 lanes per row
            float *ar = A + (size_t)r * SA;
            float sum = 0.0f;
            for (int j = tid & 15; j < N; j += 16) sum += ar[j];
            const float dn = row_sum16(sum) + 1e-12f;
            if ((tid & 15) == 0) den[r] = dn;
            for (int j = tid & 15; j < N; j += 16) ar[j] = ar[j] / dn;
        }
        if (d_bias) for (int r = rg; r < N; r += TPB / 64) dbias_acc += DP[(size_t)r * SP + o];
        __syncthreads();
        if (stop == 2) { __syncthreads(); continue; }
        // ---- d_hw = A^T . dP: this wave's 16 output features, every row tile (kept in registers for now) ----
        v4f ahw[MAXNT];
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) ahw[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        for (int kk = 0; kk < NP / 4; ++kk) {                                     // k = source row i = 4 kk + g
            const float b = DP[(size_t)(4 * kk + g) * SP + 16 * wave + c];
            const float *ar = A + (size_t)(4 * kk + g) * SA + c;
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
                if (t < NT) ahw[t] = mfma4(ar[16 * t], b, ahw[t]);
        }
        __syncthreads();                                                          // every wave is done with the whole of A
        if (stop == 3) { __syncthreads(); continue; }
        // ---- dA = dP . hw^T, one row tile per wave at a time; through the renormalisation in registers,
        //      dM_ij = mask_ij (dA_ij - sum_k dA_ik A_ik) / den_i, written over A IN PLACE without the mask (each position by
        //      the lane that read it) so that it can leave in whole rows below ----
        for (int it = wave; it < NT; it += TPB / 64) {
            v4f acc[MAXNT];
#pragma unroll
            for (int t = 0; t < MAXNT; ++t) acc[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
            for (int kk = 0; kk < E / 4; ++kk) {                                  // k = feature 4 kk + g
                const float a = DP[(size_t)(16 * it + c) * SP + 4 * kk + g];
                const float *hr = HW + (size_t)c * SP + 4 * kk + g;
#pragma unroll
                for (int t = 0; t < MAXNT; ++t)
                    if (t < NT) acc[t] = mfma4(a, hr[(size_t)16 * t * SP], acc[t]);
            }
            float tt[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
                if (t < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) tt[r] = fmaf(acc[t][r], A[(size_t)(16 * it + 4 * g + r) * SA + 16 * t + c], tt[r]);
                }
            float rden[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { tt[r] = row_sum16(tt[r]); rden[r] = 1.0f / den[min(16 * it + 4 * g + r, N - 1)]; }
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
                if (t < NT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * it + 4 * g + r < N && 16 * t + c < N)                     // the zero padding stays zero
                            A[(size_t)(16 * it + 4 * g + r) * SA + 16 * t + c] = (acc[t][r] - tt[r]) * rden[r];
                }
        }
        __syncthreads();                                                          // HW and dP are dead, A holds dM (unmasked)
#pragma unroll
        for (int t = 0; t < MAXNT; ++t)
            if (t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) HW[(size_t)(16 * t + 4 * g + r) * SP + 16 * wave + c] = ahw[t][r];
            }
        __syncthreads();
        // ---- both results leave in whole rows: 16-byte stores (the accumulator layout would give 64-byte pieces of rows) ----
        {
            float4 *o4 = reinterpret_cast<float4 *>(d_hw + row0);
            for (int k = tid; k < N * (E / 4); k += TPB) o4[k] = *reinterpret_cast<const float4 *>(HW + (size_t)(k >> 4) * SP + 4 * (k & 15));
            if ((NN & 3) == 0) {
                const float4 *j4 = adj ? reinterpret_cast<const float4 *>(adj + base) : nullptr;
                const float4 *c4 = chan ? reinterpret_cast<const float4 *>(chan + (size_t)s * ch_stride) : nullptr;
                const bool c_al = !chan || ((((size_t)s * ch_stride) & 3) == 0);
                float4 *m4 = reinterpret_cast<float4 *>(d_attn + base);
                for (int k4 = tid; k4 < NN / 4; k4 += TPB) {
                    float4 mk = j4 ? j4[k4] : make_float4(1.f, 1.f, 1.f, 1.f);
                    if (chan) {
                        if (c_al) { const float4 u = c4[k4]; mk.x *= u.x; mk.y *= u.y; mk.z *= u.z; mk.w *= u.w; }
                        else { const float *cp = chan + (size_t)s * ch_stride + 4 * k4; mk.x *= cp[0]; mk.y *= cp[1]; mk.z *= cp[2]; mk.w *= cp[3]; }
                    }
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int k = 4 * k4 + u, i = (int)(((float)k + 0.5f) * rcpN), j = k - i * N;
                        v[u] = A[(size_t)i * SA + j];
                    }
                    m4[k4] = make_float4(mk.x * v[0], mk.y * v[1], mk.z * v[2], mk.w * v[3]);
                }
            } else {
                for (int k = tid; k < NN; k += TPB) {
                    const int i = (int)(((float)k + 0.5f) * rcpN), j = k - i * N;
                    float mk = adj ? adj[base + k] : 1.0f;
                    if (chan) mk *= chan[(size_t)s * ch_stride + k];
                    d_attn[base + k] = mk * A[(size_t)i * SA + j];
                }
            }
        }
        __syncthreads();
    }
    if (d_bias) {
        dbs[rg * 64 + o] = dbias_acc;
        __syncthreads();
        if (rg == 0) {
            float v = 0.0f;
            for (int q = 0; q < TPB / 64; ++q) v += dbs[q * 64 + o];
            atomicAdd(d_bias + o, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// attention softmax backward:  dS = M (dM - rowsum(dM M));  d_q = dS . e;  d_e = dS^T . q (+ two optional addends)
// ---------------------------------------------------------------------------------------------------------------
template <int MAXNT>
__global__ __launch_bounds__(TPB) void attn_bwd_kernel(int S, int N, const float *__restrict__ q, const float *__restrict__ e,
                                                      const float *__restrict__ m, const float *__restrict__ d_m,
                                                      const float *__restrict__ add0, const float *__restrict__ add1,
                                                      float *__restrict__ d_q, float *__restrict__ d_e) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int NP = np_of(N), NT = NP >> 4, SA = sa_of(NP), NN = N * N;
    float *DS = lds;                                      // [NP][SA]; rows / columns >= N stay zero
    float *Q = DS + (size_t)NP * SA;                      // [NP][SR]
    float *K = Q + (size_t)NP * SR;                       // [NP][SR]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    for (int k = tid; k < NP * SA + 2 * NP * SR; k += TPB) lds[k] = 0.0f;
    __syncthreads();
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const size_t base = (size_t)s * NN, row0 = (size_t)s * N * E;
        {
            const float4 *q4 = reinterpret_cast<const float4 *>(q + row0), *e4 = reinterpret_cast<const float4 *>(e + row0);
            const int n4 = N * (E / 4);
            constexpr int U4 = MAXNT <= 5 ? MAXNT : 4;
            const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int k0 = tid; k0 < n4; k0 += U4 * TPB) {
                float4 qv[U4], ev[U4];
#pragma unroll
                for (int u = 0; u < U4; ++u) { const int k = k0 + u * TPB; qv[u] = k < n4 ? q4[k] : z4; ev[u] = k < n4 ? e4[k] : z4; }
#pragma unroll
                for (int u = 0; u < U4; ++u) {
                    const int k = k0 + u * TPB;
                    if (k < n4) {
                        const int r = k >> 4, x = k & 15;
                        *reinterpret_cast<float4 *>(Q + (size_t)r * SR + 4 * x) = qv[u];
                        *reinterpret_cast<float4 *>(K + (size_t)r * SR + 4 * x) = ev[u];
                    }
                }
            }
            // softmax backward, 16 lanes per row, (m, dm) in registers; up to N = 80 every row of a lane group is requested in
            // ONE batch (MAXNT rows x MAXNT columns per lane) instead of one HBM round trip per row
            constexpr int RB = MAXNT <= 5 ? MAXNT : 1;
            for (int rb = tid >> 4; rb < N; rb += RB * (TPB / 16)) {
                float mv[RB][MAXNT], dv[RB][MAXNT];
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) {
                    const int r = rb + ri * (TPB / 16);
                    const float *mr = m + base + (size_t)min(r, N - 1) * N, *dr = d_m + base + (size_t)min(r, N - 1) * N;
#pragma unroll
                    for (int u = 0; u < MAXNT; ++u) { const int j = c + 16 * u; const bool in = r < N && j < N; mv[ri][u] = in ? mr[j] : 0.0f; dv[ri][u] = in ? dr[j] : 0.0f; }
                }
#pragma unroll
                for (int ri = 0; ri < RB; ++ri) {
                    const int r = rb + ri * (TPB / 16);
                    float t = 0.0f;
#pragma unroll
                    for (int u = 0; u < MAXNT; ++u) t = fmaf(dv[ri][u], mv[ri][u], t);
                    t = row_sum16(t);
                    if (r < N) {
#pragma unroll
                        for (int u = 0; u < MAXNT; ++u) { const int j = c + 16 * u; if (j < N) DS[(size_t)r * SA + j] = mv[ri][u] * (dv[ri][u] - t); }
                    }
                }
            }
        }
        __syncthreads();
        // this wave's 16 output features of both products, every row tile
        v4f aq[MAXNT], ae[MAXNT];
#pragma unroll
        for (int t = 0; t < MAXNT; ++t) { aq[t] = (v4f){ 0.f, 0.f, 0.f, 0.f }; ae[t] = aq[t]; }
        for (int kk = 0; kk < NP / 4; ++kk) {
            const float be = K[(size_t)(4 * kk + g) * SR + 16 * wave + c];        // e[j = 4 kk + g]
            const float bq = Q[(size_t)(4 * kk + g) * SR + 16 * wave + c];        // q[i = 4 kk + g]
            const float *dsr = DS + (size_t)c * SA + 4 * kk + g;                   // dS[i = 16 t + c][j = 4 kk + g]
            const float *dst = DS + (size_t)(4 * kk + g) * SA + c;                 // dS[i = 4 kk + g][j = 16 t + c]
#pragma unroll
            for (int t = 0; t < MAXNT; ++t)
                if (t < NT) {
                    aq[t] = mfma4(dsr[(size_t)16 * t * SA], be, aq[t]);
                    ae[t] = mfma4(dst[16 * t], bq, ae[t]);
                }
        }
        __syncthreads();                                                          // Q and K are dead: the results take their place
#pragma unroll
        for (int t = 0; t < MAXNT; ++t)
            if (t < NT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    Q[(size_t)(16 * t + 4 * g + r) * SR + 16 * wave + c] = aq[t][r];
                    K[(size_t)(16 * t + 4 * g + r) * SR + 16 * wave + c] = ae[t][r];
                }
            }
        __syncthreads();
        {   // whole rows out, 16 bytes per lane; the addends of d_e come in the same way
            float4 *q4o = reinterpret_cast<float4 *>(d_q + row0), *e4o = reinterpret_cast<float4 *>(d_e + row0);
            const float4 *a04 = add0 ? reinterpret_cast<const float4 *>(add0 + row0) : nullptr;
            const float4 *a14 = add1 ? reinterpret_cast<const float4 *>(add1 + row0) : nullptr;
            for (int k = tid; k < N * (E / 4); k += TPB) {
                const int r = k >> 4, x = k & 15;
                q4o[k] = *reinterpret_cast<const float4 *>(Q + (size_t)r * SR + 4 * x);
                float4 v = *reinterpret_cast<const float4 *>(K + (size_t)r * SR + 4 * x);
                if (a04) { const float4 u = a04[k]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                if (a14) { const float4 u = a14[k]; v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                e4o[k] = v;
            }
        }
        __syncthreads();
    }
}

static size_t agg_lds(int N) { const int NP = np_of(N); return ((size_t)NP * sa_of(NP) + 2 * (size_t)NP * SP + NP + TPB) * sizeof(float); }
static size_t attn_lds(int N) { const int NP = np_of(N); return ((size_t)NP * sa_of(NP) + 2 * (size_t)NP * SR) * sizeof(float); }

static int blocks_for(int S, size_t lds) {
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / std::max<size_t>(lds, 1)));
    return (int)std::min<long>(S, 256L * per_cu);
}

template <int MAXNT>
static int launch_agg(int S, int N, const float *attn, const float *adj, const float *chan, long ch_stride, const float *hw, const float *out,
                      const float *out_minus, const float *d_out, float *d_attn, float *d_hw, float *d_bias, hipStream_t st) {
    const size_t lds = agg_lds(N);
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&agg_bwd_kernel<MAXNT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    static const int stop = [] { const char *e = getenv("COMMARL_NXN_STOP"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(agg_bwd_kernel<MAXNT>, dim3(blocks_for(S, lds)), dim3(TPB), lds, st, S, N, attn, adj, chan, ch_stride, hw, out, out_minus,
                       d_out, d_attn, d_hw, d_bias, stop);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

template <int MAXNT>
static int launch_attn(int S, int N, const float *q, const float *e, const float *m, const float *d_m, const float *add0, const float *add1,
                       float *d_q, float *d_e, hipStream_t st) {
    const size_t lds = attn_lds(N);
    static unsigned long long once = 0;
    if (cm::dev_first(once)) { CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_bwd_kernel<MAXNT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); }
    hipLaunchKernelGGL(attn_bwd_kernel<MAXNT>, dim3(blocks_for(S, lds)), dim3(TPB), lds, st, S, N, q, e, m, d_m, add0, add1, d_q, d_e);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace pm

static bool mfma_bwd_on() {
    static const bool v = [] { const char *e = getenv("COMMARL_NXN_BWD"); return !(e && e[0] == 'o'); }();   // "old": first-generation kernels
    return v;
}

// Return 1 when the shape is not covered (N < 8, N > 128, unaligned rows): the caller runs the first-generation kernel.
int agg_bwd_mfma(int S, int N, const float *attn, const float *adj, const float *chan, long ch_stride, const float *hw, const float *out,
                 const float *out_minus, const float *d_out, float *d_attn, float *d_hw, float *d_bias, void *stream) {
    if (!mfma_bwd_on() || N < 8 || N > 128) return 1;
    if (((uintptr_t)hw | (uintptr_t)out | (uintptr_t)out_minus | (uintptr_t)d_out | (uintptr_t)d_hw | (uintptr_t)adj | (uintptr_t)d_attn) & 15) return 1;
    if (pm::agg_lds(N) > 160 * 1024) return 1;
    const hipStream_t st = (hipStream_t)stream;
    const int NT = pm::np_of(N) / 16;
    if (NT <= 2) return pm::launch_agg<2>(S, N, attn, adj, chan, ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias, st);
    if (NT <= 5) return pm::launch_agg<5>(S, N, attn, adj, chan, ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias, st);
    return pm::launch_agg<8>(S, N, attn, adj, chan, ch_stride, hw, out, out_minus, d_out, d_attn, d_hw, d_bias, st);
}

int attn_bwd_mfma(int S, int N, const float *q, const float *e, const float *m, const float *d_m, const float *add0, const float *add1,
                  float *d_q, float *d_e, void *stream) {
    if (!mfma_bwd_on() || N < 8 || N > 128) return 1;
    if (((uintptr_t)q | (uintptr_t)e | (uintptr_t)add0 | (uintptr_t)add1 | (uintptr_t)d_q | (uintptr_t)d_e) & 15) return 1;
    if (pm::attn_lds(N) > 160 * 1024) return 1;
    const hipStream_t st = (hipStream_t)stream;
    const int NT = pm::np_of(N) / 16;
    if (NT <= 2) return pm::launch_attn<2>(S, N, q, e, m, d_m, add0, add1, d_q, d_e, st);
    if (NT <= 5) return pm::launch_attn<5>(S, N, q, e, m, d_m, add0, add1, d_q, d_e, st);
    return pm::launch_attn<8>(S, N, q, e, m, d_m, add0, add1, d_q, d_e, st);
}

}  // namespace cm
