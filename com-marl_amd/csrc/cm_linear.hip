// cm_linear.hip - the per-agent dense layers of the PPO update (gfx950), one HBM pass each way:
//
//   cm_linear_act_forward  : y = act(x.W^T + b)                               reads x, writes y
//   cm_linear_act_backward : dz = dy * act'(y);  dx = dz.W;  dW += dz^T.x;  db += colsum(dz)
//                                                                              reads dy, y, x; writes dx
// over R = P*T*N ~ 1e6 agent rows with in / out widths <= 128 (reference: nn.Linear + tanh of
// garage/torch/modules/multi_headed_mlp_module.py:134-149, GraphConvolutionModule's H.W graph_conv_module.py:63,
// AttentionModule.linear_in attention_module.py:36; their autograd is what torch runs as 5 kernels per layer:
// GEMM, tanh, tanh', input-gradient GEMM, weight-gradient GEMM = 5.O + 2.K floats of HBM traffic per row backward
// where this kernel moves 2.O + 2.K).  The update is HBM-bound on these [rows, 64..128] activations, so the lever is
// passes, not FLOPs: f32 MFMA (v_mfma_f32_16x16x4_f32) keeps pace with the streams.
//
// Backward: a workgroup streams 64-row chunks; dz (computed from dy, y on the way in) and x sit in LDS; the weight
// gradient lives in MFMA accumulators for the life of the workgroup (merged with float atomics at the end, as
// cm_linear_wgrad does); dx of the chunk is a second MFMA pass over the same dz tile against W fragments held in
// registers, stored straight to HBM.
#include <algorithm>

#include "cm_internal.h"

namespace cm {
namespace lin {

constexpr int TPB = 256, ROWS = 64;
typedef float v4f __attribute__((ext_vector_type(4)));

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units: |error| <= 2e-7 (as the rollout kernels; pinned at 1e-5
// by tests/test_hip_policy_parity.py::test_fused_linear_act_kernels against an f64 reference)
__device__ __forceinline__ float tanh_exact(float x) {
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// element (k, o) of the weight as stored: layout 0 = nn.Linear [O][K], 1 = GraphConvolution [K][O]
__device__ __forceinline__ float w_at(const float *__restrict__ W, int layout, int K, int O, int k, int o) {
    return layout == 0 ? W[(size_t)o * K + k] : W[(size_t)k * O + o];
}

// A [ROWS x W] row tile from HBM to LDS, zero rows past `rows`; with `yv`: dz = dy * (1 - y^2) on the way in.
// Widths that are a power-of-two number of float4s (16 / 32 / 64 / 128: every hidden width of the nets) take the vector
// path: ALL of a thread's loads (<= 8 float4 per operand) are issued before the first LDS write, so a chunk exposes ONE
// HBM round trip (the row-loop staging of cm_linear_wgrad exposed one per 16-row slice: ~11 us per chunk).  Other widths
// (the observation, the 5 logits, the critic's scalar) are small and go four loads at a time.
template <bool DZ>
__device__ __forceinline__ void stage(float *dst, int stride, const float *__restrict__ src, const float *__restrict__ yv, int W,
                                      long r0, int rows, int tid, const float *__restrict__ src2 = nullptr) {
    const int w4 = W >> 2;
    if ((W & 3) == 0 && (w4 & (w4 - 1)) == 0 && w4 >= 4 && w4 <= 32) {
        const int x = tid & (w4 - 1), rstep = TPB / w4, rb = tid / w4, ni = ROWS / rstep;      // ni = w4 / 4 <= 8
        float4 q[8], y[DZ ? 8 : 1];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (DZ) y[i] = q[i];
            const int r = rb + i * rstep;
            if (i < ni && r < rows) {
                q[i] = reinterpret_cast<const float4 *>(src + (r0 + r) * W)[x];
                if (src2) { const float4 u = reinterpret_cast<const float4 *>(src2 + (r0 + r) * W)[x]; q[i].x += u.x; q[i].y += u.y; q[i].z += u.z; q[i].w += u.w; }
                if (DZ) y[i] = reinterpret_cast<const float4 *>(yv + (r0 + r) * W)[x];
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < ni) {
                float4 v = q[i];
                if (DZ) { v.x *= 1.0f - y[i].x * y[i].x; v.y *= 1.0f - y[i].y * y[i].y; v.z *= 1.0f - y[i].z * y[i].z; v.w *= 1.0f - y[i].w * y[i].w; }
                *reinterpret_cast<float4 *>(dst + (size_t)(rb + i * rstep) * stride + 4 * x) = v;
            }
        }
    } else {
        for (int k0 = tid; k0 < ROWS * W; k0 += 4 * TPB) {
            float v[4], yy[4];
            int off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * TPB, r = k / W, x = k - r * W;
                off[u] = k < ROWS * W ? r * stride + x : -1;
                v[u] = 0.0f; yy[u] = 0.0f;
                if (k < ROWS * W && r < rows) { v[u] = src[(r0 + r) * W + x]; if (src2) v[u] += src2[(r0 + r) * W + x]; if (DZ) yy[u] = yv[(r0 + r) * W + x]; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (off[u] >= 0) dst[off[u]] = DZ ? v[u] * (1.0f - yy[u] * yy[u]) : v[u];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// forward: y[r][o] = act(b[o] + sum_k x[r][k] W(k, o))
// ---------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(TPB) void fwd_kernel(long R, int K, int O, const float *__restrict__ X, const float *__restrict__ W,
                                                 int layout, const float *__restrict__ bias, float *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    const int KT = (K + 15) >> 4, OT = (O + 15) >> 4;
    const int SX = KT * 16 + 4;                           // x tile row stride (words): 16-byte aligned rows
    float *Xs = lds;
    // this wave's output column tiles (<= 2) and their B fragments: b[t][4 kq + j] = W(k = 16 kq + 4 g + j, o = 16 ct + c)
    const int nct = OT >= 4 ? (OT + 3) / 4 : 1;            // tiles per wave
    const int ct0 = OT >= 4 ? wave * nct : wave % OT;
    const int rt_start = OT >= 4 ? 0 : wave / OT, rt_step = OT >= 4 ? 1 : (4 / OT > 0 ? 4 / OT : 1);
    float b[2][32], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int o = (ct0 + t) * 16 + c;
        const bool live = t < nct && (ct0 + t) < OT && o < O;
        bv[t] = (live && bias) ? bias[o] : 0.0f;
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            b[t][kk] = (live && kk < 4 * KT && k < K) ? w_at(W, layout, K, O, k, o) : 0.0f;
        }
    }
    for (int k = tid; k < ROWS * SX; k += TPB) Xs[k] = 0.0f;                       // k-padding columns stay zero
    __syncthreads();
    const long n_chunks = (R + ROWS - 1) / ROWS;
    for (long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const long r0 = ch * ROWS;
        const int rows = (int)min((long)ROWS, R - r0);
        stage<false>(Xs, SX, X, nullptr, K, r0, rows, tid);
        __syncthreads();
        for (int rt = rt_start; rt < ROWS / 16; rt += rt_step) {
            const float4 *pa = reinterpret_cast<const float4 *>(Xs + (size_t)(rt * 16 + c) * SX + 4 * g);
            v4f acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = (v4f){ bv[t], bv[t], bv[t], bv[t] };
#pragma unroll
            for (int kq = 0; kq < 8; ++kq) {
                if (kq < KT) {
                    const float4 a = pa[4 * kq];
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        if (t < nct) {
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[t][4 * kq + 0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[t][4 * kq + 1], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[t][4 * kq + 2], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[t][4 * kq + 3], acc[t], 0, 0, 0);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int o = (ct0 + t) * 16 + c;
                if (t < nct && (ct0 + t) < OT && o < O) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = rt * 16 + 4 * g + r;
                        if (row < rows) Y[(size_t)(r0 + row) * O + o] = ACT ? tanh_exact(acc[t][r]) : acc[t][r];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
template <int MAXT, int ACT>
__global__ __launch_bounds__(TPB, (MAXT == 8 ? 2 : 1)) void bwd_kernel(long R, int K, int O, const float *__restrict__ X, const float *__restrict__ W,
                                                 int layout, const float *__restrict__ DY, const float *__restrict__ DY2,
                                                 const float *__restrict__ Yv,
                                                 float *__restrict__ DX, float *__restrict__ DW, float *__restrict__ DB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    // weight-gradient tiles: C[p][q] with (P, Q) = (O, K) for nn.Linear [O][K], (K, O) for the [K][O] layout
    const int P = layout == 0 ? O : K, Q = layout == 0 ? K : O;
    const int OT = (O + 15) >> 4, KT = (K + 15) >> 4;
    const int PT = layout == 0 ? OT : KT, QT = layout == 0 ? KT : OT, NT = PT * QT;
    const int SZ = OT * 16 + 16, SXs = KT * 16 + 16;       // dz / x tile strides == 16 (mod 32): conflict-free column reads
    float *Zs = lds, *Xs = Zs + (size_t)ROWS * SZ;
    const float *As = layout == 0 ? Zs : Xs, *Bs = layout == 0 ? Xs : Zs;
    const int SA = layout == 0 ? SZ : SXs, SB = layout == 0 ? SXs : SZ;
    v4f acc[MAXT];
    int aoff[MAXT], boff[MAXT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        acc[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
        const int tile = wave + 4 * t;
        const int tl = tile < NT ? tile : 0;
        const int pt = tl / QT, qt = tl - pt * QT;
        aoff[t] = g * SA + c + pt * 16;
        boff[t] = g * SB + c + qt * 16;
    }
    // dx = dz.W: this wave's column tiles of K (<= 2) and their B fragments bw[t][4 oq + j] = W(k = 16 ct + c, o = 16 oq + 4 g + j)
    const int nkt = KT >= 4 ? (KT + 3) / 4 : 1;
    const int kt0 = KT >= 4 ? wave * nkt : wave % KT;
    const int rt_start = KT >= 4 ? 0 : wave / KT, rt_step = KT >= 4 ? 1 : (4 / KT > 0 ? 4 / KT : 1);
    float bw[2][32];
    if (DX) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = (kt0 + t) * 16 + c;
            const bool live = t < nkt && (kt0 + t) < KT && k < K;
#pragma unroll
            for (int oo = 0; oo < 32; ++oo) {
                const int o = 16 * (oo >> 2) + 4 * g + (oo & 3);
                bw[t][oo] = (live && oo < 4 * OT && o < O) ? w_at(W, layout, K, O, k, o) : 0.0f;
            }
        }
    }
    float csum = 0.0f;
    for (int k = tid; k < ROWS * SZ; k += TPB) Zs[k] = 0.0f;
    for (int k = tid; k < ROWS * SXs; k += TPB) Xs[k] = 0.0f;
    __syncthreads();
    const long n_chunks = (R + ROWS - 1) / ROWS;
    for (long ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const long r0 = ch * ROWS;
        const int rows = (int)min((long)ROWS, R - r0);
        stage<ACT != 0>(Zs, SZ, DY, Yv, O, r0, rows, tid, DY2);
        stage<false>(Xs, SXs, X, nullptr, K, r0, rows, tid);
        __syncthreads();
        if (DB && tid < O) { float s = 0.0f; for (int r = 0; r < ROWS; ++r) s += Zs[(size_t)r * SZ + tid]; csum += s; }
        // ---- weight gradient: C[p][q] += sum_r A[r][p] B[r][q] ----
#pragma unroll 2
        for (int kk = 0; kk < ROWS / 4; ++kk) {
            const float *ar = As + (size_t)(4 * kk) * SA, *br = Bs + (size_t)(4 * kk) * SB;
            float av[MAXT], bq[MAXT];
#pragma unroll
            for (int t = 0; t < MAXT; ++t) { av[t] = ar[aoff[t]]; bq[t] = br[boff[t]]; }
#pragma unroll
            for (int t = 0; t < MAXT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bq[t], acc[t], 0, 0, 0);
        }
        // ---- input gradient of this chunk: dx[r][k] = sum_o dz[r][o] W(k, o) ----
        if (DX) {
            for (int rt = rt_start; rt < ROWS / 16; rt += rt_step) {
                const float4 *pa = reinterpret_cast<const float4 *>(Zs + (size_t)(rt * 16 + c) * SZ + 4 * g);
                v4f d[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) d[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
                for (int oq = 0; oq < 8; ++oq) {
                    if (oq < OT) {
                        const float4 a = pa[4 * oq];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            if (t < nkt) {
                                d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw[t][4 * oq + 0], d[t], 0, 0, 0);
                                d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw[t][4 * oq + 1], d[t], 0, 0, 0);
                                d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw[t][4 * oq + 2], d[t], 0, 0, 0);
                                d[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw[t][4 * oq + 3], d[t], 0, 0, 0);
                            }
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int k = (kt0 + t) * 16 + c;
                    if (t < nkt && (kt0 + t) < KT && k < K) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = rt * 16 + 4 * g + r;
                            if (row < rows) DX[(size_t)(r0 + row) * K + k] = d[t][r];
                        }
                    }
                }
            }
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
                if (pp < P && q < Q) atomicAdd(DW + (size_t)pp * Q + q, acc[t][r]);
            }
        }
    }
    if (DB && tid < O) atomicAdd(DB + tid, csum);
}

}  // namespace lin
}  // namespace cm

namespace cm {
int linear_bwd_stream(long R, int K, int O, const float *x, const float *w, int layout, const float *dy, const float *dy2,
                      const float *y, float *dx, float *dw, float *db, void *stream);   // cm_linear_bwd.hip: widths 32 / 64 / 128
int encoder_bwd_chain(long R, int d, const float *obs, const float *a1, const float *e, const float *w2, const float *dy, const float *dy2,
                      float *dw2, float *db2, float *dw1, float *db1, void *stream);
}
using namespace cm;

extern "C" int cm_linear_act_forward(int64_t R, int32_t K, int32_t O, const float *x, const float *w, int32_t w_layout,
                                     const float *bias, int32_t act, float *y, void *stream) {
    if (!x || !w || !y) return set_error(CM_ERR_ARG, "cm_linear_act_forward: null argument");
    if (K < 1 || O < 1 || K > 128 || O > 128) return set_error(CM_ERR_ARG, "cm_linear_act_forward: 1 <= in, out <= 128 required");
    if (w_layout != 0 && w_layout != 1) return set_error(CM_ERR_ARG, "cm_linear_act_forward: w_layout must be 0 ([out,in]) or 1 ([in,out])");
    if (R <= 0) return CM_OK;
    const int KT = (K + 15) / 16;
    const size_t lds = (size_t)lin::ROWS * (KT * 16 + 4) * sizeof(float);
    const long chunks = (R + lin::ROWS - 1) / lin::ROWS;
    const int blocks = (int)std::min<long>(chunks, 512);
    if (act) hipLaunchKernelGGL(lin::fwd_kernel<1>, dim3(blocks), dim3(lin::TPB), lds, (hipStream_t)stream, (long)R, K, O, x, w, w_layout, bias, y);
    else hipLaunchKernelGGL(lin::fwd_kernel<0>, dim3(blocks), dim3(lin::TPB), lds, (hipStream_t)stream, (long)R, K, O, x, w, w_layout, bias, y);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_linear_act_backward(int64_t R, int32_t K, int32_t O, const float *x, const float *w, int32_t w_layout,
                                      const float *dy, const float *dy2, const float *y, float *dx, float *dw, float *db,
                                      void *stream) {
    if (!x || !w || !dy || !dw) return set_error(CM_ERR_ARG, "cm_linear_act_backward: null argument");
    if (K < 1 || O < 1 || K > 128 || O > 128) return set_error(CM_ERR_ARG, "cm_linear_act_backward: 1 <= in, out <= 128 required");
    if (w_layout != 0 && w_layout != 1) return set_error(CM_ERR_ARG, "cm_linear_act_backward: w_layout must be 0 ([out,in]) or 1 ([in,out])");
    if (R <= 0) return CM_OK;
    if (const int rc = linear_bwd_stream(R, K, O, x, w, w_layout, dy, dy2, y, dx, dw, db, stream); rc != 1) return rc;
    const int OT = (O + 15) / 16, KT = (K + 15) / 16, NT = OT * KT;
    const size_t lds = ((size_t)lin::ROWS * (OT * 16 + 16) + (size_t)lin::ROWS * (KT * 16 + 16)) * sizeof(float);
    const long chunks = (R + lin::ROWS - 1) / lin::ROWS;
    int blocks = (int)std::min<long>(chunks, 512);
    static const int force = [] { const char *e = getenv("COMMARL_LIN_BLOCKS"); return e ? atoi(e) : 0; }();
    if (force > 0) blocks = (int)std::min<long>(chunks, force);
    const hipStream_t st = (hipStream_t)stream;
    const int per_wave = (NT + 3) / 4;
    static unsigned long long once = 0;
    if (cm::dev_first(once)) {
#define CM_ATTR(M, A) CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&lin::bwd_kernel<M, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
        CM_ATTR(16, 0); CM_ATTR(8, 0); CM_ATTR(4, 0); CM_ATTR(2, 0); CM_ATTR(1, 0);
        CM_ATTR(16, 1); CM_ATTR(8, 1); CM_ATTR(4, 1); CM_ATTR(2, 1); CM_ATTR(1, 1);
#undef CM_ATTR
    }
#define CM_LB(M) do { if (y) hipLaunchKernelGGL((lin::bwd_kernel<M, 1>), dim3(blocks), dim3(lin::TPB), lds, st, (long)R, K, O, x, w, w_layout, dy, dy2, y, dx, dw, db); \
                      else hipLaunchKernelGGL((lin::bwd_kernel<M, 0>), dim3(blocks), dim3(lin::TPB), lds, st, (long)R, K, O, x, w, w_layout, dy, dy2, y, dx, dw, db); } while (0)
    if (per_wave <= 1) CM_LB(1); else if (per_wave <= 2) CM_LB(2); else if (per_wave <= 4) CM_LB(4);
    else if (per_wave <= 8) CM_LB(8); else CM_LB(16);
#undef CM_LB
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_encoder_backward(int64_t R, int32_t d, const float *obs, const float *a1, const float *e, const float *w2, const float *dy,
                                   const float *dy2, float *dw2, float *db2, float *dw1, float *db1, void *stream) {
    if (!obs || !a1 || !e || !w2 || !dy || !dw2 || !dw1) return set_error(CM_ERR_ARG, "cm_encoder_backward: null argument");
    if (R <= 0) return CM_OK;
    return encoder_bwd_chain(R, d, obs, a1, e, w2, dy, dy2, dw2, db2, dw1, db1, stream);
}
