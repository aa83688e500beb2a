// cm_policy_h_dev.h - fused Comm-DP policy / critic forward with the dense per-agent layers on the gfx950 f16 matrix
// pipe, at f32 accuracy: every operand x is carried as TWO f16 planes
//        hi = f16(x)              lo = f16(x - hi)            x = hi + lo  (error <= max(2^-22 |x|, 2^-25))
// and a 16x16x32 block of a product costs THREE v_mfma_f32_16x16x32_f16 (hi.lo, lo.hi, hi.hi into ONE accumulator; the dropped
// lo.lo term is 2^-22 relative) instead of EIGHT v_mfma_f32_16x16x4_f32.  (Rounds 1-2 stored lo scaled by 2^12 with the cross
// terms in accumulators of their own, joined by a multiply-add per value; round 3 found the f16 MFMA honours subnormals - see
// split2 below.)  Measured on MI355X for the scaled form (tools/micro/layer_split_schemes.hip, 32 rows x 128 -> 64,
// tanh, chained layers, two workgroups per CU): 1871 clk per layer against 3678, max |error| against an f64 reference
// 3.4e-7 against 7.0e-7 for the f32 MFMA form (the f16 instruction accumulates its 32 products more accurately than a
// chain of eight f32 MFMAs does) - far inside the 1e-5 parity bar, and pinned by the reference fixtures.
//
// Same computation, arguments and LDS-resident structure as cm_policy_mfma_dev.h (reference:
// comm_categorical_mlp_policy.py:48-119, comm_base_net.py:80-108, attention_module.py:26-51,
// graph_conv_module.py:51-72, comm_base_critic.py:91-114); what changes:
//   * TRANSPOSED formulation of every dense layer: D[feature][row] = sum_k W[feature][k] X[row][k].  The weights are
//     the A operand (registers, fetched from the operand pack as 16-byte fragments), the activations the B operand -
//     8 consecutive k of one row = one ds_read_b128 per plane - and a lane's four D values are four CONSECUTIVE
//     features of one row: the epilogue (bias, tanh, split) ends in one ds_write_b64 per plane.
//   * activations live in LDS as f16 plane pairs [row][K + 8] (row stride == 4 words mod 32: the 16 rows of a
//     quarter-wave's b128 read land on distinct bank groups); f32 copies exist only where a non-dense consumer needs
//     them (H.Wg for the aggregation, the logits, and E / Q for the small-team VALU attention).
//   * the N x N products of large teams (scores, aggregation) read the same planes: scores on the f16 pipe as well,
//     the aggregation (K = N, f32 attention weights) stays on v_mfma_f32_16x16x4_f32 with swapped operands so that it
//     too ends in four consecutive features per lane.
// COMMARL_POLICY_KERNEL=f32 selects the round-1 all-f32 kernel (cm_policy_mfma_dev.h) for A/B runs.
#pragma once
#include "cm_policy_mfma_dev.h"

namespace cm {
namespace mh {

using mf::FwdArgs;
using mf::CritHead;
using mf::EH; using mf::EMB; using mf::H1; using mf::H2; using mf::H3; using mf::DH; using mf::MAX_ACT;
using mf::v4f;
using mf::fast_tanh;
using mf::lds_barrier;
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
typedef _Float16 h16;

constexpr int SHP = 8;                                  // halves of padding per plane row
constexpr int SF = 68;                                  // f32 row stride (words) of the 64-wide f32 tiles (as mf::SE)

// Round 3: the lo plane is stored UNSCALED, lo = f16(x - hi).  v_mfma_f32_*_f16 honours f16 subnormals on gfx950
// (tools/micro/mfma_f16_subnormal.hip: 2^-24 comes through exactly), so the residual needs no 2^12 lift to survive: its
// absolute error is <= 2^-25 (subnormal spacing) or 2^-11 of itself, i.e. <= max(3e-8, 2^-22 |x|).  Every product term then has the
// same scale and ONE accumulator takes hi.lo + lo.hi + hi.hi (small terms first): no join multiply-add per value, a third of the
// accumulator registers, one multiply less per split.
__host__ __device__ inline void split2(float x, h16 &h, h16 &l) {
    h = (h16)x;
    l = (h16)(x - (float)h);
}
__device__ __forceinline__ float join2(h16 h, h16 l) { return (float)h + (float)l; }

// A pair of f16 planes [rows][stride halves] in LDS
struct Planes {
    h16 *hi, *lo;
    int stride;                                         // halves; (stride / 2) % 32 == 4
};
__device__ __forceinline__ Planes planes_at(void *base, int rows_cap, int K) {
    Planes p;
    p.stride = K + SHP;
    p.hi = reinterpret_cast<h16 *>(base);
    p.lo = p.hi + (size_t)rows_cap * p.stride;
    return p;
}
__host__ __device__ inline size_t planes_bytes(int rows_cap, int K) { return (size_t)rows_cap * (K + SHP) * 2 * sizeof(h16); }

// weight pointers into the f16 operand pack; biases stay plain f32
struct TrunkH { const uint4 *enc1_p; const float *enc_b1; const uint4 *enc2_p; const float *enc_b2; const uint4 *attn_p, *gcn_p; const float *gcn_b; };
struct PolHeadH { const uint4 *h1_p; const float *b1; const uint4 *h2_p; const float *b2; const uint4 *h3_p; const float *b3; const uint4 *h4_p; const float *b4; int n_act; };
struct CritHeadH { const uint4 *d1_p; const float *b1; const float *w2t, *b2; };

enum { OUT_PLANES = 1, OUT_F32 = 2 };

// One dense layer, transposed:  out[row][o] = act(bias[o] + sum_k in[row][k] * W[o][k]),  o < OUT, k < KP (zero padded).
// load() pulls this wave's A fragments (its 16-feature tiles, all k blocks, both planes) one layer AHEAD of run().
// Pack layout (pack_layer_h_kernel): uint4 index ((ct * KB + q) * 2 + plane) * 64 + lane = the 8 halves
// W[16 ct + (lane & 15)][32 q + 8 (lane >> 4) + e].
template <int KP, int OUT, int NW = 4>
struct LayerH {
    static_assert(KP % 32 == 0 && OUT % 16 == 0, "f16 layers are built from 16x16x32 blocks");
    static constexpr int CT = OUT / 16;
    static constexpr int NCT = CT >= NW ? CT / NW : 1;
    static constexpr int KB = KP / 32;
    static constexpr size_t PACK_U4 = (size_t)CT * KB * 2 * 64;
    v8h wh[NCT][KB], wl[NCT][KB];
    float bv[NCT][4];

    __device__ __forceinline__ void load(const uint4 *__restrict__ P, const float *__restrict__ bias, int wave, int lane,
                                         int out_real = OUT) {
        const int ct0 = CT >= NW ? wave * NCT : (wave % CT);
        const int g = lane >> 4;
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = (ct0 + t) * 16 + 4 * g + r;
                bv[t][r] = (bias && o < out_real) ? bias[o] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                const uint4 a = P[(((size_t)(ct0 + t) * KB + q) * 2 + 0) * 64 + lane];
                const uint4 b = P[(((size_t)(ct0 + t) * KB + q) * 2 + 1) * 64 + lane];
                wh[t][q] = __builtin_bit_cast(v8h, a);
                wl[t][q] = __builtin_bit_cast(v8h, b);
            }
        }
    }

    // in: planes; outputs: planes (OUT_PLANES) and / or f32 [row][fstride] (OUT_F32).  Rows of a tile that lie beyond
    // `rows` hold zeros on input and receive finite garbage, exactly as in the f32 kernel.
    template <bool TANH, int OUTS>
    __device__ __forceinline__ void run(const Planes in, const Planes out, float *fout, int fstride, int row_tiles, int wave,
                                        int lane) const {
        const int ct0 = CT >= NW ? wave * NCT : (wave % CT);
        const int rt_start = CT >= NW ? 0 : wave / CT;
        const int rt_step = CT >= NW ? 1 : NW / CT;
        const int c = lane & 15, g = lane >> 4;
        for (int rt = rt_start; rt < row_tiles; rt += 2 * rt_step) {
            const int rtB = rt + rt_step;
            const bool hasB = rtB < row_tiles;
            const int row0 = rt * 16 + c, row1 = (hasB ? rtB : rt) * 16 + c;
            const v8h *ph0 = reinterpret_cast<const v8h *>(in.hi + (size_t)row0 * in.stride + 8 * g);
            const v8h *pl0 = reinterpret_cast<const v8h *>(in.lo + (size_t)row0 * in.stride + 8 * g);
            const v8h *ph1 = reinterpret_cast<const v8h *>(in.hi + (size_t)row1 * in.stride + 8 * g);
            const v8h *pl1 = reinterpret_cast<const v8h *>(in.lo + (size_t)row1 * in.stride + 8 * g);
            v8h xh0[KB], xl0[KB], xh1[KB], xl1[KB];
#pragma unroll
            for (int q = 0; q < KB; ++q) { xh0[q] = ph0[4 * q]; xl0[q] = pl0[4 * q]; xh1[q] = ph1[4 * q]; xl1[q] = pl1[4 * q]; }
            // one accumulator per (feature tile, row tile), bias riding in it: hi.lo + lo.hi + hi.hi
            v4f hh0[NCT], hh1[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                hh0[t] = (v4f){ bv[t][0], bv[t][1], bv[t][2], bv[t][3] };
                hh1[t] = hh0[t];
            }
#define CM_MFH(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)
            if (hasB) {
#pragma unroll
                for (int q = 0; q < KB; ++q)
#pragma unroll
                    for (int t = 0; t < NCT; ++t) {
                        CM_MFH(wh[t][q], xl0[q], hh0[t]); CM_MFH(wh[t][q], xl1[q], hh1[t]);
                        CM_MFH(wl[t][q], xh0[q], hh0[t]); CM_MFH(wl[t][q], xh1[q], hh1[t]);
                        CM_MFH(wh[t][q], xh0[q], hh0[t]); CM_MFH(wh[t][q], xh1[q], hh1[t]);
                    }
            } else {
#pragma unroll
                for (int q = 0; q < KB; ++q)
#pragma unroll
                    for (int t = 0; t < NCT; ++t) {
                        CM_MFH(wh[t][q], xl0[q], hh0[t]); CM_MFH(wl[t][q], xh0[q], hh0[t]); CM_MFH(wh[t][q], xh0[q], hh0[t]);
                    }
            }
#undef CM_MFH
            // D layout: lane (c, g) holds features 16 ct + 4 g + r (r = 0..3) of row (row tile) * 16 + c
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int f0 = (ct0 + t) * 16 + 4 * g;
                float y0[4], y1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = hh0[t][r];
                    y0[r] = TANH ? fast_tanh(v0) : v0;
                    const float v1 = hh1[t][r];
                    y1[r] = TANH ? fast_tanh(v1) : v1;
                }
                if (OUTS & OUT_F32) {
                    *reinterpret_cast<float4 *>(fout + (size_t)row0 * fstride + f0) = make_float4(y0[0], y0[1], y0[2], y0[3]);
                    if (hasB) *reinterpret_cast<float4 *>(fout + (size_t)row1 * fstride + f0) = make_float4(y1[0], y1[1], y1[2], y1[3]);
                }
                if (OUTS & OUT_PLANES) {
                    v4h oh, ol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { h16 h, l; split2(y0[r], h, l); oh[r] = h; ol[r] = l; }
                    *reinterpret_cast<v4h *>(out.hi + (size_t)row0 * out.stride + f0) = oh;
                    *reinterpret_cast<v4h *>(out.lo + (size_t)row0 * out.stride + f0) = ol;
                    if (hasB) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { h16 h, l; split2(y1[r], h, l); oh[r] = h; ol[r] = l; }
                        *reinterpret_cast<v4h *>(out.hi + (size_t)row1 * out.stride + f0) = oh;
                        *reinterpret_cast<v4h *>(out.lo + (size_t)row1 * out.stride + f0) = ol;
                    }
                }
            }
        }
    }

    // The GCN weight product H.Wg in the NON-transposed formulation D[row][feature] = sum_k X[row][k] W[feature][k]:
    // the same fragments with the MFMA operands swapped.  Lane (c, g) then holds rows 16 rt + 4 g + r of feature
    // 16 ct + c - four consecutive SOURCE rows of one feature - and writes them k-contiguously into the per-env
    // transposed planes HWt[env][feature][k = row - env * N] (kstride halves per feature), the layout the aggregation's
    // A operand reads with one ds_read_b128 per 8 source rows.  Rows beyond `rows` are not written; the k-padding
    // [N, Kp) of every feature row is zeroed once by the caller.
    __device__ __forceinline__ void run_hwt(const Planes in, h16 *hwt_hi, h16 *hwt_lo, int kstride, int N, int rows, int envs,
                                            int row_tiles, int wave, int lane) const {
        const int ct0 = CT >= NW ? wave * NCT : (wave % CT);
        const int rt_start = CT >= NW ? 0 : wave / CT;
        const int rt_step = CT >= NW ? 1 : NW / CT;
        const int c = lane & 15, g = lane >> 4;
        for (int rt = rt_start; rt < row_tiles; rt += 2 * rt_step) {
            const int rtB = rt + rt_step;
            const bool hasB = rtB < row_tiles;
            const int row0 = rt * 16 + c, row1 = (hasB ? rtB : rt) * 16 + c;
            const v8h *ph0 = reinterpret_cast<const v8h *>(in.hi + (size_t)row0 * in.stride + 8 * g);
            const v8h *pl0 = reinterpret_cast<const v8h *>(in.lo + (size_t)row0 * in.stride + 8 * g);
            const v8h *ph1 = reinterpret_cast<const v8h *>(in.hi + (size_t)row1 * in.stride + 8 * g);
            const v8h *pl1 = reinterpret_cast<const v8h *>(in.lo + (size_t)row1 * in.stride + 8 * g);
            v8h xh0[KB], xl0[KB], xh1[KB], xl1[KB];
#pragma unroll
            for (int q = 0; q < KB; ++q) { xh0[q] = ph0[4 * q]; xl0[q] = pl0[4 * q]; xh1[q] = ph1[4 * q]; xl1[q] = pl1[4 * q]; }
            v4f hh0[NCT], hh1[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) hh0[t] = hh1[t] = (v4f){ 0.f, 0.f, 0.f, 0.f };
#define CM_MFH(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)
#pragma unroll
            for (int q = 0; q < KB; ++q) {
#pragma unroll
                for (int t = 0; t < NCT; ++t) { CM_MFH(xl0[q], wh[t][q], hh0[t]); if (hasB) CM_MFH(xl1[q], wh[t][q], hh1[t]); }
#pragma unroll
                for (int t = 0; t < NCT; ++t) { CM_MFH(xh0[q], wl[t][q], hh0[t]); if (hasB) CM_MFH(xh1[q], wl[t][q], hh1[t]); }
#pragma unroll
                for (int t = 0; t < NCT; ++t) { CM_MFH(xh0[q], wh[t][q], hh0[t]); if (hasB) CM_MFH(xh1[q], wh[t][q], hh1[t]); }
            }
#undef CM_MFH
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int f = (ct0 + t) * 16 + c;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int r0 = (half == 0 ? rt : rtB) * 16 + 4 * g;          // four consecutive rows, never straddling envs (N % 4 == 0 or one env)
                    if ((half == 0 || hasB) && r0 < rows) {
                        const int e = envs == 1 ? 0 : r0 / N, k = r0 - e * N;
                        v4h oh, ol;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float y = half == 0 ? hh0[t][r] : hh1[t][r];
                            h16 h, l; split2((r0 + r < rows) ? y : 0.0f, h, l); oh[r] = h; ol[r] = l;
                        }
                        const size_t o = ((size_t)e * EMB + f) * kstride + k;
                        *reinterpret_cast<v4h *>(hwt_hi + o) = oh;
                        *reinterpret_cast<v4h *>(hwt_lo + o) = ol;
                    }
                }
            }
        }
    }
};

// scores[i][j] = sum_k Q[i][k] E[j][k] over K = 64 of one 16 x 16 tile pair, on the f16 pipe from the plane pairs:
// A operand = 8 consecutive k of Q row (ra), B operand = 8 consecutive k of E row (rb).  D: lane (c, g) holds
// rows 4g + r of column c - the layout of the f32 instruction, so everything downstream is unchanged.
__device__ __forceinline__ v4f scores_tile_h(const Planes Q, const Planes E, int ra, int rb, int g) {
    const v8h *qh = reinterpret_cast<const v8h *>(Q.hi + (size_t)ra * Q.stride + 8 * g);
    const v8h *ql = reinterpret_cast<const v8h *>(Q.lo + (size_t)ra * Q.stride + 8 * g);
    const v8h *eh = reinterpret_cast<const v8h *>(E.hi + (size_t)rb * E.stride + 8 * g);
    const v8h *el = reinterpret_cast<const v8h *>(E.lo + (size_t)rb * E.stride + 8 * g);
    v8h a0 = qh[0], a1 = qh[4], al0 = ql[0], al1 = ql[4], b0 = eh[0], b1 = eh[4], bl0 = el[0], bl1 = el[4];
    v4f hh = (v4f){ 0.f, 0.f, 0.f, 0.f };
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, bl0, hh, 0, 0, 0);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, b0, hh, 0, 0, 0);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, bl1, hh, 0, 0, 0);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, b1, hh, 0, 0, 0);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, hh, 0, 0, 0);
    hh = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, hh, 0, 0, 0);
    v4f sc = hh;
    return sc;
}

// stage the observation tile [RT*16][KH] into planes (zero k-padding, zero padded rows).  A 16-lane group takes a row,
// its lanes stride the features (coalesced 64-byte segments, no division by the runtime obs dim); every load of the
// tile is issued before the first is consumed - the loop below is fully unrolled up to MAXR row rounds, so the global
// latency is paid once, not once per round.
template <int KH, int TPBW>
__device__ __forceinline__ void stage_obs(const FwdArgs &a, const Planes X, int s0, int rows, int RT, int tid) {
    constexpr int NGR = TPBW / 16, FB = KH / 16, MAXR = 3;       // row groups, feature blocks, unrolled row rounds
    const float *src = a.obs + (size_t)s0 * a.N * a.d;
    const int gq = tid >> 4, sl = tid & 15, nrow = RT * 16;
    float v[MAXR][FB];
#pragma unroll
    for (int rr = 0; rr < MAXR; ++rr) {
        const int r = rr * NGR + gq;
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            const int f = fb * 16 + sl;
            v[rr][fb] = (r < rows && f < a.d) ? src[(size_t)r * a.d + f] : 0.0f;
        }
    }
#pragma unroll
    for (int rr = 0; rr < MAXR; ++rr) {
        const int r = rr * NGR + gq;
        if (r < nrow) {
#pragma unroll
            for (int fb = 0; fb < FB; ++fb) {
                h16 h, l;
                split2(v[rr][fb], h, l);
                X.hi[(size_t)r * X.stride + fb * 16 + sl] = h;
                X.lo[(size_t)r * X.stride + fb * 16 + sl] = l;
            }
        }
    }
    for (int r = MAXR * NGR + gq; r < nrow; r += NGR) {            // tiles of more than MAXR * NGR rows (none of the BASELINE shapes)
#pragma unroll
        for (int fb = 0; fb < FB; ++fb) {
            const int f = fb * 16 + sl;
            h16 h, l;
            split2((r < rows && f < a.d) ? src[(size_t)r * a.d + f] : 0.0f, h, l);
            X.hi[(size_t)r * X.stride + f] = h;
            X.lo[(size_t)r * X.stride + f] = l;
        }
    }
}

// Training forward: a finished LDS tile leaves once, coalesced, as f32 [rows][W] at dst + row0 * W.  Called in the phase
// that CONSUMES the tile (it is stable there); planes are joined to hi + 2^-12 lo - exactly the value the rest of the
// network saw.
template <int W, int TPBW>
__device__ __forceinline__ void dump_planes(const Planes P, float *__restrict__ dst, size_t row0, int rows, int tid) {
    if (!dst) return;
    constexpr int Q4 = W / 4;
    for (int k = tid; k < rows * Q4; k += TPBW) {
        const int r = k / Q4, q = k - r * Q4;
        const v4h h = *reinterpret_cast<const v4h *>(P.hi + (size_t)r * P.stride + 4 * q);
        const v4h l = *reinterpret_cast<const v4h *>(P.lo + (size_t)r * P.stride + 4 * q);
        *reinterpret_cast<float4 *>(dst + (row0 + r) * W + 4 * q) = make_float4(join2(h[0], l[0]), join2(h[1], l[1]), join2(h[2], l[2]), join2(h[3], l[3]));
    }
}
template <int W, int TPBW>
__device__ __forceinline__ void dump_f32(const float *src, int stride, float *__restrict__ dst, size_t row0, int rows, int tid) {
    if (!dst) return;
    constexpr int Q4 = W / 4;
    for (int k = tid; k < rows * Q4; k += TPBW) {
        const int r = k / Q4, q = k - r * Q4;
        *reinterpret_cast<float4 *>(dst + (row0 + r) * W + 4 * q) = *reinterpret_cast<const float4 *>(src + (size_t)r * stride + 4 * q);
    }
}

// H.Wg of the large-team path lives TRANSPOSED ([env][feature][source row] planes): row-major f32 [rows][64] to HBM
template <int TPBW>
__device__ __forceinline__ void dump_hwt(const h16 *hi, const h16 *lo, int kstride, int N, int rows, float *__restrict__ dst, size_t row0, int tid) {
    if (!dst) return;
    for (int k = tid; k < rows * EMB; k += TPBW) {
        const int f = k & (EMB - 1), rj = k >> 6, e = rj / N, j = rj - e * N;
        const size_t at = ((size_t)e * EMB + f) * kstride + j;
        dst[(row0 + rj) * EMB + f] = join2(hi[at], lo[at]);
    }
}

// ---- LDS map (bytes), shared by every path; regions that are never live together overlay each other -------------
//   R1  planes 128 wide  : enc1 output, then the A tile of the aggregation (f32 [rows][NPA]), then head layer 1 output
//   EP  planes 64 wide   : E
//   HP  planes 64 wide   : H_l  (the observation planes overlay HP..T before the encoder has run)
//   T   288 B per row    : Q planes | H.Wg f32 [rows][SF] (large teams: H.Wg TRANSPOSED planes [env][64][Kp + 8]) | head layer 2 planes
//   EF  f32 [rows][SF]   : E in f32 - small-team VALU attention only (0 bytes otherwise)
//   QF  f32 [rows][SF]   : Q in f32 - small-team VALU attention only
//   M   f32 [EPB*N][NP]  : scores / attention (0 bytes on the teams-of-4 path)
//   rs  f32 [rows]       : critic per-agent values
struct LdsMap { size_t r1, ep, hp, t, ef, qf, m, rs, total; };
__host__ __device__ inline LdsMap lds_map(int rows_cap, int epb, int N, int mode /* -1 quad, 0 small, >0 big */) {
    LdsMap o;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 15) & ~(size_t)15; return at; };
    o.r1 = take(planes_bytes(rows_cap, 128));
    o.ep = take(planes_bytes(rows_cap, 64));
    o.hp = take(planes_bytes(rows_cap, 64));
    const size_t kp = (size_t)((N + 31) & ~31);                  // big path: per-env transposed H.Wg planes [64][Kp + 8]
    const size_t hwt = mode > 0 ? (size_t)epb * 64 * (kp + SHP) * 2 * sizeof(h16) : 0;
    o.t = take(planes_bytes(rows_cap, 64) > hwt ? planes_bytes(rows_cap, 64) : hwt);
    o.ef = take(mode == 0 ? (size_t)rows_cap * SF * 4 : 0);
    o.qf = take(mode == 0 ? (size_t)rows_cap * SF * 4 : 0);
    o.m = take(mode < 0 ? 0 : (size_t)epb * N * (N | 1) * 4);
    o.rs = take((size_t)rows_cap * 4);
    o.total = off;
    return o;
}

// HEAD 0 = policy, 1 = critic; KH = obs dim rounded up to 32; MAXMK as in mf::fwd_body (-1 teams of 4, 0 small teams on
// the VALU, > 0 large teams on MFMA tiles); NW = waves per workgroup
// SAVES = false compiles the training-forward stores out (the fused rollout kernels: their per-step copy of the argument
// block then has no dynamically indexed member and stays in registers instead of scratch)
// LATE (teams of 4 only) = every layer's weight fragments are fetched right before the layer instead of one or two layers
// ahead: fewer live registers, for the high-occupancy build that training-size grids use (cm_policy_h.hip)
#define CM_EARLY(x) do { if constexpr (!LATE) { x; } } while (0)
#define CM_JIT(x) do { if constexpr (LATE) { x; } } while (0)
// FULL (teams of 4 only) = every workgroup of the launch holds 8 whole envs (32 rows, 2 row tiles): team size, row counts
// and the LDS map become compile-time constants and the ragged-tile branches of every layer fold away
template <int HEAD, int KH, int MAXMK, int NW = 4, bool SAVES = true, bool LATE = false, bool FULL = false>
__device__ __forceinline__ void fwd_body_h(const FwdArgs &a, const TrunkH &tw, const PolHeadH &ph, const CritHeadH &chd,
                                           unsigned char *lds, int blk, int32_t *act_lds) {
    constexpr int TPBW = 64 * NW, NG = 4 * NW;
    static_assert(NW == 4 || (NW == 8 && MAXMK > 0), "8-wave workgroups are built for the large-team path only");
    static_assert(!LATE || MAXMK < 0, "the late-fetch build exists for the teams-of-4 path");
    constexpr bool quad_path = MAXMK < 0;
    constexpr bool big = MAXMK > 0;
    const int tid = thread_x(), wave = tid >> 6, lane = tid & 63;
    const bool sv_on = SAVES && a.sv_on;
    static_assert(!FULL || MAXMK < 0, "full-workgroup constants are those of the teams-of-4 path");
    const int N = quad_path ? 4 : a.N, L = a.L, NN = N * N, NP = N | 1;
    const int EPBc = FULL ? 8 : a.EPB;
    const int s0 = blk * EPBc;
    const int envs = FULL ? 8 : min(a.EPB, a.S - s0);
    const int rows = envs * N, rows_cap = (EPBc * N + 15) & ~15, RT = (rows + 15) >> 4;
    const LdsMap lm = lds_map(rows_cap, EPBc, N, quad_path ? -1 : (big ? 1 : 0));
    const Planes Ap = planes_at(lds + lm.r1, rows_cap, 128);
    const Planes Ep = planes_at(lds + lm.ep, rows_cap, 64);
    const Planes Hp = planes_at(lds + lm.hp, rows_cap, 64);
    const Planes Tp = planes_at(lds + lm.t, rows_cap, 64);
    const Planes Xp = planes_at(lds + lm.hp, rows_cap, KH);     // over HP (+ T for KH = 96): dead before either is written
    float *HW = reinterpret_cast<float *>(lds + lm.t);          // [rows_cap][SF] (272 B per row <= the 288 of the planes)
    float *EF = reinterpret_cast<float *>(lds + lm.ef);
    float *QF = reinterpret_cast<float *>(lds + lm.qf);
    float *M = reinterpret_cast<float *>(lds + lm.m);
    float *rs = reinterpret_cast<float *>(lds + lm.rs);
    float *Amat = reinterpret_cast<float *>(lds + lm.r1);       // [rows][NPA] f32, NPA <= 132
    const Planes Gp = planes_at(lds + lm.ep, rows_cap, 32);     // head layer 3 output (32 wide) over EP: E is dead by then
    float *LG = reinterpret_cast<float *>(lds + lm.t);          // logits f32 [rows_cap][20] over T (head layer 2 output is
    constexpr int SLG = 20;                                     // consumed before they are written - see the barriers)

    // ---- weights one layer ahead, observation tile first (vector-memory results return in issue order) ----
    stage_obs<KH, TPBW>(a, Xp, s0, rows, RT, tid);
    LayerH<KH, EH, NW> l_enc1;
    l_enc1.load(tw.enc1_p, tw.enc_b1, wave, lane);
    LayerH<EH, EMB, NW> l_enc2;
    CM_EARLY(l_enc2.load(tw.enc2_p, tw.enc_b2, wave, lane));
    const uint32_t draw_step = a.policy_step + (a.step_base ? *a.step_base : 0u);
    lds_barrier();
    if (a.stop == 1) return;
    l_enc1.template run<true, OUT_PLANES>(Xp, Ap, nullptr, 0, RT, wave, lane);
    LayerH<EMB, EMB, NW> l_sq;                               // 64 x 64 square layers: attention, then (non-quad) the hops
    CM_EARLY(l_sq.load(tw.attn_p, nullptr, wave, lane));
    LayerH<EMB, EMB, NW> l_g;                                // quad path: GCN weights, one hop ahead
    if (quad_path && L > 0) CM_EARLY(l_g.load(tw.gcn_p, nullptr, wave, lane));
    lds_barrier();
    if (a.stop == 2) return;
    CM_JIT(l_enc2.load(tw.enc2_p, tw.enc_b2, wave, lane));
    if (quad_path || big) l_enc2.template run<true, OUT_PLANES>(Ap, Ep, nullptr, 0, RT, wave, lane);
    else l_enc2.template run<true, OUT_PLANES | OUT_F32>(Ap, Ep, EF, SF, RT, wave, lane);
    const size_t grow0 = (size_t)s0 * N;                     // first global agent row of this workgroup
    if (sv_on) dump_planes<128, TPBW>(Ap, a.sv_a1, grow0, rows, tid);                        // encoder hidden layer
    LayerH<EMB, HEAD == 0 ? H1 : DH, NW> l_x1;               // first head layer (policy 64 -> 128, critic 64 -> 64)
    LayerH<H1, H2, NW> l_h2;
    if (quad_path) CM_EARLY(l_x1.load(HEAD == 0 ? ph.h1_p : chd.d1_p, HEAD == 0 ? ph.b1 : chd.b1, wave, lane));
    lds_barrier();
    if (a.stop == 3) return;

    if (quad_path) {
        // ---- teams of 4: attention and aggregation in registers (see cm_policy_mfma_dev.h for the layout argument) ----
        const int c = lane & 15, g = lane >> 4, q = lane & 3;
        const bool diag = (c >> 2) == g;
        CM_JIT(l_sq.load(tw.attn_p, nullptr, wave, lane));
        l_sq.template run<false, OUT_PLANES>(Ep, Tp, nullptr, 0, RT, wave, lane);           // Q = E.Wa^T  -> planes in T
        float *HW0 = reinterpret_cast<float *>(lds + lm.r1);                               // H.Wg_l f32: hop parity picks the
        float *HW1 = HW0 + (size_t)rows_cap * SF;                                           // half of R1 (enc1 output is dead)
        if (L > 0) {
            CM_JIT(l_g.load(tw.gcn_p, nullptr, wave, lane));
            l_g.template run<false, OUT_F32>(Ep, Ep, HW0, SF, RT, wave, lane);             // H.Wg_0 (hop 0 reads E)
            if (L > 1) CM_EARLY(l_g.load(tw.gcn_p + LayerH<EMB, EMB, NW>::PACK_U4, nullptr, wave, lane));
        }
        if (HEAD == 0) CM_EARLY(l_h2.load(ph.h2_p, ph.b2, wave, lane));
        if (sv_on) dump_planes<64, TPBW>(Ep, a.sv_e, grow0, rows, tid);
        lds_barrier();
        if (a.stop == 4) return;
        if (sv_on) {
            dump_planes<64, TPBW>(Tp, a.sv_q, grow0, rows, tid);
            if (L > 0) dump_f32<64, TPBW>(HW0, SF, a.sv_hw[0], grow0, rows, tid);
        }
        const int tw_ = (RT > 1) ? (wave & 1) : 0, ch = wave >> 1;
        const int rb = 16 * tw_;
        const v4f sc = scores_tile_h(Tp, Ep, rb + c, rb + c, g);
        float m[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ex = __builtin_amdgcn_exp2f((sc[r] - mf::quad_max(sc[r])) * 1.4426950408889634f);
            m[r] = ex * __builtin_amdgcn_rcpf(mf::quad_sum(ex));
        }
        if (a.stop == 42) return;
        const int env_l = 4 * tw_ + g;
        const bool live = diag && env_l < envs;
        const size_t env_g = (size_t)s0 + min(env_l, envs - 1);
        if (a.attn && ch == 0 && live) {
            float *dst = a.attn + env_g * 16 + q;
            // (trajectory outputs nobody reads during the rollout: streamed past the caches)
            __builtin_nontemporal_store(m[0], dst); __builtin_nontemporal_store(m[1], dst + 4);
            __builtin_nontemporal_store(m[2], dst + 8); __builtin_nontemporal_store(m[3], dst + 12);
        }
        if (a.stop == 5) return;
        for (int l = 0; l < L; ++l) {
            const float *HWl = (l & 1) ? HW1 : HW0;
            const bool last = l == L - 1;
            float v[4], w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                // A = M * Range * Chan_l (comm_base_net.py:101)
                float x = m[r];
                if (a.adj) x *= a.adj[env_g * 16 + 4 * r + q];
                if (a.chan) x *= a.chan[(env_g * L + l) * 16 + 4 * r + q];
                v[r] = x * __builtin_amdgcn_rcpf(mf::quad_sum(x) + 1e-12f);  // :102-103
            }
            mf::quad_transpose(v, q, w);
            // swapped operands: D'[feature][row] = sum_k HW[k][feature] A[row][k]; lane (c, g) ends up with features
            // 32 ch + 16 t + 4 g + r of row rb + c
            v4f acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[t][r] = tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + 32 * ch + 16 * t + 4 * g + r] : 0.0f;
            }
            float hb[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) hb[t][j] = HWl[(size_t)(rb + 4 * g + j) * SF + 32 * ch + 16 * t + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float aop = diag ? w[j] : 0.0f;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[0][j], aop, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(hb[1][j], aop, acc[1], 0, 0, 0);
            }
            if (RT > 1 || (wave & 1) == 0) {                              // single-tile workgroups: odd waves duplicate tile 0
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int f0 = 32 * ch + 16 * t + 4 * g;
                    const size_t o = (size_t)(rb + c) * Hp.stride + f0;
                    v4h eh, el;
                    if (last && !a.no_residual) {
                        eh = *reinterpret_cast<const v4h *>(Ep.hi + (size_t)(rb + c) * Ep.stride + f0);
                        el = *reinterpret_cast<const v4h *>(Ep.lo + (size_t)(rb + c) * Ep.stride + f0);
                    }
                    v4h oh, ol;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float hv = fast_tanh(acc[t][r]);                 // graph_conv_module.py:65-70
                        if (last && !a.no_residual) hv += join2(eh[r], el[r]);   // policy :74-77
                        h16 h, lo_; split2(hv, h, lo_); oh[r] = h; ol[r] = lo_;
                    }
                    *reinterpret_cast<v4h *>(Hp.hi + o) = oh;
                    *reinterpret_cast<v4h *>(Hp.lo + o) = ol;
                }
            }
            lds_barrier();
            if (a.stop == 61 + l) return;
            if (sv_on && l < 4) dump_planes<64, TPBW>(Hp, a.sv_h[l], grow0, rows, tid);      // hop output (last: + residual)
            if (!last) {
                CM_JIT(l_g.load(tw.gcn_p + (size_t)(l + 1) * LayerH<EMB, EMB, NW>::PACK_U4, nullptr, wave, lane));
                l_g.template run<false, OUT_F32>(Hp, Hp, (l & 1) ? HW0 : HW1, SF, RT, wave, lane);     // H.Wg_{l+1}
                if (l + 2 < L) CM_EARLY(l_g.load(tw.gcn_p + (size_t)(l + 2) * LayerH<EMB, EMB, NW>::PACK_U4, nullptr, wave, lane));
                lds_barrier();
                if (sv_on && l + 1 < 4) dump_f32<64, TPBW>((l & 1) ? HW0 : HW1, SF, a.sv_hw[l + 1], grow0, rows, tid);
            }
        }
    } else {
        // ---- general teams ----
        if (HEAD == 0 && !big) l_h2.load(ph.h2_p, ph.b2, wave, lane);
        if (big) l_sq.template run<false, OUT_PLANES>(Ep, Tp, nullptr, 0, RT, wave, lane);      // Q -> planes in T
        else l_sq.template run<false, OUT_F32>(Ep, Ep, QF, SF, RT, wave, lane);                 // Q -> f32 (VALU scores)
        if (L > 0) l_sq.load(tw.gcn_p, nullptr, wave, lane);
        lds_barrier();
        if (a.stop == 4) return;
        if (sv_on) {                                             // E and Q are stable until the hops reuse T
            dump_planes<64, TPBW>(Ep, a.sv_e, grow0, rows, tid);
            if (big) dump_planes<64, TPBW>(Tp, a.sv_q, grow0, rows, tid);
            else dump_f32<64, TPBW>(QF, SF, a.sv_q, grow0, rows, tid);
        }
        if (big) {
            {   // scores on the f16 pipe: 16 x 16 tiles dealt round-robin to waves
                const int c = lane & 15, g = lane >> 4, NT = (N + 15) >> 4, per_env = NT * NT;
                for (int t = wave; t < envs * per_env; t += NW) {
                    const int e = t / per_env, rc = t - e * per_env, rt = rc / NT, ct = rc - rt * NT;
                    const int ra = e * N + min(rt * 16 + c, N - 1), rbb = e * N + min(ct * 16 + c, N - 1);
                    const v4f sc = scores_tile_h(Tp, Ep, ra, rbb, g);
                    const int j = ct * 16 + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = rt * 16 + 4 * g + r;
                        if (i < N && j < N) M[(size_t)(e * N + i) * NP + j] = sc[r];
                    }
                }
            }
            lds_barrier();
            if (a.stop == 41) return;
            {   // softmax: 16 lanes per matrix row, DPP row reductions; the row stays in registers between the passes
                constexpr int JBS = (MAXMK == 25 || MAXMK == 15) ? 5 : 8;          // column blocks per row (N <= 16 * JBS)
                for (int r0 = 0; r0 < rows; r0 += NG) {
                    const int r = min(r0 + (tid >> 4), rows - 1), sl = tid & 15;
                    float *m = M + (size_t)r * NP;
                    float x[JBS];
                    float mx = -INFINITY;
#pragma unroll
                    for (int jb = 0; jb < JBS; ++jb) { const int j = jb * 16 + sl; x[jb] = j < N ? m[j] : -INFINITY; mx = fmaxf(mx, x[jb]); }
                    mx = mf::row16_max(mx);
                    float sum = 0.0f;
#pragma unroll
                    for (int jb = 0; jb < JBS; ++jb) { x[jb] = __builtin_amdgcn_exp2f((x[jb] - mx) * 1.4426950408889634f); sum += x[jb]; }   // exp2(-inf) = 0
                    sum = mf::row16_sum(sum);
                    const float rsum = __builtin_amdgcn_rcpf(sum);
                    if (r0 + (tid >> 4) < rows) {
#pragma unroll
                        for (int jb = 0; jb < JBS; ++jb) { const int j = jb * 16 + sl; if (j < N) m[j] = x[jb] * rsum; }
                    }
                }
            }
        } else {
            for (int k = tid; k < envs * NN; k += TPBW) {
                const int e = k / NN, ij = k - e * NN, i = ij / N, j = ij - i * N;
                const float4 *qv = reinterpret_cast<const float4 *>(QF + (size_t)(e * N + i) * SF);
                const float4 *cv = reinterpret_cast<const float4 *>(EF + (size_t)(e * N + j) * SF);
                float acc = 0.0f;
#pragma unroll
                for (int kk = 0; kk < EMB / 4; ++kk) {
                    const float4 x = qv[kk], y = cv[kk];
                    acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
                }
                M[(size_t)(e * N + i) * NP + j] = acc;
            }
            lds_barrier();
            for (int r = tid; r < rows; r += TPBW) {
                float *m = M + (size_t)r * NP;
                float mx = -INFINITY, sum = 0.0f;
                for (int j = 0; j < N; ++j) mx = fmaxf(mx, m[j]);
                for (int j = 0; j < N; ++j) { const float ex = expf(m[j] - mx); m[j] = ex; sum += ex; }
                for (int j = 0; j < N; ++j) m[j] = m[j] / sum;
            }
        }
        lds_barrier();
        if (a.stop == 42) return;
        if (big && L > 0) {
            // k-padding of the aggregation operands, zeroed once (the hops only ever write k < N): the tail [N, Kp) of every
            // HWt feature row (Q in T is dead: the scores are done) and the A tile's columns the build loop never reaches
            constexpr int JBZ = (MAXMK == 25 || MAXMK == 15) ? 5 : 8;
            const int Kp0 = (N + 31) & ~31, ks0 = Kp0 + SHP, pad = Kp0 - N;
            h16 *zh = reinterpret_cast<h16 *>(lds + lm.t), *zl = zh + (size_t)EPBc * EMB * ks0;
            for (int k = tid; k < envs * EMB * pad; k += TPBW) {
                const int fe = k / pad, j = N + (k - fe * pad);
                zh[(size_t)fe * ks0 + j] = (h16)0.0f; zl[(size_t)fe * ks0 + j] = (h16)0.0f;
            }
            const int z0 = JBZ * 16 < Kp0 ? JBZ * 16 : Kp0, zc = Kp0 - z0;
            const Planes Az = planes_at(lds + lm.r1, rows_cap, Kp0);
            for (int k = tid; k < rows * zc; k += TPBW) {
                const int r = k / zc, j = z0 + (k - r * zc);
                Az.hi[(size_t)r * Az.stride + j] = (h16)0.0f; Az.lo[(size_t)r * Az.stride + j] = (h16)0.0f;
            }
            // (ordered before the first use by the barriers inside hop 0: run_hwt / the A-tile build write disjoint addresses)
        }
        if (a.stop == 5) return;
        for (int l = 0; l < L; ++l) {
            const Planes Hin = (l == 0) ? Ep : Hp;
            const bool last = l == L - 1;
            constexpr int JB = (MAXMK == 25 || MAXMK == 15) ? 5 : 8;
            float mk[MAXMK > 0 ? MAXMK : 1];
            const bool masked = a.adj || a.chan;
            if (MAXMK > 0 && masked) {
                const int gq = tid >> 4, sl = tid & 15;
#pragma unroll
                for (int qq = 0; qq < MAXMK; ++qq) {
                    const int r = (qq / JB) * NG + gq, j = (qq % JB) * 16 + sl;
                    float v = 1.0f;
                    if (r < rows && j < N) {
                        const int e = envs == 1 ? 0 : r / N, i = r - e * N;
                        const size_t off = (size_t)i * N + j;
                        if (a.adj) v = a.adj[(size_t)(s0 + e) * NN + off];
                        if (a.chan) v *= a.chan[((size_t)(s0 + e) * L + l) * NN + off];
                    }
                    mk[qq] = v;
                }
            }
            // large teams: aggregation on the f16 pipe as well - H.Wg_l as transposed planes, the A tile as planes
            const int Kp = (N + 31) & ~31, KBQ = Kp >> 5, kstride = Kp + SHP;
            h16 *hwt_hi = reinterpret_cast<h16 *>(lds + lm.t), *hwt_lo = hwt_hi + (size_t)EPBc * EMB * kstride;
            const Planes Am = planes_at(lds + lm.r1, rows_cap, Kp);                             // A tile [rows][Kp] over R1
            if (big) l_sq.run_hwt(Hin, hwt_hi, hwt_lo, kstride, N, rows, envs, RT, wave, lane);  // H.Wg_l -> HWt planes in T
            else l_sq.template run<false, OUT_F32>(Hin, Hin, HW, SF, RT, wave, lane);           // H.Wg_l -> f32 in T
            if (a.stop == 61 + l) return;
            if (l + 1 < L) l_sq.load(tw.gcn_p + (size_t)(l + 1) * LayerH<EMB, EMB, NW>::PACK_U4, nullptr, wave, lane);
            if (MAXMK > 0 && big) {
                {   // A row = M row * mask, renormalised: 16 lanes per row, DPP row sum, one pass; written as f16 planes
                    const int gq = tid >> 4, sl = tid & 15;
#pragma unroll
                    for (int rbk = 0; rbk < MAXMK / JB; ++rbk) {
                        const int r = rbk * NG + gq;
                        if (rbk * NG < rows) {
                            const bool lv = r < rows;
                            const float *mr = M + (size_t)(lv ? r : 0) * NP;
                            float v[JB];
                            float sum = 0.0f;
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb) {
                                const int j = jb * 16 + sl;
                                v[jb] = (lv && j < N) ? mr[j] * (masked ? mk[rbk * JB + jb] : 1.0f) : 0.0f;
                                sum += v[jb];
                            }
                            const float rden = __builtin_amdgcn_rcpf(mf::row16_sum(sum) + 1e-12f);
                            if (lv) {
#pragma unroll
                                for (int jb = 0; jb < JB; ++jb) {
                                    const int j = jb * 16 + sl;
                                    if (j < Kp) {
                                        h16 h, lo_; split2(j < N ? v[jb] * rden : 0.0f, h, lo_);
                                        Am.hi[(size_t)r * Am.stride + j] = h; Am.lo[(size_t)r * Am.stride + j] = lo_;
                                    }
                                }
                            }
                        }
                    }
                }
                lds_barrier();
                if (a.stop == 51 + l) return;
                if (sv_on && l < 4) dump_hwt<TPBW>(hwt_hi, hwt_lo, kstride, N, rows, a.sv_hw[l], grow0, tid);
                {   // D'[feature][row] = sum_k HWt[feature][k] A[row][k]: both operands are 8 consecutive k per lane (b128).
                    // Wave w owns features 16 (w & 3) .. +15; with 8 waves two waves share them and split the row tiles.
                    constexpr int MAXKB = (MAXMK == 25 || MAXMK == 15) ? 3 : 4;        // N <= 80 -> Kp <= 96; N <= 128 -> Kp <= 128
                    const int c = lane & 15, g = lane >> 4, NT = (N + 15) >> 4;
                    const int ft = wave & 3, f0 = ft * 16 + 4 * g;
                    const int rt0 = 2 * (wave >> 2), rt_stride = 2 * (NW >> 2);
                    float bvr[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bvr[r] = tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + f0 + r] : 0.0f;
                    for (int e = 0; e < envs; ++e) {
                        v8h ah[MAXKB], al[MAXKB];
                        const size_t ao = ((size_t)e * EMB + ft * 16 + c) * kstride + 8 * g;
#pragma unroll
                        for (int q = 0; q < MAXKB; ++q) {
                            const int qc = q < KBQ ? q : KBQ - 1;                     // branch-free: steps past KBQ are skipped below
                            ah[q] = *reinterpret_cast<const v8h *>(hwt_hi + ao + 32 * qc);
                            al[q] = *reinterpret_cast<const v8h *>(hwt_lo + ao + 32 * qc);
                        }
                        for (int rt = rt0; rt < NT; rt += rt_stride) {
                            const bool hasB = rt + 1 < NT;
                            const size_t ra = (size_t)(e * N + min(rt * 16 + c, N - 1)) * Am.stride + 8 * g;
                            const size_t rb2 = (size_t)(e * N + min((hasB ? rt + 1 : rt) * 16 + c, N - 1)) * Am.stride + 8 * g;
                            v8h xh0[MAXKB], xl0[MAXKB], xh1[MAXKB], xl1[MAXKB];
#pragma unroll
                            for (int q = 0; q < MAXKB; ++q) {
                                const int qc = q < KBQ ? q : KBQ - 1;
                                xh0[q] = *reinterpret_cast<const v8h *>(Am.hi + ra + 32 * qc); xl0[q] = *reinterpret_cast<const v8h *>(Am.lo + ra + 32 * qc);
                                xh1[q] = *reinterpret_cast<const v8h *>(Am.hi + rb2 + 32 * qc); xl1[q] = *reinterpret_cast<const v8h *>(Am.lo + rb2 + 32 * qc);
                            }
                            v4f hh0 = (v4f){ bvr[0], bvr[1], bvr[2], bvr[3] }, hh1 = hh0;
#pragma unroll
                            for (int q = 0; q < MAXKB; ++q) {
                                if (q < KBQ) {
                                    hh0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], xl0[q], hh0, 0, 0, 0);
                                    hh1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], xl1[q], hh1, 0, 0, 0);
                                    hh0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q], xh0[q], hh0, 0, 0, 0);
                                    hh1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[q], xh1[q], hh1, 0, 0, 0);
                                    hh0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], xh0[q], hh0, 0, 0, 0);
                                    hh1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[q], xh1[q], hh1, 0, 0, 0);
                                }
                            }
                            // lane (c, g): features f0 .. f0+3 of rows rt*16 + c and (rt+1)*16 + c
#pragma unroll
                            for (int half = 0; half < 2; ++half) {
                                const int i = (rt + half) * 16 + c;
                                if ((half == 0 || hasB) && i < N) {
                                    const size_t row = (size_t)(e * N + i);
                                    v4h eh, el, oh, ol;
                                    if (last && !a.no_residual) {
                                        eh = *reinterpret_cast<const v4h *>(Ep.hi + row * Ep.stride + f0);
                                        el = *reinterpret_cast<const v4h *>(Ep.lo + row * Ep.stride + f0);
                                    }
#pragma unroll
                                    for (int r = 0; r < 4; ++r) {
                                        const float pre = half == 0 ? hh0[r] : hh1[r];
                                        float hv = fast_tanh(pre);
                                        if (last && !a.no_residual) hv += join2(eh[r], el[r]);
                                        h16 h, lo_; split2(hv, h, lo_); oh[r] = h; ol[r] = lo_;
                                    }
                                    *reinterpret_cast<v4h *>(Hp.hi + row * Hp.stride + f0) = oh;
                                    *reinterpret_cast<v4h *>(Hp.lo + row * Hp.stride + f0) = ol;
                                }
                            }
                        }
                    }
                }
                lds_barrier();
                if (sv_on && l < 4) dump_planes<64, TPBW>(Hp, a.sv_h[l], grow0, rows, tid);  // hop output (last: + residual)
                continue;
            }
            // small teams (VALU): masked + renormalised rows of A, then A.(HW) per (row, feature)
            if (N <= 16) {
                for (int r = tid; r < rows; r += TPBW) {
                    const int e = r / N, i = r - e * N;
                    const float *mr = M + (size_t)r * NP;
                    float *ar = Amat + (size_t)r * NP;
                    float sum = 0.0f;
                    for (int j = 0; j < N; ++j) {
                        float v = mr[j];
                        if (a.adj) v *= a.adj[(size_t)(s0 + e) * NN + i * N + j];
                        if (a.chan) v *= a.chan[((size_t)(s0 + e) * L + l) * NN + i * N + j];
                        ar[j] = v; sum += v;
                    }
                    const float den = sum + 1e-12f;
                    for (int j = 0; j < N; ++j) ar[j] = ar[j] / den;
                }
            } else {
                for (int k = tid; k < envs * NN; k += TPBW) {
                    const int e = k / NN, ij = k - e * NN, r = k / N, j = k - r * N;
                    float v = M[(size_t)r * NP + j];
                    if (a.adj) v *= a.adj[(size_t)(s0 + e) * NN + ij];
                    if (a.chan) v *= a.chan[((size_t)(s0 + e) * L + l) * NN + ij];
                    Amat[(size_t)r * NP + j] = v;
                }
                lds_barrier();
                for (int r = tid; r < rows; r += TPBW) {
                    float *ar = Amat + (size_t)r * NP;
                    float sum = 0.0f;
                    for (int j = 0; j < N; ++j) sum += ar[j];
                    const float den = sum + 1e-12f;
                    for (int j = 0; j < N; ++j) ar[j] = ar[j] / den;
                }
            }
            lds_barrier();
            if (sv_on && l < 4) dump_f32<64, TPBW>(HW, SF, a.sv_hw[l], grow0, rows, tid);
            {
                const int o = tid & (EMB - 1), rg = tid >> 6;
                const float bvv = tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + o] : 0.0f;
                for (int r0 = rg * 4; r0 < rows; r0 += 4 * NW) {
                    const int e = r0 / N;
                    const float *hw = HW + (size_t)e * N * SF + o;
                    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                    const float *ar[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) ar[i] = Amat + (size_t)min(r0 + i, rows - 1) * NP;
                    for (int j = 0; j < N; ++j) {
                        const float h = hw[(size_t)j * SF];
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[i] = fmaf(ar[i][j], h, acc[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (r0 + i < rows && (r0 + i) / N == e) {
                            float hv = fast_tanh(acc[i] + bvv);
                            if (last && !a.no_residual) hv += EF[(size_t)(r0 + i) * SF + o];
                            h16 h, lo_; split2(hv, h, lo_);
                            Hp.hi[(size_t)(r0 + i) * Hp.stride + o] = h;
                            Hp.lo[(size_t)(r0 + i) * Hp.stride + o] = lo_;
                        }
                }
            }
            lds_barrier();
            if (sv_on && l < 4) dump_planes<64, TPBW>(Hp, a.sv_h[l], grow0, rows, tid);      // hop output (last: + residual)
        }
    }
    if (a.stop == 6) return;
    // ---- no hops: embeddings[-1] is E itself, so x = E + E with the residual (:74-77), else E ----
    if (L == 0) {
        const float f = a.no_residual ? 1.0f : 2.0f;
        for (int k = tid; k < rows * EMB; k += TPBW) {
            const int r = k >> 6, o = k & 63;
            const float e = join2(Ep.hi[(size_t)r * Ep.stride + o], Ep.lo[(size_t)r * Ep.stride + o]);
            h16 h, lo_; split2(f * e, h, lo_);
            Hp.hi[(size_t)r * Hp.stride + o] = h; Hp.lo[(size_t)r * Hp.stride + o] = lo_;
        }
        lds_barrier();
    }

    // attention output of the general path: written LAST.  Vector-memory operations retire in issue order, so these
    // ~20 KB of stores per env would sit in front of every later mask / weight load if issued right after the softmax;
    // M is not touched by the hops or the head, so it can leave at the very end and drain while the workgroup retires.
    auto store_attention = [&]() {
        if (quad_path || !a.attn) return;
        float *dst = a.attn + (size_t)s0 * NN;
        if (big) {
            for (int r = tid >> 4; r < rows; r += NG)
                for (int j = tid & 15; j < N; j += 16) dst[(size_t)r * N + j] = M[(size_t)r * NP + j];
        } else {
            for (int k = tid; k < envs * NN; k += TPBW) { const int r = k / N, j = k - r * N; dst[k] = M[(size_t)r * NP + j]; }
        }
    };
    if (!quad_path) l_x1.load(HEAD == 0 ? ph.h1_p : chd.d1_p, HEAD == 0 ? ph.b1 : chd.b1, wave, lane);
    if (HEAD == 0) {
        if (big) l_h2.load(ph.h2_p, ph.b2, wave, lane);
        LayerH<H2, H3, NW> l_h3;
        CM_EARLY(l_h3.load(ph.h3_p, ph.b3, wave, lane));
        CM_JIT(l_x1.load(ph.h1_p, ph.b1, wave, lane));
        l_x1.template run<true, OUT_PLANES>(Hp, Ap, nullptr, 0, RT, wave, lane);             // 64 -> 128 into R1
        lds_barrier();
        CM_JIT(l_h2.load(ph.h2_p, ph.b2, wave, lane));
        l_h2.template run<true, OUT_PLANES>(Ap, Tp, nullptr, 0, RT, wave, lane);             // 128 -> 64 into T
        if (sv_on) dump_planes<128, TPBW>(Ap, a.sv_x1, grow0, rows, tid);
        const int A = ph.n_act;
        LayerH<H3, 16, NW> l_h4;                             // 32 -> n_act (<= 8) logits, zero-padded to one feature tile
        CM_EARLY(l_h4.load(ph.h4_p, ph.b4, wave, lane, A));
        lds_barrier();
        CM_JIT(l_h3.load(ph.h3_p, ph.b3, wave, lane));
        l_h3.template run<true, OUT_PLANES>(Tp, Gp, nullptr, 0, RT, wave, lane);             // 64 -> 32 into EP
        if (sv_on) dump_planes<64, TPBW>(Tp, a.sv_x2, grow0, rows, tid);
        lds_barrier();
        if (a.stop == 7) return;
        // the sampler's uniforms do not depend on the logits: the last wave (idle in the 32 -> n_act layer unless the
        // workgroup has NW row tiles) draws them now, off the critical path of the softmax / selection chain below
        const bool draw = (a.actions || act_lds) && !a.greedy;
        if (draw && wave == NW - 1)
            for (int r = lane; r < rows; r += 64) {
                const int e = r / N, i = r - e * N;
                const u32x4 xr = philox4x32_10((uint32_t)(a.env_id_offset + s0 + e), draw_step, SITE_ACTION, (uint32_t)i, a.key0, a.key1);
                rs[r] = unit_f32(xr.x);
            }
        CM_JIT(l_h4.load(ph.h4_p, ph.b4, wave, lane, A));
        l_h4.template run<false, OUT_F32>(Gp, Gp, LG, SLG, RT, wave, lane);                  // logits f32 into T
        if (sv_on) dump_planes<32, TPBW>(Gp, a.sv_x3, grow0, rows, tid);
        lds_barrier();
        if (sv_on && a.sv_out)
            for (int k = tid; k < rows * A; k += TPBW) { const int r = k / A, cc = k - r * A; a.sv_out[(grow0 + r) * A + cc] = LG[(size_t)r * SLG + cc]; }
        for (int r = tid; r < rows; r += TPBW) {
            float lg[MAX_ACT], p[MAX_ACT];
            const float *x = LG + (size_t)r * SLG;
#pragma unroll
            for (int cc = 0; cc < MAX_ACT; ++cc) lg[cc] = (cc < A) ? x[cc] : 0.0f;
            float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
#pragma unroll
            for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) mx = fmaxf(mx, lg[cc]);
#pragma unroll
            for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) { p[cc] = __builtin_amdgcn_exp2f((lg[cc] - mx) * 1.4426950408889634f); sum += p[cc]; }
            const size_t grow = (size_t)s0 * N + r;
            const float rsum = __builtin_amdgcn_rcpf(sum);
#pragma unroll
            for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) {
                const float av = a.avail ? a.avail[grow * A + cc] : 1.0f;
                p[cc] = (p[cc] * rsum) * av; msum += p[cc];
            }
            const float rmsum = __builtin_amdgcn_rcpf(msum);
#pragma unroll
            for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) p[cc] = p[cc] * rmsum;
            if (a.probs) {
#pragma unroll
                for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) __builtin_nontemporal_store(p[cc], a.probs + grow * A + cc);
            }
            if (a.actions || act_lds) {
                int act = 0;
                if (a.greedy) {
                    float best = p[0];
#pragma unroll
                    for (int cc = 1; cc < MAX_ACT; ++cc) if (cc < A && p[cc] > best) { best = p[cc]; act = cc; }
                } else {
                    const float u = rs[r];
                    float acc = 0.0f;
                    int sel = -1, lastc = 0;
#pragma unroll
                    for (int cc = 0; cc < MAX_ACT; ++cc) if (cc < A) { if (p[cc] > 0.0f) lastc = cc; acc += p[cc]; if (sel < 0 && u < acc) sel = cc; }
                    act = sel < 0 ? lastc : sel;
                }
                if (a.actions) a.actions[grow] = act;
                if (act_lds) act_lds[r] = act;
            }
        }
        store_attention();
    } else {
        float *XF = reinterpret_cast<float *>(lds + lm.r1);                                  // critic: tanh(x1) f32 [rows][SF] in R1
        CM_JIT(l_x1.load(chd.d1_p, chd.b1, wave, lane));
        l_x1.template run<true, OUT_F32>(Hp, Hp, XF, SF, RT, wave, lane);
        lds_barrier();
        if (sv_on) dump_f32<64, TPBW>(XF, SF, a.sv_x1, grow0, rows, tid);
        {
            // value head 64 -> 1: eight lanes per row, eight features each (one thread per row left 7/8 of the workgroup
            // idle behind a 64-deep dependent chain of LDS reads)
            static_assert(DH == 64, "value head split assumes 64 hidden units");
            const int part = tid & 7;
            const float4 w0 = *reinterpret_cast<const float4 *>(chd.w2t + 8 * part), w1 = *reinterpret_cast<const float4 *>(chd.w2t + 8 * part + 4);
            for (int r = tid >> 3; r < rows_cap; r += TPBW / 8) {
                const float *x = XF + (size_t)min(r, rows - 1) * SF + 8 * part;
                const float4 x0 = *reinterpret_cast<const float4 *>(x), x1 = *reinterpret_cast<const float4 *>(x + 4);
                float acc = x0.x * w0.x;
                acc = fmaf(x0.y, w0.y, acc); acc = fmaf(x0.z, w0.z, acc); acc = fmaf(x0.w, w0.w, acc);
                acc = fmaf(x1.x, w1.x, acc); acc = fmaf(x1.y, w1.y, acc); acc = fmaf(x1.z, w1.z, acc); acc = fmaf(x1.w, w1.w, acc);
                acc += __shfl_xor(acc, 1);
                acc += __shfl_xor(acc, 2);
                acc += __shfl_xor(acc, 4);
                acc += chd.b2 ? chd.b2[0] : 0.0f;
                if (part == 0 && r < rows) {
                    rs[r] = acc;
                    if (sv_on && a.sv_out) a.sv_out[grow0 + r] = acc;
                }
            }
        }
        lds_barrier();
        for (int e = tid; e < envs; e += TPBW) {
            float v = 0.0f;
            for (int i = 0; i < N; ++i) v += rs[e * N + i];
            a.values[s0 + e] = v;
        }
        store_attention();                                   // the training forward's backward reads it (cm_critic_forward_saved)
    }
}

// ---- host-side helpers ----
inline int kh_of(int d) { const int k = (d + 31) & ~31; return (k == 32 || k == 64 || k == 96) ? k : 0; }
struct PackLayoutH { size_t enc1, enc2, attn, gcn, x1, h2, h3, h4, total; };      // offsets in uint4 (16-byte) units
inline PackLayoutH pack_layout_h(int kh, int L, bool policy) {
    PackLayoutH o{};
    size_t off = 0;
    auto lay = [&](int K, int OUT) { const size_t at = off; off += (size_t)(OUT / 16) * (K / 32) * 2 * 64; return at; };
    o.enc1 = lay(kh, EH);
    o.enc2 = lay(EH, EMB);
    o.attn = lay(EMB, EMB);
    o.gcn = off;
    for (int l = 0; l < L; ++l) lay(EMB, EMB);
    o.x1 = policy ? lay(EMB, H1) : lay(EMB, DH);
    if (policy) { o.h2 = lay(H1, H2); o.h3 = lay(H2, H3); o.h4 = lay(H3, 16); }
    o.total = off;
    return o;
}

}  // namespace mh
}  // namespace cm
