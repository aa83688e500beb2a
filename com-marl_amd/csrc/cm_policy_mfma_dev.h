// cm_policy_mfma_dev.h - device code of the fused Comm-DP policy / critic forward on the gfx950 matrix cores (see
// cm_policy_mfma.hip for the design notes and reference citations).  Header so that the stand-alone forward kernel and
// the fused rollout kernel (cm_fused.hip) instantiate the same body.
#pragma once
#include "cm_internal.h"
#include "cm_rng.h"
#include <stdlib.h>

namespace cm {
namespace mf {

constexpr int TPB = 256;
constexpr int EH = 128, EMB = 64, H1 = 128, H2 = 64, H3 = 32, DH = 64;
constexpr int SA = 132;        // row strides (words): multiples of 4 so every lane's A chunk is a 16-byte
constexpr int SE = 68;         // aligned ds_read_b128; the +4 skews rows across banks
constexpr int MAX_ACT = 8;
typedef float v4f __attribute__((ext_vector_type(4)));

// Workgroup barrier for LDS hand-offs only.  __syncthreads() also drains vmcnt(0), i.e. it would wait for the
// NEXT layer's weight prefetch (global loads issued one layer ahead) at every phase boundary and expose the L2
// latency ~13 times per kernel.  Here only the LDS counter is drained; the compiler still places the vmcnt wait
// in front of the first use of a prefetched register.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units: ~8 instructions instead of the
// ~40 of the libm-grade tanhf.  Absolute error <= 2e-7 over the whole range (measured against
// double tanh in tests/test_hip_policy_parity.py), far inside the 1e-5 parity bar; saturates to
// +-1 without NaN (exp -> inf gives 1 - 0, exp -> 0 gives 1 - 2).
#ifndef CM_DIAG
#define CM_DIAG 0
#endif
__device__ __forceinline__ float fast_tanh(float x) {
#if (CM_DIAG & 2)
    return x * 0.5f;                                                       // diagnostic: no transcendental work
#endif
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);     // exp(2x) = 2^(2x*log2 e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// weight pointers: the *_p members point into the operand pack (B fragments), biases and the critic's 64 -> 1 output
// row stay plain
struct TrunkW { const float *enc1_p, *enc_b1, *enc2_p, *enc_b2, *attn_p, *gcn_p, *gcn_b; };
struct PolHead { const float *h1_p, *b1, *h2_p, *b2, *h3_p, *b3, *h4_p, *b4; int n_act; };
struct CritHead { const float *d1_p, *b1, *w2t, *b2; };
struct FwdArgs {
    int S, N, d, L, EPB;
    const float *obs, *avail, *adj, *chan;
    uint32_t key0, key1, policy_step;
    const uint32_t *step_base;
    int env_id_offset, greedy, no_residual;
    int32_t *actions;
    float *probs, *attn, *values;
    // training forward (cm_policy_forward_saved / cm_critic_forward_saved; teams-of-4 path of cm_policy_h_dev.h): when sv_on,
    // every activation the backward pass needs leaves LDS once, as f32 [R = S*N rows, width]: encoder hidden (128), E (64),
    // Q (64), per hop H.Wg_l (64) and the hop's output (64; the last one includes the residual), head hidden layers
    // (128 | 64 | 32; critic: 64), logits [R, n_act] or the critic's per-agent value [R]
    int sv_on;
    float *sv_a1, *sv_e, *sv_q, *sv_hw[4], *sv_h[4], *sv_x1, *sv_x2, *sv_x3, *sv_out;
    int stop;          // diagnostic: return after phase `stop` (0 = run everything); COMMARL_FWD_STOP
    unsigned long long *probe;   // diagnostic (COMMARL_FWD_PROBE): [blocks][NPROBE] shader-clock stamps of thread 0
};
constexpr int NPROBE = 16;
#define CM_PROBE(i) do { if (a.probe && tid == 0) a.probe[(size_t)blk * NPROBE + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// One dense layer  out[r][o] = act(sum_k in[r][k] * Wt[k][o] + bias[o]),  r < 16*row_tiles, o < OUT,
// k < kreal <= KPAD.  load() pulls this wave's B fragments (weights) into registers - it is issued one
// layer AHEAD of run() so the L2 latency hides under the previous layer's MFMAs; run() streams the A
// operand from LDS.
template <int KPAD, int OUT, int NW = 4>
struct Layer {
    static constexpr int CT = OUT / 16;                // column tiles of the layer
    static constexpr int NCT = CT >= NW ? CT / NW : 1; // column tiles per wave (NW waves per workgroup)
    static constexpr int KS = KPAD / 4;                // k-steps
    float b[NCT][KS];
    float bv[NCT];

    // k-slot mapping of the 16x16x4 MFMA: lane group g = lane>>4 supplies k = 16*kq + 4*g + j at step
    // (kq, j).  Each lane's A operands are then 4 CONTIGUOUS words per kq (one ds_read_b128), all of a
    // row tile's reads are issued up front and the MFMAs run back to back behind counted waits.
    // The B operands come from the operand pack (pack_layer below): [column tile][kq][lane][4 = j], zero padding
    // baked in, so a wave fetches one unconditional, fully coalesced 1 KB load per (tile, kq).
    static constexpr int PACK_FLOATS = CT * (KS / 4) * 64 * 4;
    __device__ __forceinline__ void load(const float *__restrict__ P, const float *__restrict__ bias, int wave, int lane,
                                         int out_real = OUT) {
        const int ct0 = CT >= NW ? wave * NCT : (wave % CT);
        const float4 *p4 = reinterpret_cast<const float4 *>(P);
#pragma unroll
        for (int t = 0; t < NCT; ++t) {
            const int col = (ct0 + t) * 16 + (lane & 15);
            bv[t] = (bias && col < out_real) ? bias[col] : 0.0f;
#pragma unroll
            for (int kq = 0; kq < KS / 4; ++kq) {
#if (CM_DIAG & 1)
                const float4 v = make_float4(0.001f, 0.002f, 0.003f, 0.004f);        // diagnostic: no weight loads
#else
                const float4 v = p4[((size_t)(ct0 + t) * (KS / 4) + kq) * 64 + lane];
#endif
                b[t][4 * kq + 0] = v.x; b[t][4 * kq + 1] = v.y; b[t][4 * kq + 2] = v.z; b[t][4 * kq + 3] = v.w;
            }
        }
    }

    template <bool TANH>
    __device__ __forceinline__ void run(const float *in, int in_stride, float *out, int out_stride, int row_tiles,
                                        int wave, int lane) const {
        const int ct0 = CT >= NW ? wave * NCT : (wave % CT);
        const int rt_start = CT >= NW ? 0 : wave / CT;
        const int rt_step = CT >= NW ? 1 : NW / CT;
        const int c = lane & 15, g = lane >> 4;
        for (int rt = rt_start; rt < row_tiles; rt += 2 * rt_step) {
            const int rtB = rt + rt_step;
            const bool hasB = rtB < row_tiles;
            const float4 *pa = reinterpret_cast<const float4 *>(in + (size_t)(rt * 16 + c) * in_stride + 4 * g);
            const float4 *pb = reinterpret_cast<const float4 *>(in + (size_t)((hasB ? rtB : rt) * 16 + c) * in_stride + 4 * g);
            float4 a0[KS / 4], a1[KS / 4];
#pragma unroll
            for (int kq = 0; kq < KS / 4; ++kq) { a0[kq] = pa[4 * kq]; a1[kq] = pb[4 * kq]; }
            v4f acc0[NCT], acc1[NCT];
#pragma unroll
            for (int t = 0; t < NCT; ++t) { acc0[t] = (v4f){ bv[t], bv[t], bv[t], bv[t] }; acc1[t] = acc0[t]; }   // bias rides in the accumulator:
                                                                       // column = lane&15 is the same for a lane's 4 rows
            if (hasB) {
#pragma unroll
                for (int kq = 0; kq < KS / 4; ++kq) {
                    const float x0[4] = { a0[kq].x, a0[kq].y, a0[kq].z, a0[kq].w };
                    const float x1[4] = { a1[kq].x, a1[kq].y, a1[kq].z, a1[kq].w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int t = 0; t < NCT; ++t) {
#if (CM_DIAG & 4)
                            acc0[t][j] += x0[j] * b[t][4 * kq + j];                  // diagnostic: no matrix-core work
                            acc1[t][j] += x1[j] * b[t][4 * kq + j];
#else
                            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], b[t][4 * kq + j], acc0[t], 0, 0, 0);
                            acc1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], b[t][4 * kq + j], acc1[t], 0, 0, 0);
#endif
                        }
                    }
                }
            } else {                                    // odd tile count: the unpaired last tile runs a single chain
#pragma unroll
                for (int kq = 0; kq < KS / 4; ++kq) {
                    const float x0[4] = { a0[kq].x, a0[kq].y, a0[kq].z, a0[kq].w };
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int t = 0; t < NCT; ++t) {
#if (CM_DIAG & 4)
                            acc0[t][j] += x0[j] * b[t][4 * kq + j];
#else
                            acc0[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], b[t][4 * kq + j], acc0[t], 0, 0, 0);
#endif
                        }
                    }
                }
            }
            // D layout: col = lane&15, row = 4*(lane>>4) + reg
#pragma unroll
            for (int t = 0; t < NCT; ++t) {
                const int col = (ct0 + t) * 16 + c;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v0 = acc0[t][r];
                    out[(size_t)(rt * 16 + 4 * g + r) * out_stride + col] = TANH ? fast_tanh(v0) : v0;
                    if (hasB) {
                        const float v1 = acc1[t][r];
                        out[(size_t)(rtB * 16 + 4 * g + r) * out_stride + col] = TANH ? fast_tanh(v1) : v1;
                    }
                }
            }
        }
    }
};

// Reductions over a 16-lane DPP row (all 16 lanes receive the result): quad swaps, then row_half_mirror and
// row_mirror.  Pure VALU+DPP - __shfl_xor lowers to ds_bpermute (an LDS-crossbar round trip per step), which made
// a 64-lane shuffle reduction cost ~1 us per matrix row.
__device__ __forceinline__ float dpp_f(float v, int ctrl_sel) {
    const int x = __float_as_int(v);
    int y;
    switch (ctrl_sel) {
    case 0: y = __builtin_amdgcn_update_dpp(x, x, 0xB1, 0xF, 0xF, false); break;    // quad_perm [1,0,3,2]
    case 1: y = __builtin_amdgcn_update_dpp(x, x, 0x4E, 0xF, 0xF, false); break;    // quad_perm [2,3,0,1]
    case 2: y = __builtin_amdgcn_update_dpp(x, x, 0x141, 0xF, 0xF, false); break;   // row_half_mirror
    default: y = __builtin_amdgcn_update_dpp(x, x, 0x140, 0xF, 0xF, false); break;  // row_mirror
    }
    return __int_as_float(y);
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f(v, 0)); v = fmaxf(v, dpp_f(v, 1)); v = fmaxf(v, dpp_f(v, 2)); v = fmaxf(v, dpp_f(v, 3));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f(v, 0); v += dpp_f(v, 1); v += dpp_f(v, 2); v += dpp_f(v, 3);
    return v;
}

// Per-env N x N products on the matrix cores (used when N >= 16; the VALU forms below are kept for small
// teams where an N x N tile would be mostly padding: N < 32).
// scores[e][i][j] = sum_k Q[e*N+i][k] * K[e*N+j][k]   (K = 64): 16x16 output tiles dealt round-robin to waves;
// both operands are 16-byte LDS reads of one activation row.
__device__ __forceinline__ void scores_mfma(const float *Q, const float *K, float *M, int N, int NP, int envs, int wave, int lane,
                                            int nw = 4) {
    const int c = lane & 15, g = lane >> 4, NT = (N + 15) >> 4, per_env = NT * NT;
    for (int t = wave; t < envs * per_env; t += nw) {
        const int e = t / per_env, rc = t - e * per_env, rt = rc / NT, ct = rc - rt * NT;
        const int ra = min(rt * 16 + c, N - 1), rb = min(ct * 16 + c, N - 1);          // clamped rows: results masked below
        const float4 *pa = reinterpret_cast<const float4 *>(Q + (size_t)(e * N + ra) * SE + 4 * g);
        const float4 *pb = reinterpret_cast<const float4 *>(K + (size_t)(e * N + rb) * SE + 4 * g);
        float4 av[EMB / 16], bw[EMB / 16];
#pragma unroll
        for (int kq = 0; kq < EMB / 16; ++kq) { av[kq] = pa[4 * kq]; bw[kq] = pb[4 * kq]; }
        v4f acc = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int kq = 0; kq < EMB / 16; ++kq) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kq].x, bw[kq].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kq].y, bw[kq].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kq].z, bw[kq].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kq].w, bw[kq].w, acc, 0, 0, 0);
        }
        const int j = ct * 16 + c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = rt * 16 + 4 * g + r;
            if (i < N && j < N) M[(size_t)(e * N + i) * NP + j] = acc[r];
        }
    }
}

// H'[e*N+i][o] = tanh(sum_j A[e*N+i][j] * HW[e*N+j][o] + b[o]) (+ E residual on the last hop).  Wave w owns
// output columns 16w..16w+15: its B fragment (a K x 16 slab of HW, K = N padded to 16) is loaded once per env
// and reused by every row tile; A rows are 16-byte reads of the zero-padded [rows][NPA] tile.
template <int MAXKS>
__device__ __forceinline__ void agg_mfma(const float *A, int NPA, const float *HW, const float *bias, const float *Eres,
                                         float *Hout, int N, int envs, int wave, int lane, int nw = 4) {
    const int c = lane & 15, g = lane >> 4, NT = (N + 15) >> 4, KQ = NT;            // k-steps of 16
    const int col = (wave & 3) * 16 + c;              // 4 column tiles; with 8 waves two waves share one and split the row tiles
    const int rt0 = 2 * (wave >> 2), rt_stride = 2 * (nw >> 2);
    const float bv = bias ? bias[col] : 0.0f;
    for (int e = 0; e < envs; ++e) {
        float b[MAXKS];
#pragma unroll
        for (int kk = 0; kk < MAXKS; ++kk) {
            const int k = 16 * (kk >> 2) + 4 * g + (kk & 3);
            b[kk] = (kk < 4 * KQ && k < N) ? HW[(size_t)(e * N + k) * SE + col] : 0.0f;
        }
        for (int rt = rt0; rt < NT; rt += rt_stride) {   // two row tiles in flight: independent accumulator chains
            const bool hasB = rt + 1 < NT;
            const int ra = min(rt * 16 + c, N - 1), rb = min((hasB ? rt + 1 : rt) * 16 + c, N - 1);
            const float4 *pa = reinterpret_cast<const float4 *>(A + (size_t)(e * N + ra) * NPA + 4 * g);
            const float4 *pb = reinterpret_cast<const float4 *>(A + (size_t)(e * N + rb) * NPA + 4 * g);
            float4 xa[MAXKS / 4], xb[MAXKS / 4];
#pragma unroll
            for (int kq = 0; kq < MAXKS / 4; ++kq) {    // branch-free: steps past KQ re-read step KQ-1 and meet b == 0
                const int kc = kq < KQ ? kq : KQ - 1;
                xa[kq] = pa[4 * kc]; xb[kq] = pb[4 * kc];
            }
            v4f acc0 = (v4f){ 0.f, 0.f, 0.f, 0.f }, acc1 = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
            for (int kq = 0; kq < MAXKS / 4; ++kq) {
                if (kq < KQ) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kq].x, b[4 * kq + 0], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[kq].x, b[4 * kq + 0], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kq].y, b[4 * kq + 1], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[kq].y, b[4 * kq + 1], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kq].z, b[4 * kq + 2], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[kq].z, b[4 * kq + 2], acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[kq].w, b[4 * kq + 3], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[kq].w, b[4 * kq + 3], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i0 = rt * 16 + 4 * g + r, i1 = i0 + 16;
                if (i0 < N) {
                    const size_t o = (size_t)(e * N + i0) * SE + col;
                    const float hv = fast_tanh(acc0[r] + bv);
                    Hout[o] = Eres ? Eres[o] + hv : hv;
                }
                if (hasB && i1 < N) {
                    const size_t o = (size_t)(e * N + i1) * SE + col;
                    const float hv = fast_tanh(acc1[r] + bv);
                    Hout[o] = Eres ? Eres[o] + hv : hv;
                }
            }
        }
    }
}

// ---- teams of 4 (the headline config): attention and aggregation without leaving the registers -------------------
// A 16-row activation tile holds 4 whole envs, so the per-env 4 x 4 score blocks are the DIAGONAL 4 x 4 blocks of the
// tile's 16 x 16 product Q.E^T: one MFMA chain per tile.  In the D layout lane (c = lane&15, g = lane>>4) holds
// rows 4g..4g+3 of column c, i.e. for the lanes with (c>>2) == g - whole DPP quads - register r is
// score[env g][i = r][j = c&3]: softmax over j, the mask product and the row renormalisation are quad reductions,
// and a 4 x 4 transpose inside the quad turns the result into the A operand of the aggregation MFMA
// (block-diagonal 16 x 16 A times the tile's 16 rows of H.W).  No LDS round trip, no workgroup barrier.
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
    const int x = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(x, x, CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, quad_dpp<0xB1>(v)); return fmaxf(v, quad_dpp<0x4E>(v)); }
__device__ __forceinline__ float quad_sum(float v) { v += quad_dpp<0xB1>(v); return v + quad_dpp<0x4E>(v); }
// lane q of a quad holds column q of a 4 x 4 block in v[0..3] (v[r] = block[r][q]); returns row q: w[j] = block[q][j]
__device__ __forceinline__ void quad_transpose(const float v[4], int q, float w[4]) {
#define CM_QT(JJ, CTRL) { const float t0 = quad_dpp<CTRL>(v[0]), t1 = quad_dpp<CTRL>(v[1]), t2 = quad_dpp<CTRL>(v[2]), \
                                       t3 = quad_dpp<CTRL>(v[3]); w[JJ] = q == 0 ? t0 : (q == 1 ? t1 : (q == 2 ? t2 : t3)); }
    CM_QT(0, 0x00) CM_QT(1, 0x55) CM_QT(2, 0xAA) CM_QT(3, 0xFF)
#undef CM_QT
}

__host__ __device__ inline size_t lds_floats(int rows_pad, int epb, int N) {
    const int NP = N | 1;
    return (size_t)rows_pad * (SA + 3 * SE) + (size_t)epb * N * NP + rows_pad;
}

// HEAD 0 = policy, 1 = critic; KPAD = obs dim rounded up to 16; MAXMK = mask elements per thread held in
// registers across a hop's MFMAs (0 for small teams: N*N <= MAXMK*256)
template <int HEAD, int KPAD, int MAXMK, int NW = 4>
// `lds` = the workgroup's dynamic LDS block, `blk` = workgroup index (envs blk*EPB ..), `act_lds` = optional [rows] LDS
// array that also receives the sampled actions (fused rollout kernel: the env step of the same envs reads them there)
__device__ __forceinline__ void fwd_body(const FwdArgs &a, const TrunkW &tw, const PolHead &ph, const CritHead &chd, float *lds,
                                         int blk, int32_t *act_lds) {
    constexpr int TPBW = 64 * NW, NG = 4 * NW;      // threads and 16-lane groups per workgroup (NW = 4 or 8 waves)
    static_assert(NW == 4 || (NW == 8 && MAXMK > 0), "8-wave workgroups are built for the large-team path only");
    const int tid = thread_x(), wave = tid >> 6, lane = tid & 63;
    const int N = a.N, d = a.d, L = a.L, NN = N * N, NP = N | 1;
    const int s0 = blk * a.EPB;
    const int envs = min(a.EPB, a.S - s0);
    const int rows = envs * N, rows_cap = (a.EPB * N + 15) & ~15, RT = (rows + 15) >> 4;
    float *bufA = lds;                                  // [rows_cap][SA]
    float *E = bufA + (size_t)rows_cap * SA;            // [rows_cap][SE]
    float *H = E + (size_t)rows_cap * SE;               // [rows_cap][SE]
    float *T = H + (size_t)rows_cap * SE;               // [rows_cap][SE]
    float *M = T + (size_t)rows_cap * SE;               // [EPB*N][NP]
    float *rs = M + (size_t)a.EPB * N * NP;             // [rows_cap]
    float *X = H;                                       // obs staging [rows_cap][SX] over H|T
    constexpr int SX = 2 * SE;                          // 136 words: d <= 128
    CM_PROBE(0);

    // Global-memory schedule.  Vector-memory results return in issue order, so whatever is needed first is issued
    // first and every later operand is requested one or two phases before its use: the observation tile, then the
    // encoder weights; small vectors used deep inside the kernel (GCN biases, the sampler's step counter) ride along
    // here instead of exposing an L2 round trip in the middle of a phase.
    // MAXMK < 0 selects the register-resident path for teams of 4 (host guarantees N == 4 and <= 32 rows per
    // workgroup): its own instantiation, so it neither carries the general path's code nor burdens it with the
    // early head prefetch its free registers allow
    constexpr bool quad_path = MAXMK < 0;
    constexpr bool EARLY = quad_path;
    constexpr int OBSR = 8;
    const float *src = a.obs + (size_t)s0 * N * d;
    const int obs_total = RT * 16 * KPAD;
    const bool obs_pre = obs_total <= OBSR * TPBW;
    float ox[OBSR];
    if (obs_pre) {
#pragma unroll
        for (int qq = 0; qq < OBSR; ++qq) {
            const int k = tid + qq * TPBW, r = k / KPAD, f = k - r * KPAD;
            ox[qq] = (k < obs_total && r < rows && f < d) ? src[(size_t)r * d + f] : 0.0f;
        }
    }
    Layer<KPAD, EH, NW> l_enc1;
    l_enc1.load(tw.enc1_p, tw.enc_b1, wave, lane);
    Layer<EH, EMB, NW> l_enc2;
    l_enc2.load(tw.enc2_p, tw.enc_b2, wave, lane);
    const uint32_t draw_step = a.policy_step + (a.step_base ? *a.step_base : 0u);
    float gbias[4];                                     // quad path: GCN biases of hops 0/1 for this wave's two column tiles
#pragma unroll
    for (int k = 0; k < 4; ++k)
        gbias[k] = (quad_path && tw.gcn_b && (k >> 1) < L) ? tw.gcn_b[(size_t)(k >> 1) * EMB + 32 * (wave >> 1) + 16 * (k & 1) + (lane & 15)] : 0.0f;
    // ---- stage observations (coalesced), zero the k-padding and the padded rows ----
    if (obs_pre) {
#pragma unroll
        for (int qq = 0; qq < OBSR; ++qq) {
            const int k = tid + qq * TPBW, r = k / KPAD, f = k - r * KPAD;
            if (k < obs_total) X[(size_t)r * SX + f] = ox[qq];
        }
    } else {
        for (int k = tid; k < obs_total; k += TPBW) {
            const int r = k / KPAD, f = k - r * KPAD;
            X[(size_t)r * SX + f] = (r < rows && f < d) ? src[(size_t)r * d + f] : 0.0f;
        }
    }
    Layer<EMB, EMB, NW> l_sq;                               // 64x64 square layers: attention, then the GCN hops
    l_sq.load(tw.attn_p, nullptr, wave, lane);
    lds_barrier();
    if (a.stop == 1) return;
    CM_PROBE(1);
    l_enc1.template run<true>(X, SX, bufA, SA, RT, wave, lane);
    Layer<EMB, EMB, NW> l_g;                                // quad path: GCN weights, one hop ahead
    if (quad_path && L > 0) l_g.load(tw.gcn_p, nullptr, wave, lane);
    lds_barrier();
    if (a.stop == 2) return;
    CM_PROBE(2);
    l_enc2.template run<true>(bufA, SA, E, SE, RT, wave, lane);
    Layer<EMB, HEAD == 0 ? H1 : DH, NW> l_x1;               // first head layer (policy 64 -> 128, critic 64 -> 64)
    if (EARLY) l_x1.load(HEAD == 0 ? ph.h1_p : chd.d1_p, HEAD == 0 ? ph.b1 : chd.b1, wave, lane);
    Layer<H1, H2, NW> l_h2;
    lds_barrier();
    if (a.stop == 3) return;
    CM_PROBE(3);
    if (quad_path) {
        const int c = lane & 15, g = lane >> 4, q = lane & 3;
        const bool diag = (c >> 2) == g;                                 // this lane's quad holds a diagonal block
        l_sq.template run<false>(E, SE, T, SE, RT, wave, lane);          // Q = E.Wa^T
        if (L > 0) {
            l_g.template run<false>(E, SE, bufA, SA, RT, wave, lane);    // H.Wg_0 (hop 0 reads E): same barrier as Q
            if (L > 1) l_g.load(tw.gcn_p + (size_t)EMB * EMB, nullptr, wave, lane);
        }
        if (EARLY && HEAD == 0) l_h2.load(ph.h2_p, ph.b2, wave, lane);
        lds_barrier();
        if (a.stop == 4) return;
        CM_PROBE(4);
        // Wave w owns row tile tw = w & 1 (4 envs) and the output columns 32*(w>>1) .. +31 of every hop: it computes the
        // scores of ITS tile (16 MFMAs, shared with the wave of the other column half instead of an LDS round trip and
        // a barrier), keeps attention / A in registers and aggregates two column tiles per hop.
        const int tw_ = (RT > 1) ? (wave & 1) : 0, ch = wave >> 1;
        const int rb = 16 * tw_;                                          // first row of this wave's tile
        v4f sc = (v4f){ 0.f, 0.f, 0.f, 0.f };
        {
            const float4 *q0 = reinterpret_cast<const float4 *>(T + (size_t)(rb + c) * SE + 4 * g);
            const float4 *e0 = reinterpret_cast<const float4 *>(E + (size_t)(rb + c) * SE + 4 * g);
            float4 qa[EMB / 16], ea[EMB / 16];
#pragma unroll
            for (int kq = 0; kq < EMB / 16; ++kq) { qa[kq] = q0[4 * kq]; ea[kq] = e0[4 * kq]; }
#pragma unroll
            for (int kq = 0; kq < EMB / 16; ++kq) {
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kq].x, ea[kq].x, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kq].y, ea[kq].y, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kq].z, ea[kq].z, sc, 0, 0, 0);
                sc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[kq].w, ea[kq].w, sc, 0, 0, 0);
            }
        }
        // softmax over j (the quad), exp / reciprocal on the hardware units (1 ulp: far inside the 1e-5 bar)
        float m[4];                                                      // m[r] = attention[env 4*tw+g][i = r][j = q]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ex = __builtin_amdgcn_exp2f((sc[r] - quad_max(sc[r])) * 1.4426950408889634f);
            m[r] = ex * __builtin_amdgcn_rcpf(quad_sum(ex));
        }
        if (a.stop == 42) return;
        CM_PROBE(5);
        const int env_l = 4 * tw_ + g;                                    // this quad's env (valid on diag lanes)
        const bool live = diag && env_l < envs;
        const size_t env_g = (size_t)s0 + min(env_l, envs - 1);
        if (a.attn && ch == 0 && live) {                                  // one wave per tile stores it: 64 B per env
            float *dst = a.attn + env_g * 16 + q;
            dst[0] = m[0]; dst[4] = m[1]; dst[8] = m[2]; dst[12] = m[3];
        }
        if (a.stop == 5) return;
        for (int l = 0; l < L; ++l) {
            const float *HW = (l & 1) ? T : bufA;                        // hop l's H.Wg_l (Q in T is dead after the scores)
            const int hws = (l & 1) ? SE : SA;
            const bool last = l == L - 1;
            float v[4], w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                // A = M * Range * Chan_l (comm_base_net.py:101)
                float x = m[r];
                if (a.adj) x *= a.adj[env_g * 16 + 4 * r + q];
                if (a.chan) x *= a.chan[(env_g * L + l) * 16 + 4 * r + q];
                v[r] = x * __builtin_amdgcn_rcpf(quad_sum(x) + 1e-12f);  // :102-103
            }
            quad_transpose(v, q, w);
            v4f acc[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {                                // bias rides in the accumulator
                const float bv = l < 2 ? (l == 0 ? gbias[t] : gbias[2 + t])
                                       : (tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + 32 * ch + 16 * t + c] : 0.0f);
                acc[t] = (v4f){ bv, bv, bv, bv };
            }
            float hb[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) hb[t][j] = HW[(size_t)(rb + 4 * g + j) * hws + 32 * ch + 16 * t + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float aop = diag ? w[j] : 0.0f;
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, hb[0][j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, hb[1][j], acc[1], 0, 0, 0);
            }
            if (RT > 1 || (wave & 1) == 0) {                              // single-tile workgroups: odd waves duplicate tile 0
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int col = 32 * ch + 16 * t + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const size_t o = (size_t)(rb + 4 * g + r) * SE + col;
                        const float hv = fast_tanh(acc[t][r]);           // graph_conv_module.py:65-70
                        H[o] = (last && !a.no_residual) ? E[o] + hv : hv;        // policy :74-77
                    }
                }
            }
            lds_barrier();
            if (a.stop == 61 + l) return;
            CM_PROBE(6 + 2 * l);
            if (!last) {
                l_g.template run<false>(H, SE, (l & 1) ? bufA : T, (l & 1) ? SA : SE, RT, wave, lane);   // H.Wg_{l+1}
                if (l + 2 < L) l_g.load(tw.gcn_p + (size_t)(l + 2) * EMB * EMB, nullptr, wave, lane);
                lds_barrier();
                CM_PROBE(7 + 2 * l);
            }
        }
    } else {
    if (EARLY && HEAD == 0) l_h2.load(ph.h2_p, ph.b2, wave, lane);
    l_sq.template run<false>(E, SE, T, SE, RT, wave, lane);                                  // Q = E.Wa^T
    if (L > 0) l_sq.load(tw.gcn_p, nullptr, wave, lane);
    lds_barrier();
    if (a.stop == 4) return;
    // ---- attention scores + softmax: N x N per env ----
    const bool big = MAXMK > 0;                         // matrix-core path for the N x N products (N >= 16)
    if (big) {
        scores_mfma(T, E, M, N, NP, envs, wave, lane, NW);
        lds_barrier();
        if (a.stop == 41) return;
        for (int r0 = 0; r0 < rows; r0 += NG) {   // 16 lanes per matrix row, DPP row reductions
            const int r = min(r0 + (tid >> 4), rows - 1), sl = tid & 15;
            float *m = M + (size_t)r * NP;
            float mx = -INFINITY;
            for (int j = sl; j < N; j += 16) mx = fmaxf(mx, m[j]);
            mx = row16_max(mx);
            float sum = 0.0f;
            for (int j = sl; j < N; j += 16) { const float ex = __builtin_amdgcn_exp2f((m[j] - mx) * 1.4426950408889634f); sum += ex; if (r0 + (tid >> 4) < rows) m[j] = ex; }   // hardware exp2 / rcp: 1 ulp
            sum = row16_sum(sum);
            const float rsum = __builtin_amdgcn_rcpf(sum);
            if (r0 + (tid >> 4) < rows)
                for (int j = sl; j < N; j += 16) m[j] = m[j] * rsum;
        }
    } else {
        for (int k = tid; k < envs * NN; k += TPBW) {
            const int e = k / NN, ij = k - e * NN, i = ij / N, j = ij - i * N;
            const float4 *q = reinterpret_cast<const float4 *>(T + (size_t)(e * N + i) * SE);
            const float4 *c = reinterpret_cast<const float4 *>(E + (size_t)(e * N + j) * SE);
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < EMB / 4; ++kk) {
                const float4 x = q[kk], y = c[kk];
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
    if (a.attn) {
        float *dst = a.attn + (size_t)s0 * NN;
        if (big) {                                      // 16 lanes per row: no per-element division
            for (int r = tid >> 4; r < rows; r += NG)
                for (int j = tid & 15; j < N; j += 16) dst[(size_t)r * N + j] = M[(size_t)r * NP + j];
        } else {
            for (int k = tid; k < envs * NN; k += TPBW) { const int r = k / N, j = k - r * N; dst[k] = M[(size_t)r * NP + j]; }
        }
    }
    if (a.stop == 5) return;
    // ---- L GCN hops: HW on the matrix cores, masked aggregation on the VALU ----
    float *Amat = bufA;                                 // [rows][NP]
    for (int l = 0; l < L; ++l) {
        const float *Hin = (l == 0) ? E : H;
        // big teams: the hop's mask product Range*Chan_l is fetched (coalesced) BEFORE the H.Wg MFMAs so the HBM
        // latency hides under them; registers hold it until the A tile is written
        // 16-lane group gq = tid>>4 owns matrix rows gq, gq+16, ...; lane sl = tid&15 owns columns sl, sl+16, ...
        // mk[rb*JB + jb] is element (row rb*16+gq, column jb*16+sl): no divisions anywhere in the mask path
        constexpr int JB = (MAXMK == 25 || MAXMK == 15) ? 5 : 8;            // column blocks per row (N <= 16*JB)
        float mk[MAXMK > 0 ? MAXMK : 1];
        const bool masked = a.adj || a.chan;
        if (MAXMK > 0 && masked) {
            const int gq = tid >> 4, sl = tid & 15;
#pragma unroll
            for (int q = 0; q < MAXMK; ++q) {
                const int r = (q / JB) * NG + gq, j = (q % JB) * 16 + sl;
                float v = 1.0f;
                if (r < rows && j < N) {
                    const int e = envs == 1 ? 0 : r / N, i = r - e * N;
                    const size_t off = (size_t)i * N + j;
                    if (a.adj) v = a.adj[(size_t)(s0 + e) * NN + off];
                    if (a.chan) v *= a.chan[((size_t)(s0 + e) * L + l) * NN + off];
                }
                mk[q] = v;
            }
        }
        l_sq.template run<false>(Hin, SE, T, SE, RT, wave, lane);                              // H.Wg_l
        if (a.stop == 61 + l) return;
        if (l + 1 < L) l_sq.load(tw.gcn_p + (size_t)(l + 1) * EMB * EMB, nullptr, wave, lane);
        if (MAXMK > 0 && big) {
            // masked + renormalised rows of A, one wave per row (coalesced mask reads along j), written into the
            // zero-padded [rows][NPA] tile the aggregation MFMA reads with 16-byte loads
            const int NPA = (((N + 15) >> 4) << 4) + 4;
            {   // A row = M row * mask, renormalised: 16 lanes per row, DPP row sum, one pass
                const int gq = tid >> 4, sl = tid & 15;
#pragma unroll
                for (int rb = 0; rb < MAXMK / JB; ++rb) {
                    const int r = rb * NG + gq;
                    if (rb * NG < rows) {               // uniform: every group of the block shares rb
                        const bool live = r < rows;
                        const float *mr = M + (size_t)(live ? r : 0) * NP;
                        float v[JB];
                        float sum = 0.0f;
#pragma unroll
                        for (int jb = 0; jb < JB; ++jb) {
                            const int j = jb * 16 + sl;
                            v[jb] = (live && j < N) ? mr[j] * (masked ? mk[rb * JB + jb] : 1.0f) : 0.0f;
                            sum += v[jb];
                        }
                        const float rden = __builtin_amdgcn_rcpf(row16_sum(sum) + 1e-12f);
                        if (live) {
                            float *ar = Amat + (size_t)r * NPA;
#pragma unroll
                            for (int jb = 0; jb < JB; ++jb) { const int j = jb * 16 + sl; if (j < NPA) ar[j] = j < N ? v[jb] * rden : 0.0f; }
                        }
                    }
                }
            }
            lds_barrier();
            if (a.stop == 51 + l) return;
            agg_mfma<32>(Amat, NPA, T, tw.gcn_b ? tw.gcn_b + (size_t)l * EMB : nullptr,
                         (l == L - 1 && !a.no_residual) ? E : nullptr, H, N, envs, wave, lane, NW);
            lds_barrier();
            continue;
        }
        if (N <= 16) {
            // small teams: one thread builds its whole masked + renormalised row (no intermediate barrier)
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
            for (int k = tid; k < envs * NN; k += TPBW) {    // A = M * Range * Chan_l (coalesced mask reads)
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
        {
            const int o = tid & (EMB - 1), rg = tid >> 6;
            const float bv = tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + o] : 0.0f;
            for (int r0 = rg * 4; r0 < rows; r0 += 16) {
                const int e = r0 / N;
                const float *hw = T + (size_t)e * N * SE + o;
                float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
                const float *ar[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ar[i] = Amat + (size_t)min(r0 + i, rows - 1) * NP;
                for (int j = 0; j < N; ++j) {
                    const float h = hw[(size_t)j * SE];
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(ar[i][j], h, acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (r0 + i < rows && (r0 + i) / N == e) {
                        const float hv = fast_tanh(acc[i] + bv);      // last hop: write E + H_L (residual, policy :74-77)
                        H[(size_t)(r0 + i) * SE + o] = (l == L - 1 && !a.no_residual) ? E[(size_t)(r0 + i) * SE + o] + hv : hv;
                    }
            }
        }
        lds_barrier();
    }
    }   // !quad_path
    if (a.stop == 6) return;
    CM_PROBE(10);
    // ---- residual ----
    if (L == 0) {      // no hops: embeddings[-1] is E itself, so x = E + E with the residual (:74-77), else E
        const float f = a.no_residual ? 1.0f : 2.0f;    // (with hops the last hop's epilogue adds the residual)
        for (int k = tid; k < rows * EMB; k += TPBW) { const int r = k >> 6, o = k & 63; H[(size_t)r * SE + o] = f * E[(size_t)r * SE + o]; }
        lds_barrier();
    }

    if (!EARLY) l_x1.load(HEAD == 0 ? ph.h1_p : chd.d1_p, HEAD == 0 ? ph.b1 : chd.b1, wave, lane);
    if (HEAD == 0) {
        if (!EARLY) l_h2.load(ph.h2_p, ph.b2, wave, lane);
        Layer<H2, H3, NW> l_h3;
        l_h3.load(ph.h3_p, ph.b3, wave, lane);
        l_x1.template run<true>(H, SE, bufA, SA, RT, wave, lane);
        lds_barrier();
        CM_PROBE(11);
        l_h2.template run<true>(bufA, SA, T, SE, RT, wave, lane);
        const int A = ph.n_act;
        Layer<H3, 16, NW> l_h4;                              // 32 -> n_act (<= 8) logits, zero-padded to one column tile
        l_h4.load(ph.h4_p, ph.b4, wave, lane, A);
        lds_barrier();
        CM_PROBE(12);
        l_h3.template run<true>(T, SE, E, SE, RT, wave, lane);
        lds_barrier();
        CM_PROBE(13);
        if (a.stop == 7) return;
        l_h4.template run<false>(E, SE, bufA, SA, RT, wave, lane);
        lds_barrier();
        CM_PROBE(14);
        for (int r = tid; r < rows; r += TPBW) {
            float lg[MAX_ACT], p[MAX_ACT];
            const float *x = bufA + (size_t)r * SA;
#pragma unroll
            for (int c = 0; c < MAX_ACT; ++c) lg[c] = (c < A) ? x[c] : 0.0f;
            float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
#pragma unroll
            for (int c = 0; c < MAX_ACT; ++c) if (c < A) mx = fmaxf(mx, lg[c]);
#pragma unroll
            for (int c = 0; c < MAX_ACT; ++c) if (c < A) { p[c] = __builtin_amdgcn_exp2f((lg[c] - mx) * 1.4426950408889634f); sum += p[c]; }
            const size_t grow = (size_t)s0 * N + r;
            const float rsum = __builtin_amdgcn_rcpf(sum);              // hardware exp2 / rcp: 1 ulp, far inside 1e-5
#pragma unroll
            for (int c = 0; c < MAX_ACT; ++c) if (c < A) {
                const float av = a.avail ? a.avail[grow * A + c] : 1.0f;
                p[c] = (p[c] * rsum) * av; msum += p[c];
            }
            const float rmsum = __builtin_amdgcn_rcpf(msum);
#pragma unroll
            for (int c = 0; c < MAX_ACT; ++c) if (c < A) p[c] = p[c] * rmsum;
            if (a.probs) {
#pragma unroll
                for (int c = 0; c < MAX_ACT; ++c) if (c < A) a.probs[grow * A + c] = p[c];
            }
            if (a.actions || act_lds) {
                int act = 0;
                if (a.greedy) {
                    float best = p[0];
#pragma unroll
                    for (int c = 1; c < MAX_ACT; ++c) if (c < A && p[c] > best) { best = p[c]; act = c; }
                } else {
                    const int e = r / N, i = r - e * N;
                    const u32x4 xr = philox4x32_10((uint32_t)(a.env_id_offset + s0 + e),
                                                   draw_step, SITE_ACTION,
                                                   (uint32_t)i, a.key0, a.key1);
                    const float u = unit_f32(xr.x);
                    float acc = 0.0f;
                    int sel = -1, last = 0;
#pragma unroll
                    for (int c = 0; c < MAX_ACT; ++c) if (c < A) { if (p[c] > 0.0f) last = c; acc += p[c]; if (sel < 0 && u < acc) sel = c; }
                    act = sel < 0 ? last : sel;
                }
                if (a.actions) a.actions[grow] = act;
                if (act_lds) act_lds[r] = act;
            }
        }
        CM_PROBE(15);
    } else {
        l_x1.template run<true>(H, SE, T, SE, RT, wave, lane);
        lds_barrier();
        for (int r = tid; r < rows; r += TPBW) {
            const float *x = T + (size_t)r * SE;
            float acc = chd.b2 ? chd.b2[0] : 0.0f;
            for (int k = 0; k < DH; ++k) acc = fmaf(x[k], chd.w2t[k], acc);
            rs[r] = acc;
        }
        lds_barrier();
        for (int e = tid; e < envs; e += TPBW) {
            float v = 0.0f;
            for (int i = 0; i < N; ++i) v += rs[e * N + i];
            a.values[s0 + e] = v;
        }
    }
}



// ---- host-side helpers shared by the stand-alone and the fused launchers ----
inline int pick_epb(int N) {
    if (N % 4 != 0) return 1;                 // aggregation tiles must not straddle envs
    static const int forced = [] { const char *e = getenv("COMMARL_FWD_ROWS"); return e ? atoi(e) : 0; }();
    // ~32 rows per workgroup (2 row tiles, >= 2 workgroups per CU); mid-size teams take two envs so that the 16-row
    // tiles are full and every weight fragment serves three of them (N = 24: 48 rows, measured 145 -> 117 us;
    // three envs = 72 rows drop to one workgroup per CU and lose: 184 us)
    const int target = forced ? forced : ((N > 16 && N < 32) ? 2 * N : 32);
    const int e = target / N;
    return e > 0 ? e : 1;
}
inline int kpad_of(int d) { const int k = (d + 15) & ~15; return (k == 32 || k == 64 || k == 80) ? k : (k == 16 ? 32 : (k == 48 ? 64 : 0)); }
struct PackLayout { size_t enc1, enc2, attn, gcn, x1, h2, h3, h4, total; };
inline PackLayout pack_layout(int kpad, int L, bool policy) {
    PackLayout o{};
    size_t off = 0;
    o.enc1 = off; off += (size_t)kpad * EH;
    o.enc2 = off; off += (size_t)EH * EMB;
    o.attn = off; off += (size_t)EMB * EMB;
    o.gcn = off; off += (size_t)L * EMB * EMB;
    o.x1 = off; off += policy ? (size_t)EMB * H1 : (size_t)EMB * DH;
    if (policy) { o.h2 = off; off += (size_t)H1 * H2; o.h3 = off; off += (size_t)H2 * H3; o.h4 = off; off += (size_t)H3 * 16; }
    o.total = off;
    return o;
}

}  // namespace mf

bool policy_shape_ok(const cm_policy_weights *w);   // cm_policy_mfma.hip

}  // namespace cm
