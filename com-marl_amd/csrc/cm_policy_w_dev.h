// cm_policy_w_dev.h - Comm-DP policy forward + sample for teams of 4 with WAVE-OWNED row tiles: one wave carries its 16 agent
// rows (4 whole envs) through the entire network in REGISTERS - no activation ever touches LDS, no workgroup barrier
// separates two layers.  (reference: comm_categorical_mlp_policy.py:48-119, comm_base_net.py:80-108,
// attention_module.py:26-51, graph_conv_module.py:51-72, categorical_mlp_module.py:64-80)
//
// Why: the 32-row workgroup of cm_policy_h_dev.h is a chain of ~12 barrier-separated stages, each
// an LDS round trip of the activations (b128 reads -> MFMA -> epilogue -> b64 writes -> s_barrier) of ~1 us; at 4096 envs
// there is exactly ONE 16-row tile per SIMD, so the step time is that chain and nothing else.  Here the chain is broken
// by a layout identity of the 16x16 MFMA family: in the transposed formulation  D[feature][row] = sum_k W[feature][k] X[row][k]
// a lane (c = lane & 15, g = lane >> 4) ends with features 16 ct + 4 g + r (r = 0..3) of row c - and the B operand of the
// NEXT layer wants, from that same lane, 8 k-values of row c.  Two column tiles (ct = 2q, 2q + 1) are exactly such an
// 8-pack, provided the next layer's weights are packed with the matching k order
//        k-slot (q, g, e)  <->  feature 32 q + 16 (e >> 2) + 4 g + (e & 3)
// (cm_policy_pack does that, pack_layer_w_kernel).  So a layer's output registers ARE the next layer's operand.
// The N x N part keeps to registers as well:
//   * scores as  S^T = E . Q^T  (A = E operand, B = Q operand, both in the shared k order): lane (c, g) gets
//     score[i = c][j = 4 g + r] - for g == c >> 2 (the env of row c) that is the WHOLE softmax row of agent c in one lane:
//     max / exp / sum / mask / renormalise without a single cross-lane operation;
//   * H.Wg in the NON-transposed form (operands swapped): lane (c, g) holds HW[source row 4 g + r][feature 16 ct + c] -
//     the A operand of v_mfma_f32_16x16x16_f16 (k = 4 per lane group) for  out^T[feature][i] = sum_j HW[j][feature] A[i][j],
//     whose B operand is the lane's own coefficient row (zero off the block diagonal) and whose result is again
//     "features 16 ct + 4 g + r of row c": bias, tanh, residual and the head follow in place.
// Weights: all four waves of a workgroup (16 envs, ONE workgroup per CU = one wave per SIMD at 4096 envs) read their A
// operands from LDS, where the operand pack is staged once per launch (128 KB; the last three head layers, 44 KB, overlay
// the encoder's slots once every wave has left the encoder).  Arithmetic: the f16-split scheme of cm_policy_h_dev.h
// (x = hi + 2^-12 lo, three MFMAs per block), f32-grade; pinned by the same reference fixtures at 1e-5.
#pragma once
#include "cm_policy_h_dev.h"

namespace cm {
namespace mw {

using mf::FwdArgs;
using mf::v4f;
using mf::fast_tanh;
using mf::EH; using mf::EMB; using mf::H1; using mf::H2; using mf::H3; using mf::MAX_ACT;
using mh::v8h; using mh::v4h; using mh::h16;

constexpr int WG_ENVS = 16, WG_ROWS = 64, KH = 32;       // teams of 4: 16 envs = 64 rows = 4 wave tiles per workgroup
constexpr int FRAG = 64;                                 // uint4 per (column tile, k block, plane): one 16-byte chunk per lane

// operand pack of the wave-owned kernel, in uint4 (16-byte) units.  Layer (K -> OUT): (OUT / 16) x (K / 32) x 2 planes x 64.
struct PackW { int enc1, enc2, attn, gcn, x1, h3, h4, bias, lds_u4, h2, total; };
__host__ __device__ constexpr int frag_u4(int K, int OUT) { return (OUT / 16) * (K / 32) * 2 * FRAG; }
constexpr int BIAS_U4 = 144;                             // 576 floats: the bias block (BiasMap), written by pack_bias_w_kernel
// [0, lds_u4) is the workgroup's LDS image, copied once per launch; the 128 -> 64 head layer (h2) stays in REGISTERS for the life
// of a wave (its 32 KB do not fit beside the rest in the 160 KB of LDS)
__host__ __device__ constexpr PackW pack_w(int L) {
    PackW o{};
    int off = 0;
    o.enc1 = off; off += frag_u4(KH, EH);
    o.enc2 = off; off += frag_u4(EH, EMB);
    o.attn = off; off += frag_u4(EMB, EMB);
    o.gcn = off; off += L * frag_u4(EMB, EMB);
    o.x1 = off; off += frag_u4(EMB, H1);
    o.h3 = off; off += frag_u4(H2, H3);
    o.h4 = off; off += frag_u4(H3, 32);                  // logits as two column tiles: actions 0..3 -> rows 0..3, action 4 -> row 16
    o.bias = off; off += BIAS_U4;
    o.lds_u4 = off;
    o.h2 = off; off += frag_u4(H1, H2);
    o.total = off;
    return o;
}

// biases in LDS (floats): enc_b1[128] enc_b2[64] gcn_b[2][64] b1[128] b2[64] b3[32] b4 as [32] (feature 16 ct + 4 g + r order)
struct BiasMap { int e1, e2, g, b1, b2, b3, b4, total; };
__host__ __device__ constexpr BiasMap bias_map(int L) {
    BiasMap o{};
    int off = 0;
    o.e1 = off; off += EH; o.e2 = off; off += EMB; o.g = off; off += (L > 0 ? L : 1) * EMB;
    o.b1 = off; off += H1; o.b2 = off; off += H2; o.b3 = off; off += H3; o.b4 = off; off += 32;
    o.total = off;
    return o;
}
// LDS bytes of the policy part: the image (fragments | biases) | sampled actions
__host__ __device__ inline size_t lds_policy_bytes(int L) { return (size_t)pack_w(L).lds_u4 * 16 + WG_ROWS * 4; }

struct WeightsW {
    const uint4 *pack;                                   // cm_policy_pack's wave-owned section (fragments, then the bias block)
    int n_act;
};

// k order of every layer but the first (see the header)
__host__ __device__ inline int kmap(int q, int g, int e) { return 32 * q + 16 * (e >> 2) + 4 * g + (e & 3); }

// An activation of width 32 KB as MFMA operand registers: hi / lo planes, 8 halves per k block
template <int KB> struct Act { v8h hi[KB], lo[KB]; };

#define CM_MFW(A, B, ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0)

// One layer's A fragments in registers.  fetch() is issued one layer AHEAD of run(): with one wave per SIMD nothing else
// hides the LDS latency (a read issued next to its MFMA cost ~60 exposed cycles per (tile, k block): 86 of them per step).
template <int KB, int CT>
struct Frags {
    v8h h[CT][KB], l[CT][KB];
    __device__ __forceinline__ void fetch(const uint4 *W, int lane) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < KB; ++q) {
                h[ct][q] = __builtin_bit_cast(v8h, W[((ct * KB + q) * 2 + 0) * FRAG + lane]);
                l[ct][q] = __builtin_bit_cast(v8h, W[((ct * KB + q) * 2 + 1) * FRAG + lane]);
            }
        // left to itself the scheduler sinks every read to just in front of its MFMA (one exposed LDS round trip per tile and
        // k block); the barrier keeps the whole batch of reads where it was written: ahead of the layer that runs meanwhile
        __builtin_amdgcn_sched_barrier(0);
    }
};

// ---- epilogue arithmetic, written STAGE-WISE over small arrays: with one wave per SIMD a dependent instruction costs ~1.7x
// an independent one (and a transcendental twice a plain one), so every stage below is N independent instructions ----------
constexpr float TANH_PRESCALE = 2.8853900817779268f;     // 2 log2(e): folded into the weights and biases of every tanh layer by the pack

// v[i] holds 2 log2(e) x (pack-time prescale): tanh(x) = 1 - 2 / (2^v + 1)
template <int N>
__device__ __forceinline__ void tanh_stage(float (&v)[N]) {
    float e[N];
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_exp2f(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] += 1.0f;
#pragma unroll
    for (int i = 0; i < N; ++i) e[i] = __builtin_amdgcn_rcpf(e[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = fmaf(-2.0f, e[i], 1.0f);
}
// y -> (hi, lo) f16 planes:  hi = f16(y), lo = f16(y - hi), UNSCALED: v_mfma_f32_*_f16 honours f16 subnormals on gfx950
// (tools/micro/mfma_f16_subnormal.hip: 2^-24 comes through exactly), so the residual needs no 2^12 lift to survive - its absolute
// error is <= 2^-25 (subnormal spacing) or 2^-11 of itself, i.e. <= max(3e-8, 2^-22 |y|).  Every product term then carries the same
// scale and ONE accumulator takes hi.hi + hi.lo + lo.hi: no join multiply-add per value, half the accumulator registers.
// Pairs go through v_cvt_pk_f16_f32 (two values per instruction).
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
__host__ __device__ inline void split_u(float x, h16 &h, h16 &l) { h = (h16)x; l = (h16)(x - (float)h); }
template <int N>
__device__ __forceinline__ void split_stage(const float (&y)[N], h16 (&h)[N], h16 (&l)[N]) {
    static_assert(N % 2 == 0, "pairs");
    float d[N];
#pragma unroll
    for (int i = 0; i < N; i += 2) { const v2h p = __builtin_convertvector((v2f){ y[i], y[i + 1] }, v2h); h[i] = p[0]; h[i + 1] = p[1]; }
#pragma unroll
    for (int i = 0; i < N; ++i) d[i] = y[i] - (float)h[i];
#pragma unroll
    for (int i = 0; i < N; i += 2) { const v2h p = __builtin_convertvector((v2f){ d[i], d[i + 1] }, v2h); l[i] = p[0]; l[i + 1] = p[1]; }
}

// One dense layer on this wave's 16 rows, transposed form: a lane ends with features 16 ct + 4 g + r of row c.  Tiles are
// taken in PAIRS (2p, 2p + 1) = k block p of the next layer's operand; the epilogue of pair p - 1 (join, tanh, split) is written
// behind the MFMAs of pair p, so the vector ALU works in the matrix pipe's shadow.  keep (may be null): the f32 results
// [CT] (the embedding E, which the residual needs again).
// sv (may be null): this lane's row of a [rows][16 CT] f32 matrix that receives the layer's output (training forward: what the
// backward pass reads back) - features 16 ct + 4 g .. + 3 of row c are one 16-byte store.
template <int KB, int CT, bool TANH, bool BIAS>
__device__ __forceinline__ void dense_act(const Frags<KB, CT> &f, const float *bias, const Act<KB> &x, Act<CT / 2> &y, v4f *keep, int lane,
                                          float *sv = nullptr) {
    const int g = lane >> 4;
    const v4f zero = (v4f){ 0.f, 0.f, 0.f, 0.f };
    v4f acc[CT];
#pragma unroll
    for (int p = 0; p <= CT / 2; ++p) {
        if (p < CT / 2) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ct = 2 * p + t;
                if (BIAS) { const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * ct + 4 * g); acc[ct] = (v4f){ b.x, b.y, b.z, b.w }; }
#pragma unroll
                for (int q = 0; q < KB; ++q) {
                    const v4f a0 = (q == 0 && !BIAS) ? zero : acc[ct];
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[ct][q], x.lo[q], a0, 0, 0, 0);      // small terms first
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.l[ct][q], x.hi[q], acc[ct], 0, 0, 0);
                    acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[ct][q], x.hi[q], acc[ct], 0, 0, 0);
                }
            }
        }
        if (p > 0) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * (p - 1) + (e >> 2)][e & 3];
            if (TANH) tanh_stage<8>(v);
            if (keep) {
#pragma unroll
                for (int e = 0; e < 8; ++e) keep[2 * (p - 1) + (e >> 2)][e & 3] = v[e];
            }
            if (sv) {
                *reinterpret_cast<float4 *>(sv + 16 * (2 * (p - 1)) + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4 *>(sv + 16 * (2 * (p - 1) + 1) + 4 * g) = make_float4(v[4], v[5], v[6], v[7]);
            }
            h16 h[8], l[8];
            split_stage<8>(v, h, l);
#pragma unroll
            for (int e = 0; e < 8; ++e) { y.hi[p - 1][e] = h[e]; y.lo[p - 1][e] = l[e]; }
        }
    }
}

// The same product with the operands swapped (non-transposed form): out[ct][r] = row 4 g + r, feature 16 ct + c, joined f32 - H.Wg
// for the aggregation; and the plain transposed form returning f32 (the logits).
template <int KB, int CT, bool SWAP, bool BIAS>
__device__ __forceinline__ void dense_f32(const Frags<KB, CT> &f, const float *bias, const Act<KB> &x, v4f (&out)[CT], int lane) {
    const int g = lane >> 4;
    const v4f zero = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        v4f acc = zero;
        if (BIAS) { const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * ct + 4 * g); acc = (v4f){ b.x, b.y, b.z, b.w }; }
#pragma unroll
        for (int q = 0; q < KB; ++q) {
            if (!SWAP) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[ct][q], x.lo[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.l[ct][q], x.hi[q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(f.h[ct][q], x.hi[q], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.lo[q], f.h[ct][q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.hi[q], f.l[ct][q], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x.hi[q], f.h[ct][q], acc, 0, 0, 0);
            }
        }
        out[ct] = acc;
    }
}

#define CM_WPROBE(i) do { if (a.probe && tid == 0) a.probe[(size_t)blk * mf::NPROBE + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

// ---- staging: the LDS image [fragments | biases], all 256 threads, once per launch (no barrier inside) ---------------------
template <int LHOPS>
__device__ __forceinline__ void stage_w(const WeightsW &w, unsigned char *lds, int tid) {
    constexpr PackW pk = pack_w(LHOPS);
    uint4 *WL = reinterpret_cast<uint4 *>(lds);
    // two rounds of ~18 chunks per thread, every load of a round in flight before its first store (at this point of the kernel
    // the registers are free): two L2 round trips instead of five
    constexpr int PER = (pk.lds_u4 + 255) / 256, BATCH = (PER + 1) / 2;
    // every workgroup of the launch reads the SAME 145 KB at the same time: each starts at another 4 KB chunk, so that at any
    // moment the CUs spread over the L2 channels instead of queueing on the same lines
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)(2 * BATCH));
    auto chunk_of = [&](int j) { const int c = j + rot; return c >= 2 * BATCH ? c - 2 * BATCH : c; };
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        uint4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const int i = chunk_of(round * BATCH + k) * 256 + tid; v[k] = w.pack[i < pk.lds_u4 ? i : 0]; }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) { const int i = chunk_of(round * BATCH + k) * 256 + tid; if (i < pk.lds_u4) WL[i] = v[k]; }
    }
}

// the register-resident layer (128 -> 64 of the head): fetched once per wave, from the pack in global memory
struct ResidentW {
    Frags<4, 4> h2;
    template <int LHOPS>
    __device__ __forceinline__ void fetch(const WeightsW &w, int lane) { h2.fetch(w.pack + pack_w(LHOPS).h2, lane); }
};

// ---- one wave's tile: rows [16 * wave, 16 * wave + 16) of workgroup blk.  No workgroup barrier: everything is the wave's own. ----
// lds: the staged image.  act_lds (may be null): sampled actions of the workgroup's 64 rows for the env phase of the fused step.
// A ragged last workgroup (S % 16 != 0) computes on zero rows and stores nothing for them.
// OBS_LDS: the observation rows come from an LDS copy the caller keeps (byte offset `obs_row` of this lane's row c, rows of
// at least 24 floats with zeros behind the d real entries) instead of from a.obs - the carried rollout (cm_rollout_w.hip).
// TRAIN: the training forward (cm_policy_forward_saved_wave) - every activation the backward pass needs goes to the a.sv_* matrices
// (layouts of cm_fwd_saves, include/commarl.h) straight from the epilogue registers; nothing is sampled.
// HEADK 1: the critic's head behind the same trunk (training forward only; cm_critic_forward_saved_wave) - decoder layer 64 -> 64
// (fragments in the policy's x1 slot of the image), value layer 64 -> 1 as a 16-wide tile (h3 slot), per-agent values to a.sv_out,
// their sum over the team to a.values.  `res` is not touched then.
template <int LHOPS, bool OBS_LDS = false, bool TRAIN = false, int HEADK = 0>
__device__ __forceinline__ void policy_tile_w(const FwdArgs &a, int n_act, const ResidentW &res, const unsigned char *lds, int blk,
                                              int32_t *act_lds, int obs_row = 0) {
    static_assert(HEADK == 0 || TRAIN, "the critic head exists as training forward only");
    static_assert(LHOPS >= 1 && LHOPS <= 2, "wave-owned forward: one or two hops");
    const int tid = thread_x(), wave = tid >> 6, lane = tid & 63, c = lane & 15, g = lane >> 4;
    constexpr PackW pk = pack_w(LHOPS);
    constexpr BiasMap bm = bias_map(LHOPS);
    const uint4 *WL = reinterpret_cast<const uint4 *>(lds);
    const float *BL = reinterpret_cast<const float *>(lds + (size_t)pk.bias * 16);
    const int s0 = blk * WG_ENVS;                              // first env of the workgroup
    const int row = wave * 16 + c;                             // this lane's row inside the workgroup
    const size_t grow = (size_t)s0 * 4 + row;                  // global agent row
    const int env_l = wave * 4 + (c >> 2), agent = c & 3;      // env inside the workgroup / agent of row c
    const bool diag = g == (c >> 2);                           // lane holds its own env's source rows j = 4 g + r
    const bool rv = s0 + env_l < a.S;                          // row belongs to an env of this launch

    CM_WPROBE(2);
    // ---- observation rows straight into operand form: lane (c, g) takes features 8 g .. 8 g + 7 of row c ----
    Act<1> xo;
    {
        float ov[8];
        if constexpr (OBS_LDS) {
            const float4 *src = reinterpret_cast<const float4 *>(lds + obs_row + 32 * (g < 3 ? g : 0));
            const float4 v0 = src[0], v1 = src[1];
            const float vv[8] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w };
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = (rv && g < 3) ? vv[e] : 0.0f;
        } else {
        const float *src = a.obs + (rv ? grow : 0) * a.d;       // every load is in range: no predicated (branchy) loads
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = 8 * g + e; const float v = src[k < a.d ? k : 0]; ov[e] = (rv && k < a.d) ? v : 0.0f; }
        }
        h16 h[8], l[8];
        split_stage<8>(ov, h, l);
#pragma unroll
        for (int e = 0; e < 8; ++e) { xo.hi[0][e] = h[e]; xo.lo[0][e] = l[e]; }
    }
    const uint32_t draw_step = a.policy_step + (a.step_base ? *a.step_base : 0u);

    // ---- encoder (every layer's fragments are fetched while the layer before it runs) ----
    Frags<1, 8> f_e1; f_e1.fetch(WL + pk.enc1, lane);
    Frags<4, 4> f_e2; f_e2.fetch(WL + pk.enc2, lane);
    auto sv_row = [&](float *base, int width) -> float * { return (TRAIN && base && rv) ? base + grow * (size_t)width : nullptr; };
    Act<4> a1;
    dense_act<1, 8, true, true>(f_e1, BL + bm.e1, xo, a1, nullptr, lane, sv_row(a.sv_a1, EH));
    Frags<2, 4> f_at; f_at.fetch(WL + pk.attn, lane);
    v4f E[4];
    Act<2> xe;
    dense_act<4, 4, true, true>(f_e2, BL + bm.e2, a1, xe, E, lane, sv_row(a.sv_e, EMB));
    Frags<2, 4> f_g; f_g.fetch(WL + pk.gcn, lane);
    CM_WPROBE(3);

    // ---- attention: Q = E.Wa^T, scores^T = E.Q^T, softmax row in the diagonal lanes ----
    float m[4];
    {
        Act<2> xq;
        dense_act<2, 4, false, false>(f_at, nullptr, xe, xq, nullptr, lane, sv_row(a.sv_q, EMB));
        v4f hh = (v4f){ 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int q = 0; q < 2; ++q) { CM_MFW(xe.hi[q], xq.lo[q], hh); CM_MFW(xe.lo[q], xq.hi[q], hh); CM_MFW(xe.hi[q], xq.hi[q], hh); }
        float sc[4], mx = -INFINITY, sum = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { sc[r] = hh[r]; mx = fmaxf(mx, sc[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) { m[r] = __builtin_amdgcn_exp2f((sc[r] - mx) * 1.4426950408889634f); sum += m[r]; }
        const float rs = __builtin_amdgcn_rcpf(sum);
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] *= rs;
    }
    const size_t env_g = (size_t)s0 + env_l;
    if (a.attn && diag && rv) {                                 // row `agent` of the env's 4 x 4 matrix: one 16-byte store
        float *dst = a.attn + env_g * 16 + 4 * agent;
        __builtin_nontemporal_store(m[0], dst); __builtin_nontemporal_store(m[1], dst + 1);
        __builtin_nontemporal_store(m[2], dst + 2); __builtin_nontemporal_store(m[3], dst + 3);
    }

    CM_WPROBE(5);
    // ---- hops: H_{l+1} = tanh(A_l.(H_l.Wg_l) + b_l), A_l = M * Range * Chan_l renormalised (comm_base_net.py:99-105) ----
    Act<2> xh = xe;
    Frags<2, HEADK == 0 ? 8 : 4> f_x1;                          // first head layer: policy 64 -> 128, critic 64 -> 64
#pragma unroll
    for (int l = 0; l < LHOPS; ++l) {
        v4f hw[4];                                             // H.Wg_l: rows 4 g + r, feature 16 ct + c
        dense_f32<2, 4, true, false>(f_g, nullptr, xh, hw, lane);
        if constexpr (TRAIN) {
            // H.Wg_l in the non-transposed form: this lane holds rows 4 g + r of the tile (env g of the wave), feature 16 ct + c.
            // The pack folded the tanh prescale into Wg_l (the aggregation's output goes through tanh): taken out again here.
            if (a.sv_hw[l] && s0 + wave * 4 + g < a.S) {
                float *dst = a.sv_hw[l] + ((size_t)s0 * 4 + wave * 16 + 4 * g) * EMB + c;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(size_t)r * EMB + 16 * ct] = hw[ct][r] * (1.0f / TANH_PRESCALE);
            }
        }
        if (l + 1 < LHOPS) f_g.fetch(WL + pk.gcn + (l + 1) * frag_u4(EMB, EMB), lane);
        else f_x1.fetch(WL + pk.x1, lane);
        float cf[4], den = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = m[r];
            if (a.adj && rv) x *= a.adj[env_g * 16 + 4 * agent + r];
            if (a.chan && rv) x *= a.chan[(env_g * LHOPS + l) * 16 + 4 * agent + r];
            cf[r] = x; den += x;
        }
        const float rden = __builtin_amdgcn_rcpf(den + 1e-12f);
        v4h bh, bl;
        {
            float cn[4];
            h16 h[4], lo_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) cn[r] = diag ? cf[r] * rden : 0.0f;
            split_stage<4>(cn, h, lo_);
#pragma unroll
            for (int r = 0; r < 4; ++r) { bh[r] = h[r]; bl[r] = lo_[r]; }
        }
        const bool last = l == LHOPS - 1;
        // A operands of all four feature tiles (H.Wg_l as f16 planes), then the twelve small MFMAs, then ONE epilogue
        v4h ah[4], al[4];
        {
            float hv[16];
            h16 h[16], lo_[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) hv[i] = hw[i >> 2][i & 3];
            split_stage<16>(hv, h, lo_);
#pragma unroll
            for (int i = 0; i < 16; ++i) { ah[i >> 2][i & 3] = h[i]; al[i >> 2][i & 3] = lo_[i]; }
        }
        v4f acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float4 b = *reinterpret_cast<const float4 *>(BL + bm.g + l * EMB + 16 * ct + 4 * g);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], bl, (v4f){ b.x, b.y, b.z, b.w }, 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(al[ct], bh, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_16x16x16f16(ah[ct], bh, acc[ct], 0, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[2 * p + (e >> 2)][e & 3];
            tanh_stage<8>(v);                                             // graph_conv_module.py:65-70 (pre-activation prescaled)
            if (last && !a.no_residual) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += E[2 * p + (e >> 2)][e & 3];   // policy :74-77
            }
            if constexpr (TRAIN) {
                if (float *sv = sv_row(a.sv_h[l], EMB)) {
                    *reinterpret_cast<float4 *>(sv + 16 * (2 * p) + 4 * g) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4 *>(sv + 16 * (2 * p + 1) + 4 * g) = make_float4(v[4], v[5], v[6], v[7]);
                }
            }
            h16 h[8], lo_[8];
            split_stage<8>(v, h, lo_);
#pragma unroll
            for (int e = 0; e < 8; ++e) { xh.hi[p][e] = h[e]; xh.lo[p][e] = lo_[e]; }
        }
    }

    CM_WPROBE(6);
    if constexpr (HEADK == 1) {
        // ---- critic head: 64 -> 64 (tanh) -> 1 (comm_base_critic.py:110-112), value of row c in the lanes g == 0 ----
        Frags<2, 1> f_d2; f_d2.fetch(WL + pk.h3, lane);
        Act<2> xd;
        dense_act<2, 4, true, true>(f_x1, BL + bm.b1, xh, xd, nullptr, lane, sv_row(a.sv_x1, EMB));
        v4f val[1];
        dense_f32<2, 1, false, true>(f_d2, BL + bm.b2, xd, val, lane);
        float v = (g == 0 && rv) ? val[0][0] : 0.0f;
        if (a.sv_out && g == 0 && rv) a.sv_out[grow] = v;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);                                   // the env's four agents are four neighbouring lanes
        if (a.values && g == 0 && rv && agent == 0) a.values[(size_t)s0 + env_l] = v;
        return;
    } else {
    // ---- head: 64 -> 128 -> 64 (resident fragments) -> 32 -> logits ----
    Act<4> x1;
    dense_act<2, 8, true, true>(f_x1, BL + bm.b1, xh, x1, nullptr, lane, sv_row(a.sv_x1, H1));
    Frags<2, 2> f_h3; f_h3.fetch(WL + pk.h3, lane);
    Frags<1, 2> f_h4; f_h4.fetch(WL + pk.h4, lane);
    // the sampler's uniforms do not depend on the logits
    float u = 0.0f;
    if constexpr (!TRAIN) {
        const u32x4 xr = philox4x32_10((uint32_t)(a.env_id_offset + (int)env_g), draw_step, SITE_ACTION, (uint32_t)agent, a.key0, a.key1);
        u = unit_f32(xr.x);
    }
    CM_WPROBE(7);
    Act<2> x2;
    dense_act<4, 4, true, true>(res.h2, BL + bm.b2, x1, x2, nullptr, lane, sv_row(a.sv_x2, H2));
    Act<1> x3;
    dense_act<2, 2, true, true>(f_h3, BL + bm.b3, x2, x3, nullptr, lane, sv_row(a.sv_x3, H3));
    v4f lg[2];
    dense_f32<1, 2, false, true>(f_h4, BL + bm.b4, x3, lg, lane);    // lanes g == 0: logits 0..3 in lg[0], logit 4 in lg[1][0]

    CM_WPROBE(9);
    // ---- softmax x avail, renormalise, sample / argmax (categorical_mlp_module.py:64-80): the 16 lanes with g == 0 ----
    const int A = n_act;
    float p[MAX_ACT];
#pragma unroll
    for (int cc = 0; cc < MAX_ACT; ++cc) p[cc] = cc < 4 ? lg[0][cc] : (cc == 4 ? lg[1][0] : 0.0f);
    if constexpr (TRAIN) {
        if (a.sv_out && g == 0 && rv) {                          // the logits, before softmax / avail mask
#pragma unroll
            for (int cc = 0; cc < 5; ++cc) if (cc < A) a.sv_out[grow * A + cc] = p[cc];
        }
    }
    float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) if (cc < A) mx = fmaxf(mx, p[cc]);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) if (cc < A) { p[cc] = __builtin_amdgcn_exp2f((p[cc] - mx) * 1.4426950408889634f); sum += p[cc]; }
    const float rsum = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) if (cc < A) {
        const float av = (a.avail && rv) ? a.avail[grow * A + cc] : 1.0f;
        p[cc] = (p[cc] * rsum) * av; msum += p[cc];
    }
    const float rmsum = __builtin_amdgcn_rcpf(msum);
#pragma unroll
    for (int cc = 0; cc < 5; ++cc) if (cc < A) p[cc] = p[cc] * rmsum;
    if (g == 0 && rv) {
        if (a.probs) {
#pragma unroll
            for (int cc = 0; cc < 5; ++cc) if (cc < A) __builtin_nontemporal_store(p[cc], a.probs + grow * A + cc);
        }
        if (a.actions || act_lds) {
            int act = 0;
            if (a.greedy) {
                float best = p[0];
#pragma unroll
                for (int cc = 1; cc < 5; ++cc) if (cc < A && p[cc] > best) { best = p[cc]; act = cc; }
            } else {
                float acc = 0.0f;
                int sel = -1, lastc = 0;
#pragma unroll
                for (int cc = 0; cc < 5; ++cc) if (cc < A) { if (p[cc] > 0.0f) lastc = cc; acc += p[cc]; if (sel < 0 && u < acc) sel = cc; }
                act = sel < 0 ? lastc : sel;
            }
            if (a.actions) a.actions[grow] = act;
            if (act_lds) act_lds[row] = act;
        }
    }
    }
    CM_WPROBE(10);
}
#undef CM_MFW

// stand-alone form: stage, one barrier, the resident layer, the tile
template <int LHOPS>
__device__ __forceinline__ void fwd_body_w(const FwdArgs &a, const WeightsW &w, unsigned char *lds, int blk, int32_t *act_lds) {
    const int tid = thread_x();
    CM_WPROBE(0);
    stage_w<LHOPS>(w, lds, tid);
    ResidentW res;
    res.fetch<LHOPS>(w, tid & 63);
    CM_WPROBE(1);
    __syncthreads();
    policy_tile_w<LHOPS>(a, w.n_act, res, lds, blk, act_lds);
}

// shapes the wave-owned forward takes: teams of 4, observation <= 32 wide, 1-2 hops, at most 5 actions
__host__ inline bool shape_ok_w(int N, int d, int L, int n_act) { return N == 4 && d <= KH && (L == 1 || L == 2) && n_act >= 1 && n_act <= 5; }

}  // namespace mw
}  // namespace cm
