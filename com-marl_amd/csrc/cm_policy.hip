// cm_policy.hip - fused Comm-DP policy / critic forward for rollouts (gfx950).
//
// One launch does, for S env states, everything CommCategoricalMLPPolicy.get_actions does
// through ~40 tiny torch ops (com_marl/torch/policies/comm_categorical_mlp_policy.py:48-119):
//   encoder MLP (garage/torch/modules/multi_headed_mlp_module.py:134-149, tanh out)
//   -> general attention softmax((E.Wa^T).E^T)            (modules/attention_module.py:26-51)
//   -> L x { A = M*Range*Chan_l ; A /= rowsum + 1e-12 ;     (modules/comm_base_net.py:99-103)
//            H' = tanh(A.(H.Wg_l) + bg_l) }                 (modules/graph_conv_module.py:51-72)
//   -> residual E + H_L -> head 64-128-64-32-5 -> softmax -> x avail -> renorm -> sample/argmax
// and CommBaseCritic.forward (baselines/comm_base_critic.py:91-114) with the value head.
//
// Layout: a 256-thread workgroup owns EPB whole envs (= EPB*N agent rows).  All activations of
// those rows stay in LDS from the observation load to the sampled action; HBM sees only the
// algorithmic traffic (obs + masks in, actions / probs / attention out).  Weights (<= 170 KB,
// stored transposed [in,out] so lanes read consecutive outputs) stream from L2.
// This file is the f32 VALU version (fmaf chains, k ascending); see DESIGN.md for the
// f32-MFMA plan.  f32 is required: the parity bar is 1e-5 on probabilities.
#include <stdlib.h>

#include "cm_internal.h"
#include "cm_rng.h"

namespace cm {

constexpr int TPB = 256;
constexpr int EH = 128, EMB = 64, H1 = 128, H2 = 64, H3 = 32, DH = 64;   // reference defaults (env_uitils.py:84,140,182-192)
constexpr int SA = EH + 4;     // LDS row strides (floats), +4 keeps float4 alignment and staggers banks
constexpr int SE = EMB + 4;
constexpr int RC = 4;          // rows per register tile
constexpr int MAX_ACT = 8;

struct TrunkW { const float *enc_w1t, *enc_b1, *enc_w2t, *enc_b2, *attn_wt, *gcn_w, *gcn_b; };
struct PolHead { const float *w1t, *b1, *w2t, *b2, *w3t, *b3, *w4t, *b4; int n_act; };
struct CritHead { const float *w1t, *b1, *w2t, *b2; };

struct FwdArgs {
    int S, N, d, L, EPB;
    const float *obs, *avail, *adj, *chan;
    uint32_t key0, key1, policy_step;
    const uint32_t *step_base;
    int env_id_offset, greedy, no_residual;
    int32_t *actions;
    float *probs, *attn, *values;
};

// y[r][o] = act(sum_k in[r][k] * Wt[k][o] + b[o]) for r < rows.  Threads: o = tid % OUT, row group = tid / OUT.
template <int OUT, bool TANH>
__device__ __forceinline__ void dense(const float *in, int in_stride, int K, const float *__restrict__ Wt,
                                      const float *__restrict__ bias, float *outp, int out_stride, int rows, int tid) {
    constexpr int GROUPS = TPB / OUT;
    const int o = tid % OUT, rg = tid / OUT;
    const float bv = bias ? bias[o] : 0.0f;
    for (int r0 = rg * RC; r0 < rows; r0 += GROUPS * RC) {
        float acc[RC];
        const float *row[RC];
#pragma unroll
        for (int i = 0; i < RC; ++i) { acc[i] = bv; row[i] = in + (size_t)min(r0 + i, rows - 1) * in_stride; }
        int k = 0;
        for (; k + 4 <= K; k += 4) {
            const float w0 = Wt[(size_t)(k + 0) * OUT + o], w1 = Wt[(size_t)(k + 1) * OUT + o],
                        w2 = Wt[(size_t)(k + 2) * OUT + o], w3 = Wt[(size_t)(k + 3) * OUT + o];
#pragma unroll
            for (int i = 0; i < RC; ++i) {
                const float4 x = *reinterpret_cast<const float4 *>(row[i] + k);
                acc[i] = fmaf(x.x, w0, acc[i]); acc[i] = fmaf(x.y, w1, acc[i]);
                acc[i] = fmaf(x.z, w2, acc[i]); acc[i] = fmaf(x.w, w3, acc[i]);
            }
        }
        for (; k < K; ++k) {
            const float w0 = Wt[(size_t)k * OUT + o];
#pragma unroll
            for (int i = 0; i < RC; ++i) acc[i] = fmaf(row[i][k], w0, acc[i]);
        }
#pragma unroll
        for (int i = 0; i < RC; ++i)
            if (r0 + i < rows) outp[(size_t)(r0 + i) * out_stride + o] = TANH ? tanhf(acc[i]) : acc[i];
    }
}

// floats of LDS needed by a block that owns `rows` rows of N-agent envs
__host__ __device__ inline size_t fwd_lds_floats(int rows, int epb, int N) {
    const int NP = N | 1;
    return (size_t)rows * (SA + 3 * SE) + (size_t)epb * N * NP + rows;
}

template <int HEAD>   // 0 = policy, 1 = critic
__global__ __launch_bounds__(TPB) void fwd_kernel(FwdArgs a, TrunkW tw, PolHead ph, CritHead chd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, N = a.N, d = a.d, L = a.L, NN = N * N, NP = N | 1;
    const int s0 = blockIdx.x * a.EPB;
    const int envs = min(a.EPB, a.S - s0);
    const int rows = envs * N, rows_max = a.EPB * N;
    float *bufA = lds;                                  // [rows][SA]  H1 / masked A / head hidden 1
    float *E = bufA + (size_t)rows_max * SA;            // [rows][SE]
    float *H = E + (size_t)rows_max * SE;               // [rows][SE]  (H,T contiguous: also the obs staging area)
    float *T = H + (size_t)rows_max * SE;               // [rows][SE]
    float *M = T + (size_t)rows_max * SE;               // [envs][N][NP] attention
    float *rs = M + (size_t)a.EPB * N * NP;             // [rows] row sums / per-row values
    float *X = H;                                       // [rows][SX]
    const int SX = 2 * SE;                              // d <= 132 checked on the host

    // ---- stage observations: coalesced HBM read of rows*d floats ----
    {
        const float *src = a.obs + (size_t)s0 * N * d;
        const int total = rows * d;
        for (int k = tid; k < total; k += TPB) { const int r = k / d, f = k - r * d; X[(size_t)r * SX + f] = src[k]; }
    }
    __syncthreads();
    // ---- encoder ----
    dense<EH, true>(X, SX, d, tw.enc_w1t, tw.enc_b1, bufA, SA, rows, tid);
    __syncthreads();
    dense<EMB, true>(bufA, SA, EH, tw.enc_w2t, tw.enc_b2, E, SE, rows, tid);
    __syncthreads();
    // ---- attention: Q = E.Wa^T ; scores = Q.E^T ; softmax over j ----
    dense<EMB, false>(E, SE, EMB, tw.attn_wt, nullptr, T, SE, rows, tid);
    __syncthreads();
    for (int k = tid; k < envs * NN; k += TPB) {
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
    __syncthreads();
    for (int r = tid; r < rows; r += TPB) {
        float *m = M + (size_t)r * NP;
        float mx = -INFINITY, sum = 0.0f;
        for (int j = 0; j < N; ++j) mx = fmaxf(mx, m[j]);
        for (int j = 0; j < N; ++j) { const float ex = expf(m[j] - mx); m[j] = ex; sum += ex; }
        for (int j = 0; j < N; ++j) m[j] = m[j] / sum;
    }
    __syncthreads();
    if (a.attn) {                                       // attention_weights output [S,N,N]
        float *dst = a.attn + (size_t)s0 * NN;
        for (int k = tid; k < envs * NN; k += TPB) { const int r = k / N, j = k - r * N; dst[k] = M[(size_t)r * NP + j]; }
    }
    // ---- L GCN hops ----
    float *Amat = bufA;                                 // [rows][NP] masked + renormalised attention
    for (int l = 0; l < L; ++l) {
        const float *Hin = (l == 0) ? E : H;
        dense<EMB, false>(Hin, SE, EMB, tw.gcn_w + (size_t)l * EMB * EMB, nullptr, T, SE, rows, tid);   // H.Wg ([in,out])
        // A = M * Range * Chan_l : coalesced mask reads
        for (int k = tid; k < envs * NN; k += TPB) {
            const int e = k / NN, ij = k - e * NN, r = k / N, j = k - r * N;
            float v = M[(size_t)r * NP + j];
            if (a.adj) v *= a.adj[(size_t)(s0 + e) * NN + ij];
            if (a.chan) v *= a.chan[((size_t)(s0 + e) * L + l) * NN + ij];
            Amat[(size_t)r * NP + j] = v;
        }
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            float sum = 0.0f;
            float *ar = Amat + (size_t)r * NP;
            for (int j = 0; j < N; ++j) sum += ar[j];
            const float den = sum + 1e-12f;
            for (int j = 0; j < N; ++j) ar[j] = ar[j] / den;
        }
        __syncthreads();
        // H' = tanh(A.(HW) + b): o = tid%64, 4 row groups, RC rows per tile (tiles never straddle envs: host picks EPB)
        {
            const int o = tid & (EMB - 1), rg = tid >> 6;
            const float bv = tw.gcn_b ? tw.gcn_b[(size_t)l * EMB + o] : 0.0f;
            for (int r0 = rg * RC; r0 < rows; r0 += (TPB / EMB) * RC) {
                const int e = r0 / N;
                const float *hw = T + (size_t)e * N * SE + o;
                float acc[RC] = { 0.0f, 0.0f, 0.0f, 0.0f };
                const float *ar[RC];
#pragma unroll
                for (int i = 0; i < RC; ++i) ar[i] = Amat + (size_t)min(r0 + i, rows - 1) * NP;
                for (int j = 0; j < N; ++j) {
                    const float h = hw[(size_t)j * SE];
#pragma unroll
                    for (int i = 0; i < RC; ++i) acc[i] = fmaf(ar[i][j], h, acc[i]);
                }
#pragma unroll
                for (int i = 0; i < RC; ++i)
                    if (r0 + i < rows && (r0 + i) / N == e) H[(size_t)(r0 + i) * SE + o] = tanhf(acc[i] + bv);
            }
        }
        __syncthreads();
    }
    // ---- residual (comm_categorical_mlp_policy.py:74-77) ----
    for (int k = tid; k < rows * EMB; k += TPB) { const int r = k >> 6, o = k & 63; H[(size_t)r * SE + o] = (L > 0 ? H[(size_t)r * SE + o] : E[(size_t)r * SE + o]) + (a.no_residual ? 0.0f : E[(size_t)r * SE + o]); }   // embeddings[-1] is E when there are no hops
    __syncthreads();

    if (HEAD == 0) {
        dense<H1, true>(H, SE, EMB, ph.w1t, ph.b1, bufA, SA, rows, tid);
        __syncthreads();
        dense<H2, true>(bufA, SA, H1, ph.w2t, ph.b2, T, SE, rows, tid);
        __syncthreads();
        dense<H3, true>(T, SE, H2, ph.w3t, ph.b3, E, SE, rows, tid);
        __syncthreads();
        const int A = ph.n_act;
        for (int r = tid; r < rows; r += TPB) {
            float lg[MAX_ACT], p[MAX_ACT];
            const float *x = E + (size_t)r * SE;
            for (int c = 0; c < A; ++c) {
                float acc = ph.b4 ? ph.b4[c] : 0.0f;
                for (int k = 0; k < H3; ++k) acc = fmaf(x[k], ph.w4t[(size_t)k * A + c], acc);
                lg[c] = acc;
            }
            float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
            for (int c = 0; c < A; ++c) mx = fmaxf(mx, lg[c]);
            for (int c = 0; c < A; ++c) { p[c] = expf(lg[c] - mx); sum += p[c]; }
            const size_t grow = (size_t)s0 * N + r;
            for (int c = 0; c < A; ++c) {                       // probs * avail, renormalise (:81-88)
                const float av = a.avail ? a.avail[grow * A + c] : 1.0f;
                p[c] = (p[c] / sum) * av; msum += p[c];
            }
            for (int c = 0; c < A; ++c) p[c] = p[c] / msum;
            if (a.probs) for (int c = 0; c < A; ++c) a.probs[grow * A + c] = p[c];
            if (a.actions) {
                int act;
                if (a.greedy) {                                 // np.argmax: first maximum (:112)
                    act = 0;
                    for (int c = 1; c < A; ++c) if (p[c] > p[act]) act = c;
                } else {                                        // inverse CDF on one Philox uniform per agent
                    const int e = r / N, i = r - e * N;
                    const u32x4 xr = philox4x32_10((uint32_t)(a.env_id_offset + s0 + e), a.policy_step + (a.step_base ? *a.step_base : 0u), SITE_ACTION, (uint32_t)i, a.key0, a.key1);
                    const float u = unit_f32(xr.x);
                    float acc = 0.0f;
                    int sel = -1, last = 0;
                    for (int c = 0; c < A; ++c) { if (p[c] > 0.0f) last = c; acc += p[c]; if (sel < 0 && u < acc) sel = c; }
                    act = sel < 0 ? last : sel;
                }
                a.actions[grow] = act;
            }
        }
    } else {
        dense<DH, true>(H, SE, EMB, chd.w1t, chd.b1, T, SE, rows, tid);
        __syncthreads();
        for (int r = tid; r < rows; r += TPB) {
            const float *x = T + (size_t)r * SE;
            float acc = chd.b2 ? chd.b2[0] : 0.0f;
            for (int k = 0; k < DH; ++k) acc = fmaf(x[k], chd.w2t[k], acc);
            rs[r] = acc;
        }
        __syncthreads();
        for (int e = tid; e < envs; e += TPB) {                 // sum over agents (comm_base_critic.py:112-114)
            float v = 0.0f;
            for (int i = 0; i < N; ++i) v += rs[e * N + i];
            a.values[s0 + e] = v;
        }
    }
}

static int pick_epb(int N) { return (N % RC == 0) ? (48 / N > 0 ? 48 / N : 1) : 1; }

template <int HEAD>
static int launch_fwd(FwdArgs a, const TrunkW &tw, const PolHead &ph, const CritHead &chd, void *stream) {
    a.EPB = pick_epb(a.N);
    const size_t lds = fwd_lds_floats(a.EPB * a.N, a.EPB, a.N) * sizeof(float);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "policy forward: n_agents too large for the 160 KB LDS tile");
    static unsigned long long attr_set[2] = { 0, 0 };
    if (cm::dev_first(attr_set[HEAD])) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_kernel<HEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int blocks = (a.S + a.EPB - 1) / a.EPB;
    hipLaunchKernelGGL(fwd_kernel<HEAD>, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, a, tw, ph, chd);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace cm

using namespace cm;

namespace cm {
int policy_forward_mfma(const cm_policy_weights *w, int32_t S, const float *obs, const float *avail, const float *adj,
                        const float *chan, uint64_t seed, int32_t env_id_offset, uint32_t policy_step,
                        const uint32_t *step_base, int32_t greedy, int32_t *actions, float *probs, float *attn,
                        void *stream);
int critic_forward_mfma(const cm_critic_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                        float *values, void *stream);
}

// COMMARL_POLICY_KERNEL=valu forces the generic VALU kernel (A/B timing, obs dims without an MFMA build)
static bool use_mfma() {
    static const bool v = [] { const char *e = getenv("COMMARL_POLICY_KERNEL"); return !(e && e[0] == 'v'); }();
    return v;
}

extern "C" int cm_policy_forward(const cm_policy_weights *w, int32_t n_samples, const float *obs, const float *avail,
                                 const float *dist_adj, const float *channels, uint64_t seed, int32_t env_id_offset,
                                 uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy,
                                 int32_t *actions, float *probs, float *attn, void *stream) {
    if (!w || !obs) return set_error(CM_ERR_ARG, "cm_policy_forward: null weights / obs");
    if (n_samples <= 0) return CM_OK;
    if (w->enc_hidden != EH || w->emb != EMB || w->h1 != H1 || w->h2 != H2 || w->h3 != H3)
        return set_error(CM_ERR_ARG, "cm_policy_forward: only the reference layer sizes (128 | 64 | 128,64,32) are built");
    if (w->n_act < 1 || w->n_act > MAX_ACT || w->d < 1 || w->d > 2 * SE - 4 || w->n_agents < 1 || w->n_hops < 0)
        return set_error(CM_ERR_ARG, "cm_policy_forward: bad dims");
    if (use_mfma()) {
        const int rc = policy_forward_mfma(w, n_samples, obs, avail, dist_adj, channels, seed, env_id_offset, policy_step,
                                           policy_step_base, greedy, actions, probs, attn, stream);
        if (rc <= 0) return rc;
    }
    FwdArgs a{};
    a.S = n_samples; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.avail = avail; a.adj = dist_adj; a.chan = channels;
    a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.policy_step = policy_step; a.step_base = policy_step_base;
    a.env_id_offset = env_id_offset; a.greedy = greedy; a.no_residual = w->no_residual;
    a.actions = actions; a.probs = probs; a.attn = attn;
    TrunkW tw{ w->enc_w1t, w->enc_b1, w->enc_w2t, w->enc_b2, w->attn_wt, w->gcn_w, w->gcn_b };
    PolHead ph{ w->hd_w1t, w->hd_b1, w->hd_w2t, w->hd_b2, w->hd_w3t, w->hd_b3, w->hd_w4t, w->hd_b4, w->n_act };
    return launch_fwd<0>(a, tw, ph, CritHead{}, stream);
}

extern "C" int cm_critic_forward(const cm_critic_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                                 const float *channels, float *values, void *stream) {
    if (!w || !obs || !values) return set_error(CM_ERR_ARG, "cm_critic_forward: null argument");
    if (n_samples <= 0) return CM_OK;
    if (w->enc_hidden != EH || w->emb != EMB || w->dec_hidden != DH)
        return set_error(CM_ERR_ARG, "cm_critic_forward: only the reference layer sizes (128 | 64 | 64) are built");
    if (w->d < 1 || w->d > 2 * SE - 4 || w->n_agents < 1 || w->n_hops < 0) return set_error(CM_ERR_ARG, "cm_critic_forward: bad dims");
    if (use_mfma()) {
        const int rc = critic_forward_mfma(w, n_samples, obs, dist_adj, channels, values, stream);
        if (rc <= 0) return rc;
    }
    FwdArgs a{};
    a.S = n_samples; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = dist_adj; a.chan = channels; a.values = values; a.no_residual = w->no_residual;
    TrunkW tw{ w->enc_w1t, w->enc_b1, w->enc_w2t, w->enc_b2, w->attn_wt, w->gcn_w, w->gcn_b };
    CritHead chd{ w->dec_w1t, w->dec_b1, w->dec_w2t, w->dec_b2 };
    return launch_fwd<1>(a, tw, PolHead{}, chd, stream);
}
