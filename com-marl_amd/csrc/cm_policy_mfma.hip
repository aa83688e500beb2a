// cm_policy_mfma.hip - fused Comm-DP policy / critic forward on the gfx950 matrix cores.
//
// Same computation and C-ABI as cm_policy.hip (reference:
// com_marl/torch/policies/comm_categorical_mlp_policy.py:48-119, modules/comm_base_net.py:80-108,
// attention_module.py:26-51, graph_conv_module.py:51-72, baselines/comm_base_critic.py:91-114),
// with every dense per-agent layer on v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: exact
// f32, needed for the 1e-5 parity bar; this is the 157 TFLOP/s roof the kernel is priced on).
//
// Work decomposition (256 threads = 4 waves per workgroup, EPB whole envs = rows agent rows):
//   * activations live in LDS [rows_pad][stride] with stride == 2 (mod 32) words, so the A-operand
//     read  A[row = lane&15][k = 4*kk + (lane>>4)]  is bank-conflict free;
//   * a wave owns 1-2 column tiles (16 outputs) of a layer and ALL row tiles: its B fragments
//     (weights, k-major [in][out], read from L2 once per layer per wave) sit in registers for the
//     whole layer, two row tiles are accumulated alternately so the 40-cycle dependent-accumulator
//     latency of the 16x16x4 instruction is covered by the 32-cycle issue of the other tile;
//   * the tiny per-env parts (N x N attention softmax, mask + renorm, A.(HW), 32->5 head,
//     categorical sample) stay on the VALU against the same LDS tiles.
// HBM traffic is the algorithmic minimum: obs (+ masks) in, actions / probs / attention out.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "cm_internal.h"
#include "cm_rng.h"
#include "cm_policy_mfma_dev.h"

namespace cm {
namespace mf {

template <int HEAD, int KPAD, int MAXMK, int NW>
__global__ __launch_bounds__(64 * NW) void fwd_mfma_kernel(FwdArgs a, TrunkW tw, PolHead ph, CritHead chd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fwd_body<HEAD, KPAD, MAXMK, NW>(a, tw, ph, chd, lds, blockIdx.x, nullptr);
}


template <int HEAD, int KPAD, int MAXMK, int NW = 4>
static int launch(FwdArgs a, const TrunkW &tw, const PolHead &ph, const CritHead &chd, void *stream) {
    a.EPB = pick_epb(a.N);
    const int rows_cap = (a.EPB * a.N + 15) & ~15;
    const size_t lds = lds_floats(rows_cap, a.EPB, a.N) * sizeof(float);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "policy forward: n_agents too large for the 160 KB LDS tile");
    static unsigned long long attr_set = 0;
    if (cm::dev_first(attr_set)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_mfma_kernel<HEAD, KPAD, MAXMK, NW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int blocks = (a.S + a.EPB - 1) / a.EPB;
    static const bool want_probe = getenv("COMMARL_FWD_PROBE") != nullptr;
    if (want_probe) {                                    // diagnostic: per-phase shader clocks, averaged over workgroups
        unsigned long long *dbuf = nullptr;
        const size_t nb = (size_t)blocks * NPROBE * sizeof(unsigned long long);
        CM_HIP(hipMalloc(&dbuf, nb));
        CM_HIP(hipMemset(dbuf, 0, nb));
        a.probe = dbuf;
        for (int rep = 0; rep < 3; ++rep)
            hipLaunchKernelGGL((fwd_mfma_kernel<HEAD, KPAD, MAXMK, NW>), dim3(blocks), dim3(64 * NW), lds, (hipStream_t)stream, a, tw, ph, chd);
        CM_HIP(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)blocks * NPROBE);
        CM_HIP(hipMemcpy(h.data(), dbuf, nb, hipMemcpyDeviceToHost));
        CM_HIP(hipFree(dbuf));
        double sum[NPROBE] = { 0 }; int cnt[NPROBE] = { 0 };
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < blocks; ++b) {
            int prev = 0;
            tmin = std::min(tmin, h[(size_t)b * NPROBE]);
            for (int i = 1; i < NPROBE; ++i) {
                const unsigned long long t = h[(size_t)b * NPROBE + i];
                if (!t) continue;
                sum[i] += (double)(t - h[(size_t)b * NPROBE + prev]); ++cnt[i]; prev = i; tmax = std::max(tmax, t);
            }
        }
        fprintf(stderr, "[fwd probe] HEAD=%d blocks=%d span=%llu clk; mean clk per phase:", HEAD, blocks, tmax - tmin);
        for (int i = 1; i < NPROBE; ++i) if (cnt[i]) fprintf(stderr, " p%d=%.0f", i, sum[i] / cnt[i]);
        fprintf(stderr, "\n");
        a.probe = nullptr;
        return CM_OK;
    }
    hipLaunchKernelGGL((fwd_mfma_kernel<HEAD, KPAD, MAXMK, NW>), dim3(blocks), dim3(64 * NW), lds, (hipStream_t)stream, a, tw, ph, chd);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

template <int HEAD>
static int dispatch(const FwdArgs &a, const TrunkW &tw, const PolHead &ph, const CritHead &chd, void *stream) {
    int kpad = (a.d + 15) & ~15;
    if (kpad == 16) kpad = 32;            // smallest instantiation (obs dims 1..32)
    if (kpad == 48) kpad = 64;
    if (a.N > 128) return 1;             // N x N MFMA tiles are built for teams of up to 128 agents
    const int nn = a.N * a.N;
    // N x N products on MFMA from N = 32 up (measured: at N = 24 the padded 32 x 32 tiles lose to the VALU form)
    static const int mk_min = [] { const char *e = getenv("COMMARL_MK_MIN"); return e ? atoi(e) : 16; }();   // N x N products on MFMA tiles from 16 agents up
    const int mk = a.N < mk_min ? 0 : (a.N <= 80 ? 25 : 64);    // (row blocks) x (column blocks) of 16
    (void)nn;
    const bool quad = a.N == 4 && pick_epb(4) * 4 <= 32;    // teams of 4: the register-resident attention / aggregation kernel
    // large teams run 8-wave workgroups (one workgroup per CU fits in LDS: two waves per SIMD hide each other's
    // latencies); COMMARL_FWD_WAVES=4 selects the 4-wave build of the same code for A/B timing
    static const bool w8_on = [] { const char *e = getenv("COMMARL_FWD_WAVES"); return !(e && e[0] == '4'); }();
    static const int w8_min = [] { const char *e = getenv("COMMARL_FWD_W8MIN"); return e ? atoi(e) : 32; }();
    const bool w8 = w8_on && a.N >= w8_min;   // measured: N = 54 245 -> 178 us, N = 72 280 -> 213 us; N = 24 (48 rows) is faster on 4 waves
#define CM_FWD(K) (quad ? launch<HEAD, K, -1>(a, tw, ph, chd, stream) : mk == 0 ? launch<HEAD, K, 0>(a, tw, ph, chd, stream) \
                   : mk == 25 ? (w8 ? launch<HEAD, K, 15, 8>(a, tw, ph, chd, stream) : launch<HEAD, K, 25>(a, tw, ph, chd, stream)) \
                              : (w8 ? launch<HEAD, K, 32, 8>(a, tw, ph, chd, stream) : launch<HEAD, K, 64>(a, tw, ph, chd, stream)))
    switch (kpad) {      // obs dims of the reference scenarios: PP sen1 21, CO sen1 29, PP sen2 53, CO sen2 77 (+clock 78)
    case 32: return CM_FWD(32);
    case 64: return CM_FWD(64);
    case 80: return CM_FWD(80);
    default: return 1;   // caller falls back to the generic VALU kernel
    }
}

// ---- operand pack ---------------------------------------------------------------------------------------------
// dst[((ct*KQ + kq)*64 + lane)*4 + j] = Wt[k = 16kq + 4(lane>>4) + j][col = 16ct + (lane&15)], zero outside [K, OUT]
__global__ void pack_layer_kernel(const float *__restrict__ Wt, int K, int OUT, int KQ, int CT, float *__restrict__ dst) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CT * KQ * 256) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8, kq = blk % KQ, ct = blk / KQ;
    const int k = 16 * kq + 4 * (lane >> 4) + j, col = 16 * ct + (lane & 15);
    dst[idx] = (k < K && col < OUT) ? Wt[(size_t)k * OUT + col] : 0.0f;
}



int pack_one(const float *Wt, int K, int OUT, int kpad, int out_pad, float *dst, void *stream) {
    if (!Wt) return set_error(CM_ERR_ARG, "weight pack: null layer weight");
    const int KQ = kpad / 16, CT = out_pad / 16, total = CT * KQ * 256;
    hipLaunchKernelGGL(pack_layer_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, Wt, K, OUT, KQ, CT, dst);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

static int pack_trunk(int d, int L, const float *w1t, const float *w2t, const float *wat, const float *gw, int kpad,
                      const PackLayout &lo, float *pack, void *stream) {
    if (int rc = pack_one(w1t, d, EH, kpad, EH, pack + lo.enc1, stream)) return rc;
    if (int rc = pack_one(w2t, EH, EMB, EH, EMB, pack + lo.enc2, stream)) return rc;
    if (int rc = pack_one(wat, EMB, EMB, EMB, EMB, pack + lo.attn, stream)) return rc;
    for (int l = 0; l < L; ++l)
        if (int rc = pack_one(gw ? gw + (size_t)l * EMB * EMB : nullptr, EMB, EMB, EMB, EMB, pack + lo.gcn + (size_t)l * EMB * EMB, stream)) return rc;
    return CM_OK;
}

}  // namespace mf

// cm_policy_h.hip: the f16-split kernel (default) and its operand pack, stored BEHIND the f32 pack in the caller's buffer
bool policy_h_enabled();
size_t policy_pack_h_bytes(int d, int L, bool policy);
int policy_pack_h(const cm_policy_weights *w, void *dst, int sections, void *stream);
int critic_pack_h(const cm_critic_weights *w, void *dst, int sections, void *stream);
int policy_forward_h(const cm_policy_weights *w, const void *h_pack, mf::FwdArgs a, void *stream);
// cm_policy_w.hip: the wave-owned teams-of-4 kernel (default where the shape allows); its fragments sit behind the f16 pack
size_t policy_pack_w_bytes(const cm_policy_weights *w);
int policy_forward_w(const cm_policy_weights *w, const void *w_pack, mf::FwdArgs a, void *stream);
int critic_forward_h(const cm_critic_weights *w, const void *h_pack, mf::FwdArgs a, void *stream);

bool policy_shape_ok(const cm_policy_weights *w) {
    return w && w->enc_hidden == mf::EH && w->emb == mf::EMB && w->h1 == mf::H1 && w->h2 == mf::H2 && w->h3 == mf::H3 &&
           w->n_act >= 1 && w->n_act <= mf::MAX_ACT && w->n_agents >= 1 && w->n_agents <= 128 && w->n_hops >= 0 &&
           mf::kpad_of(w->d) != 0;
}
static bool critic_shape_ok(const cm_critic_weights *w) {
    return w && w->enc_hidden == mf::EH && w->emb == mf::EMB && w->dec_hidden == mf::DH && w->n_agents >= 1 &&
           w->n_agents <= 128 && w->n_hops >= 0 && mf::kpad_of(w->d) != 0;
}

// entry points used by cm_policy.hip's C-ABI functions; return 1 when the shape has no MFMA instantiation or the
// caller supplied no operand pack (the generic VALU kernel then runs)
int policy_forward_mfma(const cm_policy_weights *w, int32_t S, const float *obs, const float *avail, const float *adj,
                        const float *chan, uint64_t seed, int32_t env_id_offset, uint32_t policy_step,
                        const uint32_t *step_base, int32_t greedy, int32_t *actions, float *probs, float *attn,
                        void *stream) {
    if (!w->mfma_pack || !policy_shape_ok(w)) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.avail = avail; a.adj = adj; a.chan = chan;
    a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.policy_step = policy_step; a.step_base = step_base;
    a.env_id_offset = env_id_offset; a.greedy = greedy; a.no_residual = w->no_residual;
    a.actions = actions; a.probs = probs; a.attn = attn;
    { const char *e = getenv("COMMARL_FWD_STOP"); a.stop = e ? atoi(e) : 0; }
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, true);
    const float *P = w->mfma_pack;
    if (policy_pack_w_bytes(w)) {
        const int rc = policy_forward_w(w, reinterpret_cast<const char *>(P + lo.total) + policy_pack_h_bytes(w->d, w->n_hops, true), a, stream);
        if (rc <= 0) return rc;
    }
    if (policy_h_enabled()) {
        const int rc = policy_forward_h(w, P + lo.total, a, stream);
        if (rc <= 0) return rc;
    }
    mf::TrunkW tw{ P + lo.enc1, w->enc_b1, P + lo.enc2, w->enc_b2, P + lo.attn, P + lo.gcn, w->gcn_b };
    mf::PolHead ph{ P + lo.x1, w->hd_b1, P + lo.h2, w->hd_b2, P + lo.h3, w->hd_b3, P + lo.h4, w->hd_b4, w->n_act };
    return mf::dispatch<0>(a, tw, ph, mf::CritHead{}, stream);
}

int critic_forward_mfma(const cm_critic_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                        float *values, void *stream) {
    if (!w->mfma_pack || !critic_shape_ok(w)) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = adj; a.chan = chan; a.values = values; a.no_residual = w->no_residual;
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, false);
    const float *P = w->mfma_pack;
    if (policy_h_enabled()) {
        const int rc = critic_forward_h(w, P + lo.total, a, stream);
        if (rc <= 0) return rc;
    }
    mf::TrunkW tw{ P + lo.enc1, w->enc_b1, P + lo.enc2, w->enc_b2, P + lo.attn, P + lo.gcn, w->gcn_b };
    mf::CritHead chd{ P + lo.x1, w->dec_b1, w->dec_w2t, w->dec_b2 };
    return mf::dispatch<1>(a, tw, mf::PolHead{}, chd, stream);
}

}  // namespace cm

static bool saved_shape_ok(int N, int L, int d) { return N >= 1 && N <= 128 && L <= 4 && L >= 0 && d <= 96; }
static void set_saves(cm::mf::FwdArgs &a, const cm_fwd_saves *sv) {
    a.sv_on = 1;
    a.sv_a1 = sv->a1; a.sv_e = sv->e; a.sv_q = sv->q; a.sv_x1 = sv->x1; a.sv_x2 = sv->x2; a.sv_x3 = sv->x3; a.sv_out = sv->out;
    for (int l = 0; l < 4; ++l) { a.sv_hw[l] = sv->hw[l]; a.sv_h[l] = sv->h[l]; }
}

extern "C" int cm_policy_forward_saved(const cm_policy_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                                       float *attn, const cm_fwd_saves *sv, void *stream) {
    using namespace cm;
    if (!w || !obs || !sv) return set_error(CM_ERR_ARG, "cm_policy_forward_saved: null argument");
    if (S <= 0) return CM_OK;
    if (!w->mfma_pack || !policy_shape_ok(w) || !saved_shape_ok(w->n_agents, w->n_hops, w->d)) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = adj; a.chan = chan; a.attn = attn; a.no_residual = w->no_residual;
    set_saves(a, sv);
    a.probs = sv->probs;
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, true);
    return policy_forward_h(w, w->mfma_pack + lo.total, a, stream);
}

namespace cm {                                       // cm_policy_w.hip
int policy_forward_w_train(const cm_policy_weights *w, const void *w_pack, mf::FwdArgs a, void *stream);
int critic_forward_w_train(const cm_critic_weights *w, const void *w_pack, mf::FwdArgs a, void *stream);
size_t critic_pack_w_bytes(const cm_critic_weights *w);
}

extern "C" int cm_policy_forward_saved_wave(const cm_policy_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                                            float *attn, const cm_fwd_saves *sv, void *stream) {
    using namespace cm;
    if (!w || !obs || !sv) return set_error(CM_ERR_ARG, "cm_policy_forward_saved_wave: null argument");
    if (S <= 0) return CM_OK;
    if (!w->mfma_pack || !policy_shape_ok(w) || policy_pack_w_bytes(w) == 0) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = adj; a.chan = chan; a.attn = attn; a.no_residual = w->no_residual;
    set_saves(a, sv);
    a.probs = sv->probs;
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, true);
    return policy_forward_w_train(w, reinterpret_cast<const char *>(w->mfma_pack + lo.total) + policy_pack_h_bytes(w->d, w->n_hops, true), a, stream);
}

extern "C" int cm_critic_forward_saved(const cm_critic_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                                       float *attn, float *values, const cm_fwd_saves *sv, void *stream) {
    using namespace cm;
    if (!w || !obs || !sv || !values) return set_error(CM_ERR_ARG, "cm_critic_forward_saved: null argument");
    if (S <= 0) return CM_OK;
    if (!w->mfma_pack || !critic_shape_ok(w) || !saved_shape_ok(w->n_agents, w->n_hops, w->d)) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = adj; a.chan = chan; a.attn = attn; a.values = values; a.no_residual = w->no_residual;
    set_saves(a, sv);
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, false);
    return critic_forward_h(w, w->mfma_pack + lo.total, a, stream);
}

extern "C" size_t cm_policy_pack_bytes(const cm_policy_weights *w) {
    if (!cm::policy_shape_ok(w)) return 0;
    return cm::mf::pack_layout(cm::mf::kpad_of(w->d), w->n_hops, true).total * sizeof(float) + cm::policy_pack_h_bytes(w->d, w->n_hops, true) +
           cm::policy_pack_w_bytes(w);
}

extern "C" int cm_policy_pack_sections(const cm_policy_weights *w, float *pack, int32_t sections, void *stream) {
    using namespace cm;
    if (!w || !pack) return set_error(CM_ERR_ARG, "cm_policy_pack: null argument");
    if (!policy_shape_ok(w)) return set_error(CM_ERR_ARG, "cm_policy_pack: this shape has no matrix-core instantiation (cm_policy_pack_bytes() == 0)");
    const int kpad = mf::kpad_of(w->d);
    const mf::PackLayout lo = mf::pack_layout(kpad, w->n_hops, true);
    if (sections & CM_PACK_F32) {
        if (int rc = mf::pack_trunk(w->d, w->n_hops, w->enc_w1t, w->enc_w2t, w->attn_wt, w->gcn_w, kpad, lo, pack, stream)) return rc;
        if (int rc = mf::pack_one(w->hd_w1t, mf::EMB, mf::H1, mf::EMB, mf::H1, pack + lo.x1, stream)) return rc;
        if (int rc = mf::pack_one(w->hd_w2t, mf::H1, mf::H2, mf::H1, mf::H2, pack + lo.h2, stream)) return rc;
        if (int rc = mf::pack_one(w->hd_w3t, mf::H2, mf::H3, mf::H2, mf::H3, pack + lo.h3, stream)) return rc;
        if (int rc = mf::pack_one(w->hd_w4t, mf::H3, w->n_act, mf::H3, 16, pack + lo.h4, stream)) return rc;
    }
    return policy_pack_h(w, pack + lo.total, sections, stream);   // the f16 (hi, lo) fragments (+ the wave-owned section), behind the f32 ones
}

extern "C" int cm_policy_pack(const cm_policy_weights *w, float *pack, void *stream) {
    return cm_policy_pack_sections(w, pack, CM_PACK_ALL, stream);
}

extern "C" size_t cm_critic_pack_bytes(const cm_critic_weights *w) {
    if (!cm::critic_shape_ok(w)) return 0;
    return cm::mf::pack_layout(cm::mf::kpad_of(w->d), w->n_hops, false).total * sizeof(float) + cm::policy_pack_h_bytes(w->d, w->n_hops, false) +
           cm::critic_pack_w_bytes(w);
}

extern "C" int cm_critic_forward_saved_wave(const cm_critic_weights *w, int32_t S, const float *obs, const float *adj, const float *chan,
                                            float *attn, float *values, const cm_fwd_saves *sv, void *stream) {
    using namespace cm;
    if (!w || !obs || !sv) return set_error(CM_ERR_ARG, "cm_critic_forward_saved_wave: null argument");
    if (S <= 0) return CM_OK;
    if (!w->mfma_pack || !critic_shape_ok(w) || critic_pack_w_bytes(w) == 0) return 1;
    mf::FwdArgs a{};
    a.S = S; a.N = w->n_agents; a.d = w->d; a.L = w->n_hops;
    a.obs = obs; a.adj = adj; a.chan = chan; a.attn = attn; a.values = values; a.no_residual = w->no_residual;
    set_saves(a, sv);
    const mf::PackLayout lo = mf::pack_layout(mf::kpad_of(w->d), w->n_hops, false);
    return critic_forward_w_train(w, reinterpret_cast<const char *>(w->mfma_pack + lo.total) + policy_pack_h_bytes(w->d, w->n_hops, false), a, stream);
}

extern "C" int cm_critic_pack_sections(const cm_critic_weights *w, float *pack, int32_t sections, void *stream) {
    using namespace cm;
    if (!w || !pack) return set_error(CM_ERR_ARG, "cm_critic_pack: null argument");
    if (!critic_shape_ok(w)) return set_error(CM_ERR_ARG, "cm_critic_pack: this shape has no matrix-core instantiation (cm_critic_pack_bytes() == 0)");
    const int kpad = mf::kpad_of(w->d);
    const mf::PackLayout lo = mf::pack_layout(kpad, w->n_hops, false);
    if (sections & CM_PACK_F32) {
        if (int rc = mf::pack_trunk(w->d, w->n_hops, w->enc_w1t, w->enc_w2t, w->attn_wt, w->gcn_w, kpad, lo, pack, stream)) return rc;
        if (int rc = mf::pack_one(w->dec_w1t, mf::EMB, mf::DH, mf::EMB, mf::DH, pack + lo.x1, stream)) return rc;
    }
    return critic_pack_h(w, pack + lo.total, sections, stream);
}

extern "C" int cm_critic_pack(const cm_critic_weights *w, float *pack, void *stream) {
    return cm_critic_pack_sections(w, pack, CM_PACK_ALL, stream);
}
