// cm_policy_w.hip - operand pack and stand-alone launcher of the wave-owned teams-of-4 policy forward (cm_policy_w_dev.h;
// reference: comm_categorical_mlp_policy.py:98-119 get_actions).  The fused rollout step (cm_fused.hip) instantiates the
// same device body in front of the env phase; cm_policy_forward reaches this launcher for shapes mw::shape_ok_w accepts.
// COMMARL_POLICY_KERNEL=h keeps the workgroup-tiled f16-split kernel (cm_policy_h.hip), =f32 the all-f32 one.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "cm_internal.h"
#include "cm_policy_w_dev.h"

namespace cm {
namespace mw {

template <int LHOPS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void fwd_w_kernel(FwdArgs a, WeightsW w) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_w[];
    fwd_body_w<LHOPS>(a, w, lds_w, blockIdx.x, nullptr);
}

// Training forward: a persistent workgroup per CU stages the weights ONCE and walks its share of the 64-row blocks (a wave owns a
// 16-row tile at a time, one wave per SIMD, nothing but registers between two layers); every activation the backward needs is
// stored from the epilogue registers (policy_tile_w<.., TRAIN>).
template <int LHOPS, int HEADK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void train_fwd_w_kernel(FwdArgs a, WeightsW w, int n_blk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_w[];
    stage_w<LHOPS>(w, lds_w, thread_x());
    ResidentW res{};
    if constexpr (HEADK == 0) res.fetch<LHOPS>(w, thread_x() & 63);      // (the critic's head has no register-resident layer)
    __syncthreads();
    for (int blk = blockIdx.x; blk < n_blk; blk += gridDim.x) {
        asm volatile("" ::: "memory");                       // keep each block's loads inside its iteration
        policy_tile_w<LHOPS, false, true, HEADK>(a, w.n_act, res, lds_w, blk, nullptr);
    }
}

// the critic's bias block in the policy's BiasMap: trunk as there, dec_b1 in the b1 slot (prescaled: tanh), dec_b2[0] in the b2 slot
__global__ void pack_bias_wc_kernel(const float *e1, const float *e2, const float *gb, const float *d1, const float *d2, int L, float *__restrict__ dst) {
    const BiasMap bm = bias_map(L);
    for (int i = threadIdx.x; i < BIAS_U4 * 4; i += blockDim.x) {
        float v = 0.0f;
        if (i < bm.e2) v = e1[i] * TANH_PRESCALE;
        else if (i < bm.g) v = e2[i - bm.e2] * TANH_PRESCALE;
        else if (i < bm.b1) v = ((gb && i - bm.g < L * EMB) ? gb[i - bm.g] : 0.0f) * TANH_PRESCALE;
        else if (i < bm.b1 + EMB) v = d1[i - bm.b1] * TANH_PRESCALE;
        else if (i == bm.b2) v = d2[0];
        dst[i] = v;
    }
}

// Wt [K][OUT] f32 (the ABI's transposed weights) -> A fragments with the wave-owned k order.
// dst uint4 index ((ct * KB + q) * 2 + plane) * 64 + lane = halves e = 0..7 of W[o(ct, lane & 15)][k(q, lane >> 4, e)];
//   natural = 1 (first layer: the observation arrives in memory order): k = 32 q + 8 g + e;  else k = kmap(q, g, e)
//   logits  = 1 (last layer, 32 output slots): slot o < 4 -> action o, slot 16 -> action 4, every other slot zero
//   scale: 2 log2(e) for the layers whose output goes through tanh (the kernel's tanh is 1 - 2 / (2^v + 1)), else 1
__global__ void pack_layer_w_kernel(const float *__restrict__ Wt, int K, int OUT, int KB, int CT, int natural, int logits, float scale,
                                    uint4 *__restrict__ dst, int *__restrict__ bad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= CT * KB * 64) return;
    const int lane = idx & 63, blk = idx >> 6, q = blk % KB, ct = blk / KB, g = lane >> 4;
    int o = 16 * ct + (lane & 15);
    if (logits) o = o < 4 ? o : (o == 16 ? 4 : -1);
    v8h hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = natural ? 32 * q + 8 * g + e : kmap(q, g, e);
        const float w = ((k < K && o >= 0 && o < OUT) ? Wt[(size_t)k * OUT + o] : 0.0f) * scale;
        if (bad && !(fabsf(w) <= 65504.0f)) atomicOr(bad, 1);
        h16 h, l;
        split_u(w, h, l);                                    // unscaled lo plane (cm_policy_w_dev.h)
        hi[e] = h; lo[e] = l;
    }
    dst[((size_t)blk * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    dst[((size_t)blk * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

// the bias block as the kernel keeps it in LDS (BiasMap order; the logits' bias in the two-tile slot order)
__global__ void pack_bias_w_kernel(const float *e1, const float *e2, const float *gb, const float *b1, const float *b2, const float *b3,
                                   const float *b4, int L, int n_act, float *__restrict__ dst) {
    const BiasMap bm = bias_map(L);
    for (int i = threadIdx.x; i < BIAS_U4 * 4; i += blockDim.x) {
        float v = 0.0f;
        if (i >= bm.total) v = 0.0f;
        else if (i < bm.e2) v = e1[i];
        else if (i < bm.g) v = e2[i - bm.e2];
        else if (i < bm.b1) v = (gb && i - bm.g < L * EMB) ? gb[i - bm.g] : 0.0f;
        else if (i < bm.b2) v = b1[i - bm.b1];
        else if (i < bm.b3) v = b2[i - bm.b2];
        else if (i < bm.b4) v = b3[i - bm.b3];
        else { const int o = i - bm.b4, act = o < 4 ? o : (o == 16 ? 4 : -1); v = (act >= 0 && act < n_act) ? b4[act] : 0.0f; }
        dst[i] = i < bm.b4 ? v * TANH_PRESCALE : v;            // every bias but the logits' feeds a tanh
    }
}

static int pack_one_w(const float *Wt, int K, int OUT, int kp, int out_pad, bool natural, bool logits, bool tanh_layer, uint4 *dst, void *stream,
                      int *bad) {
    if (!Wt) return set_error(CM_ERR_ARG, "weight pack: null layer weight");
    const int KB = kp / 32, CT = out_pad / 16, total = CT * KB * 64;
    hipLaunchKernelGGL(pack_layer_w_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, Wt, K, OUT, KB, CT, natural ? 1 : 0,
                       logits ? 1 : 0, tanh_layer ? TANH_PRESCALE : 1.0f, dst, bad);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace mw

// COMMARL_POLICY_KERNEL: unset / "w" = wave-owned kernel where the shape allows; "h" / "f32" / "valu" = the older kernels
bool policy_w_enabled() {
    static const bool v = [] { const char *e = getenv("COMMARL_POLICY_KERNEL"); return !(e && (e[0] == 'h' || e[0] == 'f' || e[0] == 'v')); }();
    return v;
}

size_t policy_pack_w_bytes(const cm_policy_weights *w) {
    return mw::shape_ok_w(w->n_agents, w->d, w->n_hops, w->n_act) ? (size_t)mw::pack_w(w->n_hops).total * sizeof(uint4) : 0;
}

// `bad` = the caller's range flag (cm_policy_h.hip: a weight the f16 pair cannot carry), may be null
int policy_pack_w(const cm_policy_weights *w, void *dst, void *stream, int *bad) {
    if (!mw::shape_ok_w(w->n_agents, w->d, w->n_hops, w->n_act)) return CM_OK;
    const mw::PackW pk = mw::pack_w(w->n_hops);
    uint4 *P = reinterpret_cast<uint4 *>(dst);
    using namespace mf;
    if (int rc = mw::pack_one_w(w->enc_w1t, w->d, EH, mw::KH, EH, true, false, true, P + pk.enc1, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->enc_w2t, EH, EMB, EH, EMB, false, false, true, P + pk.enc2, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->attn_wt, EMB, EMB, EMB, EMB, false, false, false, P + pk.attn, stream, bad)) return rc;
    for (int l = 0; l < w->n_hops; ++l)
        if (int rc = mw::pack_one_w(w->gcn_w ? w->gcn_w + (size_t)l * EMB * EMB : nullptr, EMB, EMB, EMB, EMB, false, false, true,
                                    P + pk.gcn + l * mw::frag_u4(EMB, EMB), stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->hd_w1t, EMB, H1, EMB, H1, false, false, true, P + pk.x1, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->hd_w2t, H1, H2, H1, H2, false, false, true, P + pk.h2, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->hd_w3t, H2, H3, H2, H3, false, false, true, P + pk.h3, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->hd_w4t, H3, w->n_act, H3, 32, false, true, false, P + pk.h4, stream, bad)) return rc;
    if (!w->enc_b1 || !w->enc_b2 || !w->hd_b1 || !w->hd_b2 || !w->hd_b3 || !w->hd_b4) return set_error(CM_ERR_ARG, "weight pack: null bias");
    hipLaunchKernelGGL(mw::pack_bias_w_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w->enc_b1, w->enc_b2, w->gcn_b, w->hd_b1, w->hd_b2,
                       w->hd_b3, w->hd_b4, w->n_hops, w->n_act, reinterpret_cast<float *>(P + pk.bias));
    CM_HIP(hipGetLastError());
    return CM_OK;
}

// cm_policy_forward_saved_wave: 1 = no wave-owned instantiation for the shape (nothing launched)
int policy_forward_w_train(const cm_policy_weights *w, const void *w_pack, mf::FwdArgs a, void *stream) {
    if (!policy_w_enabled() || !mw::shape_ok_w(w->n_agents, w->d, w->n_hops, w->n_act) || a.avail) return 1;
    const mw::WeightsW ww{ reinterpret_cast<const uint4 *>(w_pack), w->n_act };
    const size_t lds = mw::lds_policy_bytes(w->n_hops);
    const int n_blk = (a.S + mw::WG_ENVS - 1) / mw::WG_ENVS, blocks = std::min(n_blk, cm::cu_count());
#define CM_TW(LH, HK)                                                                                                          \
    do {                                                                                                                       \
        static unsigned long long done = 0;                                                                                    \
        if (cm::dev_first(done))                                                                                               \
            CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mw::train_fwd_w_kernel<LH, HK>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       160 * 1024));                                                                           \
        hipLaunchKernelGGL((mw::train_fwd_w_kernel<LH, HK>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, ww, n_blk); \
    } while (0)
    if (w->n_hops == 1) CM_TW(1, 0); else CM_TW(2, 0);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

// ---- the critic on the same kernel (training forward only): its fragments in the policy's image layout ----
static bool critic_w_ok(const cm_critic_weights *w) {
    return policy_w_enabled() && w->n_agents == 4 && w->d <= mw::KH && (w->n_hops == 1 || w->n_hops == 2) && w->enc_hidden == mf::EH &&
           w->emb == mf::EMB && w->dec_hidden == mf::DH && mf::DH == mf::EMB;
}
size_t critic_pack_w_bytes(const cm_critic_weights *w) { return critic_w_ok(w) ? (size_t)mw::pack_w(w->n_hops).lds_u4 * sizeof(uint4) : 0; }

int critic_pack_w(const cm_critic_weights *w, void *dst, void *stream, int *bad) {
    if (!critic_w_ok(w)) return CM_OK;
    const mw::PackW pk = mw::pack_w(w->n_hops);
    uint4 *P = reinterpret_cast<uint4 *>(dst);
    using namespace mf;
    if (int rc = mw::pack_one_w(w->enc_w1t, w->d, EH, mw::KH, EH, true, false, true, P + pk.enc1, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->enc_w2t, EH, EMB, EH, EMB, false, false, true, P + pk.enc2, stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->attn_wt, EMB, EMB, EMB, EMB, false, false, false, P + pk.attn, stream, bad)) return rc;
    for (int l = 0; l < w->n_hops; ++l)
        if (int rc = mw::pack_one_w(w->gcn_w ? w->gcn_w + (size_t)l * EMB * EMB : nullptr, EMB, EMB, EMB, EMB, false, false, true,
                                    P + pk.gcn + l * mw::frag_u4(EMB, EMB), stream, bad)) return rc;
    if (int rc = mw::pack_one_w(w->dec_w1t, EMB, DH, EMB, DH, false, false, true, P + pk.x1, stream, bad)) return rc;      // x1 slot
    if (int rc = mw::pack_one_w(w->dec_w2t, DH, 1, DH, 16, false, false, false, P + pk.h3, stream, bad)) return rc;       // h3 slot, 16-wide tile
    if (!w->enc_b1 || !w->enc_b2 || !w->dec_b1 || !w->dec_b2) return set_error(CM_ERR_ARG, "weight pack: null bias");
    hipLaunchKernelGGL(mw::pack_bias_wc_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, w->enc_b1, w->enc_b2, w->gcn_b, w->dec_b1, w->dec_b2,
                       w->n_hops, reinterpret_cast<float *>(P + pk.bias));
    CM_HIP(hipGetLastError());
    return CM_OK;
}

// cm_critic_forward_saved_wave: 1 = no wave-owned instantiation for the shape (nothing launched)
int critic_forward_w_train(const cm_critic_weights *w, const void *w_pack, mf::FwdArgs a, void *stream) {
    if (!critic_w_ok(w)) return 1;
    const mw::WeightsW ww{ reinterpret_cast<const uint4 *>(w_pack), 0 };
    const size_t lds = mw::lds_policy_bytes(w->n_hops);
    const int n_blk = (a.S + mw::WG_ENVS - 1) / mw::WG_ENVS, blocks = std::min(n_blk, cm::cu_count());
    if (w->n_hops == 1) CM_TW(1, 1); else CM_TW(2, 1);
#undef CM_TW
    CM_HIP(hipGetLastError());
    return CM_OK;
}

mw::WeightsW weights_w(const cm_policy_weights *w, const void *w_pack) {
    return mw::WeightsW{ reinterpret_cast<const uint4 *>(w_pack), w->n_act };
}

// Returns 1 when the shape has no wave-owned instantiation (the caller runs the workgroup-tiled kernel).
int policy_forward_w(const cm_policy_weights *w, const void *w_pack, mf::FwdArgs a, void *stream) {
    if (!policy_w_enabled() || !mw::shape_ok_w(w->n_agents, w->d, w->n_hops, w->n_act) || a.sv_on) return 1;
    const mw::WeightsW ww = weights_w(w, w_pack);
    const size_t lds = mw::lds_policy_bytes(w->n_hops);
    const int blocks = (a.S + mw::WG_ENVS - 1) / mw::WG_ENVS;
#define CM_FW(LH)                                                                                                              \
    do {                                                                                                                       \
        static unsigned long long done = 0;                                                                                    \
        if (cm::dev_first(done))                                                                                               \
            CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mw::fwd_w_kernel<LH>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       160 * 1024));                                                                           \
        hipLaunchKernelGGL((mw::fwd_w_kernel<LH>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, ww);                  \
    } while (0)
    static const bool want_probe = getenv("COMMARL_FWD_PROBE") != nullptr;
    unsigned long long *dbuf = nullptr;
    if (want_probe) {                                    // diagnostic: per-phase shader clocks of thread 0 of every workgroup
        CM_HIP(hipMalloc(&dbuf, (size_t)blocks * mf::NPROBE * sizeof(unsigned long long)));
        CM_HIP(hipMemset(dbuf, 0, (size_t)blocks * mf::NPROBE * sizeof(unsigned long long)));
        a.probe = dbuf;
        if (w->n_hops == 1) CM_FW(1); else CM_FW(2);     // warm (instruction cache, L2)
    }
    if (w->n_hops == 1) CM_FW(1); else CM_FW(2);
    if (want_probe) {
        CM_HIP(hipDeviceSynchronize());
        std::vector<unsigned long long> h((size_t)blocks * mf::NPROBE);
        CM_HIP(hipMemcpy(h.data(), dbuf, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        CM_HIP(hipFree(dbuf));
        double sum[mf::NPROBE] = { 0 };
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < blocks; ++b) {
            tmin = std::min(tmin, h[(size_t)b * mf::NPROBE]);
            for (int i = 1; i <= 10; ++i) { sum[i] += (double)(h[(size_t)b * mf::NPROBE + i] - h[(size_t)b * mf::NPROBE]); tmax = std::max(tmax, h[(size_t)b * mf::NPROBE + i]); }
        }
        fprintf(stderr, "[fwd_w probe] blocks=%d span=%llu ticks; mean ticks since entry:", blocks, tmax - tmin);
        for (int i = 1; i <= 10; ++i) fprintf(stderr, " p%d=%.0f", i, sum[i] / blocks);
        fprintf(stderr, "  (block 0:");
        for (int i = 1; i <= 10; ++i) fprintf(stderr, " %llu", h[i] - h[0]);
        fprintf(stderr, ")\n");
    }
#undef CM_FW
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace cm
