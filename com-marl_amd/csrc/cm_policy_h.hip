// cm_policy_h.hip - launchers and operand pack of the f16-split policy / critic forward (cm_policy_h_dev.h): the dense
// per-agent layers of CommBaseNet / the policy head / the critic head (reference: comm_base_net.py:80-108,
// comm_categorical_mlp_policy.py:48-96, comm_base_critic.py:91-114) on v_mfma_f32_16x16x32_f16 with every operand
// carried as an f16 (hi, 2^12-scaled lo) pair - f32-grade results at a third of the matrix-pipe time of the f32
// instruction.  Entered from cm_policy_mfma.hip's policy_forward_mfma / critic_forward_mfma; COMMARL_POLICY_KERNEL=f32
// keeps the all-f32 kernel.
#include <stdio.h>
#include <stdlib.h>

#include "cm_internal.h"
#include "cm_rng.h"
#include "cm_policy_h_dev.h"

namespace cm {
namespace mh {

template <int HEAD, int KH, int MAXMK, int NW>
__global__ __launch_bounds__(64 * NW) void fwd_h_kernel(FwdArgs a, TrunkH tw, PolHeadH ph, CritHeadH chd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_h[];
    fwd_body_h<HEAD, KH, MAXMK, NW>(a, tw, ph, chd, lds_h, blockIdx.x, nullptr);
}

// teams of 4, every workgroup full (n_samples a multiple of 8): the constant-shape build
template <int HEAD, int KH>
__global__ __launch_bounds__(256) void fwd_h_full_kernel(FwdArgs a, TrunkH tw, PolHeadH ph, CritHeadH chd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_h[];
    fwd_body_h<HEAD, KH, -1, 4, true, false, true>(a, tw, ph, chd, lds_h, blockIdx.x, nullptr);
}

// Teams of 4 at TRAINING batch sizes (tens of thousands of workgroups): the same body with late weight fetches, held to
// three waves per SIMD (<= 168 VGPRs) so that three workgroups share a CU instead of two - the rollout's 512 workgroups
// cannot use a third slot, a 34 k-workgroup grid can.
template <int HEAD, int KH>
__global__ __launch_bounds__(256, 3) void fwd_h_occ3_kernel(FwdArgs a, TrunkH tw, PolHeadH ph, CritHeadH chd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_h[];
    fwd_body_h<HEAD, KH, -1, 4, true, true>(a, tw, ph, chd, lds_h, blockIdx.x, nullptr);
}

template <int HEAD, int KH, int MAXMK, int NW = 4>
static int launch_h(FwdArgs a, const TrunkH &tw, const PolHeadH &ph, const CritHeadH &chd, void *stream) {
    a.EPB = mf::pick_epb(a.N);
    const int rows_cap = (a.EPB * a.N + 15) & ~15;
    const size_t lds = lds_map(rows_cap, a.EPB, a.N, MAXMK < 0 ? -1 : (MAXMK > 0 ? 1 : 0)).total;
    if (lds > 160 * 1024) return 1;                      // caller falls back (and reports the size limit there)
    static unsigned long long attr_set = 0;
    if (cm::dev_first(attr_set)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_h_kernel<HEAD, KH, MAXMK, NW>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int blocks = (a.S + a.EPB - 1) / a.EPB;
    if constexpr (MAXMK < 0 && NW == 4) {
        static const int occ_min = [] { const char *e = getenv("COMMARL_FWD_OCC3_MIN"); return e ? atoi(e) : 4096; }();   // workgroups; 0 = never
        if (occ_min > 0 && blocks >= occ_min) {
            static unsigned long long attr3 = 0;
            if (cm::dev_first(attr3)) {
                CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_h_occ3_kernel<HEAD, KH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            hipLaunchKernelGGL((fwd_h_occ3_kernel<HEAD, KH>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, tw, ph, chd);
            CM_HIP(hipGetLastError());
            return CM_OK;
        }
        static const bool full_on = [] { const char *e = getenv("COMMARL_FWD_FULL"); return !(e && e[0] == '0'); }();
        if (full_on && a.EPB == 8 && a.S % 8 == 0) {
            static unsigned long long attrf = 0;
            if (cm::dev_first(attrf)) {
                CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&fwd_h_full_kernel<HEAD, KH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            hipLaunchKernelGGL((fwd_h_full_kernel<HEAD, KH>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, tw, ph, chd);
            CM_HIP(hipGetLastError());
            return CM_OK;
        }
    }
    hipLaunchKernelGGL((fwd_h_kernel<HEAD, KH, MAXMK, NW>), dim3(blocks), dim3(64 * NW), lds, (hipStream_t)stream, a, tw, ph, chd);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

template <int HEAD>
static int dispatch_h(const FwdArgs &a, const TrunkH &tw, const PolHeadH &ph, const CritHeadH &chd, void *stream) {
    const int kh = kh_of(a.d);
    if (!kh || a.N > 128) return 1;
    static const int mk_min = [] { const char *e = getenv("COMMARL_MK_MIN"); return e ? atoi(e) : 16; }();
    const int mk = a.N < mk_min ? 0 : (a.N <= 80 ? 25 : 64);
    const bool quad = a.N == 4 && mf::pick_epb(4) * 4 <= 32;
    static const bool w8_on = [] { const char *e = getenv("COMMARL_FWD_WAVES"); return !(e && e[0] == '4'); }();
    static const int w8_min = [] { const char *e = getenv("COMMARL_FWD_W8MIN"); return e ? atoi(e) : 32; }();
    const bool w8 = w8_on && a.N >= w8_min;
#define CM_FWH(K) (quad ? launch_h<HEAD, K, -1>(a, tw, ph, chd, stream) : mk == 0 ? launch_h<HEAD, K, 0>(a, tw, ph, chd, stream) \
                   : mk == 25 ? (w8 ? launch_h<HEAD, K, 15, 8>(a, tw, ph, chd, stream) : launch_h<HEAD, K, 25>(a, tw, ph, chd, stream)) \
                              : (w8 ? launch_h<HEAD, K, 32, 8>(a, tw, ph, chd, stream) : launch_h<HEAD, K, 64>(a, tw, ph, chd, stream)))
    switch (kh) {
    case 32: return CM_FWH(32);
    case 64: return CM_FWH(64);
    case 96: return CM_FWH(96);
    default: return 1;
    }
#undef CM_FWH
}

// ---- operand pack: Wt [K][OUT] f32 (the ABI's transposed weights) -> A fragments of the transposed layer ----------
// dst uint4 index ((ct * KB + q) * 2 + plane) * 64 + lane = halves e = 0..7 of W[o = 16 ct + (lane & 15)][k = 32 q + 8 (lane >> 4) + e]
// *bad is raised when a weight cannot be carried by the (hi, lo) f16 pair: |w| > 65504 (the largest f16), inf or NaN would
// turn into +-inf planes silently - cm_policy_pack / cm_critic_pack refuse such a net (pack_range_check below)
// Every layer of a net in ONE launch: a table of (source, shape, destination) entries, block b works on the entry whose block range
// holds it (a PPO optimiser step re-packs both nets, ~16 layers: one launch each instead of one per layer).
constexpr int PACK_MAX_LAYERS = 12;
struct PackJobsH {
    const float *W[PACK_MAX_LAYERS]; uint4 *dst[PACK_MAX_LAYERS];
    int K[PACK_MAX_LAYERS], OUT[PACK_MAX_LAYERS], KB[PACK_MAX_LAYERS], CT[PACK_MAX_LAYERS], first_block[PACK_MAX_LAYERS + 1];
    int count;
};
__global__ void pack_layer_h_kernel(PackJobsH t, int *__restrict__ bad) {
    int j = 0;
    while (j + 1 < t.count && (int)blockIdx.x >= t.first_block[j + 1]) ++j;
    const float *__restrict__ Wt = t.W[j];
    uint4 *__restrict__ dst = t.dst[j];
    const int K = t.K[j], OUT = t.OUT[j], KB = t.KB[j], CT = t.CT[j];
    const int idx = ((int)blockIdx.x - t.first_block[j]) * 256 + threadIdx.x;   // one (ct, q, lane): both planes
    if (idx >= CT * KB * 64) return;
    const int lane = idx & 63, blk = idx >> 6, q = blk % KB, ct = blk / KB;
    const int o = 16 * ct + (lane & 15), k0 = 32 * q + 8 * (lane >> 4);
    v8h hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = k0 + e;
        const float w = (k < K && o < OUT) ? Wt[(size_t)k * OUT + o] : 0.0f;
        if (bad && !(fabsf(w) <= 65504.0f)) atomicOr(bad, 1);
        h16 h, l;
        split2(w, h, l);
        hi[e] = h; lo[e] = l;
    }
    dst[((size_t)blk * 2 + 0) * 64 + lane] = __builtin_bit_cast(uint4, hi);
    dst[((size_t)blk * 2 + 1) * 64 + lane] = __builtin_bit_cast(uint4, lo);
}

// One range flag per device: a device word the pack kernels raise and a pinned host word it is copied into.
struct RangeFlag { int *dev = nullptr, *host = nullptr; };
static RangeFlag *range_flag() {
    static RangeFlag flags[64];
    static const bool on = [] { const char *e = getenv("COMMARL_PACK_CHECK"); return !(e && e[0] == '0'); }();
    if (!on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    RangeFlag &f = flags[dev & 63];
    if (!f.dev) {
        if (hipMalloc(&f.dev, sizeof(int)) != hipSuccess) { f.dev = nullptr; return nullptr; }
        if (hipHostMalloc(&f.host, sizeof(int), hipHostMallocDefault) != hipSuccess) { (void)hipFree(f.dev); f.dev = nullptr; return nullptr; }
    }
    return &f;
}
// Opens a checked pack: clears the flag on `stream`; nullptr (no check) while the stream is being captured - a capture cannot
// wait for the device - or with COMMARL_PACK_CHECK=0.
static int *range_check_begin(void *stream) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    RangeFlag *f = range_flag();
    if (!f) return nullptr;
    if (hipMemsetAsync(f->dev, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) return nullptr;
    return f->dev;
}
// Closes it: one small copy + a wait for the pack kernels (a pack happens once per weight update, not per step).
static int range_check_end(int *bad, void *stream, const char *what) {
    if (!bad) return CM_OK;
    RangeFlag *f = range_flag();
    CM_HIP(hipMemcpyAsync(f->host, f->dev, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CM_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (*f->host)
        return set_error(CM_ERR_ARG, std::string(what) + ": a weight lies outside the f16 range (|w| > 65504, inf or NaN) - the f16-split "
                                     "matrix-core kernels cannot carry it; COMMARL_POLICY_KERNEL=f32 selects the all-f32 kernels");
    return CM_OK;
}

static int pack_flush_h(PackJobsH &t, void *stream, int *bad) {
    if (t.count == 0) return CM_OK;
    hipLaunchKernelGGL(pack_layer_h_kernel, dim3(t.first_block[t.count]), dim3(256), 0, (hipStream_t)stream, t, bad);
    CM_HIP(hipGetLastError());
    t.count = 0;
    return CM_OK;
}
// queues one layer (launched by pack_flush_h; a full table is flushed on the way)
static int pack_one_h(PackJobsH &t, const float *Wt, int K, int OUT, int kp, int out_pad, uint4 *dst, void *stream, int *bad) {
    if (!Wt) return set_error(CM_ERR_ARG, "weight pack: null layer weight");
    if (t.count == PACK_MAX_LAYERS)
        if (int rc = pack_flush_h(t, stream, bad)) return rc;
    const int KB = kp / 32, CT = out_pad / 16, total = CT * KB * 64, j = t.count;
    if (j == 0) t.first_block[0] = 0;
    t.W[j] = Wt; t.dst[j] = dst; t.K[j] = K; t.OUT[j] = OUT; t.KB[j] = KB; t.CT[j] = CT;
    t.first_block[j + 1] = t.first_block[j] + (total + 255) / 256;
    t.count = j + 1;
    return CM_OK;
}

static int pack_trunk_h(PackJobsH &t, int d, int L, const float *w1t, const float *w2t, const float *wat, const float *gw, int kh,
                        const PackLayoutH &lo, uint4 *pack, void *stream, int *bad) {
    if (int rc = pack_one_h(t, w1t, d, EH, kh, EH, pack + lo.enc1, stream, bad)) return rc;
    if (int rc = pack_one_h(t, w2t, EH, EMB, EH, EMB, pack + lo.enc2, stream, bad)) return rc;
    if (int rc = pack_one_h(t, wat, EMB, EMB, EMB, EMB, pack + lo.attn, stream, bad)) return rc;
    const size_t per = LayerH<EMB, EMB>::PACK_U4;
    for (int l = 0; l < L; ++l)
        if (int rc = pack_one_h(t, gw ? gw + (size_t)l * EMB * EMB : nullptr, EMB, EMB, EMB, EMB, pack + lo.gcn + (size_t)l * per, stream, bad)) return rc;
    return CM_OK;
}

}  // namespace mh

// COMMARL_POLICY_KERNEL=f32: the round-1 all-f32 MFMA kernel; anything else (default): the f16-split kernel
bool policy_h_enabled() {
    static const bool v = [] { const char *e = getenv("COMMARL_POLICY_KERNEL"); return !(e && e[0] == 'f'); }();
    return v;
}

int policy_pack_w(const cm_policy_weights *w, void *dst, void *stream, int *bad);   // cm_policy_w.hip

size_t policy_pack_h_bytes(int d, int L, bool policy) {
    const int kh = mh::kh_of(d);
    return kh ? mh::pack_layout_h(kh, L, policy).total * sizeof(uint4) : 0;
}

int policy_pack_h(const cm_policy_weights *w, void *dst, int sections, void *stream) {
    const int kh = mh::kh_of(w->d);
    if (!kh) return CM_OK;                               // no f16 instantiation for this obs dim: nothing to pack
    const mh::PackLayoutH lo = mh::pack_layout_h(kh, w->n_hops, true);
    uint4 *pack = reinterpret_cast<uint4 *>(dst);
    int *bad = (sections & CM_PACK_CHECK) ? mh::range_check_begin(stream) : nullptr;
    if (sections & CM_PACK_F16) {
        mh::PackJobsH t{};
        if (int rc = mh::pack_trunk_h(t, w->d, w->n_hops, w->enc_w1t, w->enc_w2t, w->attn_wt, w->gcn_w, kh, lo, pack, stream, bad)) return rc;
        if (int rc = mh::pack_one_h(t, w->hd_w1t, mf::EMB, mf::H1, mf::EMB, mf::H1, pack + lo.x1, stream, bad)) return rc;
        if (int rc = mh::pack_one_h(t, w->hd_w2t, mf::H1, mf::H2, mf::H1, mf::H2, pack + lo.h2, stream, bad)) return rc;
        if (int rc = mh::pack_one_h(t, w->hd_w3t, mf::H2, mf::H3, mf::H2, mf::H3, pack + lo.h3, stream, bad)) return rc;
        if (int rc = mh::pack_one_h(t, w->hd_w4t, mf::H3, w->n_act, mf::H3, 16, pack + lo.h4, stream, bad)) return rc;
        if (int rc = mh::pack_flush_h(t, stream, bad)) return rc;
    }
    // teams of 4: the wave-owned kernel's fragments (another k order, cm_policy_w.hip), behind this section
    if (sections & CM_PACK_WAVE)
        if (int rc = policy_pack_w(w, reinterpret_cast<char *>(dst) + lo.total * sizeof(uint4), stream, bad)) return rc;
    return mh::range_check_end(bad, stream, "cm_policy_pack");
}

int critic_pack_w(const cm_critic_weights *w, void *dst, void *stream, int *bad);   // cm_policy_w.hip

int critic_pack_h(const cm_critic_weights *w, void *dst, int sections, void *stream) {
    const int kh = mh::kh_of(w->d);
    if (!kh || !(sections & (CM_PACK_F16 | CM_PACK_WAVE))) return CM_OK;
    const mh::PackLayoutH lo = mh::pack_layout_h(kh, w->n_hops, false);
    uint4 *pack = reinterpret_cast<uint4 *>(dst);
    int *bad = (sections & CM_PACK_CHECK) ? mh::range_check_begin(stream) : nullptr;
    if (sections & CM_PACK_F16) {
        mh::PackJobsH t{};
        if (int rc = mh::pack_trunk_h(t, w->d, w->n_hops, w->enc_w1t, w->enc_w2t, w->attn_wt, w->gcn_w, kh, lo, pack, stream, bad)) return rc;
        if (int rc = mh::pack_one_h(t, w->dec_w1t, mf::EMB, mf::DH, mf::EMB, mf::DH, pack + lo.x1, stream, bad)) return rc;
        if (int rc = mh::pack_flush_h(t, stream, bad)) return rc;
    }
    // teams of 4: the wave-owned training forward's fragments (cm_critic_forward_saved_wave), behind this section
    if (sections & CM_PACK_WAVE)
        if (int rc = critic_pack_w(w, reinterpret_cast<char *>(dst) + lo.total * sizeof(uint4), stream, bad)) return rc;
    return mh::range_check_end(bad, stream, "cm_critic_pack");
}

// h_pack = the f16 operand pack (behind the f32 one in the caller's pack buffer).  Returns 1 when this shape has no
// f16 instantiation (the caller then runs the f32 kernel).
int policy_forward_h(const cm_policy_weights *w, const void *h_pack, mf::FwdArgs a, void *stream) {
    const int kh = mh::kh_of(w->d);
    if (!kh || !h_pack) return 1;
    const mh::PackLayoutH lo = mh::pack_layout_h(kh, w->n_hops, true);
    const uint4 *P = reinterpret_cast<const uint4 *>(h_pack);
    const mh::TrunkH tw{ P + lo.enc1, w->enc_b1, P + lo.enc2, w->enc_b2, P + lo.attn, P + lo.gcn, w->gcn_b };
    const mh::PolHeadH ph{ P + lo.x1, w->hd_b1, P + lo.h2, w->hd_b2, P + lo.h3, w->hd_b3, P + lo.h4, w->hd_b4, w->n_act };
    return mh::dispatch_h<0>(a, tw, ph, mh::CritHeadH{}, stream);
}

int critic_forward_h(const cm_critic_weights *w, const void *h_pack, mf::FwdArgs a, void *stream) {
    const int kh = mh::kh_of(w->d);
    if (!kh || !h_pack) return 1;
    const mh::PackLayoutH lo = mh::pack_layout_h(kh, w->n_hops, false);
    const uint4 *P = reinterpret_cast<const uint4 *>(h_pack);
    const mh::TrunkH tw{ P + lo.enc1, w->enc_b1, P + lo.enc2, w->enc_b2, P + lo.attn, P + lo.gcn, w->gcn_b };
    const mh::CritHeadH chd{ P + lo.x1, w->dec_b1, w->dec_w2t, w->dec_b2 };
    return mh::dispatch_h<1>(a, tw, mh::PolHeadH{}, chd, stream);
}

}  // namespace cm
