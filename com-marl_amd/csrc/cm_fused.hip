// cm_fused.hip - one rollout step in ONE launch: the Comm-DP policy forward + sample of a workgroup's envs
// (cm_policy_mfma_dev.h, reference: comm_categorical_mlp_policy.py:98-119) followed, in the same workgroup, by the
// env step of those envs on the sampled actions (cm_env_dev.h, reference: vec_env_executor.py:19-45 over
// predator_prey.py:494-519 / coverage.py:319-401 + env_communication.py:91-157).
//
// Why: at the headline config the stand-alone env kernel runs one wave per SIMD on a quarter of the machine and is
// pure dependent-instruction latency (~12 us for ~2500 instructions per wave), and the policy kernel leaves issue
// slots free whenever its waves wait on the matrix pipe.  Inside one workgroup the env phase of one workgroup
// overlaps the matrix phases of its CU neighbour, the actions go from the sampler to the env step through LDS, and a
// step costs one launch / one grid drain instead of two.  Results are bit-identical to cm_policy_forward followed by
// cm_env_step (same device bodies, same Philox counters); tests/test_hip_fused_parity.py checks exactly that.
#include <stdlib.h>

#include "cm_env_dev.h"
#include "cm_policy_mfma_dev.h"
#include "cm_policy_h_dev.h"

namespace cm {

// cm_rollout_w.hip: teams of 4 on wave-owned rows (single step, or a persistent chunk); 1 = not available for this handle
int launch_rollout_w(mf::FwdArgs a, const cm_policy_weights *w, const void *w_pack, const cm_env *h, const cm_rng_tape &t, const cm_step_out &out,
                     void *stream, const ChunkArgs *chunk);
bool shape_ok_rollout_w(int N, int d, int L, int n_act);

bool policy_w_enabled();                                 // cm_policy_w.hip: wave-owned teams-of-4 kernel (default where the shape allows)
size_t policy_pack_h_bytes(int d, int L, bool policy);   // cm_policy_h.hip: size of the f16 pack the wave-owned fragments sit behind
bool policy_h_enabled();                                 // cm_policy_h.hip: f16-split dense layers (default) or the all-f32 body

// POL selects the policy body: 0 = cm_policy_mfma_dev.h (all f32), 1 = cm_policy_h_dev.h (f16-split dense layers).  Both
// forms take the workgroup's dynamic LDS block; `act` = the sampled actions behind the policy tiles.
template <int POL, int KPAD, int MAXMK, bool FULL = false>
__device__ __forceinline__ void policy_body(const mf::FwdArgs &a, const mf::TrunkW &tw, const mf::PolHead &ph, const mh::TrunkH &twh,
                                            const mh::PolHeadH &phh, float *lds, int32_t *act) {
    if constexpr (POL == 0) mf::fwd_body<0, KPAD, MAXMK>(a, tw, ph, mf::CritHead{}, lds, blockIdx.x, act);
    else mh::fwd_body_h<0, KPAD, MAXMK, 4, false, false, FULL>(a, twh, phh, mh::CritHeadH{}, reinterpret_cast<unsigned char *>(lds), blockIdx.x, act);
}

// FULL: the constant-shape build for teams of 4 with every workgroup full (8 envs, S % 8 == 0) - team size, rows and
// the LDS map are compile-time constants in the policy body (cm_policy_h_dev.h) and the env count here.
template <int SCEN, int LPE, int KPAD, int MAXMK, int POL, bool FULL = false, bool PRE = false>
__global__ __launch_bounds__(mf::TPB) void rollout_step_kernel(mf::FwdArgs a, mf::TrunkW tw, mf::PolHead ph, mh::TrunkH twh,
                                                               mh::PolHeadH phh, EnvDev p, cm_rng_tape tape, cm_step_out out,
                                                               int act_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int32_t *act = reinterpret_cast<int32_t *>(lds + act_off);           // [EPB*N] sampled actions, behind the policy tiles
    const int tx = thread_x(), grp = tx / LPE;
    const int EPBc = FULL ? 8 : a.EPB;
    const int envs = FULL ? 8 : min(a.EPB, a.S - (int)blockIdx.x * a.EPB);
    constexpr bool ALL = FULL && LPE == 32;                              // 8 envs x 32 lanes: every group of every wave has its env
    const bool live = ALL || grp < envs;
    const bool env_wave = ALL || (tx & ~63) / LPE < envs;               // a wave with an env of its own (the env body syncs wave-locally)
    const int b_raw = blockIdx.x * EPBc + (live ? grp : 0);
    // PRE (host-checked: env_prefetch_ok): the env state does not depend on the actions - requested in front of the
    // policy forward, consumed behind it
    EnvPre pre{};
    if constexpr (PRE) pre = env_prefetch<SCEN, LPE>(p, b_raw, live);
    policy_body<POL, KPAD, MAXMK, FULL>(a, tw, ph, twh, phh, lds, act);
    __syncthreads();                                                     // actions visible; the policy tiles are dead
    if (!env_wave) return;
    const int32_t *my_act = act + (live ? grp : 0) * p.N;
    if constexpr (PRE) {
        const bool bad = env_stage<SCEN, LPE>(p, pre, my_act, grp, 0);
        env_body<SCEN, LPE>(p, nullptr, my_act, tape, out, 0, grp, b_raw, live, 0, nullptr, true, pre.rng_step, pre.step_count_in, pre.succ,
                            pre.t_row, pre.t_col, pre.t_step0, pre.t_step, pre.t_rew, bad, ALL);
    } else env_body<SCEN, LPE>(p, nullptr, my_act, tape, out, 0, grp, b_raw, live, 0);
}

// Persistent form: the workgroup keeps its envs for n_steps consecutive steps (policy -> env -> policy ...), pointers
// advancing by the per-step strides of the time-major trajectory buffers.  No grid-wide synchronisation between steps
// and one launch per chunk: a step costs the workgroup's own policy + env chain (28.6 us at the headline config), without
// the grid drain of a launch per step.  Step t+1 reads what step t wrote (observation, masks, env state) through the
// CU's own write-through L1 / L2: a workgroup-scope release / acquire pair around the workgroup barrier orders them.

// Two waves per SIMD as the single-step kernel: without the bound the compiler hoists every layer's (step-invariant) weight
// fragment loads out of the step loop, ends at 506 VGPRs and one workgroup per CU - half the grid waits for the other
// half (62 us per step measured, against 28 for the single-step kernel).
template <int SCEN, int LPE, int KPAD, int MAXMK, int POL>
__global__ __launch_bounds__(mf::TPB, 2) void rollout_chunk_kernel(mf::FwdArgs a, mf::TrunkW tw, mf::PolHead ph, mh::TrunkH twh,
                                                                mh::PolHeadH phh, EnvDev p, cm_step_out out, ChunkArgs c,
                                                                int act_off) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int32_t *act = reinterpret_cast<int32_t *>(lds + act_off);
    const int envs = min(a.EPB, a.S - (int)blockIdx.x * a.EPB);
    const cm_rng_tape no_tape{};
    // De-phase the workgroups that share a CU: the grid is dispatched round-robin, so workgroup b and b + gridDim / 2 are
    // neighbours; the late one's matrix phases then meet the early one's env phases for the whole chunk.
    if (c.stagger > 0 && blockIdx.x >= gridDim.x / 2)
        for (int i = 0; i < c.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    for (int t = 0; t < c.n_steps; ++t) {
        asm volatile("" ::: "memory");                                   // keep each step's loads inside the step
        const int grp = thread_x() / LPE;
        const bool live = grp < envs;
        mf::FwdArgs at = a;
        at.obs = a.obs + t * c.obs;
        at.adj = a.adj ? a.adj + t * c.dist_adj : nullptr;
        at.chan = a.chan ? a.chan + t * c.channels : nullptr;
        at.policy_step = a.policy_step + (uint32_t)t;
        at.actions = a.actions ? a.actions + t * c.actions : nullptr;
        at.probs = a.probs ? a.probs + t * c.probs : nullptr;
        at.attn = a.attn ? a.attn + t * c.attn : nullptr;
        policy_body<POL, KPAD, MAXMK>(at, tw, ph, twh, phh, lds, act);
        __syncthreads();                                                 // actions visible; the policy tiles are dead
        cm_step_out ot = out;
        if (ot.obs) ot.obs += t * c.obs;
        if (ot.reward) ot.reward += t * c.reward;
        if (ot.reward_f64) ot.reward_f64 += t * c.reward_f64;
        if (ot.done) ot.done += t * c.done;
        if (ot.details) ot.details += t * c.details;
        if (ot.dist_adj) ot.dist_adj += t * c.dist_adj;
        if (ot.channels) ot.channels += t * c.channels;
        if (ot.prey_alive) ot.prey_alive += t * c.prey_alive;
        if (ot.success) ot.success += t * c.success;
        if (ot.path_len) ot.path_len += t * c.path_len;
        if ((thread_x() & ~63) / LPE < envs)                             // waves without an env of their own skip to the barrier
            env_body<SCEN, LPE>(p, nullptr, act + (live ? grp : 0) * p.N, no_tape, ot, 0, grp, blockIdx.x * a.EPB + (live ? grp : 0), live, 0);
        // workgroup-scope release / acquire around the barrier: every wave's stores of this step are performed
        // (vmcnt drained) before any wave issues the next step's loads.  All waves of a workgroup share their CU's
        // write-through vector L1, so no cache maintenance is needed (an agent-scope pair would write back and
        // invalidate the XCD's L2 every step - measured 2.3x slower than stepping with one launch each).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// diagnostic (COMMARL_ENV_STOP=-1): phase clocks of workgroup 0's env phase inside the fused step (ENV_PROBE, cm_env_dev.h)
static void probe_dump(const EnvDev &d, void *stream) {
    if (d.stop >= 0) return;
    unsigned long long h_probe[16];
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return;
    if (hipMemcpyFromSymbol(h_probe, HIP_SYMBOL(g_env_probe), sizeof(h_probe)) != hipSuccess) return;
    fprintf(stderr, "[fused env probe] clk since env entry:");
    for (int i = 1; i < 10; ++i) fprintf(stderr, " p%d=%lld", i, (long long)(h_probe[i] - h_probe[0]));
    fprintf(stderr, "\n");
}

template <int SCEN, int LPE, int KPAD, int MAXMK, int POL>
static int launch_fused(mf::FwdArgs a, const mf::TrunkW &tw, const mf::PolHead &ph, const mh::TrunkH &twh, const mh::PolHeadH &phh,
                        const cm_env *h, const cm_rng_tape &t, const cm_step_out &out, void *stream, const ChunkArgs *chunk = nullptr) {
    const EnvDev &d = h->dev;
    a.EPB = mf::pick_epb(a.N);
    constexpr int GROUPS = mf::TPB / LPE;
    if (a.EPB > GROUPS) return 1;                                        // more envs per workgroup than env groups
    const int rows_cap = (a.EPB * a.N + 15) & ~15;
    const size_t pol_bytes = POL == 0 ? mf::lds_floats(rows_cap, a.EPB, a.N) * sizeof(float)
                                      : mh::lds_map(rows_cap, a.EPB, a.N, MAXMK < 0 ? -1 : (MAXMK > 0 ? 1 : 0)).total;
    const size_t pol_floats = (pol_bytes + 3) / 4;
    const size_t env_bytes = (size_t)d.lds_env * GROUPS;
    if (env_bytes > pol_floats * sizeof(float)) return 1;                // env area would reach into the action array
    const size_t lds = (pol_floats + (size_t)a.EPB * a.N) * sizeof(float);
    if (lds > 160 * 1024) return 1;
    static unsigned long long attr_set = 0;
    if (cm::dev_first(attr_set)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_step_kernel<SCEN, LPE, KPAD, MAXMK, POL>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int blocks = (a.S + a.EPB - 1) / a.EPB;
    static const int pre_flag = [] { const char *e = getenv("COMMARL_ENV_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }();
    if (chunk) {
        static unsigned long long attr_set_c = 0;
        if (cm::dev_first(attr_set_c)) {
            CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_chunk_kernel<SCEN, LPE, KPAD, MAXMK, POL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipLaunchKernelGGL((rollout_chunk_kernel<SCEN, LPE, KPAD, MAXMK, POL>), dim3(blocks), dim3(mf::TPB), lds, (hipStream_t)stream, a,
                           tw, ph, twh, phh, d, out, *chunk, (int)pol_floats);
        CM_HIP(hipGetLastError());
        return CM_OK;
    }
    if constexpr (MAXMK < 0 && POL == 1) {
        static const bool full_on = [] { const char *e = getenv("COMMARL_FWD_FULL"); return !(e && e[0] == '0'); }();
        if (full_on && a.EPB == 8 && a.S % 8 == 0) {
#define CM_FULL_LAUNCH(PRE)                                                                                                       \
    do {                                                                                                                          \
        static unsigned long long attr_set_f = 0;                                                                                           \
        if (cm::dev_first(attr_set_f)) {                                                                                                        \
            CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_step_kernel<SCEN, LPE, KPAD, MAXMK, POL, true, PRE>), \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                  \
        }                                                                                                                         \
        hipLaunchKernelGGL((rollout_step_kernel<SCEN, LPE, KPAD, MAXMK, POL, true, PRE>), dim3(blocks), dim3(mf::TPB), lds,       \
                           (hipStream_t)stream, a, tw, ph, twh, phh, d, t, out, (int)pol_floats);                                 \
        CM_HIP(hipGetLastError());                                                                                                \
        probe_dump(d, stream);                                                                                                    \
        return CM_OK;                                                                                                             \
    } while (0)
            if (pre_flag && env_prefetch_ok<SCEN, LPE>(d)) CM_FULL_LAUNCH(true);
            CM_FULL_LAUNCH(false);
#undef CM_FULL_LAUNCH
        }
    }
    hipLaunchKernelGGL((rollout_step_kernel<SCEN, LPE, KPAD, MAXMK, POL>), dim3(blocks), dim3(mf::TPB), lds, (hipStream_t)stream, a, tw,
                       ph, twh, phh, d, t, out, (int)pol_floats);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

}  // namespace cm

using namespace cm;

// COMMARL_FUSED=0 disables the fused step (A/B timing): the entry point then reports "not available"
static bool fused_enabled() {
    static const bool v = [] { const char *e = getenv("COMMARL_FUSED"); return !(e && e[0] == '0'); }();
    return v;
}

static int rollout_impl(cm_env_t h, const cm_policy_weights *w, const float *obs, const float *avail,
                        const float *dist_adj, const float *channels, uint64_t seed, int32_t env_id_offset,
                        uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                        float *probs, float *attn, const cm_rng_tape *tape, const cm_step_out *out, void *stream,
                        const ChunkArgs *chunk) {
    if (!h || !w || !obs || !out) return set_error(CM_ERR_ARG, "cm_rollout_step: null argument");
    const EnvDev &d = h->dev;
    if (w->n_agents != d.N || w->d != d.d || w->n_hops != d.L)
        return set_error(CM_ERR_ARG, "cm_rollout_step: policy shape (n_agents, d, n_hops) does not match the env handle");
    if (!fused_enabled() || !w->mfma_pack || !policy_shape_ok(w)) return 1;
    if (int rc = check_tape(h, tape, false)) return rc;
    cm_rng_tape t{};
    if (tape) t = *tape;
    mf::FwdArgs a{};
    a.S = d.B; a.N = d.N; a.d = d.d; a.L = d.L;
    a.obs = obs; a.avail = avail; a.adj = dist_adj; a.chan = channels;
    a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.policy_step = policy_step; a.step_base = policy_step_base;
    a.env_id_offset = env_id_offset; a.greedy = greedy; a.no_residual = w->no_residual;
    a.actions = actions; a.probs = probs; a.attn = attn;
    const int kpad = mf::kpad_of(d.d);
    const mf::PackLayout lo = mf::pack_layout(kpad, d.L, true);
    const float *P = w->mfma_pack;
    const mf::TrunkW tw{ P + lo.enc1, w->enc_b1, P + lo.enc2, w->enc_b2, P + lo.attn, P + lo.gcn, w->gcn_b };
    const mf::PolHead ph{ P + lo.x1, w->hd_b1, P + lo.h2, w->hd_b2, P + lo.h3, w->hd_b3, P + lo.h4, w->hd_b4, w->n_act };
    static const int mk_min = [] { const char *e = getenv("COMMARL_MK_MIN"); return e ? atoi(e) : 16; }();   // N x N products on MFMA tiles from 16 agents up
    const int mk = d.N < mk_min ? 0 : (d.N <= 80 ? 25 : 64);                 // as mf::dispatch
    // instantiations: the four BASELINE shapes (PP sen1 small teams; CO sen2 mid teams; PP / CO sen2 large teams)
    const bool quad = d.N == 4 && mf::pick_epb(4) * 4 <= 32;
    const int kh = mh::kh_of(d.d);
    if (policy_w_enabled() && shape_ok_rollout_w(d.N, d.d, d.L, w->n_act)) {    // teams of 4: wave-owned rows, single step or persistent chunk
        const int rc = launch_rollout_w(a, w, reinterpret_cast<const char *>(P + lo.total) + policy_pack_h_bytes(d.d, d.L, true), h, t, *out, stream, chunk);
        if (rc != 1) return rc;
    }
    if (policy_h_enabled() && kh) {                  // f16-split dense layers: the operand pack behind the f32 one
        const mh::PackLayoutH lh = mh::pack_layout_h(kh, d.L, true);
        const uint4 *Q = reinterpret_cast<const uint4 *>(P + lo.total);
        const mh::TrunkH twh{ Q + lh.enc1, w->enc_b1, Q + lh.enc2, w->enc_b2, Q + lh.attn, Q + lh.gcn, w->gcn_b };
        const mh::PolHeadH phh{ Q + lh.x1, w->hd_b1, Q + lh.h2, w->hd_b2, Q + lh.h3, w->hd_b3, Q + lh.h4, w->hd_b4, w->n_act };
        // teams of 4: 8 envs per workgroup.  With 32 lanes per env the env phase occupies all four waves (16 lanes leave two of
        // them idle): its lane-parallel loops (observation emission, tile rebuild) halve - 28.5 -> 27.4 us per step at the
        // headline config.  The stand-alone env kernel keeps the handle's own choice (16: more envs per wave); the env body
        // is bit-identical at every width (tests/test_hip_scale_parity.py).  COMMARL_FUSED_LPE=16 for the A/B.
        static const int fused_lpe = [] { const char *e = getenv("COMMARL_FUSED_LPE"); return e ? atoi(e) : 32; }();
        if (d.scen == CM_PP && d.lpe <= 32 && kh == 32 && quad && fused_lpe == 32 && !chunk)
            return launch_fused<CM_PP, 32, 32, -1, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        if (d.scen == CM_PP && d.lpe == 16 && kh == 32 && quad) return launch_fused<CM_PP, 16, 32, -1, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        if (d.scen == CM_PP && d.lpe == 16 && kh == 32 && mk == 0) return launch_fused<CM_PP, 16, 32, 0, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        if (d.scen == CM_CO && d.lpe == 64 && kh == 96 && mk == 0) return launch_fused<CM_CO, 64, 96, 0, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        if (d.scen == CM_PP && d.lpe == 64 && kh == 64 && mk == 25) return launch_fused<CM_PP, 64, 64, 25, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        if (d.scen == CM_CO && d.lpe == 64 && kh == 96 && mk == 25) return launch_fused<CM_CO, 64, 96, 25, 1>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
        return 1;
    }
    const mh::TrunkH twh{};
    const mh::PolHeadH phh{};
    if (d.scen == CM_PP && d.lpe == 16 && kpad == 32 && quad) return launch_fused<CM_PP, 16, 32, -1, 0>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
    if (d.scen == CM_PP && d.lpe == 16 && kpad == 32 && mk == 0) return launch_fused<CM_PP, 16, 32, 0, 0>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
    if (d.scen == CM_CO && d.lpe == 64 && kpad == 80 && mk == 0) return launch_fused<CM_CO, 64, 80, 0, 0>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
    if (d.scen == CM_PP && d.lpe == 64 && kpad == 64 && mk == 25) return launch_fused<CM_PP, 64, 64, 25, 0>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
    if (d.scen == CM_CO && d.lpe == 64 && kpad == 80 && mk == 25) return launch_fused<CM_CO, 64, 80, 25, 0>(a, tw, ph, twh, phh, h, t, *out, stream, chunk);
    return 1;
}

extern "C" int cm_rollout_step(cm_env_t h, const cm_policy_weights *w, const float *obs, const float *avail,
                               const float *dist_adj, const float *channels, uint64_t seed, int32_t env_id_offset,
                               uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                               float *probs, float *attn, const cm_rng_tape *tape, const cm_step_out *out, void *stream) {
    return rollout_impl(h, w, obs, avail, dist_adj, channels, seed, env_id_offset, policy_step, policy_step_base, greedy, actions,
                        probs, attn, tape, out, stream, nullptr);
}

extern "C" int cm_rollout_chunk(cm_env_t h, const cm_policy_weights *w, int32_t n_steps, const cm_chunk_strides *st,
                                const float *obs, const float *dist_adj, const float *channels, uint64_t seed,
                                int32_t env_id_offset, uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy,
                                int32_t *actions, float *probs, float *attn, const cm_step_out *out, void *stream) {
    if (!st) return set_error(CM_ERR_ARG, "cm_rollout_chunk: null strides");
    if (n_steps < 0) return set_error(CM_ERR_ARG, "cm_rollout_chunk: negative step count");
    if (h && h->cfg.rng_mode == CM_RNG_TAPE) return set_error(CM_ERR_ARG, "cm_rollout_chunk: tape mode steps one launch at a time");
    if (n_steps == 0) return CM_OK;
    static const int stagger = [] { const char *e = getenv("COMMARL_CHUNK_STAGGER"); return e ? atoi(e) : 0; }();
    ChunkArgs c{ n_steps, stagger, st->obs, st->actions, st->probs, st->attn, st->reward, st->reward_f64, st->done, st->details,
                 st->dist_adj, st->channels, st->prey_alive, st->success, st->path_len };
    return rollout_impl(h, w, obs, nullptr, dist_adj, channels, seed, env_id_offset, policy_step, policy_step_base, greedy, actions,
                        probs, attn, nullptr, out, stream, &c);
}

extern "C" int cm_rollout_chunk_tail(cm_env_t h, const cm_policy_weights *w, int32_t n_steps, const cm_chunk_strides *st,
                                     const float *obs, const float *dist_adj, const float *channels, uint64_t seed,
                                     int32_t env_id_offset, uint32_t policy_step, uint32_t *policy_step_base, int32_t greedy,
                                     int32_t *actions, float *probs, float *attn, const cm_step_out *out, float *obs_next,
                                     float *dist_adj_next, float *channels_next, void *stream) {
    if (!st) return set_error(CM_ERR_ARG, "cm_rollout_chunk_tail: null strides");
    if (n_steps < 1) return set_error(CM_ERR_ARG, "cm_rollout_chunk_tail: at least one step");
    if (!h || !out || !out->obs || !obs_next || !policy_step_base) return set_error(CM_ERR_ARG, "cm_rollout_chunk_tail: null argument");
    if (h->cfg.rng_mode == CM_RNG_TAPE) return set_error(CM_ERR_ARG, "cm_rollout_chunk_tail: tape mode steps one launch at a time");
    static const int stagger = [] { const char *e = getenv("COMMARL_CHUNK_STAGGER"); return e ? atoi(e) : 0; }();
    static const bool fold = [] { const char *e = getenv("COMMARL_FOLD_TAIL"); return !(e && e[0] == '0'); }();
    int folded = 0;
    ChunkArgs c{ n_steps, stagger, st->obs, st->actions, st->probs, st->attn, st->reward, st->reward_f64, st->done, st->details,
                 st->dist_adj, st->channels, st->prey_alive, st->success, st->path_len };
    if (fold) { c.tail_obs = obs_next; c.tail_base = policy_step_base; c.tail_folded = &folded; }
    const int rc = rollout_impl(h, w, obs, nullptr, dist_adj, channels, seed, env_id_offset, policy_step, policy_step_base, greedy, actions,
                                probs, attn, nullptr, out, stream, &c);
    if (rc != CM_OK || folded) return rc;
    const EnvDev &d = h->dev;
    const size_t last = (size_t)(n_steps - 1);
    const bool adj = out->dist_adj && dist_adj_next, ch = out->channels && channels_next;
    return cm_chunk_tail(policy_step_base, (uint32_t)n_steps, out->obs + last * st->obs, obs_next, (size_t)d.B * d.N * d.d * sizeof(float),
                         adj ? out->dist_adj + last * st->dist_adj : nullptr, adj ? dist_adj_next : nullptr,
                         adj ? (size_t)d.B * d.N * d.N * sizeof(float) : 0, ch ? out->channels + last * st->channels : nullptr,
                         ch ? channels_next : nullptr, ch ? (size_t)d.B * d.L * d.N * d.N * sizeof(float) : 0, stream);
}
