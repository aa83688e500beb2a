// cm_rollout_w.hip - the rollout step of teams of 4 on WAVE-OWNED rows: Comm-DP policy forward + sample (cm_policy_w_dev.h) and
// the env step (cm_env_dev.h) of a wave's four envs, by that wave alone, for one step or a whole chunk of steps per launch
// (reference: centralized_ma_on_policy_vectorized_sampler.py:119-232 - get_actions, vec_env.step, obses = next_obses).
// Its own translation unit: built with -fno-slp-vectorize (Makefile), which the older kernels of cm_fused.hip are not.
#include <stdio.h>
#include <stdlib.h>

#include "cm_env_dev.h"
#include "cm_policy_w_dev.h"

namespace cm {

bool policy_w_enabled();                                 // cm_policy_w.hip

// diagnostic (COMMARL_ENV_STOP=-2): shader clocks of workgroup 0 / thread 0, summed over the launch's steps: [0] steps, [1] policy
// tile, [2] env phase, [3] weight staging
static __device__ unsigned long long g_w_probe[5];

// ---- teams of 4, wave-owned rows (cm_policy_w_dev.h): a workgroup = 16 envs = four waves, ONE per SIMD; a wave carries its four
// envs through policy forward, sample AND env step by itself - the actions go through LDS words only that wave touches, the env
// phase's 16-lane groups are the wave's own envs - so no workgroup barrier exists after the one behind the weight staging, and
// with n_steps > 1 the wave simply loops (weights stay where they are: LDS image + the register-resident 128 -> 64 layer).
// LDS: [policy image | 64 actions | 16 env areas].
// Scalar registers are the scarce resource of this kernel (every kernel argument lives in SGPRs for the whole step loop; what does not
// fit is spilled to VGPR lanes and comes back through v_readlane): the per-step strides travel as 32-bit element counts and the
// RNG tape - test-only, single-step launches - is a compile-time variant.
// folded chunk tail (CARRY builds): after its last step a wave writes its envs' next observation into slot 0 and takes a ticket;
// the wave that takes the last one advances the sampler's Philox base - every other wave has read it for the last time
struct TailW { float *obs_dst; uint32_t *base; unsigned int *ticket; int on; };
struct StridesW { int n_steps, obs, actions, probs, attn, reward, reward_f64, done, details, dist_adj, channels, prey_alive, success, path_len; };

// CARRY (PRE builds, constant adjacency, no channel model, no tape): a wave's envs hand observation and state from step to step
// through LDS and registers (EnvCarry, env_pre_carry, the observation copy in the env area's unused claim table); no load of a
// step depends on a store of the launch, so the fence between two steps goes and the trajectory stores of step t drain under
// the policy forward of step t + 1.
// SHAPE 1 (carried builds): the grid of BASELINE config 2 - 10 x 10 cells, 4 preys, sensing range 1 (3 x 3 window, 21 observation
// entries) - as compile-time constants: index divisions by constants, constant LDS offsets, unrolled element loops.
template <int LHOPS, bool PRE, bool FULLWG, bool TAPE, bool CARRY = false, int SHAPE = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void rollout_w_kernel(mf::FwdArgs a, mw::WeightsW w, EnvDev p, cm_rng_tape tape_arg, cm_step_out out, StridesW c, TailW tl) {
    static_assert(SHAPE == 0 || CARRY, "shape constants are built into the carried form only");
    if constexpr (SHAPE == 1) {
        p.S = 10; p.M = 4; p.R = 1; p.W = 3; p.d = 21; p.rcp_d = 1.0f / 21.0f; p.rcp_W = 1.0f / 3.0f; p.rcp_WW = 1.0f / 9.0f;
        p.lds_env = lds_env_bytes(10, 4, 4);
        a.N = 4; a.d = 21; a.L = LHOPS;
    }
    static_assert(!CARRY || (PRE && !TAPE), "the carried form is the prefetching, tape-less build");
    const cm_rng_tape tape = TAPE ? tape_arg : cm_rng_tape{};
    // what the launcher has already established, as compile-time constants of the by-value config: the branches on them fold away
    if constexpr (!TAPE) p.rng_mode = CM_RNG_PHILOX;                     // no tape pointers
    p.scen = CM_PP; p.N = 4; p.lpe = 16; p.rcp_N = 0.25f; p.rcp_NN = 0.0625f;
    if constexpr (CARRY) { p.adj_const = 1; p.ch_const = 1; p.channel = CM_CH_FC; }   // constant adjacency, no channel model
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_w[];
    constexpr int LPE = 16;
    constexpr int ACT_OFF = mw::pack_w(LHOPS).lds_u4 * 16, ENV_BASE = ACT_OFF + mw::WG_ROWS * 4;
    int32_t *act = reinterpret_cast<int32_t *>(lds_w + ACT_OFF);
    const bool probe = p.stop == -2 && blockIdx.x == 0 && thread_x() == 0;
    const unsigned long long t_in = probe ? __builtin_amdgcn_s_memtime() : 0ull;
    mw::stage_w<LHOPS>(w, lds_w, thread_x());
    mw::ResidentW res;
    res.fetch<LHOPS>(w, thread_x() & 63);
    __syncthreads();                                                     // the only workgroup barrier of the launch
    if (probe) { g_w_probe[3] = __builtin_amdgcn_s_memtime() - t_in; g_w_probe[0] = g_w_probe[1] = g_w_probe[2] = g_w_probe[4] = 0; }
    const int envs = FULLWG ? mw::WG_ENVS : min(mw::WG_ENVS, a.S - (int)blockIdx.x * mw::WG_ENVS);
    EnvPre pre{};
    int obs_row = 0, obs_env = -1;                                       // LDS copy: this lane's row (policy) / this group's env (emission)
    if constexpr (CARRY) {
        const int tx = thread_x(), grp = tx / LPE, lane = tx & 63, cc = lane & 15;
        const bool live = FULLWG || grp < envs;
        const Lds l0 = make_lds(p.S, p.N, p.M, ENV_BASE + p.lds_env * grp, p.status);
        obs_env = l0.win;                                                // teams of 4 never run agents_parallel: its claim table is free
        const int env_l = (tx >> 6) * 4 + (cc >> 2);                     // env of the policy's row c
        obs_row = make_lds(p.S, p.N, p.M, ENV_BASE + p.lds_env * env_l, p.status).win + (cc & 3) * OBS_COPY_STRIDE * 4;
        // step 0: the observation of slot 0 into the copy (rows of this group's env; zeros behind the d entries), the env
        // state from the global arrays - the only loads of the launch that read what an earlier launch wrote
        float *oc = reinterpret_cast<float *>(lds_w + obs_env);
        const int b0 = blockIdx.x * mw::WG_ENVS + (live ? grp : 0);
        const bool ok = live && b0 < a.S;
        for (int k = tx % LPE; k < 4 * OBS_COPY_STRIDE; k += LPE) {
            const int i = k / OBS_COPY_STRIDE, f = k - i * OBS_COPY_STRIDE;
            oc[k] = (ok && f < a.d) ? a.obs[((size_t)b0 * 4 + i) * a.d + f] : 0.0f;
        }
        pre = env_prefetch<CM_PP, LPE>(p, b0, live);
    }
    for (int t = 0; t < c.n_steps; ++t) {
        asm volatile("" ::: "memory");                                   // keep each step's loads inside the step
        const int tx = thread_x(), grp = tx / LPE;
        const bool live = FULLWG || grp < envs;
        const bool env_wave = FULLWG || (tx & ~63) / LPE < envs;        // a wave with an env of its own
        const int b_raw = blockIdx.x * mw::WG_ENVS + (live ? grp : 0);
        mf::FwdArgs at = a;
        at.obs = a.obs + (size_t)t * c.obs;
        at.adj = a.adj ? a.adj + (size_t)t * c.dist_adj : nullptr;
        at.chan = a.chan ? a.chan + (size_t)t * c.channels : nullptr;
        at.policy_step = a.policy_step + (uint32_t)t;
        at.actions = a.actions ? a.actions + (size_t)t * c.actions : nullptr;
        at.probs = a.probs ? a.probs + (size_t)t * c.probs : nullptr;
        at.attn = a.attn ? a.attn + (size_t)t * c.attn : nullptr;
        const unsigned long long t0 = probe ? __builtin_amdgcn_s_memtime() : 0ull;
        if constexpr (PRE && !CARRY) pre = env_prefetch<CM_PP, LPE>(p, b_raw, live);   // env state requested in front of the policy forward
        mw::policy_tile_w<LHOPS, CARRY>(at, w.n_act, res, lds_w, blockIdx.x, act, obs_row);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's action words are in LDS
        const unsigned long long t1 = probe ? __builtin_amdgcn_s_memtime() : 0ull;
        cm_step_out ot = out;
        if (ot.obs) ot.obs += (size_t)t * c.obs;
        if (ot.reward) ot.reward += (size_t)t * c.reward;
        if (ot.reward_f64) ot.reward_f64 += (size_t)t * c.reward_f64;
        if (ot.done) ot.done += (size_t)t * c.done;
        if (ot.details) ot.details += (size_t)t * c.details;
        if (ot.dist_adj) ot.dist_adj += (size_t)t * c.dist_adj;
        if (ot.channels) ot.channels += (size_t)t * c.channels;
        if (ot.prey_alive) ot.prey_alive += (size_t)t * c.prey_alive;
        if (ot.success) ot.success += (size_t)t * c.success;
        if (ot.path_len) ot.path_len += (size_t)t * c.path_len;
        if (env_wave) {
            const int32_t *my_act = act + (live ? grp : 0) * 4;
            if constexpr (CARRY) {
                const bool bad = env_stage<CM_PP, LPE>(p, pre, my_act, grp, ENV_BASE);
                EnvCarry carry{ pre.step_count_in, pre.succ, 0 };
                env_body<CM_PP, LPE>(p, nullptr, my_act, tape, ot, 0, grp, b_raw, live, ENV_BASE, nullptr, true, pre.rng_step, pre.step_count_in,
                                     pre.succ, pre.t_row, pre.t_col, pre.t_step0, pre.t_step, pre.t_rew, bad, FULLWG, &carry, obs_env);
                pre = env_pre_carry<CM_PP, LPE>(p, pre, carry, grp, ENV_BASE);
            } else if constexpr (PRE) {
                const bool bad = env_stage<CM_PP, LPE>(p, pre, my_act, grp, ENV_BASE);
                env_body<CM_PP, LPE>(p, nullptr, my_act, tape, ot, 0, grp, b_raw, live, ENV_BASE, nullptr, true, pre.rng_step, pre.step_count_in,
                                     pre.succ, pre.t_row, pre.t_col, pre.t_step0, pre.t_step, pre.t_rew, bad, FULLWG);
            } else env_body<CM_PP, LPE>(p, nullptr, my_act, tape, ot, 0, grp, b_raw, live, ENV_BASE);
        }
        // step t + 1 reads what this WAVE wrote (observation, masks, env state): its stores are performed before its next loads;
        // the CU's vector L1 is write-through and shared, so workgroup scope needs no cache maintenance (as rollout_chunk_kernel)
        const unsigned long long tf = probe ? __builtin_amdgcn_s_memtime() : 0ull;
        if constexpr (!CARRY) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (probe) { const unsigned long long t2 = __builtin_amdgcn_s_memtime(); g_w_probe[0] += 1; g_w_probe[1] += t1 - t0; g_w_probe[2] += t2 - t1; g_w_probe[4] += t2 - tf; }
    }
    if constexpr (CARRY) {
        if (tl.on) {                                                     // `obses = next_obses` + the counter advance (cm_chunk_tail) in here
            const int tx = thread_x(), grp = tx / LPE, sl = tx % LPE;
            const int b0 = blockIdx.x * mw::WG_ENVS + grp;
            if ((FULLWG || grp < envs) && b0 < a.S) {
                const float *oc = reinterpret_cast<const float *>(lds_w + obs_env);
                float *dst = tl.obs_dst + (size_t)b0 * 4 * a.d;
                const float rcp_d = p.rcp_d;
                for (int k = sl; k < 4 * a.d; k += LPE) { const int i = fdiv(k, a.d, rcp_d), f = k - i * a.d; dst[k] = oc[i * OBS_COPY_STRIDE + f]; }
            }
            if ((tx & 63) == 0) {
                const unsigned int last = gridDim.x * (blockDim.x >> 6) - 1;
                if (atomicAdd(tl.ticket, 1u) == last) { *tl.base += (uint32_t)c.n_steps; *tl.ticket = 0u; }
            }
        }
    }
}

bool shape_ok_rollout_w(int N, int d, int L, int n_act) { return policy_w_enabled() && mw::shape_ok_w(N, d, L, n_act); }

// Launch of the wave-owned rollout kernel; 1 = not available for this shape / handle.
int launch_rollout_w(mf::FwdArgs a, const cm_policy_weights *w, const void *w_pack, const cm_env *h, const cm_rng_tape &t, const cm_step_out &out,
                    void *stream, const ChunkArgs *chunk) {
    const EnvDev &d = h->dev;
    if (d.scen != CM_PP || d.lpe != 16 || d.M > 16 || d.N != 4) return 1;
    const size_t lds = mw::lds_policy_bytes(d.L) + (size_t)d.lds_env * mw::WG_ENVS;
    if (lds > 160 * 1024) return 1;                                      // larger maps: the env areas do not fit beside the weights
    const mw::WeightsW ww{ reinterpret_cast<const uint4 *>(w_pack), w->n_act };
    StridesW c{};
    c.n_steps = 1;
    if (chunk) {
        const long long st[13] = { chunk->obs, chunk->actions, chunk->probs, chunk->attn, chunk->reward, chunk->reward_f64, chunk->done, chunk->details,
                                   chunk->dist_adj, chunk->channels, chunk->prey_alive, chunk->success, chunk->path_len };
        for (long long v : st) if (v < 0 || v > 0x7fffffffLL) return 1;   // strides beyond 2^31 elements: the workgroup-tiled kernels take it
        c = StridesW{ chunk->n_steps, (int)st[0], (int)st[1], (int)st[2], (int)st[3], (int)st[4], (int)st[5], (int)st[6], (int)st[7], (int)st[8],
                      (int)st[9], (int)st[10], (int)st[11], (int)st[12] };
    }
    const bool use_tape = t.prey || t.spawn || t.iid_u || t.ge_u || t.ge_init_u;
    const int blocks = (a.S + mw::WG_ENVS - 1) / mw::WG_ENVS;
    static const int pre_flag = [] { const char *e = getenv("COMMARL_ENV_PREFETCH"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool pre = pre_flag && env_prefetch_ok<CM_PP, 16>(d), full = a.S % mw::WG_ENVS == 0;
    // carried form (state and observation from step to step inside the wave, no fence between steps): multi-step launches on
    // a constant adjacency without a channel model; the observation copy needs 4 rows x 24 floats in the env area's claim table
    static const int carry_flag = [] { const char *e = getenv("COMMARL_ROLLOUT_CARRY"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool carry = carry_flag && pre && !use_tape && c.n_steps > 1 && d.adj_const && d.ch_const && !a.adj && !a.chan &&
                       a.d <= OBS_COPY_STRIDE && 4 * d.S * d.S >= 4 * OBS_COPY_STRIDE * 4;
    TailW tl{};
    if (carry && chunk && chunk->tail_obs && chunk->tail_base && chunk->tail_folded) {
        tl = TailW{ chunk->tail_obs, chunk->tail_base, d.tail_ticket, 1 };
        *chunk->tail_folded = 1;
    }
    static const int shape_flag = [] { const char *e = getenv("COMMARL_ROLLOUT_SHAPE"); return (e && e[0] == '0') ? 0 : 1; }();
    const bool map10 = shape_flag && carry && d.S == 10 && d.M == 4 && d.R == 1 && d.W == 3 && d.d == 21 && d.lds_env == lds_env_bytes(10, 4, 4);
#define CM_RW_(LH, PR, FU, TP, CA, SH)                                                                                          \
    do {                                                                                                                        \
        static unsigned long long done = 0;                                                                                     \
        if (cm::dev_first(done))                                                                                                \
            CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&rollout_w_kernel<LH, PR, FU, TP, CA, SH>),               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                \
        hipLaunchKernelGGL((rollout_w_kernel<LH, PR, FU, TP, CA, SH>), dim3(blocks), dim3(256), lds, (hipStream_t)stream, a, ww, d, t, out, c, tl); \
    } while (0)
#define CM_RW(LH, PR, FU, TP, CA) do { if (CA && map10) CM_RW_(LH, PR, FU, TP, CA, (CA ? 1 : 0)); else CM_RW_(LH, PR, FU, TP, CA, 0); } while (0)
#define CM_RW2(LH) do { if (use_tape) CM_RW(LH, false, false, true, false);                                                     \
                        else if (carry) { if (full) CM_RW(LH, true, true, false, true); else CM_RW(LH, true, false, false, true); } \
                        else if (pre) { if (full) CM_RW(LH, true, true, false, false); else CM_RW(LH, true, false, false, false); } \
                        else { if (full) CM_RW(LH, false, true, false, false); else CM_RW(LH, false, false, false, false); } } while (0)
    if (d.L == 1) CM_RW2(1); else CM_RW2(2);
#undef CM_RW2
#undef CM_RW
#undef CM_RW_
    CM_HIP(hipGetLastError());
    if (d.stop == -1) {                                                  // env phase clocks of workgroup 0 (ENV_PROBE, cm_env_dev.h)
        unsigned long long hp[16];
        if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess && hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_env_probe), sizeof(hp)) == hipSuccess) {
            fprintf(stderr, "[rollout_w env probe] clk since env entry:");
            for (int i = 1; i < 10; ++i) fprintf(stderr, " p%d=%lld", i, (long long)(hp[i] - hp[0]));
            fprintf(stderr, "\n");
        }
    }
    if (d.stop == -2) {
        unsigned long long hp[5];
        if (hipStreamSynchronize((hipStream_t)stream) == hipSuccess && hipMemcpyFromSymbol(hp, HIP_SYMBOL(g_w_probe), sizeof(hp)) == hipSuccess && hp[0])
            fprintf(stderr, "[rollout_w probe] steps=%llu staging=%llu clk; per step: policy=%llu env=%llu clk (of which the closing fence %llu)\n", hp[0],
                    hp[3], hp[1] / hp[0], hp[2] / hp[0], hp[4] / hp[0]);
    }
    return CM_OK;
}

}  // namespace cm
