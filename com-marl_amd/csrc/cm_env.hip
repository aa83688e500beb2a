// cm_env.hip - batched Predator-Prey / Coverage env step for gfx950 (MI355X).
//
// Replaces, for B independent envs per launch, what the reference does one env at a time in
// Python:  VecEnvExecutor.step (garage/sampler/vec_env_executor.py:19-45) over
// PredatorPrey.step (envs/ma_gym/envs/predator_prey/predator_prey.py:494-519) /
// Coverage.step (envs/ma_gym/envs/coverage/coverage.py:319-401), the observation builders
// (predator_prey.py:173-204, coverage.py:198-212,448-480) and update_communication_state
// (custom_implement/env_communication.py:91-157,200-243; gilbert_elliot_loss_model.py:121-150).
//
// Mapping: a single-wave workgroup steps G = 64/LPE envs, LPE = 16 / 32 / 64 lanes per env (the smallest
// group that gives each agent / prey a lane): at N = 4 four envs share a wave and run in lockstep.
//   * SoA state lives in HBM ([B,N] int2 positions, [B,M] alive bytes, [B,S] visited row
//     bitmasks ...); a step reads it once, rebuilds the S x S occupancy tile in LDS (the
//     reference's string grid is derived state and never stored), and writes it back once.
//   * The reference's order-dependent parts (agent i sees the grid after agents < i moved;
//     prey j sees preys < j moved/captured) run as short wave-uniform loops against the LDS
//     tile.  Everything that does NOT depend on that order was hoisted out and runs one
//     lane per item: per-prey predator counts and the <=5 move trials depend only on the
//     (static) agent layer, so only the final "is the target still vacant" test is serial.
//   * Observation windows, the range adjacency and the channel masks are emitted with the 64
//     lanes striding the flattened [N*d] / [N*N] / [L*N*N] rows -> 256-B coalesced stores.
// The kernel is HBM/latency bound integer work; there is nothing GEMM-shaped here.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "cm_internal.h"
#include "cm_rng.h"
#include "cm_env_dev.h"

namespace cm {

// ---------------------------------------------------------------------------------------
// the step kernel: a single-wave workgroup steps G = 64 / LPE envs
// ---------------------------------------------------------------------------------------
template <int SCEN, int LPE>
__global__ __launch_bounds__(WAVE) void env_kernel(EnvDev p, const int32_t *__restrict__ actions, cm_rng_tape tape,
                                                  cm_step_out out, int reset_only) {
    constexpr int G = WAVE / LPE;
    const int grp = threadIdx.x / LPE;
    env_body<SCEN, LPE>(p, actions, nullptr, tape, out, reset_only, grp, blockIdx.x * G + grp, true, 0);
}

// Wide form: a 256-thread workgroup for the G = 64 / LPE envs that share one wave in the narrow kernel.  Wave 0 runs
// the step of those envs - everything order-dependent is wave-local - and leaves the new state in LDS; then all four
// waves emit observations, adjacency, channel draws and the state write-back, 256 / G lanes per env.  Pays whenever
// the narrow kernel cannot put more than one wave on a SIMD (the step is then one wave's instruction stream): for
// N = 54 with an IID channel the emission is ~1500 Philox calls and ~13 k stores per env, 56 of the narrow 78 us.
template <int SCEN, int LPE>
__global__ __launch_bounds__(256) void env_kernel_wide(EnvDev p, const int32_t *__restrict__ actions, cm_rng_tape tape,
                                                      cm_step_out out, int reset_only) {
    constexpr int G = WAVE / LPE, EL = 256 / G;        // envs per workgroup, emission lanes per env
    __shared__ int hand[4 * G];                        // per env: [0] emit?  [1] step count  [2] comm slot  [3] Philox step
    if (threadIdx.x < 4 * G) hand[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x < WAVE) {
        const int grp = threadIdx.x / LPE;
        env_body<SCEN, LPE>(p, actions, nullptr, tape, out, reset_only, grp, blockIdx.x * G + grp, true, 0, hand + 4 * grp);
    }
    __syncthreads();
    const int grp = threadIdx.x / EL, b = blockIdx.x * G + grp;
    if (b >= p.B || !hand[4 * grp]) return;
    Grp<EL> g;
    g.sub = 0; g.sl = threadIdx.x % EL;
    const Lds l = make_lds(p.S, p.N, p.M, p.lds_env * grp, p.status);
    const Rng rng{ (uint32_t)(p.env_id_offset + b), (uint32_t)hand[4 * grp + 3], p.key0, p.key1 };
    emit<SCEN, EL>(p, l, rng, tape, out, b, g, hand[4 * grp + 1], hand[4 * grp + 2]);
}

__global__ void fill_const_kernel(float *adj, float *ch, int B, int N, int L, int channel) {
    const size_t nA = (size_t)B * N * N, nC = (size_t)B * L * N * N;
    for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < nA + nC; k += (size_t)gridDim.x * blockDim.x) {
        if (k < nA) { if (adj) adj[k] = 1.0f; }
        else if (ch) {
            const size_t q = k - nA;
            const int ij = (int)(q % ((size_t)N * N)), i = ij / N, j = ij - i * N;
            ch[q] = (channel == CM_CH_FL) ? (i == j ? 1.0f : 0.0f) : 1.0f;   // env_communication.py:93-100
        }
    }
}

}  // namespace cm

using namespace cm;

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static double py_round2(double x) {         // Python round(x, 2): correctly rounded decimal, then back to binary
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, nullptr);
}

static void co_walls(const cm_env_cfg &c, std::vector<uint8_t> &g) {
    // wall ring + fixed obstacle rectangles scaled by r = m/10 (coverage.py:44,69-80,165-168,482-500)
    const int S = c.grid + 2, r = c.grid / 10;
    g.assign((size_t)S * S, C_EMPTY);
    for (int i = 0; i < S; ++i) g[i] = g[(size_t)(S - 1) * S + i] = g[(size_t)i * S] = g[(size_t)i * S + S - 1] = C_WALL;
    struct Rect { int r0, c0, h, w; };
    std::vector<Rect> ob = { { 2 * r + 1, 2 * r + 1, 6 * r, r }, { 3 * r + 1, 8 * r + 1, 4 * r, 2 * r } };   // 'Easy'
    if (c.obst_hard) {
        ob.push_back({ 1, 2 * r + 1, r, 3 * r });
        ob.push_back({ 1, 7 * r + 1, 2 * r, r });
        ob.push_back({ 4 * r + 1, 4 * r + 1, 2 * r, 3 * r });
        ob.push_back({ 8 * r + 1, 5 * r + 1, 2 * r, 2 * r });
        ob.push_back({ 8 * r + 1, 8 * r + 1, r, r });
    }
    for (const Rect &o : ob)
        for (int i = 0; i < o.h; ++i)
            for (int j = 0; j < o.w; ++j) {
                const int rr = o.r0 + i, cc = o.c0 + j;
                if (rr >= 0 && rr < S && cc >= 0 && cc < S) g[(size_t)rr * S + cc] = C_WALL;
            }
}

extern "C" int cm_env_create(const cm_env_cfg *cfg, cm_env_t *out) {
    if (!cfg || !out) return set_error(CM_ERR_ARG, "cm_env_create: null argument");
    const cm_env_cfg &c = *cfg;
    if (c.scenario != CM_PP && c.scenario != CM_CO) return set_error(CM_ERR_ARG, "scenario must be CM_PP or CM_CO");
    if (c.n_envs <= 0 || c.n_agents <= 0 || c.n_agents > 255) return set_error(CM_ERR_ARG, "n_envs > 0 and 0 < n_agents <= 255 required");
    const int S = c.scenario == CM_PP ? c.grid : c.grid + 2;
    if (S < 2 || S > 64) return set_error(CM_ERR_ARG, "grid side (incl. wall ring) must be in [2, 64]: cell coordinates travel as 6-bit fields, "
                                                     "visited rows as one or two 32-bit words");
    if (c.scenario == CM_PP && (c.n_preys < 0 || c.n_preys > 255)) return set_error(CM_ERR_ARG, "0 <= n_preys <= 255 required");
    if (c.scenario == CM_PP && (c.load < 2 || c.load > 4)) return set_error(CM_ERR_LOAD, "PP load must be 2, 3 or 4 (capv undefined otherwise, predator_prey.py:77-79)");
    if (c.scenario == CM_CO && c.grid % 10 != 0) return set_error(CM_ERR_ARG, "CO map must be a multiple of 10 (coverage.py:67)");
    if (c.n_hops < 1 || c.n_hops > 8) return set_error(CM_ERR_ARG, "1 <= n_hops <= 8 required");
    if (c.max_steps < 1 || c.max_path_length < 1) return set_error(CM_ERR_ARG, "max_steps and max_path_length must be >= 1");
    if (c.channel < CM_CH_FC || c.channel > CM_CH_GE) return set_error(CM_ERR_ARG, "bad channel");
    if (c.ge_flags < 0 || c.ge_flags > 5 || ((c.ge_flags >> 1) & 3) == 3) return set_error(CM_ERR_ARG, "bad ge_flags");
    if ((c.ge_flags & 1) && ((c.ge_flags >> 1) & 3) == 2)
        return set_error(CM_ERR_ARG, "GE: random initial state with one transition per env step (GE_INIT random + loss_apply 0) "
                                     "is shape-inconsistent in the reference (env_communication.py:121) and not built");
    const int M = c.scenario == CM_PP ? c.n_preys : 0;
    if ((long)c.n_agents + M > (long)(S * S) / 2) return set_error(CM_ERR_ARG, "too many agents+preys for the grid");

    cm_env *h = new cm_env();
    h->cfg = c;
    EnvDev &d = h->dev;
    memset(&d, 0, sizeof d);
    d.scen = c.scenario; d.B = c.n_envs; d.N = c.n_agents; d.M = M; d.S = S; d.R = c.rsen; d.W = 2 * c.rsen + 1;
    d.d = c.scenario == CM_PP ? 2 * d.W * d.W + 3 : 3 * d.W * d.W + 2 + (c.add_clock ? 1 : 0);
    d.load = c.load; d.max_steps = c.max_steps; d.mpl = c.max_path_length; d.L = c.n_hops;
    const int rc = (c.rcom + 1 >= c.grid) ? 0 : c.rcom;                 // env_communication.py:71-72
    d.adj_const = rc == 0; d.rc2 = 2 * rc * rc;
    d.channel = c.channel; d.ch_const = (c.channel == CM_CH_FC || c.channel == CM_CH_FL);
    d.ge_flags = c.ge_flags;
    d.add_clock = c.add_clock; d.rng_mode = c.rng_mode; d.env_id_offset = c.env_id_offset;
    d.ploss = c.ploss; d.pgb = c.pgb; d.pbg = c.pbg;
    d.cap_rew = c.capture_reward; d.step_cost = c.step_cost; d.move_cost = c.move_cost; d.penalty = c.penalty;
    d.lazy = c.lazy_penalty; d.revisit = c.revisit_penalty; d.final_reward = c.final_reward;
    d.key0 = (uint32_t)c.seed; d.key1 = (uint32_t)(c.seed >> 32);

    std::vector<uint8_t> walls((size_t)S * S, C_EMPTY);
    if (c.scenario == CM_CO) {
        co_walls(c, walls);
        int n = 0;
        for (uint8_t v : walls) n += (v == C_EMPTY);
        d.n_empty = n - c.n_agents;                                     // coverage.py:228-230
    }
    std::vector<double> rew_lut;
    if (c.scenario == CM_PP) {           // [0..M]: step + cap*c ; [M+1 .. M+1+N]: (mc*m)/N   (predator_prey.py:434,480)
        for (int k = 0; k <= M; ++k) rew_lut.push_back(c.step_cost + c.capture_reward * (double)k);
        for (int k = 0; k <= c.n_agents; ++k) rew_lut.push_back((c.move_cost * (double)k) / (double)c.n_agents);
    } else {                             // 5 x [0..N]: coef * (count / N)               (coverage.py:299-306)
        const double coef[5] = { c.capture_reward, c.move_cost, c.penalty, c.lazy_penalty, c.revisit_penalty };
        for (int t = 0; t < 5; ++t)
            for (int k = 0; k <= c.n_agents; ++k) rew_lut.push_back(coef[t] * ((double)k / (double)c.n_agents));
    }
    std::vector<float> lut_row(S), lut_col(S), lut_step(c.max_steps + 1);
    for (int i = 0; i < S; ++i) {
        if (c.scenario == CM_PP) { lut_row[i] = (float)((double)i / (double)S); lut_col[i] = (float)((double)i / (double)(S - 1)); }
        else lut_row[i] = lut_col[i] = (float)py_round2((double)i / (double)(S - 1));
    }
    for (int t = 0; t <= c.max_steps; ++t) lut_step[t] = (float)((double)t / (double)c.max_steps);

    // one arena for all state
    const size_t B = c.n_envs, N = c.n_agents;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    const size_t o_ap = take(B * N * sizeof(int2)), o_pp = take(B * (M ? M : 1) * sizeof(int2)), o_al = take(B * (M ? M : 1)),
                 o_vis = take(B * S * ((S + 31) / 32) * 4), o_sc = take(B * 4), o_tc = take(B * 4), o_su = take(B * 4),
                 o_ge = take(c.channel == CM_CH_GE ? B * N * N : 1), o_rs = take(B * 4), o_ac = take(B * N), o_st = take(4), o_tk = take(4),
                 o_bg = take((size_t)S * S), o_lr = take(S * 4), o_lc = take(S * 4), o_ls = take((c.max_steps + 1) * 4),
                 o_rl = take(rew_lut.size() * 8);
    h->arena_bytes = off;
    hipError_t e = hipMalloc(&h->arena, off);
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(env arena)"); }
    char *base = (char *)h->arena;
    e = hipMemset(base, 0, off);
    if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return hip_fail(e, "hipMemset"); }
    d.agent_pos = (int2 *)(base + o_ap); d.prey_pos = (int2 *)(base + o_pp); d.alive = (uint8_t *)(base + o_al);
    d.visited = (uint32_t *)(base + o_vis); d.step_count = (int32_t *)(base + o_sc); d.total_capture = (int32_t *)(base + o_tc);
    d.success = (int32_t *)(base + o_su); d.ge_state = (uint8_t *)(base + o_ge); d.rng_step = (uint32_t *)(base + o_rs);
    d.agent_cond = (uint8_t *)(base + o_ac);
    d.status = (int32_t *)(base + o_st);
    d.tail_ticket = (unsigned int *)(base + o_tk);
    d.base_grid = (const uint8_t *)(base + o_bg); d.lut_row = (const float *)(base + o_lr); d.lut_col = (const float *)(base + o_lc);
    d.lut_step = (const float *)(base + o_ls);
    d.rew_lut = (const double *)(base + o_rl);
    {   // constant tables and the initial GE state: a failed upload would leave silently wrong LUTs, so every copy is checked
        struct Up { size_t off; const void *src; size_t bytes; const char *what; };
        const Up ups[] = { { o_rl, rew_lut.data(), rew_lut.size() * 8, "reward LUT" }, { o_bg, walls.data(), walls.size(), "wall grid" },
                           { o_lr, lut_row.data(), (size_t)S * 4, "row LUT" }, { o_lc, lut_col.data(), (size_t)S * 4, "column LUT" },
                           { o_ls, lut_step.data(), (size_t)(c.max_steps + 1) * 4, "step LUT" } };
        for (const Up &u : ups) {
            e = hipMemcpy(base + u.off, u.src, u.bytes, hipMemcpyHostToDevice);
            if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return hip_fail(e, u.what); }
        }
        if (c.channel == CM_CH_GE) {
            e = hipMemset(base + o_ge, 1, B * N * N);
            if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return hip_fail(e, "hipMemset(GE state)"); }
        }
        e = hipMemset(base + o_ac, 1, B * N);                            // agent_condition = ones (predator_prey.py:74)
        if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return hip_fail(e, "hipMemset(agent condition)"); }
    }
    d.lds_env = lds_env_bytes(S, c.n_agents, M ? M : 1);
    {   // lanes per env: the smallest sub-wave group that still gives every agent / prey its own lane
        // measured (tools/envscale.py): at N <= 8 the step is dominated by the group-uniform serial loops, so
        // four envs per wave win at every batch size; with more agents the emit loops dominate and a full
        // wave per env has the lower latency until the batch is large enough to be throughput-bound.
        const int big = c.n_agents > M ? c.n_agents : M;
        d.lpe = big <= 8 ? 16 : ((big <= 32 && c.n_envs >= 8192) ? 32 : 64);
        if (const char *e = getenv("COMMARL_ENV_LPE")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) d.lpe = v; }
    }
    h->lds_bytes = (size_t)d.lds_env * (WAVE / d.lpe);
    { const char *e = getenv("COMMARL_ENV_STOP"); d.stop = e ? atoi(e) : 0; }
    { const char *e = getenv("COMMARL_ENV_SMALL"); d.no_small = (e && e[0] == '0') ? 1 : 0; }
    d.rcp_d = 1.0f / (float)d.d; d.rcp_W = 1.0f / (float)d.W; d.rcp_N = 1.0f / (float)d.N;
    d.rcp_WW = 1.0f / (float)(d.W * d.W); d.rcp_NN = 1.0f / (float)(d.N * d.N);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(h->arena); delete h; return hip_fail(e, "cm_env_create sync"); }
    *out = h;
    return CM_OK;
}

extern "C" int cm_env_destroy(cm_env_t h) {
    if (!h) return CM_OK;
    hipFree(h->arena);
    delete h;
    return CM_OK;
}

extern "C" int cm_env_obs_dim(cm_env_t h) { return h ? h->dev.d : set_error(CM_ERR_ARG, "null handle"); }
extern "C" int cm_env_n_empty_cells(cm_env_t h) { return h ? h->dev.n_empty : set_error(CM_ERR_ARG, "null handle"); }
extern "C" int cm_env_adj_is_const(cm_env_t h) { return h ? h->dev.adj_const : set_error(CM_ERR_ARG, "null handle"); }
extern "C" int cm_env_channels_are_const(cm_env_t h) { return h ? h->dev.ch_const : set_error(CM_ERR_ARG, "null handle"); }

extern "C" int cm_env_fill_constants(cm_env_t h, float *dist_adj, float *channels, void *stream) {
    if (!h) return set_error(CM_ERR_ARG, "null handle");
    const EnvDev &d = h->dev;
    hipLaunchKernelGGL(fill_const_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, dist_adj, channels, d.B, d.N, d.L, d.channel);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

int cm::check_tape(const cm_env *h, const cm_rng_tape *tape, bool is_reset) {
    if (h->cfg.rng_mode != CM_RNG_TAPE) return CM_OK;
    if (!tape) return set_error(CM_ERR_ARG, "rng_mode is TAPE but no tape was passed");
    if (!tape->spawn || tape->spawn_cap <= 0) return set_error(CM_ERR_ARG, "tape.spawn required in tape mode");
    if (!is_reset && h->dev.scen == CM_PP && h->dev.M > 0 && !tape->prey) return set_error(CM_ERR_ARG, "tape.prey required for PP steps");
    if (h->dev.channel == CM_CH_IID && !tape->iid_u) return set_error(CM_ERR_ARG, "tape.iid_u required for IID channel");
    if (h->dev.channel == CM_CH_GE && !tape->ge_u) return set_error(CM_ERR_ARG, "tape.ge_u required for GE channel");
    if (h->dev.channel == CM_CH_GE && ((h->dev.ge_flags >> 1) & 3) == 2 && !tape->ge_init_u)
        return set_error(CM_ERR_ARG, "tape.ge_init_u required for GE with a random initial state");
    return CM_OK;
}

static int launch(cm_env_t h, const int32_t *actions, const cm_rng_tape *tape, const cm_step_out *out, void *stream, int reset_only) {
    if (!h || !out) return set_error(CM_ERR_ARG, "null handle / out");
    if (!reset_only && !actions) return set_error(CM_ERR_ARG, "actions is null");
    int rc = check_tape(h, tape, reset_only);
    if (rc) return rc;
    cm_rng_tape t{};
    if (tape) t = *tape;
    const EnvDev &d = h->dev;
    const int G = WAVE / d.lpe;
    const dim3 grid((d.B + G - 1) / G), block(WAVE);
    const hipStream_t st = (hipStream_t)stream;
    static const bool wide_on = [] { const char *e = getenv("COMMARL_ENV_WIDE"); return !(e && e[0] == '0'); }();
    // wide form when the narrow kernel would put at most one wave on each of the 1024 SIMDs (measured, narrow -> wide:
    // N = 72 x 1024 envs 82 -> 66 us, N = 54 x 1024 78 -> 56 us; N = 24 x 2048 is two waves per SIMD and stays narrow, 36 vs 44 us)
    const int G_ = WAVE / d.lpe;
    // (small teams, LPE 16: 12.6 -> 13.1 us alone and 115 -> 101 M env-steps/s at config 2, the extra waves take the
    // registers the overlapping policy kernel needs - large teams only)
    if (wide_on && d.lpe == 64 && d.B <= 1536) {
        const dim3 wgrid((d.B + G_ - 1) / G_), wblock(256);
#define CM_WIDE(SC, LP) hipLaunchKernelGGL((env_kernel_wide<SC, LP>), wgrid, wblock, h->lds_bytes, st, d, actions, t, *out, reset_only)
        if (d.scen == CM_PP) CM_WIDE(CM_PP, 64); else CM_WIDE(CM_CO, 64);      // (d.lpe == 64 here)
#undef CM_WIDE
        CM_HIP(hipGetLastError());
        return CM_OK;
    }
#define CM_LAUNCH(SC, LP) hipLaunchKernelGGL((env_kernel<SC, LP>), grid, block, h->lds_bytes, st, d, actions, t, *out, reset_only)
    if (d.scen == CM_PP) { if (d.lpe == 16) CM_LAUNCH(CM_PP, 16); else if (d.lpe == 32) CM_LAUNCH(CM_PP, 32); else CM_LAUNCH(CM_PP, 64); }
    else { if (d.lpe == 16) CM_LAUNCH(CM_CO, 16); else if (d.lpe == 32) CM_LAUNCH(CM_CO, 32); else CM_LAUNCH(CM_CO, 64); }
#undef CM_LAUNCH
    CM_HIP(hipGetLastError());
    if (d.stop < 0) {                                                    // diagnostic: phase clocks of workgroup 0 (ENV_PROBE)
        unsigned long long h_probe[16];
        CM_HIP(hipStreamSynchronize(st));
        CM_HIP(hipMemcpyFromSymbol(h_probe, HIP_SYMBOL(g_env_probe), sizeof(h_probe)));
        fprintf(stderr, "[env probe] clk since entry:");
        for (int i = 1; i < 10; ++i) fprintf(stderr, " p%d=%lld", i, (long long)(h_probe[i] - h_probe[0]));
        fprintf(stderr, "\n");
    }
    return CM_OK;
}

extern "C" int cm_env_reset(cm_env_t h, const cm_rng_tape *tape, const cm_step_out *out, void *stream) {
    return launch(h, nullptr, tape, out, stream, 1);
}

extern "C" int cm_env_step(cm_env_t h, const int32_t *actions, const cm_rng_tape *tape, const cm_step_out *out, void *stream) {
    return launch(h, actions, tape, out, stream, 0);
}

extern "C" int cm_env_status(cm_env_t h) {
    if (!h) return set_error(CM_ERR_ARG, "null handle");
    CM_HIP(hipDeviceSynchronize());
    int32_t s = 0;
    CM_HIP(hipMemcpy(&s, h->dev.status, 4, hipMemcpyDeviceToHost));
    if (s) {
        int32_t z = 0;
        hipMemcpy(h->dev.status, &z, 4, hipMemcpyHostToDevice);
        return set_error(s, s == CM_ERR_ACTION ? "action outside 0..4" : (s == CM_ERR_TAPE ? "spawn tape exhausted" : (s == CM_ERR_TAPE_PREY ? "prey-move tape exhausted" : "kernel-side error")));
    }
    return CM_OK;
}

static int copy_state(cm_env_t h, const cm_env_state *s, bool to_host) {
    if (!h || !s) return set_error(CM_ERR_ARG, "null handle / state");
    CM_HIP(hipDeviceSynchronize());
    const EnvDev &d = h->dev;
    const size_t B = d.B, N = d.N, M = d.M, S = d.S;
    struct Item { void *host; void *dev; size_t bytes; };
    const Item items[] = {
        { s->agent_pos, d.agent_pos, B * N * 8 }, { s->prey_pos, d.prey_pos, B * M * 8 }, { s->prey_alive, d.alive, B * M },
        { s->visited, d.visited, B * S * ((S + 31) / 32) * 4 }, { s->step_count, d.step_count, B * 4 }, { s->total_capture, d.total_capture, B * 4 },
        { s->success, d.success, B * 4 }, { s->ge_state, d.ge_state, d.channel == CM_CH_GE ? B * N * N : 0 },
        { s->rng_step, d.rng_step, B * 4 } };
    for (const Item &it : items) {
        if (!it.host || !it.bytes) continue;
        if (to_host) CM_HIP(hipMemcpy(it.host, it.dev, it.bytes, hipMemcpyDeviceToHost));
        else CM_HIP(hipMemcpy(it.dev, it.host, it.bytes, hipMemcpyHostToDevice));
    }
    return CM_OK;
}

extern "C" int cm_env_get_state(cm_env_t h, const cm_env_state *host) { return copy_state(h, host, true); }
extern "C" int cm_env_set_state(cm_env_t h, const cm_env_state *host) { return copy_state(h, host, false); }

// ---------------------------------------------------------------------------------------------------------------
// dormant fault / delay helpers of custom_implement/env_communication.py:270-301
// ---------------------------------------------------------------------------------------------------------------
namespace cm {

// mode 1: iid_fault (:290-292)   cond[i] = 0 iff u_i < p                      (one uniform per agent)
// mode 2: GE_fault  (:294-301)   AS WRITTEN: np.where's tuple has length 1, so ONE draw decides all good agents
//                                (stay good iff u_G < 1 - p) and ONE all bad ones (recover iff u_B < r)
__global__ void agent_fault_kernel(EnvDev d, int mode, float p, float r, const float *__restrict__ tape_u, uint32_t fault_step) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= d.B) return;
    uint8_t *cond = d.agent_cond + (size_t)b * d.N;
    const uint32_t env = (uint32_t)(d.env_id_offset + b);
    if (mode == 1) {
        for (int i = 0; i < d.N; ++i) {
            const float u = tape_u ? tape_u[(size_t)b * d.N + i]
                                   : unit_f32(pick(philox4x32_10(env, fault_step, SITE_FAULT, (uint32_t)(i >> 2), d.key0, d.key1), i & 3));
            cond[i] = u < p ? 0 : 1;
        }
    } else {
        float ug, ub;
        if (tape_u) { ug = tape_u[2 * b]; ub = tape_u[2 * b + 1]; }
        else { const u32x4 x = philox4x32_10(env, fault_step, SITE_FAULT, 0u, d.key0, d.key1); ug = unit_f32(x.x); ub = unit_f32(x.y); }
        const uint8_t g = ug < 1.0f - p ? 1 : 0, bd = ub < r ? 1 : 0;
        for (int i = 0; i < d.N; ++i) cond[i] = cond[i] ? g : bd;
    }
}

// delays_init (:271-279, init = 1: hop 0 from the adjacency alone, later hops from the link masks alone) and calc_delays
// (:281-286, init = 0: loss = adjacency * link for every hop, hop 0 counts on from old_delays [B,N,N])
__global__ void delays_kernel(int B, int L, int N, const float *__restrict__ adj, const float *__restrict__ link,
                              const int32_t *__restrict__ old_delays, int delay_th, int init, int32_t *__restrict__ delays) {
    const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x, NN = (size_t)N * N;
    if (k >= (size_t)B * NN) return;
    const size_t b = k / NN, ij = k - b * NN;
    const float a = adj ? adj[k] : 1.0f;
    int prev = 0;
    for (int l = 0; l < L; ++l) {
        const float lk = link ? link[(b * L + l) * NN + ij] : 1.0f;
        int v;
        if (init) v = l == 0 ? (a == 0.0f ? delay_th : 1) : (lk == 0.0f ? prev + 1 : 1);
        else v = (a * lk == 0.0f) ? (l == 0 ? old_delays[k] : prev) + 1 : 1;
        delays[(b * L + l) * NN + ij] = v;
        prev = v;
    }
}

}  // namespace cm

extern "C" int cm_env_agent_condition(cm_env_t h, const uint8_t *set_host, uint8_t *get_host) {
    if (!h) return set_error(CM_ERR_ARG, "null handle");
    CM_HIP(hipDeviceSynchronize());
    const size_t n = (size_t)h->dev.B * h->dev.N;
    if (set_host) CM_HIP(hipMemcpy(h->dev.agent_cond, set_host, n, hipMemcpyHostToDevice));
    if (get_host) CM_HIP(hipMemcpy(get_host, h->dev.agent_cond, n, hipMemcpyDeviceToHost));
    return CM_OK;
}

extern "C" int cm_env_agent_fault(cm_env_t h, int32_t mode, float p, float r, const float *tape_u, uint32_t fault_step, void *stream) {
    if (!h) return set_error(CM_ERR_ARG, "null handle");
    if (mode != 1 && mode != 2) return set_error(CM_ERR_ARG, "cm_env_agent_fault: mode 1 (iid_fault) or 2 (GE_fault)");
    if (h->cfg.rng_mode == CM_RNG_TAPE && !tape_u) return set_error(CM_ERR_ARG, "cm_env_agent_fault: rng_mode is TAPE but no uniforms were passed");
    hipLaunchKernelGGL(agent_fault_kernel, dim3((h->dev.B + 63) / 64), dim3(64), 0, (hipStream_t)stream, h->dev, mode, p, r, tape_u, fault_step);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_comm_delays(int32_t B, int32_t L, int32_t N, const float *dist_adj, const float *link_loss, const int32_t *old_delays,
                              int32_t delay_th, int32_t init, int32_t *delays, void *stream) {
    if (!delays) return set_error(CM_ERR_ARG, "cm_comm_delays: null output");
    if (!init && !old_delays) return set_error(CM_ERR_ARG, "cm_comm_delays: calc_delays needs old_delays");
    if (B <= 0 || L <= 0 || N <= 0) return CM_OK;
    const size_t n = (size_t)B * N * N;
    hipLaunchKernelGGL(delays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, L, N, dist_adj, link_loss,
                       old_delays, delay_th, init, delays);
    CM_HIP(hipGetLastError());
    return CM_OK;
}
