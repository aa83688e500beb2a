/*
 * commarl.h - C ABI of libcommarl_hip.so: the MI355X (gfx950) batched rollout + GNN-PPO
 * hot path of Com-MARL.
 *
 * The reference (cnuns/Com-MARL) is pure Python and has no FFI of its own; the drop-in
 * boundary is the set of Python classes its runner scripts import by name
 * (exp_runners/predatorprey/runner_pp_commDP.py:22-28).  Those classes are re-provided by
 * the host package (com-marl_amd/) as thin veneers over THIS C ABI.  Each entry point
 * below cites the reference interface it replaces (file:line relative to the reference
 * tree).  Conventions:
 *   - every function returns 0 on success, <0 on error; cm_last_error() gives the text;
 *     no exception ever crosses the ABI;
 *   - all data pointers are DEVICE pointers unless the name ends in _host; buffers are
 *     caller-owned (torch-allocated in the Python veneer) except the env state, which the
 *     handle owns;
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  Launches are
 *     asynchronous; nothing here synchronises except get/set_state and cm_env_status;
 *   - one handle is not thread-safe; distinct handles are independent (one process per GPU).
 */
#ifndef COMMARL_H
#define COMMARL_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM_ABI_VERSION 3

enum { CM_PP = 0, CM_CO = 1 };                                  /* scenario */
enum { CM_CH_FC = 0, CM_CH_FL = 1, CM_CH_IID = 2, CM_CH_GE = 3 }; /* channel model */
enum { CM_RNG_PHILOX = 0, CM_RNG_TAPE = 1 };

enum {
    CM_OK = 0,
    CM_ERR_ARG = -1,        /* bad argument / unsupported config */
    CM_ERR_TAPE = -2,       /* RNG tape exhausted (parity mode) */
    CM_ERR_TAPE_PREY = -3,
    CM_ERR_ACTION = -4,     /* action outside 0..4: predator_prey.py:255, coverage.py:347 raise */
    CM_ERR_LOAD = -5,       /* PP load not in {2,3,4}: capv undefined (predator_prey.py:77-79) */
    CM_ERR_HIP = -6,
    CM_ERR_NOMEM = -7
};

/* Env configuration = the `params` dict the reference envs read
 * (predator_prey.py:52-82, coverage.py:40-96, env_communication.py:10-75) after the arg
 * post-processing of exp_runners/env_uitils.py:174-217. */
typedef struct cm_env_cfg {
    int32_t scenario;        /* CM_PP / CM_CO */
    int32_t n_envs;          /* B independent envs owned by this handle */
    int32_t n_agents;        /* N */
    int32_t n_preys;         /* M (PP) */
    int32_t grid;            /* PP: G = --map; CO: m = --map (grid side m+2 incl. wall ring) */
    int32_t rsen;            /* --sen: window (2R+1)^2 */
    int32_t load;            /* --cap: 2 reward_default, 3/4 reward_individual */
    int32_t max_steps;       /* env time limit (PP max_env_steps; CO ctor max_steps=400) */
    int32_t max_path_length; /* VecEnvExecutor truncation (vec_env_executor.py:33-34) */
    int32_t n_hops;          /* n_gcn_layers L */
    int32_t rcom;            /* trRcom (rule Rcom+1>=map -> fully connected applied inside) */
    int32_t channel;         /* CM_CH_* */
    int32_t obst_hard;       /* CO obstComplex: 0 Easy / 1 Hard */
    int32_t add_clock;       /* CO */
    int32_t rng_mode;        /* CM_RNG_PHILOX production / CM_RNG_TAPE parity */
    int32_t env_id_offset;   /* global id of local env 0: rank r of k owns [r*B, (r+1)*B) */
    float ploss, pgb, pbg;
    int32_t ge_flags;        /* GE channel variants (env_communication.py:106-157), 0 = the runners' defaults:
                                bit 0 set  = loss_apply 0: ONE state transition per env step shared by all hops (else one per hop);
                                bits 1-2   = GE_INIT: 0 all links good, 1 all bad, 2 random with the stationary bad rate
                                             Pgb/(Pgb+Pbg) (not together with bit 0: the reference's state is shape-
                                             inconsistent for that pair, :121) */
    double capture_reward, step_cost, move_cost, penalty, lazy_penalty, revisit_penalty, final_reward;
    uint64_t seed;
} cm_env_cfg;

/* Recorded draws for parity mode (device pointers, any may be NULL when unused):
 * SURVEY.md App. A-6 call sites.  slot 0 = the step's comm update, slot 1 = the reset's. */
typedef struct cm_rng_tape {
    const uint8_t *prey;     /* [B,M,5]  np.random.choice outcomes (predator_prey.py:401), 255 unused */
    const int32_t *spawn;    /* [B,cap,2] random.randint pairs (predator_prey.py:156,165; coverage.py:185) */
    int32_t spawn_cap;
    int32_t _pad;
    const float *iid_u;      /* [B,2,L,N,N]   torch.rand (env_communication.py:212) */
    const float *ge_u;       /* [B,2,L,2,N,N] torch.rand (gilbert_elliot_loss_model.py:138,142) */
    const float *ge_init_u;  /* [B,N,N] torch.rand of get_init_state (:84-87), GE_INIT random only */
} cm_rng_tape;

/* What VecEnvExecutor.step returns + the env attributes the sampler reads each step
 * (centralized_ma_on_policy_vectorized_sampler.py:123-141).  Any pointer may be NULL to
 * skip that output; dist_adj / channels are skipped automatically when constant
 * (fully connected / FC / FL: filled once by cm_env_fill_constants). */
typedef struct cm_step_out {
    float *obs;              /* [B,N,d]   next observation (reset obs where done) */
    float *reward;           /* [B]       f32 view of the reward */
    double *reward_f64;      /* [B]       reward as the Python float the reference produces */
    uint8_t *done;           /* [B] */
    int32_t *details;        /* [B,6] capture, move, penalty, lazy, revisit|watching, final (sums over agents) */
    float *dist_adj;         /* [B,N,N] */
    float *channels;         /* [B,L,N,N] */
    uint8_t *prey_alive;     /* [B,M] env_infos['prey_alive'] (pre-reset) */
    int32_t *success;        /* [B] env.success after the step */
    int32_t *path_len;       /* [B] length of the path that ended at this step (0 = still running):
                                what the sampler adds to n_samples, x N (sampler.py:225) */
} cm_step_out;

/* Host-side snapshot of the SoA state (for parity fixtures / checkpoints). */
typedef struct cm_env_state {
    int32_t *agent_pos;      /* [B,N,2] */
    int32_t *prey_pos;       /* [B,M,2] */
    uint8_t *prey_alive;     /* [B,M] */
    uint32_t *visited;       /* [B,S,W] row bitmasks (CO): W = ceil(S / 32) words per grid row - [B,S] up to map 30; cell (r, c) =
                              * bit c & 31 of word (r, c >> 5) */
    int32_t *step_count;     /* [B] */
    int32_t *total_capture;  /* [B] */
    int32_t *success;        /* [B] */
    uint8_t *ge_state;       /* [B,N,N] */
    uint32_t *rng_step;      /* [B] */
} cm_env_state;

typedef struct cm_env *cm_env_t;

int cm_abi_version(void);
const char *cm_last_error(void);

/* PredatorPreyWrapper(...)/CoverageWrapper(...) construction (predatorprey_wrapper.py:23-44,
 * coverage_wrapper.py:16-33) for B envs at once. */
int cm_env_create(const cm_env_cfg *cfg, cm_env_t *out);
int cm_env_destroy(cm_env_t h);
int cm_env_obs_dim(cm_env_t h);          /* d per agent */
int cm_env_n_empty_cells(cm_env_t h);    /* coverage.py:228-230 */
int cm_env_adj_is_const(cm_env_t h);     /* 1 when Rcom rule gives fully connected (env_communication.py:71-72) */
int cm_env_channels_are_const(cm_env_t h);
/* write the constant dist_adj [B,N,N] / channels [B,L,N,N] once */
int cm_env_fill_constants(cm_env_t h, float *dist_adj, float *channels, void *stream);

/* VecEnvExecutor.reset (vec_env_executor.py:47-54): reset every env, emit obs + comm state. */
int cm_env_reset(cm_env_t h, const cm_rng_tape *tape, const cm_step_out *out, void *stream);
/* VecEnvExecutor.step (vec_env_executor.py:19-45) over env.step (predator_prey.py:494-519,
 * coverage.py:319-401): actions int32 [B,N]. */
int cm_env_step(cm_env_t h, const int32_t *actions, const cm_rng_tape *tape, const cm_step_out *out, void *stream);
/* synchronises; returns the first kernel-side error (CM_ERR_TAPE / CM_ERR_ACTION ...) or 0 */
int cm_env_status(cm_env_t h);
int cm_env_get_state(cm_env_t h, const cm_env_state *host);
int cm_env_set_state(cm_env_t h, const cm_env_state *host);

/* PP agent_condition (predator_prey.py:74,152,258): [B,N] bytes, 0 = the agent's moves are not applied; every env reset
 * puts its row back to ones.  HOST pointers; either may be NULL.  Synchronises. */
int cm_env_agent_condition(cm_env_t h, const uint8_t *set_host, uint8_t *get_host);
/* The reference's dormant agent-fault models, applied to agent_condition (custom_implement/env_communication.py:290-301;
 * never called by the reference's own code): mode 1 = iid_fault(n, p_fault = p); mode 2 = GE_fault(condition, p, r) as
 * written (ONE draw for all good agents, ONE for all bad ones).  tape_u (DEVICE, [B,N] for mode 1, [B,2] for mode 2):
 * the uniforms np.random.choice would consume, or NULL = Philox site 9 at (global env id, fault_step). */
int cm_env_agent_fault(cm_env_t h, int32_t mode, float p, float r, const float *tape_u, uint32_t fault_step, void *stream);
/* Link-delay counters (env_communication.py:271-286), a pure function of DEVICE arrays: init != 0 = delays_init(adjacency
 * [B,N,N], link_loss [B,L,N,N], delay_th); init == 0 = calc_delays(adjacency, link_loss, old_delays [B,N,N]).
 * delays [B,L,N,N] int32.  NULL adjacency / link_loss = ones. */
int cm_comm_delays(int32_t B, int32_t L, int32_t N, const float *dist_adj, const float *link_loss, const int32_t *old_delays,
                   int32_t delay_th, int32_t init, int32_t *delays, void *stream);

/* Comm-DP policy weights (device pointers), reference state_dict names in comments
 * (SURVEY.md §8 a-16).  Linear weights are passed TRANSPOSED [in,out] (contiguous over the
 * output index) - the veneer keeps a transposed device copy; GCN weights are already [in,out]. */
typedef struct cm_policy_weights {
    int32_t d, n_agents, n_hops, enc_hidden, emb, h1, h2, h3, n_act;
    int32_t no_residual;             /* 0 (default): x = E + H_L (comm_categorical_mlp_policy.py:74-77); 1: x = H_L */
    const float *enc_w1t, *enc_b1;   /* encoder._layers.0.linear.{weight^T,bias}          [d,128],[128] */
    const float *enc_w2t, *enc_b2;   /* encoder._output_layers.0.linear.*                 [128,64],[64] */
    const float *attn_wt;            /* attention_layer.linear_in.weight^T                [64,64] */
    const float *gcn_w, *gcn_b;      /* gcn_layers.{l}.{weight,bias}                      [L,64,64],[L,64] */
    const float *hd_w1t, *hd_b1;     /* categorical_output_layer._layers.0.linear.*       [64,128] */
    const float *hd_w2t, *hd_b2;     /* ..._layers.1.linear.*                             [128,64] */
    const float *hd_w3t, *hd_b3;     /* ..._layers.2.linear.*                             [64,32] */
    const float *hd_w4t, *hd_b4;     /* ..._output_layers.0.linear.*                      [32,5] */
    const float *mfma_pack;          /* cm_policy_pack() output, or NULL: without it the generic VALU kernel runs */
} cm_policy_weights;

typedef struct cm_critic_weights {
    int32_t d, n_agents, n_hops, enc_hidden, emb, dec_hidden;
    int32_t no_residual, _pad;
    const float *enc_w1t, *enc_b1, *enc_w2t, *enc_b2, *attn_wt, *gcn_w, *gcn_b;
    const float *dec_w1t, *dec_b1;   /* baseline_aggregator._mean_module._layers.0.linear.*        [64,64] */
    const float *dec_w2t, *dec_b2;   /* baseline_aggregator._mean_module._output_layers.0.linear.* [64,1] */
    const float *mfma_pack;          /* cm_critic_pack() output, or NULL */
} cm_critic_weights;

/* Matrix-core operand pack.  The MFMA forward kernels read every dense layer's weights as ready-made
 * v_mfma_f32_16x16x4_f32 B fragments - [column tile][16-deep k block][lane][4] with the zero padding baked in - so a
 * wave fetches 1 KB per instruction, unconditionally, instead of 4 predicated dword gathers per fragment.
 * cm_*_pack_bytes: size of the pack for this net, 0 when the shape has no matrix-core instantiation (then leave
 * mfma_pack NULL).  cm_*_pack: (re)build it on `stream` from the plain [in,out] weights in *w - call after every
 * weight update; the buffer is caller-owned and may be rewritten in place (captured hipGraphs keep working).
 * The default kernels carry every weight as an (hi, lo) pair of f16 values: a weight with |w| > 65504, inf or NaN cannot
 * be carried, and cm_*_pack REFUSE such a net (CM_ERR_ARG, text in cm_last_error()) instead of packing +-inf planes.  The
 * check costs one 4-byte copy and a wait for the pack kernels on `stream`; it is skipped while `stream` is being captured
 * (a capture cannot wait) and with COMMARL_PACK_CHECK=0.  Observations are expected in the envs' range ([-1, 1]); a
 * foreign caller's |obs| > 65504 saturates the same way and is NOT checked per step. */
size_t cm_policy_pack_bytes(const cm_policy_weights *w);
int cm_policy_pack(const cm_policy_weights *w, float *pack, void *stream);
/* The pack has sections - the all-f32 fragments (COMMARL_POLICY_KERNEL=f32, shapes without an f16 instantiation), the f16-split
 * fragments (every default kernel incl. the training forward), the wave-owned rollout kernel's fragments (teams of 4) - and a
 * caller that knows its next consumer may refresh only what that consumer reads: an optimiser step followed by a training
 * forward needs CM_PACK_F16 alone (10 launches and no wait instead of ~30 and one), the next rollout CM_PACK_ALL.
 * cm_policy_pack / cm_critic_pack = CM_PACK_ALL.  A section that was skipped is STALE until a later call packs it. */
#define CM_PACK_F32 1
#define CM_PACK_F16 2
#define CM_PACK_WAVE 4
#define CM_PACK_CHECK 8           /* the f16 range check (waits for the pack kernels) */
#define CM_PACK_ALL 15
int cm_policy_pack_sections(const cm_policy_weights *w, float *pack, int32_t sections, void *stream);
size_t cm_critic_pack_bytes(const cm_critic_weights *w);
int cm_critic_pack(const cm_critic_weights *w, float *pack, void *stream);
int cm_critic_pack_sections(const cm_critic_weights *w, float *pack, int32_t sections, void *stream);

/* CommCategoricalMLPPolicy.get_actions (comm_categorical_mlp_policy.py:98-119) for S env
 * states in one fused launch: encoder -> attention -> L x (mask, renorm, GCN) -> residual ->
 * head -> softmax x avail -> renorm -> sample (inverse CDF on the Philox stream) or argmax.
 *   obs [S,N,d]; avail [S,N,A] or NULL (= all ones, predatorprey_wrapper.py:46-51);
 *   dist_adj [S,N,N] or NULL (= ones); channels [S,L,N,N] or NULL (= ones);
 *   out: actions int32 [S,N] (or NULL), probs [S,N,A] (or NULL), attn [S,N,N] (or NULL).
 *   The sampler's Philox counter word is policy_step + (policy_step_base ? *policy_step_base : 0);
 *   the device-side base lets a captured hipGraph be replayed with fresh draws. */
int cm_policy_forward(const cm_policy_weights *w, int32_t n_samples, const float *obs, const float *avail,
                      const float *dist_adj, const float *channels, uint64_t seed, int32_t env_id_offset,
                      uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                      float *probs, float *attn, void *stream);

/* CommBaseCritic.forward (comm_base_critic.py:91-114): values [S] = sum over agents. */
int cm_critic_forward(const cm_critic_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                      const float *channels, float *values, void *stream);

/* Training forward (PPO update): the same fused forward, which additionally stores every activation the backward pass
 * needs ONCE, as f32 [R = S * n_agents rows, width] (the value the rest of the network saw): what torch's autograd would
 * keep layer by layer through ~20 separate kernels.  Pointers that are NULL are skipped.
 *   a1 [R,128] encoder hidden layer     e [R,64] encoder output E     q [R,64] attention query E.Wa^T
 *   hw[l] [R,64] H_l.Wg_l               h[l] [R,64] hop l's output tanh(A_l.hw[l] + b_l); the LAST hop's entry includes the
 *                                       residual (x = E + H_L, comm_categorical_mlp_policy.py:74-77) unless no_residual
 *   x1, x2, x3  head hidden layers: policy [R,128], [R,64], [R,32]; critic x1 [R,64] only
 *   out         policy: logits [R, n_act] (before softmax / avail mask); critic: per-agent value [R] (before the sum)
 *   probs       policy only: the action probabilities [R, n_act] exactly as cm_policy_forward returns them (no avail mask)
 * attn [S,N,N] is written as in cm_policy_forward.  Returns 1 - nothing done - when there is no saved-forward
 * instantiation for the shape (teams of <= 128 agents, n_hops <= 4, obs dim <= 96): the caller then runs layer by layer. */
typedef struct cm_fwd_saves {
    float *a1, *e, *q, *hw[4], *h[4], *x1, *x2, *x3, *out, *probs;
} cm_fwd_saves;
int cm_policy_forward_saved(const cm_policy_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                            const float *channels, float *attn, const cm_fwd_saves *sv, void *stream);
int cm_critic_forward_saved(const cm_critic_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                            const float *channels, float *attn, float *values, const cm_fwd_saves *sv, void *stream);
/* cm_policy_forward_saved / cm_critic_forward_saved for teams of 4 on the wave-owned kernel of the rollout (csrc/cm_policy_w_dev.h): one persistent
 * workgroup per CU stages the weights once and walks the batch, a wave carries 16 agent rows through the whole net in registers
 * and stores the saved activations from there.  Needs the CM_PACK_WAVE section of the operand pack to be current (between
 * optimiser steps: cm_policy_pack_sections(..., CM_PACK_F16 | CM_PACK_WAVE, ...)).  Same saves, same values to the f16-split
 * kernels' rounding (1e-6 relative).  Returns 1 - nothing done - for every other shape: call cm_policy_forward_saved / cm_critic_forward_saved. */
int cm_policy_forward_saved_wave(const cm_policy_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                                 const float *channels, float *attn, const cm_fwd_saves *sv, void *stream);
int cm_critic_forward_saved_wave(const cm_critic_weights *w, int32_t n_samples, const float *obs, const float *dist_adj,
                                 const float *channels, float *attn, float *values, const cm_fwd_saves *sv, void *stream);

/* One rollout step in ONE launch: cm_policy_forward over the B envs of `h` followed, inside the same workgroups, by
 * cm_env_step on the sampled actions (which travel through LDS and are also written to `actions`): what one iteration
 * of the sampler loop does (centralized_ma_on_policy_vectorized_sampler.py:133-141).  Arguments and results are those
 * of the two calls (n_samples is the handle's B; obs / dist_adj / channels are the CURRENT step's inputs, `out` receives
 * the next step's), bit-identical to calling them back to back.
 * Returns 0 on success, < 0 on error, and 1 - having done nothing - when this (scenario, team size, obs dim) has no
 * fused instantiation, no operand pack was supplied, or COMMARL_FUSED=0: call the two entry points instead. */
int cm_rollout_step(cm_env_t h, const cm_policy_weights *w, const float *obs, const float *avail,
                    const float *dist_adj, const float *channels, uint64_t seed, int32_t env_id_offset,
                    uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                    float *probs, float *attn, const cm_rng_tape *tape, const cm_step_out *out, void *stream);

/* n_steps consecutive rollout steps in ONE launch: step t = cm_rollout_step with every pointer advanced by t x its
 * per-step stride (time-major trajectory buffers [T(+1), B, ...]: obs / dist_adj / channels are read at slot t and -
 * through `out` - written at slot t + 1) and policy_step + t.  A workgroup keeps its envs for the whole chunk and no
 * grid-wide synchronisation separates the steps, so env phases overlap other workgroups' matrix phases.  Production
 * RNG only (tape mode is refused); no avail mask.  Results are bit-identical to n_steps calls of cm_rollout_step.
 * Strides are in ELEMENTS of the respective buffer; unused outputs may have stride 0.  Returns as cm_rollout_step. */
typedef struct cm_chunk_strides {
    int64_t obs, actions, probs, attn, reward, reward_f64, done, details, dist_adj, channels, prey_alive, success, path_len;
} cm_chunk_strides;
int cm_rollout_chunk(cm_env_t h, const cm_policy_weights *w, int32_t n_steps, const cm_chunk_strides *strides,
                     const float *obs, const float *dist_adj, const float *channels, uint64_t seed,
                     int32_t env_id_offset, uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy,
                     int32_t *actions, float *probs, float *attn, const cm_step_out *out, void *stream);

/* cm_rollout_chunk followed by cm_chunk_tail (below) for the same chunk: the observation (and masks, where the env produces
 * them) the last step wrote - out->obs + (n_steps - 1) * strides->obs, ... - carried into obs_next / dist_adj_next /
 * channels_next (slot 0 of the caller's ring; the mask pointers may be NULL) and *policy_step_base += n_steps.  Where the wave-owned
 * kernel takes the chunk the tail is part of that launch (a wave writes its envs' next observation itself, the last wave to
 * finish advances the counter): one launch per chunk; everywhere else the two launches of the separate calls.  Returns as
 * cm_rollout_chunk (1: nothing was launched, tail included). */
int cm_rollout_chunk_tail(cm_env_t h, const cm_policy_weights *w, int32_t n_steps, const cm_chunk_strides *strides,
                          const float *obs, const float *dist_adj, const float *channels, uint64_t seed,
                          int32_t env_id_offset, uint32_t policy_step, uint32_t *policy_step_base, int32_t greedy,
                          int32_t *actions, float *probs, float *attn, const cm_step_out *out, float *obs_next,
                          float *dist_adj_next, float *channels_next, void *stream);

/* End of a chunk of rollout steps, in ONE launch: what the sampler loop does between two steps that a captured chunk
 * cannot leave to the host - `obses = next_obses` (centralized_ma_on_policy_vectorized_sampler.py:232): the slot the
 * last step wrote (src) becomes slot 0 (dst) for up to three buffers (observations, dist_adj, channels; bytes multiple
 * of 4, pointers 4-byte aligned, a NULL / 0-byte entry is skipped) - and the advance of the device-side Philox counter
 * base of the action sampler by n_steps.  Replaces two framework kernels + a blit per chunk (and the idle gaps
 * around them) by one kernel of this library. */
int cm_chunk_tail(uint32_t *policy_step_base, uint32_t n_steps, const void *src0, void *dst0, size_t bytes0,
                  const void *src1, void *dst1, size_t bytes1, const void *src2, void *dst2, size_t bytes2, void *stream);

/* Plain row-wise MLPs: the non-communicating policies and the Gaussian baseline of the reference's Obs-DP / CENT
 * runners.  Layer l:  y = x . wt[l] + b[l]  (wt TRANSPOSED [in,out] as above), followed by tanh
 * when bit l of tanh_mask is set.  First layer at most 128 outputs; any layer at most 1024. */
#define CM_MLP_MAX_LAYERS 6
typedef struct cm_mlp_weights {
    int32_t in_dim, n_layers;
    int32_t out_dim[CM_MLP_MAX_LAYERS];
    int32_t tanh_mask, _pad;
    const float *wt[CM_MLP_MAX_LAYERS];
    const float *b[CM_MLP_MAX_LAYERS];   /* NULL = no bias */
    const float *mfma_pack;              /* cm_mlp_pack() output (all layers, B fragments as in cm_policy_pack), or NULL:
                                            the kernel then gathers the plain weights (slower) */
} cm_mlp_weights;
size_t cm_mlp_pack_bytes(const cm_mlp_weights *w);
int cm_mlp_pack(const cm_mlp_weights *w, float *pack, void *stream);

/* DecCategoricalMLPPolicy.get_actions (com_marl/torch/policies/dec_categorical_mlp_policy.py:106-176; encoder 2
 * layers + head 2 layers = one 4-layer chain per AGENT row: rows = S*N, groups = 1) and
 * CentralizedCategoricalMLPPolicy.get_actions (centralized_categorical_mlp_policy.py:61-118; one chain per ENV row
 * with N*n_act logits: rows = S, groups = N).  x [rows,in_dim]; the last layer has groups*n_act outputs; per group:
 * softmax x avail ([rows,groups,n_act] or NULL = ones), renormalise, then argmax (greedy) or inverse-CDF sample on the
 * Philox stream (counter: env = env_id_offset + flat/agents_per_env, step, site 7, agent = flat % agents_per_env with
 * flat = row*groups + group - the same stream cm_policy_forward uses).
 *   out: actions int32 [rows,groups] (or NULL), probs [rows,groups,n_act] (or NULL). */
int cm_mlp_policy_forward(const cm_mlp_weights *w, int32_t rows, int32_t groups, int32_t n_act, int32_t agents_per_env,
                          const float *x, const float *avail, uint64_t seed, int32_t env_id_offset,
                          uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                          float *probs, void *stream);
/* GaussianMLPBaseline.forward mean (com_marl/torch/baselines/gaussian_mlp_baseline.py:100-115): the last layer has
 * one output; values [rows]. */
int cm_mlp_value_forward(const cm_mlp_weights *w, int32_t rows, const float *x, float *values, void *stream);

/* Adjacency-masked aggregation, one GCN hop, for S samples (comm_base_net.py:101-105 +
 * graph_conv_module.py:63-70):  A = M*R*C; A /= rowsum+1e-12; out = tanh(A.(HW) + b).
 *   attn M [S,N,N]; dist_adj R [S,N,N] or NULL; channel C_l [S,N,N] with element stride
 *   ch_stride between samples (so a [S,L,N,N] tensor can be sliced) or NULL;
 *   hw [S,N,E]; bias [E] or NULL; out [S,N,E].  Saves nothing: backward recomputes A. */
int cm_masked_agg_forward(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                          const float *chan, int64_t ch_stride, const float *hw, const float *bias, float *out,
                          void *stream);
/* grads of the op above: given out and d_out returns d_attn [S,N,N], d_hw [S,N,E], d_bias [E] (accumulated with
 * atomics; caller zeroes).  out_minus (or NULL): the op's output is out - out_minus (the caller holds x = E + H of
 * comm_base_net.py:107 and E: saves the subtraction pass). */
int cm_masked_agg_backward(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                           const float *chan, int64_t ch_stride, const float *hw, const float *out, const float *out_minus,
                           const float *d_out, float *d_attn, float *d_hw, float *d_bias, void *stream);
/* The same with the bias gradient spread over `bias_replicas` >= 1 rows: d_bias [bias_replicas][E], zero on entry like every
 * accumulated gradient; a workgroup adds its column sums into ONE of the rows and the caller sums the rows.  (All workgroups adding
 * into one 256-byte row serialise in the L2: a third of the teams-of-4 kernel's time at a million agent rows.) */
int cm_masked_agg_backward_r(int32_t S, int32_t N, int32_t E, const float *attn, const float *dist_adj,
                             const float *chan, int64_t ch_stride, const float *hw, const float *out, const float *out_minus,
                             const float *d_out, float *d_attn, float *d_hw, float *d_bias, int32_t bias_replicas, void *stream);

/* Attention scores + softmax for the autograd path (attention_module.py:39-49):
 *   m[s,i,:] = softmax_j(q[s,i,:] . e[s,j,:]),  q = linear_in(e) [S,N,E], e [S,N,E], m [S,N,N].
 * backward: d_q, d_e [S,N,E] from d_m [S,N,N] (the d_e returned is only the key-side term) plus, when given, the
 * other gradients that flow into E: d_e = key-side term + d_e_add0 + d_e_add1 (each [S,N,E] or NULL; must not alias d_e). */
int cm_attention_forward(int32_t S, int32_t N, int32_t E, const float *q, const float *e, float *m, void *stream);
int cm_attention_backward(int32_t S, int32_t N, int32_t E, const float *q, const float *e, const float *m,
                          const float *d_m, const float *d_e_add0, const float *d_e_add1, float *d_q, float *d_e, void *stream);

/* Weight gradient of a per-agent dense layer over R rows: c[p][q] += sum_r a[r][p] * b[r][q] (c [P,Q] must be
 * zeroed by the caller; accumulated with float atomics), colsum_a[p] += sum_r a[r][p] (or NULL).
 * nn.Linear backward: a = dY [R,out], b = X [R,in] -> c = dW [out,in], colsum_a = db.
 * GraphConvolutionModule (H.W, graph_conv_module.py:63): a = H [R,in], b = dZ [R,out] -> c = dW [in,out]. */
int cm_linear_wgrad(int64_t R, int32_t P, int32_t Q, const float *a, const float *b, float *c, float *colsum_a,
                    void *stream);

/* One dense per-agent layer of the PPO update, one HBM pass each way (csrc/cm_linear.hip).
 * w_layout 0: w is nn.Linear's [out,in] (multi_headed_mlp_module.py:134-149, attention_module.py:36);
 * w_layout 1: w is GraphConvolutionModule's [in,out] (graph_conv_module.py:63).  1 <= in, out <= 128.
 *   forward :  y[r][o] = act(bias[o] + sum_k x[r][k] w(k,o)),  act 0 = identity, 1 = tanh; bias may be NULL.
 *   backward:  dz = (dy + dy2) * (1 - y^2) when y != NULL (tanh layer), dz = dy + dy2 when y == NULL; dy2 [R,out] or NULL
 *              is a second gradient flowing into the same output (saves the caller's accumulation pass);
 *              dx[r][k] = sum_o dz[r][o] w(k,o)   (dx may be NULL: first layer);
 *              dw += dz^T.x in w's own layout and db += colsum(dz) (db may be NULL), float atomics:
 *              the caller zeroes dw / db. */
int cm_linear_act_forward(int64_t R, int32_t in_dim, int32_t out_dim, const float *x, const float *w, int32_t w_layout,
                          const float *bias, int32_t act, float *y, void *stream);
int cm_linear_act_backward(int64_t R, int32_t in_dim, int32_t out_dim, const float *x, const float *w, int32_t w_layout,
                           const float *dy, const float *dy2, const float *y, float *dx, float *dw, float *db, void *stream);

/* Backward of the two-layer observation encoder (mlp_encoder_module: obs [R,d] -> a1 = tanh(W1 obs + b1) [R,128] ->
 * e = tanh(W2 a1 + b2) [R,64]; comm_base_net.py:80-84) in ONE pass over the saved activations: dz2 = (dy + dy2) * (1 - e^2),
 * dw2 += dz2^T a1, db2 += colsum(dz2); the gradient wrt a1 stays in the workgroup (times tanh' it overwrites the a1 tile) and
 * gives dw1 += dz1^T obs [128,d], db1 += colsum(dz1).  dy2 / db2 / db1 may be NULL; the caller zeroes dw* / db*.
 * Returns 1 - nothing done - when the shape is not covered (d > 64, tensors not 16-byte aligned): run the two layers with
 * cm_linear_act_backward instead. */
int cm_encoder_backward(int64_t R, int32_t d, const float *obs, const float *a1, const float *e, const float *w2, const float *dy,
                        const float *dy2, float *dw2, float *db2, float *dw1, float *db1, void *stream);

/* n tensors in one launch: src[k] is [rows[k], cols[k]] row-major f32; dst[k] receives its transpose ([cols, rows]) when
 * transpose[k] != 0, else a plain copy.  The veneer refreshes a net's flat [in,out] weight copy with it (one launch instead of
 * one framework copy per tensor).  n <= 40. */
int cm_multi_copy_t(int32_t n, const float *const *src, float *const *dst, const int32_t *rows, const int32_t *cols,
                    const int32_t *transpose, void *stream);
/* Multi-tensor Adam step with optional gradient-norm clip, two launches for a whole net (csrc/cm_ppo.hip): the vendored
 * torch-1.9 Adam of the reference (com_marl/torch/algos/my_optimizer/_functional.py:72-98, no weight decay / amsgrad) preceded
 * - when norm_ws != NULL - by torch.nn.utils.clip_grad_norm_(params, max_norm) (centralized_ma_ppo.py:253-255), the clipped
 * gradient written back.  params / grads / exp_avg / exp_avg_sq: HOST arrays of n <= 40 DEVICE pointers, sizes[k] elements
 * each; norm_ws: DEVICE workspace of CM_ADAM_NORM_FLOATS floats, no initial contents required: [0] receives |g|^2 before
 * the clip, the rest holds per-workgroup partial sums added in index order - the norm, and with it the step, is
 * bit-reproducible (data-parallel replicas stay identical).  step = 1, 2, ... (bias correction). */
#define CM_ADAM_NORM_FLOATS 65
int cm_multi_adam_step(int32_t n, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                       const int64_t *sizes, float *norm_ws, float max_norm, float lr, float beta1,
                       float beta2, float eps, int32_t step, void *stream);
/* The same step for a captured hipGraph that is replayed for SUCCESSIVE optimiser steps (the 10 mini-epochs over one minibatch,
 * centralized_ma_ppo.py:211-268, are the same launches on the same buffers): the bias-correction factors of step `first + *cursor`
 * are read from a DEVICE table [table_steps][2] = (1 - beta1^t, sqrt(1 - beta2^t)) indexed by the DEVICE counter `cursor` (the caller
 * advances it between steps; clamped to the table).  cm_adam_bias_corrections fills a HOST table with exactly the factors
 * cm_multi_adam_step computes for steps first_step .. first_step + n_steps - 1 (same f64 -> f32 arithmetic: the replayed steps
 * are bit-identical to eager ones). */
void cm_adam_bias_corrections(float beta1, float beta2, int32_t first_step, int32_t n_steps, float *table);
int cm_multi_adam_step_dev(int32_t n, float *const *params, float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                           const int64_t *sizes, float *norm_ws, float max_norm, float lr, float beta1, float beta2, float eps,
                           const float *bc_table, const int32_t *cursor, int32_t table_steps, void *stream);

/* The critic's loss (comm_base_critic.py:59-89: -Normal(values, std.mean()).log_prob(returns).mean(); values = the per-agent
 * outputs summed over the team, :110-112; std = exp(clamp(log_std, min)) of gaussian_mlp_module.py:62-188) in one launch, its
 * gradient in one more.  per_agent [S,N], returns [S], log_std: the DEVICE scalar parameter; has_min = 0 ignores min_log_std.
 *   forward : out[0] = loss, out[1] = mean_s (returns - values)^2 (read by the backward);  ws: CM_GAUSS_WS_BYTES of DEVICE memory,
 *             zero before the first use - the launch leaves it zero again (f64 sum of squares + a block counter)
 *   backward: d_per_agent [S,N] and d_log_std [1] (may be NULL), both times the DEVICE scalar *g (NULL = 1). */
#define CM_GAUSS_WS_BYTES 16
int cm_gauss_nll_forward(int64_t S, int32_t N, const float *per_agent, const float *returns, const float *log_std, float min_log_std,
                         int32_t has_min, float *out, void *ws, void *stream);
int cm_gauss_nll_backward(int64_t S, int32_t N, const float *per_agent, const float *returns, const float *log_std, float min_log_std,
                          int32_t has_min, const float *out, const float *g, float *d_per_agent, float *d_log_std, void *stream);

/* tensor_utils.discount_cumsum (garage/misc/tensor_utils.py:7-23) per path over a padded
 * [P,T] batch: f64 recurrence, f32 result, zero past lens[p]. */
int cm_discount_returns(int32_t P, int32_t T, const double *rewards, const int32_t *lens, double gamma,
                        float *returns, void *stream);
/* compute_advantages (garage/torch/algos/_utils.py:56-113) + per-path normalisation over the
 * valid steps (centralized_ma_ppo.py:422-426); normalize=0 skips the second part. */
int cm_gae(int32_t P, int32_t T, const float *rewards, const float *baselines, const int32_t *lens, float gamma,
           float lam, int32_t normalize, float eps, float *adv, void *stream);

/* PPO clipped-surrogate loss with entropy bonus over a padded [P,T] batch and its gradient wrt the policy logits
 * [P*T,N,A] (centralized_ma_ppo.py:390-438 _compute_loss, :540-589 _compute_objective; Categorical(probs=...) as built by
 * comm_categorical_mlp_policy.py:48-96, entropy / log_prob :121-137):
 *   *total = - sum over valid steps of [ min(r adv, clamp(r, 1 - clip, 1 + clip) adv) + ent_coeff * mean_i H_i ]   (f64)
 *   *count = number of valid steps (t < lens[p]);  dlogits (nullable) = d total / d logits, zero on padded steps.
 * add_entropy = 0 drops the entropy term (entropy_method != "regularized"). */
int cm_ppo_surrogate(int32_t P, int32_t T, int32_t N, int32_t A, const float *logits, const int32_t *actions,
                     const float *old_ll, const float *adv, const int32_t *lens, float clip, float ent_coeff,
                     int32_t add_entropy, double *total, int64_t *count, float *dlogits, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COMMARL_H */
