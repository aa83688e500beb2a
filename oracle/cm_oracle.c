/*
 * cm_oracle.c - sequential CPU restatement of the Com-MARL rollout hot path.
 * TEST INFRASTRUCTURE (see cm_oracle.h).  Every function cites the reference
 * file:line (relative to /root/reference) whose behaviour it restates.  The env code
 * deliberately keeps the reference's *sequential* order (agent by agent, prey by prey)
 * and a character grid, so that it is an independent check of the parallel
 * reformulation used by the HIP kernels.
 */
#include "cm_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXS 64           /* max grid side incl. walls */
#define MAXN 256          /* max agents */
#define C_EMPTY 0
#define C_AGENT 1
#define C_PREY 2
#define C_WALL 3

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11) - production RNG stream shared with the HIP path.
 * counter = (global env id, rng_step, site, idx), key = (seed lo, seed hi).
 * ---------------------------------------------------------------------------------- */
void cmo_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

enum { SITE_SPAWN = 1, SITE_PREY = 2, SITE_IID_STEP = 3, SITE_IID_RESET = 4, SITE_GE_STEP = 5, SITE_GE_RESET = 6,
       SITE_ACTION = 7, SITE_GE_INIT = 8 };

static inline float u32_to_unit_f32(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

typedef struct rng_ctx {
    const cmo_cfg *cfg;
    const cmo_tape *tape;
    int b;                   /* local env index */
    uint32_t env_gid, rng_step, key[2];
    int spawn_cursor;
    int err;
} rng_ctx;

static void philox_at(const rng_ctx *c, uint32_t site, uint32_t idx, uint32_t out[4]) {
    uint32_t ctr[4] = { c->env_gid, c->rng_step, site, idx };
    cmo_philox4x32_10(ctr, c->key, out);
}

/* one (row, col) spawn candidate: random.randint(lo, hi) twice (predator_prey.py:156,165; coverage.py:185) */
static void draw_spawn(rng_ctx *c, int lo, int hi, int *r, int *col) {
    if (c->cfg->rng_mode == CMO_RNG_TAPE) {
        if (c->spawn_cursor >= c->tape->spawn_cap) { c->err = -2; *r = lo; *col = lo; return; }
        const int32_t *p = c->tape->spawn + ((size_t)c->b * c->tape->spawn_cap + c->spawn_cursor) * 2;
        if (p[0] < 0) { c->err = -2; *r = lo; *col = lo; return; }
        *r = p[0]; *col = p[1];
    } else {
        uint32_t x[4];
        philox_at(c, SITE_SPAWN, (uint32_t)c->spawn_cursor, x);
        uint32_t span = (uint32_t)(hi - lo + 1);
        *r = lo + (int)mulhi32(x[0], span);
        *col = lo + (int)mulhi32(x[1], span);
    }
    c->spawn_cursor++;
}

/* np.random.choice(5, 1, p=(.175,.175,.175,.175,.3)) outcome (predator_prey.py:401) */
static int draw_prey_move(rng_ctx *c, int j, int trial) {
    if (c->cfg->rng_mode == CMO_RNG_TAPE) {
        uint8_t v = c->tape->prey[((size_t)c->b * c->cfg->n_preys + j) * 5 + trial];
        if (v > 4) { c->err = -3; return 4; }
        return v;
    }
    uint32_t x[4];
    philox_at(c, SITE_PREY, (uint32_t)(j * 2 + (trial >> 2)), x);
    uint32_t v = x[trial & 3];
    /* floor(cdf * 2^32) for cdf = .175, .35, .525, .7 */
    return (v >= 751619276u) + (v >= 1503238553u) + (v >= 2254857830u) + (v >= 3006477107u);
}

static float draw_link_uniform(rng_ctx *c, int ge, int slot, int flat_idx) {
    if (c->cfg->rng_mode == CMO_RNG_TAPE) {
        const int N = c->cfg->n_agents, L = c->cfg->n_hops;
        if (!ge) return c->tape->iid_u[((size_t)c->b * 2 + slot) * L * N * N + flat_idx];
        return c->tape->ge_u[((size_t)c->b * 2 + slot) * L * 2 * N * N + flat_idx];
    }
    uint32_t x[4];
    uint32_t site = ge ? (slot ? SITE_GE_RESET : SITE_GE_STEP) : (slot ? SITE_IID_RESET : SITE_IID_STEP);
    philox_at(c, site, (uint32_t)flat_idx >> 2, x);
    return u32_to_unit_f32(x[flat_idx & 3]);
}

/* ------------------------------------------------------------------------------------
 * geometry helpers
 * ---------------------------------------------------------------------------------- */
static const int DR[5] = { 1, 0, -1, 0, 0 };   /* 0 down(+row) 1 left(-col) 2 up(-row) 3 right(+col) 4 noop */
static const int DC[5] = { 0, -1, 0, 1, 0 };   /* predator_prey.py:244-253, coverage.py:336-345 */

int cmo_obs_dim(const cmo_cfg *cfg) {
    int w = 2 * cfg->rsen + 1;
    if (cfg->scenario == CMO_PP) return 2 * w * w + 3;                 /* predator_prey.py:183-204 */
    return 3 * w * w + 2 + (cfg->add_clock ? 1 : 0);                   /* coverage.py:198-212 */
}

static int side(const cmo_cfg *cfg) { return cfg->scenario == CMO_PP ? cfg->grid : cfg->grid + 2; }

/* CO base grid: wall ring + fixed obstacle rectangles scaled by r = m/10 (coverage.py:44,69-80,165-168,482-500) */
static void co_base_grid(const cmo_cfg *cfg, uint8_t *g /* [S*S] */) {
    const int S = cfg->grid + 2, r = cfg->grid / 10;
    memset(g, C_EMPTY, (size_t)S * S);
    for (int i = 0; i < S; ++i) { g[i] = C_WALL; g[(S - 1) * S + i] = C_WALL; g[i * S] = C_WALL; g[i * S + S - 1] = C_WALL; }
    /* {row0, col0, rows, cols} */
    const int easy[2][4] = { { 2 * r + 1, 2 * r + 1, 6 * r, 1 * r }, { 3 * r + 1, 8 * r + 1, 4 * r, 2 * r } };
    const int hard[7][4] = { { 2 * r + 1, 2 * r + 1, 6 * r, 1 * r }, { 3 * r + 1, 8 * r + 1, 4 * r, 2 * r },
                             { 1, 2 * r + 1, 1 * r, 3 * r },         { 1, 7 * r + 1, 2 * r, 1 * r },
                             { 4 * r + 1, 4 * r + 1, 2 * r, 3 * r }, { 8 * r + 1, 5 * r + 1, 2 * r, 2 * r },
                             { 8 * r + 1, 8 * r + 1, 1 * r, 1 * r } };
    const int (*ob)[4] = cfg->obst_hard ? hard : easy;
    const int n_ob = cfg->obst_hard ? 7 : 2;
    for (int k = 0; k < n_ob; ++k)
        for (int i = 0; i < ob[k][2]; ++i)
            for (int j = 0; j < ob[k][3]; ++j) {
                int rr = ob[k][0] + i, cc = ob[k][1] + j;
                if (rr >= 0 && rr < S && cc >= 0 && cc < S) g[rr * S + cc] = C_WALL;
            }
}

/* n_empty_cells = #'0' cells after the first spawn (coverage.py:228-230) */
int cmo_n_empty_cells(const cmo_cfg *cfg) {
    if (cfg->scenario != CMO_CO) return 0;
    uint8_t g[MAXS * MAXS];
    const int S = cfg->grid + 2;
    co_base_grid(cfg, g);
    int n = 0;
    for (int i = 0; i < S * S; ++i) n += (g[i] == C_EMPTY);
    return n - cfg->n_agents;
}

/* ------------------------------------------------------------------------------------
 * one env's working set
 * ---------------------------------------------------------------------------------- */
typedef struct env_view {
    const cmo_cfg *cfg;
    int S, N, M;
    uint8_t grid[MAXS * MAXS];
    int32_t *apos, *ppos;
    uint8_t *alive;
    uint32_t *visited;
    int32_t *step_count, *total_capture, *success;
    uint8_t *ge_state;
    uint8_t *cond;           /* agent_condition of this env, or NULL */
} env_view;

/* visited bitmap: VW = ceil(S / 32) words per grid row; cell (r, c) = bit c & 31 of word r * VW + (c >> 5) - one word per row
 * up to map 30, two for maps 40 / 50 (the reference keeps a float map, coverage.py:44,165-168) */
#define VW(S) (((S) + 31) >> 5)
static int vis_get(const env_view *e, int r, int c) { return (int)((e->visited[r * VW(e->S) + (c >> 5)] >> (c & 31)) & 1u); }
static void vis_set(env_view *e, int r, int c) { e->visited[r * VW(e->S) + (c >> 5)] |= 1u << (c & 31); }

static int in_grid(const env_view *e, int r, int c) { return r >= 0 && r < e->S && c >= 0 && c < e->S; }
/* _is_cell_vacant (predator_prey.py:234-238, coverage.py:258-259) */
static int vacant(const env_view *e, int r, int c) { return in_grid(e, r, c) && e->grid[r * e->S + c] == C_EMPTY; }

static void view_init(env_view *e, const cmo_cfg *cfg, cmo_state *st, int b) {
    e->cfg = cfg; e->S = side(cfg); e->N = cfg->n_agents; e->M = cfg->scenario == CMO_PP ? cfg->n_preys : 0;
    e->apos = st->agent_pos + (size_t)b * e->N * 2;
    e->ppos = st->prey_pos ? st->prey_pos + (size_t)b * e->M * 2 : NULL;
    e->alive = st->prey_alive ? st->prey_alive + (size_t)b * e->M : NULL;
    e->visited = st->visited ? st->visited + (size_t)b * e->S * VW(e->S) : NULL;
    e->step_count = st->step_count + b;
    e->total_capture = st->total_capture ? st->total_capture + b : NULL;
    e->success = st->success + b;
    e->ge_state = st->ge_state ? st->ge_state + (size_t)b * e->N * e->N : NULL;
    e->cond = st->agent_cond ? st->agent_cond + (size_t)b * e->N : NULL;
}

/* rebuild the character grid (_full_obs) from positions: agents + live preys (+ walls) */
static void build_grid(env_view *e) {
    if (e->cfg->scenario == CMO_CO) co_base_grid(e->cfg, e->grid);
    else memset(e->grid, C_EMPTY, (size_t)e->S * e->S);
    for (int i = 0; i < e->N; ++i) e->grid[e->apos[2 * i] * e->S + e->apos[2 * i + 1]] = C_AGENT;
    for (int j = 0; j < e->M; ++j)
        if (e->alive[j]) e->grid[e->ppos[2 * j] * e->S + e->ppos[2 * j + 1]] = C_PREY;
}

/* _neighbour_agents count (predator_prey.py:309-329): probes D,U,R,L, each bounds-checked */
static int count_adjacent(const env_view *e, int r, int c, uint8_t kind) {
    int n = 0;
    if (in_grid(e, r + 1, c) && e->grid[(r + 1) * e->S + c] == kind) n++;
    if (in_grid(e, r - 1, c) && e->grid[(r - 1) * e->S + c] == kind) n++;
    if (in_grid(e, r, c + 1) && e->grid[r * e->S + c + 1] == kind) n++;
    if (in_grid(e, r, c - 1) && e->grid[r * e->S + c - 1] == kind) n++;
    return n;
}

/* ------------------------------------------------------------------------------------
 * communication model (custom_implement/env_communication.py)
 * ---------------------------------------------------------------------------------- */
static int rcom_effective(const cmo_cfg *cfg) {           /* :71-72  Rcom+1 >= maps -> fully connected */
    return (cfg->rcom + 1 >= cfg->grid) ? 0 : cfg->rcom;
}

/* get_graph (:218-243).  f32 cdist <= Rcom_th is exactly dx^2+dy^2 <= 2*Rcom^2 on integer coords (SURVEY A-3) */
static void comm_adjacency(const env_view *e, float *adj) {
    const int N = e->N, rc = rcom_effective(e->cfg);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            if (rc == 0) { adj[i * N + j] = 1.0f; continue; }
            int dr = e->apos[2 * i] - e->apos[2 * j], dc = e->apos[2 * i + 1] - e->apos[2 * j + 1];
            adj[i * N + j] = (dr * dr + dc * dc <= 2 * rc * rc) ? 1.0f : 0.0f;
        }
}

/* get_next_state_matrix body (gilbert_elliot_loss_model.py:136-146) */
void cmo_ge_transition(int n, const uint8_t *s, const float *u_gb, const float *u_bg, float pgb, float pbg,
                       uint8_t *s_next) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const float eye = (i == j) ? 1.0f : 0.0f;
            const int k = i * n + j;
            int e_gb = (u_gb[k] + eye) < pgb;
            int g_next = s[k] && !(s[k] && e_gb);
            int e_bg = (u_bg[k] + eye) < pbg;
            int b_next = (!s[k]) && e_bg;
            s_next[k] = (uint8_t)(g_next || b_next);
        }
}

/* update_communication_state (:91-157); slot 0 = called from step, slot 1 = called from reset */
static void comm_channels(env_view *e, rng_ctx *rc, int slot, float *ch) {
    const int N = e->N, L = e->cfg->n_hops, NN = N * N;
    switch (e->cfg->channel) {
    case CMO_CH_FC:                                                     /* :93-95 */
        for (int k = 0; k < L * NN; ++k) ch[k] = 1.0f;
        break;
    case CMO_CH_FL:                                                     /* :97-100 */
        for (int l = 0; l < L; ++l)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) ch[l * NN + i * N + j] = (i == j) ? 1.0f : 0.0f;
        break;
    case CMO_CH_IID:                                                    /* get_iid_channel :200-214 */
        for (int l = 0; l < L; ++l)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    int k = l * NN + i * N + j;
                    float u = draw_link_uniform(rc, 0, slot, k) + ((i == j) ? 1.0f : 0.0f);
                    ch[k] = (u >= e->cfg->ploss) ? 1.0f : 0.0f;
                }
        break;
    case CMO_CH_GE: {                                                   /* env_communication.py:106-157 */
        /* ge_flags bit 0 = loss_apply == 0: one transition per ENV STEP shared by all hops (:144-149; reset :108-123:
         * the initial state for every hop, no draws) instead of one per GCN hop (:151-156; reset :124-141: hop 0 = the
         * initial state, then L-1 transitions).  bits 1-2 = GE_INIT: good (ones), bad (zeros) or random
         * (get_init_state, gilbert_elliot_loss_model.py:84-87: rand(n,n) >= Pgb/(Pgb+Pbg), no identity added). */
        uint8_t cur[MAXN * MAXN / 4], nxt[MAXN * MAXN / 4];
        float ugb[MAXN * MAXN / 4], ubg[MAXN * MAXN / 4];
        const int per_step = e->cfg->ge_flags & 1, init_mode = (e->cfg->ge_flags >> 1) & 3;
        int l0 = 0, l1 = per_step ? 1 : L;
        if (slot == 1) {                                                /* step_count == 0 */
            if (init_mode == 2) {
                const float bad_rate = (float)((double)e->cfg->pgb / ((double)e->cfg->pgb + (double)e->cfg->pbg));
                for (int k = 0; k < NN; ++k) {
                    float u;
                    if (e->cfg->rng_mode == CMO_RNG_TAPE) u = rc->tape->ge_init_u[(size_t)rc->b * NN + k];
                    else { uint32_t x[4]; philox_at(rc, SITE_GE_INIT, (uint32_t)k >> 2, x); u = u32_to_unit_f32(x[k & 3]); }
                    cur[k] = (u >= bad_rate) ? 1 : 0;
                }
            } else memset(cur, init_mode == 1 ? 0 : 1, (size_t)NN);
            for (int k = 0; k < NN; ++k) ch[k] = (float)cur[k];
            l0 = 1;
            if (per_step) l1 = 1;                                       /* no transition at reset */
        } else {
            memcpy(cur, e->ge_state, (size_t)NN);
        }
        for (int l = l0; l < l1; ++l) {
            for (int k = 0; k < NN; ++k) {
                ugb[k] = draw_link_uniform(rc, 1, slot, (l * 2 + 0) * NN + k);
                ubg[k] = draw_link_uniform(rc, 1, slot, (l * 2 + 1) * NN + k);
            }
            cmo_ge_transition(N, cur, ugb, ubg, e->cfg->pgb, e->cfg->pbg, nxt);
            memcpy(cur, nxt, (size_t)NN);
            for (int k = 0; k < NN; ++k) ch[l * NN + k] = (float)cur[k];
        }
        if (per_step)                                                   /* .expand(GCNHops, ...): every hop sees the same state */
            for (int l = 1; l < L; ++l) for (int k = 0; k < NN; ++k) ch[l * NN + k] = (float)cur[k];
        memcpy(e->ge_state, cur, (size_t)NN);
        break;
    }
    }
}

/* ------------------------------------------------------------------------------------
 * observations
 * ---------------------------------------------------------------------------------- */
/* PredatorPrey.get_agent_obs (predator_prey.py:183-204) + get_neighbors (:173-181) */
static void pp_obs(const env_view *e, float *obs) {
    const int R = e->cfg->rsen, W = 2 * R + 1, d = cmo_obs_dim(e->cfg), G = e->S;
    for (int i = 0; i < e->N; ++i) {
        float *o = obs + (size_t)i * d;
        const int r0 = e->apos[2 * i], c0 = e->apos[2 * i + 1];
        for (int k = 0; k < 2 * W * W; ++k) o[k] = 0.0f;
        for (int row = r0 - R; row <= r0 + R; ++row)
            for (int col = c0 - R; col <= c0 + R; ++col) {
                if (!in_grid(e, row, col)) continue;
                uint8_t g = e->grid[row * G + col];
                int k = (row - (r0 - R)) * W + (col - (c0 - R));
                if (g == C_AGENT) o[k] = 1.0f;
                if (g == C_PREY) o[W * W + k] = 1.0f;
            }
        o[2 * W * W + 0] = (float)((double)r0 / (double)G);            /* :195 row / G     */
        o[2 * W * W + 1] = (float)((double)c0 / (double)(G - 1));      /* :195 col / (G-1) */
        o[2 * W * W + 2] = (float)((double)*e->step_count / (double)e->cfg->max_steps);   /* :196 */
    }
}

/* Python round(x, 2): correctly rounded decimal, half-even on the exact binary value */
static double py_round2(double x) {
    char buf[64];
    snprintf(buf, sizeof buf, "%.2f", x);
    return strtod(buf, NULL);
}

/* Coverage.get_agent_obs (coverage.py:198-212) + get_local_view (:448-480) */
static void co_obs(const env_view *e, float *obs) {
    const int R = e->cfg->rsen, W = 2 * R + 1, d = cmo_obs_dim(e->cfg), S = e->S;
    for (int i = 0; i < e->N; ++i) {
        float *o = obs + (size_t)i * d;
        const int r0 = e->apos[2 * i], c0 = e->apos[2 * i + 1];
        for (int k = 0; k < 3 * W * W; ++k) o[k] = 0.0f;
        for (int row = r0 - R; row <= r0 + R; ++row)
            for (int col = c0 - R; col <= c0 + R; ++col) {
                int k = (row - (r0 - R)) * W + (col - (c0 - R));
                if (!in_grid(e, row, col)) { o[k] = 1.0f; continue; }             /* out of grid = wall */
                uint8_t g = e->grid[row * S + col];
                if (g == C_WALL) o[k] = 1.0f;
                if (g == C_AGENT) o[W * W + k] = 1.0f;
                if (vis_get(e, row, col)) o[2 * W * W + k] = 1.0f;
            }
        o[3 * W * W + 0] = (float)py_round2((double)r0 / (double)(S - 1));        /* :206 */
        o[3 * W * W + 1] = (float)py_round2((double)c0 / (double)(S - 1));
        if (e->cfg->add_clock) o[3 * W * W + 2] = (float)((double)*e->step_count / (double)e->cfg->max_steps);
    }
}

/* ------------------------------------------------------------------------------------
 * reset
 * ---------------------------------------------------------------------------------- */
/* PredatorPrey.reset (:206-232) + __init_full_obs (:150-171) */
static void pp_reset(env_view *e, rng_ctx *rc) {
    const int G = e->S;
    memset(e->grid, C_EMPTY, (size_t)G * G);
    for (int i = 0; i < e->N && !rc->err; ++i)
        for (;;) {
            int r, c;
            draw_spawn(rc, 0, G - 1, &r, &c);
            if (rc->err) break;
            if (vacant(e, r, c)) { e->apos[2 * i] = r; e->apos[2 * i + 1] = c; e->grid[r * G + c] = C_AGENT; break; }
        }
    for (int j = 0; j < e->M && !rc->err; ++j)
        for (;;) {
            int r, c;
            draw_spawn(rc, 0, G - 1, &r, &c);
            if (rc->err) break;
            if (vacant(e, r, c) && count_adjacent(e, r, c, C_AGENT) == 0) {
                e->ppos[2 * j] = r; e->ppos[2 * j + 1] = c; e->grid[r * G + c] = C_PREY; break;
            }
        }
    *e->step_count = 0;
    for (int j = 0; j < e->M; ++j) e->alive[j] = 1;
    if (e->cond) memset(e->cond, 1, (size_t)e->N);                    /* __init_full_obs :152 */
}

/* Coverage.reset (:221-246) + __init_full_obs (:172-196); team split only colours cells */
static void co_reset(env_view *e, rng_ctx *rc) {
    const int S = e->S, m = e->cfg->grid;
    co_base_grid(e->cfg, e->grid);
    for (int r = 0; r < S * VW(S); ++r) e->visited[r] = 0;
    for (int i = 0; i < e->N && !rc->err; ++i)
        for (;;) {
            int r, c;
            draw_spawn(rc, 1, m, &r, &c);
            if (rc->err) break;
            if (vacant(e, r, c)) {
                e->apos[2 * i] = r; e->apos[2 * i + 1] = c; e->grid[r * S + c] = C_AGENT;
                vis_set(e, r, c);
                break;
            }
        }
    *e->total_capture = 0;
    *e->step_count = 0;
}

static void emit_obs_comm(env_view *e, rng_ctx *rc, int slot, const cmo_cfg *cfg, cmo_out *out, int b) {
    const int N = e->N, d = cmo_obs_dim(cfg), L = cfg->n_hops;
    if (cfg->scenario == CMO_PP) pp_obs(e, out->obs + (size_t)b * N * d);
    else co_obs(e, out->obs + (size_t)b * N * d);
    comm_adjacency(e, out->dist_adj + (size_t)b * N * N);
    comm_channels(e, rc, slot, out->channels + (size_t)b * L * N * N);
}

static void rng_begin(rng_ctx *rc, const cmo_cfg *cfg, const cmo_tape *tape, cmo_state *st, int b) {
    rc->cfg = cfg; rc->tape = tape; rc->b = b; rc->err = 0; rc->spawn_cursor = 0;
    rc->env_gid = (uint32_t)(cfg->env_id_offset + b);
    rc->rng_step = st->rng_step[b];
    rc->key[0] = (uint32_t)cfg->seed; rc->key[1] = (uint32_t)(cfg->seed >> 32);
    st->rng_step[b] += 1;
}

int cmo_reset(const cmo_cfg *cfg, cmo_state *st, const cmo_tape *tape, cmo_out *out) {
    int status = 0;
    for (int b = 0; b < cfg->n_envs; ++b) {
        env_view e; rng_ctx rc;
        view_init(&e, cfg, st, b);
        rng_begin(&rc, cfg, tape, st, b);
        if (cfg->scenario == CMO_PP) pp_reset(&e, &rc); else co_reset(&e, &rc);
        emit_obs_comm(&e, &rc, 1, cfg, out, b);
        if (rc.err) status = rc.err;
    }
    return status;
}

/* ------------------------------------------------------------------------------------
 * step
 * ---------------------------------------------------------------------------------- */
/* PredatorPrey.step (:494-519) + __update_agent_pos (:240-261) + reward_default (:409-450)
 * + reward_individual (:452-492) + prey_random_move (:396-407) + __update_prey_pos (:276-301) */
static int pp_step(env_view *e, rng_ctx *rc, const int32_t *act, double *reward, int32_t *details,
                   uint8_t *alive_info) {
    const cmo_cfg *cfg = e->cfg;
    const int G = e->S, N = e->N, M = e->M;
    if (cfg->load < 2 || cfg->load > 4) return -5;                    /* capv undefined (App. B-6) */
    *e->step_count += 1;
    int moving = 0;
    for (int i = 0; i < N; ++i) {
        int a = act[i];
        if (a < 0 || a > 4) return -4;
        if (a != 4) {
            moving++;
            int r = e->apos[2 * i], c = e->apos[2 * i + 1], nr = r + DR[a], nc = c + DC[a];
            /* :257-261: a vacant target moves the agent only if agent_condition[i] != 0 (otherwise its cell is
             * cleared and re-marked by __update_agent_view: the agent stays where it is) */
            if (vacant(e, nr, nc) && (!e->cond || e->cond[i] != 0)) {
                e->grid[r * G + c] = C_EMPTY; e->grid[nr * G + nc] = C_AGENT;
                e->apos[2 * i] = nr; e->apos[2 * i + 1] = nc;
            }
        }
    }
    int capture = 0, penalty = 0;
    uint8_t watching[MAXN];
    memset(watching, 0, (size_t)N);
    for (int j = 0; j < M; ++j) {
        if (!e->alive[j]) continue;
        const int r = e->ppos[2 * j], c = e->ppos[2 * j + 1];
        const int n_ag = count_adjacent(e, r, c, C_AGENT);
        /* prey_watching: ids of the adjacent agents */
        for (int i = 0; i < N; ++i) {
            int dr = e->apos[2 * i] - r, dc = e->apos[2 * i + 1] - c;
            if ((dr == 0 && (dc == 1 || dc == -1)) || (dc == 0 && (dr == 1 || dr == -1))) watching[i] = 1;
        }
        if (n_ag >= 1) {
            int need;
            if (cfg->load == 2) need = cfg->load;                       /* :425 */
            else {                                                      /* :469-470, __create_edges :123-144 */
                const int on_r = (r == 0 || r == G - 1), on_c = (c == 0 || c == G - 1);
                const int adj = (on_r && on_c) ? 2 : ((on_r || on_c) ? 3 : cfg->load);
                const int avail = adj - count_adjacent(e, r, c, C_PREY);
                need = cfg->load < avail ? cfg->load : avail;
            }
            if (need <= n_ag) { capture++; e->alive[j] = 0; }
            else penalty++;
        }
        /* prey_random_move */
        int mv = -1;
        if (e->alive[j]) {
            for (int t = 0; t < 5; ++t) {
                int m = draw_prey_move(rc, j, t);
                if (count_adjacent(e, r + DR[m], c + DC[m], C_AGENT) == 0) { mv = m; break; }
            }
            if (mv < 0) mv = 4;
            if (mv != 4 && vacant(e, r + DR[mv], c + DC[mv])) {
                e->grid[r * G + c] = C_EMPTY;
                e->ppos[2 * j] = r + DR[mv]; e->ppos[2 * j + 1] = c + DC[mv];
                e->grid[e->ppos[2 * j] * G + e->ppos[2 * j + 1]] = C_PREY;
            }
        } else {
            e->grid[r * G + c] = C_EMPTY;                               /* :301 */
        }
    }
    double rew = (cfg->step_cost + cfg->capture_reward * capture) + (cfg->move_cost * moving) / (double)N;
    if (cfg->load == 2) rew = rew + cfg->penalty * penalty;             /* :434 vs :480 */
    *reward = rew;
    int wsum = 0;
    for (int i = 0; i < N; ++i) wsum += watching[i];
    details[0] = capture; details[1] = moving; details[2] = penalty; details[3] = 0; details[4] = wsum; details[5] = 0;
    int any_alive = 0;
    for (int j = 0; j < M; ++j) { any_alive |= e->alive[j]; if (alive_info) alive_info[j] = e->alive[j]; }
    int done = (*e->step_count >= cfg->max_steps) || !any_alive;       /* :511-517 */
    if (done) *e->success = any_alive ? 0 : 1;
    return done;
}

/* Coverage.step (:319-401) + get_reward (:299-317) */
static int co_step(env_view *e, const int32_t *act, int n_empty, double *reward, int32_t *details) {
    const cmo_cfg *cfg = e->cfg;
    const int S = e->S, N = e->N;
    *e->step_count += 1;
    int cap = 0, mov = 0, pen = 0, lazy = 0, rev = 0;
    for (int i = 0; i < N; ++i) {
        int a = act[i];
        if (a < 0 || a > 4) return -4;
        if (a == 4) { lazy++; continue; }
        mov++;
        int r = e->apos[2 * i], c = e->apos[2 * i + 1], nr = r + DR[a], nc = c + DC[a];
        if (vacant(e, nr, nc)) {
            e->apos[2 * i] = nr; e->apos[2 * i + 1] = nc;
            if (!vis_get(e, nr, nc)) { vis_set(e, nr, nc); cap++; }
            else rev++;
            e->grid[r * S + c] = C_EMPTY; e->grid[nr * S + nc] = C_AGENT;
        } else pen++;
    }
    *e->total_capture += cap;
    int done = 0;
    double fin = 0.0;
    if (*e->total_capture == n_empty) { fin = cfg->final_reward; done = 1; }    /* :381-385 */
    if (*e->step_count >= cfg->max_steps) { *e->success = done ? 1 : 0; done = 1; }  /* :388-393 */
    const double n = (double)N;
    double rew = cfg->step_cost + cfg->capture_reward * ((double)cap / n);
    rew = rew + cfg->move_cost * ((double)mov / n);
    rew = rew + cfg->penalty * ((double)pen / n);
    rew = rew + cfg->lazy_penalty * ((double)lazy / n);
    rew = rew + cfg->revisit_penalty * ((double)rev / n);
    rew = rew + fin;
    *reward = rew;
    details[0] = cap; details[1] = mov; details[2] = pen; details[3] = lazy; details[4] = rev; details[5] = fin != 0.0;
    return done;
}

/* env.step + VecEnvExecutor.step (garage/sampler/vec_env_executor.py:19-45): truncate at
 * max_path_length, auto-reset and substitute the reset observation */
int cmo_step(const cmo_cfg *cfg, cmo_state *st, const int32_t *actions, const cmo_tape *tape, cmo_out *out,
             int n_threads) {
    int status = 0;
    const int n_empty = cmo_n_empty_cells(cfg);
#pragma omp parallel for num_threads(n_threads > 0 ? n_threads : 1) schedule(static)
    for (int b = 0; b < cfg->n_envs; ++b) {
        env_view e; rng_ctx rc;
        view_init(&e, cfg, st, b);
        rng_begin(&rc, cfg, tape, st, b);
        build_grid(&e);
        const int32_t *act = actions + (size_t)b * e.N;
        int32_t *det = out->details + (size_t)b * 6;
        int done;
        if (cfg->scenario == CMO_PP)
            done = pp_step(&e, &rc, act, out->reward + b, det,
                           out->prey_alive_info ? out->prey_alive_info + (size_t)b * e.M : NULL);
        else
            done = co_step(&e, act, n_empty, out->reward + b, det);
        if (done < 0) {
#pragma omp critical
            status = done;
            continue;
        }
        if (*e.step_count >= cfg->max_path_length) done = 1;           /* ts >= max_path_length */
        out->done[b] = (uint8_t)done;
        if (done) {
            if (cfg->scenario == CMO_PP) pp_reset(&e, &rc); else co_reset(&e, &rc);
            emit_obs_comm(&e, &rc, 1, cfg, out, b);
        } else {
            emit_obs_comm(&e, &rc, 0, cfg, out, b);
        }
        if (rc.err) {
#pragma omp critical
            status = rc.err;
        }
    }
    return status;
}

/* ------------------------------------------------------------------------------------
 * Comm-DP policy / critic forward (rollout form, one sample = one env state)
 * ---------------------------------------------------------------------------------- */
/* y[n,out] = act(x[n,in] . W[out,in]^T + b) - nn.Linear */
static void linear(int n, int in, int out, const float *x, const float *W, const float *b, int do_tanh, float *y) {
    for (int i = 0; i < n; ++i)
        for (int o = 0; o < out; ++o) {
            float acc = b ? b[o] : 0.0f;
            const float *xi = x + (size_t)i * in, *wo = W + (size_t)o * in;
            for (int k = 0; k < in; ++k) acc += xi[k] * wo[k];
            y[(size_t)i * out + o] = do_tanh ? tanhf(acc) : acc;
        }
}

/* CommBaseNet.forward (comm_base_net.py:80-108): encoder -> attention -> L x (mask, renorm, GCN).
 * emb: [ (L+1), N, E ] ; attn: [N,N] */
static void comm_trunk(int d, int N, int L, int EH, int E, const float *enc_w1, const float *enc_b1,
                       const float *enc_w2, const float *enc_b2, const float *attn_w, const float *gcn_w,
                       const float *gcn_b, const float *obs, const float *adj, const float *ch, float *emb,
                       float *attn, float *scratch) {
    float *h1 = scratch;                          /* [N,EH] */
    float *q = h1 + (size_t)N * EH;               /* [N,E]  */
    float *hw = q + (size_t)N * E;                /* [N,E]  */
    float *A = hw + (size_t)N * E;                /* [N,N]  */
    linear(N, d, EH, obs, enc_w1, enc_b1, 1, h1);                        /* multi_headed_mlp_module.py:145-149 */
    linear(N, EH, E, h1, enc_w2, enc_b2, 1, emb);                        /* output_nonlinearity = tanh :53 */
    linear(N, E, E, emb, attn_w, NULL, 0, q);                            /* attention_module.py:39-41 */
    for (int i = 0; i < N; ++i) {                                        /* :42-49 scores + softmax */
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
            float s = 0.0f;
            for (int k = 0; k < E; ++k) s += q[(size_t)i * E + k] * emb[(size_t)j * E + k];
            attn[i * N + j] = s;
            if (s > mx) mx = s;
        }
        float sum = 0.0f;
        for (int j = 0; j < N; ++j) { attn[i * N + j] = expf(attn[i * N + j] - mx); sum += attn[i * N + j]; }
        for (int j = 0; j < N; ++j) attn[i * N + j] /= sum;
    }
    for (int l = 0; l < L; ++l) {
        const float *H = emb + (size_t)l * N * E;
        float *Hn = emb + (size_t)(l + 1) * N * E;
        const float *Wg = gcn_w + (size_t)l * E * E, *bg = gcn_b ? gcn_b + (size_t)l * E : NULL;
        const float *C = ch + (size_t)l * N * N;
        for (int i = 0; i < N; ++i) {                                    /* comm_base_net.py:101-103 */
            float rs = 0.0f;
            for (int j = 0; j < N; ++j) { A[i * N + j] = attn[i * N + j] * adj[i * N + j] * C[i * N + j]; rs += A[i * N + j]; }
            for (int j = 0; j < N; ++j) A[i * N + j] = A[i * N + j] / (rs + 1e-12f);
        }
        for (int i = 0; i < N; ++i)                                      /* graph_conv_module.py:63 H.W ([in,out]) */
            for (int o = 0; o < E; ++o) {
                float acc = 0.0f;
                for (int k = 0; k < E; ++k) acc += H[(size_t)i * E + k] * Wg[(size_t)k * E + o];
                hw[(size_t)i * E + o] = acc;
            }
        for (int i = 0; i < N; ++i)                                      /* :65-70 A.(HW) + b, tanh */
            for (int o = 0; o < E; ++o) {
                float acc = 0.0f;
                for (int j = 0; j < N; ++j) acc += A[i * N + j] * hw[(size_t)j * E + o];
                Hn[(size_t)i * E + o] = tanhf(acc + (bg ? bg[o] : 0.0f));
            }
    }
}

/* CommCategoricalMLPPolicy.forward (comm_categorical_mlp_policy.py:48-96) */
void cmo_policy_forward(const cmo_policy_weights *w, int S, const float *obs, const float *avail,
                        const float *dist_adj, const float *channels, float *probs, float *attn, float *emb_out,
                        int n_threads) {
    const int N = w->n_agents, d = w->d, L = w->n_hops, E = w->emb, EH = w->enc_hidden, A = w->n_act;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        size_t nscr = (size_t)N * (EH + 2 * E + N) + (size_t)(L + 1) * N * E + (size_t)N * N
                      + (size_t)N * (E + w->h1 + w->h2 + w->h3 + A);
        float *scr = (float *)malloc(nscr * sizeof(float));
        float *emb = scr + (size_t)N * (EH + 2 * E + N);
        float *att = emb + (size_t)(L + 1) * N * E;
        float *x = att + (size_t)N * N, *a1 = x + (size_t)N * E, *a2 = a1 + (size_t)N * w->h1,
              *a3 = a2 + (size_t)N * w->h2, *lg = a3 + (size_t)N * w->h3;
#pragma omp for schedule(static)
        for (int s = 0; s < S; ++s) {
            comm_trunk(d, N, L, EH, E, w->enc_w1, w->enc_b1, w->enc_w2, w->enc_b2, w->attn_w, w->gcn_w, w->gcn_b,
                       obs + (size_t)s * N * d, dist_adj + (size_t)s * N * N, channels + (size_t)s * L * N * N, emb,
                       att, scr);
            for (int k = 0; k < N * E; ++k) x[k] = (w->no_residual ? 0.0f : emb[k]) + emb[(size_t)L * N * E + k];   /* :74-77 */
            linear(N, E, w->h1, x, w->hd_w1, w->hd_b1, 1, a1);
            linear(N, w->h1, w->h2, a1, w->hd_w2, w->hd_b2, 1, a2);
            linear(N, w->h2, w->h3, a2, w->hd_w3, w->hd_b3, 1, a3);
            linear(N, w->h3, A, a3, w->hd_w4, w->hd_b4, 0, lg);
            for (int i = 0; i < N; ++i) {                                /* softmax, x avail, renorm :81-91 */
                float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
                float *p = probs + ((size_t)s * N + i) * A;
                const float *av = avail + ((size_t)s * N + i) * A;
                for (int a = 0; a < A; ++a) if (lg[i * A + a] > mx) mx = lg[i * A + a];
                for (int a = 0; a < A; ++a) { p[a] = expf(lg[i * A + a] - mx); sum += p[a]; }
                for (int a = 0; a < A; ++a) { p[a] = (p[a] / sum) * av[a]; msum += p[a]; }
                for (int a = 0; a < A; ++a) p[a] = p[a] / msum;
            }
            if (attn) memcpy(attn + (size_t)s * N * N, att, sizeof(float) * N * N);
            if (emb_out) memcpy(emb_out + (size_t)s * (L + 1) * N * E, emb, sizeof(float) * (L + 1) * N * E);
        }
        free(scr);
    }
}

/* Row-wise MLP chain (garage MultiHeadedMLPModule.forward, multi_headed_mlp_module.py:134-149): layer l is
 * nn.Linear (weight [out,in]) followed by tanh when bit l of tanh_mask is set.  Used for the non-communicating
 * policies (dec_categorical_mlp_policy.py:106-122: encoder then head per agent row;
 * centralized_categorical_mlp_policy.py:61-97: one chain per env row) and GaussianMLPBaseline.forward
 * (gaussian_mlp_baseline.py:100-115).  y [rows, out_dim[n_layers-1]]. */
void cmo_mlp_forward(int rows, int in_dim, int n_layers, const int32_t *out_dim, int tanh_mask, const float *const *W,
                     const float *const *b, const float *x, float *y) {
    int maxw = in_dim;
    for (int l = 0; l < n_layers; ++l) if (out_dim[l] > maxw) maxw = out_dim[l];
    float *t0 = (float *)malloc(sizeof(float) * maxw), *t1 = (float *)malloc(sizeof(float) * maxw);
    for (int r = 0; r < rows; ++r) {
        memcpy(t0, x + (size_t)r * in_dim, sizeof(float) * in_dim);
        int in = in_dim;
        for (int l = 0; l < n_layers; ++l) {
            linear(1, in, out_dim[l], t0, W[l], b[l], (tanh_mask >> l) & 1, t1);
            float *t = t0; t0 = t1; t1 = t;
            in = out_dim[l];
        }
        memcpy(y + (size_t)r * in, t0, sizeof(float) * in);
    }
    free(t0); free(t1);
}

/* softmax per group of n_act logits, x avail, renormalise (dec_categorical_mlp_policy.py:113-121,
 * centralized_categorical_mlp_policy.py:83-97); logits / avail / probs [items, n_act]. */
void cmo_group_softmax(int items, int n_act, const float *logits, const float *avail, float *probs) {
    for (int i = 0; i < items; ++i) {
        const float *lg = logits + (size_t)i * n_act, *av = avail + (size_t)i * n_act;
        float *p = probs + (size_t)i * n_act;
        float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
        for (int a = 0; a < n_act; ++a) if (lg[a] > mx) mx = lg[a];
        for (int a = 0; a < n_act; ++a) { p[a] = expf(lg[a] - mx); sum += p[a]; }
        for (int a = 0; a < n_act; ++a) { p[a] = (p[a] / sum) * av[a]; msum += p[a]; }
        for (int a = 0; a < n_act; ++a) p[a] = p[a] / msum;
    }
}

/* inverse-CDF categorical sample; u from Philox(counter = (env, policy_step, 7, agent)).x0 */
void cmo_sample_actions(int S, int n_agents, int n_act, const float *probs, uint64_t seed, int env_id_offset,
                        uint32_t policy_step, int32_t *actions) {
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < n_agents; ++i) {
            uint32_t ctr[4] = { (uint32_t)(env_id_offset + s), policy_step, SITE_ACTION, (uint32_t)i }, x[4];
            cmo_philox4x32_10(ctr, key, x);
            const float u = u32_to_unit_f32(x[0]);
            const float *p = probs + ((size_t)s * n_agents + i) * n_act;
            float acc = 0.0f;
            int a_sel = -1, last = 0;
            for (int a = 0; a < n_act; ++a) {
                if (p[a] > 0.0f) last = a;
                acc += p[a];
                if (a_sel < 0 && u < acc) a_sel = a;
            }
            actions[(size_t)s * n_agents + i] = a_sel < 0 ? last : a_sel;
        }
}

/* CommBaseCritic.forward (comm_base_critic.py:91-114), aggregator 'sum' */
void cmo_critic_forward(const cmo_critic_weights *w, int S, const float *obs, const float *dist_adj,
                        const float *channels, float *values, int n_threads) {
    const int N = w->n_agents, d = w->d, L = w->n_hops, E = w->emb, EH = w->enc_hidden, DH = w->dec_hidden;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
    {
        size_t nscr = (size_t)N * (EH + 2 * E + N) + (size_t)(L + 1) * N * E + (size_t)N * N
                      + (size_t)N * (E + DH + 1);
        float *scr = (float *)malloc(nscr * sizeof(float));
        float *emb = scr + (size_t)N * (EH + 2 * E + N);
        float *att = emb + (size_t)(L + 1) * N * E;
        float *x = att + (size_t)N * N, *a1 = x + (size_t)N * E, *v = a1 + (size_t)N * DH;
#pragma omp for schedule(static)
        for (int s = 0; s < S; ++s) {
            comm_trunk(d, N, L, EH, E, w->enc_w1, w->enc_b1, w->enc_w2, w->enc_b2, w->attn_w, w->gcn_w, w->gcn_b,
                       obs + (size_t)s * N * d, dist_adj + (size_t)s * N * N, channels + (size_t)s * L * N * N, emb,
                       att, scr);
            for (int k = 0; k < N * E; ++k) x[k] = (w->no_residual ? 0.0f : emb[k]) + emb[(size_t)L * N * E + k];
            linear(N, E, DH, x, w->dec_w1, w->dec_b1, 1, a1);
            linear(N, DH, 1, a1, w->dec_w2, w->dec_b2, 0, v);
            float sum = 0.0f;
            for (int i = 0; i < N; ++i) sum += v[i];
            values[s] = sum;
        }
        free(scr);
    }
}

/* discount_cumsum (garage/misc/tensor_utils.py:7-23): scipy lfilter([1],[1,-g]) on the reversed
 * series == f64 recurrence y_t = x_t + g*y_{t+1}; the caller's torch.Tensor() cast gives f32 */
void cmo_discount_cumsum(int T, const double *x, double gamma, float *out) {
    double y = 0.0;
    for (int t = T - 1; t >= 0; --t) { y = x[t] + gamma * y; out[t] = (float)y; }
}

/* compute_advantages (garage/torch/algos/_utils.py:106-113): delta_t = r_t + g*V_{t+1} - V_t with
 * V past the padded end = 0; A_t = sum_k (g*lam)^k delta_{t+k}, the filter being an f32 cumprod */
void cmo_gae(int P, int T, const float *rewards, const float *baselines, float gamma, float lam, float *adv) {
    float *filt = (float *)malloc(sizeof(float) * (size_t)T);
    float *delta = (float *)malloc(sizeof(float) * (size_t)T);
    const float gl = gamma * lam;
    filt[0] = 1.0f;
    for (int k = 1; k < T; ++k) filt[k] = filt[k - 1] * gl;
    for (int p = 0; p < P; ++p) {
        const float *r = rewards + (size_t)p * T, *v = baselines + (size_t)p * T;
        for (int t = 0; t < T; ++t) delta[t] = (r[t] + gamma * (t + 1 < T ? v[t + 1] : 0.0f)) - v[t];
        for (int t = 0; t < T; ++t) {
            float a = 0.0f;
            for (int k = 0; t + k < T; ++k) a += delta[t + k] * filt[k];
            adv[(size_t)p * T + t] = a;
        }
    }
    free(filt); free(delta);
}
