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

namespace cm {

constexpr int C_EMPTY = 0, C_AGENT = 1, C_PREY = 2, C_WALL = 3;
constexpr int WAVE = 64;

__device__ __forceinline__ int dr_of(int a) { return a == 0 ? 1 : (a == 2 ? -1 : 0); }   // predator_prey.py:244-253
__device__ __forceinline__ int dc_of(int a) { return a == 1 ? -1 : (a == 3 ? 1 : 0); }
__device__ __forceinline__ bool in_grid(int r, int c, int S) { return (unsigned)r < (unsigned)S && (unsigned)c < (unsigned)S; }
// (cell / count_adj are defined after the LDS accessors)
// _neighbour_agents / _neighbour_preys count (predator_prey.py:309-351): D,U,R,L, each bounds-checked
__device__ __forceinline__ void raise(const EnvDev &p, int code) { atomicCAS(p.status, 0, code); }

// Dynamic LDS of the single-wave workgroup.  Everything is addressed as smem + integer offset so that
// the compiler keeps the accesses in the LDS address space (ds_read/ds_write), never as flat pointers.
extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

struct Lds {           // byte offsets into smem (kept in registers: always passed by value)
    int g;             // [S*S] u8 occupancy tile
    int ar, ac;        // [N] i16
    int pr, pc;        // [M] i16
    int act;           // [N] u8
    int alive;         // [M] u8
    int pcnt;          // [M] u8 predator count around prey j
    int pmv;           // [M] u8 chosen prey move | 8 = tape ran out
    int vis;           // [S] u32
#ifdef CM_BOUNDS
    int nS2, nN, nM, nS;
    int32_t *status;
#endif
};

__host__ __device__ inline int lds_take(int &off, int bytes) { const int o = off; off += (bytes + 15) & ~15; return o; }
__host__ __device__ inline int lds_env_bytes(int S, int N, int M) {
    int off = 0;
    lds_take(off, S * S); lds_take(off, 2 * N); lds_take(off, 2 * N); lds_take(off, 2 * M); lds_take(off, 2 * M);
    lds_take(off, N); lds_take(off, M); lds_take(off, M); lds_take(off, M); lds_take(off, 4 * S);
    return off;
}
__device__ __forceinline__ Lds make_lds(int S, int N, int M, int base, int32_t *status) {
    int off = base;
    Lds l;
    l.g = lds_take(off, S * S); l.ar = lds_take(off, 2 * N); l.ac = lds_take(off, 2 * N); l.pr = lds_take(off, 2 * M);
    l.pc = lds_take(off, 2 * M); l.act = lds_take(off, N); l.alive = lds_take(off, M); l.pcnt = lds_take(off, M);
    l.pmv = lds_take(off, M); l.vis = lds_take(off, 4 * S);
#ifdef CM_BOUNDS
    l.nS2 = S * S; l.nN = N; l.nM = M; l.nS = S; l.status = status;
#endif
    return l;
}

// -DCM_BOUNDS builds a checked variant: an out-of-range LDS index raises status -100-site instead of
// silently reading 0 (LDS out-of-range reads are not faults), used to hunt indexing bugs on the GPU.
#ifdef CM_BOUNDS
__device__ __forceinline__ int chk_(const Lds &l, int i, int n, int site) {
    if ((unsigned)i < (unsigned)n) return i;
    atomicCAS(l.status, 0, -100 - site);
    return 0;
}
#define chk(l, i, n, site) chk_(l, i, (l).n, site)
#else
#define chk(l, i, n, site) (i)
#endif
__device__ __forceinline__ uint8_t &Gc(const Lds l, int i) { return smem[l.g + chk(l, i, nS2, 1)]; }
__device__ __forceinline__ int16_t &AR(const Lds l, int i) { return reinterpret_cast<int16_t *>(smem + l.ar)[chk(l, i, nN, 2)]; }
__device__ __forceinline__ int16_t &AC(const Lds l, int i) { return reinterpret_cast<int16_t *>(smem + l.ac)[chk(l, i, nN, 3)]; }
__device__ __forceinline__ int16_t &PR(const Lds l, int i) { return reinterpret_cast<int16_t *>(smem + l.pr)[chk(l, i, nM, 4)]; }
__device__ __forceinline__ int16_t &PC(const Lds l, int i) { return reinterpret_cast<int16_t *>(smem + l.pc)[chk(l, i, nM, 5)]; }
__device__ __forceinline__ uint8_t &ACT(const Lds l, int i) { return smem[l.act + chk(l, i, nN, 6)]; }
__device__ __forceinline__ uint8_t &ALV(const Lds l, int i) { return smem[l.alive + chk(l, i, nM, 7)]; }
__device__ __forceinline__ uint8_t &PCNT(const Lds l, int i) { return smem[l.pcnt + chk(l, i, nM, 8)]; }
__device__ __forceinline__ uint8_t &PMV(const Lds l, int i) { return smem[l.pmv + chk(l, i, nM, 9)]; }
__device__ __forceinline__ uint32_t &VIS(const Lds l, int i) { return reinterpret_cast<uint32_t *>(smem + l.vis)[chk(l, i, nS, 10)]; }

// Branch-free probes: the address is clamped into the tile and the result masked by the bounds test, so the
// four neighbour reads of count_adj are independent LDS loads (one round trip) instead of four dependent
// short-circuit branches.
__device__ __forceinline__ int cell(const Lds l, int r, int c, int S) {
    const int rr = min(max(r, 0), S - 1), cc = min(max(c, 0), S - 1);
    const int v = (int)Gc(l, rr * S + cc);
    return in_grid(r, c, S) ? v : -1;
}
// _neighbour_agents / _neighbour_preys count (predator_prey.py:309-351): D,U,R,L, each bounds-checked
__device__ __forceinline__ int count_adj(const Lds l, int r, int c, int S, int kind) {
    const int a = cell(l, r + 1, c, S), b = cell(l, r - 1, c, S), d = cell(l, r, c + 1, S), e = cell(l, r, c - 1, S);
    return (a == kind) + (b == kind) + (d == kind) + (e == kind);
}

struct Rng {
    uint32_t gid, step, k0, k1;
    __device__ __forceinline__ u32x4 at(uint32_t site, uint32_t idx) const { return philox4x32_10(gid, step, site, idx, k0, k1); }
};

// 4 consecutive uniforms of a link stream starting at flat index f0 (<= 2 Philox calls)
__device__ __forceinline__ void uniform4(const Rng &rng, uint32_t site, uint32_t f0, float u[4]) {
    const uint32_t q = f0 >> 2, o = f0 & 3;
    const u32x4 a = rng.at(site, q);
    if (o == 0) { u[0] = unit_f32(a.x); u[1] = unit_f32(a.y); u[2] = unit_f32(a.z); u[3] = unit_f32(a.w); return; }
    const u32x4 b = rng.at(site, q + 1);
    // window of 4 words starting at component o of (a, b), without a dynamically indexed array
    const uint32_t w0 = o == 1 ? a.y : (o == 2 ? a.z : a.w);
    const uint32_t w1 = o == 1 ? a.z : (o == 2 ? a.w : b.x);
    const uint32_t w2 = o == 1 ? a.w : (o == 2 ? b.x : b.y);
    const uint32_t w3 = o == 1 ? b.x : (o == 2 ? b.y : b.z);
    u[0] = unit_f32(w0); u[1] = unit_f32(w1); u[2] = unit_f32(w2); u[3] = unit_f32(w3);
}

// ---------------------------------------------------------------------------------------
// Sub-wave groups: LPE lanes per env (16 / 32 / 64), G = 64 / LPE envs per wave.  All G envs of a
// wave run the same program in lockstep; "group-uniform" values are identical within a group.
// ---------------------------------------------------------------------------------------
template <int LPE>
struct Grp {
    int sub, sl;                 // group index inside the wave, lane inside the group
    __device__ __forceinline__ unsigned long long mask() const {
        return LPE == 64 ? ~0ull : (((1ull << (LPE & 63)) - 1ull) << (sub * LPE));
    }
    __device__ __forceinline__ bool any(bool pred) const { return (__ballot(pred) & mask()) != 0ull; }
    __device__ __forceinline__ int count(bool pred) const { return __popcll(__ballot(pred) & mask()); }
};

// floor(k / d) for 0 <= k < 2^22, d >= 1, with a precomputed float reciprocal (instead of the ~35
// instruction integer division): the float product is off by at most one, fixed up exactly.
__device__ __forceinline__ int fdiv(int k, int d, float rcp) {
    int q = (int)((float)k * rcp);
    int r = k - q * d;
    if (r >= d) { ++q; } else if (r < 0) { --q; }
    return q;
}

// ---------------------------------------------------------------------------------------
// reset: rejection-sampled spawn (predator_prey.py:150-171,206-232; coverage.py:172-196,221-246).
// Every lane of a group evaluates the same candidate, lane 0 of the group commits it; groups that do
// not reset (need == false) idle through the loop.
// ---------------------------------------------------------------------------------------
template <int SCEN, int LPE>
__device__ __forceinline__ void do_reset(const EnvDev &p, const Lds l, const Rng rng, const cm_rng_tape &tape, int b,
                                         const Grp<LPE> g, bool need) {
    const int S = p.S, N = p.N, M = p.M, sl = g.sl;
    if (need) {
        for (int k = sl; k < S * S; k += LPE) Gc(l, k) = (SCEN == CM_CO) ? p.base_grid[k] : (uint8_t)C_EMPTY;
        if (SCEN == CM_CO) for (int r = sl; r < S; r += LPE) VIS(l, r) = 0u;
    }
    __syncthreads();
    const int lo = (SCEN == CM_CO) ? 1 : 0;                              // randint(1, m) vs randint(0, G-1)
    const int total = N + M;
    int e = need ? 0 : total, cursor = 0;
    bool fail = false;
    while (__any(e < total)) {
        const bool act = e < total;
        const bool is_prey = e >= N;
        int r = 0, c = 0;
        bool ok = false;
        if (act) {
            if (p.rng_mode == CM_RNG_TAPE) {
                if (cursor >= tape.spawn_cap) { fail = true; }
                else {
                    const int32_t *t = tape.spawn + ((size_t)b * tape.spawn_cap + cursor) * 2;
                    r = t[0]; c = t[1];
                    if (r < 0) fail = true;
                }
            } else {
                const u32x4 x = rng.at(SITE_SPAWN, (uint32_t)cursor);
                const uint32_t sp = (uint32_t)((SCEN == CM_CO) ? S - 2 : S);
                r = lo + (int)__umulhi(x.x, sp);
                c = lo + (int)__umulhi(x.y, sp);
            }
            ++cursor;
            if (!fail) {
                ok = in_grid(r, c, S) && Gc(l, r * S + c) == C_EMPTY;            // _is_cell_vacant
                if (ok && is_prey) ok = count_adj(l, r, c, S, C_AGENT) == 0;      // predator_prey.py:166
            }
        }
        __syncthreads();                                   // all probes done before anybody commits
        if (ok && sl == 0) {
            if (!is_prey) { AR(l, e) = (int16_t)r; AC(l, e) = (int16_t)c; Gc(l, r * S + c) = C_AGENT;
                            if (SCEN == CM_CO) VIS(l, r) |= (1u << c); }          // coverage.py:187
            else { PR(l, e - N) = (int16_t)r; PC(l, e - N) = (int16_t)c; Gc(l, r * S + c) = C_PREY; }
        }
        __syncthreads();
        if (ok) ++e;
        if (fail) e = total;
    }
    if (fail && sl == 0) raise(p, CM_ERR_TAPE);
    if (need) for (int j = sl; j < M; j += LPE) ALV(l, j) = 1;
    __syncthreads();
}

// ---------------------------------------------------------------------------------------
// emission: obs + dist_adj + channels + state write-back
// ---------------------------------------------------------------------------------------
template <int SCEN, int LPE>
__device__ __forceinline__ void emit(const EnvDev &p, const Lds l, const Rng rng, const cm_rng_tape &tape,
                                     const cm_step_out &out, int b, const Grp<LPE> g, int step_count, int slot) {
    const int S = p.S, N = p.N, M = p.M, R = p.R, W = p.W, d = p.d, WW = W * W, sl = g.sl;
    const float rcp_d = p.rcp_d, rcp_W = p.rcp_W, rcp_N = p.rcp_N, rcp_WW = p.rcp_WW, rcp_NN = p.rcp_NN;
    // ---- observations [N*d], lanes stride the flattened row -> coalesced stores ----
    if (out.obs) {
        float *o = out.obs + (size_t)b * N * d;
        const int total = N * d;
        for (int k = sl; k < total; k += LPE) {
            const int i = fdiv(k, d, rcp_d), f = k - i * d;
            const int r0 = AR(l, i), c0 = AC(l, i);
            float v;
            if (SCEN == CM_PP) {
                if (f < 2 * WW) {                                   // get_neighbors (predator_prey.py:173-181)
                    const int chn = f >= WW, w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                    v = (cell(l, r0 - R + wr, c0 - R + wc, S) == (chn ? C_PREY : C_AGENT)) ? 1.0f : 0.0f;
                } else if (f == 2 * WW) v = p.lut_row[r0];          // row / G          (:195)
                else if (f == 2 * WW + 1) v = p.lut_col[c0];        // col / (G-1)      (:195)
                else v = p.lut_step[step_count];                    // step / Tmax      (:196)
            } else {
                if (f < 3 * WW) {                                   // get_local_view (coverage.py:448-480)
                    const int chn = fdiv(f, WW, rcp_WW), w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                    const int rr = r0 - R + wr, cc = c0 - R + wc;
                    const bool in = in_grid(rr, cc, S);
                    if (chn == 0) v = (!in || Gc(l, rr * S + cc) == C_WALL) ? 1.0f : 0.0f;
                    else if (chn == 1) v = (in && Gc(l, rr * S + cc) == C_AGENT) ? 1.0f : 0.0f;
                    else v = (in && ((VIS(l, rr) >> cc) & 1u)) ? 1.0f : 0.0f;
                } else if (f == 3 * WW) v = p.lut_row[r0];          // round(row/(S-1), 2) (:206)
                else if (f == 3 * WW + 1) v = p.lut_col[c0];
                else v = p.lut_step[step_count];
            }
            o[k] = v;
        }
    }
    // ---- range adjacency (env_communication.py:218-243): integer form of cdist <= Rcom_th ----
    if (out.dist_adj && !p.adj_const) {
        float *a = out.dist_adj + (size_t)b * N * N;
        for (int k = sl; k < N * N; k += LPE) {
            const int i = fdiv(k, N, rcp_N), j = k - i * N;
            const int dr = AR(l, i) - AR(l, j), dc = AC(l, i) - AC(l, j);
            a[k] = (dr * dr + dc * dc <= p.rc2) ? 1.0f : 0.0f;
        }
    }
    // ---- channel masks ----
    const int NN = N * N, L = p.L;
    if (p.channel == CM_CH_IID && out.channels) {                   // get_iid_channel (:200-214)
        float *ch = out.channels + (size_t)b * L * NN;
        const int total = L * NN;
        if (p.rng_mode == CM_RNG_TAPE) {
            const float *u = tape.iid_u + ((size_t)b * 2 + slot) * total;
            for (int k = sl; k < total; k += LPE) {
                const int ij = k - fdiv(k, NN, rcp_NN) * NN, i = fdiv(ij, N, rcp_N), j = ij - i * N;
                ch[k] = ((u[k] + (i == j ? 1.0f : 0.0f)) >= p.ploss) ? 1.0f : 0.0f;
            }
        } else {
            const uint32_t site = slot ? SITE_IID_RESET : SITE_IID_STEP;
            for (int q = sl; q * 4 < total; q += LPE) {
                const u32x4 x = rng.at(site, (uint32_t)q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = q * 4 + e;
                    if (k < total) {
                        const int ij = k - fdiv(k, NN, rcp_NN) * NN, i = fdiv(ij, N, rcp_N), j = ij - i * N;
                        ch[k] = ((unit_f32(pick(x, e)) + (i == j ? 1.0f : 0.0f)) >= p.ploss) ? 1.0f : 0.0f;
                    }
                }
            }
        }
    } else if (p.channel == CM_CH_GE) {                              // env_communication.py:106-157, GE model :121-150
        float *ch = out.channels ? out.channels + (size_t)b * L * NN : nullptr;
        uint8_t *gs = p.ge_state + (size_t)b * NN;
        const uint32_t site = slot ? SITE_GE_RESET : SITE_GE_STEP;
        const int l0 = slot ? 1 : 0;                                 // reset: hop 0 = all good, then L-1 transitions
        for (int k0 = sl * 4; k0 < NN; k0 += LPE * 4) {
            uint8_t s[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] = (k0 + e < NN) ? (slot ? (uint8_t)1 : gs[k0 + e]) : (uint8_t)0;
            if (slot && ch)
                for (int e = 0; e < 4; ++e) if (k0 + e < NN) ch[k0 + e] = 1.0f;
            for (int hop = l0; hop < L; ++hop) {
                float ugb[4], ubg[4];
                if (p.rng_mode == CM_RNG_TAPE) {
                    const float *u = tape.ge_u + (((size_t)b * 2 + slot) * L + hop) * 2 * NN;
                    for (int e = 0; e < 4; ++e) { const int k = k0 + e < NN ? k0 + e : NN - 1; ugb[e] = u[k]; ubg[e] = u[NN + k]; }
                } else {
                    uniform4(rng, site, (uint32_t)((hop * 2 + 0) * NN + k0), ugb);
                    uniform4(rng, site, (uint32_t)((hop * 2 + 1) * NN + k0), ubg);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + e;
                    if (k >= NN) continue;
                    const int i = fdiv(k, N, rcp_N), j = k - i * N;
                    const float eye = (i == j) ? 1.0f : 0.0f;
                    const bool e_gb = (ugb[e] + eye) < p.pgb, e_bg = (ubg[e] + eye) < p.pbg;
                    const bool g_next = s[e] && !(s[e] && e_gb), b_next = (!s[e]) && e_bg;
                    s[e] = (uint8_t)(g_next || b_next);
                    if (ch) ch[(size_t)hop * NN + k] = (float)s[e];
                }
            }
            for (int e = 0; e < 4; ++e) if (k0 + e < NN) gs[k0 + e] = s[e];
        }
    }
    // ---- state write-back ----
    for (int i = sl; i < N; i += LPE) p.agent_pos[(size_t)b * N + i] = make_int2(AR(l, i), AC(l, i));
    if (SCEN == CM_PP) {
        for (int j = sl; j < M; j += LPE) {
            p.prey_pos[(size_t)b * M + j] = make_int2(PR(l, j), PC(l, j));
            p.alive[(size_t)b * M + j] = ALV(l, j);
        }
    } else {
        for (int r = sl; r < S; r += LPE) p.visited[(size_t)b * S + r] = VIS(l, r);
    }
}

// ---------------------------------------------------------------------------------------
// the step kernel
// ---------------------------------------------------------------------------------------
template <int SCEN, int LPE>
__global__ __launch_bounds__(WAVE) void env_kernel(EnvDev p, const int32_t *__restrict__ actions, cm_rng_tape tape,
                                                  cm_step_out out, int reset_only) {
    constexpr int G = WAVE / LPE;
    Grp<LPE> g;
    g.sub = threadIdx.x / LPE; g.sl = threadIdx.x % LPE;
    const int sl = g.sl;
    const int b_raw = blockIdx.x * G + g.sub;
    const bool valid = b_raw < p.B;
    const int b = valid ? b_raw : p.B - 1;            // idle groups shadow the last env and never commit
    const Lds l = make_lds(p.S, p.N, p.M, p.lds_env * g.sub, p.status);
    const int S = p.S, N = p.N, M = p.M;
    Rng rng{ (uint32_t)(p.env_id_offset + b), p.rng_step[b], p.key0, p.key1 };
    cm_step_out o = out;
    if (!valid) { o.obs = nullptr; o.dist_adj = nullptr; o.channels = nullptr; }

    if (reset_only) {
        do_reset<SCEN, LPE>(p, l, rng, tape, b, g, true);
        if (!valid) return;
        if (sl == 0) { p.step_count[b] = 0; if (SCEN == CM_CO) p.total_capture[b] = 0; p.rng_step[b] = rng.step + 1; }
        emit<SCEN, LPE>(p, l, rng, tape, o, b, g, 0, 1);
        return;
    }

    // ---- load SoA state, rebuild the occupancy tile in LDS ----
    bool bad_action = false;
    for (int i = sl; i < N; i += LPE) {
        const int2 q = p.agent_pos[(size_t)b * N + i];
        AR(l, i) = (int16_t)q.x; AC(l, i) = (int16_t)q.y;
        const int a = actions[(size_t)b * N + i];
        const bool bad = (unsigned)a > 4u;
        bad_action |= bad;
        ACT(l, i) = (uint8_t)(bad ? 4 : a);
    }
    if (SCEN == CM_PP)
        for (int j = sl; j < M; j += LPE) {
            const int2 q = p.prey_pos[(size_t)b * M + j];
            PR(l, j) = (int16_t)q.x; PC(l, j) = (int16_t)q.y;
            ALV(l, j) = p.alive[(size_t)b * M + j];
        }
    for (int k = sl; k < S * S; k += LPE) Gc(l, k) = (SCEN == CM_CO) ? p.base_grid[k] : (uint8_t)C_EMPTY;
    if (SCEN == CM_CO) for (int r = sl; r < S; r += LPE) VIS(l, r) = p.visited[(size_t)b * S + r];
    // the reference raises on a bad action (predator_prey.py:255): flag it; the env is left untouched
    const bool env_bad = g.any(bad_action);
    if (env_bad && sl == 0 && valid) raise(p, CM_ERR_ACTION);
    const bool commit = valid && !env_bad;
    __syncthreads();
    if (p.stop == 1) return;
    for (int i = sl; i < N; i += LPE) Gc(l, AR(l, i) * S + AC(l, i)) = C_AGENT;
    if (SCEN == CM_PP)
        for (int j = sl; j < M; j += LPE) if (ALV(l, j)) Gc(l, PR(l, j) * S + PC(l, j)) = C_PREY;
    __syncthreads();

    if (p.stop == 2) return;
    int step_count = p.step_count[b] + 1;
    int succ = p.success[b];
    int done = 0;
    double reward;
    int det0 = 0, det1 = 0, det2 = 0, det3 = 0, det4 = 0, det5 = 0;

    if (SCEN == CM_PP) {
        // ---- agents move in index order (predator_prey.py:497-500, :240-261): group-uniform loop ----
        int moving = 0;
        for (int i = 0; i < N; ++i) {
            const int a = ACT(l, i);
            bool mv = false;
            int r = 0, c = 0, nr = 0, nc = 0;
            if (a != 4) {
                ++moving;
                r = AR(l, i); c = AC(l, i); nr = r + dr_of(a); nc = c + dc_of(a);
                mv = in_grid(nr, nc, S) && Gc(l, nr * S + nc) == C_EMPTY;
            }
            __syncthreads();
            if (mv && sl == 0) { Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_AGENT; AR(l, i) = (int16_t)nr; AC(l, i) = (int16_t)nc; }
            __syncthreads();
        }
        if (p.stop == 3) return;
        // ---- per-prey work that only depends on the (now static) agent layer: one lane per prey ----
        for (int j = sl; j < M; j += LPE) {
            int cnt = 0, mv = 4;
            if (ALV(l, j)) {
                const int r = PR(l, j), c = PC(l, j);
                cnt = count_adj(l, r, c, S, C_AGENT);
                // prey_random_move (:396-407): first of <=5 draws whose target has no predator neighbour
                const bool captured_now = (p.load == 2) && cnt >= 1 && p.load <= cnt;
                if (!captured_now) {
                    bool found = false;
                    u32x4 x = { 0, 0, 0, 0 };
                    for (int t = 0; t < 5 && !found; ++t) {
                        int m;
                        if (p.rng_mode == CM_RNG_TAPE) {
                            m = tape.prey[((size_t)b * M + j) * 5 + t];
                            if (m > 4) { mv = 4 | 8; break; }     // recorded tape ended: only legal if prey gets captured
                        } else {
                            if ((t & 3) == 0) x = rng.at(SITE_PREY, (uint32_t)(2 * j + (t >> 2)));
                            m = prey_move_from_u32(pick(x, t & 3));
                        }
                        if (count_adj(l, r + dr_of(m), c + dc_of(m), S, C_AGENT) == 0) { mv = m; found = true; }
                    }
                }
            }
            PCNT(l, j) = (uint8_t)cnt; PMV(l, j) = (uint8_t)mv;
        }
        // prey_watching (:419-423): agents 4-adjacent to a live prey (prey layer still at start-of-phase positions)
        int wsum = 0;
        for (int i0 = 0; i0 < N; i0 += LPE) {
            const int i = i0 + sl;
            const bool w = i < N && count_adj(l, AR(l, i), AC(l, i), S, C_PREY) > 0;
            wsum += g.count(w);
        }
        __syncthreads();
        if (p.stop == 4) return;
        // ---- captures + prey moves in index order (:416-432 / :460-478, :276-301): group-uniform loop ----
        int capture = 0, penalty = 0;
        bool tape_short = false;
        for (int j = 0; j < M; ++j) {
            const bool alive = ALV(l, j) != 0;
            bool captured = false, moved = false;
            int r = 0, c = 0, nr = 0, nc = 0;
            if (alive) {
                r = PR(l, j); c = PC(l, j);
                const int cnt = PCNT(l, j), mvb = PMV(l, j);
                if (cnt >= 1) {
                    int need = p.load;
                    if (p.load != 2) {                                       // reward_individual :467-470
                        const bool on_r = (r == 0 || r == S - 1), on_c = (c == 0 || c == S - 1);
                        const int adj = (on_r && on_c) ? 2 : ((on_r || on_c) ? 3 : p.load);   // __create_edges :123-144
                        const int avail = adj - count_adj(l, r, c, S, C_PREY);
                        need = p.load < avail ? p.load : avail;
                    }
                    if (need <= cnt) { captured = true; ++capture; } else ++penalty;
                }
                if (!captured) {
                    if (mvb & 8) tape_short = true;
                    const int mv = mvb & 7;
                    if (mv != 4) {
                        nr = r + dr_of(mv); nc = c + dc_of(mv);
                        moved = in_grid(nr, nc, S) && Gc(l, nr * S + nc) == C_EMPTY;
                    }
                }
            }
            __syncthreads();
            if (sl == 0) {
                if (captured) { ALV(l, j) = 0; Gc(l, r * S + c) = C_EMPTY; }      // :301
                else if (moved) { Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_PREY; PR(l, j) = (int16_t)nr; PC(l, j) = (int16_t)nc; }
            }
            __syncthreads();
        }
        if (p.stop == 5) return;
        if (tape_short && sl == 0 && commit) raise(p, CM_ERR_TAPE_PREY);
        // reward in f64 exactly as the Python expression evaluates (:434 / :480); no FMA contraction (build flag)
        // (step + cap*c) + (mc*m)/N [+ pen*p]: the two count-indexed terms come from host tables built with the
        // same f64 operations (no f64 division on the device)
        reward = p.rew_lut[capture] + p.rew_lut[(M + 1) + moving];
        if (p.load == 2) reward = reward + p.penalty * (double)penalty;
        det0 = capture; det1 = moving; det2 = penalty; det4 = wsum;
        bool any_alive = false;
        for (int j0 = 0; j0 < M; j0 += LPE) any_alive |= g.any(j0 + sl < M && ALV(l, j0 + sl));
        if (o.prey_alive && commit) for (int j = sl; j < M; j += LPE) o.prey_alive[(size_t)b * M + j] = ALV(l, j);
        done = (step_count >= p.max_steps) || !any_alive;               // :511-517
        if (done) succ = any_alive ? 0 : 1;
    } else {
        // ---- Coverage.step (:319-378): sequential agents against tile + visited bitmap ----
        int cap = 0, mov = 0, pen = 0, lazy = 0, rev = 0;
        for (int i = 0; i < N; ++i) {
            const int a = ACT(l, i);
            bool mv = false, seen = false;
            int r = 0, c = 0, nr = 0, nc = 0;
            if (a == 4) ++lazy;
            else {
                ++mov;
                r = AR(l, i); c = AC(l, i); nr = r + dr_of(a); nc = c + dc_of(a);
                mv = in_grid(nr, nc, S) && Gc(l, nr * S + nc) == C_EMPTY;
                if (mv) { seen = (VIS(l, nr) >> nc) & 1u; if (seen) ++rev; else ++cap; }
                else ++pen;
            }
            __syncthreads();
            if (mv && sl == 0) {
                VIS(l, nr) |= (1u << nc);
                Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_AGENT; AR(l, i) = (int16_t)nr; AC(l, i) = (int16_t)nc;
            }
            __syncthreads();
        }
        const int total = p.total_capture[b] + cap;
        double fin = 0.0;
        if (total == p.n_empty) { fin = p.final_reward; done = 1; }     // :381-385
        if (step_count >= p.max_steps) { succ = done ? 1 : 0; done = 1; }   // :388-393
        if (sl == 0 && commit) p.total_capture[b] = total;
        // get_reward (:299-317), left-to-right; term_k[count] = coef_k * (count / N) tabulated on the host in f64
        const double *T = p.rew_lut;
        reward = p.step_cost + T[cap];
        reward = reward + T[(N + 1) + mov];
        reward = reward + T[2 * (N + 1) + pen];
        reward = reward + T[3 * (N + 1) + lazy];
        reward = reward + T[4 * (N + 1) + rev];
        reward = reward + fin;
        det0 = cap; det1 = mov; det2 = pen; det3 = lazy; det4 = rev; det5 = fin != 0.0;
    }

    if (step_count >= p.mpl) done = 1;                                   // vec_env_executor.py:33-34
    if (sl == 0 && commit) {
        if (o.reward) o.reward[b] = (float)reward;
        if (o.reward_f64) o.reward_f64[b] = reward;
        if (o.done) o.done[b] = (uint8_t)done;
        if (o.path_len) o.path_len[b] = done ? step_count : 0;
        if (o.details) {
            int32_t *dd = o.details + (size_t)b * 6;
            dd[0] = det0; dd[1] = det1; dd[2] = det2; dd[3] = det3; dd[4] = det4; dd[5] = det5;
        }
        p.rng_step[b] = rng.step + 1;
    }
    __syncthreads();
    if (p.stop == 6) return;
    // auto-reset (:36-43): groups whose env finished re-spawn and emit the reset observation
    do_reset<SCEN, LPE>(p, l, rng, tape, b, g, done != 0);
    if (done) step_count = 0;
    if (!commit || p.stop == 7) return;
    if (sl == 0) {
        if (done && SCEN == CM_CO) p.total_capture[b] = 0;
        p.step_count[b] = step_count;
        p.success[b] = succ;
        if (o.success) o.success[b] = succ;
    }
    emit<SCEN, LPE>(p, l, rng, tape, o, b, g, step_count, done ? 1 : 0);
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
    if (S < 2 || S > 32) return set_error(CM_ERR_ARG, "grid side (incl. wall ring) must be in [2, 32]: visited rows are 32-bit masks");
    if (c.scenario == CM_PP && (c.n_preys < 0 || c.n_preys > 255)) return set_error(CM_ERR_ARG, "0 <= n_preys <= 255 required");
    if (c.scenario == CM_PP && (c.load < 2 || c.load > 4)) return set_error(CM_ERR_LOAD, "PP load must be 2, 3 or 4 (capv undefined otherwise, predator_prey.py:77-79)");
    if (c.scenario == CM_CO && c.grid % 10 != 0) return set_error(CM_ERR_ARG, "CO map must be a multiple of 10 (coverage.py:67)");
    if (c.n_hops < 1 || c.n_hops > 8) return set_error(CM_ERR_ARG, "1 <= n_hops <= 8 required");
    if (c.max_steps < 1 || c.max_path_length < 1) return set_error(CM_ERR_ARG, "max_steps and max_path_length must be >= 1");
    if (c.channel < CM_CH_FC || c.channel > CM_CH_GE) return set_error(CM_ERR_ARG, "bad channel");
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
                 o_vis = take(B * S * 4), o_sc = take(B * 4), o_tc = take(B * 4), o_su = take(B * 4),
                 o_ge = take(c.channel == CM_CH_GE ? B * N * N : 1), o_rs = take(B * 4), o_st = take(4),
                 o_bg = take((size_t)S * S), o_lr = take(S * 4), o_lc = take(S * 4), o_ls = take((c.max_steps + 1) * 4),
                 o_rl = take(rew_lut.size() * 8);
    h->arena_bytes = off;
    hipError_t e = hipMalloc(&h->arena, off);
    if (e != hipSuccess) { delete h; return hip_fail(e, "hipMalloc(env arena)"); }
    char *base = (char *)h->arena;
    e = hipMemset(base, 0, off);
    if (e != hipSuccess) { hipFree(h->arena); delete h; return hip_fail(e, "hipMemset"); }
    d.agent_pos = (int2 *)(base + o_ap); d.prey_pos = (int2 *)(base + o_pp); d.alive = (uint8_t *)(base + o_al);
    d.visited = (uint32_t *)(base + o_vis); d.step_count = (int32_t *)(base + o_sc); d.total_capture = (int32_t *)(base + o_tc);
    d.success = (int32_t *)(base + o_su); d.ge_state = (uint8_t *)(base + o_ge); d.rng_step = (uint32_t *)(base + o_rs);
    d.status = (int32_t *)(base + o_st);
    d.base_grid = (const uint8_t *)(base + o_bg); d.lut_row = (const float *)(base + o_lr); d.lut_col = (const float *)(base + o_lc);
    d.lut_step = (const float *)(base + o_ls);
    d.rew_lut = (const double *)(base + o_rl);
    hipMemcpy(base + o_rl, rew_lut.data(), rew_lut.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(base + o_bg, walls.data(), walls.size(), hipMemcpyHostToDevice);
    hipMemcpy(base + o_lr, lut_row.data(), S * 4, hipMemcpyHostToDevice);
    hipMemcpy(base + o_lc, lut_col.data(), S * 4, hipMemcpyHostToDevice);
    hipMemcpy(base + o_ls, lut_step.data(), (c.max_steps + 1) * 4, hipMemcpyHostToDevice);
    if (c.channel == CM_CH_GE) hipMemset(base + o_ge, 1, B * N * N);
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
    d.rcp_d = 1.0f / (float)d.d; d.rcp_W = 1.0f / (float)d.W; d.rcp_N = 1.0f / (float)d.N;
    d.rcp_WW = 1.0f / (float)(d.W * d.W); d.rcp_NN = 1.0f / (float)(d.N * d.N);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) { hipFree(h->arena); delete h; return hip_fail(e, "cm_env_create sync"); }
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

static int check_tape(const cm_env *h, const cm_rng_tape *tape, bool is_reset) {
    if (h->cfg.rng_mode != CM_RNG_TAPE) return CM_OK;
    if (!tape) return set_error(CM_ERR_ARG, "rng_mode is TAPE but no tape was passed");
    if (!tape->spawn || tape->spawn_cap <= 0) return set_error(CM_ERR_ARG, "tape.spawn required in tape mode");
    if (!is_reset && h->dev.scen == CM_PP && h->dev.M > 0 && !tape->prey) return set_error(CM_ERR_ARG, "tape.prey required for PP steps");
    if (h->dev.channel == CM_CH_IID && !tape->iid_u) return set_error(CM_ERR_ARG, "tape.iid_u required for IID channel");
    if (h->dev.channel == CM_CH_GE && !tape->ge_u) return set_error(CM_ERR_ARG, "tape.ge_u required for GE channel");
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
#define CM_LAUNCH(SC, LP) hipLaunchKernelGGL((env_kernel<SC, LP>), grid, block, h->lds_bytes, st, d, actions, t, *out, reset_only)
    if (d.scen == CM_PP) { if (d.lpe == 16) CM_LAUNCH(CM_PP, 16); else if (d.lpe == 32) CM_LAUNCH(CM_PP, 32); else CM_LAUNCH(CM_PP, 64); }
    else { if (d.lpe == 16) CM_LAUNCH(CM_CO, 16); else if (d.lpe == 32) CM_LAUNCH(CM_CO, 32); else CM_LAUNCH(CM_CO, 64); }
#undef CM_LAUNCH
    CM_HIP(hipGetLastError());
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
        { s->visited, d.visited, B * S * 4 }, { s->step_count, d.step_count, B * 4 }, { s->total_capture, d.total_capture, B * 4 },
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
