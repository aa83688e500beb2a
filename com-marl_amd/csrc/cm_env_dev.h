// cm_env_dev.h - device code of the batched Predator-Prey / Coverage env step (see cm_env.hip for the mapping and the
// reference citations).  Header so that the stand-alone env kernel (cm_env.hip) and the fused rollout kernel
// (cm_fused.hip) instantiate the same body.
#pragma once
#include "cm_internal.h"
#include "cm_rng.h"

namespace cm {

constexpr int C_EMPTY = 0, C_AGENT = 1, C_PREY = 2, C_WALL = 3;
constexpr int WAVE = 64;
constexpr int PAR_AGENTS_MIN = 16;  // teams larger than this resolve their moves in parallel rounds (agents_parallel)

// Every LDS hand-off in the env code is between lanes of ONE wave (an env's group never spans waves), and a wave's LDS
// operations execute in issue order: draining the LDS counter is all the synchronisation needed.  (A workgroup barrier
// would also drain the vector-memory counter - exposing every outstanding global load - and could not be used inside
// the data-dependent reset loop once several waves share a workgroup.)
#define ENV_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// Diagnostic (COMMARL_ENV_STOP=-1): shader-clock stamps of thread 0 of workgroup 0 at the phase boundaries of one step;
// the launching translation unit prints the differences (cm_env.hip).
static __device__ unsigned long long g_env_probe[16];
#define ENV_PROBE(i) do { asm volatile("; ENV_PROBE " #i); if (p.stop < 0 && blockIdx.x == 0 && thread_x() == 0) g_env_probe[i] = __builtin_amdgcn_s_memtime(); } while (0)

// displacement of action a (predator_prey.py:244-253; 0 down, 1 left, 2 up, 3 right, 4 stay, 5 = the faulty agent's
// "no displacement"): delta + 1 as six 2-bit fields of a constant - shift, field extract, add instead of two compare / select
// pairs (these sit in every iteration of the order-dependent loops)
__device__ __forceinline__ int dr_of(int a) { return (int)((1350u >> (2 * a)) & 3u) - 1; }   // +1 0 -1 0 0 0
__device__ __forceinline__ int dc_of(int a) { return (int)((1425u >> (2 * a)) & 3u) - 1; }   // 0 -1 0 +1 0 0
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
    int vis, vw;       // [S * vw] u32 visited bitmap: vw = ceil(S / 32) words per grid row (bit c & 31 of word r * vw + (c >> 5))
    int own, win, st;  // parallel agent resolution: [S*S] u8 owner index + 1, [S*S] u32 lowest contender, [N] u8 status
#ifdef CM_BOUNDS
    int nS2, nN, nM, nS;
    int32_t *status;
#endif
};

__host__ __device__ inline int lds_take(int &off, int bytes) { const int o = off; off += (bytes + 15) & ~15; return o; }
__host__ __device__ inline int lds_env_bytes(int S, int N, int M) {
    int off = 0;
    lds_take(off, S * S); lds_take(off, 2 * N); lds_take(off, 2 * N); lds_take(off, 2 * M); lds_take(off, 2 * M);
    lds_take(off, N); lds_take(off, M); lds_take(off, M); lds_take(off, M); lds_take(off, 4 * S * ((S + 31) >> 5));
    lds_take(off, S * S); lds_take(off, 4 * S * S); lds_take(off, N);
    return off;
}
__device__ __forceinline__ Lds make_lds(int S, int N, int M, int base, int32_t *status) {
    int off = base;
    Lds l;
    l.g = lds_take(off, S * S); l.ar = lds_take(off, 2 * N); l.ac = lds_take(off, 2 * N); l.pr = lds_take(off, 2 * M);
    l.pc = lds_take(off, 2 * M); l.act = lds_take(off, N); l.alive = lds_take(off, M); l.pcnt = lds_take(off, M);
    l.pmv = lds_take(off, M); l.vw = (S + 31) >> 5; l.vis = lds_take(off, 4 * S * l.vw);
    l.own = lds_take(off, S * S); l.win = lds_take(off, 4 * S * S); l.st = lds_take(off, N);
#ifdef CM_BOUNDS
    l.nS2 = S * S; l.nN = N; l.nM = M; l.nS = S * ((S + 31) >> 5); l.status = status;
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
__device__ __forceinline__ uint8_t &OWN(const Lds l, int i) { return smem[l.own + chk(l, i, nS2, 12)]; }
__device__ __forceinline__ uint32_t &WIN(const Lds l, int i) { return reinterpret_cast<uint32_t *>(smem + l.win)[chk(l, i, nS2, 13)]; }
__device__ __forceinline__ uint8_t &ST(const Lds l, int i) { return smem[l.st + chk(l, i, nN, 14)]; }
__device__ __forceinline__ uint32_t &VIS(const Lds l, int i) { return reinterpret_cast<uint32_t *>(smem + l.vis)[chk(l, i, nS, 10)]; }   // word i
__device__ __forceinline__ uint32_t &VISW(const Lds l, int r, int c) { return VIS(l, r * l.vw + (c >> 5)); }                        // word of cell (r, c)
__device__ __forceinline__ uint32_t vbit(int c) { return 1u << (c & 31); }

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
        if (SCEN == CM_CO) for (int r = sl; r < S * l.vw; r += LPE) VIS(l, r) = 0u;
    }
    ENV_SYNC();
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
        ENV_SYNC();                                   // all probes done before anybody commits
        if (ok && sl == 0) {
            if (!is_prey) { AR(l, e) = (int16_t)r; AC(l, e) = (int16_t)c; Gc(l, r * S + c) = C_AGENT;
                            if (SCEN == CM_CO) VISW(l, r, c) |= vbit(c); }          // coverage.py:187
            else { PR(l, e - N) = (int16_t)r; PC(l, e - N) = (int16_t)c; Gc(l, r * S + c) = C_PREY; }
        }
        ENV_SYNC();
        if (ok) ++e;
        if (fail) e = total;
    }
    if (fail && sl == 0) raise(p, CM_ERR_TAPE);
    if (need) for (int j = sl; j < M; j += LPE) ALV(l, j) = 1;
    if (need && SCEN == CM_PP) for (int i = sl; i < N; i += LPE) p.agent_cond[(size_t)b * N + i] = 1;   // __init_full_obs :152
    ENV_SYNC();
}

// ---------------------------------------------------------------------------------------
// emission: obs + dist_adj + channels + state write-back
// ---------------------------------------------------------------------------------------
// The observation's coordinate / clock entries come from host-built tables (exact reference arithmetic).  Read inside the
// element loop each is a dependent global load in front of that iteration's store (1.7 k of the emission's 6.6 k clocks at
// the headline shape); a caller that knows the step early passes them in registers instead: lane s of an env's group holds
// row[s] and col[s] (grid side <= lanes per env) and the element fetches its entry with a lane shuffle.
struct ObsTabs { bool held; float row, col, step; };
// Carried rollout (cm_rollout_w.hip): inside a multi-step launch a wave's envs hand their state from step to step through LDS
// and registers instead of through the global arrays (which are still written every step): the next step loads nothing a store
// of this launch produced, so no fence stands between two steps and the trajectory stores drain under the next policy forward.
constexpr int OBS_COPY_STRIDE = 24;                      // floats per observation row of the LDS copy (d <= 24)
struct EnvCarry { int step_count, succ, done; };

template <int SCEN, int LPE>
__device__ __forceinline__ void emit(const EnvDev &p, const Lds l, const Rng rng, const cm_rng_tape &tape,
                                     const cm_step_out &out, int b, const Grp<LPE> g, int step_count, int slot,
                                     const ObsTabs tabs = ObsTabs{ false, 0.0f, 0.0f, 0.0f }, int obs_copy = -1) {
    const int S = p.S, N = p.N, M = p.M, R = p.R, W = p.W, d = p.d, WW = W * W, sl = g.sl;
    const float rcp_d = p.rcp_d, rcp_W = p.rcp_W, rcp_N = p.rcp_N, rcp_WW = p.rcp_WW, rcp_NN = p.rcp_NN;
    // ---- observations [N*d], lanes stride the flattened row -> coalesced stores ----
    if (out.obs && tabs.held && LPE == 16) {
        // 16 lanes per env (small teams; the carried rollout): 6 elements per lane and pass, in stages - every position read of the pass
        // is requested before the first tile probe, every probe before the first store (the LDS copy's stores would otherwise order every
        // later LDS read behind them).  +1.5 % on the headline step; with 64 lanes per env (large teams) the plain loop below is as fast.
        float *o = out.obs + (size_t)b * N * d;
        const int total = N * d;
        constexpr int U = 6;
        for (int k0 = 0; k0 < total; k0 += U * LPE) {                   // uniform trip count: every lane takes part in the shuffles
            int kk[U], ii[U], ff[U], r0[U], c0[U];
            bool on[U];
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + u * LPE + sl;
                on[u] = k < total;
                kk[u] = on[u] ? k : 0;
                ii[u] = fdiv(kk[u], d, rcp_d); ff[u] = kk[u] - ii[u] * d;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { r0[u] = AR(l, ii[u]); c0[u] = AC(l, ii[u]); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int f = ff[u];
                const float vrow = __shfl(tabs.row, r0[u], LPE), vcol = __shfl(tabs.col, c0[u], LPE);
                float v;
                if (SCEN == CM_PP) {
                    if (f < 2 * WW) {                                   // get_neighbors (predator_prey.py:173-181)
                        const int chn = f >= WW, w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                        v = (cell(l, r0[u] - R + wr, c0[u] - R + wc, S) == (chn ? C_PREY : C_AGENT)) ? 1.0f : 0.0f;
                    } else v = f == 2 * WW ? vrow : (f == 2 * WW + 1 ? vcol : tabs.step);      // (:195-196)
                } else {
                    if (f < 3 * WW) {                                   // get_local_view (coverage.py:448-480)
                        const int chn = fdiv(f, WW, rcp_WW), w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                        const int rr = r0[u] - R + wr, cc = c0[u] - R + wc;
                        const bool in = in_grid(rr, cc, S);
                        if (chn == 0) v = (!in || Gc(l, rr * S + cc) == C_WALL) ? 1.0f : 0.0f;
                        else if (chn == 1) v = (in && Gc(l, rr * S + cc) == C_AGENT) ? 1.0f : 0.0f;
                        else v = (in && ((VISW(l, rr, cc) >> (cc & 31)) & 1u)) ? 1.0f : 0.0f;
                    } else v = f == 3 * WW ? vrow : (f == 3 * WW + 1 ? vcol : tabs.step);      // (:206)
                }
                vv[u] = v;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (on[u]) o[kk[u]] = vv[u];
                // the persistent rollout keeps a copy for its next policy forward (rows of OBS_COPY_STRIDE floats in the env's LDS area)
                if (on[u] && obs_copy >= 0) reinterpret_cast<float *>(smem + obs_copy)[ii[u] * OBS_COPY_STRIDE + ff[u]] = vv[u];
            }
        }
    } else if (out.obs && tabs.held) {
        float *o = out.obs + (size_t)b * N * d;
        const int total = N * d;
        for (int k0 = 0; k0 < total; k0 += LPE) {                       // uniform trip count: every lane takes part in the shuffles
            const int k = k0 + sl;
            const bool on = k < total;
            const int kk = on ? k : 0;
            const int i = fdiv(kk, d, rcp_d), f = kk - i * d;
            const int r0 = AR(l, i), c0 = AC(l, i);
            const float vrow = __shfl(tabs.row, r0, LPE), vcol = __shfl(tabs.col, c0, LPE);
            float v;
            if (SCEN == CM_PP) {
                if (f < 2 * WW) {                                   // get_neighbors (predator_prey.py:173-181)
                    const int chn = f >= WW, w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                    v = (cell(l, r0 - R + wr, c0 - R + wc, S) == (chn ? C_PREY : C_AGENT)) ? 1.0f : 0.0f;
                } else v = f == 2 * WW ? vrow : (f == 2 * WW + 1 ? vcol : tabs.step);      // (:195-196)
            } else {
                if (f < 3 * WW) {                                   // get_local_view (coverage.py:448-480)
                    const int chn = fdiv(f, WW, rcp_WW), w = f - chn * WW, wr = fdiv(w, W, rcp_W), wc = w - wr * W;
                    const int rr = r0 - R + wr, cc = c0 - R + wc;
                    const bool in = in_grid(rr, cc, S);
                    if (chn == 0) v = (!in || Gc(l, rr * S + cc) == C_WALL) ? 1.0f : 0.0f;
                    else if (chn == 1) v = (in && Gc(l, rr * S + cc) == C_AGENT) ? 1.0f : 0.0f;
                    else v = (in && ((VISW(l, rr, cc) >> (cc & 31)) & 1u)) ? 1.0f : 0.0f;
                } else v = f == 3 * WW ? vrow : (f == 3 * WW + 1 ? vcol : tabs.step);      // (:206)
            }
            if (on) o[k] = v;
            if (on && obs_copy >= 0) reinterpret_cast<float *>(smem + obs_copy)[i * OBS_COPY_STRIDE + f] = v;
        }
    } else if (out.obs) {
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
                    else v = (in && ((VISW(l, rr, cc) >> (cc & 31)) & 1u)) ? 1.0f : 0.0f;
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
            // one Philox call = 4 consecutive links.  The diagonal (i == j <=> ij is a multiple of N + 1) is found with one
            // exact float division per element, and when L*N*N is a multiple of 4 (every env's block is then 16-byte
            // aligned) the four masks leave as one 16-byte store
            const float rcp_N1 = 1.0f / (float)(N + 1);
            const bool vec = (total & 3) == 0;
            for (int q = sl; q * 4 < total; q += LPE) {
                const u32x4 x = rng.at(site, (uint32_t)q);
                const int hop = fdiv(q * 4, NN, rcp_NN);
                float m[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = q * 4 + e;
                    int ij = k - hop * NN;
                    if (ij >= NN) ij -= NN;                            // the 4 links may straddle a hop boundary
                    const bool diag = fdiv(ij, N + 1, rcp_N1) * (N + 1) == ij;
                    m[e] = ((unit_f32(pick(x, e)) + (diag ? 1.0f : 0.0f)) >= p.ploss) ? 1.0f : 0.0f;
                }
                if (vec) *reinterpret_cast<float4 *>(ch + q * 4) = make_float4(m[0], m[1], m[2], m[3]);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (q * 4 + e < total) ch[q * 4 + e] = m[e];
                }
            }
        }
    } else if (p.channel == CM_CH_GE) {                              // env_communication.py:106-157, GE model :121-150
        float *ch = out.channels ? out.channels + (size_t)b * L * NN : nullptr;
        uint8_t *gs = p.ge_state + (size_t)b * NN;
        const uint32_t site = slot ? SITE_GE_RESET : SITE_GE_STEP;
        // ge_flags bit 0 = loss_apply 0: one transition per ENV STEP shared by all hops (:144-149; at reset the initial
        // state fills every hop, no draws, :108-123) instead of one per hop (:151-156; reset: hop 0 = initial state, then
        // L-1 transitions, :124-141).  bits 1-2 = GE_INIT: good / bad / random (get_init_state, GE model :84-87).
        const bool per_step = p.ge_flags & 1;
        const int init_mode = (p.ge_flags >> 1) & 3;
        const int l0 = slot ? 1 : 0, l1 = per_step ? 1 : L;
        const float bad_rate = (float)((double)p.pgb / ((double)p.pgb + (double)p.pbg));
        for (int k0 = sl * 4; k0 < NN; k0 += LPE * 4) {
            uint8_t s[4];
            if (slot) {
                float ui[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
                if (init_mode == 2) {
                    if (p.rng_mode == CM_RNG_TAPE) { for (int e = 0; e < 4; ++e) ui[e] = tape.ge_init_u[(size_t)b * NN + (k0 + e < NN ? k0 + e : NN - 1)]; }
                    else uniform4(rng, SITE_GE_INIT, (uint32_t)k0, ui);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    s[e] = (k0 + e < NN) ? (uint8_t)(init_mode == 0 ? 1 : (init_mode == 1 ? 0 : (ui[e] >= bad_rate))) : (uint8_t)0;
                if (ch)
                    for (int e = 0; e < 4; ++e) if (k0 + e < NN) ch[k0 + e] = (float)s[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = (k0 + e < NN) ? gs[k0 + e] : (uint8_t)0;
            }
            for (int hop = l0; hop < l1; ++hop) {
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
            if (per_step && ch)                                          // .expand(GCNHops, ...): all hops see one state
                for (int hop = 1; hop < L; ++hop)
                    for (int e = 0; e < 4; ++e) if (k0 + e < NN) ch[(size_t)hop * NN + k0 + e] = (float)s[e];
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
        for (int r = sl; r < S * l.vw; r += LPE) p.visited[(size_t)b * S * l.vw + r] = VIS(l, r);
    }
}

// ---------------------------------------------------------------------------------------
// the step kernel
// ---------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------
// Agent moves of one step for large teams, resolved in parallel instead of N sequential iterations.
// Reference semantics (predator_prey.py:497-500 / coverage.py:330-365): agents act in index order; agent i moves
// iff its target cell is inside the grid and empty AT ITS TURN (neither wall, live prey, nor any agent - agents < i
// at their new cells, agents > i at their old ones).  Hence for a target cell X that is no wall / prey:
//   * X is the start cell of agent j > i            -> i stays (j has not moved yet);
//   * several agents target X                        -> only the lowest index among those not excluded above can
//                                                      ever enter it (if it cannot, X stays occupied for the others);
//   * that lowest contender i moves iff X was empty at the start, or its owner j < i moved away.
// The only dependencies are on LOWER indices (owner of the target), so a few rounds of "look up my owner's
// outcome" settle everything; chains are as long as a queue of agents walking behind each other.
// Leaves positions and the occupancy tile updated; returns the per-step counters.
// ---------------------------------------------------------------------------------------
struct MoveOut { int moving, lazy, cap, rev, pen; };

template <int SCEN, int LPE>
__device__ __forceinline__ MoveOut agents_parallel(const EnvDev &p, const Lds l, const Grp<LPE> g) {
    constexpr int MAXA = 256 / LPE;                     // agents per lane (n_agents <= 255)
    const int S = p.S, N = p.N, sl = g.sl;
    MoveOut mo{ 0, 0, 0, 0, 0 };
    for (int k = sl; k < S * S; k += LPE) { OWN(l, k) = 0; WIN(l, k) = 0xFFFFFFFFu; }
    ENV_SYNC();
    int act[MAXA], r0[MAXA], c0[MAXA], tgt[MAXA], own[MAXA], st[MAXA];     // st: 0 stay, 1 moved, 2 pending on own[]
#pragma unroll
    for (int q = 0; q < MAXA; ++q) {
        const int i = sl + q * LPE;
        act[q] = 4; r0[q] = c0[q] = 0; tgt[q] = -1; own[q] = -1; st[q] = 0;
        if (i < N) { act[q] = ACT(l, i); r0[q] = AR(l, i); c0[q] = AC(l, i); OWN(l, r0[q] * S + c0[q]) = (uint8_t)(i + 1); }
    }
    ENV_SYNC();
#pragma unroll
    for (int q = 0; q < MAXA; ++q) {
        const int i = sl + q * LPE;
        if (i < N && act[q] != 4) {
            const int nr = r0[q] + dr_of(act[q]), nc = c0[q] + dc_of(act[q]);
            if (in_grid(nr, nc, S)) {
                const int X = nr * S + nc, cellv = Gc(l, X);
                if (cellv == C_EMPTY || cellv == C_AGENT) {
                    const int j = (int)OWN(l, X) - 1;                     // -1: empty at the start of the step
                    if (j < i) { tgt[q] = X; own[q] = j; atomicMin(&WIN(l, X), (uint32_t)i); }
                }
            }
        }
    }
    ENV_SYNC();
    bool pend = false;
#pragma unroll
    for (int q = 0; q < MAXA; ++q) {
        const int i = sl + q * LPE;
        if (tgt[q] >= 0) {
            if (WIN(l, tgt[q]) != (uint32_t)i) tgt[q] = -1;               // a lower index owns the claim on this cell
            else st[q] = own[q] < 0 ? 1 : 2;
        }
        if (i < N) ST(l, i) = (uint8_t)st[q];
        pend |= st[q] == 2;
    }
    ENV_SYNC();
    while (g.any(pend)) {                               // follow the owner chains (lower indices settle first)
        pend = false;
#pragma unroll
        for (int q = 0; q < MAXA; ++q) {
            if (st[q] == 2) {
                const int sj = ST(l, own[q]);
                if (sj != 2) { st[q] = sj; ST(l, sl + q * LPE) = (uint8_t)sj; }
                pend |= st[q] == 2;
            }
        }
        ENV_SYNC();
    }
    // counters + the visited test against the start-of-step bitmap (a cell entered this step is occupied, so nobody
    // else can "see" it visited or unvisited afterwards)
    bool seen[MAXA];
#pragma unroll
    for (int q = 0; q < MAXA; ++q) {
        const int i = sl + q * LPE;
        const bool is = i < N, moved = st[q] == 1;
        seen[q] = false;
        if (SCEN == CM_CO && moved) { const int X = tgt[q], nr = fdiv(X, S, 1.0f / (float)S); seen[q] = (VISW(l, nr, X - nr * S) >> ((X - nr * S) & 31)) & 1u; }
        mo.moving += g.count(is && act[q] != 4);
        mo.lazy += g.count(is && act[q] == 4);
        mo.pen += g.count(is && act[q] != 4 && !moved);
        mo.cap += g.count(moved && !seen[q]);
        mo.rev += g.count(moved && seen[q]);
    }
    ENV_SYNC();
#pragma unroll
    for (int q = 0; q < MAXA; ++q) if (st[q] == 1) Gc(l, r0[q] * S + c0[q]) = C_EMPTY;       // leave ...
    ENV_SYNC();
#pragma unroll
    for (int q = 0; q < MAXA; ++q) {
        if (st[q] == 1) {                                                                    // ... then enter
            const int X = tgt[q], nr = fdiv(X, S, 1.0f / (float)S), nc = X - nr * S;
            Gc(l, X) = C_AGENT;
            AR(l, sl + q * LPE) = (int16_t)nr; AC(l, sl + q * LPE) = (int16_t)nc;
            if (SCEN == CM_CO) atomicOr(&VISW(l, nr, nc), vbit(nc));
        }
    }
    ENV_SYNC();
    return mo;
}

// One env per LPE-lane group.  `grp` = group index inside the workgroup (LDS slot), `b_raw` = env index (groups with
// b_raw >= p.B or !grp_live shadow the last env and never commit), `lds_base` = byte offset of the env area in the
// dynamic LDS block, `act_lds` = optional [N] action bytes already in LDS for this env (fused rollout kernel), else
// the actions come from `actions` in HBM.  All synchronisation is wave-local (ENV_SYNC), so the body runs unchanged
// inside the single-wave env kernel and inside the 4-wave fused rollout workgroup.
// ---------------------------------------------------------------------------------------
// Small PP teams (n_agents, n_preys <= R <= 8, several envs per wave): the order-dependent part of the step with every
// position in registers.  The tile walk below costs three dependent LDS round trips per agent / prey (action -> position
// -> target cell, then the commit) - 11.5 k of the env step's 25 k clocks at the headline shape (ENV_PROBE).  Here every
// lane of an env's group holds ALL its agents and preys and replays the sequential loops redundantly: "is the target
// cell empty" becomes <= 2R register compares (a cell holds at most one entity), "predators / preys next to (r, c)" a
// count of entities at Manhattan distance 1 (exactly the four bounds-checked neighbour cells of count_adj).  Only the
// prey trials stay one lane per prey (Philox per prey), their results crossing lanes once through LDS.
// In: AR / AC / ACT / PR / PC / ALV staged in LDS, tile all C_EMPTY.  Out: the same arrays and the tile at their
// end-of-step values, as the tile walk leaves them.  Reference lines as in env_body below.
// ---------------------------------------------------------------------------------------
struct SmallOut { int moving, capture, penalty, wsum; bool tape_short, any_alive; };

// Straight-line code throughout (selects, no short-circuit operators: hipcc turns every `&&` on lane data into an
// exec-mask branch), and no long-lived booleans (each one is a 64-bit lane mask in scalar registers: an `alive[R]` array
// spills them).  Coordinates are held +1 so that a target one step outside the grid stays non-negative and |a - b| is one
// v_sad_u32; absent entities AND dead preys sit at (100, 100): never equal to, never next to anything real, so liveness
// needs no flag of its own.
constexpr int GONE = 100;
// |a - b| + c in one instruction (hipcc expands __usad into max / min / sub / add)
__device__ __forceinline__ unsigned sad(unsigned a, unsigned b, unsigned c) {
    unsigned d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned sad0(unsigned a, unsigned b) {
    unsigned d;
    asm("v_sad_u32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ unsigned dist1(int r0, int c0, int r1, int c1) {  // Manhattan distance
    return sad((unsigned)c0, (unsigned)c1, sad0((unsigned)r0, (unsigned)r1));
}
// 1 if no entity of either layer stands on (r, c)
template <int R>
__device__ __forceinline__ int cell_free(const int (&ar)[R], const int (&ac)[R], const int (&pr)[R], const int (&pc)[R], int r, int c) {
    unsigned m = 1u;
#pragma unroll
    for (int k = 0; k < R; ++k) m = min(m, min(dist1(ar[k], ac[k], r, c), dist1(pr[k], pc[k], r, c)));
    return (int)m;
}
// number of entities of one layer at distance exactly 1 from (r, c)
template <int R>
__device__ __forceinline__ int next_to(const int (&er)[R], const int (&ec)[R], int r, int c) {
    int n = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        unsigned e;                                    // |d - 1|: 0 exactly at distance 1
        asm("v_sad_u32 %0, %1, 1, 0" : "=v"(e) : "v"(dist1(er[k], ec[k], r, c)));
        n += 1 - (int)min(e, 1u);
    }
    return n;
}

template <int LPE, int R>
__device__ __forceinline__ SmallOut pp_small_step(const EnvDev &p, const Lds l, const Rng &rng, const cm_rng_tape &tape, int b,
                                                  const Grp<LPE> g) {
    const int N = p.N, M = p.M, S = p.S, sl = g.sl;
    int ar[R], ac[R], act[R], pr[R], pc[R];
    {
        // every array starts on a 16-byte boundary of its own 16-byte-rounded slot (lds_take): the <= 8 int16 / u8 entries of
        // one array are ONE wide read (entries past N / M are padding and masked below)
        uint32_t w_ar[R / 2], w_ac[R / 2], w_pr[R / 2], w_pc[R / 2], w_act[R / 4], w_alv[R / 4];
        if constexpr (R == 8) {
            const uint4 a = *reinterpret_cast<const uint4 *>(smem + l.ar), c = *reinterpret_cast<const uint4 *>(smem + l.ac);
            const uint4 q = *reinterpret_cast<const uint4 *>(smem + l.pr), d = *reinterpret_cast<const uint4 *>(smem + l.pc);
            const uint2 t = *reinterpret_cast<const uint2 *>(smem + l.act), v = *reinterpret_cast<const uint2 *>(smem + l.alive);
            w_ar[0] = a.x; w_ar[1] = a.y; w_ar[2] = a.z; w_ar[3] = a.w; w_ac[0] = c.x; w_ac[1] = c.y; w_ac[2] = c.z; w_ac[3] = c.w;
            w_pr[0] = q.x; w_pr[1] = q.y; w_pr[2] = q.z; w_pr[3] = q.w; w_pc[0] = d.x; w_pc[1] = d.y; w_pc[2] = d.z; w_pc[3] = d.w;
            w_act[0] = t.x; w_act[1] = t.y; w_alv[0] = v.x; w_alv[1] = v.y;
        } else {
            const uint2 a = *reinterpret_cast<const uint2 *>(smem + l.ar), c = *reinterpret_cast<const uint2 *>(smem + l.ac);
            const uint2 q = *reinterpret_cast<const uint2 *>(smem + l.pr), d = *reinterpret_cast<const uint2 *>(smem + l.pc);
            w_ar[0] = a.x; w_ar[1] = a.y; w_ac[0] = c.x; w_ac[1] = c.y; w_pr[0] = q.x; w_pr[1] = q.y; w_pc[0] = d.x; w_pc[1] = d.y;
            w_act[0] = *reinterpret_cast<const uint32_t *>(smem + l.act); w_alv[0] = *reinterpret_cast<const uint32_t *>(smem + l.alive);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int sh = (i & 1) * 16, sb = (i & 3) * 8;
            const int r = (int)(int16_t)(w_ar[i / 2] >> sh), c = (int)(int16_t)(w_ac[i / 2] >> sh);
            const int qr = (int)(int16_t)(w_pr[i / 2] >> sh), qc = (int)(int16_t)(w_pc[i / 2] >> sh);
            const int a = (w_act[i / 4] >> sb) & 0xff, alive = (w_alv[i / 4] >> sb) & 0xff;
            ar[i] = i < N ? r + 1 : GONE; ac[i] = i < N ? c + 1 : GONE; act[i] = i < N ? a : 4;
            pr[i] = ((i < M) & (alive != 0)) ? qr + 1 : GONE; pc[i] = ((i < M) & (alive != 0)) ? qc + 1 : GONE;
        }
    }
    const bool mine = sl < M;                          // this lane's prey for the trials: start-of-step values
    int my_r, my_c;
    {
        const int ip = mine ? sl : 0;
        const int qr = PR(l, ip), qc = PC(l, ip), alive = ALV(l, ip);
        my_r = (mine & (alive != 0)) ? qr + 1 : GONE; my_c = (mine & (alive != 0)) ? qc + 1 : GONE;
    }
    const bool taped = p.rng_mode == CM_RNG_TAPE;
    // the prey's first four trial words depend on nothing the step computes: issued under the LDS latency
    u32x4 x0 = { 0, 0, 0, 0 };
    if (mine && !taped) x0 = rng.at(SITE_PREY, (uint32_t)(2 * sl));
    ENV_SYNC();

    SmallOut o{ 0, 0, 0, 0, false, false };
    // ---- agents move in index order (predator_prey.py:497-500, :240-261) ----
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int a = act[i];
        const int active = (i < N) & (a != 4);         // pseudo-action 5 (faulty agent): target = own cell, rejected by k == i
        o.moving += active;
        const int nr = ar[i] + dr_of(a), nc = ac[i] + dc_of(a);
        const int ok = active & ((unsigned)(nr - 1) < (unsigned)S) & ((unsigned)(nc - 1) < (unsigned)S) & cell_free<R>(ar, ac, pr, pc, nr, nc);
        ar[i] = ok ? nr : ar[i]; ac[i] = ok ? nc : ac[i];
    }
    ENV_PROBE(3);
    // ---- per-prey work against the (now static) agent layer: one lane per prey (:396-407) ----
    {
        const int my_alive = my_r != GONE;
        const int cnt = next_to<R>(ar, ac, my_r, my_c);                      // 0 for a dead prey
        int mv = 4;
        int found = (my_alive ^ 1) | ((p.load == 2) & (cnt >= 2));          // dead, or captured this step: no trial
        if (taped) {
            for (int t = 0; t < 5 && !found; ++t) {
                const int m = tape.prey[((size_t)b * M + sl) * 5 + t];
                if (m > 4) { mv = 4 | 8; break; }      // recorded tape ended: only legal if prey gets captured
                if (next_to<R>(ar, ac, my_r + dr_of(m), my_c + dc_of(m)) == 0) { mv = m; found = 1; }
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) {              // first of <= 5 draws whose target has no predator neighbour
                const int m = prey_move_from_u32(pick(x0, t));
                const int ok = (next_to<R>(ar, ac, my_r + dr_of(m), my_c + dc_of(m)) == 0) & (found ^ 1);
                mv = ok ? m : mv; found |= ok;
            }
            if (!found) {                              // fifth draw: second Philox call, rare
                const u32x4 x1 = rng.at(SITE_PREY, (uint32_t)(2 * sl + 1));
                const int m = prey_move_from_u32(x1.x);
                mv = next_to<R>(ar, ac, my_r + dr_of(m), my_c + dc_of(m)) == 0 ? m : mv;
            }
        }
        if (mine) { PCNT(l, sl) = (uint8_t)cnt; PMV(l, sl) = (uint8_t)mv; }
    }
    ENV_PROBE(8);
    // prey_watching (:419-423): agents 4-adjacent to a live prey (prey layer at start-of-phase positions)
    int my_ar = GONE, my_ac = GONE;                    // this lane's agent, for the count and the write-back
#pragma unroll
    for (int i = 0; i < R; ++i) { my_ar = sl == i ? ar[i] : my_ar; my_ac = sl == i ? ac[i] : my_ac; }
    o.wsum = g.count(next_to<R>(pr, pc, my_ar, my_ac) != 0);                 // absent agents sit next to nothing
    ENV_SYNC();
    int pcnt[R], pmv[R];
    {
        uint32_t w_cnt[R / 4], w_mv[R / 4];
        if constexpr (R == 8) {
            const uint2 a = *reinterpret_cast<const uint2 *>(smem + l.pcnt), c = *reinterpret_cast<const uint2 *>(smem + l.pmv);
            w_cnt[0] = a.x; w_cnt[1] = a.y; w_mv[0] = c.x; w_mv[1] = c.y;
        } else {
            w_cnt[0] = *reinterpret_cast<const uint32_t *>(smem + l.pcnt); w_mv[0] = *reinterpret_cast<const uint32_t *>(smem + l.pmv);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int sb = (j & 3) * 8;
            pcnt[j] = j < M ? (int)((w_cnt[j / 4] >> sb) & 0xff) : 0; pmv[j] = j < M ? (int)((w_mv[j / 4] >> sb) & 0xff) : 4;
        }
    }
    ENV_SYNC();
    ENV_PROBE(4);
    // ---- captures + prey moves in index order (:416-432 / :460-478, :276-301) ----
    int tshort = 0, alive_any = 0;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int r = pr[j], c = pc[j], cnt = pcnt[j], mvb = pmv[j];
        const int alive = r != GONE;
        int need = p.load;
        if (p.load != 2) {                                                       // reward_individual :467-470 (uniform branch)
            const int on_r = (r == 1) | (r == S), on_c = (c == 1) | (c == S);
            const int adj = (on_r & on_c) ? 2 : ((on_r | on_c) ? 3 : p.load);    // __create_edges :123-144
            const int avail = adj - next_to<R>(pr, pc, r, c);
            need = p.load < avail ? p.load : avail;
        }
        const int hit = alive & (cnt >= 1), captured = hit & (need <= cnt);
        o.capture += captured; o.penalty += hit & (captured ^ 1);
        const int stays = alive & (captured ^ 1);                                // :301 otherwise
        tshort |= stays & ((mvb >> 3) & 1);
        const int mv = mvb & 7;
        const int nr = r + dr_of(mv), nc = c + dc_of(mv);
        const int ok = stays & (mv != 4) & ((unsigned)(nr - 1) < (unsigned)S) & ((unsigned)(nc - 1) < (unsigned)S) &
                       cell_free<R>(ar, ac, pr, pc, nr, nc);
        pr[j] = stays ? (ok ? nr : r) : GONE; pc[j] = stays ? (ok ? nc : c) : GONE;
        alive_any |= stays;
    }
    o.tape_short = tshort != 0; o.any_alive = alive_any != 0;
    // ---- end-of-step state back to LDS: lane i its agent and its prey, then the tile marks ----
    int my_pr = GONE, my_pc = GONE;
#pragma unroll
    for (int j = 0; j < R; ++j) { my_pr = sl == j ? pr[j] : my_pr; my_pc = sl == j ? pc[j] : my_pc; }
    if (sl < N) { AR(l, sl) = (int16_t)(my_ar - 1); AC(l, sl) = (int16_t)(my_ac - 1); Gc(l, (my_ar - 1) * S + my_ac - 1) = C_AGENT; }
    if (mine) {
        const bool my_al = my_pr != GONE;
        ALV(l, sl) = (uint8_t)my_al;
        if (my_al) { PR(l, sl) = (int16_t)(my_pr - 1); PC(l, sl) = (int16_t)(my_pc - 1); Gc(l, (my_pr - 1) * S + my_pc - 1) = C_PREY; }
    }
    ENV_SYNC();
    return o;
}

// Everything the step reads from HBM that does not depend on the actions: a caller with other work in front of the env
// step (the fused rollout kernel: the whole policy forward) requests it first and hands it over in registers, so the
// env phase starts without a memory round trip.  One lane = one agent and one prey (teams <= lanes per env).
struct EnvPre {
    uint32_t rng_step;
    int step_count_in, succ;
    float t_row, t_col, t_step0, t_step;
    double t_rew;
    int ax, ay, px, py;
    int cond, alive;
};

template <int SCEN, int LPE>
__host__ __device__ __forceinline__ bool env_prefetch_ok(const EnvDev &p) {
    return SCEN == CM_PP && LPE < 64 && p.N <= LPE && p.M <= LPE && p.S <= LPE && (p.M + 1) + (p.N + 1) <= LPE;
}

template <int SCEN, int LPE>
__device__ __forceinline__ EnvPre env_prefetch(const EnvDev &p, int b_raw, bool grp_live) {
    const int sl = thread_x() % LPE;
    const int b = (grp_live && b_raw < p.B) ? b_raw : p.B - 1;
    const int S = p.S, N = p.N, M = p.M;
    EnvPre e;
    e.step_count_in = p.step_count[b];
    e.succ = p.success[b];
    e.rng_step = p.rng_step[b];
    const int s0 = sl < S ? sl : S - 1;
    e.t_row = p.lut_row[s0]; e.t_col = p.lut_col[s0]; e.t_step0 = p.lut_step[0];
    e.t_rew = p.rew_lut[sl < (M + 1) + (N + 1) ? sl : 0];
    const int ia = sl < N ? sl : 0, ip = sl < M ? sl : 0;
    { const int2 q = p.agent_pos[(size_t)b * N + ia]; e.ax = q.x; e.ay = q.y; }
    e.cond = p.agent_cond[(size_t)b * N + ia];
    { const int2 q = p.prey_pos[(size_t)b * M + ip]; e.px = q.x; e.py = q.y; }
    e.alive = p.alive[(size_t)b * M + ip];
    const int sc = e.step_count_in + 1;
    e.t_step = p.lut_step[sc <= p.max_steps ? sc : p.max_steps];
    return e;
}

// The prefetched agent / prey rows into the env's LDS arrays (what env_body's own load loops do), one lane each.
// Returns the lane's bad-action flag.
template <int SCEN, int LPE>
__device__ __forceinline__ bool env_stage(const EnvDev &p, const EnvPre &e, const int32_t *act_lds, int grp, int lds_base) {
    const int sl = thread_x() % LPE;
    const Lds l = make_lds(p.S, p.N, p.M, lds_base + p.lds_env * grp, p.status);
    bool bad = false;
    if (sl < p.N) {
        AR(l, sl) = (int16_t)e.ax; AC(l, sl) = (int16_t)e.ay;
        const int a = act_lds[sl];
        bad = (unsigned)a > 4u;
        const bool faulty = SCEN == CM_PP && e.cond == 0;             // pseudo-action 5, as in env_body
        ACT(l, sl) = (uint8_t)(bad ? 4 : ((faulty && a != 4) ? 5 : a));
    }
    if (SCEN == CM_PP && sl < p.M) { PR(l, sl) = (int16_t)e.px; PC(l, sl) = (int16_t)e.py; ALV(l, sl) = (uint8_t)e.alive; }
    return bad;
}

// The next step's EnvPre from what the step just taken left behind: positions / alive flags in the env's LDS arrays (after the
// auto-reset), the scalars in `carry`; the lane's table entries (row, col, reward terms) are launch constants and stay.  What
// env_prefetch would have loaded from the global state arrays - written by this very wave a moment ago - is never read.
//   agent_condition: do_reset re-arms every agent of a finished env (:152), nothing else changes it inside a launch
//   rng_step       : advanced by one per step (env_body: p.rng_step[b] = rng.step + 1)
template <int SCEN, int LPE>
__device__ __forceinline__ EnvPre env_pre_carry(const EnvDev &p, const EnvPre &prev, const EnvCarry &carry, int grp, int lds_base) {
    const int sl = thread_x() % LPE;
    const Lds l = make_lds(p.S, p.N, p.M, lds_base + p.lds_env * grp, p.status);
    EnvPre e = prev;
    e.step_count_in = carry.step_count;
    e.succ = carry.succ;
    e.rng_step = prev.rng_step + 1u;
    const int ia = sl < p.N ? sl : 0, ip = sl < p.M ? sl : 0;
    e.ax = AR(l, ia); e.ay = AC(l, ia);
    e.cond = carry.done ? 1 : prev.cond;
    e.px = PR(l, ip); e.py = PC(l, ip); e.alive = ALV(l, ip);
    const int sc = e.step_count_in + 1;
    e.t_step = p.lut_step[sc <= p.max_steps ? sc : p.max_steps];
    return e;
}

template <int SCEN, int LPE>
__device__ __forceinline__ void env_body(const EnvDev &p, const int32_t *__restrict__ actions, const int32_t *act_lds,
                                         const cm_rng_tape &tape, const cm_step_out &out, int reset_only, int grp, int b_raw,
                                         bool grp_live, int lds_base, int *defer = nullptr,
                                         // PRE: state already staged in LDS by the caller (env_stage) and the scalars below prefetched
                                         bool PRE = false, uint32_t pre_rng_step = 0, int pre_step_count_in = 0, int pre_succ = 0,
                                         float pre_t_row = 0.0f, float pre_t_col = 0.0f, float pre_t_step0 = 0.0f,
                                         float pre_t_step = 0.0f, double pre_t_rew = 0.0,
                                         bool pre_bad = false,
                                         bool all_valid = false /* the caller vouches: every group has an env of its own below p.B */,
                                         EnvCarry *carry = nullptr, int obs_copy = -1) {
    Grp<LPE> g;
    const int tx = thread_x();
    g.sub = (tx & (WAVE - 1)) / LPE; g.sl = tx % LPE;
    const int sl = g.sl;
    ENV_PROBE(0);
    const bool valid = all_valid || (grp_live && b_raw < p.B);
    const int b = valid ? b_raw : p.B - 1;            // idle groups shadow the last env and never commit
    const Lds l = make_lds(p.S, p.N, p.M, lds_base + p.lds_env * grp, p.status);
    const int S = p.S, N = p.N, M = p.M;
    Rng rng{ (uint32_t)(p.env_id_offset + b), PRE ? pre_rng_step : p.rng_step[b], p.key0, p.key1 };
    cm_step_out o = out;
    if (!valid) { o.obs = nullptr; o.dist_adj = nullptr; o.channels = nullptr; }

    if (reset_only) {
        do_reset<SCEN, LPE>(p, l, rng, tape, b, g, true);
        if (!valid) return;
        if (sl == 0) { p.step_count[b] = 0; if (SCEN == CM_CO) p.total_capture[b] = 0; p.rng_step[b] = rng.step + 1; }
        if (defer) { if (sl == 0) { defer[1] = 0; defer[2] = 1; defer[3] = (int)rng.step; defer[0] = 1; } return; }
        emit<SCEN, LPE>(p, l, rng, tape, o, b, g, 0, 1);
        return;
    }

    const int step_count_in = PRE ? pre_step_count_in : p.step_count[b];   // requested first: the clock entry below depends on it
    int succ = PRE ? pre_succ : p.success[b];
    // tables the end of the step needs, requested now (ObsTabs; the reward terms likewise: lane s holds rew_lut[s])
    const bool tabs_held = PRE || S <= LPE;
    const bool rew_held = PRE || (SCEN == CM_PP && LPE < 64 && (M + 1) + (N + 1) <= LPE);
    float t_row = pre_t_row, t_col = pre_t_col, t_step0 = pre_t_step0;
    double t_rew = pre_t_rew;
    if (!PRE) {
        if (tabs_held) { const int s0 = sl < S ? sl : S - 1; t_row = p.lut_row[s0]; t_col = p.lut_col[s0]; t_step0 = p.lut_step[0]; }
        if (rew_held) t_rew = p.rew_lut[sl < (M + 1) + (N + 1) ? sl : 0];
    }
    // ---- load SoA state, rebuild the occupancy tile in LDS ----
    bool bad_action = pre_bad;
    if (!PRE) {
    for (int i = sl; i < N; i += LPE) {
        const int2 q = p.agent_pos[(size_t)b * N + i];
        AR(l, i) = (int16_t)q.x; AC(l, i) = (int16_t)q.y;
        const int a = act_lds ? act_lds[i] : actions[(size_t)b * N + i];
        const bool bad = (unsigned)a > 4u;
        bad_action |= bad;
        // agent_condition gate (predator_prey.py:257-261): a faulty agent's move is computed but not applied - it counts
        // as a moving agent and stays put.  Encoded as the pseudo-action 5 (no displacement): its "target" is its own
        // occupied cell, which every move resolver below rejects.
        const bool faulty = SCEN == CM_PP && p.agent_cond[(size_t)b * N + i] == 0;
        ACT(l, i) = (uint8_t)(bad ? 4 : ((faulty && a != 4) ? 5 : a));
    }
    if (SCEN == CM_PP)
        for (int j = sl; j < M; j += LPE) {
            const int2 q = p.prey_pos[(size_t)b * M + j];
            PR(l, j) = (int16_t)q.x; PC(l, j) = (int16_t)q.y;
            ALV(l, j) = p.alive[(size_t)b * M + j];
        }
    }
    for (int k = sl; k < S * S; k += LPE) Gc(l, k) = (SCEN == CM_CO) ? p.base_grid[k] : (uint8_t)C_EMPTY;
    if (SCEN == CM_CO) for (int r = sl; r < S * l.vw; r += LPE) VIS(l, r) = p.visited[(size_t)b * S * l.vw + r];
    // the reference raises on a bad action (predator_prey.py:255): flag it; the env is left untouched
    const bool env_bad = g.any(bad_action);
    if (env_bad && sl == 0 && valid) raise(p, CM_ERR_ACTION);
    const bool commit = valid && !env_bad;
    ENV_SYNC();
    ENV_PROBE(1);
    if (p.stop == 1) return;
    // small PP teams keep their positions in registers (pp_small_step) and mark the tile once, at the end of the step
    const bool small = SCEN == CM_PP && LPE < 64 && N <= 8 && M <= 8 && M <= LPE && !p.no_small;
    if (!small) {
        for (int i = sl; i < N; i += LPE) Gc(l, AR(l, i) * S + AC(l, i)) = C_AGENT;
        if (SCEN == CM_PP)
            for (int j = sl; j < M; j += LPE) if (ALV(l, j)) Gc(l, PR(l, j) * S + PC(l, j)) = C_PREY;
        ENV_SYNC();
    }

    ENV_PROBE(2);
    if (p.stop == 2) return;
    int step_count = step_count_in + 1;
    float t_step = pre_t_step;
    if (!PRE && tabs_held) t_step = p.lut_step[step_count <= p.max_steps ? step_count : p.max_steps];
    int done = 0;
    double reward;
    int det0 = 0, det1 = 0, det2 = 0, det3 = 0, det4 = 0, det5 = 0;

    if (SCEN == CM_PP) {
        // ---- agents move in index order (predator_prey.py:497-500, :240-261) ----
        int moving = 0;
        int capture = 0, penalty = 0, wsum = 0;
        bool tape_short = false, any_alive = false;
        if (LPE < 64 && small) {
            const SmallOut so = (N <= 4 && M <= 4) ? pp_small_step<LPE, 4>(p, l, rng, tape, b, g) : pp_small_step<LPE, 8>(p, l, rng, tape, b, g);
            moving = so.moving; capture = so.capture; penalty = so.penalty; wsum = so.wsum; tape_short = so.tape_short;
            any_alive = so.any_alive;
            ENV_PROBE(5);
        } else {
        if (N > PAR_AGENTS_MIN) moving = agents_parallel<SCEN, LPE>(p, l, g).moving;      // large teams: parallel resolution
        else for (int i = 0; i < N; ++i) {                                                // small teams: group-uniform loop
            const int a = ACT(l, i);
            bool mv = false;
            int r = 0, c = 0, nr = 0, nc = 0;
            if (a != 4) {
                ++moving;
                r = AR(l, i); c = AC(l, i); nr = r + dr_of(a); nc = c + dc_of(a);
                mv = in_grid(nr, nc, S) && Gc(l, nr * S + nc) == C_EMPTY;
            }
            ENV_SYNC();
            if (mv && sl == 0) { Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_AGENT; AR(l, i) = (int16_t)nr; AC(l, i) = (int16_t)nc; }
            ENV_SYNC();
        }
        ENV_PROBE(3);
        if (p.stop == 3) return;
        if (LPE == 64) {
        // ---- one wave per env: the per-prey inputs of the sequential loop (alive, position, predator count, chosen
        // move - all start-of-phase values, nothing an earlier prey can change) stay in the registers of the lane that
        // computed them; iteration j fetches them with v_readlane (no LDS round trip) and issues its five tile probes
        // together, so an iteration is ONE LDS latency instead of four dependent ones ----
        constexpr int MAXP = 4;                              // n_preys <= 255
        int pk[MAXP];                                        // alive | r << 1 | c << 7 | cnt << 13 | move(+8) << 17
#pragma unroll
        for (int q = 0; q < MAXP; ++q) {
            const int j = sl + q * LPE;
            int cnt = 0, mv = 4, alive = 0, r = 0, c = 0;
            if (j < M && ALV(l, j)) {
                alive = 1; r = PR(l, j); c = PC(l, j);
                cnt = count_adj(l, r, c, S, C_AGENT);
                const bool captured_now = (p.load == 2) && cnt >= 1 && p.load <= cnt;
                if (!captured_now) {                          // prey_random_move (:396-407)
                    bool found = false;
                    u32x4 x = { 0, 0, 0, 0 };
                    for (int t = 0; t < 5 && !found; ++t) {
                        int m;
                        if (p.rng_mode == CM_RNG_TAPE) {
                            m = tape.prey[((size_t)b * M + j) * 5 + t];
                            if (m > 4) { mv = 4 | 8; break; }
                        } else {
                            if ((t & 3) == 0) x = rng.at(SITE_PREY, (uint32_t)(2 * j + (t >> 2)));
                            m = prey_move_from_u32(pick(x, t & 3));
                        }
                        if (count_adj(l, r + dr_of(m), c + dc_of(m), S, C_AGENT) == 0) { mv = m; found = true; }
                    }
                }
            }
            pk[q] = alive | (r << 1) | (c << 7) | (cnt << 13) | (mv << 17);
        }
        for (int i0 = 0; i0 < N; i0 += LPE) {                 // prey_watching (:419-423)
            const int i = i0 + sl;
            const bool w = i < N && count_adj(l, AR(l, i), AC(l, i), S, C_PREY) > 0;
            wsum += g.count(w);
        }
        ENV_SYNC();
        ENV_PROBE(4);
        if (p.stop == 4) return;
        for (int j = 0; j < M; ++j) {
            const int lanej = j & (LPE - 1), qj = j >> 6;
            const int v0 = __builtin_amdgcn_readlane(pk[0], lanej), v1 = __builtin_amdgcn_readlane(pk[1], lanej),
                      v2 = __builtin_amdgcn_readlane(pk[2], lanej), v3 = __builtin_amdgcn_readlane(pk[3], lanej);
            const int v = qj == 0 ? v0 : (qj == 1 ? v1 : (qj == 2 ? v2 : v3));
            if (!(v & 1)) continue;                            // dead before this step (uniform)
            const int r = (v >> 1) & 63, c = (v >> 7) & 63, cnt = (v >> 13) & 15, mvb = (v >> 17) & 15, mv = mvb & 7;
            const int nr = r + dr_of(mv), nc = c + dc_of(mv);
            const int npre = count_adj(l, r, c, S, C_PREY);                     // 4 probes + the target probe: one round trip
            const int tcell = cell(l, nr, nc, S);
            bool captured = false, moved = false;
            if (cnt >= 1) {
                int need = p.load;
                if (p.load != 2) {                                               // reward_individual :467-470
                    const bool on_r = (r == 0 || r == S - 1), on_c = (c == 0 || c == S - 1);
                    const int adj = (on_r && on_c) ? 2 : ((on_r || on_c) ? 3 : p.load);   // __create_edges :123-144
                    const int avail = adj - npre;
                    need = p.load < avail ? p.load : avail;
                }
                if (need <= cnt) { captured = true; ++capture; } else ++penalty;
            }
            if (!captured) {
                if (mvb & 8) tape_short = true;
                moved = mv != 4 && tcell == C_EMPTY;                             // cell() is -1 outside the grid
            }
            if (sl == 0) {
                if (captured) { ALV(l, j) = 0; Gc(l, r * S + c) = C_EMPTY; }      // :301
                else if (moved) { Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_PREY; PR(l, j) = (int16_t)nr; PC(l, j) = (int16_t)nc; }
            }
            ENV_SYNC();
        }
        } else {
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
        ENV_PROBE(8);
        // prey_watching (:419-423): agents 4-adjacent to a live prey (prey layer still at start-of-phase positions)
        for (int i0 = 0; i0 < N; i0 += LPE) {
            const int i = i0 + sl;
            const bool w = i < N && count_adj(l, AR(l, i), AC(l, i), S, C_PREY) > 0;
            wsum += g.count(w);
        }
        ENV_SYNC();
        ENV_PROBE(4);
        if (p.stop == 4) return;
        // ---- captures + prey moves in index order (:416-432 / :460-478, :276-301): group-uniform loop ----
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
            ENV_SYNC();
            if (sl == 0) {
                if (captured) { ALV(l, j) = 0; Gc(l, r * S + c) = C_EMPTY; }      // :301
                else if (moved) { Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_PREY; PR(l, j) = (int16_t)nr; PC(l, j) = (int16_t)nc; }
            }
            ENV_SYNC();
        }
        }   // LPE != 64
        for (int j0 = 0; j0 < M; j0 += LPE) any_alive |= g.any(j0 + sl < M && ALV(l, j0 + sl));
        ENV_PROBE(5);
        }   // tile walk
        if (p.stop == 5) return;
        if (tape_short && sl == 0 && commit) raise(p, CM_ERR_TAPE_PREY);
        // reward in f64 exactly as the Python expression evaluates (:434 / :480); no FMA contraction (build flag)
        // (step + cap*c) + (mc*m)/N [+ pen*p]: the two count-indexed terms come from host tables built with the
        // same f64 operations (no f64 division on the device)
        if (rew_held) reward = __shfl(t_rew, capture, LPE) + __shfl(t_rew, (M + 1) + moving, LPE);
        else reward = p.rew_lut[capture] + p.rew_lut[(M + 1) + moving];
        if (p.load == 2) reward = reward + p.penalty * (double)penalty;
        det0 = capture; det1 = moving; det2 = penalty; det4 = wsum;
        if (o.prey_alive && commit) for (int j = sl; j < M; j += LPE) o.prey_alive[(size_t)b * M + j] = ALV(l, j);
        done = (step_count >= p.max_steps) || !any_alive;               // :511-517
        if (done) succ = any_alive ? 0 : 1;
    } else {
        // ---- Coverage.step (:319-378): sequential agents against tile + visited bitmap ----
        int cap = 0, mov = 0, pen = 0, lazy = 0, rev = 0;
        if (N > PAR_AGENTS_MIN) {
            const MoveOut mo = agents_parallel<SCEN, LPE>(p, l, g);
            cap = mo.cap; mov = mo.moving; pen = mo.pen; lazy = mo.lazy; rev = mo.rev;
        } else for (int i = 0; i < N; ++i) {
            const int a = ACT(l, i);
            bool mv = false, seen = false;
            int r = 0, c = 0, nr = 0, nc = 0;
            if (a == 4) ++lazy;
            else {
                ++mov;
                r = AR(l, i); c = AC(l, i); nr = r + dr_of(a); nc = c + dc_of(a);
                mv = in_grid(nr, nc, S) && Gc(l, nr * S + nc) == C_EMPTY;
                if (mv) { seen = (VISW(l, nr, nc) >> (nc & 31)) & 1u; if (seen) ++rev; else ++cap; }
                else ++pen;
            }
            ENV_SYNC();
            if (mv && sl == 0) {
                VISW(l, nr, nc) |= vbit(nc);
                Gc(l, r * S + c) = C_EMPTY; Gc(l, nr * S + nc) = C_AGENT; AR(l, i) = (int16_t)nr; AC(l, i) = (int16_t)nc;
            }
            ENV_SYNC();
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
            int2 *dd = reinterpret_cast<int2 *>(o.details + (size_t)b * 6);      // 24-byte rows: three 8-byte stores
            dd[0] = make_int2(det0, det1); dd[1] = make_int2(det2, det3); dd[2] = make_int2(det4, det5);
        }
        p.rng_step[b] = rng.step + 1;
    }
    ENV_SYNC();
    ENV_PROBE(6);
    if (p.stop == 6) return;
    // auto-reset (:36-43): groups whose env finished re-spawn and emit the reset observation
    do_reset<SCEN, LPE>(p, l, rng, tape, b, g, done != 0);
    if (done) step_count = 0;
    ENV_PROBE(7);
    if (!commit || p.stop == 7) return;
    if (sl == 0) {
        if (done && SCEN == CM_CO) p.total_capture[b] = 0;
        p.step_count[b] = step_count;
        p.success[b] = succ;
        if (o.success) o.success[b] = succ;
    }
    // wide kernel: the emission (order-independent, the bulk of the instructions for large teams) is done by all the
    // workgroup's waves after this wave has left the state in LDS; hand over (step count, slot, Philox step)
    if (defer) { if (sl == 0) { defer[1] = step_count; defer[2] = done ? 1 : 0; defer[3] = (int)rng.step; defer[0] = 1; } return; }
    if (carry) { carry->step_count = step_count; carry->succ = succ; carry->done = done; }
    emit<SCEN, LPE>(p, l, rng, tape, o, b, g, step_count, done ? 1 : 0, ObsTabs{ tabs_held, t_row, t_col, done ? t_step0 : t_step }, obs_copy);
    ENV_PROBE(9);
}


}  // namespace cm
