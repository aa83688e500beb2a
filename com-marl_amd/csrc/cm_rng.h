// cm_rng.h - counter-based Philox4x32-10 stream of the production path.
//
// counter = (global env id, rng_step, site, idx), key = (seed lo, seed hi):
//   site 1 spawn candidate (idx = attempt, x0 -> row, x1 -> col)
//   site 2 prey move trial  (idx = 2*prey + trial/4, component trial%4)
//   site 3/4 IID uniforms   (step / reset comm update; idx = flat/4, component flat%4)
//   site 5/6 GE uniforms    (step / reset; flat = (2*hop + which)*N*N + link)
//   site 7 action sample    (rng_step := policy_step, idx = agent, x0)
//   site 9 agent faults     (rng_step := fault_step; iid: idx = agent/4, component agent%4; GE: idx 0, x0 good / x1 bad)
// Nothing is stored: any draw can be recomputed from (seed, env, step), which is what makes
// the rollout reproducible across GPU counts (env ids are global).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cm {

enum : uint32_t { SITE_SPAWN = 1, SITE_PREY = 2, SITE_IID_STEP = 3, SITE_IID_RESET = 4, SITE_GE_STEP = 5,
                  SITE_GE_RESET = 6, SITE_ACTION = 7, SITE_GE_INIT = 8, SITE_FAULT = 9 };

struct u32x4 { uint32_t x, y, z, w; };

__host__ __device__ inline u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                               uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return { c0, c1, c2, c3 };
}

__host__ __device__ inline float unit_f32(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
__host__ __device__ inline uint32_t pick(const u32x4 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }

// prey move from a raw word: thresholds floor(cdf * 2^32), cdf = .175 .35 .525 .7 (predator_prey.py:55,401)
__host__ __device__ inline int prey_move_from_u32(uint32_t v) {
    return (v >= 751619276u) + (v >= 1503238553u) + (v >= 2254857830u) + (v >= 3006477107u);
}

}  // namespace cm
