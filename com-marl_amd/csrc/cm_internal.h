// cm_internal.h - shared between the translation units of libcommarl_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/commarl.h"

namespace cm {

int set_error(int code, const std::string &msg);   // stores thread-local text, returns code
int hip_fail(hipError_t e, const char *what);      // -> CM_ERR_HIP

#define CM_HIP(call)                                          \
    do {                                                      \
        hipError_t _e = (call);                               \
        if (_e != hipSuccess) return ::cm::hip_fail(_e, #call); \
    } while (0)

// Function attributes (hipFuncAttributeMaxDynamicSharedMemorySize) and CU counts belong to a DEVICE, not to the process:
// a launcher's `static unsigned long long done` carries one bit per device ordinal, so a process that moves on to a second
// GPU sets the attribute there too.  (Host launch paths are single-threaded per handle, include/commarl.h.)
inline bool dev_first(unsigned long long &done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done & bit) return false;
    done |= bit;
    return true;
}
inline int cu_count() {
    static int cached[64] = { 0 };
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    int &n = cached[dev & 63];
    if (n <= 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

// threadIdx.x through an opaque register: inside the step loop of the persistent rollout kernel the compiler otherwise
// hoists every lane-derived address computation of both bodies out of the loop (several hundred live VGPRs: one
// workgroup per CU, or 256 spilled registers when held to two).  Costs one move in the single-step kernels.
__device__ __forceinline__ int thread_x() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// Device-side view of one batched env set (passed to kernels by value).
struct EnvDev {
    int scen, B, N, M, S, R, W, d, load, max_steps, mpl, L, rc2, channel, add_clock, n_empty, rng_mode, env_id_offset;
    int adj_const, ch_const;
    int ge_flags;             // cm_env_cfg.ge_flags (bit 0: one GE transition per env step; bits 1-2: initial state)
    int lpe, lds_env;         // lanes per env (16/32/64) and LDS bytes per env
    int stop;                 // diagnostic (COMMARL_ENV_STOP): return after phase `stop`; 0 = run everything
    int no_small;             // COMMARL_ENV_SMALL=0: tile walk also for small PP teams (A/B of pp_small_step)
    float rcp_d, rcp_W, rcp_N, rcp_WW, rcp_NN;   // float reciprocals for the exact fast division in the emit loops
    float ploss, pgb, pbg;
    double cap_rew, step_cost, move_cost, penalty, lazy, revisit, final_reward;
    uint32_t key0, key1;
    // SoA state in HBM
    int2 *agent_pos;          // [B,N]  (row, col)
    int2 *prey_pos;           // [B,M]
    uint8_t *alive;           // [B,M]
    uint32_t *visited;        // [B,S]  row bitmasks (CO)
    int32_t *step_count;      // [B]
    int32_t *total_capture;   // [B]
    int32_t *success;         // [B]
    uint8_t *ge_state;        // [B,N,N]
    uint32_t *rng_step;       // [B]
    uint8_t *agent_cond;      // [B,N] PP agent_condition (predator_prey.py:74,152,258): 0 = the agent cannot move
    int32_t *status;          // [1] first kernel-side error
    unsigned int *tail_ticket;  // [1] zero at rest: waves that finished a chunk whose tail is folded into the kernel (cm_rollout_w.hip)
    // read-only tables
    const uint8_t *base_grid; // [S*S] CO walls (0 empty / 3 wall)
    const float *lut_row;     // [S]   obs row coordinate
    const float *lut_col;     // [S]
    const float *lut_step;    // [max_steps+1] clock
    const double *rew_lut;    // count-indexed f64 reward terms (no f64 division on the device)
};

}  // namespace cm

struct cm_env {
    cm_env_cfg cfg;
    cm::EnvDev dev;
    size_t lds_bytes;
    void *arena;              // one hipMalloc for all state
    size_t arena_bytes;
};

namespace cm {
// Per-step element strides of the time-major trajectory buffers, for the kernels that run several rollout steps per launch
// (cm_rollout_chunk): step t reads / writes base + t * stride.
struct ChunkArgs {
    int n_steps;
    int stagger;              // late start of the second half of the grid, in units of s_sleep 32 (~2048 clocks)
    long long obs, actions, probs, attn, reward, reward_f64, done, details, dist_adj, channels, prey_alive, success, path_len;
    // cm_rollout_chunk_tail: the tail the caller wants behind the chunk - the last step's observation into `tail_obs` (slot 0) and
    // *tail_base += n_steps.  A kernel that does it itself sets *tail_folded (host) to 1; otherwise the entry point launches cm_chunk_tail.
    float *tail_obs = nullptr;
    uint32_t *tail_base = nullptr;
    int *tail_folded = nullptr;
};

int check_tape(const cm_env *h, const cm_rng_tape *tape, bool is_reset);   // cm_env.hip: tape pointers the config needs
namespace mf {
// cm_policy_mfma.hip: one layer's weights [K,OUT] -> MFMA B fragments [out_pad/16][kpad/16][64 lanes][4], zero padded
int pack_one(const float *Wt, int K, int OUT, int kpad, int out_pad, float *dst, void *stream);
}
}
