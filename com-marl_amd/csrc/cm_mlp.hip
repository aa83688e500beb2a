// cm_mlp.hip - fused row-wise MLP forward on the gfx950 matrix cores for the two non-communicating
// policies of the reference and the plain Gaussian baseline:
//   * DecCategoricalMLPPolicy.get_actions  (com_marl/torch/policies/dec_categorical_mlp_policy.py:106-176):
//       per agent row  obs[d] -> 128 tanh -> 64 tanh | 32 tanh -> 5 logits -> softmax * avail, renorm, sample
//   * CentralizedCategoricalMLPPolicy.get_actions (centralized_categorical_mlp_policy.py:61-118):
//       per env row  obs[N*d] -> 128 tanh -> 64 tanh -> 32 tanh -> N*5 logits -> per-agent softmax ..., sample
//   * GaussianMLPBaseline.forward (com_marl/torch/baselines/gaussian_mlp_baseline.py:100-115):
//       per env row  obs[N*d] -> 64 tanh -> 64 tanh -> 64 tanh -> 1
// Layer sizes are run-time values (cm_mlp_weights); every layer is  v_mfma_f32_16x16x4_f32  (f32 in, f32
// accumulate - the 1e-5 parity bar), 32 rows (two 16-row tiles) per 256-thread workgroup, activations ping-pong
// between two LDS tiles, the first layer streams its (possibly thousands of columns wide) input through LDS in
// 128-column chunks with the accumulators held in registers.  HBM traffic = input rows in, actions / probs /
// values out; weights are read from L2.
#include "cm_internal.h"
#include "cm_rng.h"

namespace cm {
namespace mlp {

constexpr int TPB = 256, ROWS = 32, CHUNK = 128, MAXL = CM_MLP_MAX_LAYERS, MAX_ACT = 8;
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_tanh(float x) {      // same form as cm_policy_mfma.hip (abs err <= 2e-7)
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

struct Args {
    int rows, in_dim, n_layers, tanh_mask, sw;
    int out_dim[MAXL];
    const float *wt[MAXL], *b[MAXL];
    const float *pk[MAXL];          // per-layer B fragments (cm_mlp_pack) or NULL: [ct][kq][lane][4], kq over ceil(K/16)
    const float *x, *avail;
    int groups, n_act, agents_per_env, env_id_offset, greedy;
    uint32_t key0, key1, policy_step;
    const uint32_t *step_base;
    int32_t *actions;
    float *probs, *values;
};

extern __shared__ float smem[];

// k-slot mapping shared by A and B: lane group g = lane>>4 supplies k = 16*kq + 4*g + j at MFMA (kq, j), so a
// lane's four A words per kq are contiguous (one ds_read_b128).
__device__ __forceinline__ void store_tile(float *out, int sw, int rt, int ct, int lane, const v4f &acc, float bias,
                                           bool th) {
    const int c = lane & 15, g = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float v = acc[r] + bias;
        out[(size_t)(rt * 16 + 4 * g + r) * sw + ct * 16 + c] = th ? fast_tanh(v) : v;
    }
}

__global__ __launch_bounds__(TPB) void mlp_kernel(Args a) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * ROWS;
    const int rows = min(ROWS, a.rows - row0);
    const int sw = a.sw;
    float *buf0 = smem, *buf1 = smem + (size_t)ROWS * sw;

    // ---- layer 0: input streamed from HBM through buf1 in CHUNK-column pieces --------------------------------
    {
        const int K = a.in_dim, OUT = a.out_dim[0];
        const int nct = (OUT + 15) >> 4;                     // <= 8 (host-checked: OUT <= 128)
        const int KQ0 = (K + 15) >> 4;
        const float *__restrict__ Wt = a.wt[0];
        v4f acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = (v4f){ 0.f, 0.f, 0.f, 0.f }; acc[t][1] = (v4f){ 0.f, 0.f, 0.f, 0.f }; }
        for (int c0 = 0; c0 < K; c0 += CHUNK) {
            __syncthreads();                                 // previous chunk fully consumed
            for (int i = tid; i < ROWS * CHUNK; i += TPB) {
                const int r = i >> 7, cc = i & (CHUNK - 1);
                const int k = c0 + cc;
                buf1[(size_t)r * sw + cc] = (r < rows && k < K) ? a.x[(size_t)(row0 + r) * K + k] : 0.0f;
            }
            __syncthreads();
            const int kc = min(CHUNK, K - c0);
            const int k16 = (kc + 15) >> 4;
            for (int kq = 0; kq < k16; ++kq) {
                const float4 a0 = *reinterpret_cast<const float4 *>(buf1 + (size_t)c * sw + 16 * kq + 4 * g);
                const float4 a1 = *reinterpret_cast<const float4 *>(buf1 + (size_t)(16 + c) * sw + 16 * kq + 4 * g);
                const float x0[4] = { a0.x, a0.y, a0.z, a0.w }, x1[4] = { a1.x, a1.y, a1.z, a1.w };
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int ct = wave + 4 * t;
                    if (ct >= nct) continue;                 // wave-uniform
                    const int col = ct * 16 + c;
                    float bw[4];
                    if (a.pk[0]) {                           // one unconditional 16-byte load per lane
                        const float4 v = reinterpret_cast<const float4 *>(a.pk[0])[((size_t)ct * KQ0 + (c0 >> 4) + kq) * 64 + lane];
                        bw[0] = v.x; bw[1] = v.y; bw[2] = v.z; bw[3] = v.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = c0 + 16 * kq + 4 * g + j;
                            bw[j] = (k < K && col < OUT) ? Wt[(size_t)k * OUT + col] : 0.0f;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], bw[j], acc[t][0], 0, 0, 0);
                        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], bw[j], acc[t][1], 0, 0, 0);
                    }
                }
            }
        }
        const bool th = a.tanh_mask & 1;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ct = wave + 4 * t;
            if (ct >= nct) continue;
            const int col = ct * 16 + c;
            const float bias = (a.b[0] && col < OUT) ? a.b[0][col] : 0.0f;
            store_tile(buf0, sw, 0, ct, lane, acc[t][0], bias, th);
            store_tile(buf0, sw, 1, ct, lane, acc[t][1], bias, th);
        }
    }
    __syncthreads();

    // ---- layers 1..: LDS -> LDS --------------------------------------------------------------------------------
    float *in = buf0, *out = buf1;
    for (int l = 1; l < a.n_layers; ++l) {
        const int K = a.out_dim[l - 1], OUT = a.out_dim[l];
        const int nct = (OUT + 15) >> 4, k16 = (K + 15) >> 4;
        const float *__restrict__ Wt = a.wt[l];
        const bool th = (a.tanh_mask >> l) & 1;
        for (int ct = wave; ct < nct; ct += 4) {
            const int col = ct * 16 + c;
            v4f acc0 = (v4f){ 0.f, 0.f, 0.f, 0.f }, acc1 = (v4f){ 0.f, 0.f, 0.f, 0.f };
            for (int kq = 0; kq < k16; ++kq) {
                const float4 a0 = *reinterpret_cast<const float4 *>(in + (size_t)c * sw + 16 * kq + 4 * g);
                const float4 a1 = *reinterpret_cast<const float4 *>(in + (size_t)(16 + c) * sw + 16 * kq + 4 * g);
                const float x0[4] = { a0.x, a0.y, a0.z, a0.w }, x1[4] = { a1.x, a1.y, a1.z, a1.w };
                float bw[4];
                if (a.pk[l]) {
                    const float4 v = reinterpret_cast<const float4 *>(a.pk[l])[((size_t)ct * k16 + kq) * 64 + lane];
                    bw[0] = v.x; bw[1] = v.y; bw[2] = v.z; bw[3] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = 16 * kq + 4 * g + j;
                        bw[j] = (k < K && col < OUT) ? Wt[(size_t)k * OUT + col] : 0.0f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[j], bw[j], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[j], bw[j], acc1, 0, 0, 0);
                }
            }
            const float bias = (a.b[l] && col < OUT) ? a.b[l][col] : 0.0f;
            store_tile(out, sw, 0, ct, lane, acc0, bias, th);
            store_tile(out, sw, 1, ct, lane, acc1, bias, th);
        }
        __syncthreads();
        float *t = in; in = out; out = t;
    }
    // `in` now holds the last layer's output [ROWS][>= out_dim[n_layers-1]]

    if (a.values) {                                          // GaussianMLPBaseline mean: one value per row
        for (int r = tid; r < rows; r += TPB) a.values[row0 + r] = in[(size_t)r * sw];
        return;
    }

    // ---- per-agent softmax * avail, renormalise, sample (same arithmetic order as cm_policy_mfma.hip) ------------
    const int G = a.groups, A = a.n_act;
    for (int it = tid; it < rows * G; it += TPB) {
        const int r = it / G, gi = it - r * G;
        const float *lg = in + (size_t)r * sw + gi * A;
        float p[MAX_ACT];
        float mx = -INFINITY, sum = 0.0f, msum = 0.0f;
#pragma unroll
        for (int k = 0; k < MAX_ACT; ++k) if (k < A) mx = fmaxf(mx, lg[k]);
#pragma unroll
        for (int k = 0; k < MAX_ACT; ++k) if (k < A) { p[k] = expf(lg[k] - mx); sum += p[k]; }
        const size_t flat = (size_t)(row0 + r) * G + gi;     // global agent-row index
#pragma unroll
        for (int k = 0; k < MAX_ACT; ++k) if (k < A) {
            const float av = a.avail ? a.avail[flat * A + k] : 1.0f;
            p[k] = (p[k] / sum) * av; msum += p[k];
        }
#pragma unroll
        for (int k = 0; k < MAX_ACT; ++k) if (k < A) p[k] = p[k] / msum;
        if (a.probs) {
#pragma unroll
            for (int k = 0; k < MAX_ACT; ++k) if (k < A) a.probs[flat * A + k] = p[k];
        }
        if (a.actions) {
            int act = 0;
            if (a.greedy) {
                float best = p[0];
#pragma unroll
                for (int k = 1; k < MAX_ACT; ++k) if (k < A && p[k] > best) { best = p[k]; act = k; }
            } else {
                const size_t e = flat / (size_t)a.agents_per_env;
                const uint32_t i = (uint32_t)(flat - e * a.agents_per_env);
                const u32x4 xr = philox4x32_10((uint32_t)(a.env_id_offset + (int)e),
                                               a.policy_step + (a.step_base ? *a.step_base : 0u), SITE_ACTION, i,
                                               a.key0, a.key1);
                const float u = unit_f32(xr.x);
                float acc = 0.0f;
                int sel = -1, last = 0;
#pragma unroll
                for (int k = 0; k < MAX_ACT; ++k) if (k < A) { if (p[k] > 0.0f) last = k; acc += p[k]; if (sel < 0 && u < acc) sel = k; }
                act = sel < 0 ? last : sel;
            }
            a.actions[flat] = act;
        }
    }
}

static size_t pack_floats(int K, int OUT) { return (size_t)((OUT + 15) >> 4) * ((K + 15) >> 4) * 256; }

static int launch(Args a, void *stream) {
    int maxw = CHUNK;
    for (int l = 0; l < a.n_layers; ++l) maxw = max(maxw, (a.out_dim[l] + 15) & ~15);
    a.sw = maxw + 4;                                         // +4 words: rows skewed across LDS banks, 16-byte aligned
    const size_t lds = 2ull * ROWS * a.sw * sizeof(float);
    if (lds > 160 * 1024) return set_error(CM_ERR_ARG, "mlp forward: layer too wide for the 160 KB LDS tile");
    static unsigned long long attr_set = 0;
    if (cm::dev_first(attr_set)) {
        CM_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&mlp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024));
    }
    if (a.rows == 0) return CM_OK;
    const int blocks = (a.rows + ROWS - 1) / ROWS;
    hipLaunchKernelGGL(mlp_kernel, dim3(blocks), dim3(TPB), lds, (hipStream_t)stream, a);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

static int fill(Args &a, const cm_mlp_weights *w, int32_t rows, const float *x) {
    if (!w || !x) return set_error(CM_ERR_ARG, "mlp forward: null weights / input");
    if (rows < 0) return set_error(CM_ERR_ARG, "mlp forward: negative row count");
    if (w->n_layers < 1 || w->n_layers > MAXL) return set_error(CM_ERR_ARG, "mlp forward: 1..6 linear layers supported");
    if (w->in_dim < 1) return set_error(CM_ERR_ARG, "mlp forward: in_dim < 1");
    if (w->out_dim[0] > 128) return set_error(CM_ERR_ARG, "mlp forward: first layer wider than 128 outputs");
    a.rows = rows; a.in_dim = w->in_dim; a.n_layers = w->n_layers; a.tanh_mask = w->tanh_mask; a.x = x;
    for (int l = 0; l < w->n_layers; ++l) {
        if (w->out_dim[l] < 1 || w->out_dim[l] > 1024) return set_error(CM_ERR_ARG, "mlp forward: layer width outside 1..1024");
        if (!w->wt[l]) return set_error(CM_ERR_ARG, "mlp forward: null layer weight");
        a.out_dim[l] = w->out_dim[l]; a.wt[l] = w->wt[l]; a.b[l] = w->b[l];
    }
    size_t off = 0;
    for (int l = 0; l < w->n_layers; ++l) {
        a.pk[l] = w->mfma_pack ? w->mfma_pack + off : nullptr;
        off += pack_floats(l == 0 ? w->in_dim : w->out_dim[l - 1], w->out_dim[l]);
    }
    return CM_OK;
}

}  // namespace mlp
}  // namespace cm

extern "C" {

int cm_mlp_policy_forward(const cm_mlp_weights *w, int32_t rows, int32_t groups, int32_t n_act, int32_t agents_per_env,
                          const float *x, const float *avail, uint64_t seed, int32_t env_id_offset,
                          uint32_t policy_step, const uint32_t *policy_step_base, int32_t greedy, int32_t *actions,
                          float *probs, void *stream) {
    cm::mlp::Args a{};
    if (int rc = cm::mlp::fill(a, w, rows, x)) return rc;
    if (groups < 1 || n_act < 1 || n_act > cm::mlp::MAX_ACT || agents_per_env < 1)
        return cm::set_error(CM_ERR_ARG, "mlp policy forward: groups >= 1, 1 <= n_act <= 8, agents_per_env >= 1");
    if (w->out_dim[w->n_layers - 1] != groups * n_act)
        return cm::set_error(CM_ERR_ARG, "mlp policy forward: last layer width != groups * n_act");
    if (!actions && !probs) return cm::set_error(CM_ERR_ARG, "mlp policy forward: no output requested");
    a.groups = groups; a.n_act = n_act; a.agents_per_env = agents_per_env; a.avail = avail;
    a.key0 = (uint32_t)seed; a.key1 = (uint32_t)(seed >> 32); a.policy_step = policy_step; a.step_base = policy_step_base;
    a.env_id_offset = env_id_offset; a.greedy = greedy; a.actions = actions; a.probs = probs;
    return cm::mlp::launch(a, stream);
}

int cm_mlp_value_forward(const cm_mlp_weights *w, int32_t rows, const float *x, float *values, void *stream) {
    cm::mlp::Args a{};
    if (int rc = cm::mlp::fill(a, w, rows, x)) return rc;
    if (!values) return cm::set_error(CM_ERR_ARG, "mlp value forward: null output");
    if (w->out_dim[w->n_layers - 1] != 1) return cm::set_error(CM_ERR_ARG, "mlp value forward: last layer width != 1");
    a.values = values;
    return cm::mlp::launch(a, stream);
}

size_t cm_mlp_pack_bytes(const cm_mlp_weights *w) {
    if (!w || w->n_layers < 1 || w->n_layers > cm::mlp::MAXL || w->in_dim < 1) return 0;
    size_t n = 0;
    for (int l = 0; l < w->n_layers; ++l) {
        if (w->out_dim[l] < 1) return 0;
        n += cm::mlp::pack_floats(l == 0 ? w->in_dim : w->out_dim[l - 1], w->out_dim[l]);
    }
    return n * sizeof(float);
}

int cm_mlp_pack(const cm_mlp_weights *w, float *pack, void *stream) {
    if (!w || !pack || !cm_mlp_pack_bytes(w)) return cm::set_error(CM_ERR_ARG, "cm_mlp_pack: null / malformed argument");
    size_t off = 0;
    for (int l = 0; l < w->n_layers; ++l) {
        const int K = l == 0 ? w->in_dim : w->out_dim[l - 1], OUT = w->out_dim[l];
        if (int rc = cm::mf::pack_one(w->wt[l], K, OUT, (K + 15) & ~15, (OUT + 15) & ~15, pack + off, stream)) return rc;
        off += cm::mlp::pack_floats(K, OUT);
    }
    return CM_OK;
}

}  // extern "C"
