// cm_api.hip - error plumbing + version of the C ABI (include/commarl.h)
#include "cm_internal.h"

namespace cm {

static thread_local std::string g_last_error;

int set_error(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return CM_ERR_HIP;
}

}  // namespace cm

namespace cm {

struct TailCopy { const uint32_t *src; uint32_t *dst; size_t words; };

// slot n -> slot 0 of up to three buffers + the sampler's counter bump; grid-stride dword copies (16-byte where aligned)
__global__ __launch_bounds__(256) void chunk_tail_kernel(uint32_t *step_base, uint32_t n, TailCopy c0, TailCopy c1, TailCopy c2) {
    if (step_base && blockIdx.x == 0 && threadIdx.x == 0) *step_base += n;
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, nth = (size_t)gridDim.x * blockDim.x;
    const TailCopy cs[3] = { c0, c1, c2 };
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const TailCopy c = cs[k];
        if (!c.words) continue;
        if ((((uintptr_t)c.src | (uintptr_t)c.dst) & 15) == 0) {
            const uint4 *s4 = reinterpret_cast<const uint4 *>(c.src);
            uint4 *d4 = reinterpret_cast<uint4 *>(c.dst);
            const size_t q = c.words / 4;
            for (size_t i = tid; i < q; i += nth) d4[i] = s4[i];
            for (size_t i = 4 * q + tid; i < c.words; i += nth) c.dst[i] = c.src[i];
        } else {
            for (size_t i = tid; i < c.words; i += nth) c.dst[i] = c.src[i];
        }
    }
}

}  // namespace cm

extern "C" int cm_chunk_tail(uint32_t *policy_step_base, uint32_t n_steps, const void *src0, void *dst0, size_t bytes0,
                             const void *src1, void *dst1, size_t bytes1, const void *src2, void *dst2, size_t bytes2,
                             void *stream) {
    using namespace cm;
    const void *srcs[3] = { src0, src1, src2 };
    void *dsts[3] = { dst0, dst1, dst2 };
    size_t bytes[3] = { bytes0, bytes1, bytes2 };
    TailCopy c[3];
    size_t total = 0;
    for (int k = 0; k < 3; ++k) {
        if (!srcs[k] || !dsts[k]) bytes[k] = 0;
        if (bytes[k] & 3) return set_error(CM_ERR_ARG, "cm_chunk_tail: byte counts must be multiples of 4");
        if (bytes[k] && ((((uintptr_t)srcs[k]) | ((uintptr_t)dsts[k])) & 3)) return set_error(CM_ERR_ARG, "cm_chunk_tail: pointers must be 4-byte aligned");
        c[k] = TailCopy{ (const uint32_t *)srcs[k], (uint32_t *)dsts[k], bytes[k] / 4 };
        total += bytes[k] / 16 + 1;
    }
    if (!policy_step_base && total == 3) return CM_OK;
    size_t blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(chunk_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, policy_step_base, n_steps, c[0], c[1], c[2]);
    CM_HIP(hipGetLastError());
    return CM_OK;
}

extern "C" int cm_abi_version(void) { return CM_ABI_VERSION; }
extern "C" const char *cm_last_error(void) { return cm::g_last_error.c_str(); }
