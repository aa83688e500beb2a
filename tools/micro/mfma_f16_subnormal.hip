// Does v_mfma_f32_16x16x32_f16 honour f16 SUBNORMAL inputs, or flush them to zero?  (decides whether the lo plane of the (hi, lo)
// operand pairs could be stored unscaled.)  A = 1.0 in one k slot, B = a subnormal f16 in that slot: D should be the subnormal's value.
//   hipcc -O2 --offload-arch=gfx950 mfma_f16_subnormal.hip -o mfma_f16_subnormal
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v4h __attribute__((ext_vector_type(4)));
__global__ void k(float *out, float val, float cvt_in) {
    const int lane = threadIdx.x;
    v8h a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.0f; b[e] = (_Float16)0.0f; }
    if ((lane >> 4) == 0) { a[0] = (_Float16)1.0f; b[0] = (_Float16)val; }        // k = 0 of every row / column
    v4f acc = { 0.f, 0.f, 0.f, 0.f };
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    v4h a4, b4;
    for (int e = 0; e < 4; ++e) { a4[e] = (_Float16)0.0f; b4[e] = (_Float16)0.0f; }
    if ((lane >> 4) == 0) { a4[0] = (_Float16)1.0f; b4[0] = (_Float16)val; }
    v4f acc2 = { 0.f, 0.f, 0.f, 0.f };
    acc2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc2, 0, 0, 0);
    if (lane == 0) { out[0] = acc[0]; out[1] = acc2[0]; out[2] = (float)(_Float16)cvt_in; out[3] = (float)b[0]; }
}
int main() {
    float *d; hipMalloc(&d, 16);
    const float vals[] = { 6.103515625e-05f /* 2^-14 min normal */, 3.0517578125e-05f /* 2^-15 subnormal */, 5.9604644775390625e-08f /* 2^-24 smallest */, 1.2345e-6f };
    for (float v : vals) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v, v);
        float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("input %.6e : mfma16x16x32 -> %.6e   mfma16x16x16 -> %.6e   f16 round trip (cvt) -> %.6e  operand as stored -> %.6e\n", v, h[0], h[1], h[2], h[3]);
    }
    return 0;
}
