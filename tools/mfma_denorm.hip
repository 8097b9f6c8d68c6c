// Are f16 denormal inputs honoured by v_mfma_f32_32x32x16_f16 on gfx950?  (A = 2^-20, a subnormal f16; B = 1)  expect 16 * 2^-20 = 1.526e-05
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float aval) {
    h8 a, b;
    for (int s = 0; s < 8; ++s) { a[s] = (_Float16)aval; b[s] = (_Float16)1.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 8);
    for (float v : {9.5367431640625e-07f, 5.9604644775390625e-08f, 3.0517578125e-05f, 6.103515625e-05f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
        float h[2]; (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %.6e (as f16 -> %.6e): mfma sum over K=16 = %.6e, expected %.6e\n", v, h[1], h[0], 16.0 * h[1]);
    }
    return 0;
}
