// f16 denormals on gfx950: (1) v_fma_mixlo_f16 results, (2) v_mfma_f32_16x16x32_f16 inputs (A and B side), (3) v_cvt_f16_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float aval) {
    const float pre = 1.0f / 2048.0f;
    uint32_t hiu;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(hiu) : "v"(aval), "s"(pre));
    _Float16 hmix; __builtin_memcpy(&hmix, &hiu, 2);
    const _Float16 hcvt = (_Float16)(aval * pre);
    h8 a, b, one;
    for (int s = 0; s < 8; ++s) { a[s] = hcvt; b[s] = hcvt; one[s] = (_Float16)1.0f; }
    f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 da = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, one, z, 0, 0, 0);
    const f32x4 db = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, b, z, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = (float)hmix; out[1] = (float)hcvt; out[2] = da[0]; out[3] = db[0]; }
}
int main() {
    float* d; (void)hipMalloc(&d, 16);
    for (float v : {1.0f, 0.1f, 0.01f, 0.001f, 1e-4f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, v);
        float h[4]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("x = %.4e  x/2048 = %.6e | fma_mixlo -> %.6e  cvt -> %.6e | mfma16x16x32 sum K=32: A-side %.6e  B-side %.6e  expected %.6e\n", v, v / 2048.0, h[0], h[1], h[2], h[3], 32.0 * h[1]);
    }
    return 0;
}
