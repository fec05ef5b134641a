// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal INPUTS, and does the f32 -> f16 conversion of the split produce them?
// (The f16 x 3 split's lo piece of an activation |x| < 2^-8 is subnormal after the 2^5 scaling: urnn_common.h.)
// hipcc --offload-arch=gfx950 -O2 tools/ubench/f16_denorm.hip -o /tmp/f16_denorm && /tmp/f16_denorm
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, float xs)
{
    f16x8 a, b;
    const _Float16 sub = __builtin_bit_cast(_Float16, (unsigned short)0x0010);   // 2^-20: subnormal in f16
    for (int i = 0; i < 8; ++i) { a[i] = sub; b[i] = (_Float16)1.0f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    f32x16 acc2;
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc2, 0, 0, 0);
    // conversion: a float in f16's subnormal range
    const f32x2 v = {xs, xs * 3.0f};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    if (threadIdx.x == 0) {
        out[0] = acc[0];
        out[1] = acc2[0];
        out[2] = (float)h[0];
        out[3] = (float)h[1];
        // VALU f16 arithmetic on a subnormal
        _Float16 t = sub * (_Float16)2.0f;
        out[4] = (float)t;
    }
}
int main()
{
    float *d, h[5];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 3.0e-6f);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mfma(A = 16 x 2^-20 subnormal, B = 1): %.9g (exact 1.52587891e-05; 0 = inputs flushed)\n", h[0]);
    printf("mfma(A = 1, B = subnormal):            %.9g\n", h[1]);
    printf("cvt f32 3.0e-6 -> f16 -> f32:           %.9g (f16 subnormal grid 5.96e-8; 0 = flushed)\n", h[2]);
    printf("cvt f32 9.0e-6 -> f16 -> f32:           %.9g\n", h[3]);
    printf("f16 VALU subnormal * 2:                 %.9g (exact 1.90734863e-06)\n", h[4]);
    return 0;
}
