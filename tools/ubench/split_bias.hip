// Is the f16 hi + lo split of an activation (urnn_common.h split2_pair) unbiased?  Mean signed relative error of (hi + lo) / 32 - x over
// random x, per magnitude range; and the rounding mode of the conversion instruction it compiles to.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I u-rnn_amd/csrc tools/ubench/split_bias.hip -o tools/ubench/split_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define URNN_ALLOW_PACKED_F32 1   // a probe, not the library: no MFMA next to packed fp32 here
#include "urnn_common.h"
__global__ void k(const float *x, float *y, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 < n) {
        unsigned ph, pl;
        split2_pair(x[2 * i], x[2 * i + 1], URNN_F16_ASCALE, ph, pl);
        const f16x2 hi = __builtin_bit_cast(f16x2, ph), lo = __builtin_bit_cast(f16x2, pl);
        y[2 * i] = ((float)hi.x + (float)lo.x);          // exact in fp32 (22 bits)
        y[2 * i + 1] = ((float)hi.y + (float)lo.y);
    }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), y(n);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        x[i] = (float)(exp(u * 14.0 - 12.0) * ((st & 1) ? 1.0 : -1.0));      // |x| log-uniform in [6e-6, 7.4]
    }
    x[0] = 1.0f + 0.625f / 1024.0f;   // rounding-mode probe: RNE -> 1 + 2^-10 as hi, RTZ -> 1.0
    x[1] = 1.0f + 0.375f / 1024.0f;
    float *dx, *dy;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dy, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, dx, dy, n);
    (void)hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost);
    const double edges[] = {0, 1e-4, 1e-3, 4e-3, 1e-2, 1e-1, 1.0, 10.0};
    for (int r = 0; r < 7; ++r) {
        double sum = 0, sq = 0, sabs = 0; long cnt = 0;
        for (int i = 2; i < n; ++i) {
            const double v = x[i];
            if (fabs(v) < edges[r] || fabs(v) >= edges[r + 1]) continue;
            const double e = ((double)y[i] / 32.0 - v) / fabs(v) * (v > 0 ? 1 : -1);   // > 0: magnitude overestimated
            sum += e; sq += e * e; sabs += fabs(e); ++cnt;
        }
        printf("|x| in [%7.0e, %7.0e): n %7ld  mean signed rel err (of |x|) %+9.2e  rms %8.2e\n", edges[r], edges[r + 1], cnt, sum / cnt, sqrt(sq / cnt));
    }
    printf("hi+lo of 1 + 0.625 * 2^-10, x32: %.9g (input %.9g)   of 1 + 0.375 * 2^-10: %.9g (input %.9g)\n", y[0], x[0] * 32.0, y[1], x[1] * 32.0);
    return 0;
}
