// Accuracy AND BIAS of the product's activations (u-rnn_amd/csrc/urnn_common.h: sigmoidf_fast, tanhf_fast) against double, in ulps of
// the result: mean signed error (a bias accumulates over a recurrent rollout, a rounding error does not), rms, max; per range.
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I u-rnn_amd/csrc tools/ubench/act_accuracy.hip -o tools/ubench/act_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define URNN_ALLOW_PACKED_F32 1   // a probe, not the library: no MFMA next to packed fp32 here
#include "urnn_common.h"
__global__ void k(const float *x, float *s, float *t, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = sigmoidf_fast(x[i]); t[i] = tanhf_fast(x[i]); }
}
static double ulp(double v) { int e; frexp(fabs(v), &e); return ldexp(1.0, e - 24); }
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), s(n), t(n);
    unsigned long long st = 88172645463325252ULL;
    for (int i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; x[i] = (float)((double)(st >> 11) / 9007199254740992.0 * 24.0 - 12.0); }
    float *dx, *ds, *dt;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&ds, n * 4); (void)hipMalloc(&dt, n * 4);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, ds, dt, n);
    (void)hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(t.data(), dt, n * 4, hipMemcpyDeviceToHost);
    const double edges[] = {0.0, 0.1, 0.4, 1.0, 2.0, 4.0, 8.0, 12.0};
    for (int f = 0; f < 2; ++f)
        for (int sign = -1; sign <= 1; sign += 2)
            for (int r = 0; r < 7; ++r) {
                double sum = 0, sq = 0, mx = 0; long cnt = 0;
                for (int i = 0; i < n; ++i) {
                    const double v = x[i];
                    if ((v < 0) != (sign < 0) || fabs(v) < edges[r] || fabs(v) >= edges[r + 1]) continue;
                    const double ref = f ? tanh(v) : 1.0 / (1.0 + exp(-v));
                    const double got = f ? t[i] : s[i];
                    const double e = (got - ref) / ulp(ref);
                    sum += e; sq += e * e; mx = fmax(mx, fabs(e)); ++cnt;
                }
                printf("%s  v in %c[%4.1f, %4.1f): n %7ld  bias %+7.3f ulp  rms %6.3f  max %6.2f\n", f ? "tanh   " : "sigmoid", sign < 0 ? '-' : '+', edges[r], edges[r + 1], cnt,
                       sum / cnt, sqrt(sq / cnt), mx);
            }
    return 0;
}
