// act_determinism.hip -- development probe: are the activation functions of urnn_common.h bit-reproducible from launch to launch
// under load?  Each variant maps a fixed input array through one function 300 times; every output is compared with the first
// launch's.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DURNN_ACT=0|1] tools/ubench/act_determinism.hip -o act_det
#define URNN_ALLOW_PACKED_F32 1   // a probe, not the library: no MFMA next to packed fp32 here
#include "../../u-rnn_amd/csrc/urnn_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

template <int WHICH>
__global__ __launch_bounds__(256) void k(const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ o, int n)
{
    const int base = blockIdx.x * (256 * 4 * 8);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int p = base + (it * 256 + threadIdx.x) * 4;
        if (p >= n) break;
        const f32x4 a = *reinterpret_cast<const f32x4 *>(x + p), b = *reinterpret_cast<const f32x4 *>(y + p);
        f32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (WHICH == 0) r[q] = exp_neg(-fabsf(a[q]));
            else if (WHICH == 1) r[q] = sigmoidf_fast(a[q]);
            else if (WHICH == 2) r[q] = tanhf_fast(a[q]);
            else if (WHICH == 3) { const float z = sigmoidf_fast(a[q] * 1.3f + 0.1f), t = tanhf_fast(b[q] * 0.7f - 0.2f); r[q] = (1.f - z) * b[q] + z * t; }
            else if (WHICH == 4) r[q] = ldexpf(a[q], (int)rintf(b[q] * 3.f));
            else if (WHICH == 5) r[q] = __builtin_amdgcn_exp2f(a[q] * 0.25f) * b[q];
            else r[q] = siluf_fast(a[q]);
        }
        *reinterpret_cast<f32x4 *>(o + p) = r;
    }
}

template <int WHICH>
static int run(const char *name, const float *dx, const float *dy, float *dout, float *dref, int n, int reps)
{
    const int grid = (n + 256 * 4 * 8 - 1) / (256 * 4 * 8);
    hipLaunchKernelGGL(k<WHICH>, dim3(grid), dim3(256), 0, 0, dx, dy, dref, n);
    hipDeviceSynchronize();
    std::vector<float> ref(n), got(n);
    hipMemcpy(ref.data(), dref, n * 4, hipMemcpyDeviceToHost);
    long bad_launches = 0, bad_values = 0;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k<WHICH>, dim3(grid), dim3(256), 0, 0, dx, dy, dout, n);
        hipDeviceSynchronize();
        hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long nb = 0;
        int first = -1;
        for (int i = 0; i < n; ++i)
            if (memcmp(&got[i], &ref[i], 4)) { if (first < 0) first = i; ++nb; }
        if (nb) {
            ++bad_launches;
            bad_values += nb;
            if (bad_launches <= 3) printf("   %s launch %d: %ld values differ, first at %d (lane %d): %.9g vs %.9g\n", name, r, nb, first, first / 4 % 64, got[first], ref[first]);
        }
    }
    printf("%-28s %ld of %d launches differ from the first (%ld values)\n", name, bad_launches, reps, bad_values);
    return bad_launches != 0;
}

int main()
{
    const int n = 1 << 24, reps = 300;
    float *dx, *dy, *dout, *dref;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, n * 4); hipMalloc(&dref, n * 4);
    std::vector<float> hx(n), hy(n);
    srand(7);
    for (int i = 0; i < n; ++i) { hx[i] = (rand() / (float)RAND_MAX - 0.5f) * 12.f; hy[i] = (rand() / (float)RAND_MAX - 0.5f) * 2.f; }
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, hy.data(), n * 4, hipMemcpyHostToDevice);
    printf("URNN_ACT = %d\n", URNN_ACT);
    int bad = 0;
    bad |= run<0>("exp_neg", dx, dy, dout, dref, n, reps);
    bad |= run<1>("sigmoidf_fast", dx, dy, dout, dref, n, reps);
    bad |= run<2>("tanhf_fast", dx, dy, dout, dref, n, reps);
    bad |= run<3>("blend expression", dx, dy, dout, dref, n, reps);
    bad |= run<4>("ldexp(rint)", dx, dy, dout, dref, n, reps);
    bad |= run<5>("exp2 * y", dx, dy, dout, dref, n, reps);
    bad |= run<6>("siluf_fast", dx, dy, dout, dref, n, reps);
    return bad;
}
