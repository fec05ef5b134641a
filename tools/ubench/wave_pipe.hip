// wave_pipe.hip -- development experiment for the next step of the GEMM kernels ("what is left" item 0):
// can ONE wave per SIMD keep the matrix pipe busy while it also does the bf16 x 6 split's VALU work, and what does an
// interleaved 16-byte store cost a wave whose loads retire through the same in-order vmcnt counter?
//
// part 1: cycles per 16-k group of a gate tile (NB = 2, PB = 4: 48 x v_mfma_f32_32x32x16_bf16 + the split of 16 fragment pairs,
//         176 VALU) for   mfma only | valu only | burst (all VALU, then all MFMAs: today's order) | interleaved (the VALU of
//         the NEXT group spread between this group's MFMAs) -- with one wave per SIMD and with two.
// part 2: a wave streams 1-KB rows with 8 loads in flight (the ring); every step it also stores 1 KB (or not): time per step.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split_pair(float xe, float xo, unsigned &ph, unsigned &pm, unsigned &pl)
{
    const unsigned ue = __float_as_uint(xe), uo = __float_as_uint(xo);
    ph = __builtin_amdgcn_perm(uo, ue, 0x07060302u);
    const float re = xe - __uint_as_float(ue & 0xffff0000u), ro = xo - __uint_as_float(uo & 0xffff0000u);
    const unsigned ve = __float_as_uint(re), vo = __float_as_uint(ro);
    pm = __builtin_amdgcn_perm(vo, ve, 0x07060302u);
    const float se = re - __uint_as_float(ve & 0xffff0000u), so = ro - __uint_as_float(vo & 0xffff0000u);
    pl = __builtin_amdgcn_perm(__float_as_uint(so), __float_as_uint(se), 0x07060302u);
}

// mode 0 mfma only, 1 valu only, 2 burst, 3 interleaved
template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void group_kernel(float *out, long long *cyc, int groups, float seed)
{
    const int lane = threadIdx.x & 63;
    f32x16 acc[2][4];
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b)
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float raw[4][8];                                     // [pb][k-pair of the group]: stand-ins for the ring fragments
    for (int b = 0; b < 4; ++b)
        for (int q = 0; q < 8; ++q) raw[b][q] = seed * (float)(lane + 1) * (1.f + 0.01f * (float)(q + 8 * b));
    unsigned pc[3][4][4], pn[3][4][4];                   // pieces of the current / next group: [piece][pb][dword]
    for (int p = 0; p < 3; ++p)
        for (int b = 0; b < 4; ++b)
            for (int d = 0; d < 4; ++d) pc[p][b][d] = pn[p][b][d] = 0x3f803f80u;
    u32x4 w[2][3];
    for (int a = 0; a < 2; ++a)
        for (int p = 0; p < 3; ++p) w[a][p] = u32x4{0x3f803f80u, 0x3c003c00u, 0x38003800u, 0x3f803f80u};
    auto mfma = [&](int a, int b, int qa, int qb) {
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w[a][qa]),
                                                           __builtin_bit_cast(bf16x8, u32x4{pc[qb][b][0], pc[qb][b][1], pc[qb][b][2], pc[qb][b][3]}), acc[a][b], 0, 0, 0);
    };
    auto split_one = [&](int b, int d) {                 // one fragment pair of the next group: 11 VALU
        asm volatile("" : "+v"(raw[b][2 * d]), "+v"(raw[b][2 * d + 1]));      // opaque: a fresh fragment every group
        split_pair(raw[b][2 * d], raw[b][2 * d + 1], pn[0][b][d], pn[1][b][d], pn[2][b][d]);
    };
    const int qa[6] = {1, 2, 0, 1, 0, 0}, qb[6] = {1, 0, 2, 0, 1, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int g = 0; g < groups; ++g) {
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int d = 0; d < 4; ++d) split_one(b, d);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int p = 0; p < 6; ++p)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) mfma(a, b, qa[p], qb[p]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) {
            // 48 MFMAs, 16 splits: one split (11 VALU) after every third MFMA, pinned
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                const int p = i / 8, a = (i / 4) & 1, b = i & 3;
                mfma(a, b, qa[p], qb[p]);
                if (i % 3 == 2) {
                    const int s = i / 3;
                    split_one(s >> 2, s & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (MODE != 0) {
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int d = 0; d < 4; ++d) pc[p][b][d] = pn[p][b][d];
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 4; ++b)
            for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    for (int p = 0; p < 3; ++p)
        for (int b = 0; b < 4; ++b)
            for (int d = 0; d < 4; ++d) s += (float)(pc[p][b][d] & 0xff);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

// part 2: per wave a ring of 8 x 16-byte loads in flight; STORE: one 16-byte store per step as well
template <int STORE>
__global__ __launch_bounds__(512) void stream_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, long long *cyc, int steps, size_t stride4)
{
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const f32x4 *p = src + (size_t)wave * steps * 64 + lane;          // each wave its own contiguous stream, 1 KB per step
    f32x4 *q = dst + (size_t)wave * steps * 64 + lane;
    f32x4 ring[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ring[i] = __builtin_nontemporal_load(p + (size_t)i * 64);
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_readcyclecounter();
    for (int s0 = 0; s0 + 16 <= steps; s0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = ring[i];                                   // waits for the oldest load only (counted vmcnt)
            ring[i] = __builtin_nontemporal_load(p + (size_t)(s0 + 8 + i) * 64);
            sum += v;
            if (STORE) __builtin_nontemporal_store(v, q + (size_t)(s0 + i) * 64);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += ring[i];
    if (sum.x == 12345.f) q[0] = sum;
    if (lane == 0) cyc[wave] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
static double run_group(int blocks, int waves_per_block, int groups, float *out, long long *cyc, std::vector<long long> &h)
{
    if (waves_per_block == 4) hipLaunchKernelGGL((group_kernel<MODE, 256>), dim3(blocks), dim3(256), 0, 0, out, cyc, groups, 1e-3f);
    else hipLaunchKernelGGL((group_kernel<MODE, 512>), dim3(blocks), dim3(512), 0, 0, out, cyc, groups, 1e-3f);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks * waves_per_block, hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks * waves_per_block; ++i) s += (double)h[i];
    return s / (blocks * waves_per_block) / groups;
}

int main()
{
    const int groups = 2000;
    float *out; long long *cyc;
    CK(hipMalloc(&out, sizeof(float) * 256 * 512 * 2));
    CK(hipMalloc(&cyc, sizeof(long long) * 8192));
    std::vector<long long> h(8192);
    printf("part 1: cycles (s_memtime-class counter) per 16-k group: 48 MFMA 32x32x16 bf16 + 16 fragment-pair splits (176 VALU)\n");
    for (int wps = 1; wps <= 2; ++wps) {
        const int wpb = 4 * wps;    // 256 blocks: one per CU, wps waves per SIMD
        printf("  %d wave(s) per SIMD: mfma only %.0f | valu only %.0f | burst %.0f | interleaved %.0f\n", wps,
               run_group<0>(256, wpb, groups, out, cyc, h), run_group<1>(256, wpb, groups, out, cyc, h), run_group<2>(256, wpb, groups, out, cyc, h),
               run_group<3>(256, wpb, groups, out, cyc, h));
    }
    // part 2
    const int steps = 512, waves = 256 * 8;
    const size_t n4 = (size_t)waves * steps * 64;
    f32x4 *src, *dst;
    CK(hipMalloc(&src, n4 * 16));
    CK(hipMalloc(&dst, n4 * 16));
    CK(hipMemset(src, 0, n4 * 16));
    printf("part 2: 2048 waves each streaming %d x 1 KB with 8 loads in flight (%.0f MB read)\n", steps, n4 * 16 / 1e6);
    for (int rep = 0; rep < 2; ++rep) {
        for (int st = 0; st < 2; ++st) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            if (st) hipLaunchKernelGGL(stream_kernel<1>, dim3(256), dim3(512), 0, 0, src, dst, cyc, steps, 0);
            else hipLaunchKernelGGL(stream_kernel<0>, dim3(256), dim3(512), 0, 0, src, dst, cyc, steps, 0);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), cyc, sizeof(long long) * waves, hipMemcpyDeviceToHost);
            double s = 0;
            for (int i = 0; i < waves; ++i) s += (double)h[i];
            printf("  %s: %.1f us, %.2f TB/s moved, %.0f counter ticks per step per wave\n", st ? "load + store per step" : "loads only           ", ms * 1e3,
                   (st ? 2.0 : 1.0) * n4 * 16 / (ms * 1e-3) / 1e12, s / waves / (steps - 8));
        }
    }
    return 0;
}
