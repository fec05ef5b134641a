// mall_roundtrip.hip -- two questions VERDICT r3 left open about the "HBM traffic" figures (MI355X_MICROARCH.md: 256 MiB Infinity
// Cache; FETCH_SIZE appears to count its hits; WRITE_SIZE uncalibrated):
//   1. does a 64-MB plane written by one kernel (the raw gate / candidate planes of a full-resolution cell) come back from the
//      Infinity Cache when the next kernel reads it -- and does a non-temporal store (what the GEMM epilogues use) change that?
//      -> read time of a buffer right after it was written (default stores / nontemporal stores) against the same read after 1 GiB of
//      other traffic has gone through the chip ("cold");
//   2. what do FETCH_SIZE and WRITE_SIZE report for a KNOWN byte count in these access shapes (16 B per lane, coalesced)?
//      -> run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` and divide (tools/pmc_summary.py prints the
//      per-kernel averages; the kernels carry their byte counts in their names' template arguments).
// build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/ubench/mall_roundtrip.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MB, int NT>
__global__ __launch_bounds__(256) void write_kernel(f32x4 *dst, size_t n4, float seed)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = {seed, seed + 1.f, seed + 2.f, (float)(i & 1023)};
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <int MB, int TAG>     // TAG: 0 read after default stores, 1 after nontemporal stores, 2 cold
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *src, size_t n4, float *sink)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) *sink = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MB>
static void run(f32x4 *buf, f32x4 *scratch, size_t scratch4, float *sink)
{
    const size_t n4 = (size_t)MB * 1024 * 1024 / 16;
    const int grid = 256 * 8;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms[3] = {0, 0, 0}, wms[2] = {0, 0};
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        for (int mode = 0; mode < 3; ++mode) {
            float t;
            // the producer
            CK(hipEventRecord(a));
            if (mode == 1) hipLaunchKernelGGL((write_kernel<MB, 1>), dim3(grid), dim3(256), 0, 0, buf, n4, (float)r);
            else hipLaunchKernelGGL((write_kernel<MB, 0>), dim3(grid), dim3(256), 0, 0, buf, n4, (float)r);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&t, a, b));
            if (mode < 2) wms[mode] += t;
            if (mode == 2) {            // cold: 1 GiB of other traffic between the producer and the consumer
                hipLaunchKernelGGL((write_kernel<1024, 0>), dim3(grid), dim3(256), 0, 0, scratch, scratch4, 1.f);
                hipLaunchKernelGGL((read_kernel<1024, 2>), dim3(grid), dim3(256), 0, 0, scratch, scratch4, sink);
                CK(hipDeviceSynchronize());
            }
            CK(hipEventRecord(a));
            if (mode == 0) hipLaunchKernelGGL((read_kernel<MB, 0>), dim3(grid), dim3(256), 0, 0, buf, n4, sink);
            else if (mode == 1) hipLaunchKernelGGL((read_kernel<MB, 1>), dim3(grid), dim3(256), 0, 0, buf, n4, sink);
            else hipLaunchKernelGGL((read_kernel<MB, 2>), dim3(grid), dim3(256), 0, 0, buf, n4, sink);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&t, a, b));
            ms[mode] += t;
        }
    }
    const double gb = (double)MB * 1024 * 1024 / 1e9;
    printf("%4d MB: write default %6.1f us (%5.2f TB/s)  nontemporal %6.1f us (%5.2f TB/s) | read after default stores %6.1f us (%5.2f TB/s)  after nontemporal "
           "stores %6.1f us (%5.2f TB/s)  cold %6.1f us (%5.2f TB/s)\n", MB, wms[0] / reps * 1e3, gb / (wms[0] / reps), wms[1] / reps * 1e3, gb / (wms[1] / reps),
           ms[0] / reps * 1e3, gb / (ms[0] / reps), ms[1] / reps * 1e3, gb / (ms[1] / reps), ms[2] / reps * 1e3, gb / (ms[2] / reps));
}

int main()
{
    f32x4 *buf, *scratch;
    float *sink;
    const size_t scratch4 = (size_t)1024 * 1024 * 1024 / 16;
    CK(hipMalloc(&buf, (size_t)1024 * 1024 * 1024));
    CK(hipMalloc(&scratch, scratch4 * 16));
    CK(hipMalloc(&sink, 4));
    printf("# producer -> consumer round trip of one buffer through the memory system (16 B per lane, coalesced; times by HIP events, mean of 5)\n");
    run<16>(buf, scratch, scratch4, sink);
    run<64>(buf, scratch, scratch4, sink);
    run<128>(buf, scratch, scratch4, sink);
    run<256>(buf, scratch, scratch4, sink);
    run<512>(buf, scratch, scratch4, sink);
    return 0;
}
