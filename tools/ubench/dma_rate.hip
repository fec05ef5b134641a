// LDS-DMA issue/throughput microbenchmark (development tool): how many cycles does a CU need per buffer_load ... lds
// instruction, as a function of waves per CU, bytes per lane and where the data lives?
//   hipcc --offload-arch=gfx950 -O3 -o dma_rate dma_rate.hip && ./dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
extern __shared__ __attribute__((aligned(16))) char smem[];

template <int BYTES, int MODE>   // MODE 0: LDS-DMA, 1: plain global load to registers (dwordx4 / dword)
__global__ void k(const float *src, size_t span_bytes, int iters, int rowstride_bytes, unsigned long long *out, float *sink)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    char *slot = smem + wave * 4096;
    rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, (int)0x7fffffff, 0x00020000);
    // each wave walks its own stream: like the GEMM ring, lanes 0-31 row k, lanes 32-63 row k+1, 16 B per lane
    unsigned base = (unsigned)(((size_t)(blockIdx.x * nw + wave) * 512) % (span_bytes / 4));
    unsigned vo = base + (lane >> 5) * rowstride_bytes + (lane & 31) * BYTES;
    unsigned step = 2u * rowstride_bytes;
    float acc = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        unsigned off = (vo + (unsigned)i * step) % (unsigned)(span_bytes - 4096);
        off &= ~15u;
        if constexpr (MODE == 0) {
            if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(slot + (i & 3) * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(slot + (i & 3) * 1024), 4, off, 0, 0, 0);
            if ((i & 3) == 3) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            if constexpr (BYTES == 16) {
                float4 v = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(src) + off);
                acc += v.x + v.w;
            } else {
                acc += *reinterpret_cast<const float *>(reinterpret_cast<const char *>(src) + off);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * nw + wave] = t1 - t0;
    if (acc == 12345.f) sink[0] = acc;
}

template <int BYTES, int MODE>
static void run(const char *name, const float *src, size_t span, int waves, int iters, unsigned long long *dout, float *sink)
{
    const int blocks = 256;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<BYTES, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<BYTES, MODE>), dim3(blocks), dim3(64 * waves), 90 * 1024, 0, src, span, iters, 1000000, dout, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks * waves);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= h.size();
    const double bytes = (double)blocks * waves * iters * 64 * BYTES;
    printf("%-28s waves/CU %2d  span %6.0f MB: %7.1f cyc per instr per wave, %6.1f cyc per instr per CU, %5.1f B/cyc/CU, %6.2f TB/s (kernel %.1f us)\n",
           name, waves, span / 1e6, mean / iters, mean / iters / waves, 64.0 * BYTES * waves * iters / mean, bytes / (ms * 1e-3) / 1e12, ms * 1e3);
}

int main()
{
    const size_t big = (size_t)1 << 30;
    float *src; hipMalloc(&src, big); hipMemset(src, 0, big);
    unsigned long long *dout; hipMalloc(&dout, 256 * 16 * 8);
    float *sink; hipMalloc(&sink, 4);
    for (size_t span : {(size_t)16 << 20, big}) {
        for (int waves : {4, 8, 16}) {
            run<16, 0>("lds-dma 16B/lane", src, span, waves, 2000, dout, sink);
            run<4, 0>("lds-dma 4B/lane", src, span, waves, 2000, dout, sink);
            run<16, 1>("global_load_dwordx4", src, span, waves, 2000, dout, sink);
        }
    }
    return 0;
}
