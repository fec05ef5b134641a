// operand_stream.hip -- development microbenchmark: at the product's residency (256 blocks x 8 waves, one
// private LDS-DMA ring per wave, counted vmcnt) stream K fp32 channel planes of P pixels through the CUs in several REQUEST SHAPES
// and print TB/s for each.  The question: is the 3.0-4.8 TB/s of the full-resolution cell GEMMs a property of how they ask for
// their operands (2 rows x 256 B per DMA instruction, planes 1 MB apart), and would wider rows / a pixel-blocked panel do better?
//
//   shape            layout                 one DMA instruction moves                     slot     tile
//   rows256          NCHW planes            32 lanes x 16 B: 2 rows x 256 B               512 B    64 px    (today: gate GEMM, cand_fused)
//   rows256x4        NCHW planes            64 lanes x 16 B: 4 rows x 256 B               1 KB     64 px    (two k-pairs per instruction)
//   rows512          NCHW planes            64 lanes x 16 B: 2 rows x 512 B               1 KB     128 px   (today: 128-pixel candidate)
//   rows1024         NCHW planes            64 lanes x 16 B: 1 row x 1 KB (2 instr/slot)  2 KB     256 px   (wave pair shares the tile: alternate slots)
//   panel512         [tile][K][64 px]       32 lanes x 16 B: 512 B contiguous             512 B    64 px
//   panel1024        [tile][K][64 px]       64 lanes x 16 B: 1 KB contiguous              1 KB     64 px
//
// Optional per tile: `gap` cycles of s_sleep (a stand-in for the MFMA-only phase 2) and an epilogue store of `erows` rows of the
// tile (NCHW rows or one contiguous block), so that the read stream meets what it meets in the product.
//   hipcc --offload-arch=gfx950 -O3 -o operand_stream operand_stream.hip && ./operand_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
extern __shared__ __attribute__((aligned(16))) char smem[];

enum { ROWS256 = 0, ROWS256X4 = 1, ROWS512 = 2, ROWS1024 = 3, PANEL512 = 4, PANEL1024 = 5 };

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int SHAPE> struct Shape;
template <> struct Shape<ROWS256>   { static constexpr int TILE = 64,  SLOT = 512,  ROWS = 2, NLOAD = 1; };
template <> struct Shape<ROWS256X4> { static constexpr int TILE = 64,  SLOT = 1024, ROWS = 4, NLOAD = 1; };
template <> struct Shape<ROWS512>   { static constexpr int TILE = 128, SLOT = 1024, ROWS = 2, NLOAD = 1; };
template <> struct Shape<ROWS1024>  { static constexpr int TILE = 256, SLOT = 2048, ROWS = 2, NLOAD = 2; };
template <> struct Shape<PANEL512>  { static constexpr int TILE = 64,  SLOT = 512,  ROWS = 2, NLOAD = 1; };
template <> struct Shape<PANEL1024> { static constexpr int TILE = 64,  SLOT = 1024, ROWS = 4, NLOAD = 1; };

struct Args {
    const float *src;     // K planes of P pixels (NCHW) or ceil(P/64) panels of K x 64
    float *dst;           // epilogue target (erows planes / panels)
    float *sink;
    int K, P, tiles, D_unused, gap, erows, epanel;
};

template <int SHAPE, int D>
__global__ __launch_bounds__(512) void stream_kernel(const Args a)
{
    using S = Shape<SHAPE>;
    constexpr bool PAIRED = SHAPE == ROWS1024;            // two waves share a tile and take alternate slots
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char *ring = smem + wave * (D * S::SLOT);
    const unsigned P4 = 4u * (unsigned)a.P;
    const size_t bytes = (size_t)a.K * ((SHAPE >= PANEL512) ? (size_t)((a.P + 63) / 64) * 64 : (size_t)a.P) * 4;
    const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.src), 0, (int)(bytes > 0xffffffffull ? 0xffffffffu : (unsigned)bytes), 0x00020000);
    const int slots_per_tile = a.K / S::ROWS;
    const int wslot = PAIRED ? (blockIdx.x * 8 + wave) >> 1 : blockIdx.x * 8 + wave;      // wave slot in the tile walk
    const int wstep = PAIRED ? gridDim.x * 4 : gridDim.x * 8;
    const int s_first = PAIRED ? (wave & 1) : 0, s_step = PAIRED ? 2 : 1;

    // per-lane byte offset inside a slot's source (constant per tile up to the tile base)
    unsigned lane_off;
    if constexpr (SHAPE == ROWS256) lane_off = ((lane >> 4) & 1) * P4 + (lane & 15) * 16;                 // lanes 0-31 active
    else if constexpr (SHAPE == ROWS256X4) lane_off = (lane >> 4) * P4 + (lane & 15) * 16;
    else if constexpr (SHAPE == ROWS512) lane_off = (lane >> 5) * P4 + (lane & 31) * 16;
    else if constexpr (SHAPE == ROWS1024) lane_off = lane * 16;                                            // + row * P4 per instruction
    else lane_off = lane * 16;                                                                             // panels: contiguous

    // the wave's stream: (tile, slot) pairs in order; issue state and consume state walk it D slots apart
    int it_tile = wslot, it_s = s_first;
    auto tile_base = [&](int t) -> unsigned {
        if constexpr (SHAPE >= PANEL512) return (unsigned)t * (unsigned)a.K * 256u;
        else return (unsigned)t * (unsigned)(S::TILE * 4);
    };
    auto issue = [&](int rslot) {
        const bool live = it_tile < a.tiles;
        unsigned off;
        if constexpr (SHAPE >= PANEL512) off = tile_base(it_tile) + (unsigned)it_s * S::SLOT + lane_off;
        else off = tile_base(it_tile) + (unsigned)it_s * S::ROWS * P4 + lane_off;
        if (!live) off = 0xF0000000u;
        char *dst = ring + rslot * S::SLOT;
        if constexpr (SHAPE == ROWS256 || SHAPE == PANEL512) {
            if (lane < 32) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        } else if constexpr (SHAPE == ROWS1024) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, off, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + 1024), 16, off + P4, 0, 0, 0);
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, off, 0, 0, 0);
        }
        it_s += s_step;
        if (it_s >= slots_per_tile) { it_s = s_first; it_tile += wstep; }
    };

    for (int i = 0; i < D; ++i) issue(i);
    float acc = 0.f;
    int rslot = 0;
    for (int tile = wslot; tile < a.tiles; tile += wstep) {
        for (int s = s_first; s < slots_per_tile; s += s_step) {
            wait_vmcnt<(D - 1) * S::NLOAD>();
            // the lane's own bytes of the slot (what the GEMM's fragment read does)
            if constexpr (S::SLOT == 512) {
                const float2 v = *reinterpret_cast<const float2 *>(ring + rslot * S::SLOT + lane * 8);
                acc += v.x + v.y;
            } else {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(ring + rslot * S::SLOT + lane * 16);
                acc += v.x + v.w;
                if constexpr (S::SLOT == 2048) {
                    const f32x4 u = *reinterpret_cast<const f32x4 *>(ring + rslot * S::SLOT + 1024 + lane * 16);
                    acc += u.x + u.w;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(rslot);
            rslot = rslot + 1 == D ? 0 : rslot + 1;
        }
        if (a.gap) for (int i = 0; i < a.gap; ++i) __builtin_amdgcn_s_sleep(16);     // ~1k cycles each
        if (a.erows && (!PAIRED || true)) {
            // epilogue: erows rows of the tile, 16 B per lane (NCHW: rows P apart; panel: contiguous)
            const int er = PAIRED ? a.erows / 2 : a.erows;
            const int r0 = PAIRED ? (wave & 1) * er : 0;
            const f32x4 val = {acc, acc, acc, acc};
            if (a.epanel) {
                float *base = a.dst + (size_t)tile * a.erows * S::TILE;
                for (int q = lane; q < er * S::TILE / 4; q += 64)
                    __builtin_nontemporal_store(val, reinterpret_cast<f32x4 *>(base + (size_t)r0 * S::TILE) + q);
            } else {
                constexpr int LPR = S::TILE / 4;             // lanes per row
                constexpr int RPI = 64 / LPR < 1 ? 1 : 64 / LPR;   // rows per store instruction
                for (int r = r0; r < r0 + er; r += RPI) {
                    if constexpr (LPR <= 64) {
                        const int rr = r + lane / LPR, px = tile * S::TILE + (lane % LPR) * 4;
                        if (px < a.P && rr < r0 + er) __builtin_nontemporal_store(val, reinterpret_cast<f32x4 *>(a.dst + (size_t)rr * a.P + px));
                    }
                }
            }
        }
    }
    wait_vmcnt<0>();
    if (acc == 12345.678f) a.sink[0] = acc;
}

// plain read ceiling: grid-stride 16 B per lane, U loads in flight per lane
template <int U>
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *src, size_t n16, float *sink)
{
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void copy_kernel(const f32x4 *src, f32x4 *dst, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

static const int NBUF = 3;
static float *g_src[NBUF], *g_dst, *g_sink;

template <int SHAPE, int D>
static void run(const char *name, int K, int P, int gap, int erows, int epanel)
{
    using S = Shape<SHAPE>;
    Args a;
    a.K = K; a.P = P; a.tiles = (P + S::TILE - 1) / S::TILE; a.gap = gap; a.erows = erows; a.epanel = epanel;
    a.dst = g_dst; a.sink = g_sink; a.D_unused = D;
    const size_t lds = (size_t)8 * D * S::SLOT;
    hipFuncSetAttribute(reinterpret_cast<const void *>(stream_kernel<SHAPE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> us;
    for (int rep = 0; rep < 7; ++rep) {
        a.src = g_src[rep % NBUF];
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_kernel<SHAPE, D>), dim3(256), dim3(512), lds, 0, a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) us.push_back(ms * 1e3f);
    }
    std::sort(us.begin(), us.end());
    const double med = us[us.size() / 2];
    const double rd = (double)K * P * 4, wr = (double)erows * P * 4;
    printf("%-10s D=%2d K=%3d gap=%2d erows=%2d%s | %7.1f us  read %5.2f TB/s  read+write %5.2f TB/s  (in flight/CU %5.1f KB)\n", name, D, K, gap, erows,
           epanel ? "p" : " ", med, rd / med / 1e6, (rd + wr) / med / 1e6, 8.0 * D * S::SLOT / 1024);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int P = 250000;
    const size_t bytes = (size_t)224 * 250048 * 4;
    for (int i = 0; i < NBUF; ++i) { hipMalloc(&g_src[i], bytes); hipMemset(g_src[i], 0, bytes); }
    hipMalloc(&g_dst, (size_t)64 * 250112 * 4);
    hipMalloc(&g_sink, 4);
    hipDeviceSynchronize();

    // ceilings
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int blocks : {2048, 8192}) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL((read_kernel<4>), dim3(blocks), dim3(256), 0, 0, reinterpret_cast<const f32x4 *>(g_src[rep % NBUF]), bytes / 16, g_sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("read ceiling (global_load_dwordx4 nt, %4d blocks x 256, 4 in flight): %6.1f us  %5.2f TB/s\n", blocks, best * 1e3, bytes / (best * 1e-3) / 1e12);
        }
        float best = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(copy_kernel, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const f32x4 *>(g_src[rep % NBUF]), reinterpret_cast<f32x4 *>(g_src[(rep + 1) % NBUF]), bytes / 16);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 2 && ms < best) best = ms;
        }
        printf("copy ceiling (8192 blocks x 256): %6.1f us  %5.2f TB/s (read + write)\n", best * 1e3, 2.0 * bytes / (best * 1e-3) / 1e12);
    }

    for (int K : {80, 224}) {
        printf("---- K = %d planes of %d pixels, read stream only\n", K, P);
        run<ROWS256, 8>("rows256", K, P, 0, 0, 0);
        run<ROWS256, 16>("rows256", K, P, 0, 0, 0);
        run<ROWS256X4, 4>("rows256x4", K, P, 0, 0, 0);
        run<ROWS256X4, 8>("rows256x4", K, P, 0, 0, 0);
        run<ROWS512, 4>("rows512", K, P, 0, 0, 0);
        run<ROWS512, 8>("rows512", K, P, 0, 0, 0);
        run<ROWS1024, 4>("rows1024", K, P, 0, 0, 0);
        run<ROWS1024, 8>("rows1024", K, P, 0, 0, 0);
        run<PANEL512, 8>("panel512", K, P, 0, 0, 0);
        run<PANEL512, 16>("panel512", K, P, 0, 0, 0);
        run<PANEL1024, 4>("panel1024", K, P, 0, 0, 0);
        run<PANEL1024, 8>("panel1024", K, P, 0, 0, 0);
        run<PANEL1024, 16>("panel1024", K, P, 0, 0, 0);
        printf("---- K = %d, + epilogue store of 64 rows per tile (gate / candidate GEMM)\n", K);
        run<ROWS256, 8>("rows256", K, P, 0, 64, 0);
        run<ROWS256X4, 8>("rows256x4", K, P, 0, 64, 0);
        run<ROWS512, 8>("rows512", K, P, 0, 64, 0);
        run<ROWS1024, 4>("rows1024", K, P, 0, 64, 0);
        run<PANEL512, 8>("panel512", K, P, 0, 64, 1);
        run<PANEL1024, 8>("panel1024", K, P, 0, 64, 1);
        printf("---- K = %d, + 24k idle cycles per tile (phase 2 + epilogue arithmetic) + the stores\n", K);
        run<ROWS256, 8>("rows256", K, P, 24, 64, 0);
        run<ROWS256X4, 8>("rows256x4", K, P, 24, 64, 0);
        run<PANEL1024, 8>("panel1024", K, P, 24, 64, 1);
        run<PANEL1024, 16>("panel1024", K, P, 24, 64, 1);
    }
    return 0;
}
