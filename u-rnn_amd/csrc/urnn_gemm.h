// urnn_gemm.h -- the weights-stationary GEMM kernel template and its launch machinery, shared by the translation units that
// instantiate it (urnn_gemm.hip: stage convs; urnn_gemm_deconv.hip; urnn_gemm_gates.hip; urnn_gemm_cand.hip; urnn_cand_fused.hip).
// One template, several objects: the instantiations compile in parallel.  Define URNN_TU (a C identifier) before including.
//
//
// Every convolution of the network is 1x1: out[n][p] = sum_k W[n][k] * in[k][p] over the pixels p of an NCHW
// plane.  A wavefront owns a tile of 32*PB pixels x NB*32 output channels ("n-group") and runs v_mfma_f32_32x32x2_f32 with
//   A = weights      (lane l: n = g*NB*32 + nb*32 + (l & 31), k = 2*kp + (l >> 5))
//   B = activations  (lane l: p = pixel(l & 31, pb),          k = 2*kp + (l >> 5))
//   D[n][p] accumulates in AGPRs; epilogues fuse bias, LeakyReLU, AvgPool, the ConvTranspose scatter or the GroupNorm
//   partial statistics.  fp32-input MFMA is bit-exact fp32 FMA (gfx950 has no TF32 path); under this load the chip clocks
//   to about 2.1 GHz.
//
// Structure (measurements in profiles/ and DESIGN.md):
//   * WEIGHTS STATIONARY: a block stages the whole weight slab of ONE n-group (<= 110 KiB, K x 96 columns) into LDS once and
//     then loops over pixel tiles (persistent blocks of 4 or 8 waves, one tile per wave at a time).  Streaming the weights per tile
//     (v2) doubled the bytes through the L2->CU load path, which saturates near 4.7 TB/s and was the measured bottleneck.
//   * ACTIVATIONS are consumed by exactly the lane that loads them, so each wave owns a private LDS ring of D k-pair slots
//     filled by asynchronous LDS-DMA (buffer_load ... lds) D slots ahead and drained with ds_read of the lane's own bytes:
//     a per-lane FIFO -- no bank conflicts, no barriers, no VGPRs spent on prefetch, counted s_waitcnt vmcnt.
//   * blocks b and b + 8 run on the same XCD (observed dispatch b % 8), so the NG n-groups that read the same pixel tiles are
//     placed there: the second group's activation reads hit that XCD's L2.  Placement only affects speed.
//
// One kernel template, conv_gemm_kernel, with five epilogues: stage convs, pooled convs, deconvs, the GRU gate GEMM and the
// GRU candidate GEMM (whose hidden-state rows enter as sigmoid(GN(r)) * h, formed on the fly from two DMA streams).
#pragma once
#include "urnn_common.h"
#include "urnn_kernels.h"

#include <atomic>
#include <stdlib.h>
#include <type_traits>

#ifndef URNN_ABL
#define URNN_ABL 0   // tuning builds only: 2 skip activation DMA, 4 skip epilogue stores, 8 skip LDS fragment reads
#endif
#ifdef URNN_TRACE
// tuning builds: [wave slot][item (8)][8] s_memtime stamps.  One buffer pointer and one setter per translation unit (no relocatable
// device code): urnn_debug_set_trace_<URNN_TU> -- tools/trace_gates.py sets them all.
static __device__ unsigned long long *urnn_trace_buf = nullptr;
#define URNN_TRACE_CAT2(a, b) a##b
#define URNN_TRACE_CAT(a, b) URNN_TRACE_CAT2(a, b)
extern "C" int URNN_TRACE_CAT(urnn_debug_set_trace_, URNN_TU)(unsigned long long *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(urnn_trace_buf), &p, sizeof(p)); }
#define TRACE_STAMP(k) do { if (urnn_trace_buf && lane == 0 && tr_n < 8) urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + tr_n) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TRACE_STAMP(k) do { } while (0)
#endif
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

extern __shared__ __attribute__((aligned(16))) char urnn_smem[];

__device__ __forceinline__ void dma16(const float *g, char *l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
__device__ __forceinline__ void dma4(const float *g, char *l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 4, 0, 0); }

// Buffer-addressed LDS-DMA: address = descriptor base (SGPRs) + per-lane byte offset (tile constant + the uniform row walk,
// one v_add per issue).  Reads past the descriptor's size return 0 -- exactly what the zero-weight pad row of an odd
// channel count and the count-keeping dummy issues need.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void bdma16(rsrc_t r, unsigned voff, unsigned soff, char *l) { __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)l, 16, voff, soff, 0, 0); }
__device__ __forceinline__ void bdma4(rsrc_t r, unsigned voff, unsigned soff, char *l) { __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)l, 4, voff, soff, 0, 0); }

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------
// pixel geometry: which plane offset does (lane column j, pixel block pb) of tile t address?
//   MAP_VEC      p = t*32*PB + (pb/4)*128 + 4*j + pb%4     16-byte loads/stores (planes 16-B aligned, P % 4 == 0)
//   MAP_PAIR     p = t*64 + 2*j + pb            (PB == 2)  8-byte stores, dword DMA (P % 2 == 0)
//   MAP_PAIR16   same pixels; the k-pair's 2 x 64 floats arrive as ONE 16-byte DMA issued by lanes 0-31 (P % 4 == 0)
//   MAP_QUAD16   same pixels; TWO k-pairs (4 rows x 64 floats = 1 KB) arrive as one 16-byte DMA issued by ALL 64 lanes: a DMA
//                instruction costs the CU the same whether it moves 512 B or 1 KB (tools/ubench/operand_stream.hip: 2.0-2.7 TB/s
//                against 5.3-5.9), so the half-wave form of MAP_PAIR16 streams at half the rate.  Split k-loops only.
//   MAP_STRIDED  p = t*32*PB + 32*pb + j                    dword everything (any P)
//   MAP_POOL     pooled pixel q = t*32 + j, input pixel (2*y2 + pb/2, 2*x2 + pb%2)   (PB == 4)
// ------------------------------------------------------------------------------------------------------------------
constexpr bool is_pair(int map) { return map == MAP_PAIR || map == MAP_PAIR16 || map == MAP_QUAD16; }

template <int MAP, int PB>
struct PixelMap {
    int off[PB];     // clamped (always in-bounds) offsets inside an input plane
    bool valid[PB];  // false: out of range, contributes nothing and is never stored
    int q;           // MAP_POOL: pooled output pixel index
    int dma_off;     // MAP_PAIR16: plane offset of the 4 pixels this lane's DMA moves (lanes 0-31 only)

    __device__ __forceinline__ void init(int tile, int j, int P, int W, int P2, int W2)
    {
        q = 0;
        dma_off = 0;
        if constexpr (MAP == MAP_PAIR16 || MAP == MAP_QUAD16) {
            const int p4 = tile * 64 + 4 * (j & 15);          // lanes 0-15 -> row k, lanes 16-31 -> row k+1 (same pixels; QUAD16: lanes 32-63 -> rows k+2, k+3)
            dma_off = p4 < P ? p4 : 0;
        }
        if constexpr (MAP == MAP_POOL) {
            static_assert(PB == 4, "pool tiles are 2x2 input pixels per lane");
            q = tile * 32 + j;
            const bool ok = q < P2;
            const int qq = ok ? q : 0;
            const int y2 = qq / W2, x2 = qq - y2 * W2;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                off[pb] = (2 * y2 + (pb >> 1)) * W + 2 * x2 + (pb & 1);
                valid[pb] = ok;
            }
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                int p;
                if constexpr (MAP == MAP_VEC) p = tile * (32 * PB) + (pb >> 2) * 128 + 4 * j + (pb & 3);
                else if constexpr (is_pair(MAP)) p = tile * 64 + 2 * j + pb;
                else p = tile * (32 * PB) + 32 * pb + j;
                valid[pb] = p < P;
                off[pb] = valid[pb] ? p : 0;
            }
        }
    }
};

// Plain (non-DMA) row access used by epilogues: PB values of one channel row at the tile's pixels.
template <int MAP, int PB>
__device__ __forceinline__ void load_row(const float *row, const PixelMap<MAP, PB> &pm, float (&v)[PB])
{
    if constexpr (MAP == MAP_VEC) {
#pragma unroll
        for (int qd = 0; qd < PB / 4; ++qd) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(row + pm.off[4 * qd]);
            v[4 * qd] = t.x; v[4 * qd + 1] = t.y; v[4 * qd + 2] = t.z; v[4 * qd + 3] = t.w;
        }
    } else if constexpr (is_pair(MAP)) {
        const f32x2 t = *reinterpret_cast<const f32x2 *>(row + pm.off[0]);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) v[pb] = row[pm.off[pb]];
    }
}

template <int MAP, int PB>
__device__ __forceinline__ void store_row(float *row, const PixelMap<MAP, PB> &pm, const float (&v)[PB])
{
#if (URNN_ABL & 4)
    if (pm.off[0] != -12345) { asm volatile("" ::"v"(v[0])); return; }
#endif
    if constexpr (MAP == MAP_VEC) {
#pragma unroll
        for (int qd = 0; qd < PB / 4; ++qd)
            if (pm.valid[4 * qd]) {
                // 128-pixel tiles are the big planes: tens of MB per launch that no kernel re-reads out of the 4-MiB L2s.  Non-temporal
                // stores keep them from evicting the activation rows the other n-group is about to re-read (+2.7 % frames/s)
#ifndef URNN_PLAIN_STORES
                __builtin_nontemporal_store(f32x4{v[4 * qd], v[4 * qd + 1], v[4 * qd + 2], v[4 * qd + 3]}, reinterpret_cast<f32x4 *>(row + pm.off[4 * qd]));
#else
                *reinterpret_cast<f32x4 *>(row + pm.off[4 * qd]) = f32x4{v[4 * qd], v[4 * qd + 1], v[4 * qd + 2], v[4 * qd + 3]};
#endif
            }
    } else if constexpr (is_pair(MAP)) {
        if (pm.valid[0]) *reinterpret_cast<f32x2 *>(row + pm.off[0]) = f32x2{v[0], v[1]};
    } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
            if (pm.valid[pb]) row[pm.off[pb]] = v[pb];
    }
}

// store_row with non-temporal stores on the pair maps: for the fused candidate kernel, whose 64 output planes would otherwise push the
// rows it is about to re-read (the tile's hidden state for phase 2, the next tile's inputs) out of the XCD's L2 -- same-box A/B of
// the two libraries: cand_fused_kernel enc1 57.5 -> 55.4 us, dec1 88.8 -> 81.3 us.  Not for the gate GEMMs (full resolution: no
// change; half resolution: 32.7 -> 35.5, 38.3 -> 41.2 us) nor the half-resolution candidate (no change): they keep store_row.
template <int MAP, int PB>
__device__ __forceinline__ void store_row_nt(float *row, const PixelMap<MAP, PB> &pm, const float (&v)[PB])
{
    if constexpr (is_pair(MAP)) {
#if (URNN_ABL & 4)
        if (pm.off[0] != -12345) { asm volatile("" ::"v"(v[0])); return; }
#endif
        if (pm.valid[0]) __builtin_nontemporal_store(f32x2{v[0], v[1]}, reinterpret_cast<f32x2 *>(row + pm.off[0]));
    } else {
        store_row<MAP, PB>(row, pm, v);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-wave activation ring.  One slot = one k-pair = PB*256 B; lane l's own bytes sit at  quad*1024 + l*16  (16-B DMA) or
// pb*256 + l*4  (dword DMA).
// ------------------------------------------------------------------------------------------------------------------
template <int PB, int MAP>
struct Ring {
    static constexpr bool VEC = (MAP == MAP_VEC);
    static constexpr bool Q16 = (MAP == MAP_QUAD16);
    static constexpr bool P16 = (MAP == MAP_PAIR16) || Q16;
    static constexpr int KPS = Q16 ? 2 : 1;                 // k-pairs per slot (= per DMA instruction)
#if (URNN_ABL & 2)
    static constexpr int NLOAD = 0;
#else
    static constexpr int NLOAD = P16 ? 1 : (VEC ? PB / 4 : PB);   // DMA instructions per slot
#endif
    static constexpr int SLOT = PB * 256 * KPS;             // bytes

    // Which of the k-pair's two channel rows a lane fetches: lanes 32-63 the lower one, except MAP_PAIR16 where lanes 16-31 do.
    __device__ static __forceinline__ int row_select(int lane) { return Q16 ? lane >> 4 : (P16 ? (lane >> 4) & 1 : lane >> 5); }

    // Per-lane byte offsets (constant for a tile) of the DMA pieces of one k-pair, relative to the k-pair's first row.
    static constexpr int NV = P16 ? 1 : (VEC ? PB / 4 : PB);
    __device__ static __forceinline__ void lane_offsets(const PixelMap<MAP, PB> &pm, int lane, unsigned P, unsigned (&vo)[NV])
    {
        const unsigned rowb = (unsigned)row_select(lane) * 4u * P;
        if constexpr (P16) vo[0] = rowb + 4u * (unsigned)pm.dma_off;
        else if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) vo[qd] = rowb + 4u * (unsigned)pm.off[4 * qd];
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) vo[pb] = rowb + 4u * (unsigned)pm.off[pb];
        }
    }

    __device__ static __forceinline__ void issue(char *slot, rsrc_t r, const unsigned (&vo)[NV], unsigned soff, int lane)
    {
#if (URNN_ABL & 2)
        return;
#endif
        // the row walk rides in the VECTOR offset (one v_add): gfx9 range-checks only vector + immediate offset, and the pad
        // row / past-the-end dummies rely on out-of-range reads returning 0 instead of touching memory
        if constexpr (Q16) {
            static_assert(PB == 2, "pair tiles");
            bdma16(r, vo[0] + soff, 0, slot);                 // 64 lanes x 16 B = rows k .. k+3 of the 64-pixel tile (two k-pairs)
        } else if constexpr (P16) {
            static_assert(PB == 2, "pair tiles");
            if (lane < 32) bdma16(r, vo[0] + soff, 0, slot);  // 32 lanes x 16 B = rows k and k+1 of the 64-pixel tile
        } else if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) bdma16(r, vo[qd] + soff, 0, slot + qd * 1024);
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) bdma4(r, vo[pb] + soff, 0, slot + pb * 256);
        }
    }

    // sub: which k-pair of the slot (MAP_QUAD16 only)
    __device__ static __forceinline__ void read(const char *slot, int lane, float (&b)[PB], int sub = 0)
    {
        if constexpr (P16) {
            const f32x2 t = *reinterpret_cast<const f32x2 *>(slot + sub * 512 + ((lane >> 5) * 64 + 2 * (lane & 31)) * 4);
            b[0] = t.x; b[1] = t.y;
        } else if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(slot + qd * 1024 + lane * 16);
                b[4 * qd] = t.x; b[4 * qd + 1] = t.y; b[4 * qd + 2] = t.z; b[4 * qd + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) b[pb] = *reinterpret_cast<const float *>(slot + pb * 256 + lane * 4);
        }
    }
};

// Stage `nfloats` (a multiple of 256) floats from global into LDS with the block's 4 waves, 1 KiB per DMA instruction.
__device__ __forceinline__ void stage_weights(const float *src, char *dst, int nfloats, int wave, int nwaves, int lane)
{
    for (int c = wave; c < nfloats / 256; c += nwaves) dma16(src + (size_t)c * 256 + lane * 4, dst + c * 1024);
}

// block -> (n-group, tile slot): blocks b and b + 8 share an XCD; the NG groups of one tile slot are placed there.
__device__ __forceinline__ void block_role(int NG, int &g, int &slot, int &nslots)
{
    const int xcd = blockIdx.x & 7, r = blockIdx.x >> 3;
    g = r % NG;
    slot = (r / NG) * 8 + xcd;
    nslots = gridDim.x / NG;
}

// ------------------------------------------------------------------------------------------------------------------
// conv_gemm_kernel: persistent blocks of WPB waves; block owns n-group g (weights resident in LDS), each wave loops over
// pixel tiles.  gridDim.x is a multiple of 8 * NG.  Dynamic LDS = aFloats*4 + WPB * (D + 1) * SLOT + NB * 128 (bias).
// ------------------------------------------------------------------------------------------------------------------
// SPLIT = 0: v_mfma_f32_32x32x2_f32 per k-pair (exact fp32 fma chain).  SPLIT = 1: eight k-pairs at a time on
// v_mfma_f32_32x32x16_bf16 with both operands split into three bf16 pieces (six MFMAs per 16 k: 2.67x the fp32 matrix rate);
// needs (KT - kpBegin) % 8 == 0 and, in the candidate GEMM, a plain part that is a positive multiple of 8 k-pairs.
// SPLIT = 2: bf16 COMPUTE (urnn_set_matrix_mode(URNN_MATRIX_BF16), BASELINE configs[3]): activations rounded to bf16 (RNE), weights
// kept to 16 mantissa bits (hi + mid pieces), fp32 accumulate -- two MFMAs per 16 k; statistics, norms, states stay fp32.
// SPLIT = 3: f16 x 3 (urnn_common.h): both operands as two scaled f16 pieces, three v_mfma_f32_32x32x16_f16 per 16 k, fp32-class
// accuracy at half the matrix work and a third of the VALU work of SPLIT = 1 -- the forward path's arithmetic; SPLIT = 1 stays
// for the backward pass's gradients (prm.wide), whose magnitudes have no lower bound.
template <int NB, int PB, int MAP, int EPI, int D, int WPB, int SPLIT>
__global__ __launch_bounds__(64 * WPB, WPB / 4) void conv_gemm_kernel(const ConvGemmParams prm)
{
    using R = Ring<PB, MAP>;
    static_assert(!(R::Q16 && SPLIT == 0), "MAP_QUAD16 exists for the split k-loops only");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    int g, slot0, nslots;
    block_role(prm.NG, g, slot0, nslots);

    const float *A = reinterpret_cast<const float *>(urnn_smem);       // [KT][NB][64]  (SPLIT: [KT/8][NB][3][64] x 16 B of bf16 pieces)
    const size_t slabBytes = SPLIT == 3 ? (size_t)prm.fDwords * 4 : (SPLIT ? (size_t)prm.sDwords * 4 : (size_t)prm.aFloats * 4);
    char *ring = urnn_smem + slabBytes + wave * ((D + 1) * R::SLOT);
    char *scratch = ring + D * R::SLOT;                                // one extra slot: sink for the count-keeping dummy DMAs
    const int kp_begin = prm.kpBegin, KT = prm.KT;
    const int n0 = g * (NB * 32);
    // The group's bias row lives in LDS too: a GLOBAL load inside the epilogue would put an s_waitcnt vmcnt(0) in front of
    // every store (loads and stores share the counter), i.e. one full memory round trip per stored row -- measured 1000
    // cycles per store instruction, 49k cycles per 48-KiB tile epilogue.
    float *bias = reinterpret_cast<float *>(urnn_smem + slabBytes + WPB * ((D + 1) * R::SLOT));
    float *ssm = bias + NB * 32;                                       // EPI_CAND: [B][F][2] r-gate (scale, shift)
    // epilogue reads of the bias row: ONE lane-dependent base (+ 4 * half) and compile-time row offsets, so that the reads are
    // ds_read with immediate offsets instead of 16 * NB precomputed address registers kept live across the k-loop
    const float *bias_h = bias + 4 * half;
    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    constexpr bool GATED = (EPI == EPI_CAND);
    // epilogue value of an accumulator element: the f16 path's operands were scaled by 2^(AEXP + WEXP)
    auto fin = [](float a, float bv) {
        if constexpr (SPLIT == 3) return fmaf(a, URNN_F16_DESCALE, bv);
        else return a + bv;
    };
    if constexpr (SPLIT == 3) stage_weights(reinterpret_cast<const float *>(prm.wf16) + (size_t)g * prm.fDwords, urnn_smem, prm.fDwords, wave, WPB, lane);
    else if constexpr (SPLIT) stage_weights(reinterpret_cast<const float *>(prm.wsplit) + (size_t)g * prm.sDwords, urnn_smem, prm.sDwords, wave, WPB, lane);
    else stage_weights(prm.wt + (size_t)g * prm.aFloats, urnn_smem, prm.aFloats, wave, WPB, lane);
    if (threadIdx.x < NB * 32) bias[threadIdx.x] = (EPI == EPI_GRU1 && SPLIT == 3 && prm.biasf ? prm.biasf : prm.bias)[n0 + threadIdx.x];
    if constexpr (EPI == EPI_LRELU && NB == 1 && PB == 4 && MAP == MAP_VEC) {
        if (prm.stemW && threadIdx.x < 256) ssm[threadIdx.x] = prm.stemW[threadIdx.x];      // the head's stem conv, for the epilogue's statistics
    }
    if constexpr (GATED) {
        // GroupNorm of the gates is finalised HERE instead of in a launch of its own: one wave per (sample, 32-channel
        // group) folds the gate GEMM's per-tile (sum, sumsq) partials in double, in a fixed order (lane-strided, then an xor
        // butterfly) -- every block computes identical bits.  The reset-gate rows' (scale, shift) stay in LDS for the gated
        // fragments; block 0 also publishes the whole table for the blend kernel.
        const int F = prm.F, G1 = 2 * F / 32;
        for (int q = wave; q < prm.B * G1; q += WPB) {
            const int b = q / G1, grp = q - b * G1;
            const float *pp = prm.gpart + ((size_t)b * G1 + grp) * prm.gtiles * 2;
            double s1, s2;
            fold_lane_chain<32>(pp, prm.gtiles, prm.gtilePix, 32, prm.P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                s1 += __shfl_xor(s1, m, 64);
                s2 += __shfl_xor(s2, m, 64);
            }
            const double mean = s1 / prm.gcount;
            double var = s2 / prm.gcount - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
            var = var > 0.0 ? var : 0.0;
            const double rstd = 1.0 / sqrt(var + (double)prm.eps);
            if (lane < 32) {
                const int c = grp * 32 + lane;
                const double sc = (double)prm.gn_w[c] * rstd;
                const float fsc = (float)sc, fsh = (float)((double)prm.gn_b[c] - nofma(mean * sc));
                if (c >= F) {
                    ssm[((size_t)b * F + (c - F)) * 2] = fsc;
                    ssm[((size_t)b * F + (c - F)) * 2 + 1] = fsh;
                }
                if (blockIdx.x == 0) {
                    prm.ss_out[((size_t)b * 2 * F + c) * 2] = fsc;
                    prm.ss_out[((size_t)b * 2 * F + c) * 2 + 1] = fsh;
                    if (lane == 0) flag_nonfinite(prm.status, URNN_STATUS_GATES, s1, s2);
                    if (lane == 0 && prm.stat_out) {
                        prm.stat_out[((size_t)b * G1 + grp) * 2] = (float)mean;
                        prm.stat_out[((size_t)b * G1 + grp) * 2 + 1] = (float)rstd;
                    }
                }
            }
        }
    }
    wait_vmcnt<0>();
    __syncthreads();

    if constexpr (WPB > 4) {
        // Waves w and w + 4 share a SIMD.  Start offset for the later waves: prm.stagger naps of s_sleep 64 (~4 096 cycles each).  Every
        // wave of the chip starts its first tile at the same time, so all first epilogues -- 2 048 waves x 16 KB of stores -- fall
        // together, and a wave's next loads queue behind its own stores (one in-order counter): DESIGN 4.16.  A nap or two take the two
        // waves of a SIMD far enough apart where a wave has two tiles (enc1 gates, same box: 49.8 us without, 46.5 with one nap, 44.4 with
        // two; dec1 unchanged); any offset where a wave has a single tile only adds a tail (dec2 37.5 -> 41.6 us at two naps).  The
        // launcher decides (profiles/r05_ab_gate_stagger.txt).
        if (wave >= 4 && prm.stagger) {
            for (int i = 0; i < prm.stagger; ++i) __builtin_amdgcn_s_sleep(64);
        }
    }

    int tr_n = 0;
    (void)tr_n;
    for (int item = slot0 * WPB + wave; item < prm.totalTiles; item += nslots * WPB) {
        TRACE_STAMP(0);
        const int b = item / prm.tilesPerSample;
        const int tile = item - b * prm.tilesPerSample;
        PixelMap<MAP, PB> pm;
        pm.init(tile, j, prm.P, prm.W, prm.P2, prm.W2);

        f32x16 acc[NB][PB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

        // Activation rows are addressed through one buffer descriptor per K segment (x | e | h): row k-pair kp of a segment sits
        // at uniform offset 2*(kp - kp0)*P*4; the per-lane part (row select + pixel offset) is constant for the tile.  A
        // segment with an odd channel count ends in a pad row (zero weight): it lies past the descriptor and reads as 0.
        //
        // The ring is fed by a SLOT STREAM: one slot = one row pair of one input.  A plain k-pair takes one slot.  In the
        // candidate GEMM (EPI_CAND) the k-pairs of the hidden-state segment are GATED: B = sigmoid(GN(r)) * h, so they take
        // two consecutive slots (raw r-gate rows, then h rows) and the product is formed when the fragment is read.
        const int k1 = prm.segKp0[1], k2 = prm.segKp0[2];
        const int kpe = GATED ? (prm.hKp0 < KT ? prm.hKp0 : KT) : KT;      // end of the plain k-pairs
        const int nplain = kpe > kp_begin ? kpe - kp_begin : 0;
        const int total = nplain + (GATED ? 2 * (KT - kpe) : 0);           // slots this tile streams
        const rsrc_t rs0 = make_rsrc(prm.seg[0] + (size_t)b * prm.segC[0] * prm.P, 4u * (unsigned)prm.segC[0] * (unsigned)prm.P);
        const rsrc_t rs1 = make_rsrc(prm.seg[1] + (size_t)b * prm.segC[1] * prm.P, 4u * (unsigned)prm.segC[1] * (unsigned)prm.P);
        const rsrc_t rs2 = make_rsrc(prm.seg[2] + (size_t)b * prm.segC[2] * prm.P, 4u * (unsigned)prm.segC[2] * (unsigned)prm.P);
        const rsrc_t rsg = GATED ? make_rsrc(prm.gate + ((size_t)b * 2 * prm.F + prm.F) * prm.P, 4u * (unsigned)prm.F * (unsigned)prm.P) : rs0;
        const float *ssb = ssm + (size_t)b * 2 * prm.F;
        unsigned vo[R::NV];
        R::lane_offsets(pm, lane, (unsigned)prm.P, vo);
        const unsigned rstep = 8u * (unsigned)prm.P;                       // two channel rows, bytes
        int si = 0;                                                        // next slot of the stream to issue
        const int s_begin = kp_begin >= k2 ? 2 : (kp_begin >= k1 ? 1 : 0);
        rsrc_t rs = s_begin == 2 ? rs2 : (s_begin == 1 ? rs1 : rs0);       // current plain segment
        unsigned soff = rstep * (unsigned)(kp_begin - (s_begin == 2 ? k2 : (s_begin == 1 ? k1 : 0)));
        unsigned soff_g = 0;                                               // gated part: byte offset of the next row pair
        // Branch-free refills (the k loop must stay one basic block so that its instruction order can be pinned): past the
        // end of the stream the DMA goes to a scratch slot (and reads out of range = zeros), which keeps the outstanding-DMA
        // count -- and therefore every s_waitcnt immediate -- exact.  Scalar ALU only.
        auto refill_plain = [&](int slot) {
            const bool live = si < total;
            char *dst = live ? ring + slot * R::SLOT : scratch;
            R::issue(dst, rs, vo, live ? soff : 0xF0000000u, lane);
            ++si;
            soff += rstep;
            const bool sw1 = si == k1 - kp_begin, sw2 = !GATED && si == k2 - kp_begin;
            rs = sw2 ? rs2 : (sw1 ? rs1 : rs);
            soff = (sw1 || sw2) ? 0u : soff;
        };
        // gated part of the stream: slots alternate r-gate rows / h rows of the same row pair
        auto refill_gate = [&](int slot, bool hrows) {
            const bool live = si < total;
            char *dst = live ? ring + slot * R::SLOT : scratch;
            R::issue(dst, hrows ? rs2 : rsg, vo, live ? soff_g : 0xF0000000u, lane);
            ++si;
            soff_g += hrows ? rstep : 0u;
        };
        auto wrap = [](int s_) { return s_ >= D ? s_ - D : s_; };
        auto read_plain = [&](int kp, int slot, float (&a)[NB], float (&bv)[PB]) {
#if (URNN_ABL & 8)
            if (kp > kp_begin) {   // tuning build: keep the first fragments, skip the LDS traffic
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) asm volatile("" : "+v"(a[nb]));
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) asm volatile("" : "+v"(bv[pb]));
                return;
            }
#endif
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = A[(kp * NB + nb) * 64 + lane];
            R::read(ring + slot * R::SLOT, lane, bv);
        };
        auto read_gated = [&](int kp, int slot, float (&a)[NB], float (&bv)[PB]) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[nb] = A[(kp * NB + nb) * 64 + lane];
            float gg[PB], hh[PB];
            R::read(ring + slot * R::SLOT, lane, gg);
            R::read(ring + wrap(slot + 1) * R::SLOT, lane, hh);
            const f32x2 st = *reinterpret_cast<const f32x2 *>(ssb + 2 * (2 * (kp - kpe) + half));
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) bv[pb] = gate_sigmoid(gg[pb], st.x, st.y) * hh[pb];
        };

        if constexpr (SPLIT) {
            // ---- bf16 x 6 path.  The ring protocol (slots, counted waits, refills D slots ahead) is the fp32 one; a k-pair's
            // activation fragments are split into bf16 pieces as they arrive (even / odd k-pair -> low / high half of a dword),
            // the weights' pieces are read ready-made from LDS, and every eighth k-pair the 6 * NB * PB MFMAs of the 16-k group
            // are issued.  Lane l holds, as element j of its A / B vectors, k = 2 * (8 * group + j) + (l >> 5): any assignment
            // works as long as A and B agree.  Refills use uniform branches (no pinned schedule here): segment switches are rare.
            int seg_left = (kp_begin >= k2 ? INT_MAX : (kp_begin >= k1 ? (k2 == INT_MAX || GATED ? INT_MAX : k2 - kp_begin) : k1 == INT_MAX ? INT_MAX : k1 - kp_begin));
            int seg_cur = s_begin;
            // The common case -- the next row pair of the current plain segment, or a dummy past the end of the stream -- is
            // straight-line code (two scalar selects); segment switches are rare, not-taken branches.
            // MAP_QUAD16: a slot (one DMA instruction) holds TWO k-pairs; si, nplain_s and seg_left count slots.  Segments start at even
            // k-pairs (the launcher checks), so a slot never straddles two inputs.
            static_assert(!(R::Q16 && GATED), "the two-stream candidate GEMM keeps one k-pair per slot");
            constexpr int KPS = R::KPS;
            const int nplain_s = nplain / KPS;
            if constexpr (KPS == 2) seg_left = seg_left == INT_MAX ? INT_MAX : seg_left / 2;
            auto refill_s = [&](int slot) {
                if (GATED && si >= nplain && si < total) {            // candidate GEMM: half of its stream
                    const bool hrows = ((si - nplain) & 1) != 0;
                    R::issue(ring + slot * R::SLOT, hrows ? rs2 : rsg, vo, soff_g, lane);
                    soff_g += hrows ? rstep : 0u;
                    ++si;
                    return;
                }
                const bool live = si < nplain_s;                  // else past the end: the DMA reads out of range (zeros) into the
                R::issue(live ? ring + slot * R::SLOT : scratch, rs, vo, live ? soff : 0xF0000000u, lane);   // sink slot, which keeps
                soff += rstep * KPS;                                                                            // every vmcnt exact
                ++si;
                if (__builtin_expect(--seg_left == 0, 0)) {      // next K segment (x -> e -> h)
                    ++seg_cur;
                    rs = seg_cur == 1 ? rs1 : rs2;
                    soff = 0u;
                    seg_left = (seg_cur == 1 && !GATED && k2 != INT_MAX) ? (k2 - k1) / KPS : INT_MAX;
                }
            };
#ifdef URNN_POISON
            // diagnosis build: a ring slot holds NaN from the moment its fragment has been consumed until its DMA lands, so a read
            // that gets ahead of the DMA (whatever s_waitcnt said) shows up as NaN in the output instead of as plausible stale data
            auto poison = [&](int slot_) {
                const float qn = __builtin_nanf("");
#pragma unroll
                for (int qd = 0; qd < R::SLOT / 1024 + (R::SLOT % 1024 ? 1 : 0); ++qd)
                    if (qd * 1024 + lane * 16 < R::SLOT) *reinterpret_cast<f32x4 *>(ring + slot_ * R::SLOT + qd * 1024 + lane * 16) = f32x4{qn, qn, qn, qn};
            };
            for (int i = 0; i < D; ++i) poison(i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            for (int i = 0; i < D; ++i) refill_s(i);
            constexpr bool BF16C = (SPLIT == 2), F16 = (SPLIT == 3);
            constexpr int NPC = F16 ? 2 : 3;               // weight pieces per n-block in the LDS slab
            float rb[2][PB];                               // raw fp32 activation fragments: rb[0] even k-pairs, rb[1] odd ones
            unsigned bh[PB][4], bm[(BF16C || F16) ? 1 : PB][4], bl[BF16C ? 1 : PB][4];
            const float asc = URNN_F16_ASCALE;
            const char *Ap = urnn_smem + lane * 16;
            auto read_b = [&](int kp, int slot_, bool gated, float (&bv)[PB]) {
                if (!gated) {
                    R::read(ring + slot_ * R::SLOT, lane, bv);
                } else {
                    float gg[PB], hh[PB];
                    R::read(ring + slot_ * R::SLOT, lane, gg);
                    R::read(ring + wrap(slot_ + 1) * R::SLOT, lane, hh);
                    const f32x2 st = *reinterpret_cast<const f32x2 *>(ssb + 2 * (2 * (kp - kpe) + half));
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) bv[pb] = gate_sigmoid(gg[pb], st.x, st.y) * hh[pb];
                }
            };
            wait_vmcnt<(D - 1) * R::NLOAD>();
            read_b(kp_begin, 0, false, rb[0]);
            TRACE_STAMP(1);
            int slot = 0;
            auto sstep = [&](int kp, auto q_tag, auto cur_tag, auto nxt_tag) {
                constexpr int Q = decltype(q_tag)::value;
                constexpr bool CG = decltype(cur_tag)::value, NGT = decltype(nxt_tag)::value;
                const int nslot = wrap(slot + (CG ? 2 : 1));
                if constexpr (Q & 1) {
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        if constexpr (BF16C) bh[pb][Q >> 1] = round_pair(rb[0][pb], rb[1][pb]);
                        else if constexpr (F16) split2_pair(rb[0][pb], rb[1][pb], asc, bh[pb][Q >> 1], bl[pb][Q >> 1]);
                        else split_pair(rb[0][pb], rb[1][pb], bh[pb][Q >> 1], bm[pb][Q >> 1], bl[pb][Q >> 1]);
                    }
                }
                if constexpr (R::Q16) {
                    // two k-pairs per slot: an even step reads the slot's second k-pair (landed with the first); an odd step has both
                    // fragments of the slot in registers, waits for the next slot, hands this one back to the DMA and reads on
                    if constexpr ((Q & 1) == 0) {
                        R::read(ring + slot * R::SLOT, lane, rb[1], 1);
                    } else {
                        wait_vmcnt<(D - 2) * R::NLOAD>();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (both reads of the slot have left LDS: hazard note below)
                        refill_s(slot);
                        R::read(ring + nslot * R::SLOT, lane, rb[0], 0);
                    }
                } else {
                wait_vmcnt<(D - (CG ? 2 : 1) - (NGT ? 2 : 1)) * R::NLOAD>();   // the next k-pair's slot(s) have landed (or are dummies)
                // The slot(s) about to be refilled were read one step ago, but an even k-pair's fragments are not CONSUMED before
                // the next odd step, so nothing has waited for that ds_read yet: without this wait the DMA could (rarely: one
                // launch in ~30) overwrite the slot before the read had left LDS.  The read is a step old: the wait is free.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if defined(URNN_POISON) && URNN_POISON >= 2
                poison(slot);
                if constexpr (CG) poison(wrap(slot + 1));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                refill_s(slot);
                if constexpr (CG) refill_s(wrap(slot + 1));
                read_b(kp + 1 < KT ? kp + 1 : kp, nslot, NGT, rb[(Q + 1) & 1]);
                }
                if constexpr (Q == 7) {
                    const char *ag = Ap + (size_t)(kp >> 3) * (NB * NPC * 1024);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        if constexpr (F16) {
                            const f16x8 fh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * 2 + 0) * 1024));
                            const f16x8 fl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * 2 + 1) * 1024));
                            auto mf = [&](const f16x8 &wa, const unsigned (&pbv)[PB][4]) {
#pragma unroll
                                for (int pb = 0; pb < PB; ++pb)
                                    acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, as_f16x8(pbv[pb]), acc[nb][pb], 0, 0, 0);
                            };
                            mf(fl, bh); mf(fh, bl); mf(fh, bh);                                          // small terms first
                        } else {
                        const bf16x8 wh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * NPC + 0) * 1024));
                        const bf16x8 wm = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * NPC + 1) * 1024));
                        auto mm = [&](const bf16x8 &wa, const unsigned (&pbv)[PB][4]) {
#pragma unroll
                            for (int pb = 0; pb < PB; ++pb)
                                acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, as_bf16x8(pbv[pb]), acc[nb][pb], 0, 0, 0);
                        };
                        if constexpr (BF16C) {
                            mm(wm, bh); mm(wh, bh);
                        } else {
                            const bf16x8 wl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * NPC + 2) * 1024));
                            mm(wm, bm); mm(wl, bh); mm(wh, bl); mm(wm, bh); mm(wh, bm); mm(wh, bh);     // small terms first
                        }
                        }
                        __builtin_amdgcn_sched_barrier(0);       // keep the next n-block's weight pieces from being loaded early (registers)
                    }
                }
                if constexpr (!R::Q16 || (Q & 1)) slot = nslot;
            };
            using std::false_type;
            using std::true_type;
            auto group = [&](int kp0, auto cur_tag, auto last_nxt_tag) {
                sstep(kp0 + 0, std::integral_constant<int, 0>{}, cur_tag, cur_tag);
                sstep(kp0 + 1, std::integral_constant<int, 1>{}, cur_tag, cur_tag);
                sstep(kp0 + 2, std::integral_constant<int, 2>{}, cur_tag, cur_tag);
                sstep(kp0 + 3, std::integral_constant<int, 3>{}, cur_tag, cur_tag);
                sstep(kp0 + 4, std::integral_constant<int, 4>{}, cur_tag, cur_tag);
                sstep(kp0 + 5, std::integral_constant<int, 5>{}, cur_tag, cur_tag);
                sstep(kp0 + 6, std::integral_constant<int, 6>{}, cur_tag, cur_tag);
                sstep(kp0 + 7, std::integral_constant<int, 7>{}, cur_tag, last_nxt_tag);
            };
            if constexpr (GATED) {
                static_assert(D >= 6 && D % 2 == 0, "a gated k-pair and its successor hold four slots");
                int kp = kp_begin;
                for (; kp + 8 < kpe; kp += 8) group(kp, false_type{}, false_type{});
                group(kp, false_type{}, true_type{});                          // last plain group: its last step reads a gated k-pair
                for (kp += 8; kp < KT; kp += 8) group(kp, true_type{}, true_type{});
            } else {
                for (int kp = kp_begin; kp < KT; kp += 8) group(kp, false_type{}, false_type{});
            }
        } else {
        // Software pipeline.  A wave issues in order and each fp32 MFMA occupies the pipe for 64 cycles, so everything that is
        // not an MFMA must sit BETWEEN MFMAs (about ten instruction slots hide behind each one) -- traced with s_memtime, the
        // version that did its LDS reads, pointer math and DMA issue after the last MFMA of a k-pair lost 300 of every 1070
        // cycles.  Order per k-pair, pinned with sched_barrier: [wait + LDS reads of the NEXT k-pair] MFMAs(nb 0) [refill DMA of
        // the slot(s) just consumed] MFMAs(nb 1) [bookkeeping] MFMAs(nb 2) [fragment hand-over].
        for (int i = 0; i < D; ++i) {
            if (!GATED || si < nplain) refill_plain(i);
            else refill_gate(i, ((si - nplain) & 1) != 0);
        }
        float a_cur[NB], b_cur[PB], a_nxt[NB], b_nxt[PB];
        if (!GATED || nplain > 0) {
            wait_vmcnt<(D - 1) * R::NLOAD>();
            read_plain(kp_begin, 0, a_cur, b_cur);
        } else {
            wait_vmcnt<(D - 2) * R::NLOAD>();
            read_gated(kp_begin, 0, a_cur, b_cur);
        }
        TRACE_STAMP(1);
        int slot = 0;
        // One k-pair.  CG / NGT: the current / next k-pair is gated (two slots).  RF: which part of the stream the slots issued
        // here belong to -- 0 plain, 1 gated with run-time parity (the last D plain k-pairs already prefetch the gated part),
        // 2 gated with static parity (inside the gated part si - nplain is even at the top of every k-pair, D being even).
        auto step = [&](int kp, auto cur_tag, auto nxt_tag, auto rf_tag) {
            constexpr bool CG = decltype(cur_tag)::value, NGT = decltype(nxt_tag)::value;
            constexpr int RF = decltype(rf_tag)::value;
            const int nslot = wrap(slot + (CG ? 2 : 1));
            wait_vmcnt<(D - (CG ? 2 : 1) - (NGT ? 2 : 1)) * R::NLOAD>();   // the next k-pair's slot(s) have landed (or are dummies)
            const int kn = kp + 1 < KT ? kp + 1 : kp;
            if constexpr (NGT) read_gated(kn, nslot, a_nxt, b_nxt);
            else read_plain(kn, nslot, a_nxt, b_nxt);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[0][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], b_cur[pb], acc[0][pb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // refill the slot(s) just consumed: their fragments are already in registers
            if constexpr (RF == 0) refill_plain(slot);
            else if constexpr (RF == 1) refill_gate(slot, ((si - nplain) & 1) != 0);
            else {
                refill_gate(slot, false);
                refill_gate(wrap(slot + 1), true);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 1; nb < NB; ++nb) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[nb], b_cur[pb], acc[nb][pb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a_cur[nb] = a_nxt[nb];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) b_cur[pb] = b_nxt[pb];
            slot = nslot;
        };
        using std::false_type;
        using std::true_type;
        using RF0 = std::integral_constant<int, 0>;
        using RF1 = std::integral_constant<int, 1>;
        using RF2 = std::integral_constant<int, 2>;
        if constexpr (GATED) {
            static_assert(D >= 6 && D % 2 == 0, "a gated k-pair and its successor hold four slots; static parity needs an even ring");
            int kp = kp_begin;
            for (; kp + D < kpe; ++kp) step(kp, false_type{}, false_type{}, RF0{});       // slots issued here are still plain
            for (; kp + 1 < kpe; ++kp) step(kp, false_type{}, false_type{}, RF1{});
            if (kp < kpe) {
                step(kp, false_type{}, true_type{}, RF1{});                               // (the candidate always has gated rows)
                ++kp;
            }
            for (; kp < KT; ++kp) step(kp, true_type{}, true_type{}, RF2{});
        } else {
            for (int kp = kp_begin; kp < KT; ++kp) step(kp, false_type{}, false_type{}, RF0{});
        }
        }
        wait_vmcnt<0>();
        TRACE_STAMP(2);
        if constexpr (SPLIT) {
            // Re-derive the pixel map for the epilogue instead of carrying it in registers across the k-loop (the bf16 pieces
            // need them: a spilled value reloaded between two stores would serialise the stores, see the bias note above).
            int tile_e = tile, j_e = j;
            asm volatile("" : "+s"(tile_e), "+v"(j_e));
            pm.init(tile_e, j_e, prm.P, prm.W, prm.P2, prm.W2);
        }

        if constexpr (EPI == EPI_LRELU) {
            // out[b][n][p] = lrelu(acc + bias)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cib = mfma_row(r, half);
                    const int n = n0 + nb * 32 + cib;
                    if (n < prm.Cout) {
                        const float bv = bias_h[nb * 32 + row_c(r)];
                        float v[PB];
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb) v[pb] = lrelu(fin(acc[nb][pb][r], bv), prm.slope);
                        store_row<MAP, PB>(prm.out0 + ((size_t)b * prm.Cout + n) * prm.P, pm, v);
                    }
                }
            if constexpr (NB == 1 && PB == 4 && MAP == MAP_VEC) {
                if (prm.stemW) {
                    // u0 = Ws . f for the tile's 128 pixels and its LayerNorm partials (sum, squares about the tile mean): rows 0..7 of a
                    // lane are channels {0-3, 8-11} + 4 half of ONE pixel per pixel block; the partner lane (lane ^ 32) holds the other
                    // eight.  Lane (j, half) computes outputs {0-3, 8-11} + 4 half of its four pixels.  Whole waves (shuffles).
                    // On the matrix pipe: accumulator row r of a pixel block IS the B operand of v_mfma_f32_32x32x2_f32 for the k-pair
                    // (channel row_c(r), channel row_c(r) + 4) -- lanes 0-31 hold the first, lanes 32-63 the second --, and the result's
                    // rows 0..7 of a lane are the outputs {0-3, 8-11} + 4 half of its pixel: eight MFMAs per pixel block, no shuffles.
                    const float *hw = ssm;
                    float a8[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) a8[r] = j < 16 ? hw[j * 16 + row_c(r) + 4 * half] : 0.f;     // A[m = j][k = half]: Ws[m][channel]
                    float u8[PB][8];
                    float s = 0.f;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        f32x16 ua;
#pragma unroll
                        for (int r = 0; r < 16; ++r) ua[r] = 0.f;
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float f = lrelu(fin(acc[0][pb][r], bias_h[row_c(r)]), prm.slope);
                            ua = __builtin_amdgcn_mfma_f32_32x32x2f32(a8[r], f, ua, 0, 0, 0);
                        }
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            u8[pb][r] = ua[r];
                            if (pm.valid[pb]) s += ua[r];
                        }
                    }
                    s = wave_sum(s);
                    const int nvalid = tile_valid(tile, 32 * PB, prm.P);
                    const float m = s / (16.f * (float)nvalid);
                    float q = 0.f;
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            const float d = u8[pb][r] - m;
                            if (pm.valid[pb]) q = fmaf(d, d, q);
                        }
                    q = wave_sum(q);
                    if (lane == 0) {
                        float *pp = prm.stemPart + ((size_t)b * prm.tilesPerSample + tile) * 2;
                        pp[0] = s;
                        pp[1] = q;
                    }
                }
            }
        } else if constexpr (EPI == EPI_POOL) {
            // out[b][n][q] = 0.25 * sum_{2x2} lrelu(acc + bias)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cib = mfma_row(r, half);
                    const int n = n0 + nb * 32 + cib;
                    if (n < prm.Cout && pm.valid[0]) {
                        const float bv = bias_h[nb * 32 + row_c(r)];
                        float s = 0.f;
#pragma unroll
                        for (int pb = 0; pb < 4; ++pb) s += lrelu(fin(acc[nb][pb][r], bv), prm.slope);
                        prm.out0[((size_t)b * prm.Cout + n) * prm.P2 + pm.q] = 0.25f * s;
                    }
                }
        } else if constexpr (EPI == EPI_DECONV) {
            // group = output row parity a; n-blocks = (bb, co-block); out[b][co][2y+a][2x+bb] = lrelu(acc + bias)
            static_assert(NB % 2 == 0, "deconv group holds both column parities");
            constexpr int NBC = NB / 2;
            const int a = g;
            const int W2 = 2 * prm.W;
            int oy[PB], ox[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int y = pm.off[pb] / prm.W;
                oy[pb] = 2 * y + a;
                ox[pb] = 2 * (pm.off[pb] - y * prm.W);
            }
#pragma unroll
            for (int cob = 0; cob < NBC; ++cob)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int cib = mfma_row(r, half);
                    const int co = cob * 32 + cib;
                    if (co < prm.Cout) {
                        const float bv = bias_h[cob * 32 + row_c(r)];
                        float *oplane = prm.out0 + ((size_t)b * prm.Cout + co) * (4 * (size_t)prm.P);
                        if constexpr (is_pair(MAP)) {
                            // two horizontally adjacent input pixels -> four consecutive output floats (W even, p even)
                            if (pm.valid[0]) {
                                f32x4 v;
                                v.x = lrelu(fin(acc[cob][0][r], bv), prm.slope);
                                v.y = lrelu(fin(acc[NBC + cob][0][r], bv), prm.slope);
                                v.z = lrelu(fin(acc[cob][1][r], bv), prm.slope);
                                v.w = lrelu(fin(acc[NBC + cob][1][r], bv), prm.slope);
                                *reinterpret_cast<f32x4 *>(oplane + (size_t)oy[0] * W2 + ox[0]) = v;   // (non-temporal here: -1 %)
                            }
                        } else {
#pragma unroll
                            for (int pb = 0; pb < PB; ++pb)
                                if (pm.valid[pb]) {
                                    f32x2 v;
                                    v.x = lrelu(fin(acc[cob][pb][r], bv), prm.slope);
                                    v.y = lrelu(fin(acc[NBC + cob][pb][r], bv), prm.slope);
                                    *reinterpret_cast<f32x2 *>(oplane + (size_t)oy[pb] * W2 + ox[pb]) = v;
                                }
                        }
                    }
                }
        } else if constexpr (EPI == EPI_GRU1) {
            // block nb of group g is canonical block cb of [z_0 .. z_{G-1} | r_0 .. r_{G-1}] (urnn_gate_cb; the fp32 / bf16 slabs:
            // group i = z_i | r_i): raw (pre-GroupNorm) gates -> out0 (B,2F,P) channels [32 cb, 32 cb + 32) and that block's
            // GroupNorm partial sums -> partial[b][cb][tile][2].
            const int F = prm.F;
            const int G = F / 32;
            const bool grouped = SPLIT == 3 && prm.biasf != nullptr;
            const float inv_n = tile == prm.tilesPerSample - 1 ? prm.invTail : prm.invFull;   // 1 / (32 * valid pixels), from the host: no v_rcp here
            // pass 1, registers only: every n-block's tile sum -> its own mean (the NB reductions run interleaved); pass 2: squares
            // about that mean (urnn_common.h tile_x2) while the rows are stored -- an accumulator row dies with its store
            float s1[NB], s2[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                s1[nb] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[nb * 32 + row_c(r)];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        if (pm.valid[pb]) s1[nb] += fin(acc[nb][pb][r], bv);
                }
            }
            wave_sum_n<NB>(s1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int cb = grouped ? urnn_gate_cb(prm.gHalves, prm.gGS, G, g, nb) : urnn_gate_cb(0, 1, G, g, nb);
                const float mt = nofma(s1[nb] * inv_n);     // (rounded on its own: v - mt must not become an fma in one kernel and not in another)
                s2[nb] = 0.f;
                const bool keep = !(prm.zOnly && cb >= G);     // fused-reset-gate cell: r leaves only its statistics
                float *obase = prm.out0 + ((size_t)b * 2 * F + cb * 32 + 4 * half) * prm.P;   // one lane-dependent base, uniform row steps
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[nb * 32 + row_c(r)];
                    float *orow = obase + (size_t)row_c(r) * prm.P;
                    float v[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        v[pb] = fin(acc[nb][pb][r], bv);
                        const float d = v[pb] - mt;
                        if (pm.valid[pb]) s2[nb] = fmaf(d, d, s2[nb]);
                    }
                    if (keep) store_row<MAP, PB>(orow, pm, v);
                }
            }
            wave_sum_n<NB>(s2);
            if (lane == 0) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const int cb = grouped ? urnn_gate_cb(prm.gHalves, prm.gGS, G, g, nb) : urnn_gate_cb(0, 1, G, g, nb);
                    float *pp = prm.partial + (((size_t)b * 2 * G + cb) * prm.tilesPerSample + tile) * 2;
                    pp[0] = s1[nb];
                    pp[1] = s2[nb];
                }
            }
        } else if constexpr (EPI == EPI_CAND) {
            // group g owns candidate channels [g*NB*32, (g+1)*NB*32): pre-GroupNorm candidate -> out0 (B,F,P), partial statistics
            // (sum, centred second moment) per 32-channel GroupNorm group -> partial[b][F/32][tile][2]
            const int F = prm.F;
            const float inv_n = tile == prm.tilesPerSample - 1 ? prm.invTail : prm.invFull;   // 1 / (32 * valid pixels), from the host: no v_rcp here
            float s1[NB], s2[NB];                                  // pass 1 (registers only) / pass 2 (+ stores) as in the gate epilogue
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                s1[nb] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[nb * 32 + row_c(r)];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        if (pm.valid[pb]) s1[nb] += fin(acc[nb][pb][r], bv);
                }
            }
            wave_sum_n<NB>(s1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int grp = g * NB + nb;
                const float mt = nofma(s1[nb] * inv_n);     // (rounded on its own: v - mt must not become an fma in one kernel and not in another)
                s2[nb] = 0.f;
                float *obase = prm.out0 + ((size_t)b * F + grp * 32 + 4 * half) * prm.P;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[nb * 32 + row_c(r)];
                    float *orow = obase + (size_t)row_c(r) * prm.P;
                    float v[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        v[pb] = fin(acc[nb][pb][r], bv);
                        const float d = v[pb] - mt;
                        if (pm.valid[pb]) s2[nb] = fmaf(d, d, s2[nb]);
                    }
                    store_row<MAP, PB>(orow, pm, v);
                }
            }
            wave_sum_n<NB>(s2);
            if (lane == 0) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    float *pp = prm.partial + (((size_t)b * (F / 32) + g * NB + nb) * prm.tilesPerSample + tile) * 2;
                    pp[0] = s1[nb];
                    pp[1] = s2[nb];
                }
            }
        }
#ifdef URNN_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TRACE_STAMP(3);
        ++tr_n;
#endif
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------------
static constexpr int RING_D = 8;
static constexpr int NUM_CUS = 256;                   // MI355X
static constexpr size_t LDS_PER_CU = 160 * 1024;

// Persistent grid: as many blocks as fit (LDS-limited, at most 2 per CU), a multiple of 8*NG, no more than the work needs.
static int tune_block_waves()
{
    static int v = -1;
    if (v < 0) {
        v = (int)urnn_tune("URNN_TUNE_WPB", 0);   // development knob: 4 forces 4-wave blocks
    }
    return v;
}

static int tune_stagger()
{
    static int v = -1;
    if (v < 0) {
        v = (int)urnn_tune("URNN_TUNE_STAGGER", -1);   // development knob: start offset of the second wave per SIMD in naps (-1: the launcher's rule)
    }
    return v;
}

static int persistent_grid(size_t lds_bytes, int NG, int total_tiles, int wpb = 4, int max_blocks_per_cu = 2)
{
    int bpc = (int)(LDS_PER_CU / lds_bytes);
    const int cap = (wpb >= 8 && max_blocks_per_cu <= 2) ? 1 : (max_blocks_per_cu > 2 ? 2 : max_blocks_per_cu);
    bpc = bpc < 1 ? 1 : (bpc > cap ? cap : bpc);
    const int unit = 8 * NG;
    static const int cus = [] { const int v = (int)urnn_tune("URNN_TUNE_CUS", 0); return v > 0 && v < NUM_CUS ? v : NUM_CUS; }();
    int n = (cus * bpc) / unit * unit;      // development knob URNN_TUNE_CUS: leave CUs free for the other kernel chain
    int need = ((total_tiles + wpb - 1) / wpb) * NG;
    need = (need + unit - 1) / unit * unit;
    n = n < need ? n : need;
    return n < unit ? unit : n;
}

template <typename K>
static hipError_t allow_big_lds(K kernel, size_t lds)
{
    if (lds <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

// dynamic LDS: weight slab + per-wave rings (D slots + the dummy sink) + bias row + (candidate GEMM) the r-gate scale/shift table
template <int NB, int PB, int EPI>
static int split_mode(const ConvGemmParams &p);
template <int NB, int PB, int EPI>
static bool split_ok(const ConvGemmParams &p) { return split_mode<NB, PB, EPI>(p) != 0; }

template <int NB, int PB, int MAP, int EPI>
static size_t conv_lds_bytes(const ConvGemmParams &p, int D, int WPB)
{
    using R = Ring<PB, MAP>;
    const int sm = split_mode<NB, PB, EPI>(p);
    const size_t slab = sm == 3 ? (size_t)p.fDwords * 4 : (sm ? (size_t)p.sDwords * 4 : (size_t)p.aFloats * 4);
    return slab + (size_t)WPB * ((D + 1) * R::SLOT) + NB * 128 + (EPI == EPI_CAND ? (size_t)p.B * p.F * 8 : 0) + (EPI == EPI_LRELU && p.stemW ? 1024 : 0);
}

template <int NB, int PB, int MAP, int EPI, int D, int WPB, int SPLIT>
static hipError_t launch_conv_split(const ConvGemmParams &p, hipStream_t st, int max_bpc)
{
    using R = Ring<PB, MAP>;
    const size_t lds = conv_lds_bytes<NB, PB, MAP, EPI>(p, D, WPB);
    if (lds > LDS_PER_CU) return hipErrorInvalidValue;
    auto kern = conv_gemm_kernel<NB, PB, MAP, EPI, D, WPB, SPLIT>;
    // raise this instantiation's dynamic-LDS cap once (and again only if a launch needs more); one process drives one GPU
    static std::atomic<size_t> allowed{64 * 1024};
    if (lds > allowed.load(std::memory_order_relaxed)) {
        hipError_t e = allow_big_lds(kern, LDS_PER_CU);
        if (e != hipSuccess) return e;
        allowed.store(LDS_PER_CU, std::memory_order_relaxed);
    }
    const int grid = persistent_grid(lds, p.NG, p.totalTiles, WPB, max_bpc);
    ConvGemmParams q = p;
    // second wave of each SIMD two naps behind the first where the waves have two tiles each (gate GEMMs of the full-resolution cells)
    q.stagger = tune_stagger() >= 0 ? tune_stagger() : ((EPI == EPI_GRU1 && WPB == 8 && 2L * p.totalTiles >= 3L * grid * WPB) ? 2 : 0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WPB), lds, st, q);
    return hipGetLastError();
}

static int tune_ring()
{
    static int v = -1;
    if (v < 0) {
        v = (int)urnn_tune("URNN_TUNE_RING", 8);
    }
    return v;
}

extern std::atomic<int> g_matrix_mode;   // urnn_gemm.hip (urnn_set_matrix_mode)

static int tune_split()
{
    static int v = -1;
    if (v < 0) {
        v = (int)urnn_tune("URNN_TUNE_SPLIT", 1);   // development knob: 0 forces the fp32-MFMA k-loop everywhere
    }
    return v;
}

#ifndef URNN_KEEP_BF16X6
#define URNN_KEEP_BF16X6 0   // 1 (A/B builds): bf16 x 6 instantiations for every epilogue; URNN_TUNE_F16=0 then selects them
#endif
static int tune_f16()
{
    static int v = -1;
    if (v < 0) {
        v = (int)urnn_tune("URNN_TUNE_F16", 1);     // development knob: 0 = bf16 x 6 instead of f16 x 3 (URNN_KEEP_BF16X6 builds)
    }
    return v;
}

// Which k-loop (the SPLIT template argument): 3 = f16 x 3 (forward activations), 1 = bf16 x 6 (gradients: prm.wide), 2 = bf16
// compute mode -- whenever the K range is made of whole 16-k groups (every layer of the published network) and the accumulators
// leave room for the pieces; else 0 = the exact fp32-MFMA loop (odd channel counts).
template <int NB, int PB, int EPI>
static int split_mode(const ConvGemmParams &p)
{
    // accumulators + pieces: <= 128 accumulators in a 256-register wave (8-wave blocks); the deconv's 6-block tile runs one
    // wave per SIMD (4-wave blocks, 512 registers) and takes its 192
    if constexpr (EPI == EPI_DECONV ? NB * PB * 16 > 192 : NB * PB * 16 > 128) return 0;
    if (!tune_split() || g_matrix_mode.load(std::memory_order_relaxed) == URNN_MATRIX_FP32_MFMA) return 0;
    if constexpr (EPI == EPI_CAND) {
        // URNN_MATRIX_FP32_CAND: the candidate GEMM of a full-resolution cell is the one launch whose 16-bit k-loop shows in a long
        // rollout's error (profiles/r03_noise_floor_cell_gemm_arithmetic.txt): exact fp32 MFMA for it, everything else as the default mode
        if (g_matrix_mode.load(std::memory_order_relaxed) == URNN_MATRIX_FP32_CAND && p.P >= URNN_FULL_RES_PIXELS) return 0;
        if (p.candExact) return 0;
    }
    if (p.KT % 8 != 0 || p.kpBegin % 8 != 0 || p.KT <= p.kpBegin) return 0;      // whole 16-k groups, aligned with the packed ones
    if constexpr (EPI == EPI_CAND) {
        const int kpe = p.hKp0 < p.KT ? p.hKp0 : p.KT;
        if (kpe <= p.kpBegin || kpe % 8 != 0 || kpe >= p.KT) return 0;
    }
    const bool bf_ok = p.sDwords > 0 && p.wsplit && (size_t)p.sDwords * 4 <= LDS_PER_CU - 24 * 1024;
    const bool f16_ok = p.fDwords > 0 && p.wf16 && (size_t)p.fDwords * 4 <= LDS_PER_CU - 24 * 1024;
    if (g_matrix_mode.load(std::memory_order_relaxed) == URNN_MATRIX_BF16) return bf_ok ? 2 : 0;
    const bool want_bf = p.wide || (URNN_KEEP_BF16X6 && !tune_f16());
    if (want_bf && (EPI == EPI_LRELU || URNN_KEEP_BF16X6)) return bf_ok ? 1 : 0;
    // the gate GEMM in its F/32 groups of (z_i | r_i) without an f16 slab in that grouping (strips: their statistics exchange is
    // written against this tile layout): bf16 x 6 instead of dropping to the fp32 matrix instruction
    if constexpr (EPI == EPI_GRU1 && NB == 2) {
        if (!f16_ok) return bf_ok ? 1 : 0;
    }
    return f16_ok ? 3 : 0;
}

template <int NB, int PB, int MAP, int EPI, int D, int WPB>
static hipError_t launch_conv_cfg(const ConvGemmParams &p, hipStream_t st, int max_bpc = 2)
{
    if constexpr (MAP == MAP_QUAD16) {                  // split k-loops only (quad_ok checked the mode), never the two-stream candidate
        if constexpr (EPI == EPI_CAND || (EPI == EPI_DECONV ? !(NB * PB * 16 <= 192 && WPB == 4) : NB * PB * 16 > 128)) return hipErrorInvalidValue;
        else {
            const int sm = split_mode<NB, PB, EPI>(p);
            if (sm == 3) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 3>(p, st, max_bpc);
            if constexpr (!(EPI == EPI_GRU1 && NB > 2)) {
                if (sm == 2) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 2>(p, st, max_bpc);
                if constexpr (EPI == EPI_LRELU || (EPI == EPI_GRU1 && NB == 2) || URNN_KEEP_BF16X6) {
                    if (sm == 1) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 1>(p, st, max_bpc);
                }
            }
            return hipErrorInvalidValue;
        }
    } else
    if constexpr (EPI == EPI_GRU1 && NB > 2) {          // grouped gate GEMM: exists in the f16 form only (gate_grouped checked the rest)
        static_assert(NB * PB * 16 <= 128, "grouped gate tiles are 64 pixels wide at most");
        if (split_mode<NB, PB, EPI>(p) != 3) return hipErrorInvalidValue;
        return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 3>(p, st, max_bpc);
    } else
    if constexpr (EPI == EPI_DECONV ? (NB * PB * 16 <= 192 && WPB == 4) : NB * PB * 16 <= 128) {
        const int sm = split_mode<NB, PB, EPI>(p);
        if (sm == 3) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 3>(p, st, max_bpc);
        if (sm == 2) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 2>(p, st, max_bpc);
        if constexpr (EPI == EPI_LRELU || (EPI == EPI_GRU1 && NB == 2) || URNN_KEEP_BF16X6) {
            if (sm == 1) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 1>(p, st, max_bpc);
        }
    }
    if constexpr (MAP != MAP_QUAD16) return launch_conv_split<NB, PB, MAP, EPI, D, WPB, 0>(p, st, max_bpc);
}

// Block shape: 8 waves (two per SIMD: one wave's epilogue / stalls hide behind the other's MFMAs) with a 4-deep ring when
// the weight slab leaves room, else 4 waves with an 8-deep ring.
template <int NB, int PB, int MAP, int EPI>
static hipError_t launch_conv(const ConvGemmParams &p, hipStream_t st)
{
    // 8-wave blocks only pay when there is enough work to fill 2048 wave slots; small planes keep 4-wave blocks
    const bool enough = (long)p.totalTiles * p.NG >= 1024;
    if constexpr (EPI == EPI_CAND) {
        // gated k-pairs hold two ring slots each: 8-deep rings
        if constexpr (NB * PB * 16 <= 64) {
            // small accumulators: two 8-wave blocks per CU (4 waves per SIMD) when the LDS allows (6-deep rings)
#ifdef URNN_TUNING
            if (2 * conv_lds_bytes<NB, PB, MAP, EPI>(p, 6, 8) <= LDS_PER_CU && enough && tune_block_waves() == 16)
                return launch_conv_cfg<NB, PB, MAP, EPI, 6, 8>(p, st, 3);
#endif
        }
        if constexpr (NB * PB * 16 <= 128) {   // + the gate/hidden fragments of the gated rows: 192 accumulators would spill
            if (conv_lds_bytes<NB, PB, MAP, EPI>(p, 8, 8) <= LDS_PER_CU && enough && tune_block_waves() != 4)
                return launch_conv_cfg<NB, PB, MAP, EPI, 8, 8>(p, st);
        }
        return launch_conv_cfg<NB, PB, MAP, EPI, 8, 4>(p, st);
    } else {
        if constexpr (NB * PB * 16 <= 192 && EPI != EPI_DECONV) {   // accumulators + loop state fit the 256-register budget (the deconv scatter does not)
            if constexpr (NB * PB * 16 <= 64) {
#ifdef URNN_TUNING
                if (2 * conv_lds_bytes<NB, PB, MAP, EPI>(p, 4, 8) <= LDS_PER_CU && enough && tune_block_waves() == 16)
                    return launch_conv_cfg<NB, PB, MAP, EPI, 4, 8>(p, st, 3);
#endif
            }
            if constexpr (NB * PB * 16 <= 128) {
                // bf16 x 6 k-loop: a k-pair is consumed in ~200 cycles instead of 512, so a 4-deep ring (3 KiB in flight per
                // wave) no longer covers the HBM latency -- the dec1 gate GEMM stayed at 113 us with the MFMA and issue time
                // halved; 8-deep where the LDS allows (development knob URNN_TUNE_RING = 4 | 6 | 8)
                if (split_ok<NB, PB, EPI>(p) && enough && tune_block_waves() != 4) {
                    const int want = tune_ring();
                    if (want >= 8 && conv_lds_bytes<NB, PB, MAP, EPI>(p, 8, 8) <= LDS_PER_CU) return launch_conv_cfg<NB, PB, MAP, EPI, 8, 8>(p, st);
#ifdef URNN_TUNING
                    if (want >= 6 && conv_lds_bytes<NB, PB, MAP, EPI>(p, 6, 8) <= LDS_PER_CU) return launch_conv_cfg<NB, PB, MAP, EPI, 6, 8>(p, st);
#endif
                }
            }
            if (conv_lds_bytes<NB, PB, MAP, EPI>(p, 4, 8) <= LDS_PER_CU && enough && tune_block_waves() != 4)
                return launch_conv_cfg<NB, PB, MAP, EPI, 4, 8>(p, st);
        }
        return launch_conv_cfg<NB, PB, MAP, EPI, 8, 4>(p, st);
    }
}

// MAP_PAIR16 -> MAP_QUAD16 (two k-pairs per DMA instruction, all 64 lanes): whenever a split k-loop will run and every K segment
// starts at an even k-pair, so that a 4-row slot never straddles two inputs.  Same pixels, same fragments, same MFMA order: the
// results are bit-identical to the half-wave form.  Development knob URNN_TUNE_QUAD=0 keeps the half-wave DMA.
template <int NB, int PB, int EPI>
static bool quad_ok(const ConvGemmParams &p)
{
    static const int on = (int)urnn_tune("URNN_TUNE_QUAD", 1);
    if (!on || p.P % 4 != 0) return false;
    if (p.segKp0[1] != INT_MAX && (p.segKp0[1] & 1)) return false;
    if (p.segKp0[2] != INT_MAX && (p.segKp0[2] & 1)) return false;
    const int sm = split_mode<NB, PB, EPI>(p);
    if (sm == 0) return false;
    if constexpr (EPI == EPI_GRU1 && NB > 2) return sm == 3;
    else if constexpr (EPI == EPI_LRELU || (EPI == EPI_GRU1 && NB == 2) || URNN_KEEP_BF16X6) return true;
    else return sm == 3 || sm == 2;
}

// tile shape -> (PB, MAP): 16-B DMA on aligned planes, pair/strided dword DMA otherwise
template <int NB, int EPI>
static hipError_t launch_flat(const ConvGemmParams &p, int pb, int map, hipStream_t st)
{
    if constexpr (!(EPI == EPI_GRU1 && NB > 2)) {      // the grouped gate kernels run 64- / 32-pixel tiles only
        if (map == MAP_VEC && pb == 4) return launch_conv<NB, 4, MAP_VEC, EPI>(p, st);
    }
    if constexpr (EPI != EPI_CAND) {
        if (map == MAP_PAIR16 && pb == 2 && quad_ok<NB, 2, EPI>(p)) return launch_conv<NB, 2, MAP_QUAD16, EPI>(p, st);
    }
    if (map == MAP_PAIR16 && pb == 2) return launch_conv<NB, 2, MAP_PAIR16, EPI>(p, st);
    if (map == MAP_PAIR && pb == 2) return launch_conv<NB, 2, MAP_PAIR, EPI>(p, st);
    if (map == MAP_STRIDED && pb == 2) return launch_conv<NB, 2, MAP_STRIDED, EPI>(p, st);
    if (map == MAP_STRIDED && pb == 1) return launch_conv<NB, 1, MAP_STRIDED, EPI>(p, st);
    return hipErrorInvalidValue;
}


// 1 / (values per GroupNorm tile) for the epilogues' tile means (full tiles / the last, possibly partial one)
static void set_tile_means(ConvGemmParams &p, int tile_pix)
{
    const int tail = p.P - (p.tilesPerSample - 1) * tile_pix;
    p.invFull = 1.0f / (32.0f * (float)tile_pix);
    p.invTail = 1.0f / (32.0f * (float)(tail > 0 ? tail : tile_pix));
}
