// urnn_gemm.hip -- the per-pixel contractions of U-RNN as fp32 MFMA GEMMs on gfx950.
//
// Every convolution of the network is 1x1, i.e. out[n][p] = sum_k W[n][k] * in[k][p] over the pixels p of an
// NCHW plane.  One wavefront owns a tile of 32*PB pixels x NB*32 output channels and runs v_mfma_f32_32x32x2_f32 with
//   A = packed weights  (lane l: n = n0 + nb*32 + (l & 31), k = 2*kp + (l >> 5))
//   B = activations     (lane l: p = pixel(l & 31, pb),     k = 2*kp + (l >> 5))
//   D[n][p] accumulates in registers (AGPRs); epilogues fuse bias, LeakyReLU, AvgPool, the ConvTranspose scatter, or the
//   GroupNorm partial statistics.  fp32-input MFMA is bit-exact fp32 FMA (gfx950 has no TF32 path) at the fp32 peak rate.
//
// Operand streaming: both operands are laid out so that EVERY LANE CONSUMES EXACTLY THE BYTES IT LOADS (activations are used
// by one wave tile only; the packed weights are pre-arranged per lane).  Each wave therefore owns a private LDS ring of D
// k-pair slots filled by asynchronous LDS-DMA (global_load_lds) D k-pairs ahead of use and drained with ds_read of the
// lane's own 16 B: a per-lane FIFO -- no bank conflicts, no barriers, no VGPRs spent on prefetch, counted s_waitcnt vmcnt.
// The first version of this kernel prefetched one chunk into registers and measured 25-45 % of the MFMA peak because HBM
// latency (>1 us under load) exceeded the prefetch distance at 1 wave/SIMD (profiles/r01_kernel_bench_v1.txt).
//
// Kernels here: conv_gemm_kernel (stage convs, deconvs, GRU gate GEMM) and gru_cand_kernel (candidate GEMM whose B operand
// is sigmoid(GN(r)) * h computed on the fly from two DMA streams).
#include "urnn_common.h"
#include "urnn_kernels.h"

#ifndef URNN_ABL
#define URNN_ABL 0   // tuning builds only: 1 skip weight DMA, 2 skip activation DMA, 4 skip epilogue stores
#endif
typedef __attribute__((address_space(3))) void *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

extern __shared__ __attribute__((aligned(16))) char urnn_smem[];

__device__ __forceinline__ void dma16(const float *g, char *l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0); }
__device__ __forceinline__ void dma4(const float *g, char *l) { __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 4, 0, 0); }

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
//   MAP_STRIDED  p = t*32*PB + 32*pb + j                    dword everything (any P)
//   MAP_POOL     pooled pixel q = t*32 + j, input pixel (2*y2 + pb/2, 2*x2 + pb%2)   (PB == 4)
// ------------------------------------------------------------------------------------------------------------------
template <int MAP, int PB>
struct PixelMap {
    int off[PB];     // clamped (always in-bounds) offsets inside an input plane
    bool valid[PB];  // false: out of range, contributes nothing and is never stored
    int q;           // MAP_POOL: pooled output pixel index

    __device__ __forceinline__ void init(int tile, int j, int P, int W, int P2, int W2)
    {
        q = 0;
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
                else if constexpr (MAP == MAP_PAIR) p = tile * 64 + 2 * j + pb;
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
    } else if constexpr (MAP == MAP_PAIR) {
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
            if (pm.valid[4 * qd]) *reinterpret_cast<f32x4 *>(row + pm.off[4 * qd]) = f32x4{v[4 * qd], v[4 * qd + 1], v[4 * qd + 2], v[4 * qd + 3]};
    } else if constexpr (MAP == MAP_PAIR) {
        if (pm.valid[0]) *reinterpret_cast<f32x2 *>(row + pm.off[0]) = f32x2{v[0], v[1]};
    } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
            if (pm.valid[pb]) row[pm.off[pb]] = v[pb];
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Per-wave LDS ring.  One slot = one k-pair: NQ KiB of weights (4 floats per lane per quad of n-blocks) followed by
// PB*256 B of activations.  Lane l's own bytes sit at  quad*1024 + l*16  (16-B DMA) or  pb*256 + l*4  (dword DMA).
// ------------------------------------------------------------------------------------------------------------------
template <int NB, int PB, int MAP>
struct Ring {
    static constexpr int NQ = (NB + 3) / 4;
    static constexpr bool VEC = (MAP == MAP_VEC);
    static constexpr int NBLOAD = VEC ? PB / 4 : PB;
#if (URNN_ABL & 3) == 3
    static constexpr int NLOAD = 0;
#elif (URNN_ABL & 1)
    static constexpr int NLOAD = NBLOAD;
#elif (URNN_ABL & 2)
    static constexpr int NLOAD = NQ;
#else
    static constexpr int NLOAD = NQ + NBLOAD;               // DMA instructions per slot
#endif
    static constexpr int SLOT = NQ * 1024 + PB * 256;       // bytes

    // issue the DMA loads of one k-pair: weights (this lane's 4 floats per quad) and the activation row
    __device__ static __forceinline__ void issue(char *slot, const float *wq_lane, const float *row, const PixelMap<MAP, PB> &pm)
    {
#if !(URNN_ABL & 1)
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) dma16(wq_lane + qd * 256, slot + qd * 1024);
#endif
        char *bslot = slot + NQ * 1024;
#if (URNN_ABL & 2)
        return;
#endif
        if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) dma16(row + pm.off[4 * qd], bslot + qd * 1024);
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) dma4(row + pm.off[pb], bslot + pb * 256);
        }
    }

    __device__ static __forceinline__ void read(const char *slot, int lane, float (&a)[NQ * 4], float (&b)[PB])
    {
#pragma unroll
        for (int qd = 0; qd < NQ; ++qd) {
            const f32x4 t = *reinterpret_cast<const f32x4 *>(slot + qd * 1024 + lane * 16);
            a[4 * qd] = t.x; a[4 * qd + 1] = t.y; a[4 * qd + 2] = t.z; a[4 * qd + 3] = t.w;
        }
        const char *bslot = slot + NQ * 1024;
        if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) {
                const f32x4 t = *reinterpret_cast<const f32x4 *>(bslot + qd * 1024 + lane * 16);
                b[4 * qd] = t.x; b[4 * qd + 1] = t.y; b[4 * qd + 2] = t.z; b[4 * qd + 3] = t.w;
            }
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) b[pb] = *reinterpret_cast<const float *>(bslot + pb * 256 + lane * 4);
        }
    }
};

// ------------------------------------------------------------------------------------------------------------------
// conv_gemm_kernel: blockDim = 64 * NW; wave w owns output n-blocks [w*NB, (w+1)*NB) for one pixel tile.
// grid.x = B * tilesPerSample.  Dynamic LDS = NW * D * SLOT.
// ------------------------------------------------------------------------------------------------------------------
#ifndef URNN_GEMM_MINWAVES
#define URNN_GEMM_MINWAVES 1
#endif

template <int NB, int PB, int MAP, int EPI, int D>
__global__ __launch_bounds__(256, URNN_GEMM_MINWAVES) void conv_gemm_kernel(const ConvGemmParams prm)
{
    using R = Ring<NB, PB, MAP>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / prm.tilesPerSample;
    const int tile = blockIdx.x - b * prm.tilesPerSample;

    PixelMap<MAP, PB> pm;
    pm.init(tile, j, prm.P, prm.W, prm.P2, prm.W2);

    f32x16 acc[NB][PB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

    char *ring = urnn_smem + wave * (D * R::SLOT);
    // this lane's weights: Wq[wave][kp][quad][lane][4]
    const float *wq = prm.wt + ((size_t)wave * prm.KT * R::NQ) * 256 + lane * 4;
    const int kp_begin = prm.kpBegin, KT = prm.KT;

    // activation row of absolute k-pair kp (segments are concatenated, each padded to an even channel count)
    auto row_of = [&](int kp) -> const float * {
        int s = 0;
        if (kp >= prm.segKp0[1]) s = 1;
        if (kp >= prm.segKp0[2]) s = 2;
        const int C = prm.segC[s];
        int c = 2 * (kp - prm.segKp0[s]) + half;
        c = c < C ? c : C - 1;                       // pad row: weight is zero, any finite activation does
        return prm.seg[s] + ((size_t)b * C + c) * prm.P;
    };
    auto issue = [&](int kp, int slot) { R::issue(ring + slot * R::SLOT, wq + (size_t)kp * (R::NQ * 256), row_of(kp), pm); };
    auto consume = [&](int kp, int slot) {
        float a[R::NQ * 4], bv[PB];
        R::read(ring + slot * R::SLOT, lane, a, bv);
        // the hidden-state rows feed the z / r gates only; the candidate's h part waits for r (gru_cand_kernel)
        const bool hrow = (EPI == EPI_GRU1) && kp >= prm.hKp0;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (EPI == EPI_GRU1 && nb == 2 && hrow) continue;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb], bv[pb], acc[nb][pb], 0, 0, 0);
        }
    };

    const int nk = KT - kp_begin;
    {
        const int npro = nk < D ? nk : D;
        for (int i = 0; i < npro; ++i) issue(kp_begin + i, i);
    }
    int slot = 0;
    int kp = kp_begin;
    // steady state: D slots in flight; the oldest has landed once at most (D-1) slots' loads are outstanding
    for (; kp + D <= KT; ++kp) {
        wait_vmcnt<(D - 1) * R::NLOAD>();
        consume(kp, slot);
        asm volatile("" ::: "memory");
        if (kp + D < KT) issue(kp + D, slot);
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    // drain: everything left is already in flight
    wait_vmcnt<0>();
    for (; kp < KT; ++kp) {
        consume(kp, slot);
        slot = slot + 1 == D ? 0 : slot + 1;
    }

    const int n0 = wave * (NB * 32);
    const float *bias = prm.bias + n0;

    if constexpr (EPI == EPI_LRELU) {
        // out[b][n][p] = lrelu(acc + bias)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cib = mfma_row(r, half);
                const int n = n0 + nb * 32 + cib;
                if (n < prm.Cout) {
                    const float bv = bias[nb * 32 + cib];
                    float v[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) v[pb] = lrelu(acc[nb][pb][r] + bv, prm.slope);
                    store_row<MAP, PB>(prm.out0 + ((size_t)b * prm.Cout + n) * prm.P, pm, v);
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
                    const float bv = bias[nb * 32 + cib];
                    float s = 0.f;
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) s += lrelu(acc[nb][pb][r] + bv, prm.slope);
                    prm.out0[((size_t)b * prm.Cout + n) * prm.P2 + pm.q] = 0.25f * s;
                }
            }
    } else if constexpr (EPI == EPI_DECONV) {
        // wave = output row parity a; n-blocks = (bb, co-block); out[b][co][2y+a][2x+bb] = lrelu(acc + bias)
        static_assert(NB % 2 == 0, "deconv wave holds both column parities");
        constexpr int NBC = NB / 2;
        const int a = wave;
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
                    const float bv = bias[cob * 32 + cib];
                    float *oplane = prm.out0 + ((size_t)b * prm.Cout + co) * (4 * (size_t)prm.P);
                    if constexpr (MAP == MAP_PAIR) {
                        // two horizontally adjacent input pixels -> four consecutive output floats (W even, p even)
                        if (pm.valid[0]) {
                            f32x4 v;
                            v.x = lrelu(acc[cob][0][r] + bv, prm.slope);
                            v.y = lrelu(acc[NBC + cob][0][r] + bv, prm.slope);
                            v.z = lrelu(acc[cob][1][r] + bv, prm.slope);
                            v.w = lrelu(acc[NBC + cob][1][r] + bv, prm.slope);
                            *reinterpret_cast<f32x4 *>(oplane + (size_t)oy[0] * W2 + ox[0]) = v;
                        }
                    } else {
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb)
                            if (pm.valid[pb]) {
                                f32x2 v;
                                v.x = lrelu(acc[cob][pb][r] + bv, prm.slope);
                                v.y = lrelu(acc[NBC + cob][pb][r] + bv, prm.slope);
                                *reinterpret_cast<f32x2 *>(oplane + (size_t)oy[pb] * W2 + ox[pb]) = v;
                            }
                    }
                }
            }
    } else if constexpr (EPI == EPI_GRU1) {
        // wave i owns [z_i | r_i | c_i]: raw (pre-GroupNorm) gates -> out0 (B,2F,P), candidate x/e part + b2 -> out1 (B,F,P),
        // and the GroupNorm partial sums of z_i (group i) and r_i (group F/32 + i) -> partial[b][group][tile][2].
        static_assert(NB == 3, "gate tile is z|r|c");
        const int F = prm.F;
        const int i = wave;
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cib = mfma_row(r, half);
                const float bv = bias[nb * 32 + cib];
                float *orow;
                if (nb == 0) orow = prm.out0 + ((size_t)b * 2 * F + i * 32 + cib) * prm.P;
                else if (nb == 1) orow = prm.out0 + ((size_t)b * 2 * F + F + i * 32 + cib) * prm.P;
                else orow = prm.out1 + ((size_t)b * F + i * 32 + cib) * prm.P;
                float v[PB];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) {
                    v[pb] = acc[nb][pb][r] + bv;
                    if (nb < 2 && pm.valid[pb]) {
                        s1 += v[pb];
                        s2 += v[pb] * v[pb];
                    }
                }
                store_row<MAP, PB>(orow, pm, v);
            }
            if (nb < 2) {
                s1 = wave_sum(s1);
                s2 = wave_sum(s2);
                if (lane == 0) {
                    const int G = 2 * F / 32;
                    const int g = nb * (F / 32) + i;
                    float *pp = prm.partial + (((size_t)b * G + g) * prm.tilesPerSample + tile) * 2;
                    pp[0] = s1;
                    pp[1] = s2;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gru_cand_kernel: C = Cx + W2h . (r * h),  r = sigmoid(g_r * scale + shift)  (GroupNorm folded into scale/shift).
// One wave = 32*PB pixels x all F candidate channels (NBF = F/32 n-blocks); Cx is added in the epilogue and the sum goes
// back in place; GroupNorm partial sums of C per 32-channel group.  blockDim = 64 * WPB waves, one tile per wave.
// Ring slot = [weights 1 KiB | raw r gate PB*256 B | h PB*256 B]; dynamic LDS = 8*F (scale/shift) + WPB * D * SLOT.
// ------------------------------------------------------------------------------------------------------------------
template <int NBF, int PB, int MAP, int D>
__global__ __launch_bounds__(256) void gru_cand_kernel(const GruCandParams prm)
{
    constexpr int F = NBF * 32;
    constexpr bool VEC = (MAP == MAP_VEC);
    constexpr int NBLOAD = VEC ? PB / 4 : PB;
    constexpr int NLOAD = 1 + 2 * NBLOAD;
    constexpr int SLOT = 1024 + 2 * PB * 256;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / prm.blocksPerSample;
    const int tile = (blockIdx.x - b * prm.blocksPerSample) * wpb + wave;

    float *ssm = reinterpret_cast<float *>(urnn_smem);               // [F][2] scale/shift of the r gate for this sample
    char *ring = urnn_smem + 8 * F + wave * (D * SLOT);
    for (int c = threadIdx.x; c < 2 * F; c += blockDim.x) ssm[c] = prm.ss1[((size_t)b * 2 * F + F) * 2 + c];
    __syncthreads();
    if (tile >= prm.tilesPerSample) return;

    PixelMap<MAP, PB> pm;
    pm.init(tile, j, prm.P, 0, 0, 0);

    f32x16 acc[NBF][PB];
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

    const float *gr = prm.g1 + ((size_t)b * 2 * F + F + half) * prm.P;   // raw r-gate planes (row k = 2*kp + half)
    const float *hh = prm.h + ((size_t)b * F + half) * prm.P;
    const float *wq = prm.w2h + lane * 4;
    constexpr int KT = F / 2;

    auto issue = [&](int kp, int slot) {
        char *s = ring + slot * SLOT;
        dma16(wq + (size_t)kp * 256, s);
        const float *grow = gr + (size_t)(2 * kp) * prm.P;
        const float *hrow = hh + (size_t)(2 * kp) * prm.P;
        if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) {
                dma16(grow + pm.off[4 * qd], s + 1024 + qd * 1024);
                dma16(hrow + pm.off[4 * qd], s + 1024 + PB * 256 + qd * 1024);
            }
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                dma4(grow + pm.off[pb], s + 1024 + pb * 256);
                dma4(hrow + pm.off[pb], s + 1024 + PB * 256 + pb * 256);
            }
        }
    };
    auto consume = [&](int kp, int slot) {
        const char *s = ring + slot * SLOT;
        const f32x4 av = *reinterpret_cast<const f32x4 *>(s + lane * 16);
        const float a[4] = {av.x, av.y, av.z, av.w};
        float g[PB], h[PB];
        if constexpr (VEC) {
#pragma unroll
            for (int qd = 0; qd < PB / 4; ++qd) {
                const f32x4 tg = *reinterpret_cast<const f32x4 *>(s + 1024 + qd * 1024 + lane * 16);
                const f32x4 th = *reinterpret_cast<const f32x4 *>(s + 1024 + PB * 256 + qd * 1024 + lane * 16);
                g[4 * qd] = tg.x; g[4 * qd + 1] = tg.y; g[4 * qd + 2] = tg.z; g[4 * qd + 3] = tg.w;
                h[4 * qd] = th.x; h[4 * qd + 1] = th.y; h[4 * qd + 2] = th.z; h[4 * qd + 3] = th.w;
            }
        } else {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                g[pb] = *reinterpret_cast<const float *>(s + 1024 + pb * 256 + lane * 4);
                h[pb] = *reinterpret_cast<const float *>(s + 1024 + PB * 256 + pb * 256 + lane * 4);
            }
        }
        const f32x2 st = *reinterpret_cast<const f32x2 *>(ssm + 2 * (2 * kp + half));
        float bop[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) bop[pb] = sigmoidf_fast(g[pb] * st.x + st.y) * h[pb];
#pragma unroll
        for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[nb], bop[pb], acc[nb][pb], 0, 0, 0);
    };

    static_assert(KT >= D, "ring deeper than the K loop");
    for (int i = 0; i < D; ++i) issue(i, i);
    int slot = 0, kp = 0;
    for (; kp + D <= KT; ++kp) {
        wait_vmcnt<(D - 1) * NLOAD>();
        consume(kp, slot);
        asm volatile("" ::: "memory");
        if (kp + D < KT) issue(kp + D, slot);
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    wait_vmcnt<0>();
    for (; kp < KT; ++kp) {
        consume(kp, slot);
        slot = slot + 1 == D ? 0 : slot + 1;
    }

    // epilogue: add the x/e part (+ bias) written by the gate GEMM, store C in place, partial sums per 32-channel group
    float *cx = prm.cx + (size_t)b * F * prm.P;
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float *orow = cx + (size_t)(nb * 32 + mfma_row(r, half)) * prm.P;
            float v[PB];
            load_row<MAP, PB>(orow, pm, v);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                v[pb] += acc[nb][pb][r];
                if (pm.valid[pb]) {
                    s1 += v[pb];
                    s2 += v[pb] * v[pb];
                }
            }
            store_row<MAP, PB>(orow, pm, v);
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            float *pp = prm.partial + (((size_t)b * NBF + nb) * prm.tilesPerSample + tile) * 2;
            pp[0] = s1;
            pp[1] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------------
static constexpr int RING_D = 8;

template <int NB, int PB, int MAP, int EPI>
static hipError_t launch_conv(const ConvGemmParams &p, int nblocks, int nwaves, hipStream_t st)
{
    using R = Ring<NB, PB, MAP>;
    const size_t lds = (size_t)nwaves * RING_D * R::SLOT;
    hipLaunchKernelGGL((conv_gemm_kernel<NB, PB, MAP, EPI, RING_D>), dim3(nblocks), dim3(64 * nwaves), lds, st, p);
    return hipGetLastError();
}

// tile shape -> (PB, MAP): 16-B DMA on aligned planes, pair/strided dword DMA otherwise
template <int NB, int EPI>
static hipError_t launch_flat(const ConvGemmParams &p, int pb, int map, int nblocks, int nwaves, hipStream_t st)
{
    if (map == MAP_VEC && pb == 4) return launch_conv<NB, 4, MAP_VEC, EPI>(p, nblocks, nwaves, st);
    if (map == MAP_PAIR && pb == 2) return launch_conv<NB, 2, MAP_PAIR, EPI>(p, nblocks, nwaves, st);
    if (map == MAP_STRIDED && pb == 2) return launch_conv<NB, 2, MAP_STRIDED, EPI>(p, nblocks, nwaves, st);
    if (map == MAP_STRIDED && pb == 1) return launch_conv<NB, 1, MAP_STRIDED, EPI>(p, nblocks, nwaves, st);
    return hipErrorInvalidValue;
}

int urnn_conv_nb(int Cout)
{
    const int nblk = (Cout + 31) / 32;
    return nblk <= 3 ? nblk : (nblk % 3 == 0 ? 3 : (nblk % 2 == 0 ? 2 : 1));
}

// Flat 1x1 conv + LeakyReLU.  NB n-blocks per wave chosen from Cout; NW waves cover all columns.
hipError_t urnn_launch_conv_flat(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int nblk = (p.Cout + 31) / 32;
    const int NB = urnn_conv_nb(p.Cout);
    const int NW = nblk / NB;
    if (NW > 4) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    if (NB == 1) return launch_flat<1, EPI_LRELU>(p, PB, map, nblocks, NW, st);
    if (NB == 2) return launch_flat<2, EPI_LRELU>(p, PB, map, nblocks, NW, st);
    return launch_flat<3, EPI_LRELU>(p, PB, map, nblocks, NW, st);
}

hipError_t urnn_launch_conv_pool(ConvGemmParams p, int B, hipStream_t st)
{
    const int nblk = (p.Cout + 31) / 32;
    const int NB = urnn_conv_nb(p.Cout);
    const int NW = nblk / NB;
    if (NW > 4) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P2 + 31) / 32;
    const int nblocks = B * p.tilesPerSample;
    if (NB == 1) return launch_conv<1, 4, MAP_POOL, EPI_POOL>(p, nblocks, NW, st);
    if (NB == 2) return launch_conv<2, 4, MAP_POOL, EPI_POOL>(p, nblocks, NW, st);
    return launch_conv<3, 4, MAP_POOL, EPI_POOL>(p, nblocks, NW, st);
}

// Deconv: two waves (output row parity), each 2 * ceil(Cout/32) n-blocks, PB = 2 pairs (1 strided for tiny/odd planes).
hipError_t urnn_launch_deconv(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int nbc = (p.Cout + 31) / 32;
    if (nbc < 1 || nbc > 3) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    if (PB == 2 && map == MAP_PAIR) {
        if (nbc == 1) return launch_conv<2, 2, MAP_PAIR, EPI_DECONV>(p, nblocks, 2, st);
        if (nbc == 2) return launch_conv<4, 2, MAP_PAIR, EPI_DECONV>(p, nblocks, 2, st);
        return launch_conv<6, 2, MAP_PAIR, EPI_DECONV>(p, nblocks, 2, st);
    }
    if (PB != 1 || map != MAP_STRIDED) return hipErrorInvalidValue;
    if (nbc == 1) return launch_conv<2, 1, MAP_STRIDED, EPI_DECONV>(p, nblocks, 2, st);
    if (nbc == 2) return launch_conv<4, 1, MAP_STRIDED, EPI_DECONV>(p, nblocks, 2, st);
    return launch_conv<6, 1, MAP_STRIDED, EPI_DECONV>(p, nblocks, 2, st);
}

// GRU gate GEMM: F/32 waves of [z|r|c].
hipError_t urnn_launch_gru1(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int NW = p.F / 32;
    if (NW < 1 || NW > 4) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    return launch_flat<3, EPI_GRU1>(p, PB, map, nblocks, NW, st);
}

template <int NBF, int PB, int MAP>
static hipError_t launch_cand_one(const GruCandParams &p, int nblocks, int wpb, hipStream_t st)
{
    constexpr int SLOT = 1024 + 2 * PB * 256;
    const size_t lds = (size_t)8 * NBF * 32 + (size_t)wpb * RING_D * SLOT;
    hipLaunchKernelGGL((gru_cand_kernel<NBF, PB, MAP, RING_D>), dim3(nblocks), dim3(64 * wpb), lds, st, p);
    return hipGetLastError();
}

template <int NBF>
static hipError_t launch_cand_nbf(const GruCandParams &p, int PB, int map, int nblocks, int wpb, hipStream_t st)
{
    if (map == MAP_VEC && PB == 4) return launch_cand_one<NBF, 4, MAP_VEC>(p, nblocks, wpb, st);
    if (map == MAP_PAIR && PB == 2) return launch_cand_one<NBF, 2, MAP_PAIR>(p, nblocks, wpb, st);
    if (map == MAP_STRIDED && PB == 2) return launch_cand_one<NBF, 2, MAP_STRIDED>(p, nblocks, wpb, st);
    if (map == MAP_STRIDED && PB == 1) return launch_cand_one<NBF, 1, MAP_STRIDED>(p, nblocks, wpb, st);
    return hipErrorInvalidValue;
}

hipError_t urnn_launch_cand(GruCandParams p, int B, int F, int PB, int map, hipStream_t st)
{
    const int wpb = 2;   // 2 waves x 8 slots x <= 3 KiB stays under the 64 KiB dynamic-LDS default
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.blocksPerSample = (p.tilesPerSample + wpb - 1) / wpb;
    const int nblocks = B * p.blocksPerSample;
    switch (F / 32) {
    case 1: return launch_cand_nbf<1>(p, PB, map, nblocks, wpb, st);
    case 2: return launch_cand_nbf<2>(p, PB, map, nblocks, wpb, st);
    case 3: return launch_cand_nbf<3>(p, PB, map, nblocks, wpb, st);
    case 4: return launch_cand_nbf<4>(p, PB, map, nblocks, wpb, st);
    default: return hipErrorInvalidValue;
    }
}
