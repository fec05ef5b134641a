// urnn_elem.hip -- HBM-bound streaming kernels of the U-RNN timestep: GroupNorm/LayerNorm statistics finalisation,
// the GRU blend, the LayerNorm head, per-frame input assembly and the one-off weight packers.
#include "urnn_common.h"
#include "urnn_kernels.h"

#include <stdlib.h>

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm finalise: partial (sum, centred second moment) per tile (urnn_common.h tile_x2) -> per-channel (scale, shift) with
//   y = (v - mean) * rstd * gamma + beta = v * scale + shift.   One wavefront per (sample, 32-channel group).
// Partials are fp32 sums over <= 4096 values; they are combined in double in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------------------------------
// One wave per (sample, group): 32 independent partial loads in flight per lane, lane-strided double sums, xor butterfly
// (the same fixed order as the fold in the candidate GEMM's prologue).
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float *__restrict__ partial, int ntiles, int tile_pix, int P, double count,
                                                         const float *__restrict__ gamma, const float *__restrict__ beta,
                                                         float eps, float *__restrict__ ss, float *__restrict__ stat, int C, int *status, int status_bit)
{
    const int G = C / 32;
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int lane = threadIdx.x;
    const float *pp = partial + ((size_t)b * G + g) * ntiles * 2;
    double s1, s2;
    fold_lane_chain<32>(pp, ntiles, tile_pix, 32, P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane < 32) {
        const double mean = s1 / count;
        double var = s2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
        var = var > 0.0 ? var : 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        const int c = g * 32 + lane;
        const double sc = (double)gamma[c] * rstd;
        ss[((size_t)b * C + c) * 2] = (float)sc;
        ss[((size_t)b * C + c) * 2 + 1] = (float)((double)beta[c] - nofma(mean * sc));
        if (lane == 0) flag_nonfinite(status, status_bit, s1, s2);
        if (stat && lane == 0) {
            stat[((size_t)b * G + g) * 2] = (float)mean;
            stat[((size_t)b * G + g) * 2 + 1] = (float)rstd;
        }
    }
}

hipError_t urnn_launch_gn_finalize(const float *partial, int ntiles, int tile_pix, int P, double count, const float *gamma, const float *beta,
                                   float eps, float *ss, float *stat, int B, int C, int *status, int status_bit, hipStream_t st)
{
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * (C / 32)), dim3(64), 0, st, partial, ntiles, tile_pix, P, count, gamma, beta, eps, ss,
                       stat, C, status, status_bit);
    return hipGetLastError();
}

// max |v[i]| (the weight-range guard of include/urnn_hip.h): one block, atomicMax on the bits of a non-negative float
__global__ __launch_bounds__(1024) void max_abs_kernel(const float *__restrict__ v, long n, float *__restrict__ out)
{
    float m = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float a = fabsf(v[i]);
        m = (a > m || a != a) ? a : m;              // NaN wins: a NaN weight must not pass as "in range"
    }
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) {
        const float o = __shfl_xor(m, k, 64);
        m = (o > m || o != o) ? o : m;
    }
    __shared__ float sh[16];
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = (sh[w] > m || sh[w] != sh[w]) ? sh[w] : m;
        *out = m;
    }
}

hipError_t urnn_launch_max_abs(const float *v, long n, float *out, hipStream_t st)
{
    hipLaunchKernelGGL(max_abs_kernel, dim3(1), dim3(1024), 0, st, v, n, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// GRU blend: z = sigmoid(GN(g_z)); n = tanh(GN(c)); h' = (1 - z) * h + z * n          (ConvRNN.py:183-189)
// grid = (chunks, B*F): one (sample, channel) plane per blockIdx.y so scale/shift are block-uniform.
// ------------------------------------------------------------------------------------------------------------------
// FIN: the candidate's GroupNorm finalize runs in the block's prologue instead of a launch of its own (six launches fewer per
// frame): every block folds the per-tile partials of ITS channel's group in double, in a fixed order (thread-strided, xor
// butterfly, waves 0..3) -- all blocks of a group compute identical bits; the blocks of chunk 0 publish (scale, shift) and
// (mean, rstd) for the backward pass.
struct BlendFin {
    const float *partial;   // [B][F/32][ntiles][2] centred tile partials of the candidate (urnn_common.h tile_x2)
    int ntiles, tile_pix;
    double count;
    const float *gamma, *beta;
    float eps;
    float *ss2, *stat2;     // out: [B][F][2] (scale, shift), [B][F/32][2] (mean, rstd)
    int *status;            // the workspace's status word (urnn_common.h flag_nonfinite)
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 16-byte access at 4-byte alignment (planes of odd size)

// V = 4: 16-byte accesses (P % 4 == 0); V = 5: the same on planes whose size is not a multiple of four (the quarter-resolution
// 125 x 125 planes: every channel plane starts at a different 4-byte phase) -- unaligned 16-byte accesses for the first P & ~3
// pixels, dwords for the last P & 3; V = 1: dwords.
template <int V, bool FIN, int ITER_ = 0>
__global__ __launch_bounds__(256) void gru_blend_kernel(const float *__restrict__ g1, const float *__restrict__ c,
                                                        const float *h, const float *__restrict__ ss1,
                                                        const float *__restrict__ ss2, float *out, int F, int P, const BlendFin fin)
{
    const int bc = blockIdx.y;
    const int b = bc / F, f = bc - b * F;
    const float s1 = ss1[((size_t)b * 2 * F + f) * 2], t1 = ss1[((size_t)b * 2 * F + f) * 2 + 1];
    float s2, t2;
    if constexpr (FIN) {
        __shared__ double red[2][4];
        __shared__ float st[2];
        const int G = F / 32, g = f >> 5;
        const float *pp = fin.partial + ((size_t)b * G + g) * fin.ntiles * 2;
        double a1, a2;
        fold_thread_chain<256>(pp, fin.ntiles, fin.tile_pix, 32, P, threadIdx.x, a1, a2);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            a1 += __shfl_xor(a1, m, 64);
            a2 += __shfl_xor(a2, m, 64);
        }
        if ((threadIdx.x & 63) == 0) {
            red[0][threadIdx.x >> 6] = a1;
            red[1][threadIdx.x >> 6] = a2;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double S1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]), S2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
            const double mean = S1 / fin.count;
            double var = S2 / fin.count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
            var = var > 0.0 ? var : 0.0;
            const double rstd = 1.0 / sqrt(var + (double)fin.eps);
            const double sc = (double)fin.gamma[f] * rstd;
            st[0] = (float)sc;
            st[1] = (float)((double)fin.beta[f] - nofma(mean * sc));
            if (blockIdx.x == 0) {
                flag_nonfinite(fin.status, URNN_STATUS_CAND, S1, S2);
                fin.ss2[((size_t)b * F + f) * 2] = st[0];
                fin.ss2[((size_t)b * F + f) * 2 + 1] = st[1];
                if ((f & 31) == 0 && fin.stat2) {
                    fin.stat2[((size_t)b * G + g) * 2] = (float)mean;
                    fin.stat2[((size_t)b * G + g) * 2 + 1] = (float)rstd;
                }
            }
        }
        __syncthreads();
        s2 = st[0];
        t2 = st[1];
    } else {
        s2 = ss2[((size_t)b * F + f) * 2];
        t2 = ss2[((size_t)b * F + f) * 2 + 1];
    }
    const float *gz = g1 + ((size_t)b * 2 * F + f) * P;
    const float *cc = c + ((size_t)bc) * P;
    const float *hh = h + ((size_t)bc) * P;
    float *oo = out + ((size_t)bc) * P;
    // fused finalize: fewer, larger blocks (the prologue is per block) -- on large planes; a small plane (quarter resolution, the
    // small-grid configs) would leave most CUs idle while ~200 blocks walk through eight dependent memory round trips each
    constexpr int ITER = ITER_ ? ITER_ : (FIN ? 8 : 4);
    constexpr int VW = V == 5 ? 4 : V;     // pixels per thread and access
    const int base = blockIdx.x * (256 * VW * ITER);
    const int Pv = V == 5 ? (P & ~3) : P;  // pixels covered by vector accesses
    if constexpr (V == 5) {                // the last P & 3 pixels of the plane: one thread each, first block
        if (blockIdx.x == 0 && (int)threadIdx.x < P - Pv) {
            const int p = Pv + threadIdx.x;
            const float z = gate_sigmoid(gz[p], s1, t1);
            const float n = tanhf_fast(fmaf(cc[p], s2, t2));
            oo[p] = gru_blend(z, n, hh[p]);
        }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int p = base + (it * 256 + threadIdx.x) * VW;
        if (p >= Pv) break;
        if constexpr (V == 5) {
            const f32x4 g = *reinterpret_cast<const f32x4u *>(gz + p);
            const f32x4 cv = *reinterpret_cast<const f32x4u *>(cc + p);
            const f32x4 hv = *reinterpret_cast<const f32x4u *>(hh + p);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = gate_sigmoid(g[k], s1, t1);
                const float n = tanhf_fast(fmaf(cv[k], s2, t2));
                o[k] = gru_blend(z, n, hv[k]);
            }
            *reinterpret_cast<f32x4u *>(oo + p) = o;
        } else if constexpr (V == 4) {
            const f32x4 g = *reinterpret_cast<const f32x4 *>(gz + p);
            const f32x4 cv = *reinterpret_cast<const f32x4 *>(cc + p);
            const f32x4 hv = *reinterpret_cast<const f32x4 *>(hh + p);
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = gate_sigmoid(g[k], s1, t1);
                const float n = tanhf_fast(fmaf(cv[k], s2, t2));
                o[k] = gru_blend(z, n, hv[k]);
            }
            *reinterpret_cast<f32x4 *>(oo + p) = o;           // (the state is re-read soon: non-temporal stores cost 1 %)
        } else {
            const float z = gate_sigmoid(gz[p], s1, t1);
            const float n = tanhf_fast(fmaf(cc[p], s2, t2));
            oo[p] = gru_blend(z, n, hh[p]);
        }
    }
}

hipError_t urnn_launch_blend(const float *g1, const float *c, const float *h, const float *ss1, const float *ss2, float *out,
                             int B, int F, int P, hipStream_t st)
{
    const bool v4 = (P % 4) == 0, v5 = !v4 && P >= 1024;
    const int per_block = 256 * ((v4 || v5) ? 4 : 1) * 4;
    dim3 grid((P + per_block - 1) / per_block, B * F);
    const BlendFin none = {};
    if (v4) hipLaunchKernelGGL((gru_blend_kernel<4, false>), grid, dim3(256), 0, st, g1, c, h, ss1, ss2, out, F, P, none);
    else if (v5) hipLaunchKernelGGL((gru_blend_kernel<5, false>), grid, dim3(256), 0, st, g1, c, h, ss1, ss2, out, F, P, none);
    else hipLaunchKernelGGL((gru_blend_kernel<1, false>), grid, dim3(256), 0, st, g1, c, h, ss1, ss2, out, F, P, none);
    return hipGetLastError();
}

// GroupNorm finalize of the candidate + blend in one launch (see gru_blend_kernel FIN)
hipError_t urnn_launch_blend_fin(const float *g1, const float *c, const float *h, const float *ss1, float *out, int B, int F, int P,
                                 const float *partial, int ntiles, int tile_pix, double count, const float *gamma, const float *beta, float eps,
                                 float *ss2, float *stat2, int *status, hipStream_t st)
{
    const bool v4 = (P % 4) == 0, v5 = !v4 && P >= 1024;
    const int vw = (v4 || v5) ? 4 : 1;
    auto blocks = [&](int iter) { return (long)((P + 256 * vw * iter - 1) / (256 * vw * iter)) * B * F; };
    const int iter = blocks(8) >= 512 ? 8 : (blocks(2) >= 512 ? 2 : 1);
    const int per_block = 256 * vw * iter;
    dim3 grid((P + per_block - 1) / per_block, B * F);
    const BlendFin fin = {partial, ntiles, tile_pix, count, gamma, beta, eps, ss2, stat2, status};
#define URNN_BLEND_FIN(V_, I_) hipLaunchKernelGGL((gru_blend_kernel<V_, true, I_>), grid, dim3(256), 0, st, g1, c, h, ss1, nullptr, out, F, P, fin)
#define URNN_BLEND_FIN_V(V_) do { if (iter == 8) URNN_BLEND_FIN(V_, 8); else if (iter == 2) URNN_BLEND_FIN(V_, 2); else URNN_BLEND_FIN(V_, 1); } while (0)
    if (v4) URNN_BLEND_FIN_V(4);
    else if (v5) URNN_BLEND_FIN_V(5);
    else URNN_BLEND_FIN_V(1);
#undef URNN_BLEND_FIN_V
#undef URNN_BLEND_FIN
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Head (flood_head.py:131-202).  C = 16 channels.  LayerNorm([16,H,W]) statistics are whole-sample reductions => four passes:
//   k1: stats(u0 = Ws.f)                                   k2: t = SiLU(LN0(u0)); u1 = Wc1.t, u2 = Wq1.t (+stats)
//   k3: u1 <- Wc2.SiLU(LN1(u1)), u2 <- Wq2.SiLU(LN3(u2)) (+stats)      k4: cls / reg predictions + wet/dry mask
// The element-wise LayerNorm affines (16,H,W) -- ten planes sets of 16 MB at 500 x 500 -- are the dominant HBM stream.
//
// Register layout (round 5; rounds 1-4 kept a pixel's 16 channels in one thread, which allowed 4- / 8-byte accesses only -- 16-byte
// ones needed 256 registers -- and spent 256 fma per pixel and layer on the VALU): a WAVE owns a 64-pixel tile; lane l = 16 cg + pq
// holds channels 4 cg .. 4 cg + 3 of pixels 4 pq .. 4 pq + 3 as four f32x4 (x[i][j] = channel 4 cg + i, pixel 4 pq + j).  Every
// HBM access is 16 bytes per lane, 256 contiguous bytes per channel row, and the 16 x 16 channel mix runs on the matrix pipe:
// v_mfma_f32_16x16x4_f32 takes B[k = l / 16][n = l % 16] and returns D[m = 4 (l / 16) + r][n = l % 16] in four registers, so with
// k-block i = the channels {4 kk + i} the B operand of pixel j IS x[i][j] and the result lands in the same layout: 16 MFMAs per
// tile and layer (exact fp32 products, fp32 accumulation), no data movement.  A block = 4 waves = 256 pixels = one LayerNorm partial.
// Planes whose size is not a multiple of four (AL = false) use element-wise guarded accesses with the same layout and arithmetic.
// ------------------------------------------------------------------------------------------------------------------
#define HEAD_C 16
#define HEAD_BLOCK_PIX 256

struct HeadLane {
    int cg, pq;     // channel group, pixel quad of the wave's tile
    int p;          // the lane's first pixel
    int npx;        // how many of its four pixels lie inside the plane (AL: 0 or 4)
};

__device__ __forceinline__ HeadLane head_lane(int blk, int P)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    HeadLane L;
    L.cg = lane >> 4;
    L.pq = lane & 15;
    L.p = (blk * 4 + wave) * 64 + 4 * L.pq;
    const int left = P - L.p;
    L.npx = left >= 4 ? 4 : (left > 0 ? left : 0);
    return L;
}

__device__ __forceinline__ void head_zero(f32x4 (&x)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// the lane's 4 x 4 values of a (16, P) tensor; lanes / pixels outside the plane read as 0
template <bool AL>
__device__ __forceinline__ void head_load(const float *__restrict__ src, int P, const HeadLane &L, f32x4 (&x)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float *row = src + (size_t)(4 * L.cg + i) * P + L.p;
        if constexpr (AL) {
            x[i] = L.npx > 0 ? *reinterpret_cast<const f32x4 *>(row) : f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[i][j] = j < L.npx ? row[j] : 0.f;
        }
    }
}

template <bool AL>
__device__ __forceinline__ void head_store(float *dst, int P, const HeadLane &L, const f32x4 (&x)[4])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float *row = dst + (size_t)(4 * L.cg + i) * P + L.p;
        if constexpr (AL) {
            if (L.npx > 0) *reinterpret_cast<f32x4 *>(row) = x[i];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < L.npx) row[j] = x[i][j];
        }
    }
}

// A operand of the four k-blocks: a[i] = W[m = l % 16][4 (l / 16) + i]  (w: 16 x 16 row-major, 16-byte aligned)
__device__ __forceinline__ f32x4 head_weights(const float *__restrict__ w)
{
    const int lane = threadIdx.x & 63;
    return *reinterpret_cast<const f32x4u *>(w + (lane & 15) * HEAD_C + 4 * (lane >> 4));   // (a parameter view: 4-byte alignment only)
}

// u[m][pixel] = sum_c W[m][c] x[c][pixel] on the matrix pipe (whole waves only: never inside a divergent branch)
__device__ __forceinline__ void head_conv(const f32x4 &a, const f32x4 (&x)[4], f32x4 (&u)[4])
{
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], x[i][j], acc[j], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[r][j] = acc[j][r];
}

// x <- SiLU((x - mean) * rstd * gamma + beta); a pixel outside the plane (x = gamma = beta = 0) stays 0
__device__ __forceinline__ void head_ln_silu(f32x4 (&x)[4], const f32x4 (&g)[4], const f32x4 (&bt)[4], float mean, float rstd)
{
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = siluf_fast((x[i][j] - mean) * rstd * g[i][j] + bt[i][j]);
}

// Block-wide sums of two values, returned to every thread (fixed order: xor butterfly per wave, then waves 0..3).
__device__ __forceinline__ void head_block_sum2(float &a, float &b)
{
    __shared__ float sh[2][4];
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
        sh[0][wave] = a;
        sh[1][wave] = b;
    }
    __syncthreads();
    a = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    b = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
}

// LayerNorm partial statistics (sum, second moment about the block's own mean -- urnn_common.h tile_x2) of up to two tensors from
// per-lane scalars: a lane reduces its 4 x npx values to (sum s, squares q about its own mean) while they are in registers, so the
// values can be stored and die before any block-wide step; the block then combines the lanes exactly (Chan):
// Q = sum_lanes [ q + n (m_lane - m_block)^2 ].  Lanes outside the plane pass n = 0.
template <bool AL>
__device__ __forceinline__ void head_lane_stats(const f32x4 (&u)[4], int npx, float &s, float &q)
{
    s = 0.f;
    q = 0.f;
    if constexpr (AL) {            // npx = 4, or 0 with u = 0 (a conv of zeros): no predicates
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += u[i][j];
        const float m = s * (1.f / 16.f);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = u[i][j] - m;
                q = fmaf(d, d, q);
            }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += j < npx ? u[i][j] : 0.f;
        const float m = npx > 0 ? s / (float)(4 * npx) : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = j < npx ? u[i][j] - m : 0.f;
                q = fmaf(d, d, q);
            }
    }
}

// COH: the partials cross a coop_grid_barrier_nf of this launch (head_coop_kernel): 8-byte agent-scope stores (urnn_common.h publish8)
template <bool COH = false>
__device__ __forceinline__ void head_block_stats2(float sa, float qa, float sb, float qb, int n_lane, bool two, int nvalid, float *dst_a,
                                                  float *dst_b)
{
    const float fn = (float)n_lane;
    const float ma = n_lane > 0 ? sa / fn : 0.f, mb = n_lane > 0 ? sb / fn : 0.f;
    float Sa = sa, Sb = sb;
    head_block_sum2(Sa, Sb);
    const float inv = 1.f / (float)nvalid;
    const float da = ma - Sa * inv, db = mb - Sb * inv;
    float Qa = n_lane > 0 ? fmaf(fn * da, da, qa) : 0.f, Qb = n_lane > 0 ? fmaf(fn * db, db, qb) : 0.f;
    head_block_sum2(Qa, Qb);
    if (threadIdx.x == 0) {
        if constexpr (COH) {
            publish8(dst_a, Sa, Qa);
            if (two) publish8(dst_b, Sb, Qb);
        } else {
            dst_a[0] = Sa;
            dst_a[1] = Qa;
            if (two) {
                dst_b[0] = Sb;
                dst_b[1] = Qb;
            }
        }
    }
}

// partial layout: [which(5)][B][nblk][2];  stats layout: [which(5)][B][2]
__device__ __forceinline__ float *head_partial(const HeadParams &p, int which, int b, int blk)
{
    return p.partial + ((((size_t)which * p.B + b) * p.nblk) + blk) * 2;
}

// LayerNorm statistics of tensor `which` of sample b folded from the per-block partials by ONE wave: lane-strided double sums in
// ascending block order, then the xor butterfly -- the order ln_finalize_kernel uses, so every path gives identical bits.
// Who folds (round 5): a finalize launch of ONE wave per (norm, sample) between the passes (ln_finalize_kernel); the consumers read
// two floats.  Rounds 1-4 had every wave of every consuming block fold all partials in front of (or, tried this round, behind) its
// own loads -- ~1 000 blocks x 4 waves x 8 KB of L2 reads, a dependent round trip and ~60 registers per tile: head_k4 32 -> 20 us
// without it, the three finalize launches cost 3 x 3-5 us.  (Also tried: the producer's last block to finish folds -- one
// agent-scope counter and a release fence per block: 17-40 ns per block SERIALISED, head_k3 31 -> 110 us.)
// The cooperative head folds across its grid barriers (small planes); a feature map whose producer took the first norm's partials
// (the decoder's last conv, conv_gemm_kernel's stemW epilogue, or urnn_tail.hip) is finalized from there (prm.partial0).
template <bool COH = false>
__device__ __forceinline__ void head_fold_stats(const HeadParams &prm, int which, int b, bool publish, float &mean_f, float &rstd_f)
{
    const int lane = threadIdx.x & 63;
    const float *pp = prm.partial + (((size_t)which * prm.B + b) * prm.nblk) * 2;
    int nblk_used = (prm.P + HEAD_BLOCK_PIX - 1) / HEAD_BLOCK_PIX, block_pix = HEAD_BLOCK_PIX;
    if (which == 0 && prm.partial0) {                  // statistics of u0 taken by the producer of feat (urnn_tail.hip): its own block size
        pp = prm.partial0 + (size_t)b * prm.nblk0 * 2;
        nblk_used = prm.nblk0;
        block_pix = prm.bpix0;
    }
    double s1, s2;
    fold_lane_chain<8, true, COH>(pp, nblk_used, block_pix, HEAD_C, prm.P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    const double count = (double)HEAD_C * (double)prm.P;
    const double mean = s1 / count;
    double var = s2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
    var = var > 0.0 ? var : 0.0;
    mean_f = (float)mean;
    rstd_f = (float)(1.0 / sqrt(var + (double)prm.eps));
    if (publish && lane == 0) {                      // the consumers (and the backward pass) read the statistics from prm.stats
        flag_nonfinite(prm.status, URNN_STATUS_HEAD, s1, s2);
        prm.stats[((size_t)which * prm.B + b) * 2] = mean_f;
        prm.stats[((size_t)which * prm.B + b) * 2 + 1] = rstd_f;
    }
}
// (mean, rstd) of tensor `which` as its producer's last block (or a finalize launch: strips, phase-split callers) left them
__device__ __forceinline__ void head_stats(const HeadParams &prm, int which, int b, float &mean, float &rstd)
{
    mean = prm.stats[(which * prm.B + b) * 2];
    rstd = prm.stats[(which * prm.B + b) * 2 + 1];
}

// prediction layer of one branch: a[j] = sum_c w[c] x[c][pixel j] + bias -- the lane's four channels, then the four channel groups
// (xor 16, xor 32: every lane of a pixel quad ends with the same bits)
__device__ __forceinline__ f32x4 head_pred(const float *__restrict__ w, const float *__restrict__ bias, int cg, const f32x4 (&x)[4])
{
    f32x4 a;
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float wv = w[4 * cg + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(wv, x[i][j], a[j]);
    }
    const float bv = bias[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        a[j] += __shfl_xor(a[j], 16, 64);
        a[j] += __shfl_xor(a[j], 32, 64);
        a[j] += bv;
    }
    return a;
}

// cls / masked reg / raw reg of the lane's four pixels: channel group 0 stores cls, 1 the masked depth, 2 the raw one
template <bool AL>
__device__ __forceinline__ void head_outputs(const HeadParams &prm, int b, const HeadLane &L, const f32x4 &acls, const f32x4 &areg)
{
    if (L.npx == 0 || L.cg == 3) return;
    const int frame = prm.frame_index ? *prm.frame_index : 0;
    const size_t obase = ((size_t)frame * prm.B + b) * prm.P + L.p;
    f32x4 cls, reg, o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        cls[j] = sigmoidf_fast(acls[j]);
        reg[j] = lrelu(areg[j], prm.slope);
    }
    float *dst;
    if (L.cg == 0) {
        dst = prm.out_cls;
        o = cls;
    } else if (L.cg == 1) {
        dst = prm.out_masked;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = reg[j] * (cls[j] >= prm.cls_thred ? 1.f : 0.f);
    } else {
        dst = prm.out_raw;
        o = reg;
        if (!dst) return;
    }
    if constexpr (AL) *reinterpret_cast<f32x4u *>(dst + obase) = o;   // (the caller's frames: 4-byte alignment only)
    else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < L.npx) dst[obase + j] = o[j];
    }
}

template <bool AL>
__global__ __launch_bounds__(256) void head_k1(const HeadParams prm)
{
    const int b = blockIdx.y;
    const HeadLane L = head_lane(blockIdx.x, prm.P);
    if (prm.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *prm.bump = *prm.frame_index + 1;      // the NEXT head's frame word (nothing in flight reads it)
    f32x4 f[4], u[4];
    head_load<AL>(prm.feat + (size_t)b * HEAD_C * prm.P, prm.P, L, f);
    head_conv(head_weights(prm.conv_w), f, u);
    float s, q;
    head_lane_stats<AL>(u, L.npx, s, q);
    head_block_stats2(s, q, 0.f, 0.f, 4 * L.npx, false, HEAD_C * tile_valid(blockIdx.x, HEAD_BLOCK_PIX, prm.P), head_partial(prm, 0, b, blockIdx.x),
                      nullptr);
}

template <bool AL>
__global__ __launch_bounds__(256) void head_k2(const HeadParams prm)
{
    const int b = blockIdx.y;
    const HeadLane L = head_lane(blockIdx.x, prm.P);
    const size_t CP = (size_t)HEAD_C * prm.P;
    if (prm.partial0 && prm.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *prm.bump = *prm.frame_index + 1;   // (head_k1 did not run)
    f32x4 f[4], t[4], g[4], bt[4], u[4];
    head_load<AL>(prm.feat + b * CP, prm.P, L, f);               // the tile's rows travel while the statistics are folded
    head_load<AL>(prm.ln_w, prm.P, L, g);
    head_load<AL>(prm.ln_b, prm.P, L, bt);
    const f32x4 a0 = head_weights(prm.conv_w), a1 = head_weights(prm.conv_w + 1 * HEAD_C * HEAD_C), a3 = head_weights(prm.conv_w + 3 * HEAD_C * HEAD_C);
    float m0, r0;
    head_stats(prm, 0, b, m0, r0);
    head_conv(a0, f, t);
    head_ln_silu(t, g, bt, m0, r0);
    float sc, qc, sq, qq;
    head_conv(a1, t, u);
    head_lane_stats<AL>(u, L.npx, sc, qc);
    head_store<AL>(prm.u1 + b * CP, prm.P, L, u);
    head_conv(a3, t, u);
    head_lane_stats<AL>(u, L.npx, sq, qq);
    head_store<AL>(prm.u2 + b * CP, prm.P, L, u);
    head_block_stats2(sc, qc, sq, qq, 4 * L.npx, true, HEAD_C * tile_valid(blockIdx.x, HEAD_BLOCK_PIX, prm.P), head_partial(prm, 1, b, blockIdx.x),
                      head_partial(prm, 3, b, blockIdx.x));
}

// blockIdx.z: the branch (0: cls, norms 1 -> 2 in u1; 1: reg, norms 3 -> 4 in u2) -- twice the blocks at half the registers
template <bool AL>
__global__ __launch_bounds__(256) void head_k3(const HeadParams prm)
{
    const int b = blockIdx.y, br = blockIdx.z;
    const HeadLane L = head_lane(blockIdx.x, prm.P);
    const size_t CP = (size_t)HEAD_C * prm.P;
    const int win = br ? 3 : 1, wout = br ? 4 : 2;
    float *buf = (br ? prm.u2 : prm.u1) + b * CP;
    f32x4 x[4], g[4], bt[4], u[4];
    head_load<AL>(buf, prm.P, L, x);
    head_load<AL>(prm.ln_w + win * CP, prm.P, L, g);
    head_load<AL>(prm.ln_b + win * CP, prm.P, L, bt);
    const f32x4 a = head_weights(prm.conv_w + wout * HEAD_C * HEAD_C);
    float m, r;
    head_stats(prm, win, b, m, r);
    head_ln_silu(x, g, bt, m, r);
    head_conv(a, x, u);
    float s, q;
    head_lane_stats<AL>(u, L.npx, s, q);
    head_store<AL>(buf, prm.P, L, u);
    head_block_stats2(s, q, 0.f, 0.f, 4 * L.npx, false, HEAD_C * tile_valid(blockIdx.x, HEAD_BLOCK_PIX, prm.P), head_partial(prm, wout, b, blockIdx.x),
                      nullptr);
}

template <bool AL>
__global__ __launch_bounds__(256) void head_k4(const HeadParams prm)
{
    const int b = blockIdx.y;
    const HeadLane L = head_lane(blockIdx.x, prm.P);
    const size_t CP = (size_t)HEAD_C * prm.P;
    f32x4 x1[4], g1[4], b1[4], x2[4], g2[4], b2[4];
    head_load<AL>(prm.u1 + b * CP, prm.P, L, x1);
    head_load<AL>(prm.ln_w + 2 * CP, prm.P, L, g1);
    head_load<AL>(prm.ln_b + 2 * CP, prm.P, L, b1);
    head_load<AL>(prm.u2 + b * CP, prm.P, L, x2);
    head_load<AL>(prm.ln_w + 4 * CP, prm.P, L, g2);
    head_load<AL>(prm.ln_b + 4 * CP, prm.P, L, b2);
    float m2, r2, m4, r4;
    head_stats(prm, 2, b, m2, r2);
    head_stats(prm, 4, b, m4, r4);
    head_ln_silu(x1, g1, b1, m2, r2);
    const f32x4 acls = head_pred(prm.cls_w, prm.cls_b, L.cg, x1);
    head_ln_silu(x2, g2, b2, m4, r4);
    const f32x4 areg = head_pred(prm.reg_w, prm.reg_b, L.cg, x2);
    head_outputs<AL>(prm, b, L, acls, areg);
}

// The whole head of a SMALL plane in one cooperative launch (URNN head of the 64x64 / 52x120 / 128x128 configs: four launches of 5-18 us
// for microseconds of work): head_k1 | grid barrier | head_k2 | grid barrier | head_k3 | grid barrier | head_k4, a lane keeping its
// pixels' 2 x 16 branch activations in registers from pass to pass -- nothing but the partial LayerNorm statistics leaves the CU.
// Same device functions, same block geometry and summation orders as the four kernels: identical bits.  Every block must be
// resident (<= 256 blocks; with two kernel chains in flight the caller passes <= 128, as for the cooperative cells).
template <bool AL>
__global__ __launch_bounds__(256) void head_coop_kernel(const HeadParams prm, unsigned *bar, int nblocks)
{
    const int b = blockIdx.y;
    const HeadLane L = head_lane(blockIdx.x, prm.P);
    const size_t CP = (size_t)HEAD_C * prm.P;
    const int nvalid = HEAD_C * tile_valid(blockIdx.x, HEAD_BLOCK_PIX, prm.P);
    if (prm.bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *prm.bump = *prm.frame_index + 1;
    const bool pub = blockIdx.x == 0 && threadIdx.x < 64;           // who leaves (mean, rstd) in prm.stats (the backward pass reads them)
    f32x4 u1[4], u2[4], g[4], bt[4];
    // ---- pass 1 (head_k1): u0 = Ws . f and its statistics; u0 stays in u1
    {
        f32x4 f[4];
        head_load<AL>(prm.feat + b * CP, prm.P, L, f);
        head_conv(head_weights(prm.conv_w), f, u1);
        float s, q;
        head_lane_stats<AL>(u1, L.npx, s, q);
        head_block_stats2<true>(s, q, 0.f, 0.f, 4 * L.npx, false, nvalid, head_partial(prm, 0, b, blockIdx.x), nullptr);
    }
    head_load<AL>(prm.ln_w, prm.P, L, g);                          // (the next pass's affine rows travel across the barrier)
    head_load<AL>(prm.ln_b, prm.P, L, bt);
    coop_grid_barrier_nf(bar, blockIdx.y * gridDim.x + blockIdx.x, (unsigned)nblocks, prm.status);
    // ---- pass 2 (head_k2): t = SiLU(LN0(u0)); u1 = Wc1 . t, u2 = Wq1 . t
    {
        float m0, r0;
        head_fold_stats<true>(prm, 0, b, pub, m0, r0);
        float sc, qc, sq, qq;
        f32x4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = u1[i];
        head_ln_silu(t, g, bt, m0, r0);
        head_conv(head_weights(prm.conv_w + 1 * HEAD_C * HEAD_C), t, u1);
        head_lane_stats<AL>(u1, L.npx, sc, qc);
        head_conv(head_weights(prm.conv_w + 3 * HEAD_C * HEAD_C), t, u2);
        head_lane_stats<AL>(u2, L.npx, sq, qq);
        head_block_stats2<true>(sc, qc, sq, qq, 4 * L.npx, true, nvalid, head_partial(prm, 1, b, blockIdx.x), head_partial(prm, 3, b, blockIdx.x));
    }
    head_load<AL>(prm.ln_w + 1 * CP, prm.P, L, g);
    head_load<AL>(prm.ln_b + 1 * CP, prm.P, L, bt);
    coop_grid_barrier_nf(bar, blockIdx.y * gridDim.x + blockIdx.x, (unsigned)nblocks, prm.status);
    // ---- pass 3 (head_k3): u1 <- Wc2 . SiLU(LN1(u1)), u2 <- Wq2 . SiLU(LN3(u2))
    {
        float m1, r1, m3, r3;
        head_fold_stats<true>(prm, 1, b, pub, m1, r1);
        head_fold_stats<true>(prm, 3, b, pub, m3, r3);
        float sc, qc, sq, qq;
        f32x4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = u1[i];
        head_ln_silu(x, g, bt, m1, r1);
        head_load<AL>(prm.ln_w + 3 * CP, prm.P, L, g);
        head_load<AL>(prm.ln_b + 3 * CP, prm.P, L, bt);
        head_conv(head_weights(prm.conv_w + 2 * HEAD_C * HEAD_C), x, u1);
        head_lane_stats<AL>(u1, L.npx, sc, qc);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = u2[i];
        head_ln_silu(x, g, bt, m3, r3);
        head_conv(head_weights(prm.conv_w + 4 * HEAD_C * HEAD_C), x, u2);
        head_lane_stats<AL>(u2, L.npx, sq, qq);
        head_block_stats2<true>(sc, qc, sq, qq, 4 * L.npx, true, nvalid, head_partial(prm, 2, b, blockIdx.x), head_partial(prm, 4, b, blockIdx.x));
    }
    head_load<AL>(prm.ln_w + 2 * CP, prm.P, L, g);
    head_load<AL>(prm.ln_b + 2 * CP, prm.P, L, bt);
    coop_grid_barrier_nf(bar, blockIdx.y * gridDim.x + blockIdx.x, (unsigned)nblocks, prm.status);
    // ---- pass 4 (head_k4): predictions + wet / dry mask
    {
        float m2, r2, m4, r4;
        head_fold_stats<true>(prm, 2, b, pub, m2, r2);
        head_fold_stats<true>(prm, 4, b, pub, m4, r4);
        head_ln_silu(u1, g, bt, m2, r2);
        head_load<AL>(prm.ln_w + 4 * CP, prm.P, L, g);
        head_load<AL>(prm.ln_b + 4 * CP, prm.P, L, bt);
        const f32x4 acls = head_pred(prm.cls_w, prm.cls_b, L.cg, u1);
        head_ln_silu(u2, g, bt, m4, r4);
        const f32x4 areg = head_pred(prm.reg_w, prm.reg_b, L.cg, u2);
        head_outputs<AL>(prm, b, L, acls, areg);
    }
}

// LayerNorm finalise: one block per (which, sample); which = first + i * stride for i < n (blockIdx.y).
__global__ __launch_bounds__(64) void ln_finalize_kernel(const HeadParams prm, int first, int stride, int nblk_used, int block_pix)
{
    const int which = first + blockIdx.y * stride;
    const int b = blockIdx.x;
    const int lane = threadIdx.x;
    const float *pp = prm.partial + (((size_t)which * prm.B + b) * prm.nblk) * 2;
    if (which == 0 && prm.partial0) {                  // statistics of u0 taken by the producer of feat: its own block size
        pp = prm.partial0 + (size_t)b * prm.nblk0 * 2;
        nblk_used = prm.nblk0;
        block_pix = prm.bpix0;
    }
    double s1, s2;
    fold_lane_chain<16>(pp, nblk_used, block_pix, HEAD_C, prm.P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares)
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane == 0) {
        const double count = (double)HEAD_C * (double)(prm.Pglobal > 0 ? prm.Pglobal : (long)prm.P);
        const double mean = s1 / count;
        double var = s2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
        flag_nonfinite(prm.status, URNN_STATUS_HEAD, s1, s2);
        var = var > 0.0 ? var : 0.0;
        prm.stats[((size_t)which * prm.B + b) * 2] = (float)mean;
        prm.stats[((size_t)which * prm.B + b) * 2 + 1] = (float)(1.0 / sqrt(var + (double)prm.eps));
    }
}

int urnn_head_nblk(int P) { const int n = (P + 255) / 256; return n < 2 ? 2 : n; }   // >= 2: the strip mode's two pseudo-blocks
int urnn_head_nblk_used(int P) { return (P + HEAD_BLOCK_PIX - 1) / HEAD_BLOCK_PIX; }
int urnn_head_block_pix(int) { return HEAD_BLOCK_PIX; }

template <bool AL>
static hipError_t launch_head_v(const HeadParams &p, int mask, hipStream_t st)
{
    const int nb = urnn_head_nblk_used(p.P);
    const int fin = p.Pglobal > 0 ? 2 : nb;                  // strip mode: the partials hold the all-reduced totals as two pseudo-blocks
    const int bpix = p.Pglobal > 0 ? 0 : HEAD_BLOCK_PIX;     //             ... which are raw (sum, sum of squares)
    dim3 grid(nb, p.B), grid3(nb, p.B, 2), blk(256);
    const bool tail = p.partial0 != nullptr;                 // the producer of feat took norm 0's partials (the decoder's last conv, or the fused cell tail): no head_k1
    if ((mask & URNN_HEAD_K1) && !tail) hipLaunchKernelGGL(head_k1<AL>, grid, blk, 0, st, p);
    if (mask & URNN_HEAD_F1) hipLaunchKernelGGL(ln_finalize_kernel, dim3(p.B, 1), dim3(64), 0, st, p, 0, 1, fin, bpix);   // (folds p.partial0 when set)
    if (mask & URNN_HEAD_K2) hipLaunchKernelGGL(head_k2<AL>, grid, blk, 0, st, p);
    if (mask & URNN_HEAD_F2) hipLaunchKernelGGL(ln_finalize_kernel, dim3(p.B, 2), dim3(64), 0, st, p, 1, 2, fin, bpix);
    if (mask & URNN_HEAD_K3) hipLaunchKernelGGL(head_k3<AL>, grid3, blk, 0, st, p);
    if (mask & URNN_HEAD_F3) hipLaunchKernelGGL(ln_finalize_kernel, dim3(p.B, 2), dim3(64), 0, st, p, 2, 2, fin, bpix);
    if (mask & URNN_HEAD_K4) hipLaunchKernelGGL(head_k4<AL>, grid, blk, 0, st, p);
    return hipGetLastError();
}

// blocks a cooperative head launch takes; 0 when this device could not hold them all at once (one 256-thread block per CU is the
// residency the callers' rules are written for)
int urnn_head_coop_blocks(int B, int P)
{
    const int n = B * urnn_head_nblk_used(P);
    return n <= urnn_device_cus() ? n : 0;
}

hipError_t urnn_launch_head_coop(const HeadParams &p, unsigned *bar, hipStream_t st)
{
    const int nb = urnn_head_nblk_used(p.P);
    const dim3 grid(nb, p.B), blk(256);
    if (p.P % 4 == 0) hipLaunchKernelGGL(head_coop_kernel<true>, grid, blk, 0, st, p, bar, nb * p.B);
    else hipLaunchKernelGGL(head_coop_kernel<false>, grid, blk, 0, st, p, bar, nb * p.B);
    return hipGetLastError();
}

hipError_t urnn_launch_head(const HeadParams &p, int mask, hipStream_t st)
{
    if (p.P % 4 == 0) return launch_head_v<true>(p, mask, st);
    return launch_head_v<false>(p, mask, st);
}

// ------------------------------------------------------------------------------------------------------------------
// Strip mode (a plane split over ranks in horizontal strips): norm statistics leave as double (sum, sum of squares)
// and come back, all-reduced, as two RAW pseudo-tiles (hi + lo floats) that the finalizes add up in double (tile_pix = 0).
// partial[row][stride][2]; one wave per row.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void stats_reduce_kernel(const float *__restrict__ partial, int stride, int ntiles, int tile_pix, int P, int chans,
                                                          double *__restrict__ sums)
{
    const int row = blockIdx.x, lane = threadIdx.x;
    const float *pp = partial + (size_t)row * stride * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int t = lane; t < ntiles; t += 64) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(pp + 2 * t);
        s1 += (double)v.x;
        s2 += tile_x2(v.x, v.y, chans * tile_valid(t, tile_pix, P));      // centred tile partials -> raw second moment, in double
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m, 64);
        s2 += __shfl_xor(s2, m, 64);
    }
    if (lane == 0) {
        sums[2 * row] = s1;
        sums[2 * row + 1] = s2;
    }
}

__global__ void stats_scatter_kernel(const double *__restrict__ sums, int rows, int stride, float *__restrict__ partial)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (row, component)
    if (i >= rows * 2) return;
    const int row = i >> 1, comp = i & 1;
    const double v = sums[i];
    const float hi = (float)v, lo = (float)(v - (double)hi);
    float *pp = partial + (size_t)row * stride * 2;
    pp[comp] = hi;
    pp[2 + comp] = lo;
}

hipError_t urnn_launch_stats_reduce(const float *partial, int rows, int stride, int ntiles, int tile_pix, int P, int chans, double *sums,
                                    hipStream_t st)
{
    hipLaunchKernelGGL(stats_reduce_kernel, dim3(rows), dim3(64), 0, st, partial, stride, ntiles, tile_pix, P, chans, sums);
    return hipGetLastError();
}

hipError_t urnn_launch_stats_scatter(const double *sums, int rows, int stride, float *partial, hipStream_t st)
{
    hipLaunchKernelGGL(stats_scatter_kernel, dim3((rows * 2 + 255) / 256), dim3(256), 0, st, sums, rows, stride, partial);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Per-frame input assembly (Dynamic2DFlood.py:265-376).  grid = (chunks, B*C); one output plane per blockIdx.y.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void preprocess_kernel(const float *__restrict__ rain, const float *__restrict__ cumsum,
                                                         const float *__restrict__ dem, const float *__restrict__ imperv,
                                                         const float *__restrict__ manhole, float dem_min, float dem_max,
                                                         float *__restrict__ out, int t_host, const int *__restrict__ t_dev,
                                                         int T, int nums, int P, int spatial, float rain_max, float cumsum_max, int *bump)
{
    const int C = 2 * nums + 3;
    const int bc = blockIdx.y;
    const int b = bc / C, c = bc - b * C;
    const int t = t_dev ? *t_dev : t_host;
    if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump = t + 1;     // the next frame's word (never t_dev itself: other blocks still read it)
    float *dst = out + (size_t)bc * P;
    const int p0 = blockIdx.x * 1024;
    const int pend = min(P, p0 + 1024);
    if (c < 2 * nums) {
        const int which = c / nums, slot = c - which * nums;
        const int start = max(0, t - nums + 1), end = min(t + 1, T);
        const int nsteps = end - start;
        const int k = slot - (nums - nsteps);
        const float *src = which ? cumsum : rain;
        const float mx = which ? cumsum_max : rain_max;
        if (k < 0) {
            for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = 0.f;
        } else if (!spatial) {
            const float v = (src[(size_t)b * T + start + k] - 0.f) / (mx - 0.f);
            for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = v;
        } else {
            const float *plane = src + ((size_t)b * T + start + k) * P;
            for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = (plane[p] - 0.f) / (mx - 0.f);
        }
    } else if (c == 2 * nums) {
        const float *plane = dem + (size_t)b * P;
        const float span = dem_max - dem_min;
        for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = (plane[p] - dem_min) / span;
    } else if (c == 2 * nums + 1) {
        const float *plane = imperv + (size_t)b * P;
        const float lo = 0.05f, span = 0.9f;   // python-double 0.95 - 0.05 rounded once to fp32, as torch does
        for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = (plane[p] - lo) / span;
    } else {
        const float *plane = manhole + (size_t)b * P;
        for (int p = p0 + threadIdx.x; p < pend; p += 256) dst[p] = plane[p];
    }
}

hipError_t urnn_launch_preprocess(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                                  const float *manhole, float dem_min, float dem_max, float *out, int t, const int *t_dev,
                                  int B, int T, int nums, int P, int spatial, float rain_max, float cumsum_max,
                                  hipStream_t st, int *bump)
{
    const int C = 2 * nums + 3;
    dim3 grid((P + 1023) / 1024, B * C);
    hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, st, rain, cumsum, dem, imperv, manhole, dem_min, dem_max, out, t,
                       t_dev, T, nums, P, spatial, rain_max, cumsum_max, bump);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Scalar-rainfall fast path of the first encoder stage.  With one rainfall value per frame
// (UrbanFlood24), 2*nums of the 2*nums+3 input channels are spatial constants, so
//   stage1(x_t)[n][p] = lrelu( S[n][p] + v_t[n] ),   S = W[:, static] . [norm DEM, norm impervious, manhole]   (once per event)
//                                                     v_t = b + W[:, rain] . rain history(t) / max           (16 numbers per frame)
// and the (B, 2*nums+3, H, W) input tensor of preprocess_inputs (63 MB per frame at 500x500) never materialises.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage1_static_kernel(const float *__restrict__ dem, const float *__restrict__ imperv,
                                                            const float *__restrict__ manhole, float dem_min, float dem_max,
                                                            const float *__restrict__ w /* (Cout, 2*nums+3) */, float *__restrict__ S,
                                                            int nums, int Cout, int P)
{
    const int b = blockIdx.y;
    const int C = 2 * nums + 3;
    const float span = dem_max - dem_min;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const float d = (dem[(size_t)b * P + p] - dem_min) / span;
        const float im = (imperv[(size_t)b * P + p] - 0.05f) / 0.9f;
        const float mh = manhole[(size_t)b * P + p];
        for (int n = 0; n < Cout; ++n) {
            const float *wr = w + (size_t)n * C + 2 * nums;
            S[((size_t)b * Cout + n) * P + p] = fmaf(wr[2], mh, fmaf(wr[1], im, wr[0] * d));
        }
    }
}

template <int V>
__global__ __launch_bounds__(256) void stage1_scalar_kernel(const float *__restrict__ S, const float *__restrict__ rain,
                                                            const float *__restrict__ cumsum, const float *__restrict__ w,
                                                            const float *__restrict__ bias, float *__restrict__ out, int t_host,
                                                            const int *__restrict__ t_dev, int T, int nums, int Cout, int P,
                                                            float rain_max, float cumsum_max, float slope, int *bump)
{
    __shared__ float vt[64];
    const int bn = blockIdx.y;                    // (sample, output channel)
    const int b = bn / Cout, n = bn - b * Cout;
    const int C = 2 * nums + 3;
    const int t = t_dev ? *t_dev : t_host;
    if (bump && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *bump = t + 1;     // the next frame's word (never t_dev itself: other blocks still read it)
    if (threadIdx.x < 64) {
        // rain-history part of channel n, one history slot per lane (get_past_rainfall: left zero padding for t < nums)
        const int start = max(0, t - nums + 1), end = min(t + 1, T), nsteps = end - start;
        float acc = 0.f;
        for (int c = threadIdx.x; c < 2 * nums; c += 64) {
            const int which = c / nums, slot = c - which * nums;
            const int k = slot - (nums - nsteps);
            if (k >= 0) {
                const float v = (which ? cumsum : rain)[(size_t)b * T + start + k] / (which ? cumsum_max : rain_max);
                acc = fmaf(w[(size_t)n * C + c], v, acc);
            }
        }
        acc = wave_sum(acc);
        if (threadIdx.x == 0) vt[0] = acc + bias[n];
    }
    __syncthreads();
    const float v = vt[0];
    const float *src = S + (size_t)bn * P;
    float *dst = out + (size_t)bn * P;
    for (int p = (blockIdx.x * 256 + threadIdx.x) * V; p < P; p += gridDim.x * 256 * V) {
        if constexpr (V == 4) {
            f32x4 x = *reinterpret_cast<const f32x4 *>(src + p);
            x.x = lrelu(x.x + v, slope); x.y = lrelu(x.y + v, slope); x.z = lrelu(x.z + v, slope); x.w = lrelu(x.w + v, slope);
            *reinterpret_cast<f32x4 *>(dst + p) = x;
        } else {
            dst[p] = lrelu(src[p] + v, slope);
        }
    }
}

hipError_t urnn_launch_stage1_static(const float *dem, const float *imperv, const float *manhole, float dem_min, float dem_max,
                                     const float *w, float *S, int B, int nums, int Cout, int P, hipStream_t st)
{
    dim3 grid(min((P + 255) / 256, 2048), B);
    hipLaunchKernelGGL(stage1_static_kernel, grid, dim3(256), 0, st, dem, imperv, manhole, dem_min, dem_max, w, S, nums, Cout, P);
    return hipGetLastError();
}

hipError_t urnn_launch_stage1_scalar(const float *S, const float *rain, const float *cumsum, const float *w, const float *bias,
                                     float *out, int t, const int *t_dev, int B, int T, int nums, int Cout, int P, float rain_max,
                                     float cumsum_max, float slope, hipStream_t st, int *bump)
{
    const bool v4 = (P % 4) == 0;
    const int per = 256 * (v4 ? 4 : 1);
    dim3 grid(min((P + per - 1) / per, 64), B * Cout);
    if (v4) hipLaunchKernelGGL(stage1_scalar_kernel<4>, grid, dim3(256), 0, st, S, rain, cumsum, w, bias, out, t, t_dev, T, nums, Cout, P, rain_max, cumsum_max, slope, bump);
    else hipLaunchKernelGGL(stage1_scalar_kernel<1>, grid, dim3(256), 0, st, S, rain, cumsum, w, bias, out, t, t_dev, T, nums, Cout, P, rain_max, cumsum_max, slope, bump);
    return hipGetLastError();
}

__global__ void advance_kernel(int *counter, int delta) { *counter += delta; }

hipError_t urnn_launch_advance(int *counter, int delta, hipStream_t st)
{
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, counter, delta);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------------
// Weight packers (one-off).  Every conv / cell buffer ends with the SPLIT form of the same slabs (bf16 pieces, urnn_common.h
// urnn_split_slab_dwords) for the bf16 x 6 k-loop.  fp32 form: one slab per n-group g (NB 32-column blocks), [KT][NB][64] floats padded to a
// multiple of 256 floats:  slab[g][(kp*NB + nb)*64 + l] = W[row k = 2*kp + (l >> 5)][column n = (g*NB + nb)*32 + (l & 31)]
// i.e. exactly the image conv_gemm_kernel keeps in LDS (every lane reads its MFMA A operand with one conflict-free
// ds_read_b32).  The bias of every packed column follows the slabs.
// ------------------------------------------------------------------------------------------------------------------
__host__ __device__ static inline int slab_floats(int KT, int NB) { return (KT * NB * 64 + 255) / 256 * 256; }

// dword `r` of an n-group's split slab: (16-k group, n-block, piece, lane, dword) -> the two k-pairs it holds
__device__ __forceinline__ void split_slot(int r, int NB, int &kp_even, int &nb, int &piece, int &l)
{
    const int grp = r / (NB * 768), rr = r - grp * (NB * 768);
    nb = rr / 768;
    piece = (rr - nb * 768) / 256;
    l = (rr & 255) >> 2;
    kp_even = 8 * grp + 2 * (rr & 3);
}

// dword `r` of an n-group's f16 slab (two pieces per n-block)
__device__ __forceinline__ void f16_slot(int r, int NB, int &kp_even, int &nb, int &piece, int &l)
{
    const int grp = r / (NB * 512), rr = r - grp * (NB * 512);
    nb = rr / 512;
    piece = (rr - nb * 512) / 256;
    l = (rr & 255) >> 2;
    kp_even = 8 * grp + 2 * (rr & 3);
}
__device__ __forceinline__ unsigned f16_pair(float we, float wo, int piece)
{
    return f16_piece(we * URNN_F16_WSCALE, piece) | (f16_piece(wo * URNN_F16_WSCALE, piece) << 16);
}

__global__ void pack_conv_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ packed, int Cin,
                                 int Cout, int NB, int NG, int KT)
{
    const int slab = slab_floats(KT, NB), ssd = urnn_split_slab_dwords(KT, NB), fsd = urnn_f16_slab_dwords(KT, NB);
    const int nw = NG * slab, Npad = NG * NB * 32;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nw + Npad + NG * ssd + NG * fsd) return;
    auto wv = [&](int g, int kp, int nb, int l) {
        const int k = 2 * kp + (l >> 5), n = (g * NB + nb) * 32 + (l & 31);
        return (kp < KT && k < Cin && n < Cout) ? w[(size_t)n * Cin + k] : 0.f;
    };
    if (idx >= nw + Npad + NG * ssd) {
        const int q = idx - nw - Npad - NG * ssd;
        const int g = q / fsd;
        int kp, nb, piece, l;
        f16_slot(q - g * fsd, NB, kp, nb, piece, l);
        reinterpret_cast<unsigned *>(packed)[idx] = f16_pair(wv(g, kp, nb, l), wv(g, kp + 1, nb, l), piece);
        return;
    }
    if (idx < nw) {
        const int g = idx / slab, r = idx - g * slab;
        const int l = r & 63, row = r >> 6;
        const int kp = row / NB, nb = row - kp * NB;
        packed[idx] = wv(g, kp, nb, l);
    } else if (idx < nw + Npad) {
        const int n = idx - nw;
        packed[idx] = (bias && n < Cout) ? bias[n] : 0.f;
    } else {
        const int q = idx - nw - Npad;
        const int g = q / ssd;
        int kp, nb, piece, l;
        split_slot(q - g * ssd, NB, kp, nb, piece, l);
        reinterpret_cast<unsigned *>(packed)[idx] = bf16_piece(wv(g, kp, nb, l), piece) | (bf16_piece(wv(g, kp + 1, nb, l), piece) << 16);
    }
}

hipError_t urnn_launch_pack_conv(const float *w, const float *bias, float *packed, int Cin, int Cout, hipStream_t st)
{
    const int NB = urnn_conv_nb(Cout), NG = urnn_conv_ng(Cout), KT = (Cin + 1) / 2;
    const int total = NG * slab_floats(KT, NB) + NG * NB * 32 + NG * urnn_split_slab_dwords(KT, NB) + NG * urnn_f16_slab_dwords(KT, NB);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, bias, packed, Cin, Cout, NB, NG, KT);
    return hipGetLastError();
}

// GRU.  Rows (K) of both GEMMs = x (I padded to even) | e (F, decoder only) | h (F); source K order of W1 / W2 is
// cat(x, [e,] h) (ConvRNN.py:153,165-168).
//   gate GEMM:      F/32 groups [z_i | r_i] (NB = 2) from W1, then bias [F/32][z_i(32) | r_i(32)]
//   candidate GEMM: NG2 groups of NB2 = urnn_cand_nb(F) 32-channel blocks from W2, then bias b2 [F]
//   f16 forms (forward k-loop): the gate slab in the grouping of urnn_gate_groups (NGg groups of NBg blocks, block nb of group g
//   = canonical block urnn_gate_cb of [z_0 .. | r_0 ..]), the candidate slab as above, then the gate bias in the grouped order.
__global__ void pack_gru_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                                const float *__restrict__ b2, float *__restrict__ packed, int I, int F, int skip, int NB2, int NBg, int NGg,
                                int halves, int GS)
{
    const int Ie = (I + 1) & ~1;
    const int Fe = skip ? F : 0;
    const int KT = (Ie + Fe + F) / 2, NG = F / 32, Ksrc = I + Fe + F;
    const int slab1 = slab_floats(KT, 2), slab2 = slab_floats(KT, NB2), NG2 = NG / NB2;
    const int ssd1 = urnn_split_slab_dwords(KT, 2), ssd2 = urnn_split_slab_dwords(KT, NB2);
    const int fsd1 = urnn_f16_slab_dwords(KT, NBg), fsd2 = urnn_f16_slab_dwords(KT, NB2);
    const int n1 = NG * slab1, nb1 = 2 * F, n2 = NG2 * slab2, nb2 = F, s1 = NG * ssd1, s2 = NG2 * ssd2, f1 = NGg * fsd1, f2 = NG2 * fsd2;
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const FusedCandLayout fu = urnn_fused_cand_layout(I, F, skip);
    const int base_fu = n1 + nb1 + n2 + nb2 + s1 + s2 + f1 + f2 + 2 * F;
    if (idx >= base_fu + (fu.ok ? fu.dw1 + fu.dw2 + 2 * F : 0)) return;
    if (idx >= base_fu) {
        // fused-reset-gate candidate (urnn_gemm.hip cand_fused_kernel): phase-1 slab, phase-2 slab, bias [b1 r | b2]
        int q = idx - base_fu;
        const int NBF = fu.NBF;
        auto w1r = [&](int blk, int kp, int l) {                 // reset-gate rows of W1
            const int k = 2 * kp + (l >> 5);
            const int ks = k < Ie ? (k < I ? k : -1) : I + (k - Ie);
            return ks >= 0 ? W1[(size_t)(F + blk * 32 + (l & 31)) * Ksrc + ks] : 0.f;
        };
        auto w2x = [&](int blk, int kp, int l) {                 // candidate rows of W2, x | e columns
            const int k = 2 * kp + (l >> 5);
            const int ks = k < Ie ? (k < I ? k : -1) : I + (k - Ie);
            return ks >= 0 ? W2[(size_t)(blk * 32 + (l & 31)) * Ksrc + ks] : 0.f;
        };
        if (q < fu.dw1) {
            const int xe = fu.nXE * (2 * NBF * 512);
            int grp, blk, rr;
            bool hgrp = q >= xe;
            if (!hgrp) {
                grp = q / (2 * NBF * 512);
                rr = q - grp * (2 * NBF * 512);
            } else {
                grp = fu.nXE + (q - xe) / (NBF * 512);
                rr = (q - xe) % (NBF * 512);
            }
            blk = rr / 512;
            const int piece = (rr - blk * 512) / 256, l = (rr & 255) >> 2, kp = 8 * grp + 2 * (rr & 3);
            const unsigned d = blk < NBF ? f16_pair(w1r(blk, kp, l), w1r(blk, kp + 1, l), piece)
                                         : f16_pair(w2x(blk - NBF, kp, l), w2x(blk - NBF, kp + 1, l), piece);
            reinterpret_cast<unsigned *>(packed)[idx] = d;
            return;
        }
        q -= fu.dw1;
        if (q < fu.dw2) {
            // W2[:, h] for the fp32 matrix instruction of phase 2, x 2^15 (the scale of the f16 products it is added to): row (rb, r) of
            // the accumulator layout = hidden channels 32 rb + row_c(r) (lane half 0) and + 4 (lane half 1), NBF candidate blocks each
            const int l = q & 63, nb = (q >> 6) % NBF, rr = (q >> 6) / NBF;
            const int rb = rr >> 4, r = rr & 15;
            const int ch = 32 * rb + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
            packed[idx] = W2[(size_t)(nb * 32 + (l & 31)) * Ksrc + (size_t)(I + Fe) + ch] * (URNN_F16_ASCALE * URNN_F16_WSCALE);
            return;
        }
        q -= fu.dw2;
        packed[idx] = q < F ? b1[F + q] : b2[q - F];
        return;
    }
    if (idx >= n1 + nb1 + n2 + nb2 + s1 + s2 + f1 + f2) {        // gate bias in the f16 grouping's packed column order
        const int n = idx - (n1 + nb1 + n2 + nb2 + s1 + s2 + f1 + f2);
        const int g = n / (NBg * 32), nb = (n >> 5) % NBg;
        const int cb = urnn_gate_cb(halves, GS, NG, g, nb);
        packed[idx] = b1[(cb / NG) * F + (cb % NG) * 32 + (n & 31)];
        return;
    }
    auto src_col = [&](int k) {   // packed row k -> source column of W1 / W2, -1: padding row
        if (k < Ie) return k < I ? k : -1;
        return I + (k - Ie);
    };
    auto gate_w = [&](int i, int kp, int c, int l) {    // group i = [z_i | r_i], block c
        const int ks = kp < KT ? src_col(2 * kp + (l >> 5)) : -1;
        return ks >= 0 ? W1[(size_t)(c * F + i * 32 + (l & 31)) * Ksrc + ks] : 0.f;
    };
    auto cand_w = [&](int g, int kp, int nb, int l) {
        const int ks = kp < KT ? src_col(2 * kp + (l >> 5)) : -1;
        return ks >= 0 ? W2[(size_t)((g * NB2 + nb) * 32 + (l & 31)) * Ksrc + ks] : 0.f;
    };
    float v = 0.f;
    if (idx < n1) {
        const int i = idx / slab1, r = idx - i * slab1;
        const int l = r & 63, row = r >> 6;
        v = gate_w(i, row / 2, row & 1, l);
    } else if (idx < n1 + nb1) {
        const int n = idx - n1;
        const int i = n / 64, c = (n >> 5) & 1;
        v = b1[c * F + i * 32 + (n & 31)];
    } else if (idx < n1 + nb1 + n2) {
        const int q = idx - n1 - nb1;
        const int g = q / slab2, r = q - g * slab2;
        const int l = r & 63, row = r >> 6;
        const int kp = row / NB2;
        v = cand_w(g, kp, row - kp * NB2, l);
    } else if (idx < n1 + nb1 + n2 + nb2) {
        v = b2[idx - n1 - nb1 - n2];
    } else {
        int q = idx - n1 - nb1 - n2 - nb2;
        int kp, nb, piece, l;
        unsigned d;
        if (q >= s1 + s2) {                       // f16 forms: gate groups, then candidate groups
            q -= s1 + s2;
            if (q < f1) {
                const int g = q / fsd1;
                f16_slot(q - g * fsd1, NBg, kp, nb, piece, l);
                const int cb = urnn_gate_cb(halves, GS, NG, g, nb);
                d = f16_pair(gate_w(cb % NG, kp, cb / NG, l), gate_w(cb % NG, kp + 1, cb / NG, l), piece);
            } else {
                q -= f1;
                const int g = q / fsd2;
                f16_slot(q - g * fsd2, NB2, kp, nb, piece, l);
                d = f16_pair(cand_w(g, kp, nb, l), cand_w(g, kp + 1, nb, l), piece);
            }
        } else if (q < s1) {
            const int i = q / ssd1;
            split_slot(q - i * ssd1, 2, kp, nb, piece, l);
            d = bf16_piece(gate_w(i, kp, nb, l), piece) | (bf16_piece(gate_w(i, kp + 1, nb, l), piece) << 16);
        } else {
            q -= s1;
            const int g = q / ssd2;
            split_slot(q - g * ssd2, NB2, kp, nb, piece, l);
            d = bf16_piece(cand_w(g, kp, nb, l), piece) | (bf16_piece(cand_w(g, kp + 1, nb, l), piece) << 16);
        }
        reinterpret_cast<unsigned *>(packed)[idx] = d;
        return;
    }
    packed[idx] = v;
}

size_t urnn_packed_gru_total(int I, int F, int skip)
{
    const int Ie = (I + 1) & ~1;
    const int KT = (Ie + (skip ? F : 0) + F) / 2;
    const int NB2 = urnn_cand_nb(F);
    const GateGroups gg = urnn_gate_groups(F, KT);
    return (size_t)(F / 32) * slab_floats(KT, 2) + 2 * F + (size_t)((F / 32) / NB2) * slab_floats(KT, NB2) + F +
           (size_t)(F / 32) * urnn_split_slab_dwords(KT, 2) + (size_t)((F / 32) / NB2) * urnn_split_slab_dwords(KT, NB2) +
           (size_t)gg.NG * urnn_f16_slab_dwords(KT, gg.NB) + (size_t)((F / 32) / NB2) * urnn_f16_slab_dwords(KT, NB2) + 2 * F +
           urnn_packed_gru_fused_floats(I, F, skip);
}

// floats appended for the fused-reset-gate candidate kernel (0 when the cell's shape has no such form)
size_t urnn_packed_gru_fused_floats(int I, int F, int skip)
{
    const FusedCandLayout fu = urnn_fused_cand_layout(I, F, skip);
    return fu.ok ? (size_t)fu.dw1 + fu.dw2 + 2 * F : 0;
}

hipError_t urnn_launch_pack_gru(const float *W1, const float *b1, const float *W2, const float *b2, float *packed, int I,
                                int F, int skip, hipStream_t st)
{
    const int Ie = (I + 1) & ~1;
    const int KT = (Ie + (skip ? F : 0) + F) / 2;
    const int NB2 = urnn_cand_nb(F);
    const GateGroups gg = urnn_gate_groups(F, KT);
    const int total = (int)urnn_packed_gru_total(I, F, skip);
    hipLaunchKernelGGL(pack_gru_kernel, dim3((total + 255) / 256), dim3(256), 0, st, W1, b1, W2, b2, packed, I, F, skip, NB2, gg.NB, gg.NG,
                       gg.halves, gg.GS);
    return hipGetLastError();
}

// Deconv: group a (output row parity) owns n-blocks nb = bb*NBC + cob (column parity bb, 32-channel block cob):
// slab[a][(kp*NB + nb)*64 + l] = w[ci = 2*kp + half][co = cob*32 + j][a][bb]; bias[a][nb][j] = bias[co].
__global__ void pack_deconv_kernel(const float *__restrict__ w, const float *__restrict__ bias, float *__restrict__ packed, int Cin,
                                   int Cout, int NBC, int KT)
{
    const int NB = 2 * NBC;
    const int slab = slab_floats(KT, NB), ssd = urnn_split_slab_dwords(KT, NB), fsd = urnn_f16_slab_dwords(KT, NB);
    const int nw = 2 * slab, Npad = 2 * NB * 32;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nw + Npad + 2 * ssd + 2 * fsd) return;
    auto wv = [&](int a, int kp, int nb, int l) {
        const int k = 2 * kp + (l >> 5);
        const int bb = nb / NBC, co = (nb - bb * NBC) * 32 + (l & 31);
        return (kp < KT && k < Cin && co < Cout) ? w[(((size_t)k * Cout + co) * 2 + a) * 2 + bb] : 0.f;
    };
    if (idx >= nw + Npad + 2 * ssd) {
        const int q = idx - nw - Npad - 2 * ssd;
        const int a = q / fsd;
        int kp, nb, piece, l;
        f16_slot(q - a * fsd, NB, kp, nb, piece, l);
        reinterpret_cast<unsigned *>(packed)[idx] = f16_pair(wv(a, kp, nb, l), wv(a, kp + 1, nb, l), piece);
        return;
    }
    if (idx < nw) {
        const int a = idx / slab, r = idx - a * slab;
        const int l = r & 63, row = r >> 6;
        const int kp = row / NB;
        packed[idx] = wv(a, kp, row - kp * NB, l);
    } else if (idx < nw + Npad) {
        const int n = idx - nw;
        const int nb = (n % (NB * 32)) / 32;
        const int co = (nb % NBC) * 32 + (n & 31);
        packed[idx] = (bias && co < Cout) ? bias[co] : 0.f;
    } else {
        const int q = idx - nw - Npad;
        const int a = q / ssd;
        int kp, nb, piece, l;
        split_slot(q - a * ssd, NB, kp, nb, piece, l);
        reinterpret_cast<unsigned *>(packed)[idx] = bf16_piece(wv(a, kp, nb, l), piece) | (bf16_piece(wv(a, kp + 1, nb, l), piece) << 16);
    }
}

hipError_t urnn_launch_pack_deconv(const float *w, const float *bias, float *packed, int Cin, int Cout, hipStream_t st)
{
    const int NBC = (Cout + 31) / 32, NB = 2 * NBC, KT = (Cin + 1) / 2;
    const int total = 2 * slab_floats(KT, NB) + 2 * NB * 32 + 2 * urnn_split_slab_dwords(KT, NB) + 2 * urnn_f16_slab_dwords(KT, NB);
    hipLaunchKernelGGL(pack_deconv_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w, bias, packed, Cin, Cout, NBC, KT);
    return hipGetLastError();
}
