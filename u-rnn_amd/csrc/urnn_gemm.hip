// urnn_gemm.hip -- the per-pixel contractions of U-RNN as fp32 MFMA GEMMs on gfx950.
//
// Every convolution of the network is 1x1, i.e. out[n][p] = sum_k W[n][k] * in[k][p] over the pixels p of an
// NCHW plane.  One wavefront owns a tile of 32*PB pixels x NB*32 output channels and runs v_mfma_f32_32x32x2_f32 with
//   A = packed weights  Wt[k][n]   (lane l: n = n0 + (l & 31), k = 2*kp + (l >> 5))  -> 128-B coalesced rows, L2 resident
//   B = activations     in[k][p]   (lane l: p = pixel(l & 31, pb), k = 2*kp + (l >> 5)) -> straight from HBM, one vector
//                                   load per k-pair (no LDS staging: each element is used by exactly one wave tile)
//   D[n][p] accumulates in registers; epilogues fuse bias, LeakyReLU, AvgPool, the ConvTranspose scatter, or the GroupNorm
//   partial statistics.  fp32-input MFMA is bit-exact fp32 FMA (no TF32 path exists on gfx950) and runs at the fp32 peak.
//
// Kernels here: conv_gemm_kernel (stage convs, deconvs, GRU gate GEMM) and gru_cand_kernel (candidate GEMM whose B operand
// is sigmoid(GN(r)) * h computed on the fly).
#include "urnn_common.h"
#include "urnn_kernels.h"

// ------------------------------------------------------------------------------------------------------------------
// pixel geometry: which plane offset does (lane column j, pixel block pb) of tile t address?
// ------------------------------------------------------------------------------------------------------------------
template <int MODE, int PB, bool VEC>
struct PixelMap {
    int off[PB];     // clamped (always in-bounds) offsets inside an input plane
    bool valid[PB];  // false: out of range, contributes nothing and is never stored
    int q;           // MODE_POOL: pooled output pixel index

    __device__ __forceinline__ void init(int tile, int j, int P, int W, int P2, int W2)
    {
        if constexpr (MODE == MODE_POOL) {
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
            q = 0;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                const int p = VEC ? tile * (32 * PB) + PB * j + pb : tile * (32 * PB) + 32 * pb + j;
                valid[pb] = p < P;
                off[pb] = valid[pb] ? p : 0;
            }
        }
    }
};

// B-operand fetch for one k-pair: PB floats from channel row c of the current segment.
template <int MODE, int PB, bool VEC>
__device__ __forceinline__ void load_b(const float *__restrict__ row, const PixelMap<MODE, PB, VEC> &pm, float (&b)[PB])
{
    if constexpr (VEC && MODE == MODE_POOL) {
        const f32x2 v0 = *reinterpret_cast<const f32x2 *>(row + pm.off[0]);
        const f32x2 v1 = *reinterpret_cast<const f32x2 *>(row + pm.off[2]);
        b[0] = v0.x; b[1] = v0.y; b[2] = v1.x; b[3] = v1.y;
    } else if constexpr (VEC && PB == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + pm.off[0]);
        b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w;
    } else if constexpr (VEC && PB == 2) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(row + pm.off[0]);
        b[0] = v.x; b[1] = v.y;
    } else {
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) b[pb] = row[pm.off[pb]];
    }
}

// One K segment: acc[nb][pb] += Wt[k][nb] * in[k][pb] for nb < NBA, software-pipelined one chunk (KU k-pairs) ahead.
template <int NB, int NBA, int MODE, int PB, bool VEC>
__device__ __forceinline__ void gemm_segment(f32x16 (&acc)[NB][PB], const float *__restrict__ wt /* row 0 of the segment, + n0 + j */,
                                             int ldw, const float *__restrict__ src /* channel 0 plane of this sample */,
                                             int C, int P, int half, const PixelMap<MODE, PB, VEC> &pm)
{
    const int nkp = (C + 1) / 2;
    const int nchunks = (nkp + URNN_KU - 1) / URNN_KU;
    float a_cur[URNN_KU][NBA], b_cur[URNN_KU][PB];
    float a_nxt[URNN_KU][NBA], b_nxt[URNN_KU][PB];

    auto fetch = [&](int chunk, float (&a)[URNN_KU][NBA], float (&b)[URNN_KU][PB]) {
#pragma unroll
        for (int u = 0; u < URNN_KU; ++u) {
            const int k = 2 * (chunk * URNN_KU + u) + half;       // packed (padded) row index
            const int c = k < C ? k : C - 1;                        // pad rows carry zero weights: any finite activation does
            const float *wrow = wt + (size_t)k * ldw;
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) a[u][nb] = wrow[nb * 32];
            load_b<MODE, PB, VEC>(src + (size_t)c * P, pm, b[u]);
        }
    };

    fetch(0, a_cur, b_cur);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        if (chunk + 1 < nchunks) fetch(chunk + 1, a_nxt, b_nxt);
#pragma unroll
        for (int u = 0; u < URNN_KU; ++u)
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u][nb], b_cur[u][pb], acc[nb][pb], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < URNN_KU; ++u) {
#pragma unroll
            for (int nb = 0; nb < NBA; ++nb) a_cur[u][nb] = a_nxt[u][nb];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) b_cur[u][pb] = b_nxt[u][pb];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// conv_gemm_kernel: blockDim = 64 * NW; wave w owns output columns [w*NB*32, (w+1)*NB*32) of the packed matrix for one
// pixel tile.  grid.x = B * tilesPerSample.
// ------------------------------------------------------------------------------------------------------------------
template <int NB, int PB, int MODE, bool VEC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams prm)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int b = blockIdx.x / prm.tilesPerSample;
    const int tile = blockIdx.x - b * prm.tilesPerSample;

    PixelMap<MODE, PB, VEC> pm;
    pm.init(tile, j, prm.P, prm.W, prm.P2, prm.W2);

    f32x16 acc[NB][PB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

    const int n0 = wave * (NB * 32);
    const float *wt = prm.wt + n0 + j;
    int krow = 0;
#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        const int C = prm.segC[s];
        if (C <= 0) continue;
        const int Cp = (C + URNN_KPAD - 1) / URNN_KPAD * URNN_KPAD;
        if (prm.seg[s] != nullptr) {
            const float *src = prm.seg[s] + (size_t)b * C * prm.P;
            const float *w = wt + (size_t)krow * prm.ldw;
            if (MODE == MODE_GRU1 && s == prm.hseg) {
                // the hidden-state rows feed the z / r gates only; the candidate's h part waits for r (gru_cand_kernel)
                if constexpr (NB >= 3) gemm_segment<NB, 2, MODE, PB, VEC>(acc, w, prm.ldw, src, C, prm.P, half, pm);
            } else {
                gemm_segment<NB, NB, MODE, PB, VEC>(acc, w, prm.ldw, src, C, prm.P, half, pm);
            }
        }
        krow += Cp;
    }

    const float *bias = prm.wt + (size_t)prm.Kpad * prm.ldw + n0;

    if constexpr (MODE == MODE_FLAT) {
        // out[b][n][p] = lrelu(acc + bias)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cib = mfma_row(r, half);
                const int n = n0 + nb * 32 + cib;
                if (n < prm.Cout) {
                    const float bv = bias[nb * 32 + cib];
                    float *orow = prm.out0 + ((size_t)b * prm.Cout + n) * prm.P;
                    float v[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) v[pb] = lrelu(acc[nb][pb][r] + bv, prm.slope);
                    if constexpr (VEC && PB == 4) {
                        if (pm.valid[0]) *reinterpret_cast<f32x4 *>(orow + pm.off[0]) = f32x4{v[0], v[1], v[2], v[3]};
                    } else if constexpr (VEC && PB == 2) {
                        if (pm.valid[0]) *reinterpret_cast<f32x2 *>(orow + pm.off[0]) = f32x2{v[0], v[1]};
                    } else {
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb)
                            if (pm.valid[pb]) orow[pm.off[pb]] = v[pb];
                    }
                }
            }
    } else if constexpr (MODE == MODE_POOL) {
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
    } else if constexpr (MODE == MODE_DECONV) {
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
                    if constexpr (VEC && PB == 2) {
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
    } else if constexpr (MODE == MODE_GRU1) {
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
                if constexpr (VEC && PB == 4) {
                    if (pm.valid[0]) *reinterpret_cast<f32x4 *>(orow + pm.off[0]) = f32x4{v[0], v[1], v[2], v[3]};
                } else if constexpr (VEC && PB == 2) {
                    if (pm.valid[0]) *reinterpret_cast<f32x2 *>(orow + pm.off[0]) = f32x2{v[0], v[1]};
                } else {
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        if (pm.valid[pb]) orow[pm.off[pb]] = v[pb];
                }
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
// ------------------------------------------------------------------------------------------------------------------
template <int NBF, int PB, bool VEC>
__global__ __launch_bounds__(256) void gru_cand_kernel(const GruCandParams prm)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [F][2] scale/shift of the r gate for this sample
    constexpr int F = NBF * 32;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int gtile = blockIdx.x * wpb + wave;                 // blocks never straddle samples (host rounds tiles up)
    const int b = blockIdx.x / prm.blocksPerSample;
    const int tile = (blockIdx.x - b * prm.blocksPerSample) * wpb + wave;
    (void)gtile;

    // r-gate scale/shift (channels F..2F-1 of the gate GroupNorm)
    for (int c = threadIdx.x; c < 2 * F; c += blockDim.x) smem[c] = prm.ss1[((size_t)b * 2 * F + F) * 2 + c];
    __syncthreads();
    if (tile >= prm.tilesPerSample) return;

    PixelMap<MODE_FLAT, PB, VEC> pm;
    pm.init(tile, j, prm.P, 0, 0, 0);

    float *cx = prm.cx + (size_t)b * F * prm.P;
    f32x16 acc[NBF][PB];
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

    const float *gr = prm.g1 + ((size_t)b * 2 * F + F) * prm.P;   // raw r-gate planes
    const float *hh = prm.h + (size_t)b * F * prm.P;
    const float *wt = prm.w2h + j;
    constexpr int KU2 = 2;                       // shallower pipeline than the gate GEMM: two operand streams (g, h) per k-pair
    constexpr int NCH = F / 2 / KU2;

    float a_cur[KU2][NBF], g_cur[KU2][PB], h_cur[KU2][PB];
    float a_nxt[KU2][NBF], g_nxt[KU2][PB], h_nxt[KU2][PB];
    auto fetch = [&](int chunk, float (&a)[KU2][NBF], float (&g)[KU2][PB], float (&h)[KU2][PB]) {
#pragma unroll
        for (int u = 0; u < KU2; ++u) {
            const int k = 2 * (chunk * KU2 + u) + half;
            const float *wrow = wt + (size_t)k * F;
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb) a[u][nb] = wrow[nb * 32];
            load_b<MODE_FLAT, PB, VEC>(gr + (size_t)k * prm.P, pm, g[u]);
            load_b<MODE_FLAT, PB, VEC>(hh + (size_t)k * prm.P, pm, h[u]);
        }
    };
    fetch(0, a_cur, g_cur, h_cur);
#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk + 1 < NCH) fetch(chunk + 1, a_nxt, g_nxt, h_nxt);
#pragma unroll
        for (int u = 0; u < KU2; ++u) {
            const int k = 2 * (chunk * KU2 + u) + half;
            const f32x2 st = *reinterpret_cast<const f32x2 *>(smem + 2 * k);
            float bop[PB];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) bop[pb] = sigmoidf_fast(g_cur[u][pb] * st.x + st.y) * h_cur[u][pb];
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[u][nb], bop[pb], acc[nb][pb], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < KU2; ++u) {
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb) a_cur[u][nb] = a_nxt[u][nb];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                g_cur[u][pb] = g_nxt[u][pb];
                h_cur[u][pb] = h_nxt[u][pb];
            }
        }
    }

    // epilogue: store C in place, partial sums per 32-channel group
#pragma unroll
    for (int nb = 0; nb < NBF; ++nb) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float *orow = cx + (size_t)(nb * 32 + mfma_row(r, half)) * prm.P;
            float v[PB];
            load_b<MODE_FLAT, PB, VEC>(orow, pm, v);   // x/e part of the candidate (+ bias) written by the gate GEMM
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                v[pb] += acc[nb][pb][r];
                if (pm.valid[pb]) {
                    s1 += v[pb];
                    s2 += v[pb] * v[pb];
                }
            }
            if constexpr (VEC && PB == 4) {
                if (pm.valid[0]) *reinterpret_cast<f32x4 *>(orow + pm.off[0]) = f32x4{v[0], v[1], v[2], v[3]};
            } else if constexpr (VEC && PB == 2) {
                if (pm.valid[0]) *reinterpret_cast<f32x2 *>(orow + pm.off[0]) = f32x2{v[0], v[1]};
            } else {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb)
                    if (pm.valid[pb]) orow[pm.off[pb]] = v[pb];
            }
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
template <int NB, int PB, int MODE, bool VEC>
static hipError_t launch_conv(const ConvGemmParams &p, int nblocks, int nwaves, hipStream_t st)
{
    hipLaunchKernelGGL((conv_gemm_kernel<NB, PB, MODE, VEC>), dim3(nblocks), dim3(64 * nwaves), 0, st, p);
    return hipGetLastError();
}

template <int NB, int MODE>
static hipError_t launch_conv_pb(const ConvGemmParams &p, int PB, bool vec, int nblocks, int nwaves, hipStream_t st)
{
    if (PB == 4) return vec ? launch_conv<NB, 4, MODE, true>(p, nblocks, nwaves, st) : launch_conv<NB, 4, MODE, false>(p, nblocks, nwaves, st);
    if (PB == 2) return vec ? launch_conv<NB, 2, MODE, true>(p, nblocks, nwaves, st) : launch_conv<NB, 2, MODE, false>(p, nblocks, nwaves, st);
    return launch_conv<NB, 1, MODE, false>(p, nblocks, nwaves, st);
}

// Flat 1x1 conv + LeakyReLU.  NB n-blocks per wave chosen from Cout; NW waves cover all columns.
hipError_t urnn_launch_conv_flat(ConvGemmParams p, int B, int PB, bool vec, hipStream_t st)
{
    const int nblk = (p.Cout + 31) / 32;
    int NB = nblk <= 3 ? nblk : (nblk % 3 == 0 ? 3 : (nblk % 2 == 0 ? 2 : 1));
    const int NW = nblk / NB;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    if (NB == 1) return launch_conv_pb<1, MODE_FLAT>(p, PB, vec, nblocks, NW, st);
    if (NB == 2) return launch_conv_pb<2, MODE_FLAT>(p, PB, vec, nblocks, NW, st);
    return launch_conv_pb<3, MODE_FLAT>(p, PB, vec, nblocks, NW, st);
}

hipError_t urnn_launch_conv_pool(ConvGemmParams p, int B, bool vec, hipStream_t st)
{
    const int nblk = (p.Cout + 31) / 32;
    int NB = nblk <= 3 ? nblk : (nblk % 3 == 0 ? 3 : (nblk % 2 == 0 ? 2 : 1));
    const int NW = nblk / NB;
    p.tilesPerSample = (p.P2 + 31) / 32;
    const int nblocks = B * p.tilesPerSample;
    if (NB == 1) return vec ? launch_conv<1, 4, MODE_POOL, true>(p, nblocks, NW, st) : launch_conv<1, 4, MODE_POOL, false>(p, nblocks, NW, st);
    if (NB == 2) return vec ? launch_conv<2, 4, MODE_POOL, true>(p, nblocks, NW, st) : launch_conv<2, 4, MODE_POOL, false>(p, nblocks, NW, st);
    return vec ? launch_conv<3, 4, MODE_POOL, true>(p, nblocks, NW, st) : launch_conv<3, 4, MODE_POOL, false>(p, nblocks, NW, st);
}

// Deconv: two waves (output row parity), each 2 * ceil(Cout/32) n-blocks, PB = 2 (1 for tiny planes).
hipError_t urnn_launch_deconv(ConvGemmParams p, int B, int PB, bool vec, hipStream_t st)
{
    const int nbc = (p.Cout + 31) / 32;
    if (nbc < 1 || nbc > 3) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    if (PB == 2) {
        if (nbc == 1) return vec ? launch_conv<2, 2, MODE_DECONV, true>(p, nblocks, 2, st) : launch_conv<2, 2, MODE_DECONV, false>(p, nblocks, 2, st);
        if (nbc == 2) return vec ? launch_conv<4, 2, MODE_DECONV, true>(p, nblocks, 2, st) : launch_conv<4, 2, MODE_DECONV, false>(p, nblocks, 2, st);
        return vec ? launch_conv<6, 2, MODE_DECONV, true>(p, nblocks, 2, st) : launch_conv<6, 2, MODE_DECONV, false>(p, nblocks, 2, st);
    }
    if (nbc == 1) return launch_conv<2, 1, MODE_DECONV, false>(p, nblocks, 2, st);
    if (nbc == 2) return launch_conv<4, 1, MODE_DECONV, false>(p, nblocks, 2, st);
    return launch_conv<6, 1, MODE_DECONV, false>(p, nblocks, 2, st);
}

// GRU gate GEMM: F/32 waves of [z|r|c].
hipError_t urnn_launch_gru1(ConvGemmParams p, int B, int PB, bool vec, hipStream_t st)
{
    const int NW = p.F / 32;
    if (NW < 1 || NW > 4) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    const int nblocks = B * p.tilesPerSample;
    return launch_conv_pb<3, MODE_GRU1>(p, PB, vec, nblocks, NW, st);
}

template <int NBF>
static hipError_t launch_cand_nbf(const GruCandParams &p, int PB, bool vec, int nblocks, int wpb, hipStream_t st)
{
    const size_t sh = (size_t)NBF * 32 * 2 * sizeof(float);
    if (PB == 4) {
        if (vec) hipLaunchKernelGGL((gru_cand_kernel<NBF, 4, true>), dim3(nblocks), dim3(64 * wpb), sh, st, p);
        else hipLaunchKernelGGL((gru_cand_kernel<NBF, 4, false>), dim3(nblocks), dim3(64 * wpb), sh, st, p);
    } else if (PB == 2) {
        if (vec) hipLaunchKernelGGL((gru_cand_kernel<NBF, 2, true>), dim3(nblocks), dim3(64 * wpb), sh, st, p);
        else hipLaunchKernelGGL((gru_cand_kernel<NBF, 2, false>), dim3(nblocks), dim3(64 * wpb), sh, st, p);
    } else {
        hipLaunchKernelGGL((gru_cand_kernel<NBF, 1, false>), dim3(nblocks), dim3(64 * wpb), sh, st, p);
    }
    return hipGetLastError();
}

hipError_t urnn_launch_cand(GruCandParams p, int B, int F, int PB, bool vec, hipStream_t st)
{
    const int wpb = 4;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.blocksPerSample = (p.tilesPerSample + wpb - 1) / wpb;
    const int nblocks = B * p.blocksPerSample;
    switch (F / 32) {
    case 1: return launch_cand_nbf<1>(p, PB, vec, nblocks, wpb, st);
    case 2: return launch_cand_nbf<2>(p, PB, vec, nblocks, wpb, st);
    case 3: return launch_cand_nbf<3>(p, PB, vec, nblocks, wpb, st);
    case 4: return launch_cand_nbf<4>(p, PB, vec, nblocks, wpb, st);
    default: return hipErrorInvalidValue;
    }
}
