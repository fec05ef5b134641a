// urnn_tail.hip -- the END of a ConvGRU cell fused with the layer that consumes the new state ("blend folded into its consumer",
// encoder.py:170-185 / decoder.py:150-164: every cell's output goes straight into a stage's 1x1 conv).
//
// The three-kernel cell ends with gru_blend_kernel -- read z, c, h; write h' -- and the next launch, a stage conv, reads h' again:
// one F-plane pass through HBM and one launch whose only purpose is to hand the state from one kernel to the next.  Here a block
// (persistent, two per CU) owns 128 pixels of one sample and ALL channels at a time:
//   phase 1  (gru_blend_kernel's arithmetic, value for value): the candidate's GroupNorm is finalised in the prologue in the blend's own
//            summation order, every lane blends the 16 channels of one 16-k group at one pixel, stores h' -- and leaves it, split into
//            the f16 pieces of the forward GEMMs, in LDS as ready-made MFMA B fragments ([16-k group][pixel block][piece][lane] x 16 B:
//            the layout of urnn_small.hip, one ds_write_b128 per row parity and piece);
//   phase 2  the consumer's 1x1 conv as a tiny activation-stationary GEMM (F / 16 groups x 3 MFMAs per 32 x 32 tile; weight pieces
//            straight from the packed f16 slab, L2-resident), one wave per (32-channel block, 32-pixel block);
//   epilogue TAIL_POOL: LeakyReLU + AvgPool2 (the block's four pixel blocks ARE the four corners of 32 pooled pixels; summed in
//            conv_gemm_kernel<EPI_POOL>'s order) -> the pooled stage output;  TAIL_FLAT: LeakyReLU -> the decoder's last feature map,
//            and -- when the head follows -- the statistics of its first LayerNorm (head_k1: u0 = Ws . f), so that pass of the head
//            over the feature map disappears as well.
// Same pieces, same MFMA order as the stage conv's kernel: the conv output is what conv_gemm_kernel would have produced from h'.
#include "urnn_common.h"
#include "urnn_kernels.h"

#include <limits.h>
#include <stdlib.h>

extern __shared__ __attribute__((aligned(16))) char urnn_tail_smem[];

enum { TAIL_POOL = 0, TAIL_FLAT = 1 };

// KGT = F / 16: 4 -> eight waves, two blocks per CU (<= 128 registers); 6 -> twelve waves, one block per CU
template <int MODE, int KGT>
__global__ __launch_bounds__(128 * KGT, KGT == 4 ? 4 : 3) void blend_conv_kernel(const TailParams tp)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    constexpr int KG = KGT, F = 16 * KGT, G = F / 32;
    const int P = tp.P;
    const int total = tp.B * tp.blocksPerSample;

    unsigned *Bp = reinterpret_cast<unsigned *>(urnn_tail_smem);         // panel [KG][4][2][64][4] dwords; later the pooling exchange
    float *ss1z = reinterpret_cast<float *>(Bp + (size_t)KG * 4 * 2 * 256);   // [F][2] (scale, shift) of the update gate, current sample
    float *ss2 = ss1z + 2 * F;                                           // [F][2] of the candidate
    float *bias = ss2 + 2 * F;                                           // [NBO * 32]
    float *hw = bias + tp.NBO * 32;                                      // TAIL_FLAT + head: Ws [16][16]
    float *red = hw + 256;                                               // [16] block reductions
    double *dred = reinterpret_cast<double *>(red + 16);                 // [nwaves][2] the candidate statistics' per-wave sums
    u32x4 *Wl = reinterpret_cast<u32x4 *>(dred + 32);                    // the consumer conv's f16 slab, all groups (persistent block: staged once)

    const int nbo = wave >> 2, pbw = wave & 3;                           // GEMM role: (output 32-channel block, pixel block)
    const bool gemm_wave = nbo < tp.NBO;
    const int gq1 = wave >> 1, pb1 = 2 * (wave & 1) + half;              // blend role: (16-k group, pixel block); nwaves = 2 KG
    if (threadIdx.x < tp.NBO * 32) bias[threadIdx.x] = tp.bias[threadIdx.x];
    if (MODE == TAIL_FLAT && tp.head_w && threadIdx.x < 256) hw[threadIdx.x] = tp.head_w[threadIdx.x];
    for (int i = threadIdx.x; i < tp.wDwords / 4; i += blockDim.x) Wl[i] = reinterpret_cast<const u32x4 *>(tp.wf16)[i];

    // panel pixel (pixel block pb, column jj) of tile `blk` -> plane offset
    auto plane_px = [&](int blk, int pb, int jj, bool &valid) -> int {
        if constexpr (MODE == TAIL_POOL) {
            const int q = blk * 32 + jj;
            valid = q < tp.P2;
            const int qq = valid ? q : 0;
            const int y2 = qq / tp.W2, x2 = qq - y2 * tp.W2;
            return (2 * y2 + (pb >> 1)) * tp.W + 2 * x2 + (pb & 1);
        } else {
            const int p = blk * 128 + pb * 32 + jj;
            valid = p < P;
            return valid ? p : 0;
        }
    };
    // z, c, h of the 16 channels of group gq1 at this lane's pixel of tile `item` (past the end: nothing is loaded)
    float zv[16], cv[16], hv[16];
    auto request = [&](int item, bool &ok, int &px) {
        const int bb = item / tp.blocksPerSample;
        ok = false;
        px = 0;
        if (item < total) px = plane_px(item - bb * tp.blocksPerSample, pb1, j, ok);
        const float *zrow = tp.g1 + ((size_t)bb * 2 * F + 16 * gq1) * P + px;
        const float *crow = tp.cx + ((size_t)bb * F + 16 * gq1) * P + px;
        const float *hrow = tp.h + ((size_t)bb * F + 16 * gq1) * P + px;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            zv[k] = ok ? zrow[(size_t)k * P] : 0.f;
            cv[k] = ok ? crow[(size_t)k * P] : 0.f;
            hv[k] = ok ? hrow[(size_t)k * P] : 0.f;
        }
    };

    // Persistent blocks (two per CU): a block walks tiles blockIdx.x, + gridDim.x, ...  The statistics fold -- a chain of dependent L2 round
    // trips -- is paid once per block and sample instead of once per tile, and a tile's 48 operand loads per lane are requested as soon
    // as the previous tile's values have been consumed: they travel under that tile's GEMM, epilogue and block barriers.
    bool ok1;
    int px1;
    // tile walk: tp.chunk == 0: blockIdx.x, + gridDim.x, ...; else tp.chunk CONSECUTIVE tiles per block (longer contiguous runs per plane)
    const int first = tp.chunk ? blockIdx.x * tp.chunk : blockIdx.x, step = tp.chunk ? 1 : gridDim.x;
    const int last = tp.chunk ? (first + tp.chunk < total ? first + tp.chunk : total) : total;
    request(first < last ? first : total, ok1, px1);
    int cur_b = -1;
    for (int item = first; item < last; item += step) {
        const int b = item / tp.blocksPerSample, blk = item - b * tp.blocksPerSample;
        if (b != cur_b) {
            // GroupNorm of the candidate of sample b, folded exactly as gru_blend_kernel<FIN> folds it -- 256 threads stride the tiles, xor
            // butterfly per wave, waves combined as (w0 + w1) + (w2 + w3) -- with four waves of this block per 32-channel group playing the
            // blend's four (nwaves = 4 G); four tile loads in flight per lane, accumulated in tile order
            cur_b = b;
            __syncthreads();                                             // (the previous sample's tables are no longer read)
            if (threadIdx.x < F) {
                ss1z[2 * threadIdx.x] = tp.ss1[((size_t)b * 2 * F + threadIdx.x) * 2];
                ss1z[2 * threadIdx.x + 1] = tp.ss1[((size_t)b * 2 * F + threadIdx.x) * 2 + 1];
            }
            {
                const int grp = wave >> 2, w = wave & 3;
                const float *pp = tp.partial2 + ((size_t)b * G + grp) * tp.ntiles2 * 2;
                double a1 = 0.0, a2 = 0.0;
                for (int t0 = w * 64 + lane; t0 < tp.ntiles2; t0 += 4 * 256) {
                    f32x2 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = t0 + u * 256 < tp.ntiles2 ? *reinterpret_cast<const f32x2 *>(pp + 2 * (t0 + u * 256)) : f32x2{0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (t0 + u * 256 < tp.ntiles2) {
                            a1 += (double)v[u].x;
                            a2 += tile_x2(v[u].x, v[u].y, 32 * tile_valid(t0 + u * 256, tp.tile_pix2, P));
                        }
                }
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) {
                    a1 += __shfl_xor(a1, m, 64);
                    a2 += __shfl_xor(a2, m, 64);
                }
                if (lane == 0) {
                    dred[2 * wave] = a1;
                    dred[2 * wave + 1] = a2;
                }
            }
            __syncthreads();
            if ((wave & 3) == 0) {
                const int grp = wave >> 2;
                const double S1 = (dred[8 * grp] + dred[8 * grp + 2]) + (dred[8 * grp + 4] + dred[8 * grp + 6]);
                const double S2 = (dred[8 * grp + 1] + dred[8 * grp + 3]) + (dred[8 * grp + 5] + dred[8 * grp + 7]);
                const double mean = S1 / tp.count;
                double var = S2 / tp.count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
                var = var > 0.0 ? var : 0.0;
                const double rstd = 1.0 / sqrt(var + (double)tp.eps);
                if (lane < 32) {
                    const int c = grp * 32 + lane;
                    const double sc = (double)tp.gn2_w[c] * rstd;
                    const float fsc = (float)sc, fsh = (float)((double)tp.gn2_b[c] - nofma(mean * sc));
                    ss2[2 * c] = fsc;
                    ss2[2 * c + 1] = fsh;
                    if (blk == 0) {
                        if (lane == 0) flag_nonfinite(tp.status, URNN_STATUS_CAND, S1, S2);
                        tp.ss2_out[((size_t)b * F + c) * 2] = fsc;
                        tp.ss2_out[((size_t)b * F + c) * 2 + 1] = fsh;
                        if (lane == 0 && tp.stat2_out) {
                            tp.stat2_out[((size_t)b * G + grp) * 2] = (float)mean;
                            tp.stat2_out[((size_t)b * G + grp) * 2 + 1] = (float)rstd;
                        }
                    }
                }
            }
            __syncthreads();
        }

        // ---- phase 1: blend (gru_blend_kernel's arithmetic); h' -> HBM and, as f16 pieces, -> the LDS panel ----------------------------------
        {
            float *orow = tp.h_out + ((size_t)b * F + 16 * gq1) * P + px1;
            const bool okc = ok1;
            float o[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = 16 * gq1 + k;
                const float z = gate_sigmoid(zv[k], ss1z[2 * c], ss1z[2 * c + 1]);
                const float n = tanhf_fast(fmaf(cv[k], ss2[2 * c], ss2[2 * c + 1]));
                o[k] = okc ? gru_blend(z, n, hv[k]) : 0.f;
            }
            request(item + step < last ? item + step : total, ok1, px1);  // the next tile's operands travel from here on
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (okc) orow[(size_t)k * P] = o[k];
            // channel k16 of the group is k-pair k16 >> 1, row parity k16 & 1; dword d of a lane slot holds k-pairs 2d (low) and 2d + 1 (high):
            // row parity hf, dword d = the pair (o[4d + hf], o[4d + 2 + hf]); the four dwords of one parity are one 16-byte store
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                unsigned ph[4], pl[4];
#pragma unroll
                for (int d = 0; d < 4; ++d) split2_pair(o[4 * d + hf], o[4 * d + 2 + hf], URNN_F16_ASCALE, ph[d], pl[d]);
                u32x4 *dst = reinterpret_cast<u32x4 *>(Bp) + (((size_t)gq1 * 4 + pb1) * 2) * 64 + (j + 32 * hf);
                dst[0] = u32x4{ph[0], ph[1], ph[2], ph[3]};
                dst[64] = u32x4{pl[0], pl[1], pl[2], pl[3]};
            }
        }
        __syncthreads();

        // ---- phase 2: the consumer's 1x1 conv on the panel ------------------------------------------------------------------------------------
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (gemm_wave) {
            const u32x4 *Bw = reinterpret_cast<const u32x4 *>(Bp) + (size_t)pbw * 2 * 64 + lane;
            const int g = nbo / tp.NBc, nb = nbo - g * tp.NBc;
            const u32x4 *Aw = Wl + (size_t)g * (tp.fDwords / 4) + lane;
#pragma unroll
            for (int gq = 0; gq < KG; ++gq) {
                    const u32x4 bh = Bw[(size_t)gq * 4 * 2 * 64], bl = Bw[(size_t)gq * 4 * 2 * 64 + 64];
                    const u32x4 wh = Aw[((size_t)(gq * tp.NBc + nb) * 2 + 0) * 64], wl = Aw[((size_t)(gq * tp.NBc + nb) * 2 + 1) * 64];
                    const f16x8 fwh = __builtin_bit_cast(f16x8, wh), fwl = __builtin_bit_cast(f16x8, wl);
                    const f16x8 fxh = __builtin_bit_cast(f16x8, bh), fxl = __builtin_bit_cast(f16x8, bl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl, fxh, acc, 0, 0, 0);           // small terms first (conv_gemm_kernel's order)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxh, acc, 0, 0, 0);
                }
        }
        auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
        const float *bias_h = bias + nbo * 32 + 4 * half;

        if constexpr (MODE == TAIL_POOL) {
            // LeakyReLU, then the four pixel blocks of a lane column are the four corners of one pooled pixel: exchange through LDS (the
            // panel is dead), summed in pixel-block order 0..3 like conv_gemm_kernel<EPI_POOL>
            __syncthreads();
            float *xb = reinterpret_cast<float *>(Bp);                    // [NBO][4 pb][16 r][64 lanes]
            if (gemm_wave) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    xb[(((size_t)nbo * 4 + pbw) * 16 + r) * 64 + lane] = lrelu(fmaf(acc[r], URNN_F16_DESCALE, bias_h[row_c(r)]), tp.slope);
            }
            __syncthreads();
            if (gemm_wave) {
                const int q = blk * 32 + j;
                if (q < tp.P2) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = pbw * 4 + rr;                       // this wave sums rows 4 pbw .. 4 pbw + 3 of block nbo
                        const int n = nbo * 32 + row_c(r) + 4 * half;
                        float s = 0.f;
#pragma unroll
                        for (int pb = 0; pb < 4; ++pb) s += xb[(((size_t)nbo * 4 + pb) * 16 + r) * 64 + lane];
                        if (n < tp.Cout) tp.out[((size_t)b * tp.Cout + n) * tp.P2 + q] = 0.25f * s;
                    }
                }
            }
        } else {
            // LeakyReLU -> the feature map; with a head behind it, the statistics of its first LayerNorm: u0 = Ws . f over 16 channels
            bool ok = false;
            int px = 0;
            if (gemm_wave) px = plane_px(blk, pbw, j, ok);
            float f8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {                                 // rows 0..7 of a lane are channels {0-3, 8-11} + 4 half: the 16 real ones
                f8[r] = lrelu(fmaf(acc[r], URNN_F16_DESCALE, bias_h[row_c(r)]), tp.slope);
                const int n = row_c(r) + 4 * half;
                if (gemm_wave && ok && n < tp.Cout) tp.out[((size_t)b * tp.Cout + n) * P + px] = f8[r];
            }
            if (tp.head_w) {
                // all 16 channels of the pixel in channel order: this lane's eight and the partner lane's (lane ^ 32) eight
                float fa[16];
#pragma unroll
                for (int r = 0; r < 8; ++r) {                             // (static indices: rows 0..7 = channels {0-3, 8-11} + 4 half)
                    const float other = __shfl_xor(f8[r], 32, 64);
                    fa[row_c(r)] = half ? other : f8[r];
                    fa[row_c(r) + 4] = half ? f8[r] : other;
                }
                // this lane's eight outputs of the stem conv (head_conv's fma order), their sum and -- about the BLOCK mean -- their squares
                float u8[8];
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int n = row_c(r) + 4 * half;
                    float u = 0.f;
#pragma unroll
                    for (int c = 0; c < 16; ++c) u = fmaf(hw[n * 16 + c], fa[c], u);
                    u8[r] = u;
                    if (gemm_wave && ok) s += u;
                }
                s = wave_sum(s);
                if (lane == 0) red[wave] = s;
                __syncthreads();
                const int nvalid = tile_valid(blk, 128, P);
                const float S = (red[0] + red[1]) + (red[2] + red[3]);
                const float m = S / (16.f * (float)nvalid);
                float q = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float d = u8[r] - m;
                    if (gemm_wave && ok) q = fmaf(d, d, q);
                }
                q = wave_sum(q);
                __syncthreads();
                if (lane == 0) red[wave] = q;
                __syncthreads();
                if (threadIdx.x == 0) {
                    float *pp = tp.partial0 + ((size_t)b * tp.blocksPerSample + blk) * 2;
                    pp[0] = S;
                    pp[1] = (red[0] + red[1]) + (red[2] + red[3]);
                }
            }
        }
        __syncthreads();        // the panel / exchange buffer and the reduction words are free for the next tile
    }
}

// ---- host side --------------------------------------------------------------------------------------------------------------------------
static size_t tail_lds_bytes(int F, int NBO, int wDwords)
{
    const size_t panel = (size_t)(F / 16) * 4 * 2 * 1024, xbuf = (size_t)NBO * 4 * 16 * 64 * 4;
    return (panel > xbuf ? panel : xbuf) + (size_t)4 * F * 4 + (size_t)NBO * 128 + 1024 + 64 + 16 * 16 + (size_t)wDwords * 4;
}

// Can the end of a cell on (B, F, H, W) be fused with a 1x1 conv to Cout channels (pool: + AvgPool2)?
bool urnn_tail_ok(int B, int F, int H, int W, int Cin, int Cout, int pool)
{
    static const int on = (int)urnn_tune("URNN_TUNE_TAIL", 1);       // development knob (A/B)
    if (!on || B < 1) return false;
    const int mm = urnn_get_matrix_mode();
    if (mm != URNN_MATRIX_FP32 && mm != URNN_MATRIX_FP32_CAND) return false;          // the f16-piece arithmetic of the stage conv
    if ((F != 64 && F != 96) || Cin != F) return false;
    const int NBO = (Cout + 31) / 32;
    if (NBO < 1 || NBO * 4 > 2 * (F / 16)) return false;                               // one wave per (output block, pixel block)
    if (pool && ((H & 1) || (W & 1) || H < 2 || W < 2)) return false;                  // every pixel must belong to a pooled one
    const int NBc = urnn_conv_nb(Cout), NGc = urnn_conv_ng(Cout);
    return tail_lds_bytes(F, NBO, NGc * urnn_f16_slab_dwords((Cin + 1) / 2, NBc)) <= 150 * 1024;
}

hipError_t urnn_launch_tail(TailParams tp, int H, int pool, hipStream_t st)
{
    tp.NBO = (tp.Cout + 31) / 32;
    if (pool) {
        tp.W2 = tp.W / 2;
        tp.P2 = (H / 2) * (tp.W / 2);
        tp.blocksPerSample = (tp.P2 + 31) / 32;
    } else {
        tp.blocksPerSample = (tp.P + 127) / 128;
    }
    const size_t lds = tail_lds_bytes(tp.F, tp.NBO, tp.wDwords);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipSuccess;
        auto raise = [&](const void *k) { if (e == hipSuccess) e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); };
        raise(reinterpret_cast<const void *>(blend_conv_kernel<TAIL_POOL, 4>));
        raise(reinterpret_cast<const void *>(blend_conv_kernel<TAIL_POOL, 6>));
        raise(reinterpret_cast<const void *>(blend_conv_kernel<TAIL_FLAT, 4>));
        raise(reinterpret_cast<const void *>(blend_conv_kernel<TAIL_FLAT, 6>));
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int total = tp.B * tp.blocksPerSample;
    const int per_cu = tp.F == 64 ? 2 : 1;                                         // persistent blocks
    int nblocks = total < 256 * per_cu ? total : 256 * per_cu;
    static const int chunked = (int)urnn_tune("URNN_TUNE_TAIL_CHUNK", 1);   // development knob (A/B)
    tp.chunk = chunked ? (total + nblocks - 1) / nblocks : 0;
    if (tp.chunk) nblocks = (total + tp.chunk - 1) / tp.chunk;
    const dim3 grid(nblocks), blk(64 * 2 * (tp.F / 16));
    if (pool && tp.F == 64) hipLaunchKernelGGL((blend_conv_kernel<TAIL_POOL, 4>), grid, blk, lds, st, tp);
    else if (pool) hipLaunchKernelGGL((blend_conv_kernel<TAIL_POOL, 6>), grid, blk, lds, st, tp);
    else if (tp.F == 64) hipLaunchKernelGGL((blend_conv_kernel<TAIL_FLAT, 4>), grid, blk, lds, st, tp);
    else hipLaunchKernelGGL((blend_conv_kernel<TAIL_FLAT, 6>), grid, blk, lds, st, tp);
    return hipGetLastError();
}
