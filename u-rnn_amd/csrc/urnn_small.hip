// urnn_small.hip -- the gate / candidate GEMMs of the ConvGRU cells on SMALL planes (the quarter- and half-resolution stages:
// 15 625 / 62 500 pixels at 500x500), activation-stationary.
//
// conv_gemm_kernel (urnn_gemm.hip) keeps an n-group's weights in LDS and streams activations; on a small plane a wave gets a
// single 32-pixel tile and its k-loop is one serial chain of K/2 ring steps (~270 cycles each for a dozen MFMAs per 16 k):
// 20 us for 0.9 GFLOP, and every n-group re-splits the same activations.  Here the roles are swapped:
//   * a block owns 64 pixels and ALL output channels.  Its prologue loads the tile's K input rows once, forms the candidate's
//     gated rows sigmoid(GN(r)) * h once, splits everything into the three bf16 pieces once, and leaves them in LDS as
//     ready-made MFMA B fragments ([16-k group][pixel block][piece][lane] x 16 B);
//   * one wave per (32-channel block, 32-pixel block): its k-loop is K/16 steps of {3 x 16-B weight-piece loads straight from
//     the packed split slab in global memory (L2-resident, prefetched three groups ahead), 3 ds_read_b128, 6 MFMAs} -- no
//     VALU, no ring, no per-k-pair bookkeeping;
//   * epilogue per wave: bias, centred GroupNorm partials of its 32 x 32 tile (tile = 32 pixels for the consumers), stores.
// Same arithmetic as the split k-loop (three exact bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16 k, fp32 accumulate) and
// the same outputs / workspace layout, so the blend, the backward pass and the strip mode read them unchanged.
#include "urnn_common.h"
#include "urnn_kernels.h"

#include <limits.h>
#include <stdlib.h>
#ifndef URNN_SMALL_PF
#define URNN_SMALL_PF 8     // 16-k groups of weight pieces requested ahead of the MFMAs (the L2-resident slab): 4 -> 8 is +0.5 % frames/s on both schedules (profiles/r05_ab_small_pf.txt)
#endif

extern __shared__ __attribute__((aligned(16))) char urnn_small_smem[];

#ifdef URNN_TRACE
// tuning builds: [block][wave (16)][16] s_memtime stamps of the cooperative cell's phases (tools/trace_coop.py)
static __device__ unsigned long long *urnn_small_trace_buf = nullptr;
extern "C" int urnn_debug_set_trace_urnn_small(unsigned long long *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(urnn_small_trace_buf), &p, sizeof(p)); }
#define COOP_STAMP(k) do { if (urnn_small_trace_buf && lane == 0) urnn_small_trace_buf[((size_t)blockIdx.x * 16 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define COOP_STAMP(k) do { } while (0)
#endif

// GATED = 0: gate GEMM (EPI_GRU1 semantics).  GATED = 1: candidate GEMM (EPI_CAND): rows >= hKp0 are sigmoid(GN(r)) * h; the
// gates' GroupNorm is finalised in the prologue (block 0 of each sample publishes the tables, as conv_gemm_kernel does).
template <int GATED, int MODE>
__global__ __launch_bounds__(1024) void small_cell_gemm_kernel(const ConvGemmParams prm, int nblk_total, int NBG)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int blocksPerSample = (prm.P + 63) >> 6;
    const int b = blockIdx.x / blocksPerSample, blk = blockIdx.x - b * blocksPerSample;
    const int P = prm.P, F = prm.F;
    const int kg0 = prm.kpBegin >> 3, KG = (prm.KT - prm.kpBegin) >> 3;   // 16-k groups [kg0, kg0 + KG)

    constexpr int NPC = MODE == 3 ? 2 : 3;                              // pieces per operand (MODE 3: f16 hi | lo, urnn_common.h)
    unsigned *Bp = reinterpret_cast<unsigned *>(urnn_small_smem);       // [KG][2][NPC][64][4] dwords
    float *bias = reinterpret_cast<float *>(Bp + (size_t)KG * 2 * NPC * 256);
    float *ssm = bias + nblk_total * 32;                                // GATED: [F][2] r-gate (scale, shift) of sample b
    // gate GEMM, f16 form: blocks in the packed order of the grouped slab (urnn_gate_groups), bias likewise
    const bool grouped = !GATED && MODE == 3 && prm.biasf != nullptr;
    if (threadIdx.x < nblk_total * 32) bias[threadIdx.x] = (grouped ? prm.biasf : prm.bias)[threadIdx.x];

    if constexpr (GATED) {
        // GroupNorm of the gates, folded from the gate GEMM's per-tile partials in double, fixed order (as in conv_gemm_kernel):
        // one wave per 32-channel group of sample b; the r half stays in LDS, block 0 of the sample publishes everything
        const int G1 = 2 * F / 32;
        for (int grp = wave; grp < G1; grp += nwaves) {
            const float *pp = prm.gpart + ((size_t)b * G1 + grp) * prm.gtiles * 2;
            double s1, s2;
            fold_lane_chain<8>(pp, prm.gtiles, prm.gtilePix, 32, P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares)
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
                    ssm[(c - F) * 2] = fsc;
                    ssm[(c - F) * 2 + 1] = fsh;
                }
                if (blk == 0) {
                    if (lane == 0) flag_nonfinite(prm.status, URNN_STATUS_GATES, s1, s2);
                    prm.ss_out[((size_t)b * 2 * F + c) * 2] = fsc;
                    prm.ss_out[((size_t)b * 2 * F + c) * 2 + 1] = fsh;
                    if (lane == 0 && prm.stat_out) {
                        prm.stat_out[((size_t)b * G1 + grp) * 2] = (float)mean;
                        prm.stat_out[((size_t)b * G1 + grp) * 2 + 1] = (float)rstd;
                    }
                }
            }
        }
        __syncthreads();
    }

    // the wave's first weight pieces are requested BEFORE the activation prologue: their L2 / HBM round trip overlaps with it
    const int nbg = wave >> 1, pbw = wave & 1;
    const int g = nbg / NBG, nb = nbg - g * NBG;                          // n-group / block inside it, as packed
    const u32x4 *Aw = reinterpret_cast<const u32x4 *>(MODE == 3 ? prm.wf16 + (size_t)g * prm.fDwords : prm.wsplit + (size_t)g * prm.sDwords) + lane;
    auto a_ptr = [&](int gq, int piece) { return Aw + ((size_t)((kg0 + gq) * NBG + nb) * NPC + piece) * 64; };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int PF = URNN_SMALL_PF;                                     // weight pieces prefetched PF groups ahead
    u32x4 ah[PF], am[PF], al[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        const int gq = q < KG ? q : KG - 1;
        ah[q] = *a_ptr(gq, 0);
        am[q] = *a_ptr(gq, 1);
        if constexpr (MODE == 1) al[q] = *a_ptr(gq, 2);
    }
    // ---- prologue: the tile's input rows -> bf16 pieces in LDS, once for all output channels -------------------------------------
    // unit u = (group gq, dword d, row parity hf): k-pairs kp0 = 8 (kg0 + gq) + 2d and kp0 + 1, channel row hf of each; lane = pixel
    {
        const int px = blk * 64 + lane;
        const bool pix_ok = px < P;
        const int pb = lane >> 5;
        const int k1 = prm.segKp0[1], k2 = prm.segKp0[2];
        // loads only (every unit's requests go out before anything is used): a gated row returns h and leaves its raw r-gate in gv
        auto act = [&](int kp, int hf, float &gv) -> float {
            gv = 0.f;
            if (!pix_ok) return 0.f;
            if (GATED && kp >= prm.hKp0) {
                const int ch = 2 * (kp - prm.hKp0) + hf;
                gv = prm.gate[((size_t)b * 2 * F + F + ch) * P + px];
                return prm.seg[2][((size_t)b * F + ch) * P + px];
            }
            const int sg = kp >= k2 ? 2 : (kp >= k1 ? 1 : 0);
            const int ch = 2 * (kp - (sg == 2 ? k2 : (sg == 1 ? k1 : 0))) + hf;
            return ch < prm.segC[sg] ? prm.seg[sg][((size_t)b * prm.segC[sg] + ch) * P + px] : 0.f;    // pad row of an odd channel count
        };
        const int nunits = KG * 8;
        constexpr int UB = GATED ? 6 : 8;                                   // units in flight per wave (2 loads each, 4 when gated)
        for (int u0 = wave; u0 < nunits; u0 += UB * nwaves) {
            float v0[UB], v1[UB], g0[GATED ? UB : 1], g1[GATED ? UB : 1];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q * nwaves;
                if (u < nunits) {
                    const int gq = u >> 3, d = (u >> 1) & 3, hf = u & 1;
                    const int kp0 = 8 * (kg0 + gq) + 2 * d;
                    float ga, gb;
                    v0[q] = act(kp0, hf, ga);
                    v1[q] = act(kp0 + 1, hf, gb);
                    if constexpr (GATED) {
                        g0[q] = ga;
                        g1[q] = gb;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q * nwaves;
                if (u < nunits) {
                    const int gq = u >> 3, d = (u >> 1) & 3, hf = u & 1;
                    if constexpr (GATED) {
                        const int kp0 = 8 * (kg0 + gq) + 2 * d;
                        if (kp0 >= prm.hKp0) {                               // (whole 16-k groups are gated or plain: hKp0 % 8 == 0)
                            const int ch = 2 * (kp0 - prm.hKp0) + hf;
                            v0[q] *= gate_sigmoid(g0[q], ssm[2 * ch], ssm[2 * ch + 1]);
                            v1[q] *= gate_sigmoid(g1[q], ssm[2 * ch + 4], ssm[2 * ch + 5]);
                        }
                    }
                    unsigned ph, pm, pl;
                    if constexpr (MODE == 2) {
                        ph = round_pair(v0[q], v1[q]);
                        pm = pl = 0u;
                    } else if constexpr (MODE == 3) {
                        split2_pair(v0[q], v1[q], URNN_F16_ASCALE, ph, pm);
                        pl = 0u;
                    } else {
                        split_pair(v0[q], v1[q], ph, pm, pl);
                    }
                    unsigned *dst = Bp + ((((size_t)gq * 2 + pb) * NPC) * 64 + ((lane & 31) + 32 * hf)) * 4 + d;
                    dst[0] = ph;
                    dst[256] = pm;
                    if constexpr (NPC == 3) dst[512] = pl;
                }
            }
        }
    }
    __syncthreads();

    // ---- main loop: wave = (32-channel block nbg, pixel block pbw) -------------------------------------------------------------------
    const u32x4 *Bw = reinterpret_cast<const u32x4 *>(Bp) + (size_t)pbw * NPC * 64 + lane;
    for (int gq0 = 0; gq0 < KG; gq0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int gq = gq0 + q;
            if (gq < KG) {
                const u32x4 bh = Bw[(size_t)gq * 2 * NPC * 64], bm = Bw[(size_t)gq * 2 * NPC * 64 + 64];
                const bf16x8 wh = __builtin_bit_cast(bf16x8, ah[q]), wm = __builtin_bit_cast(bf16x8, am[q]);
                const bf16x8 xh = __builtin_bit_cast(bf16x8, bh);
                if constexpr (MODE == 3) {
                    const f16x8 fwh = __builtin_bit_cast(f16x8, ah[q]), fwl = __builtin_bit_cast(f16x8, am[q]);
                    const f16x8 fxh = __builtin_bit_cast(f16x8, bh), fxl = __builtin_bit_cast(f16x8, bm);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl, fxh, acc, 0, 0, 0);           // small terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxh, acc, 0, 0, 0);
                } else if constexpr (MODE == 2) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
                } else {
                    const u32x4 bl = Bw[(size_t)gq * 2 * NPC * 64 + 128];
                    const bf16x8 wl = __builtin_bit_cast(bf16x8, al[q]), xm = __builtin_bit_cast(bf16x8, bm), xl = __builtin_bit_cast(bf16x8, bl);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, acc, 0, 0, 0);       // small terms first
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xm, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, xh, acc, 0, 0, 0);
                }
                const int gn = gq + PF < KG ? gq + PF : KG - 1;           // refill this prefetch slot
                ah[q] = *a_ptr(gn, 0);
                am[q] = *a_ptr(gn, 1);
                if constexpr (MODE == 1) al[q] = *a_ptr(gn, 2);
            }
        }
    }

    // ---- epilogue: bias, centred GroupNorm partials of the 32 x 32 tile, stores -----------------------------------------------------
    const int tile = blk * 2 + pbw;                                       // 32-pixel tile index inside the sample
    const int px = tile * 32 + j;
    const bool ok = px < P;
    const int nvalid = tile_valid(tile, 32, P);
    if (nvalid <= 0) return;                                              // (whole-wave: a tile past the end of an odd plane)
    int ch0, Cout, cb = 0;                                                // first output channel of this wave's block
    if constexpr (GATED) {
        ch0 = nbg * 32;
        Cout = F;
    } else {
        // canonical block of [z_0 .. | r_0 ..]: group g = (z_g | r_g) in the fp32 / bf16 slabs, urnn_gate_cb in the grouped f16 one
        cb = grouped ? urnn_gate_cb(prm.gHalves, prm.gGS, F / 32, g, nb) : urnn_gate_cb(0, 1, F / 32, g, nb);
        ch0 = cb * 32;
        Cout = 2 * F;
    }
    const float *bias_h = bias + nbg * 32 + 4 * half;
    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto fin = [](float a, float bv) {
        if constexpr (MODE == 3) return fmaf(a, URNN_F16_DESCALE, bv);
        else return a + bv;
    };
    float s1 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (ok) s1 += fin(acc[r], bias_h[row_c(r)]);
    s1 = wave_sum(s1);
    const float mt = nofma(s1 * (nvalid == 32 ? prm.invFull : prm.invTail));      // 1 / (32 * valid pixels), from the host: no division here
    float s2 = 0.f;
    float *obase = prm.out0 + ((size_t)b * Cout + ch0 + 4 * half) * P + px;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float v = fin(acc[r], bias_h[row_c(r)]);
        const float d = v - mt;
        if (ok) {
            s2 = fmaf(d, d, s2);
            obase[(size_t)row_c(r) * P] = v;
        }
    }
    s2 = wave_sum(s2);
    if (lane == 0) {
        const int grp = GATED ? nbg : cb;
        const int G = GATED ? F / 32 : 2 * F / 32;
        float *pp = prm.partial + (((size_t)b * G + grp) * prm.tilesPerSample + tile) * 2;
        pp[0] = s1;
        pp[1] = s2;
    }
}

static int small_mode(const ConvGemmParams &p)
{
    if (urnn_get_matrix_mode() == URNN_MATRIX_BF16) return 2;
    static const int f16 = (int)urnn_tune("URNN_TUNE_F16", 1);   // development knob (A/B)
    return (f16 && p.fDwords > 0 && p.wf16) ? 3 : 1;
}

size_t urnn_small_lds_bytes(const ConvGemmParams &p, int nblk_total, int gated)
{
    const size_t KG = (size_t)(p.KT - p.kpBegin) / 8;
    return KG * 2 * (small_mode(p) == 3 ? 2 : 3) * 1024 + (size_t)nblk_total * 128 + (gated ? (size_t)p.F * 8 : 0);
}

// Eligibility mirrors split_ok() of urnn_gemm.hip (whole, aligned 16-k groups; a split slab) plus the block shape limits.
bool urnn_small_ok(const ConvGemmParams &p, int nblk_total, int gated)
{
    if (p.sDwords <= 0 || !p.wsplit) return false;
    if (urnn_get_matrix_mode() == URNN_MATRIX_FP32_MFMA) return false;      // these kernels have no fp32-MFMA form: the weights-stationary ones run
    if (p.KT % 8 != 0 || p.kpBegin % 8 != 0 || p.KT <= p.kpBegin) return false;
    if (gated) {
        const int kpe = p.hKp0 < p.KT ? p.hKp0 : p.KT;
        if (kpe % 8 != 0 || kpe >= p.KT) return false;
    }
    if (nblk_total < 1 || nblk_total * 2 > 16) return false;             // one wave per (block, pixel block): <= 1024 threads
    return urnn_small_lds_bytes(p, nblk_total, gated) <= 150 * 1024;
}

template <int GATED>
static hipError_t launch_small(const ConvGemmParams &p, int B, int nblk_total, int NBG, int mode, hipStream_t st)
{
    const size_t lds = urnn_small_lds_bytes(p, nblk_total, GATED);
    auto k1 = small_cell_gemm_kernel<GATED, 1>;
    auto k2 = small_cell_gemm_kernel<GATED, 2>;
    auto k3 = small_cell_gemm_kernel<GATED, 3>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(k3), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    ConvGemmParams q = p;
    {
        const int tail = q.P - (q.tilesPerSample - 1) * 32;
        q.invFull = 1.0f / 1024.0f;
        q.invTail = 1.0f / (32.0f * (float)(tail > 0 ? tail : 32));
    }
    const int blocks = B * ((p.P + 63) / 64);
    const dim3 grid(blocks), blk(64 * nblk_total * 2);
    if (mode == 2) hipLaunchKernelGGL(k2, grid, blk, lds, st, q, nblk_total, NBG);
    else if (mode == 3) hipLaunchKernelGGL(k3, grid, blk, lds, st, q, nblk_total, NBG);
    else hipLaunchKernelGGL(k1, grid, blk, lds, st, q, nblk_total, NBG);
    return hipGetLastError();
}

// p as for urnn_launch_gru1 / urnn_launch_cand (tilesPerSample is set here: 32-pixel tiles)
hipError_t urnn_launch_small_gates(ConvGemmParams p, int B, hipStream_t st)
{
    p.tilesPerSample = (p.P + 31) / 32;
    p.totalTiles = B * p.tilesPerSample;
    const int mode = small_mode(p);
    return launch_small<0>(p, B, p.NG * 2, mode == 3 ? p.NBf : 2, mode, st);          // blocks per group as packed (f16: grouped)
}

hipError_t urnn_launch_small_cand(ConvGemmParams p, int B, hipStream_t st)
{
    const int NB = urnn_cand_nb(p.F);
    p.B = B;
    p.NG = (p.F / 32) / NB;
    p.tilesPerSample = (p.P + 31) / 32;
    p.totalTiles = B * p.tilesPerSample;
    return launch_small<1>(p, B, p.F / 32, NB, small_mode(p), st);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// coop_cell_kernel -- a WHOLE cell of a small plane in one launch (URNN_PHASE_COOP; ConvRNN.py:111-194): gates -> grid barrier ->
// candidate -> grid barrier -> blend.  A plane of <= 16 384 pixels per launch is at most 256 blocks of 64 pixels: every block is
// resident at once (one or two per CU), so the two whole-plane GroupNorm reductions can be barriers between phases instead of
// kernel boundaries, and nothing but the partial statistics ever leaves the CU between the three phases:
//   phase A  small_cell_gemm_kernel<0, 3>'s body: the tile's input rows -> f16 pieces in LDS, gate GEMM; the raw gates stay in the
//            accumulators, their centred tile statistics go to partial1;                                        | grid barrier 1
//   phase B  every wave folds the statistics of ITS 32-channel group (the order of the GATED prologue above: identical bits);
//            update-gate waves turn their accumulators into z, reset-gate waves into r (.) h, split it into pieces and overwrite the
//            hidden-state rows of the panel (a lane's 16 accumulator rows are channels {0-3, 8-11, ...} + 4 half of its block: eight
//            dwords of the panel per piece); candidate GEMM on the reset-gate waves (same weights, pieces and MFMA order as
//            small_cell_gemm_kernel<1, 3>: identical accumulators); statistics to partial2;                      | grid barrier 2
//   phase C  candidate waves fold their group (gru_blend_kernel<FIN>'s order), take z through LDS from the partner wave, blend, store h'.
// HBM traffic = K input planes + F planes of h a second time (L2) + F planes out: the three-kernel cell moved 2F + F raw planes
// out and back and read the inputs twice.  Arithmetic and summation orders are those of the three kernels: h' is bit-identical.
// Residency: blocks <= 256 = one per CU (<= 12 waves, ~130 registers, <= 150 KB LDS).  A stream's kernels run in order, so on ONE
// kernel chain every block is resident as soon as the predecessor drains.  A rollout with TWO chains passes the flag only for
// cells of at most 128 blocks (rollout.py: two such launches fit side by side, and kernels without a grid barrier always finish
// and free their CUs).  Spins are bounded: a barrier that cannot complete in ~1 s raises a status bit
// instead of hanging the chip.
// ------------------------------------------------------------------------------------------------------------------------------------
struct CoopCellParams {
    ConvGemmParams g;            // the gate GEMM as urnn_launch_small_gates takes it (f16 slab in the grouped order, partial = partial1)
    const unsigned *cwf16;       // candidate f16 slab [NG2][KT/8][NB2][2][64][4]
    int cfDwords, cNB;           // dwords per candidate n-group, blocks per group
    const float *cbias;          // candidate bias [F]
    const float *gn2_w, *gn2_b;
    float *partial2;             // [B][F/32][tiles][2]
    float *ss2_out;              // [B][F][2] (scale, shift) of the candidate, as gru_blend_kernel<FIN> publishes them
    float *h_out;
    unsigned *bar;               // [0]: arrivals, [16]: generation (two cache lines of the workspace's status area)
    int nblocks;
    int zbufDwords;              // LDS dwords the z hand-over needs (aliases the panel)
};


__global__ __launch_bounds__(768) void coop_cell_kernel(const CoopCellParams cp, int nblk_total, int NBG)
{
    const ConvGemmParams &prm = cp.g;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nwaves = blockDim.x >> 6;
    const int j = lane & 31, half = lane >> 5;
    const int blocksPerSample = (prm.P + 63) >> 6;
    const int b = blockIdx.x / blocksPerSample, blk = blockIdx.x - b * blocksPerSample;
    const int P = prm.P, F = prm.F, G = F / 32;
    const int kg0 = prm.kpBegin >> 3, KG = (prm.KT - prm.kpBegin) >> 3;
    constexpr int NPC = 2;

    COOP_STAMP(0);
    unsigned *Bp = reinterpret_cast<unsigned *>(urnn_small_smem);       // [KG][2][2][64][4] dwords
    const int panelDw = KG * 2 * NPC * 256;
    float *zbuf = reinterpret_cast<float *>(Bp + panelDw);               // [F/32][2][16][64]: z from the update-gate waves to their partners
    float *bias = zbuf + cp.zbufDwords;
    float *cbias = bias + nblk_total * 32;                              // [F]
    float *sstab = cbias + F;                                           // [nwaves][32][2] (scale, shift) of a wave's own channels
    if (threadIdx.x < nblk_total * 32) bias[threadIdx.x] = prm.biasf[threadIdx.x];
    if (threadIdx.x < F) cbias[threadIdx.x] = cp.cbias[threadIdx.x];

    const int nbg = wave >> 1, pbw = wave & 1;
    const int g = nbg / NBG, nb = nbg - g * NBG;
    const int cb = urnn_gate_cb(prm.gHalves, prm.gGS, G, g, nb);         // canonical block of [z_0 .. z_{G-1} | r_0 .. r_{G-1}]
    const bool is_r = cb >= G;
    const int ci = is_r ? cb - G : cb;                                    // channel block of z / r / c this wave works on
    const u32x4 *Aw = reinterpret_cast<const u32x4 *>(prm.wf16 + (size_t)g * prm.fDwords) + lane;
    auto a_ptr = [&](int gq, int piece) { return Aw + ((size_t)((kg0 + gq) * NBG + nb) * NPC + piece) * 64; };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int PF = URNN_SMALL_PF;
    u32x4 ah[PF], am[PF];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
        const int gq = q < KG ? q : KG - 1;
        ah[q] = *a_ptr(gq, 0);
        am[q] = *a_ptr(gq, 1);
    }
    // ---- phase A prologue: the tile's input rows -> f16 pieces in LDS (small_cell_gemm_kernel<0, 3>) -------------------------------
    {
        const int px = blk * 64 + lane;
        const bool pix_ok = px < P;
        const int pb = lane >> 5;
        const int k1 = prm.segKp0[1], k2 = prm.segKp0[2];
        auto act = [&](int kp, int hf) -> float {
            if (!pix_ok) return 0.f;
            const int sg = kp >= k2 ? 2 : (kp >= k1 ? 1 : 0);
            const int ch = 2 * (kp - (sg == 2 ? k2 : (sg == 1 ? k1 : 0))) + hf;
            return ch < prm.segC[sg] ? prm.seg[sg][((size_t)b * prm.segC[sg] + ch) * P + px] : 0.f;
        };
        const int nunits = KG * 8;
        constexpr int UB = 8;
        for (int u0 = wave; u0 < nunits; u0 += UB * nwaves) {
            float v0[UB], v1[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q * nwaves;
                if (u < nunits) {
                    const int gq = u >> 3, d = (u >> 1) & 3, hf = u & 1;
                    const int kp0 = 8 * (kg0 + gq) + 2 * d;
                    v0[q] = act(kp0, hf);
                    v1[q] = act(kp0 + 1, hf);
                }
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int u = u0 + q * nwaves;
                if (u < nunits) {
                    const int gq = u >> 3, d = (u >> 1) & 3, hf = u & 1;
                    unsigned ph, pm;
                    split2_pair(v0[q], v1[q], URNN_F16_ASCALE, ph, pm);
                    unsigned *dst = Bp + ((((size_t)gq * 2 + pb) * NPC) * 64 + ((lane & 31) + 32 * hf)) * 4 + d;
                    dst[0] = ph;
                    dst[256] = pm;
                }
            }
        }
    }
    COOP_STAMP(1);
    __syncthreads();
    COOP_STAMP(2);

    const u32x4 *Bw = reinterpret_cast<const u32x4 *>(Bp) + (size_t)pbw * NPC * 64 + lane;
    auto mfma3 = [&](const u32x4 &wh, const u32x4 &wl, int gq) {
        const u32x4 bh = Bw[(size_t)gq * 2 * NPC * 64], bm = Bw[(size_t)gq * 2 * NPC * 64 + 64];
        const f16x8 fwh = __builtin_bit_cast(f16x8, wh), fwl = __builtin_bit_cast(f16x8, wl);
        const f16x8 fxh = __builtin_bit_cast(f16x8, bh), fxl = __builtin_bit_cast(f16x8, bm);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwl, fxh, acc, 0, 0, 0);           // small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh, fxh, acc, 0, 0, 0);
    };
    for (int gq0 = 0; gq0 < KG; gq0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int gq = gq0 + q;
            if (gq < KG) {
                mfma3(ah[q], am[q], gq);
                const int gn = gq + PF < KG ? gq + PF : KG - 1;
                ah[q] = *a_ptr(gn, 0);
                am[q] = *a_ptr(gn, 1);
            }
        }
    }

    COOP_STAMP(3);
    // ---- phase A epilogue: raw gates stay in registers, centred statistics of the 32 x 32 tile -> partial1 ---------------------------
    const int tile = blk * 2 + pbw;
    const int px = tile * 32 + j;
    const bool ok = px < P;
    const int nvalid = tile_valid(tile, 32, P);
    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto fin = [](float a, float bv) { return fmaf(a, URNN_F16_DESCALE, bv); };
    const float inv_n = nvalid == 32 ? prm.invFull : prm.invTail;
    {
        const float *bias_h = bias + nbg * 32 + 4 * half;
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = fin(acc[r], bias_h[row_c(r)]);
            if (ok) s1 += acc[r];
        }
        s1 = wave_sum(s1);
        const float mt = nofma(s1 * inv_n);
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[r] - mt;
            if (ok) s2 = fmaf(d, d, s2);
        }
        s2 = wave_sum(s2);
        if (lane == 0 && nvalid > 0)            // published across the grid barrier: one 8-byte agent-scope store (urnn_common.h coop_grid_barrier_nf)
            publish8(prm.partial + (((size_t)b * 2 * G + cb) * prm.tilesPerSample + tile) * 2, s1, s2);
    }
    const int cg = ci / cp.cNB, cnb = ci - cg * cp.cNB;
    const u32x4 *Cw = reinterpret_cast<const u32x4 *>(cp.cwf16 + (size_t)cg * cp.cfDwords) + lane;
    auto c_ptr = [&](int gq, int piece) { return Cw + ((size_t)((kg0 + gq) * cp.cNB + cnb) * NPC + piece) * 64; };
    // h at this lane's 16 (channel, pixel) positions: the reset-gate waves need it for r (.) h and again for the blend
    const float *hrow = prm.seg[prm.segKp0[2] != INT_MAX ? 2 : 1] + ((size_t)b * F + ci * 32 + 4 * half) * P + (ok ? px : 0);
    float hv[16];
    if (is_r) {
#pragma unroll
        for (int r = 0; r < 16; ++r) hv[r] = ok ? hrow[(size_t)row_c(r) * P] : 0.f;
    }

    COOP_STAMP(4);
    coop_grid_barrier_nf(cp.bar, blockIdx.x, (unsigned)cp.nblocks, prm.status);
    COOP_STAMP(5);

    // ---- phase B: GroupNorm of this wave's gate block (the GATED prologue's fold: 8 loads in flight, lane-strided, xor butterfly) --
    float *sst = sstab + wave * 64;
    {
        const float *pp = prm.partial + ((size_t)b * 2 * G + cb) * prm.tilesPerSample * 2;
        const int gtiles = prm.tilesPerSample;
        double s1, s2;
        fold_lane_chain<8, true, true>(pp, gtiles, 32, 32, P, lane, s1, s2);      // (urnn_common.h: the order every finalizer shares; agent-scope loads)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64);
            s2 += __shfl_xor(s2, m, 64);
        }
        const double count = 32.0 * (double)P;
        const double mean = s1 / count;
        double var = s2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
        var = var > 0.0 ? var : 0.0;
        const double rstd = 1.0 / sqrt(var + (double)prm.eps);
        if (lane < 32) {
            const int c = cb * 32 + lane;
            const double sc = (double)prm.gn_w[c] * rstd;
            sst[2 * lane] = (float)sc;
            sst[2 * lane + 1] = (float)((double)prm.gn_b[c] - nofma(mean * sc));
            if (blk == 0 && pbw == 0 && prm.ss_out) {                      // the tables the three-kernel cell leaves in the workspace
                prm.ss_out[((size_t)b * 2 * F + c) * 2] = sst[2 * lane];
                prm.ss_out[((size_t)b * 2 * F + c) * 2 + 1] = sst[2 * lane + 1];
            }
        }
        if (blk == 0 && pbw == 0 && lane == 0) flag_nonfinite(prm.status, URNN_STATUS_GATES, s1, s2);
    }
    COOP_STAMP(6);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the table is this wave's own: no block barrier)
    if (is_r) {                                              // the candidate's first weight pieces (L2) travel during the gating
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int gq = q < KG ? q : KG - 1;
            ah[q] = *c_ptr(gq, 0);
            am[q] = *c_ptr(gq, 1);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const f32x2 st = *reinterpret_cast<const f32x2 *>(sst + 2 * (row_c(r) + 4 * half));
        acc[r] = gate_sigmoid(acc[r], st.x, st.y);          // z (update-gate waves) / r (reset-gate waves)
    }
    if (is_r) {
        // r (.) h -> the hidden-state rows of the panel.  Accumulator row r = 4 q + i is hidden channel 32 ci + 8 q + 4 half + i:
        // k-pair 16 ci + 4 q + 2 half + (i >> 1), row parity i & 1  =>  16-k group gqH + 2 ci + (q >> 1), dword 2 (q & 1) + half,
        // (i = 0, 2) / (i = 1, 3) the low / high halves of the dword of row parity 0 / 1 (the prologue's unit layout above)
        const int gqH = (prm.hKp0 - prm.kpBegin) >> 3;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float v0 = hv[4 * q + hf] * acc[4 * q + hf], v1 = hv[4 * q + hf + 2] * acc[4 * q + hf + 2];
                unsigned ph, pm;
                split2_pair(v0, v1, URNN_F16_ASCALE, ph, pm);
                unsigned *dst = Bp + ((((size_t)(gqH + 2 * ci + (q >> 1)) * 2 + pbw) * NPC) * 64 + (j + 32 * hf)) * 4 + 2 * (q & 1) + half;
                dst[0] = ph;
                dst[256] = pm;
            }
        }
    }
    float *zb = zbuf + (size_t)(ci * 2 + pbw) * 1024 + lane;
    if (!is_r) {                                             // update-gate waves are done after handing z over
#pragma unroll
        for (int r = 0; r < 16; ++r) zb[r * 64] = acc[r];
    }
    COOP_STAMP(7);
    __syncthreads();
    COOP_STAMP(8);
    if (is_r) {
        // candidate GEMM (small_cell_gemm_kernel<1, 3>'s main loop on the panel whose hidden rows now hold r (.) h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int gq0 = 0; gq0 < KG; gq0 += PF) {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int gq = gq0 + q;
                if (gq < KG) {
                    mfma3(ah[q], am[q], gq);
                    const int gn = gq + PF < KG ? gq + PF : KG - 1;
                    ah[q] = *c_ptr(gn, 0);
                    am[q] = *c_ptr(gn, 1);
                }
            }
        }
        const float *cb_h = cbias + ci * 32 + 4 * half;
        float s1 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[r] = fin(acc[r], cb_h[row_c(r)]);
            if (ok) s1 += acc[r];
        }
        s1 = wave_sum(s1);
        const float mt = nofma(s1 * inv_n);
        float s2 = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = acc[r] - mt;
            if (ok) s2 = fmaf(d, d, s2);
        }
        s2 = wave_sum(s2);
        if (lane == 0 && nvalid > 0) publish8(cp.partial2 + (((size_t)b * G + ci) * prm.tilesPerSample + tile) * 2, s1, s2);
    }

    COOP_STAMP(9);
    coop_grid_barrier_nf(cp.bar, blockIdx.x, (unsigned)cp.nblocks, prm.status);
    COOP_STAMP(10);

    // ---- phase C: candidate GroupNorm (gru_blend_kernel<FIN>'s order), z from the partner wave's hand-over, blend ----------------------
    if (is_r) {
        // 256 threads stride the tiles, xor butterfly per wave, waves combined as (w0 + w1) + (w2 + w3): the same additions, one wave
        const float *pp = cp.partial2 + ((size_t)b * G + ci) * prm.tilesPerSample * 2;
        auto sub = [&](int w, double &o1, double &o2) {
            double a1 = 0.0, a2 = 0.0;
            for (int t = w * 64 + lane; t < prm.tilesPerSample; t += 256) {
                const f32x2 v = consume8(pp + 2 * t);
                a1 += (double)v.x;
                a2 += tile_x2(v.x, v.y, 32 * tile_valid(t, 32, P));
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                a1 += __shfl_xor(a1, m, 64);
                a2 += __shfl_xor(a2, m, 64);
            }
            o1 = a1;
            o2 = a2;
        };
        double p0, q0, p1, q1;
        sub(0, p0, q0);
        sub(1, p1, q1);
        const double S1a = p0 + p1, S2a = q0 + q1;
        sub(2, p0, q0);
        sub(3, p1, q1);
        const double S1 = S1a + (p0 + p1), S2 = S2a + (q0 + q1);
        const double count = 32.0 * (double)P;
        const double mean = S1 / count;
        double var = S2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
        var = var > 0.0 ? var : 0.0;
        const double rstd = 1.0 / sqrt(var + (double)prm.eps);
        if (lane < 32) {
            const int c = ci * 32 + lane;
            const double sc = (double)cp.gn2_w[c] * rstd;
            sst[2 * lane] = (float)sc;
            sst[2 * lane + 1] = (float)((double)cp.gn2_b[c] - nofma(mean * sc));
            if (blk == 0 && pbw == 0 && cp.ss2_out) {
                cp.ss2_out[((size_t)b * F + c) * 2] = sst[2 * lane];
                cp.ss2_out[((size_t)b * F + c) * 2 + 1] = sst[2 * lane + 1];
            }
        }
        if (blk == 0 && pbw == 0 && lane == 0) flag_nonfinite(prm.status, URNN_STATUS_CAND, S1, S2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    COOP_STAMP(11);
    if (is_r && ok) {
        float *orow = cp.h_out + ((size_t)b * F + ci * 32 + 4 * half) * P + px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const f32x2 st = *reinterpret_cast<const f32x2 *>(sst + 2 * (row_c(r) + 4 * half));
            const float z = zb[r * 64];
            const float n = tanhf_fast(fmaf(acc[r], st.x, st.y));
            orow[(size_t)row_c(r) * P] = gru_blend(z, n, hv[r]);
        }
    }
#ifdef URNN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    COOP_STAMP(12);
#endif
}

// LDS of a cooperative cell launch: panel (or the z hand-over, whichever is larger) + gate bias + candidate bias + per-wave tables
static size_t coop_lds_bytes(const ConvGemmParams &p, int nblk_total)
{
    const size_t KG = (size_t)(p.KT - p.kpBegin) / 8;
    const size_t panel = KG * 2 * 2 * 1024, zbuf = (size_t)(p.F / 32) * 2 * 4096;
    return panel + zbuf + (size_t)nblk_total * 128 + (size_t)p.F * 4 + (size_t)nblk_total * 2 * 256;
}

// Can this cell run as ONE cooperative launch?  p / c: the gate / candidate parameter blocks as gru_cell_impl builds them.
bool urnn_coop_cell_ok(const ConvGemmParams &p, const ConvGemmParams &c, int B)
{
    static const int on = (int)urnn_tune("URNN_TUNE_COOP", 1);   // development knob (A/B)
    if (!on) return false;
    const int mm = urnn_get_matrix_mode();
    if (mm != URNN_MATRIX_FP32 && mm != URNN_MATRIX_FP32_CAND) return false;           // the f16-piece arithmetic only
    if (small_mode(p) != 3 || small_mode(c) != 3 || !p.biasf) return false;
    const int nblk = 2 * p.F / 32;
    if (!urnn_small_ok(p, nblk, 0) || !urnn_small_ok(c, p.F / 32, 1)) return false;
    if (c.hKp0 % 8 != 0 || c.hKp0 >= c.KT || (c.KT - c.hKp0) * 2 != p.F) return false;
    if (nblk * 2 * 64 > 768) return false;                                               // F <= 96: twelve waves
    const long blocks = (long)B * ((p.P + 63) / 64);
    const size_t lds = coop_lds_bytes(p, nblk);
    // every block must be resident at once: one per CU.  (A caller with two kernel chains in flight passes the flag up to 128 blocks
    // only -- urnn_gru_cell_coop_blocks, include/urnn_hip.h.)
    return blocks <= (urnn_device_cus() < 256 ? urnn_device_cus() : 256) && lds <= 150 * 1024;
}

hipError_t urnn_launch_coop_cell(ConvGemmParams p, const ConvGemmParams &c, const float *gn2_w, const float *gn2_b, float *ss2_out, float *h_out,
                                 unsigned *bar, int B, hipStream_t st)
{
    p.tilesPerSample = (p.P + 31) / 32;
    p.totalTiles = B * p.tilesPerSample;
    {
        const int tail = p.P - (p.tilesPerSample - 1) * 32;
        p.invFull = 1.0f / 1024.0f;
        p.invTail = 1.0f / (32.0f * (float)(tail > 0 ? tail : 32));
    }
    const int nblk = 2 * p.F / 32;
    CoopCellParams cp;
    cp.g = p;
    cp.g.NG = p.NGf;
    cp.cwf16 = c.wf16;
    cp.cfDwords = c.fDwords;
    cp.cNB = urnn_cand_nb(p.F);
    cp.cbias = c.bias;
    cp.g.hKp0 = c.hKp0;
    cp.g.gn_w = c.gn_w;
    cp.g.gn_b = c.gn_b;
    cp.g.eps = c.eps;
    cp.gn2_w = gn2_w;
    cp.gn2_b = gn2_b;
    cp.partial2 = c.partial;
    cp.ss2_out = ss2_out;
    cp.g.ss_out = c.ss_out;
    cp.h_out = h_out;
    cp.bar = bar;
    cp.nblocks = B * ((p.P + 63) / 64);
    cp.zbufDwords = (p.F / 32) * 2 * 1024;
    const size_t lds = coop_lds_bytes(p, nblk);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(coop_cell_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    hipLaunchKernelGGL(coop_cell_kernel, dim3(cp.nblocks), dim3(64 * nblk * 2), lds, st, cp, nblk, p.NBf);
    return hipGetLastError();
}
