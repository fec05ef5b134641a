// urnn_cand_gated.hip -- the two-stream candidate GEMM of the half-resolution cells on 64-pixel tiles and a group-wise ring
#define URNN_TU urnn_cand_gated
#include "urnn_gemm.h"

// ------------------------------------------------------------------------------------------------------------------
// cand_gated_kernel -- C = W2 . [x; e; sigmoid(GN(r)) (.) h] + b2 with the hidden rows gated on the fly from the STORED raw reset
// gate (ConvRNN.py:165-180), for planes that conv_gemm_kernel<NB, 1, MAP_STRIDED, EPI_CAND> used to take with one 32-pixel tile
// per wave: the half-resolution cells of the 500 x 500 / 400 x 560 configs (62 500 / 56 000 pixels, F = 96: the r and c slabs of a
// fused kernel do not fit the LDS).  That kernel walks 80-144 k-pair steps per tile at ~600 cycles per step and wave -- a dword
// DMA, a counted wait, an LDS read and (gated rows) a sigmoid in a dependent chain per step -- 42 us alone for 88-120 MB and 99 us
// in the benchmarked schedule (profiles/r05_kernel_stats*.txt).  Here:
//   * 64-pixel tiles (MAP_QUAD16 geometry: a DMA instruction moves four rows x 256 B with all 64 lanes), ONE wave per SIMD
//     (4-wave blocks: ceil(P / 64) tiles cover the chip's 1 024 SIMDs once), NB * 2 * 16 accumulators;
//   * the k-loop one 16-k group at a time (cand_fused_kernel's protocol): wait for the group's slots, read its fragments into
//     registers -- a gated group: eight of the raw reset gate and eight of h --, hand the slots back to the DMA for the NEXT
//     group, then gate (eight independent sigmoid chains per pixel column), split and multiply.  The ring holds one gated group
//     (eight 1-KB slots); a plain group uses the first four;
//   * the gates' GroupNorm is folded while the first group's rows travel (fold_lane_chain; every block the same bits).
// Same pieces, same MFMA order per accumulator as the step-wise kernel: the candidate planes are bit-identical to it; the
// GroupNorm partials are per 64-pixel tile instead of per 32 (the blend's fold takes the tile size as a parameter).
// Needs: f16-piece matrix mode, every K segment starting at a multiple of eight k-pairs, P % 4 == 0, slab + rings within the LDS.
// ------------------------------------------------------------------------------------------------------------------
// The slot stream as a plain struct + inlined functions (not closures: see the note on CandStream in urnn_cand_fused.hip).
struct GatedStream {
    unsigned vo;                       // the lane's DMA offset inside a row quad (tile constant)
    const float *sx, *se, *sh, *sg;    // the tile's sample: x, e, h planes and the raw reset-gate planes
};
__device__ __forceinline__ const float *gated_uniform_ptr(const float *q)       // wave-uniform by construction; say so
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void gated_open(GatedStream &gs, const ConvGemmParams &prm, int item, int j, int lane)
{
    using R = Ring<2, MAP_QUAD16>;
    const int b_ = __builtin_amdgcn_readfirstlane(item / prm.tilesPerSample);
    PixelMap<MAP_QUAD16, 2> pm_;
    pm_.init(item - b_ * prm.tilesPerSample, j, prm.P, prm.W, prm.P2, prm.W2);
    unsigned vo_[R::NV];
    R::lane_offsets(pm_, lane, (unsigned)prm.P, vo_);
    gs.vo = vo_[0];
    gs.sx = prm.seg[0] + (size_t)b_ * prm.segC[0] * prm.P;
    gs.se = prm.seg[1] + (size_t)b_ * prm.segC[1] * prm.P;
    gs.sh = prm.seg[2] + (size_t)b_ * prm.segC[2] * prm.P;
    gs.sg = prm.gate + ((size_t)b_ * 2 * prm.F + prm.F) * prm.P;
}
// The stream in UNITS of four slots (half a ring): plain group i is unit i; gated group q is units nP + 2 q (its raw reset-gate rows) and
// nP + 2 q + 1 (its rows of h).  Unit u lives in ring half u & 1.  Two units are in flight at any time, so a plain group's rows were
// requested two groups before they are needed and a gated group's one group before.
__device__ __forceinline__ void gated_request_unit(const GatedStream &gs, const ConvGemmParams &prm, char *ring, int u, int nP, int lane)
{
    using R = Ring<2, MAP_QUAD16>;
    const unsigned qstep = 16u * (unsigned)prm.P;                      // one slot = four rows
    const unsigned vb[R::NV] = {gs.vo};
    char *half_ring = ring + (u & 1) * 4 * R::SLOT;
    // (plain selects on scalar values: a pointer picked through an if / else chain made hipcc keep the struct on the stack)
    const bool plain = u < nP;
    const int kp0 = prm.kpBegin + 8 * u;
    const bool in_e = plain && kp0 >= prm.segKp0[1];
    const int q = (u - nP) >> 1;
    const bool hrows = !plain && ((u - nP) & 1) != 0;
    const unsigned long long px = reinterpret_cast<unsigned long long>(gs.sx), pe = reinterpret_cast<unsigned long long>(gs.se),
                             ph = reinterpret_cast<unsigned long long>(gs.sh), pg = reinterpret_cast<unsigned long long>(gs.sg);
    const unsigned long long psel = plain ? (in_e ? pe : px) : (hrows ? ph : pg);
    const float *src = reinterpret_cast<const float *>(psel);
    const int chans = plain ? (in_e ? prm.segC[1] : prm.segC[0]) : (hrows ? prm.segC[2] : prm.F);
    const unsigned bytes = 4u * (unsigned)chans * (unsigned)prm.P;
    const unsigned base = plain ? 8u * (unsigned)prm.P * (unsigned)(kp0 - (in_e ? prm.segKp0[1] : 0))
                                : 64u * (unsigned)prm.P * (unsigned)q;   // (gated: eight k-pairs = sixteen rows per group)
    const rsrc_t rs = make_rsrc(gated_uniform_ptr(src), (unsigned)__builtin_amdgcn_readfirstlane((int)bytes));
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) R::issue(half_ring + sl * R::SLOT, rs, vb, base + sl * qstep, lane);
}

template <int NB, int WPB>
__global__ __launch_bounds__(64 * WPB, 1) void cand_gated_kernel(const ConvGemmParams prm)
{
    constexpr int PB = 2, MAP = MAP_QUAD16, D = 8;
    using R = Ring<PB, MAP>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    const size_t slabBytes = (size_t)prm.fDwords * 4;
    char *ring = urnn_smem + slabBytes + wave * (D * R::SLOT);
    float *bias = reinterpret_cast<float *>(urnn_smem + slabBytes + WPB * (D * R::SLOT));
    float *ssm = bias + NB * 32;                                       // [B][F][2] r-gate (scale, shift)
    const float *bias_h = bias + 4 * half;
    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto fin = [](float a, float bv) { return fmaf(a, URNN_F16_DESCALE, bv); };
    const int kp_begin = prm.kpBegin, KT = prm.KT, kH = prm.hKp0;
    const int g = blockIdx.x % prm.NG;                                 // n-group of this block (NG == 1 for F <= 96)
    const int n0 = g * (NB * 32);
    const int F = prm.F;

    stage_weights(reinterpret_cast<const float *>(prm.wf16) + (size_t)g * prm.fDwords, urnn_smem, prm.fDwords, wave, WPB, lane);
    if (threadIdx.x < NB * 32) bias[threadIdx.x] = prm.bias[n0 + threadIdx.x];

    // ---- the slot stream of a tile (GatedStream below): group q covers k-pairs [kp_begin + 8 q, +8): plain (one segment) or gated ----
    const int item0 = (blockIdx.x / prm.NG) * WPB + wave, istep = (gridDim.x / prm.NG) * WPB;
    GatedStream gs;
    gs.vo = 0; gs.sx = gs.se = gs.sh = gs.sg = nullptr;
    wait_vmcnt<0>();
    __syncthreads();                                                  // slab and bias are in LDS
    const int nP = (kH - kp_begin) >> 3, nG = (KT - kH) >> 3, U = nP + 2 * nG;      // plain groups, gated groups, units per tile
    if (item0 < prm.totalTiles) {
        gated_open(gs, prm, item0, j, lane);
        gated_request_unit(gs, prm, ring, 0, nP, lane);               // the first two units travel while the statistics are folded
        gated_request_unit(gs, prm, ring, 1, nP, lane);
    }
    {
        // GroupNorm of the gates from the gate GEMM's partials: the arithmetic of conv_gemm_kernel's EPI_CAND prologue, value for value
        const int G1 = 2 * F / 32;
        for (int q = wave; q < prm.B * G1; q += WPB) {
            const int b = q / G1, grp = q - b * G1;
            const float *pp = prm.gpart + ((size_t)b * G1 + grp) * prm.gtiles * 2;
            double s1, s2;
            fold_lane_chain<32>(pp, prm.gtiles, prm.gtilePix, 32, prm.P, lane, s1, s2);
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
    __syncthreads();                                                  // the (scale, shift) table is complete (LDS; drains nothing in flight but DMAs)

    const char *Ap = urnn_smem + lane * 16;
    const float asc = URNN_F16_ASCALE;
    for (int item = item0; item < prm.totalTiles; item += istep) {
        const int b = __builtin_amdgcn_readfirstlane(item / prm.tilesPerSample);
        const int tile = item - b * prm.tilesPerSample;
        const float *ssb = ssm + (size_t)b * 2 * F;                   // ([B][F][2])
        f32x16 acc[NB][PB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;
        unsigned bh[PB][4], bl[PB][4];
        if (item != item0) {                                          // (a wave's later tiles open their own stream: one exposed round trip per tile)
            gated_open(gs, prm, item, j, lane);
            gated_request_unit(gs, prm, ring, 0, nP, lane);
            gated_request_unit(gs, prm, ring, 1, nP, lane);
        }
        int next = 2;                                                 // the next unit to request
        for (int kp = kp_begin; kp < KT; kp += 8) {
            const bool gated = kp >= kH;                              // uniform
            float fr[8][PB];
            if (!gated) {
                const int u = (kp - kp_begin) >> 3;
                if (u + 1 < U) wait_vmcnt<4 * R::NLOAD>(); else wait_vmcnt<0>();      // unit u has landed; u + 1 may still travel
                const char *hr = ring + (u & 1) * 4 * R::SLOT;
#pragma unroll
                for (int q = 0; q < 8; ++q) R::read(hr + (q >> 1) * R::SLOT, lane, fr[q], q & 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments have left LDS: the half may be overwritten
                if (next < U) { gated_request_unit(gs, prm, ring, next, nP, lane); ++next; }
            } else {
                const int u = nP + 2 * ((kp - kH) >> 3);
                wait_vmcnt<0>();                                      // both of the group's units
                const char *rr = ring + (u & 1) * 4 * R::SLOT, *hr = ring + ((u + 1) & 1) * 4 * R::SLOT;
                float hf[8][PB];
                f32x2 st[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    R::read(rr + (q >> 1) * R::SLOT, lane, fr[q], q & 1);
                    R::read(hr + (q >> 1) * R::SLOT, lane, hf[q], q & 1);
                    st[q] = *reinterpret_cast<const f32x2 *>(ssb + 2 * (2 * (kp + q - kH) + half));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // (the next group's rows are requested before the sigmoids: their round trip overlaps this group's arithmetic)
                if (next < U) { gated_request_unit(gs, prm, ring, next, nP, lane); ++next; }
                if (next < U) { gated_request_unit(gs, prm, ring, next, nP, lane); ++next; }
#pragma unroll
                for (int q = 0; q < 8; ++q)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) fr[q][pb] = gate_sigmoid(fr[q][pb], st[q].x, st[q].y) * hf[q][pb];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) split2_pair(fr[2 * q][pb], fr[2 * q + 1][pb], asc, bh[pb][q], bl[pb][q]);
            const char *ag = Ap + (size_t)(kp >> 3) * (NB * 2 * 1024);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const f16x8 fh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * 2 + 0) * 1024));
                const f16x8 fl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb * 2 + 1) * 1024));
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, as_f16x8(bh[pb]), acc[nb][pb], 0, 0, 0);   // small terms first
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(bl[pb]), acc[nb][pb], 0, 0, 0);
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[nb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(bh[pb]), acc[nb][pb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue: conv_gemm_kernel's EPI_CAND ------------------------------------------------------------------------------------
        PixelMap<MAP, PB> pm;
        pm.init(tile, j, prm.P, prm.W, prm.P2, prm.W2);
        const float inv_n = tile == prm.tilesPerSample - 1 ? prm.invTail : prm.invFull;
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
}

// ---- host side --------------------------------------------------------------------------------------------------------------------
// p: the candidate GEMM's parameter block as urnn_launch_cand takes it.  Returns 1 when this kernel takes the launch (the caller then
// sizes the candidate's partials for 64-pixel tiles), 0 otherwise.
int urnn_cand_gated_plan(const ConvGemmParams &p, int B)
{
    static const int on = (int)urnn_tune("URNN_TUNE_CAND_GATED", 1);   // development knob (A/B)
    if (!on) return 0;
    const int mm = g_matrix_mode.load(std::memory_order_relaxed);
    if ((mm != URNN_MATRIX_FP32 && mm != URNN_MATRIX_FP32_CAND) || !tune_split() || !tune_f16()) return 0;
    if (p.candExact || p.wide || !p.wf16 || p.fDwords <= 0 || !p.gate || p.gtilePix <= 0) return 0;
    if (p.F % 32 != 0 || urnn_cand_nb(p.F) != 3 || p.F != 96) return 0;             // one group of three n-blocks (the network's half / quarter resolution)
    if (p.P % 4 != 0 || p.KT % 8 != 0 || p.kpBegin % 8 != 0 || p.hKp0 % 8 != 0 || p.hKp0 >= p.KT || p.kpBegin >= p.hKp0) return 0;
    if (p.segKp0[1] != INT_MAX && (p.segKp0[1] % 8 != 0 || p.segKp0[1] <= p.kpBegin)) return 0;
    if ((p.KT - p.hKp0) * 2 != p.F) return 0;
    const long tiles = (long)B * ((p.P + 63) / 64);
    if (tiles < 512 || (long)B * p.P >= 2L * URNN_FULL_RES_PIXELS) return 0;      // small planes keep their kernels; full resolution its own
    using R = Ring<2, MAP_QUAD16>;
    const size_t lds = (size_t)p.fDwords * 4 + (size_t)4 * 8 * R::SLOT + 3 * 128 + (size_t)B * p.F * 8;
    return lds <= LDS_PER_CU ? 1 : 0;
}

hipError_t urnn_launch_cand_gated(ConvGemmParams p, int B, hipStream_t st)
{
    if (!urnn_cand_gated_plan(p, B)) return hipErrorInvalidValue;
    using R = Ring<2, MAP_QUAD16>;
    p.B = B;
    p.NG = 1;
    p.tilesPerSample = (p.P + 63) / 64;
    p.totalTiles = B * p.tilesPerSample;
    set_tile_means(p, 64);
    const size_t lds = (size_t)p.fDwords * 4 + (size_t)4 * 8 * R::SLOT + 3 * 128 + (size_t)B * p.F * 8;
    auto k = cand_gated_kernel<3, 4>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = allow_big_lds(k, LDS_PER_CU);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int grid = persistent_grid(lds, 1, p.totalTiles, 4, 1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, st, p);
    return hipGetLastError();
}
