// urnn_cand_fused.hip -- the fused candidate kernel of the full-resolution cells
#define URNN_TU urnn_cand_fused
#include "urnn_gemm.h"

// ------------------------------------------------------------------------------------------------------------------
// cand_fused_kernel -- the candidate GEMM with the reset gate RECOMPUTED in place (URNN_PHASE_FUSED_R; ConvRNN.py:165-180).
// The three-pass cell writes the raw reset gate (F planes) from the gate GEMM and reads it back here; with the matrix pipe at a
// fifth of its capacity the bytes are the scarcer resource, so the gate GEMM keeps only the statistics of r (and stores z), and
// this kernel multiplies W1[r rows] . [x; e; h] again next to its own W2 . [x; e] -- the same rows of x, e and h stream through
// the ring once and feed both products:
//   phase 1 (k-loop over x | e | h):  accR += W1r . [x; e; h]   (all k)        accC += W2 . [x; e]   (the x | e groups)
//   phase 2 (registers): r = sigmoid(GN(accR + b1r)) with the gate GEMM's folded statistics; r (.) h becomes the B operand of
//           accC += W2[:, h] . (r (.) h) WITHOUT a transpose: a lane's 16 accumulator rows of a 32-channel block are 2 x 8
//           channels, i.e. two ready-made 16-k groups in a permuted channel order -- the permutation is baked into the packed
//           W2[:, h] slab (urnn_elem.hip pack_gru_kernel).  h comes back from L2 / MALL (its rows have just streamed by).
//   epilogue: the candidate GEMM's (bias, centred GroupNorm partials, raw candidate planes).
// accR is bit-identical to the gate GEMM's accumulators (same pieces, same MFMA order), so r is the r whose statistics were taken.
// 64-pixel pair tiles (MAP_PAIR16), f16 x 3 arithmetic, NBF = F / 32 blocks of r and of c per wave (F = 64: 128 accumulators).
// LDS: [phase-1 slab | phase-2 slab | rings | bias r|c | (scale, shift) of r].
// ------------------------------------------------------------------------------------------------------------------
// The slot stream of cand_fused_kernel as a plain struct + two inlined functions (NOT closures: a lambda that is called from a
// second place, or that outlives one tile, made hipcc keep every captured variable on the stack -- and a buffer descriptor or an
// LDS address loaded from the stack is "divergent": each DMA became a 64-trip waterfall loop, 20x the run time).
struct CandStream {
    const float *sp1, *sp2, *cp;      // bases of segments 1, 2 and of the current one (sample of the stream's tile)
    unsigned csz, vo0, soff;          // current segment's bytes; the lane's DMA offset in the tile; byte offset of the next row pair
    int si, total, seg_left, seg_cur;
};
__device__ __forceinline__ const float *uniform_ptr(const float *q)       // wave-uniform by construction; say so
{
    const unsigned long long v = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<const float *>(((unsigned long long)hi << 32) | lo);
}
// point the stream at tile `item_` (past the end: every refill becomes a dummy into the sink slot)
__device__ __forceinline__ void cand_stream_open(CandStream &st, const ConvGemmParams &prm, int item_, int j, int lane)
{
    using R = Ring<2, MAP_QUAD16>;                            // a slot = one DMA instruction = TWO k-pairs (4 rows x 256 B)
    const int k1 = prm.segKp0[1], k2 = prm.segKp0[2], kp_begin = prm.kpBegin;
    const int s_begin = kp_begin >= k2 ? 2 : (kp_begin >= k1 ? 1 : 0);
    st.si = 0;
    st.total = item_ < prm.totalTiles ? (prm.KT - kp_begin) / 2 : 0;
    const int it = item_ < prm.totalTiles ? item_ : 0;
    const int b_ = __builtin_amdgcn_readfirstlane(it / prm.tilesPerSample);
    PixelMap<MAP_QUAD16, 2> pm_;
    pm_.init(it - b_ * prm.tilesPerSample, j, prm.P, prm.W, prm.P2, prm.W2);
    unsigned vo_[R::NV];
    R::lane_offsets(pm_, lane, (unsigned)prm.P, vo_);
    st.vo0 = vo_[0];
    const float *sp0 = prm.seg[0] + (size_t)b_ * prm.segC[0] * prm.P;
    st.sp1 = prm.seg[1] + (size_t)b_ * prm.segC[1] * prm.P;
    st.sp2 = prm.seg[2] + (size_t)b_ * prm.segC[2] * prm.P;
    st.cp = s_begin == 2 ? st.sp2 : (s_begin == 1 ? st.sp1 : sp0);
    st.csz = 4u * (unsigned)prm.segC[s_begin] * (unsigned)prm.P;
    st.soff = 8u * (unsigned)prm.P * (unsigned)(kp_begin - (s_begin == 2 ? k2 : (s_begin == 1 ? k1 : 0)));
    st.seg_left = (kp_begin >= k2 ? INT_MAX : (kp_begin >= k1 ? (k2 == INT_MAX ? INT_MAX : (k2 - kp_begin) / 2) : k1 == INT_MAX ? INT_MAX : (k1 - kp_begin) / 2));
    st.seg_cur = s_begin;
}
// issue the stream's next slot into `dst` (conv_gemm_kernel's refill_s: straight-line in the common case, a rare branch at a segment switch)
__device__ __forceinline__ void cand_stream_refill(CandStream &st, const ConvGemmParams &prm, char *dst, char *sink, int lane)
{
    using R = Ring<2, MAP_QUAD16>;
    const bool live = st.si < st.total;                       // past the end: a dummy into the sink slot keeps every vmcnt exact
    const unsigned vo_[R::NV] = {st.vo0};
    R::issue(live ? dst : sink, make_rsrc(uniform_ptr(st.cp), (unsigned)__builtin_amdgcn_readfirstlane((int)st.csz)), vo_, live ? st.soff : 0xF0000000u, lane);
    st.soff += 16u * (unsigned)prm.P;
    ++st.si;
    if (__builtin_expect(--st.seg_left == 0, 0)) {           // next K segment (x -> e -> h)
        const int k1 = prm.segKp0[1], k2 = prm.segKp0[2];
        ++st.seg_cur;
        st.cp = st.seg_cur == 1 ? st.sp1 : st.sp2;
        st.csz = 4u * (unsigned)prm.segC[st.seg_cur == 1 ? 1 : 2] * (unsigned)prm.P;
        st.soff = 0u;
        st.seg_left = (st.seg_cur == 1 && k2 != INT_MAX) ? (k2 - k1) / 2 : INT_MAX;
    }
}

template <int NBF, int D, int WPB>
__global__ __launch_bounds__(64 * WPB, WPB / 4) void cand_fused_kernel(const ConvGemmParams prm)
{
    constexpr int PB = 2, MAP = MAP_QUAD16, NB1 = 2 * NBF;
    using R = Ring<PB, MAP>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, half = lane >> 5;
    const size_t slabBytes = ((size_t)prm.fu1Dwords + (size_t)prm.fu2Dwords) * 4;
    char *ring = urnn_smem + slabBytes + wave * ((D + 1) * R::SLOT);
    char *scratch = ring + D * R::SLOT;
    float *bias = reinterpret_cast<float *>(urnn_smem + slabBytes + WPB * ((D + 1) * R::SLOT));   // [r: F | c: F]
    float *ssm = bias + NB1 * 32;                                                                 // [B][F][2] r-gate (scale, shift)
    const float *bias_h = bias + 4 * half;
    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    auto fin = [](float a, float bv) { return fmaf(a, URNN_F16_DESCALE, bv); };
    const int kp_begin = prm.kpBegin, KT = prm.KT, kH = prm.hKp0;

    stage_weights(reinterpret_cast<const float *>(prm.wfused), urnn_smem, prm.fu1Dwords + prm.fu2Dwords, wave, WPB, lane);
    if (threadIdx.x < NB1 * 32) bias[threadIdx.x] = prm.biasfu[threadIdx.x];
    {
        // GroupNorm of the gates from the gate GEMM's partials: the arithmetic of conv_gemm_kernel's EPI_CAND prologue, value for value
        const int F = prm.F, G1 = 2 * F / 32;
        for (int q = wave; q < prm.B * G1; q += WPB) {
            const int b = q / G1, grp = q - b * G1;
            const float *pp = prm.gpart + ((size_t)b * G1 + grp) * prm.gtiles * 2;
            double s1 = 0.0, s2 = 0.0;
            for (int t0 = 0; t0 < prm.gtiles; t0 += 64 * 32) {
                f32x2 v[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    const int t = t0 + u * 64 + lane;
                    v[u] = t < prm.gtiles ? *reinterpret_cast<const f32x2 *>(pp + 2 * t) : f32x2{0.f, 0.f};
                }
#pragma unroll
                for (int u = 0; u < 32; ++u) {
                    s1 += (double)v[u].x;
                    s2 += tile_x2(v[u].x, v[u].y, 32 * tile_valid(t0 + u * 64 + lane, prm.gtilePix, prm.P));
                }
            }
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

    // The slot stream of the gate GEMM -- one slot per k-pair of x | e | h, all plain (as in conv_gemm_kernel) -- runs ACROSS
    // tiles: the last eight refills of a tile (its last 16-k group) already fetch the first eight k-pairs of the wave's next tile,
    // so that their HBM round trip overlaps with phase 2 and the epilogue instead of opening the next tile.
    // development knob URNN_TUNE_CAND_STAGGER (prm.stagger units of ~1k cycles): the second wave of every SIMD starts late, so that one
    // wave's MFMA-only phase 2 meets the other's DMA-bound phase 1 instead of its phase 2
    if (wave >= 4 && prm.stagger)
        for (int i = 0; i < prm.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    CandStream st;
    st.si = 0; st.total = 0; st.seg_left = INT_MAX; st.seg_cur = 0; st.soff = 0; st.vo0 = 0; st.csz = 0;
    st.sp1 = st.sp2 = st.cp = nullptr;
    auto wrap = [](int s_) { return s_ >= D ? s_ - D : s_; };
    static_assert(D * R::KPS == 8, "the last 16-k group's refills issue exactly the next tile's first D slots (8 k-pairs), and the steps of a group bring the ring back to its first slot");
    const int item0 = blockIdx.x * WPB + wave;
    int slot = 0;

    for (int item = item0; item < prm.totalTiles; item += gridDim.x * WPB) {
        if (item == item0) {                                  // the wave's first tile opens the stream; later ones find it running
            cand_stream_open(st, prm, item0, j, lane);
            for (int i = 0; i < D; ++i) cand_stream_refill(st, prm, ring + i * R::SLOT, scratch, lane);
        }
        const int b = __builtin_amdgcn_readfirstlane(item / prm.tilesPerSample);      // (the division runs in the VALU)
        const int tile = item - b * prm.tilesPerSample;
        PixelMap<MAP, PB> pm;

        f32x16 acc[NB1][PB];                                  // blocks 0 .. NBF-1: raw reset gate, NBF .. 2 NBF-1: candidate
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nb][pb][r] = 0.f;

        float rbf[2][PB];
        unsigned bh[PB][4], bl[PB][4];
        const float asc = URNN_F16_ASCALE;
        const char *Ap = urnn_smem + lane * 16;
        wait_vmcnt<(D - 1) * R::NLOAD>();                     // (already landed for every tile but the wave's first)
        R::read(ring + slot * R::SLOT, lane, rbf[0]);
        auto mfma3 = [&](const char *ag, int nb_slab, f32x16 (&a)[PB], const unsigned (&ph)[PB][4], const unsigned (&pl)[PB][4]) __attribute__((always_inline)) {
            const f16x8 fh = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb_slab * 2 + 0) * 1024));
            const f16x8 fl = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(ag + (nb_slab * 2 + 1) * 1024));
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) a[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, as_f16x8(ph[pb]), a[pb], 0, 0, 0);   // small terms first
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) a[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(pl[pb]), a[pb], 0, 0, 0);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) a[pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(ph[pb]), a[pb], 0, 0, 0);
        };
        auto sstep = [&](const char *ag, auto q_tag, auto hg_tag) __attribute__((always_inline)) {
            constexpr int Q = decltype(q_tag)::value;
            constexpr bool HG = decltype(hg_tag)::value;
            const int nslot = wrap(slot + 1);
            if constexpr (Q & 1) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) split2_pair(rbf[0][pb], rbf[1][pb], asc, bh[pb][Q >> 1], bl[pb][Q >> 1]);
            }
            if constexpr ((Q & 1) == 0) {
                R::read(ring + slot * R::SLOT, lane, rbf[1], 1);   // the slot's second k-pair landed with its first
            } else {
                wait_vmcnt<(D - 2) * R::NLOAD>();             // the next slot has landed (or is a dummy)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot about to be refilled has left LDS (conv_gemm_kernel, hazard note)
                cand_stream_refill(st, prm, ring + slot * R::SLOT, scratch, lane);
                R::read(ring + nslot * R::SLOT, lane, rbf[0], 0);
            }
            if constexpr (Q == 7) {
                constexpr int NBC = HG ? NBF : NB1;           // hidden-state groups feed the reset gate only
#pragma unroll
                for (int nb = 0; nb < NBC; ++nb) {
                    mfma3(ag, nb, acc[nb], bh, bl);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if constexpr (Q & 1) slot = nslot;
        };
        auto group = [&](const char *ag, auto hg_tag) __attribute__((always_inline)) {
            sstep(ag, std::integral_constant<int, 0>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 1>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 2>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 3>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 4>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 5>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 6>{}, hg_tag);
            sstep(ag, std::integral_constant<int, 7>{}, hg_tag);
        };
        for (int kp = kp_begin; kp < kH; kp += 8) group(Ap + (size_t)(kp >> 3) * (NB1 * 2048), std::false_type{});
        const char *Ah = Ap + (size_t)(kH >> 3) * (NB1 * 2048);
        for (int kp = kH; kp < KT; kp += 8) {                 // (ONE call site per group flavour: a second one and hipcc stops inlining the
            if (kp + 8 == KT) cand_stream_open(st, prm, item + gridDim.x * WPB, j, lane);   // lambda -- every captured variable moves to the stack.)  The last
            group(Ah + (size_t)((kp - kH) >> 3) * (NBF * 2048), std::true_type{});   // group's refills fetch the next tile's first k-pairs
        }
        {
            int tile_e = __builtin_amdgcn_readfirstlane(tile), j_e = j;   // derive the tile's pixel map here instead of carrying it across the k-loop
            asm volatile("" : "+s"(tile_e), "+v"(j_e));
            pm.init(tile_e, j_e, prm.P, prm.W, prm.P2, prm.W2);
        }

        // ---- phase 2: r (.) h out of the accumulators, W2[:, h] . (r (.) h) ---------------------------------------------------------
        {
            const float *hbase = prm.seg[2] + ((size_t)b * prm.F + 4 * half) * prm.P;
            const float *ssb = ssm + ((size_t)b * prm.F + 4 * half) * 2;
            const float *A2f = reinterpret_cast<const float *>(urnn_smem + (size_t)prm.fu1Dwords * 4) + lane;
            // h first (every row of the tile's hidden state: 16 x NBF loads of 8 B per lane), then the sigmoids -- which do not need h
            // -- while the loads are in flight; r replaces the accumulator it came from
            float hv[NBF][16][PB];
#pragma unroll
            for (int rb = 0; rb < NBF; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) load_row<MAP, PB>(hbase + (size_t)(rb * 32 + row_c(r)) * prm.P, pm, hv[rb][r]);
            // Software pipeline over the reset-gate blocks: the sigmoids of block rb + 1 (VALU, transcendental pipe) are issued between the fp32
            // MFMAs of block rb (matrix pipe, 64 cycles each, asynchronous), row by row, instead of all sigmoids of a block in front of all
            // its MFMAs.  Same products in the same order per accumulator: identical bits.
            auto gate_row = [&](int rb, int r) __attribute__((always_inline)) {
                const f32x2 sc = *reinterpret_cast<const f32x2 *>(ssb + 2 * (rb * 32 + row_c(r)));
                const float bv = bias_h[rb * 32 + row_c(r)];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[rb][pb][r] = sigmoidf_fast(fin(acc[rb][pb][r], bv) * sc.x + sc.y);
            };
            // W2[:, h] . (r (.) h) on the fp32 matrix instruction: of the cell's products this is the one whose 16-bit form shows in a
            // long rollout, and here it costs little -- the operand is already in registers as fp32 (no split),
            // K = 2 per instruction pairs the two lane halves' rows (channels c and c + 4), the weights sit in LDS as fp32 x 2^15
            // (the accumulators' scale) in exactly that order, 64 x 64 of them
            auto mfma_row = [&](int rb, int r) __attribute__((always_inline)) {
                float v[PB];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) v[pb] = acc[rb][pb][r] * hv[rb][r][pb];
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) {
                    const float wa = A2f[((rb * 16 + r) * NBF + nb) * 64];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc[NBF + nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, v[pb], acc[NBF + nb][pb], 0, 0, 0);
                }
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) gate_row(0, r);
#pragma unroll
            for (int rb = 0; rb < NBF; ++rb) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    mfma_row(rb, r);
                    if (rb + 1 < NBF) gate_row(rb + 1, r);
                }
            }
        }

        // ---- epilogue: conv_gemm_kernel's EPI_CAND on the candidate blocks ------------------------------------------------------------
        {
            const int F = prm.F;
            const float inv_n = tile == prm.tilesPerSample - 1 ? prm.invTail : prm.invFull;
            float s1[NBF], s2[NBF];
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb) {
                s1[nb] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[(NBF + nb) * 32 + row_c(r)];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb)
                        if (pm.valid[pb]) s1[nb] += fin(acc[NBF + nb][pb][r], bv);
                }
            }
            wave_sum_n<NBF>(s1);
#pragma unroll
            for (int nb = 0; nb < NBF; ++nb) {
                const float mt = nofma(s1[nb] * inv_n);     // (rounded on its own: v - mt must not become an fma in one kernel and not in another)
                s2[nb] = 0.f;
                float *obase = prm.out0 + ((size_t)b * F + nb * 32 + 4 * half) * prm.P;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_h[(NBF + nb) * 32 + row_c(r)];
                    float *orow = obase + (size_t)row_c(r) * prm.P;
                    float v[PB];
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) {
                        v[pb] = fin(acc[NBF + nb][pb][r], bv);
                        const float dd = v[pb] - mt;
                        if (pm.valid[pb]) s2[nb] = fmaf(dd, dd, s2[nb]);
                    }
                    store_row<MAP, PB>(orow, pm, v);
                }
            }
            wave_sum_n<NBF>(s2);
            if (lane == 0) {
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) {
                    float *pp = prm.partial + (((size_t)b * (F / 32) + nb) * prm.tilesPerSample + tile) * 2;
                    pp[0] = s1[nb];
                    pp[1] = s2[nb];
                }
            }
        }
    }
}


// ---- fused candidate (cand_fused_kernel) ----------------------------------------------------------------------------------
// p: the GATE GEMM's parameter block (plain segments x | e | h, kpBegin, KT, P, F) + hKp0 = first k-pair of h, the fused slab
// (wfused, fu1Dwords, fu2Dwords, biasfu), the gate statistics to fold (gpart, gtiles, gtilePix, gcount, gn_w, gn_b, eps, ss_out,
// stat_out), out0 / partial of the candidate.  Returns the ring depth it would run with (0: does not fit / not eligible).
int urnn_cand_fused_plan(const ConvGemmParams &p, int B)
{
    if (p.F != 64 || !p.wfused || p.fu1Dwords <= 0 || p.P % 4 != 0) return 0;
    if (g_matrix_mode.load(std::memory_order_relaxed) != URNN_MATRIX_FP32 || !tune_split() || !tune_f16()) return 0;
    static const int on = (int)urnn_tune("URNN_TUNE_FUSED_R", 1);   // development knob (A/B)
    if (!on) return 0;
    if (p.KT % 8 != 0 || p.kpBegin % 8 != 0 || p.hKp0 % 8 != 0 || p.hKp0 >= p.KT || p.kpBegin > p.hKp0) return 0;
    if ((long)B * ((p.P + 63) / 64) < 1024) return 0;              // small planes keep their own kernels
    if ((p.segKp0[1] != INT_MAX && (p.segKp0[1] & 1)) || (p.segKp0[2] != INT_MAX && (p.segKp0[2] & 1))) return 0;   // 4-row slots
    using R = Ring<2, MAP_QUAD16>;
    const size_t lds = ((size_t)p.fu1Dwords + p.fu2Dwords) * 4 + (size_t)8 * (4 + 1) * R::SLOT + 4 * 32 * 4 + (size_t)B * p.F * 8;
    return lds <= LDS_PER_CU ? 4 : 0;
}

hipError_t urnn_launch_cand_fused(ConvGemmParams p, int B, hipStream_t st)
{
    const int D = urnn_cand_fused_plan(p, B);
    if (!D) return hipErrorInvalidValue;
    using R = Ring<2, MAP_QUAD16>;
    p.B = B;
    p.NG = 1;
    p.tilesPerSample = (p.P + 63) / 64;
    p.totalTiles = B * p.tilesPerSample;
    set_tile_means(p, 64);
    const size_t lds = ((size_t)p.fu1Dwords + p.fu2Dwords) * 4 + (size_t)8 * (D + 1) * R::SLOT + 4 * 32 * 4 + (size_t)B * p.F * 8;
    auto k8 = cand_fused_kernel<2, 4, 8>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = allow_big_lds(k8, LDS_PER_CU);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int grid = persistent_grid(lds, 1, p.totalTiles, 8, 2);
    static const int stag = (int)urnn_tune("URNN_TUNE_CAND_STAGGER", 0);
    p.stagger = stag;
    hipLaunchKernelGGL(k8, dim3(grid), dim3(512), lds, st, p);
    return hipGetLastError();
}

