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

#ifdef URNN_TRACE
    if (urnn_trace_buf && lane == 0) urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + 7) * 8 + 0] = __builtin_amdgcn_s_memtime();   // kernel entry
#endif
    stage_weights(reinterpret_cast<const float *>(prm.wfused), urnn_smem, prm.fu1Dwords + prm.fu2Dwords, wave, WPB, lane);
    if (threadIdx.x < NB1 * 32) bias[threadIdx.x] = prm.biasfu[threadIdx.x];
    // hand-over words of the block: [0] folds completed (waves 4-7), [1 + w] "go" for wave 4 + w from its SIMD partner, wave w
    volatile int *flags = reinterpret_cast<volatile int *>(ssm + (size_t)prm.B * prm.F * 2);
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    wait_vmcnt<0>();
    __syncthreads();                                          // slab and bias are in LDS
    // The gates' GroupNorm is needed in phase 2 only, so the FIRST wave of every SIMD (waves 0-3) starts streaming at once, while the
    // SECOND one (waves 4-7) folds the statistics and then waits until its partner is prm.stagger 16-k groups into its first tile.
    // No block waits for a fold before its first byte moves, and the two waves of a SIMD run out of phase: one wave's phase 2 +
    // epilogue (VALU / MFMA work, no reads) meets the other's k-loop (reads, little arithmetic) instead of its phase 2.
    if (wave >= WPB / 2) {
        // GroupNorm of the gates from the gate GEMM's partials: the arithmetic of conv_gemm_kernel's EPI_CAND prologue, value for value
        const int F = prm.F, G1 = 2 * F / 32;
        for (int q = wave - WPB / 2; q < prm.B * G1; q += WPB / 2) {
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
                    // phase 2 evaluates sigmoid(GN(acc * 2^-15 + bias)) as sigmoid_of_log2arg(acc * a + b): the accumulator's scale, the bias,
                    // the norm's affine and log2(e) folded into one (a, b) per channel, in double, rounded once
                    const double L2E = 1.4426950408889634074;
                    const double shd = (double)prm.gn_b[c] - nofma(mean * sc);
                    ssm[((size_t)b * F + (c - F)) * 2] = (float)(sc * (double)URNN_F16_DESCALE * L2E);
                    ssm[((size_t)b * F + (c - F)) * 2 + 1] = (float)(((double)prm.biasfu[c - F] * sc + shd) * L2E);
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
#ifdef URNN_TRACE
    if (urnn_trace_buf && lane == 0) urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + 7) * 8 + 1] = __builtin_amdgcn_s_memtime();   // own prologue work done
#endif
    // hand-over through LDS words instead of a barrier: LDS operations of a wave execute in order, so the table is written before the count
    // moves and read after it was seen
    bool synced = wave >= WPB / 2;                            // waves 0-3: has the fold been seen complete?  (needed from phase 2 on)
    bool gone = false;                                        // waves 0-3: has the partner been released?
#define RELEASE_PARTNER() do { if (lane == 0) flags[1 + wave] = 1; gone = true; } while (0)   /* (not a closure: see the note on CandStream) */
    if (wave >= WPB / 2) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) atomicAdd(const_cast<int *>(&flags[0]), 1);
        while (flags[1 + wave - WPB / 2] == 0) __builtin_amdgcn_s_sleep(8);      // the partner's k-loop is prm.stagger groups ahead
    }
    const int gstag = prm.stagger;                            // 16-k groups of its first tile after which a wave 0-3 releases its partner
    int gdone = 0;
    if (wave < WPB / 2 && (gstag <= 0 || blockIdx.x * WPB + wave >= prm.totalTiles)) RELEASE_PARTNER();
#ifdef URNN_TRACE
    if (urnn_trace_buf && lane == 0) urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + 7) * 8 + 2] = __builtin_amdgcn_s_memtime();   // past the barrier
#endif

    // The slot stream of the gate GEMM -- one slot per k-pair of x | e | h, all plain (as in conv_gemm_kernel) -- runs ACROSS
    // tiles: the last eight refills of a tile (its last 16-k group) already fetch the first eight k-pairs of the wave's next tile,
    // so that their HBM round trip overlaps with phase 2 and the epilogue instead of opening the next tile.
#ifdef URNN_TUNING
    const int abl = prm.abl;            // 1 no phase 2, 4 no stores, 16 no phase-1 MFMAs, 32 no epilogue, 64 no k-loop
#else
    constexpr int abl = 0;
#endif
    int tr_n = 0;
    (void)tr_n;
    CandStream st;
    st.si = 0; st.total = 0; st.seg_left = INT_MAX; st.seg_cur = 0; st.soff = 0; st.vo0 = 0; st.csz = 0;
    st.sp1 = st.sp2 = st.cp = nullptr;
    static_assert(D * R::KPS == 8, "the last 16-k group's refills issue exactly the next tile's first D slots (8 k-pairs), and the steps of a group bring the ring back to its first slot");
    const int item0 = blockIdx.x * WPB + wave;

    for (int item = item0; item < prm.totalTiles; item += gridDim.x * WPB) {
        if (item == item0) {                                  // the wave's first tile opens the stream; later ones find it running
            cand_stream_open(st, prm, item0, j, lane);
            for (int i = 0; i < D; ++i) cand_stream_refill(st, prm, ring + i * R::SLOT, scratch, lane);
        }
        TRACE_STAMP(0);
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

        // The k-loop, one 16-k group at a time.  The ring holds exactly one group (D slots x 2 k-pairs = 8 k-pairs), requested one group
        // ahead: a group (1) waits for its four slots, (2) reads its eight fragments into registers, (3) hands all four slots back to
        // the DMA -- the NEXT group's rows -- and only then (4) splits and multiplies.  The requests fly during the ~1 k cycles of (4)
        // and of the SIMD's other wave's (4).  The step-wise protocol this replaces (one slot refilled every second k-pair, a counted
        // wait per step) had the look-ahead read at the end of a group wait for a slot requested six steps -- a few hundred cycles --
        // earlier: every group stalled for most of a DMA round trip (2-4 k cycles against ~1 k of arithmetic), which is where the
        // "DMA-bound phase 1" of this kernel and of the gate GEMM came from.
        unsigned bh[PB][4], bl[PB][4];
        const float asc = URNN_F16_ASCALE;
        const char *Ap = urnn_smem + lane * 16;
#ifdef URNN_TRACE
        unsigned long long tr_wait = 0;                       // cycles this tile's groups spent waiting for their slots
#endif
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
        auto group = [&](const char *ag, auto hg_tag) __attribute__((always_inline)) {
            constexpr bool HG = decltype(hg_tag)::value;
            float fr[8][PB];
#ifdef URNN_TRACE
            const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
            wait_vmcnt<0>();                                  // the group's four slots (and whatever the wave stored before them)
#ifdef URNN_TRACE
            tr_wait += __builtin_amdgcn_s_memtime() - tw0;
#endif
#pragma unroll
            for (int q = 0; q < 8; ++q) R::read(ring + (q >> 1) * R::SLOT, lane, fr[q], q & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragments have left LDS: the slots may be overwritten
#pragma unroll
            for (int sl = 0; sl < D; ++sl) cand_stream_refill(st, prm, ring + sl * R::SLOT, scratch, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) split2_pair(fr[2 * q][pb], fr[2 * q + 1][pb], asc, bh[pb][q], bl[pb][q]);
            constexpr int NBC = HG ? NBF : NB1;               // hidden-state groups feed the reset gate only
            if (!(abl & 16)) {
#pragma unroll
                for (int nb = 0; nb < NBC; ++nb) {
                    mfma3(ag, nb, acc[nb], bh, bl);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (!(abl & 64))
        for (int kp = kp_begin; kp < kH; kp += 8) {
            group(Ap + (size_t)(kp >> 3) * (NB1 * 2048), std::false_type{});
            if (!gone && ++gdone >= gstag) RELEASE_PARTNER();
        }
        const char *Ah = Ap + (size_t)(kH >> 3) * (NB1 * 2048);
        if (!(abl & 64))
        for (int kp = kH; kp < KT; kp += 8) {                 // (ONE call site per group flavour: a second one and hipcc stops inlining the
            if (kp + 8 == KT) cand_stream_open(st, prm, item + gridDim.x * WPB, j, lane);   // lambda -- every captured variable moves to the stack.)  The last
            group(Ah + (size_t)((kp - kH) >> 3) * (NBF * 2048), std::true_type{});   // group's refills fetch the next tile's first k-pairs
            if (!gone && ++gdone >= gstag) RELEASE_PARTNER();
        }
        {
            int tile_e = __builtin_amdgcn_readfirstlane(tile), j_e = j;   // derive the tile's pixel map here instead of carrying it across the k-loop
            asm volatile("" : "+s"(tile_e), "+v"(j_e));
            pm.init(tile_e, j_e, prm.P, prm.W, prm.P2, prm.W2);
        }

        if (!gone) RELEASE_PARTNER();
        if (!synced) {                                        // waves 0-3, first tile: phase 2 needs the table that waves 4-7 folded
            while (flags[0] < WPB / 2) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            synced = true;
        }
#ifdef URNN_TRACE
        if (urnn_trace_buf && lane == 0 && tr_n < 8) urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + tr_n) * 8 + 1] = urnn_trace_buf[((size_t)(blockIdx.x * WPB + wave) * 8 + tr_n) * 8 + 0] + tr_wait;
#endif
        TRACE_STAMP(2);
        // ---- phase 2: r (.) h out of the accumulators, W2[:, h] . (r (.) h) ---------------------------------------------------------
        if (!(abl & 1)) {
            const float *ssb = ssm + ((size_t)b * prm.F + 4 * half) * 2;
            const float *A2f = reinterpret_cast<const float *>(urnn_smem + (size_t)prm.fu1Dwords * 4) + lane;
            // h first (every row of the tile's hidden state: 16 x NBF loads of 8 B per lane; its rows have just streamed by: L2 / MALL)
            float hv[NBF][16][PB];
            {
                const float *hbase = prm.seg[2] + ((size_t)b * prm.F + 4 * half) * prm.P;
#pragma unroll
                for (int rb = 0; rb < NBF; ++rb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) load_row<MAP, PB>(hbase + (size_t)(rb * 32 + row_c(r)) * prm.P, pm, hv[rb][r]);
            }
            // Software pipeline over the reset-gate blocks: the sigmoids of block rb + 1 (VALU, transcendental pipe) are issued between the fp32
            // MFMAs of block rb (matrix pipe, 64 cycles each, asynchronous), row by row, instead of all sigmoids of a block in front of all
            // its MFMAs.  Same products in the same order per accumulator: identical bits.
            // Every row needs operands from LDS -- the gate's folded affine (a, b), the product's two weight fragments.  Left to the
            // compiler they are read right in front of their use (it keeps register pressure down), and a wave alone in phase 2 -- its
            // SIMD partner streaming -- then stalls on ~130 LDS round trips per tile.  They are requested PF rows ahead instead, and a
            // scheduling barrier per row keeps them there.
            #ifndef URNN_P2_PF
#define URNN_P2_PF 4
#endif
            constexpr int NROW = NBF * 16, PF = URNN_P2_PF;
            f32x2 gsc[NROW];
            float wq[NROW][NBF];
            auto gate_ops = [&](int i) __attribute__((always_inline)) {        // i = rb * 16 + r
                gsc[i] = *reinterpret_cast<const f32x2 *>(ssb + 2 * ((i >> 4) * 32 + row_c(i & 15)));
            };
            auto mfma_ops = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) wq[i][nb] = A2f[(i * NBF + nb) * 64];
            };
            auto gate_row = [&](int i) __attribute__((always_inline)) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) acc[i >> 4][pb][i & 15] = sigmoid_of_log2arg(fmaf(acc[i >> 4][pb][i & 15], gsc[i].x, gsc[i].y));
            };
            // W2[:, h] . (r (.) h) on the fp32 matrix instruction: of the cell's products this is the one whose 16-bit form shows in a
            // long rollout, and here it costs little -- the operand is already in registers as fp32 (no split),
            // K = 2 per instruction pairs the two lane halves' rows (channels c and c + 4), the weights sit in LDS as fp32 x 2^15
            // (the accumulators' scale) in exactly that order, 64 x 64 of them
            auto mfma_row = [&](int i) __attribute__((always_inline)) {
                float v[PB];
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) v[pb] = acc[i >> 4][pb][i & 15] * hv[i >> 4][i & 15][pb];
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb)
#pragma unroll
                    for (int pb = 0; pb < PB; ++pb) acc[NBF + nb][pb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wq[i][nb], v[pb], acc[NBF + nb][pb], 0, 0, 0);
            };
            TRACE_STAMP(3);
#pragma unroll
            for (int i = 0; i < PF; ++i) gate_ops(i);
#pragma unroll
            for (int i = 0; i < 16; ++i) {                            // the first block's gates: nothing to interleave them with
                if (i + PF < NROW) gate_ops(i + PF);
                if (i + PF >= 16 && i + PF - 16 < PF) mfma_ops(i + PF - 16);     // (the product's first PF rows)
                __builtin_amdgcn_sched_barrier(0);
                gate_row(i);
            }
            TRACE_STAMP(4);
#pragma unroll
            for (int i = 0; i < NROW; ++i) {
                if (i + 16 + PF < NROW) gate_ops(i + 16 + PF);
                if (i + PF < NROW) mfma_ops(i + PF);
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(i);
                if (i + 16 < NROW) gate_row(i + 16);
            }
        }

        TRACE_STAMP(5);
        // ---- epilogue: conv_gemm_kernel's EPI_CAND on the candidate blocks ------------------------------------------------------------
        if (!(abl & 32)) {
            const int F = prm.F;
            const float inv_n = tile == prm.tilesPerSample - 1 ? prm.invTail : prm.invFull;
            float s1[NBF], s2[NBF];
            bool allv = true;
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) allv = allv && pm.valid[pb];
            // (conv_gemm_kernel's epilogue: values finished in place, a select-free copy for tiles whose pixels are all valid)
            auto stats = [&](auto full_tag) __attribute__((always_inline)) {
                constexpr bool FULLT = decltype(full_tag)::value;
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) {
                    s1[nb] = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float bv = bias_h[(NBF + nb) * 32 + row_c(r)];
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb) {
                            const float v = fin(acc[NBF + nb][pb][r], bv);
                            acc[NBF + nb][pb][r] = v;
                            if (FULLT || pm.valid[pb]) s1[nb] += v;
                        }
                    }
                }
                wave_sum_n<NBF>(s1);
                TRACE_STAMP(6);
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) {
                    const float mt = nofma(s1[nb] * inv_n);     // (rounded on its own: v - mt must not become an fma in one kernel and not in another)
                    s2[nb] = 0.f;
                    float *obase = prm.out0 + ((size_t)b * F + nb * 32 + 4 * half) * prm.P;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float *orow = obase + (size_t)row_c(r) * prm.P;
                        float v[PB];
#pragma unroll
                        for (int pb = 0; pb < PB; ++pb) {
                            v[pb] = acc[NBF + nb][pb][r];
                            const float dd = v[pb] - mt;
                            if (FULLT || pm.valid[pb]) s2[nb] = fmaf(dd, dd, s2[nb]);
                        }
                        if (!(abl & 4)) store_row_nt<MAP, PB>(orow, pm, v);      // (non-temporal: urnn_gemm.h)
                    }
                }
                wave_sum_n<NBF>(s2);
            };
            if (__builtin_amdgcn_ballot_w64(!allv) == 0) stats(std::true_type{});
            else stats(std::false_type{});
            if (lane == 0) {
#pragma unroll
                for (int nb = 0; nb < NBF; ++nb) {
                    float *pp = prm.partial + (((size_t)b * (F / 32) + nb) * prm.tilesPerSample + tile) * 2;
                    pp[0] = s1[nb];
                    pp[1] = s2[nb];
                }
            }
        }
#ifdef URNN_TRACE
        TRACE_STAMP(7);
        ++tr_n;
#endif
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
    const size_t lds = ((size_t)p.fu1Dwords + p.fu2Dwords) * 4 + (size_t)8 * (4 + 1) * R::SLOT + 4 * 32 * 4 + (size_t)B * p.F * 8 + 64;
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
    const size_t lds = ((size_t)p.fu1Dwords + p.fu2Dwords) * 4 + (size_t)8 * (D + 1) * R::SLOT + 4 * 32 * 4 + (size_t)B * p.F * 8 + 64;
    auto k8 = cand_fused_kernel<2, 4, 8>;
    static bool raised = false;
    if (!raised) {
        hipError_t e = allow_big_lds(k8, LDS_PER_CU);
        if (e != hipSuccess) return e;
        raised = true;
    }
    const int grid = persistent_grid(lds, 1, p.totalTiles, 8, 2);
    p.abl = (int)urnn_tune("URNN_TUNE_ABL", 0);
    p.stagger = (int)urnn_tune("URNN_TUNE_CAND_STAG", 3);      // 16-k groups a SIMD's first wave runs ahead of its second
    hipLaunchKernelGGL(k8, dim3(grid), dim3(512), lds, st, p);
    return hipGetLastError();
}

