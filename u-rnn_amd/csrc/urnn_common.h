// urnn_common.h -- shared device helpers for the gfx950 U-RNN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// Every translation unit of the library must be compiled WITHOUT SLP vectorisation: on gfx950 the vectoriser turns pairs of scalar
// fp32 operations into v_pk_mul_f32 / v_pk_fma_f32, and next to MFMA results those gave -- about once per 10^9 instructions,
// reproducibly -- a wrong low element in 16 lanes of a wave (tools/stress_overlap.py).  The flag cannot be
// set from inside a source file, so the build says that it set it: u-rnn_amd/build_ext.py passes both.
#if !defined(URNN_NO_PACKED_F32) && !defined(URNN_ALLOW_PACKED_F32)
#error "compile liburnn_hip with -fno-slp-vectorize -DURNN_NO_PACKED_F32=1 (u-rnn_amd/build_ext.py); see DESIGN.md section 4.8"
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8-byte agent-scope (write-through / L1-bypassing) store and load of a float pair: the only accesses coop_grid_barrier_nf orders
__device__ __forceinline__ void publish8(float *p, float a, float b)
{
    const f32x2 v = {a, b};
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x2 consume8(const float *p)
{
    return __builtin_bit_cast(f32x2, __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// A value the compiler may not fold into a neighbouring operation: hipcc contracts a * b + c into an fma wherever it likes (and
// __fmul_rn / __dmul_rn are plain multiplications to it), so the SAME source line can round once in one kernel and twice in another --
// one ulp in a tile's centred sum of squares was enough to move a GroupNorm scale by a float ulp between the cooperative cell and
// the three-kernel cell.  Wherever several kernels must produce the same bits, the product is made opaque before it is added.
__device__ __forceinline__ float nofma(float v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ double nofma(double v) { asm("" : "+v"(v)); return v; }

// k-pairs (one v_mfma_f32_32x32x2_f32 consumes a k-pair) per software-pipeline chunk.  Every K segment of a
// packed weight matrix is padded to a multiple of 2*KU rows.
#ifndef URNN_KU
#define URNN_KU 4
#endif
#define URNN_KPAD (2 * URNN_KU)

static inline int urnn_round_up(int v, int m) { return (v + m - 1) / m * m; }

// Packed weights, SPLIT form (urnn_gemm.hip, bf16 x 6 k-loop): per n-group  [KT / 8 sixteen-k groups][NB][3 pieces hi|mid|lo]
// [64 lanes][4 dwords]; dword d of lane l holds the bf16 piece of W[k = 2 * (8 * group + 2d) + (l >> 5)][n] in its low half and
// of k + 2 (the next k-pair) in its high half, n = (g * NB + nb) * 32 + (l & 31): the A operand of v_mfma_f32_32x32x16_bf16 is
// one conflict-free ds_read_b128 per piece.  Dwords per n-group:
__host__ __device__ static inline int urnn_split_slab_dwords(int KT, int NB) { return ((KT + 7) / 8) * NB * 3 * 256; }

// Packed weights, F16 form (urnn_gemm.hip SPLIT = 3, the f16 x 3 k-loop): the same image with TWO f16 pieces (hi | lo) of
// W * 2^URNN_F16_WEXP per 16-k group and n-block:  [KT / 8][NB][2 pieces][64 lanes][4 dwords].  Dwords per n-group:
__host__ __device__ static inline int urnn_f16_slab_dwords(int KT, int NB) { return ((KT + 7) / 8) * NB * 2 * 256; }

// Row of the 32x32 MFMA C/D tile held in accumulator register r by a lane of the given half-wave:
// row = (r & 3) + 8 * (r >> 2) + 4 * half  (cdna_hip_programming.md section 3; col = lane & 31).
__device__ __forceinline__ int mfma_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// Sum over the 64 lanes, returned to every lane; fixed order (bit-reproducible).  The first four steps are DPP adds inside a
// 16-lane row (quad swaps, half-row mirror, row mirror: VALU speed); only the two cross-row steps go through the LDS crossbar
// (ds_bpermute, ~100 cycles of latency each).  The all-bpermute butterfly was a chain of six such round trips per sum: the GEMM
// epilogues spent a third of their time in eight of those chains per tile.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ void wave_sum_n(float (&v)[N])      // N independent sums, their steps interleaved
{
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_move<0xB1>(v[i]);      // quad_perm [1,0,3,2]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_move<0x4E>(v[i]);      // quad_perm [2,3,0,1]
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_move<0x141>(v[i]);     // row_half_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_move<0x140>(v[i]);     // row_mirror
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], 16, 64);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += __shfl_xor(v[i], 32, 64);
}
__device__ __forceinline__ float wave_sum(float v)
{
    float a[1] = {v};
    wave_sum_n<1>(a);
    return a[0];
}

// GroupNorm / LayerNorm partial statistics.  A tile (or block) of n values leaves (s1, m2) = (sum, sum of squared deviations
// from the tile's OWN mean s1 / n) in fp32; the finalizes rebuild the raw second moment in double, x2 = m2 + s1^2 / n, and
// combine in double.  Raw fp32 (sum, sum of squares) partials lose the variance to cancellation once |mean| >> std (a
// trained network's pre-norm activations: |mean| / std = 30 cost 5e-4 of rstd); centred ones keep ~2 (|mean|/std) eps.
// n == 0 marks a RAW partial (strip mode's all-reduced totals): y already is the second moment.
__device__ __forceinline__ int tile_valid(int t, int tile_pix, int P)
{
    const int left = P - t * tile_pix;
    return left < 0 ? 0 : (left > tile_pix ? tile_pix : left);
}
__device__ __forceinline__ double tile_x2(float s1, float y, int n)
{
    return n > 0 ? (double)y + (double)s1 * (double)s1 / (double)n : (double)y;
}

// One lane's share of a fold over a sample's per-tile partials pp[t] = (sum, centred second moment): tiles lane, lane + 64, ... in
// ascending order, U loads in flight -- the order EVERY finalizer of the library uses (then an xor butterfly over the lanes), so that they
// all produce the same bits.  chans * tile_pix values per full tile; tile_pix == 0: raw partials (strip mode).
// A full tile's count is a power of two for every tile shape of the GEMMs (32 channels x 32 / 64 / 128 pixels), and dividing by 2^k
// is an exact scaling: the product with the exact reciprocal has the same bits as tile_x2's quotient (s * s is exact in double: 48
// significant bits).  Only a sample's last, partial tile needs the fp64 division; it is the last element of its lane's chain, so it
// is added after the loop.  The straightforward loop -- tile_valid, an integer -> double conversion and a division per element, each
// guarded load its own branch -- was 25 k cycles for the 3 907 tiles of a 500 x 500 plane: 13 % of the candidate kernel's launch,
// spent by every block before its first byte moved.
// FAST = false: the plain loop only (the head's kernels fold ~8 elements per lane in every wave of ~500 short-lived blocks: there the
// two-path version measured slower -- head_k3 36 -> 54 us -- although it executes fewer instructions).
// COH: the partials were published by other blocks of THIS launch across coop_grid_barrier_nf: 8-byte agent-scope loads (consume8).
template <bool COH>
__device__ __forceinline__ f32x2 load_partial(const float *p)
{
    if constexpr (COH) return consume8(p);
    else return *reinterpret_cast<const f32x2 *>(p);
}
template <int U, bool FAST = true, bool COH = false>
__device__ __forceinline__ void fold_lane_chain(const float *pp, int ntiles, int tile_pix, int chans, int P, int lane, double &s1, double &s2)
{
    s1 = 0.0;
    s2 = 0.0;
    const int nfull = chans * tile_pix;
    if (FAST && tile_pix > 0 && (nfull & (nfull - 1)) == 0) {
        const int k = 31 - __builtin_clz((unsigned)nfull);
        const double inv = __longlong_as_double((long long)(1023 - k) << 52);   // 2^-k, exact
        const int tpart = (P % tile_pix) != 0 && P / tile_pix < ntiles ? P / tile_pix : -1;   // the partial tile, if any
        const int nclean = tpart >= 0 ? tpart : ntiles;                    // tiles [0, nclean) are full
        const bool mine = tpart >= 0 && (tpart & 63) == lane;              // (requested with the loop's first loads, used after them)
        const f32x2 ldp = load_partial<COH>(pp + 2 * (mine ? tpart : 0));
        for (int t0 = 0; t0 < nclean; t0 += 64 * U) {
            f32x2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * 64 + lane;
                const f32x2 ld = load_partial<COH>(pp + 2 * (t < nclean ? t : 0));   // (clamped: no branch around the load)
                v[u] = t < nclean ? ld : f32x2{0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const double dx = (double)v[u].x;
                s1 += dx;
                s2 += (double)v[u].y + dx * dx * inv;                      // = y + x * x / n: both products are exact, fused or not
            }
        }
        if (mine) {
            s1 += (double)ldp.x;
            s2 += tile_x2(ldp.x, ldp.y, chans * (P - tpart * tile_pix));
        }
        return;
    }
    for (int t0 = 0; t0 < ntiles; t0 += 64 * U) {
        f32x2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int t = t0 + u * 64 + lane;
            v[u] = t < ntiles ? load_partial<COH>(pp + 2 * t) : f32x2{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s1 += (double)v[u].x;
            s2 += tile_x2(v[u].x, v[u].y, chans * tile_valid(t0 + u * 64 + lane, tile_pix, P));
        }
    }
}

// Status words (include/urnn_hip.h): the first 256 bytes of a cell / head workspace.  The kernels that turn partial sums into a
// norm's (mean, rstd) OR a bit into word 0 when the statistics are not finite -- an operand beyond the f16 pieces' range
// (|activation| >= 2047 or |weight| >= 64 in the default matrix mode) turns into inf in the matrix pipe and surfaces here, one
// norm later at most.  The owner of the workspace zeroes it once and reads the word at a synchronisation point of its choice.
// (bits URNN_STATUS_GATES / _CAND / _HEAD: include/urnn_hip.h)
#define URNN_STATUS_BYTES 16384   // bytes 0-255: status word + the round-5 barrier words; 1024-9727: coop_grid_barrier_nf's words (include/urnn_hip.h)
#ifndef URNN_STATUS_BARRIER
#define URNN_STATUS_BARRIER 8     // a grid barrier gave up (include/urnn_hip.h)
#endif
__device__ __forceinline__ void flag_nonfinite(int *status, int bit, double s1, double s2)
{
    if (status && !(__builtin_isfinite(s1) && __builtin_isfinite(s2))) atomicOr(status, bit);   // (the sums, before any clamp)
}

// Activations.  URNN_ACT = 0: hardware exp2 / rcp on x * log2(e) (the argument's rounding costs |x| * 2^-24 relative -- a bias of
// +8.5e-8 on average -- and tanh's 1 - t cancels for small |x|); 1 (shipped): the argument's rounding error is carried along
// (exp_neg), the reciprocal is v_rcp_f32 + one Newton step, tanh uses its series for small arguments: ~1-2 ulp, unbiased.
// (Round 3 chased rare wrong values in 16 lanes of the candidate GEMM through these functions -- padded inline asm, a VALU-only
// sigmoid -- before the packed-fp32 instructions turned out to be the cause: u-rnn_amd/build_ext.py, -fno-slp-vectorize.)
#ifndef URNN_ACT
#define URNN_ACT 1
#endif
// 1 / d for d in [1, 2] (the activations' denominators 1 + e): hardware reciprocal (1 ulp) + one Newton step
__device__ __forceinline__ float rcp_1to2(float d)
{
#if URNN_ACT == 0
    return __frcp_rn(d);
#else
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
#endif
}
// exp(x) for x <= 0 (the only sign the activations need), ~1.5 ulp: v_exp_f32 does its own exact integer / fraction split, so only
// the argument needs care: t = rn(x * log2e_hi) goes to the hardware exp2, the part of x * log2(e) that t lost -- the product's
// rounding error (one fma) plus x * log2e_lo -- is at most 2^-24 |t| and enters as the first-order factor 1 + lo * ln 2.
__device__ __forceinline__ float exp_neg(float x)
{
#if URNN_ACT == 0
    return __expf(x);
#else
    const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-08f;   // log2(e) = hi + lo
    const float t = x * L2E_HI;
    float lo = fmaf(x, L2E_HI, -t);           // exact: what the rounding of t dropped
    lo = fmaf(x, L2E_LO, lo);
    const float e = __builtin_amdgcn_exp2f(t);           // v_exp_f32, 1 ulp over the whole range (underflows to 0 like expf)
    return fmaf(e, lo * 0.693147180559945f, e);
#endif
}
__device__ __forceinline__ float sigmoidf_fast(float v)
{
#if URNN_ACT == 0
    return __frcp_rn(1.0f + __expf(-v));
#else
    // sigmoid(v) = 1 / (1 + exp(-|v|)) for v >= 0, exp(-|v|) / (1 + exp(-|v|)) for v < 0: exp of a non-positive argument only
    const float e = exp_neg(-fabsf(v));
    const float r = rcp_1to2(1.0f + e);
    return v >= 0.f ? r : e * r;
#endif
}

// sigmoid(v) from t = v * log2(e), for callers that can fold log2(e) into an affine they evaluate anyway (cand_fused_kernel: the reset
// gate's GroupNorm).  Nine VALU instead of fifteen: the argument's rounding -- what exp_neg carries along -- perturbs e = 2^-|t| by at
// most ~|t| * 1e-7 relative, and sigmoid damps that by e / (1 + e): <= 1 ulp of the result, unbiased (a rounding, not a constant).
__device__ __forceinline__ float sigmoid_of_log2arg(float t)
{
    const float e = __builtin_amdgcn_exp2f(-fabsf(t));
    const float r = rcp_1to2(1.0f + e);
    return t >= 0.f ? r : e * r;
}

// The cell's gate activation  sigmoid(raw * scale + shift)  -- (scale, shift) = a GroupNorm folded per channel -- as every inference
// kernel that forms z or r evaluates it (blend, two-stream candidate, small-plane and cooperative cells, cell tail): log2(e) goes
// into the affine (two multiplies per channel, hoisted out of the pixel loops) and the sigmoid works on the exp2 argument directly.
// One definition, so that kernels which must agree bit for bit do.
__device__ __forceinline__ float gate_sigmoid(float raw, float scale, float shift)
{
    const float L2E = 1.44269502162933349609375f;
    return sigmoid_of_log2arg(fmaf(raw, nofma(scale * L2E), nofma(shift * L2E)));
}

__device__ __forceinline__ float tanhf_fast(float v)
{
#if URNN_ACT == 0
    // tanh(v) = sign(v) * (1 - t) / (1 + t), t = exp(-2|v|)  (absolute error ~1e-7)
    const float t = __expf(-2.0f * fabsf(v));
    const float r = (1.0f - t) * __frcp_rn(1.0f + t);
    return copysignf(r, v);
#else
    const float a = fabsf(v);
    // Taylor series to a^13 on [0, 0.4]: tanh(a) = a + a^3 P(a^2), truncation < 4e-9 relative (branch-free: both forms evaluated)
    const float s = a * a;
    float p = fmaf(s, 21844.0f / 6081075.0f, -1382.0f / 155925.0f);     // +a^13, -a^11
    p = fmaf(s, p, 62.0f / 2835.0f);
    p = fmaf(s, p, -17.0f / 315.0f);
    p = fmaf(s, p, 2.0f / 15.0f);
    p = fmaf(s, p, -1.0f / 3.0f);
    const float small = fmaf(a * s, p, a);
    const float t = exp_neg(-2.0f * a);
    const float big = (1.0f - t) * rcp_1to2(1.0f + t);          // t <= 0.45 here: 1 - t loses no more than one bit
    return copysignf(a < 0.4f ? small : big, v);
#endif
}

// Truncating three-way bf16 split of an fp32 value (exact: x = hi + mid + lo, 8 significant bits each); piece 0 / 1 / 2 as the
// 16 bits of the bf16 encoding.
__device__ __forceinline__ unsigned bf16_piece(float x, int piece)
{
    unsigned u = __float_as_uint(x);
    if (piece == 0) return u >> 16;
    float r = x - __uint_as_float(u & 0xffff0000u);
    u = __float_as_uint(r);
    if (piece == 1) return u >> 16;
    r = r - __uint_as_float(u & 0xffff0000u);
    return __float_as_uint(r) >> 16;
}

// The same for a fold in which thread `tid` of NT walks tiles tid, tid + NT, ... (gru_blend_kernel<FIN>'s order, which the cooperative cell's
// last phase and the cell tail reproduce): every block of the blend folds its channel group's partials before it streams -- ~2 000
// blocks x 16 fp64 divisions per thread at 500 x 500 were ~10 us of the chip's VALU per launch.
template <int NT>
__device__ __forceinline__ void fold_thread_chain(const float *pp, int ntiles, int tile_pix, int chans, int P, int tid, double &a1, double &a2)
{
    a1 = 0.0;
    a2 = 0.0;
    const int nfull = chans * tile_pix;
    if (tile_pix > 0 && (nfull & (nfull - 1)) == 0) {
        const int k = 31 - __builtin_clz((unsigned)nfull);
        const double inv = __longlong_as_double((long long)(1023 - k) << 52);   // 2^-k, exact
        const int tpart = (P % tile_pix) != 0 && P / tile_pix < ntiles ? P / tile_pix : -1;   // the partial tile, if any: the last of its thread's chain
        const int nclean = tpart >= 0 ? tpart : ntiles;
        const bool mine = tpart >= 0 && tpart % NT == tid;
        const f32x2 ldp = *reinterpret_cast<const f32x2 *>(pp + 2 * (mine ? tpart : 0));
        for (int t = tid; t < nclean; t += NT) {
            const f32x2 v = *reinterpret_cast<const f32x2 *>(pp + 2 * t);
            const double dx = (double)v.x;
            a1 += dx;
            a2 += (double)v.y + dx * dx * inv;                                   // = tile_x2: both products are exact
        }
        if (mine) {
            a1 += (double)ldp.x;
            a2 += tile_x2(ldp.x, ldp.y, chans * (P - tpart * tile_pix));
        }
        return;
    }
    for (int t = tid; t < ntiles; t += NT) {
        const f32x2 v = *reinterpret_cast<const f32x2 *>(pp + 2 * t);
        a1 += (double)v.x;
        a2 += tile_x2(v.x, v.y, chans * tile_valid(t, tile_pix, P));
    }
}

// ---- grid barriers of the cooperative launches (urnn_small.hip coop_cell_kernel, urnn_elem.hip head_coop_kernel) ------------------------
// Round 6 (tools/ubench/grid_barrier.hip, profiles/r06_grid_barrier.txt): the round-5 barrier below -- one arrival counter, one
// generation word, an agent-scope release fence before the arrival and an acquire fence after the release -- costs 7.7 us for 245
// blocks (1.1 us + ~25 ns per arrival on the one word, ~1.7 us per fence), and the quarter-resolution cell spent 17 of its 37 us in
// two of them.  coop_grid_barrier_nf: arrivals sharded 16 ways (blockIdx % 16; <= 16 arrivals per word, the last arriver of a shard
// bumps a top counter, the last of those writes 16 generation words, a block polls its shard's) and NO fences: 2.7 us for 245 blocks.
// Its contract: everything a block publishes across the barrier travels as 8-byte agent-scope atomic stores (write-through) and is
// read back with 8-byte agent-scope atomic loads (publish8 / consume8 below; MI355X_MICROARCH.md "valid forms": 8-B agent atomics on
// both sides) -- the cooperative kernels publish nothing but per-tile partial statistics, one (sum, centred second moment) pair per
// wave.  Plain stores / loads across this barrier are NOT ordered by it.
// Barrier words (dwords from `bar`, 64 apart = 256 B so that no two share a channel's line): [64 s] shard arrivals, [1024] top counter,
// [1088 + 64 s] shard generation words, s < 16: the workspace's status area holds them from byte 1024 on (URNN_STATUS_BYTES).
#define URNN_BARRIER_SHARDS 16
__device__ __forceinline__ void coop_grid_barrier_nf(unsigned *bar, unsigned block, unsigned nblocks, int *status)
{
    constexpr unsigned NS = URNN_BARRIER_SHARDS;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every wave: its published granules have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = block % NS;
        unsigned *genw = bar + 1088 + 64 * s;
        const unsigned gen = __hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned mine = nblocks / NS + (s < nblocks % NS ? 1u : 0u);     // blocks of this shard
        const unsigned used = nblocks < NS ? nblocks : NS;                      // shards with blocks
        bool last = false;
        if (__hip_atomic_fetch_add(&bar[64 * s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1) {
            __hip_atomic_store(&bar[64 * s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(&bar[1024], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1) {
                __hip_atomic_store(&bar[1024], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = true;
            }
        }
        if (last) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the counter resets have landed before anybody is released
            for (unsigned k = 0; k < NS; ++k) __hip_atomic_store(bar + 1088 + 64 * k, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) {               // ~1 s: something else holds the chip; give up loudly instead of hanging it
                    if (status) atomicOr(status, URNN_STATUS_BARRIER);
                    break;
                }
            }
        }
    }
    __syncthreads();
}

// The round-5 barrier: one monotonic generation word + an arrival counter the last arriver resets (MI355X_MICROARCH.md "barrier-counter"
// with the hand-off protocol of cdna_hip_programming.md section 6 G16: every wave drains its stores, one lane releases at agent scope,
// polls relaxed, acquires once): orders PLAIN stores / loads across it.  <= 256 arrivals.
__device__ __forceinline__ void coop_grid_barrier(unsigned *bar, unsigned nblocks, int *status)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned gen = __hip_atomic_load(&bar[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nblocks - 1) {
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&bar[16], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            int spins = 0;
            while (__hip_atomic_load(&bar[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1 << 22)) {               // ~1 s: something else holds the chip; give up loudly instead of hanging it
                    if (status) atomicOr(status, URNN_STATUS_BARRIER);
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}


// The GRU blend h' = (1 - z) h + z n (ConvRNN.py:189) with every operation rounded on its own, as the reference's eager torch ops
// round them -- and so that every kernel that blends (gru_blend_kernel in all its vector forms, coop_cell_kernel, blend_conv_kernel)
// produces the same bits.
__device__ __forceinline__ float gru_blend(float z, float n, float h) { return nofma((1.f - z) * h) + nofma(z * n); }

__device__ __forceinline__ float lrelu(float v, float slope) { return v >= 0.f ? v : v * slope; }
__device__ __forceinline__ float siluf_fast(float v) { return v * sigmoidf_fast(v); }

// bf16 pieces of fp32 operands for the 16-bit matrix pipe (SPLIT kernels).  Truncating splits are EXACT: x = hi + mid + lo with
// 8 significant bits each, so  a*b = hh + hm + mh + hl + lh + mm  up to the three dropped terms (<= 3 * 2^-24 |a*b|, one fp32
// rounding's worth); measured against float64 on K = 224 dot products: rms error 2.2e-7 of sqrt(sum (w x)^2) vs 2.7e-7 for the
// fp32 MFMA's own fma chain (tools/ubench/split_mfma.hip, profiles/r02_split_mfma.txt) -- at 16/6 of its rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One dword of each piece from two fp32 values: element 2d = the even k-pair's value, 2d + 1 = the odd one's.
__device__ __forceinline__ void split_pair(float xe, float xo, unsigned &ph, unsigned &pm, unsigned &pl)
{
    const unsigned ue = __float_as_uint(xe), uo = __float_as_uint(xo);
    ph = __builtin_amdgcn_perm(uo, ue, 0x07060302u);                       // {uo[31:16], ue[31:16]}: truncation to bf16
    const float re = xe - __uint_as_float(ue & 0xffff0000u), ro = xo - __uint_as_float(uo & 0xffff0000u);   // exact
    const unsigned ve = __float_as_uint(re), vo = __float_as_uint(ro);
    pm = __builtin_amdgcn_perm(vo, ve, 0x07060302u);
    const float se = re - __uint_as_float(ve & 0xffff0000u), so = ro - __uint_as_float(vo & 0xffff0000u);   // exact, <= 8 bits left
    pl = __builtin_amdgcn_perm(__float_as_uint(so), __float_as_uint(se), 0x07060302u);
}

// ---- f16 x 3 (SPLIT = 3): the forward GEMMs' arithmetic ------------------------------------------------------------------------
// x = hi + lo with hi = RNE_f16(x * 2^a), lo = RNE_f16(x * 2^a - hi) (the residual is exact in fp32): 22 significant bits, and
// a*b = hh + hl + lh up to 2^-22 |ab| -- three v_mfma_f32_32x32x16_f16 per 16 k instead of six bf16 ones, two VALU per element
// instead of 5.5.  f16 has 5 exponent bits, so both operands are scaled by powers of two (exact) into its comfortable range and
// the accumulator is scaled back in the epilogue: activations by 2^5 (lo keeps all its bits for |x| >= 2^-8, below that the
// absolute error is <= 2^-30; finite up to |x| < 2047), weights by 2^10 (full precision for |w| >= 2^-13, finite for |w| < 64).
// Measured against float64 (K = 224, tools/ubench/split_mfma.hip, profiles/r02_split_mfma.txt): rms 1.8e-7 of sqrt(sum (w x)^2)
// (fp32 MFMA chain 2.7e-7, bf16 x 6 2.2e-7).  Gradients (unbounded below) keep the bf16 x 6 split, which has fp32's exponent range.
// An activation beyond the f16 range turns into inf -> NaN statistics -> NaN frames: loud, not silent.
#define URNN_F16_AEXP 5
#define URNN_F16_WEXP 10
#define URNN_F16_ASCALE 32.0f
#define URNN_F16_WSCALE 1024.0f
#define URNN_F16_DESCALE 0x1p-15f
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// One dword of each f16 piece from two fp32 values (low half: xe, high half: xo), both scaled by s (a power of two).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
#ifndef URNN_SPLIT2_ASM
#define URNN_SPLIT2_ASM 0
#endif
__device__ __forceinline__ void split2_pair(float xe, float xo, float s, unsigned &ph, unsigned &pl)
{
#if URNN_SPLIT2_ASM
    // v_fma_mix*: fp32 fma whose result is rounded (RNE) into one half of the destination; op_sel_hi marks an f16 source, op_sel
    // picks its high half.  hi = rne(x*s); lo = rne(x*s - hi): four VALU for two elements (tuning builds only: partial-register
    // writes inside an asm statement, nothing pads their hazards)
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "s_nop 1\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(ph), "=&v"(pl)
        : "v"(xe), "v"(xo), "s"(s));
#else
    // hi = rne_f16(x * s) (v_cvt_pk_f16_f32: both halves in one full-register write); the residual x * s - hi is exact in fp32
    // (v_fma_mix_f32 with the f16 half as addend); lo = rne_f16(residual).  Compiler-scheduled: hazards and waits are its business.
    const float se = xe * s, so = xo * s;
    const f16x2 hi = __builtin_convertvector(f32x2v{se, so}, f16x2);
    const float re = se - (float)hi.x, ro = so - (float)hi.y;
    const f16x2 lo = __builtin_convertvector(f32x2v{re, ro}, f16x2);
    ph = __builtin_bit_cast(unsigned, hi);
    pl = __builtin_bit_cast(unsigned, lo);
#endif
}
// the same split of one value (weight packers): piece 0 = hi, 1 = lo, as the 16 bits of the f16 encoding
__host__ __device__ static inline unsigned f16_piece(float x, int piece)
{
    const _Float16 h = (_Float16)x;
    const _Float16 v = piece == 0 ? h : (_Float16)(x - (float)h);
    unsigned short u;
    __builtin_memcpy(&u, &v, 2);
    return u;
}
__device__ __forceinline__ f16x8 as_f16x8(const unsigned (&p)[4]) { return __builtin_bit_cast(f16x8, u32x4{p[0], p[1], p[2], p[3]}); }

// bf16 compute mode (SPLIT = 2): one dword of round-to-nearest-even bf16 values (v_cvt_pk_bf16_f32)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned round_pair(float xe, float xo)
{
    const bf16x2 v = {(__bf16)xe, (__bf16)xo};
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ bf16x8 as_bf16x8(const unsigned (&p)[4]) { return __builtin_bit_cast(bf16x8, u32x4{p[0], p[1], p[2], p[3]}); }

