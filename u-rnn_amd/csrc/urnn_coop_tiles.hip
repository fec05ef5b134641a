// urnn_coop_tiles.hip -- a WHOLE ConvGRU / Skip-ConvGRU cell of a HALF-RESOLUTION plane (ConvRNN.py:111-194; 62 500 pixels at
// 500x500, 56 000 at Futian's 400x560) in ONE cooperative launch whose raw gates and candidate never leave the chip.
//
// The three-kernel cell moves the raw gates (2F planes) and the raw candidate (F planes) out to HBM and back and reads its inputs
// twice: 608 MB per frame for the two half-resolution cells against 160 MB of algorithmic traffic, 166 us.  Here a block is one CU
// and owns up to FOUR 64-pixel tiles (977 tiles over 256 CUs); its output -- 3F channels (z | r | c) x 256 pixels -- stays in the
// accumulators of its twelve waves from the first k-step to the blend:
//   wave = (role wr in {z, r, c}, tile slot wc): all F channels of its role for the 64 pixels of tile wc = F/32 x 2 MFMA tiles of
//   32 x 32 (96 accumulator registers at F = 96); waves wc sit on SIMD wc (a z-, an r- and a c-wave per SIMD).
// Phase A (output-stationary GEMM, K streamed in 16-k groups):
//   * activations: the group's 16 input rows x 4 tiles arrive as fp32 through LDS-DMA (4 rows x 256 B per instruction, issued five
//     groups ahead by the z- and r-waves) into a SIX-slot ring; a wave reads its tile's B fragments straight from the ring and splits
//     them into f16 pieces in registers (the three waves of a tile redo the split: VALU is idle next to the matrix pipe here);
//   * weights: the group's 3F/32 blocks x two f16 pieces (the packed slabs of the three-kernel cell, L2-resident) arrive by LDS-DMA
//     from the c-waves two groups ahead (three buffers) -- ONE copy per CU and k-group for all four tiles (the activation-stationary
//     kernels of urnn_small.hip re-stream them per 64-pixel tile, which is what made them lose at this size);
//   * z / r accumulate all K groups, c the x | e groups; one __syncthreads per group.
//   The gates' centred tile statistics go out as 8-byte agent-scope granules.                                     | grid barrier 1
// Phase B: a wave per norm group folds the statistics (the order every finalizer shares) -> (scale, shift) table in LDS.  The ring's
//   last F/16 slots still hold the tiles' hidden state: the r-waves turn it IN PLACE into r (.) h (one dword = the value's two f16
//   pieces), the c-waves finish the candidate with W2[:, h] (staged once per CU) and publish its statistics, the z-waves apply
//   their sigmoid meanwhile.                                                                                      | grid barrier 2
// Phase C: z goes through LDS to the c-waves, which fold the candidate's statistics, re-read h from global memory (L2 / MALL) and
//   blend.  HBM traffic: K input planes + F planes of h a second time + F planes out.
// Arithmetic: the f16 x 3 pieces, MFMA order and activation functions of every other forward kernel; GroupNorm partials per 64-pixel
// tile, folded in the shared order.  Every block must be resident (one per CU); barriers: urnn_common.h coop_grid_barrier_nf.
#include "urnn_common.h"
#include "urnn_kernels.h"

#include <limits.h>

typedef __attribute__((address_space(3))) void *ct_lds_ptr_t;
typedef const __attribute__((address_space(1))) void *ct_gbl_ptr_t;
typedef __amdgpu_buffer_rsrc_t ct_rsrc_t;

extern __shared__ __attribute__((aligned(16))) char urnn_ct_smem[];

#ifdef URNN_TRACE
static __device__ unsigned long long *urnn_ct_trace_buf = nullptr;
extern "C" int urnn_debug_set_trace_urnn_coop_tiles(unsigned long long *p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(urnn_ct_trace_buf), &p, sizeof(p)); }
#define CT_STAMP(k) do { if (urnn_ct_trace_buf && lane == 0) urnn_ct_trace_buf[((size_t)blockIdx.x * 16 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CT_STAMP(k) do { } while (0)
#endif

static constexpr int CT_D = 6;                    // ring slots (16-k groups of activations in LDS): >= F/16, the hidden state stays resident
static constexpr int CT_NT = 4;                   // tiles per block
static constexpr int CT_SLOT = CT_NT * 4096;      // one group's 16 rows x 64 pixels x fp32 of every tile
static constexpr int CT_WBUF = 20 * 1024;         // one group's weight blocks (9 x 2 KB at F = 96) + the DMA padding
static constexpr int CT_NW = 3;                   // weight buffers

struct CoopTilesParams {
    const float *seg[3];      // x | e | h (encoder: x | h)
    int segC[3];
    int segG0[3];             // first absolute 16-k group of each segment (INT_MAX: unused)
    unsigned segBytes[3];     // bytes of each segment's tensor (the DMA descriptors' bounds)
    int kg0, KG, KGxe;        // first absolute group, groups of the gate GEMM, the first KGxe of them are the candidate's plain (x | e) rows
    const unsigned *wblk[9];  // per canonical block [z_0 .. | r_0 .. | c_0 ..]: its f16 pieces at absolute group 0 ...
    int wstride[9];           // ... and the dwords from one group to the next
    const float *gbias;       // gate bias in packed order; block cb's 32 values start at gbiasOff[cb]
    int gbiasOff[6];
    const float *cbias;       // [F]
    const float *gn1_w, *gn1_b, *gn2_w, *gn2_b;
    float eps;
    const float *h;
    float *h_out;
    float *partial1, *partial2;   // [B][2F/32][tiles][2], [B][F/32][tiles][2]: 64-pixel tiles
    float *ss1_out, *ss2_out;     // [B][2F][2], [B][F][2]: what the three-kernel cell leaves in the workspace
    float *st1_out, *st2_out;     // [B][2F/32][2], [B][F/32][2] (mean, rstd)
    unsigned *bar;
    int *status;
    int B, P, F, tilesPerSample, totalTiles, nblocks;
};

__device__ __forceinline__ void ct_bdma16(ct_rsrc_t r, unsigned voff, unsigned soff, char *l) { __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (ct_lds_ptr_t)l, 16, voff, soff, 0, 0); }
__device__ __forceinline__ void ct_dma16(const unsigned *g, char *l) { __builtin_amdgcn_global_load_lds((ct_gbl_ptr_t)g, (ct_lds_ptr_t)l, 16, 0, 0); }
template <int N>
__device__ __forceinline__ void ct_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void ct_wait_vm_groups(int groups, int per)       // at most `groups` x `per` VMEM operations still in flight
{
    const int n = groups * per;
    if (n >= 10) ct_wait_vmcnt<10>();
    else if (n >= 8) ct_wait_vmcnt<8>();
    else if (n >= 6) ct_wait_vmcnt<6>();
    else if (n >= 5) ct_wait_vmcnt<5>();
    else if (n >= 4) ct_wait_vmcnt<4>();
    else if (n >= 3) ct_wait_vmcnt<3>();
    else if (n >= 2) ct_wait_vmcnt<2>();
    else ct_wait_vmcnt<0>();
}

// Block barrier WITHOUT the compiler's fence: __syncthreads() makes hipcc drain vmcnt to 0 first (it must assume the LDS-DMA in flight is
// covered by the fence), which would wait for the rows requested five groups ahead at every k-step.  The waits that matter are explicit:
// counted vmcnt waits for the DMA a wave issued itself, lgkmcnt(0) for its LDS traffic.
__device__ __forceinline__ void ct_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// arrive at / leave the sharded grid barrier of urnn_common.h from ONE thread; the caller brackets it with its own __syncthreads
__device__ __forceinline__ void ct_barrier_lane(unsigned *bar, unsigned block, unsigned nblocks, int *status)
{
    constexpr unsigned NS = URNN_BARRIER_SHARDS;
    const unsigned s = block % NS;
    unsigned *genw = bar + 1088 + 64 * s;
    const unsigned gen = __hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned mine = nblocks / NS + (s < nblocks % NS ? 1u : 0u);
    const unsigned used = nblocks < NS ? nblocks : NS;
    bool last = false;
    if (__hip_atomic_fetch_add(&bar[64 * s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1) {
        __hip_atomic_store(&bar[64 * s], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&bar[1024], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == used - 1) {
            __hip_atomic_store(&bar[1024], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = true;
        }
    }
    if (last) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (unsigned k = 0; k < NS; ++k) __hip_atomic_store(bar + 1088 + 64 * k, gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        int spins = 0;
        while (__hip_atomic_load(genw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) {
                if (status) atomicOr(status, URNN_STATUS_BARRIER);
                break;
            }
        }
    }
}

// G = F / 32 channel blocks per role
// One role's whole program.  The three roles are three instantiations called from three branches of the kernel, NOT one body with
// role tests inside: with a shared body the register allocator sees one set of accumulators live on every path (an r-wave's are dead
// once r (.) h is formed, and its registers then hold the tile's previous state) and spills ~150 registers around the phases.
template <int G, int ROLE>
__device__ __forceinline__ void ct_role(const CoopTilesParams &cp, const int lane, const int wave)
{
    constexpr int wr = ROLE;                                      // role: 0 z, 1 r, 2 c
    const int wc = wave & 3;                                      // tile slot
    const int j = lane & 31, half = lane >> 5;
    const int P = cp.P, F = cp.F;
    const int KG = cp.KG, KGxe = cp.KGxe, NH = F / 16;            // NH hidden-state groups = the last groups of the k-loop
    char *ring = urnn_ct_smem;
    char *wbuf = ring + CT_D * CT_SLOT;
    float *biasl = reinterpret_cast<float *>(wbuf + CT_NW * CT_WBUF);   // [3 G 32] z | r | c channels
    // from phase B on (the weight buffers are free past the 12 G^2 KB of W2[:, h]): (scale, shift) tables, one per tile slot
    float *sstab = reinterpret_cast<float *>(wbuf + 48 * 1024);           // [4 slots][3 G 32][2]
    CT_STAMP(0);

    // ---- tiles of this block: slot ti holds global tile blockIdx * CT_NT + ti (sample b, 64-pixel tile t): consecutive tiles, so a block
    // straddles two samples only at a sample's end -----------------------------------------------------------------------------------------
    auto slot_tile = [&](int ti, int &b_, int &t_) -> bool {
        const int T = (int)blockIdx.x * CT_NT + ti;
        const bool on = T < cp.totalTiles;
        b_ = on ? T / cp.tilesPerSample : 0;
        t_ = on ? T - b_ * cp.tilesPerSample : 0;
        return on;
    };
    int tb, tt;
    const bool tile_on = slot_tile(wc, tb, tt);                   // (whole wave)
    const int nvalid = tile_on ? tile_valid(tt, 64, P) : 0;

    for (int i = threadIdx.x; i < 3 * G * 32; i += blockDim.x) {
        const int cb = i >> 5;
        biasl[i] = cb < 2 * G ? cp.gbias[cp.gbiasOff[cb] + (i & 31)] : cp.cbias[i - 2 * G * 32];
    }

    // ---- DMA duties ---------------------------------------------------------------------------------------------------------------------
    // waves 0-7: activations.  Wave w moves rows 4 qd .. 4 qd + 3 (qd = w & 3) of tile slots w >> 2 and (w >> 2) + 2: lane l = row l >> 4,
    // pixels 4 (l & 15) .. + 3.  A slot without a tile re-reads tile 0 of sample 0 (nobody looks at it; the issue count stays uniform).
    const ct_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cp.seg[0]), 0, (int)cp.segBytes[0], 0x00020000);
    const ct_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cp.seg[1]), 0, (int)cp.segBytes[1], 0x00020000);
    const ct_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(cp.seg[2]), 0, (int)cp.segBytes[2], 0x00020000);
    const int qd = wave & 3, ts0 = (wave >> 2) & 1;
    unsigned voff[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int bk, tk;
        slot_tile(ts0 + 2 * k, bk, tk);
#pragma unroll
        for (int s = 0; s < 3; ++s)
            voff[k][s] = (unsigned)((((size_t)bk * cp.segC[s] + (lane >> 4)) * P + (size_t)tk * 64 + 4 * (lane & 15)) * 4);
    }
    auto issue_acts = [&](int gn) __attribute__((always_inline)) {           // the 16 rows of relative group gn -> ring slot gn % D
        const int Ga = cp.kg0 + gn;
        const int s = Ga >= cp.segG0[2] ? 2 : (Ga >= cp.segG0[1] ? 1 : 0);
        const unsigned soff = (unsigned)((16 * (Ga - cp.segG0[s]) + 4 * qd) * P) * 4u;
        char *dst = ring + (gn % CT_D) * CT_SLOT + qd * 1024;
        if (s == 0) {
            ct_bdma16(rs0, voff[0][0], soff, dst + ts0 * 4096);
            ct_bdma16(rs0, voff[1][0], soff, dst + (ts0 + 2) * 4096);
        } else if (s == 1) {
            ct_bdma16(rs1, voff[0][1], soff, dst + ts0 * 4096);
            ct_bdma16(rs1, voff[1][1], soff, dst + (ts0 + 2) * 4096);
        } else {
            ct_bdma16(rs2, voff[0][2], soff, dst + ts0 * 4096);
            ct_bdma16(rs2, voff[1][2], soff, dst + (ts0 + 2) * 4096);
        }
    };
    // waves 8-11: weights.  2 x 3 G pieces of 1 KB per group, NQ per wave; indices past the end re-copy the first pieces into the buffer's padding
    constexpr int NPCS = 2 * 3 * G, NQ = (NPCS + 3) / 4;
    auto issue_weights = [&](int gn) __attribute__((always_inline)) {
        const int Ga = cp.kg0 + gn;
        char *dst = wbuf + (gn % CT_NW) * CT_WBUF;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int n = wc + 4 * q, nn = n < NPCS ? n : n - NPCS;
            const int blk = nn >> 1;
            ct_dma16(cp.wblk[blk] + (size_t)Ga * cp.wstride[blk] + (nn & 1) * 256 + lane * 4, dst + n * 1024);
        }
    };
    if (wr < 2) {
        for (int gn = 0; gn < CT_D && gn < KG; ++gn) issue_acts(gn);       // every ring slot
    } else {
        issue_weights(0);
        if (KG > 1) issue_weights(1);
    }

    f32x16 acc[G][2];
#pragma unroll
    for (int blk = 0; blk < G; ++blk)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[blk][pb][r] = 0.f;

    // B fragments of this wave's tile for one 16-k group: dword d of lane (j, half) = rows 4 d + half (low) and 4 d + 2 + half (high) of the
    // group at pixel 32 pb + j, as two f16 pieces.  Item i < 8 = (pb = i >> 2, d = i & 3).
    const int boff = wc * 4096 + half * 256 + j * 4;
    auto prep_item = [&](const char *slot, int i, unsigned (&ph)[2][4], unsigned (&pl)[2][4]) __attribute__((always_inline)) {
        const int pb = i >> 2, d = i & 3;
        const float v0 = *reinterpret_cast<const float *>(slot + boff + (4 * d) * 256 + pb * 128);
        const float v1 = *reinterpret_cast<const float *>(slot + boff + (4 * d + 2) * 256 + pb * 128);
        split2_pair(v0, v1, URNN_F16_ASCALE, ph[pb][d], pl[pb][d]);
    };
    auto mfma_block = [&](const char *wsrc, int blk, const unsigned (&ph)[2][4], const unsigned (&pl)[2][4]) __attribute__((always_inline)) {
        const f16x8 ah = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(wsrc + blk * 2048 + lane * 16));
        const f16x8 al = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4 *>(wsrc + blk * 2048 + 1024 + lane * 16));
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {                    // f16 x 3, small terms first (the order of every forward kernel)
            acc[blk][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, as_f16x8(ph[pb]), acc[blk][pb], 0, 0, 0);
            acc[blk][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(pl[pb]), acc[blk][pb], 0, 0, 0);
            acc[blk][pb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, as_f16x8(ph[pb]), acc[blk][pb], 0, 0, 0);
        }
    };

    // ---- phase A: the k-loop.  CT_PIPELINE = 1 (built, measured, off): group g + 1's fragments read and split under group g's MFMAs.  It fits
    // the registers since the roles are separate instantiations (166, no spill) and is SLOWER: 30.2 / 31.8 instead of 28.0 / 28.5 hundred
    // cycles per group at K = 160 / 288 (profiles/r06_trace_coop_tiles.txt).  The loop takes the same ~2 800 cycles per 16-row group whether
    // the c-waves multiply (4 of 10 groups at K = 160, 12 of 18 at K = 288) or not: it is bound by the arrival of the group's 16 KB of rows
    // (5.9 B per clock and CU = 3.4 TB/s chip-wide, the rate every row-streaming GEMM of this library reaches), not by the matrix pipe.
    CT_STAMP(1);
#ifndef CT_PIPELINE
#define CT_PIPELINE 0
#endif
#if CT_PIPELINE
    unsigned curh[2][4], curl[2][4], nxth[2][4], nxtl[2][4];
    {   // group 0's fragments
        if (wr < 2) ct_wait_vm_groups((KG < CT_D ? KG : CT_D) - 1, 2);
        else ct_wait_vm_groups(KG > 1 ? 1 : 0, NQ);
        ct_sync();
        if (tile_on) {
#pragma unroll
            for (int i = 0; i < 8; ++i) prep_item(ring, i, curh, curl);
        }
    }
    for (int g = 0; g < KG; ++g) {
        const bool more = g + 1 < KG;
        if (wr < 2) {                                   // group g + 1's rows have landed; the groups requested after it may still travel
            const int issued = g == 0 ? (KG < CT_D ? KG : CT_D) - 1 : (g - 1 + CT_D < KG - 1 ? g - 1 + CT_D : KG - 1);   // last group requested so far
            ct_wait_vm_groups(more ? issued - (g + 1) : 0, 2);
        } else {                                        // group g's weights: at most group g + 1's in flight
            ct_wait_vm_groups(more ? 1 : 0, NQ);
        }
        ct_sync();                                      // ... for every wave's share; everybody has taken group g's fragments and is done with group g - 1's weights
        if (wr < 2) {
            if (g + CT_D < KG) issue_acts(g + CT_D);                    // -> the slot of group g
        } else {
            if (g + 2 < KG) issue_weights(g + 2);                       // -> the buffer of group g - 1
        }
        const bool mul = tile_on && (wr < 2 || g < KGxe);
        const bool prep = tile_on && more && (wr < 2 || g + 1 < KGxe);
        const char *wsrc = wbuf + (g % CT_NW) * CT_WBUF + wr * G * 2048;
        const char *nslot = ring + ((g + 1) % CT_D) * CT_SLOT;
        constexpr int PER = (8 + G - 1) / G;            // fragment items per block step: 3, 3, 2 at G = 3
#pragma unroll
        for (int blk = 0; blk < G; ++blk) {
            if (mul) mfma_block(wsrc, blk, curh, curl);
            if (prep) {
#pragma unroll
                for (int i = blk * PER; i < (blk + 1) * PER && i < 8; ++i) prep_item(nslot, i, nxth, nxtl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (prep) {
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    curh[pb][d] = nxth[pb][d];
                    curl[pb][d] = nxtl[pb][d];
                }
        }
    }
#else
    // (Software-pipelining the split of group g + 1 under group g's MFMAs needs a second set of fragment registers: 96 accumulators + 2 x 16
    // pieces + A fragments + addressing exceed the 168 registers of three waves per SIMD and the accumulators spill inside the loop.)
    for (int g = 0; g < KG; ++g) {
        if (wr < 2) {                                   // group g's rows have landed; the groups requested after it may still travel
            const int issued = g == 0 ? (KG < CT_D ? KG : CT_D) - 1 : (g - 2 + CT_D < KG - 1 ? g - 2 + CT_D : KG - 1);   // last group requested so far
            ct_wait_vm_groups(issued - g, 2);
        } else {                                        // group g's weights: at most group g + 1's in flight
            ct_wait_vm_groups(g + 1 < KG ? 1 : 0, NQ);
        }
        ct_sync();                                      // ... for every wave's share; and everybody is done with group g - 1's slot and weights
        if (wr < 2) {
            if (g > 0 && g - 1 + CT_D < KG) issue_acts(g - 1 + CT_D);   // -> the slot of group g - 1
        } else {
            if (g + 2 < KG) issue_weights(g + 2);                       // -> the buffer of group g - 1
        }
        if (tile_on && (wr < 2 || g < KGxe)) {
            unsigned ph[2][4], pl[2][4];
            const char *slot = ring + (g % CT_D) * CT_SLOT;
#pragma unroll
            for (int i = 0; i < 4; ++i) prep_item(slot, i, ph, pl);
            __builtin_amdgcn_sched_barrier(0);           // (one pixel half's rows in registers at a time)
#pragma unroll
            for (int i = 4; i < 8; ++i) prep_item(slot, i, ph, pl);
            __builtin_amdgcn_sched_barrier(0);
            const char *wsrc = wbuf + (g % CT_NW) * CT_WBUF + wr * G * 2048;
#pragma unroll
            for (int blk = 0; blk < G; ++blk) {
                mfma_block(wsrc, blk, ph, pl);
                __builtin_amdgcn_sched_barrier(0);       // (one block's A fragments at a time: all three hoisted cost 16 registers the loop does not have)
            }
        }
    }
#endif
    CT_STAMP(2);

    // Everything below addresses memory through these copies: laundered so that the compiler cannot form the phases' per-lane addresses
    // BEFORE the k-loop and carry them through it (it did: ~40 registers, and an accumulator tile lived in scratch inside the loop).
    int lane_b = lane;
    asm volatile("" : "+v"(lane_b));
    const int j_b = lane_b & 31, half_b = lane_b >> 5;
    const int boff_b = wc * 4096 + half_b * 256 + j_b * 4;

    auto row_c = [](int r) { return (r & 3) + 8 * (r >> 2); };
    const float inv_n = nvalid == 64 ? 1.0f / 2048.0f : 1.0f / (32.0f * (float)(nvalid > 0 ? nvalid : 1));
    const bool okp[2] = {tt * 64 + j_b < P && tile_on, tt * 64 + 32 + j_b < P && tile_on};
    // accumulators -> raw values (bias added, in place); centred statistics of every (64-pixel tile, 32-channel group) of this wave's role
    auto finish_and_stats = [&](int role, float *partial, int ngroups) __attribute__((always_inline)) {
        float s1[G], s2[G], mt[G];
#pragma unroll
        for (int blk = 0; blk < G; ++blk) {
            const float *bias_h = biasl + (role * G + blk) * 32 + 4 * half_b;
            s1[blk] = 0.f;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[blk][pb][r] = fmaf(acc[blk][pb][r], URNN_F16_DESCALE, bias_h[row_c(r)]);
                    if (okp[pb]) s1[blk] += acc[blk][pb][r];
                }
        }
        wave_sum_n<G>(s1);                                  // (the G reductions interleaved: one latency instead of G)
#pragma unroll
        for (int blk = 0; blk < G; ++blk) {
            mt[blk] = nofma(s1[blk] * inv_n);
            s2[blk] = 0.f;
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[blk][pb][r] - mt[blk];
                    if (okp[pb]) s2[blk] = fmaf(d, d, s2[blk]);
                }
        }
        wave_sum_n<G>(s2);
        if (lane_b == 0 && nvalid > 0) {
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
                publish8(partial + (((size_t)tb * ngroups + (role == 2 ? blk : role * G + blk)) * cp.tilesPerSample + tt) * 2, s1[blk], s2[blk]);
        }
    };

    // GroupNorm of one norm group for every tile slot of this block: (scale, shift) of its 32 channels into the slot's table.  Slots of one
    // sample share the fold (consecutive tiles: a block straddles samples only where B > 1 and a sample ends inside it).
    int sb[CT_NT];
    bool son[CT_NT];
#pragma unroll
    for (int ti = 0; ti < CT_NT; ++ti) {
        int t_;
        son[ti] = slot_tile(ti, sb[ti], t_);
    }
    // l2e: the table is a gate's -- its entries are (scale, shift) x log2(e), what gate_sigmoid forms per value elsewhere (same products, same
    // bits, formed once per channel instead of once per pixel); the workspace tables keep the plain (scale, shift)
    auto fold_group = [&](const float *partial, int ngroups, int grp, const float *gam, const float *bet, int ch0, float *ss_out, float *st_out,
                          int status_bit, bool l2e) __attribute__((always_inline)) {
#pragma unroll
        for (int ti = 0; ti < CT_NT; ++ti) {
            if (!son[ti]) continue;
            float *tab = sstab + (size_t)ti * 3 * G * 64;
            if (ti > 0 && sb[ti] == sb[ti - 1]) {                       // same sample as the slot before: copy its table entries
                if (lane_b < 32) {
                    const float *prev = sstab + (size_t)(ti - 1) * 3 * G * 64;
                    tab[2 * (ch0 + lane_b)] = prev[2 * (ch0 + lane_b)];
                    tab[2 * (ch0 + lane_b) + 1] = prev[2 * (ch0 + lane_b) + 1];
                }
                continue;
            }
            const int bs = sb[ti];
            // two tiles per 16-byte write-through-coherent load (sc1, as the 8-byte agent-scope loads of consume8), eight loads in flight per
            // lane: ONE round trip for up to 1024 tiles (the 8-byte form in two dependent batches cost 6.7 us here).  Lane l takes tile pairs
            // l, l + 64, ... in ascending order, then the xor butterfly: a fixed order.
            double s1 = 0.0, s2 = 0.0;
            {
                const int nt = cp.tilesPerSample, tpart = (P & 63) != 0 ? P / 64 : -1;
                const ct_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(partial), 0, (int)0x7fffffff, 0x00020000);
                const unsigned base = (unsigned)((((size_t)bs * ngroups + grp) * nt) * 8);
                for (int u0 = 0; 2 * u0 < nt; u0 += 64 * 8) {
                    u32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int u = u0 + 64 * k + lane_b;
                        v[k] = __builtin_amdgcn_raw_buffer_load_b128(rp, 2 * u < nt ? base + 16u * (unsigned)u : base, 0, 16);      // aux 16 = sc1
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int u = u0 + 64 * k + lane_b;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int t = 2 * u + e;
                            const float a = __uint_as_float(v[k][2 * e]), m2 = __uint_as_float(v[k][2 * e + 1]);
                            if (t < nt) {
                                const double dx = (double)a;
                                s1 += dx;
                                s2 += t == tpart ? tile_x2(a, m2, 32 * (P - tpart * 64)) : (double)m2 + dx * dx * (1.0 / 2048.0);   // (full tile: 32 x 64 values)
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                s1 += __shfl_xor(s1, m, 64);
                s2 += __shfl_xor(s2, m, 64);
            }
            const double count = 32.0 * (double)P;
            const double mean = s1 / count;
            double var = s2 / count - nofma(mean * mean);   // (no contraction: every finalizer gives the same bits)
            var = var > 0.0 ? var : 0.0;
            const double rstd = 1.0 / sqrt(var + (double)cp.eps);
            int t0, b0;
            slot_tile(ti, b0, t0);                          // the block that holds a sample's first tile leaves the workspace tables
            if (lane_b < 32) {
                const int c = grp * 32 + lane_b;
                const double sc = (double)gam[c] * rstd;
                const float fsc = (float)sc, fsh = (float)((double)bet[c] - nofma(mean * sc));
                const float L2E = 1.44269502162933349609375f;       // (urnn_common.h gate_sigmoid)
                tab[2 * (ch0 + lane_b)] = l2e ? nofma(fsc * L2E) : fsc;
                tab[2 * (ch0 + lane_b) + 1] = l2e ? nofma(fsh * L2E) : fsh;
                if (t0 == 0 && ss_out) {
                    ss_out[((size_t)bs * ngroups * 32 + c) * 2] = fsc;
                    ss_out[((size_t)bs * ngroups * 32 + c) * 2 + 1] = fsh;
                }
            }
            if (t0 == 0 && lane_b == 0) {
                flag_nonfinite(cp.status, status_bit, s1, s2);
                if (st_out) {
                    st_out[((size_t)bs * ngroups + grp) * 2] = (float)mean;
                    st_out[((size_t)bs * ngroups + grp) * 2 + 1] = (float)rstd;
                }
            }
        }
    };
    const float *mytab = sstab + (size_t)wc * 3 * G * 64;
    float *zbuf = reinterpret_cast<float *>(ring) + (size_t)wc * (G * 2 * 1024);       // phase C: z, [slot][blk][pb][16][64] over the ring
    float *hbuf = reinterpret_cast<float *>(wbuf) + (size_t)wc * (G * 1024);           //          h of one pixel half_b, [slot][blk][16][64] over the weight buffers
    const int NHh = NH;

    // From here on the three roles run their own code (so that a role's dead accumulators are dead for the register allocator too); every
    // path passes the SAME sequence of block barriers: Sa Sb Sc Sd Se Sf Sg Sh Si Sj.
    if (wr == 0) {
        // =============================================== z-waves ===================================================================
        finish_and_stats(0, cp.partial1, 2 * G);
        ct_sync();                                                                  // Sa
        CT_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the statistics have been acknowledged
        ct_sync();                                                                  // Sb
        if (threadIdx.x == 0) ct_barrier_lane(cp.bar, blockIdx.x, (unsigned)cp.nblocks, cp.status);       // grid barrier 1
        ct_sync();                                                                  // Sc
        CT_STAMP(4);
        if (wc < G) fold_group(cp.partial1, 2 * G, wc, cp.gn1_w, cp.gn1_b, wc * 32, cp.ss1_out, cp.st1_out, URNN_STATUS_GATES, true);
        ct_sync();                                                                  // Sd
        CT_STAMP(5);
        if (tile_on) {                                                              // z = sigmoid(GN(raw)) in place, next to the r-waves' gating
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x2 st = *reinterpret_cast<const f32x2 *>(mytab + 2 * (blk * 32 + row_c(r) + 4 * half_b));
                        acc[blk][pb][r] = sigmoid_of_log2arg(fmaf(acc[blk][pb][r], st.x, st.y));      // (= gate_sigmoid: the table holds the folded affine)
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        ct_sync();                                                                  // Se
        CT_STAMP(6);
        CT_STAMP(7);
        ct_sync();                                                                  // Sf: the ring is dead
        if (tile_on) {
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) zbuf[((blk * 2 + pb) * 16 + r) * 64 + lane_b] = acc[blk][pb][r];
        }
        ct_sync();                                                                  // Sg
        CT_STAMP(8);
        ct_sync();                                                                  // Sh
        CT_STAMP(9);
        ct_sync();                                                                  // Si
        ct_sync();                                                                  // Sj
    } else if (wr == 1) {
        // =============================================== r-waves ===================================================================
        finish_and_stats(1, cp.partial1, 2 * G);
        ct_sync();                                                                  // Sa
        CT_STAMP(3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ct_sync();                                                                  // Sb
        ct_sync();                                                                  // Sc
        CT_STAMP(4);
        if (wc < G) fold_group(cp.partial1, 2 * G, G + wc, cp.gn1_w, cp.gn1_b, (G + wc) * 32, cp.ss1_out, cp.st1_out, URNN_STATUS_GATES, true);
        ct_sync();                                                                  // Sd
        CT_STAMP(5);
        // r (.) h in place over the hidden-state rows the ring still holds (its last NH groups): one dword = the value's two f16 pieces
        if (tile_on) {
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        // accumulator rows 4 q + i, i < 4: hidden channels 32 blk + 8 q + 4 half_b + i: group 2 blk + (q >> 1), rows 8 (q & 1) + 4 half_b + i
                        const int gi = 2 * blk + (q >> 1);
                        char *base = ring + ((KG - NHh + gi) % CT_D) * CT_SLOT + wc * 4096 + (8 * (q & 1) + 4 * half_b) * 256 + (pb * 32 + j_b) * 4;
                        float hv[4], rv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) hv[i] = *reinterpret_cast<const float *>(base + i * 256);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const f32x2 st = *reinterpret_cast<const f32x2 *>(mytab + 2 * ((G + blk) * 32 + 8 * q + 4 * half_b + i));
                            rv[i] = hv[i] * sigmoid_of_log2arg(fmaf(acc[blk][pb][4 * q + i], st.x, st.y));
                        }
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            unsigned ph, pl;
                            split2_pair(rv[i], rv[i + 1], URNN_F16_ASCALE, ph, pl);       // ph = {hi(i+1), hi(i)}, pl = {lo(i+1), lo(i)}
                            *reinterpret_cast<unsigned *>(base + i * 256) = __builtin_amdgcn_perm(pl, ph, 0x05040100u);          // {lo(i), hi(i)}
                            *reinterpret_cast<unsigned *>(base + (i + 1) * 256) = __builtin_amdgcn_perm(pl, ph, 0x07060302u);    // {lo(i+1), hi(i+1)}
                        }
                        if (q == 3) __builtin_amdgcn_sched_barrier(0);      // (16 independent chains per step)
                    }
        }
        ct_sync();                                                                  // Se
        CT_STAMP(6);
        // the tile's previous state for the blend: all of it into this wave's (now free) registers, one round trip, while the c-waves multiply
        float hreg[G][2][16];
        const float *hbase = cp.h + ((size_t)tb * F + 4 * half_b) * P + (size_t)tt * 64 + j_b;
#pragma unroll
        for (int blk = 0; blk < G; ++blk)
#pragma unroll
            for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                for (int r = 0; r < 16; ++r) hreg[blk][pb][r] = okp[pb] ? hbase[(size_t)(blk * 32 + row_c(r)) * P + pb * 32] : 0.f;
        CT_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ct_sync();                                                                  // Sf: the weight buffers are dead
        if (tile_on) {
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) hbuf[(blk * 16 + r) * 64 + lane_b] = hreg[blk][0][r];
        }
        ct_sync();                                                                  // Sg
        CT_STAMP(8);
        ct_sync();                                                                  // Sh
        CT_STAMP(9);
        ct_sync();                                                                  // Si: the c-waves are through with the first pixel half_b
        if (tile_on) {
#pragma unroll
            for (int blk = 0; blk < G; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) hbuf[(blk * 16 + r) * 64 + lane_b] = hreg[blk][1][r];
        }
        ct_sync();                                                                  // Sj
    } else {
        // =============================================== c-waves ===================================================================
        ct_sync();                                                                  // Sa: everybody is past the k-loop's last weight read
        {   // W2[:, h] (NH groups x G blocks x two pieces), once per CU, into the weight buffers
            const int total = NHh * G * 2;                 // 1-KB pieces: [group][block][piece]
            for (int n = wc; n < total; n += 4) {
                const int gi = n / (2 * G), rem = n - gi * 2 * G, blk = rem >> 1;
                ct_dma16(cp.wblk[2 * G + blk] + (size_t)(cp.kg0 + KGxe + gi) * cp.wstride[2 * G + blk] + (rem & 1) * 256 + lane_b * 4, wbuf + n * 1024);
            }
        }
        CT_STAMP(3);
        ct_sync();                                                                  // Sb
        ct_sync();                                                                  // Sc
        CT_STAMP(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // W2[:, h] has landed
        ct_sync();                                                                  // Sd
        CT_STAMP(5);
        ct_sync();                                                                  // Se: r (.) h is in the ring
        CT_STAMP(6);
        if (tile_on) {
            for (int gi = 0; gi < NHh; ++gi) {
                const char *slot = ring + ((KG - NHh + gi) % CT_D) * CT_SLOT;
                unsigned ph[2][4], pl[2][4];
#pragma unroll
                for (int pb = 0; pb < 2; ++pb)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned a = *reinterpret_cast<const unsigned *>(slot + boff_b + (4 * d) * 256 + pb * 128);
                        const unsigned b = *reinterpret_cast<const unsigned *>(slot + boff_b + (4 * d + 2) * 256 + pb * 128);
                        ph[pb][d] = __builtin_amdgcn_perm(b, a, 0x05040100u);      // {b.lo16, a.lo16}: the hi pieces
                        pl[pb][d] = __builtin_amdgcn_perm(b, a, 0x07060302u);      // {b.hi16, a.hi16}: the lo pieces
                    }
#pragma unroll
                for (int blk = 0; blk < G; ++blk) mfma_block(wbuf + gi * G * 2048, blk, ph, pl);
            }
        }
        finish_and_stats(2, cp.partial2, G);
        CT_STAMP(7);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the statistics have been acknowledged
        ct_sync();                                                                  // Sf
        if (threadIdx.x == 64 * 8) ct_barrier_lane(cp.bar, blockIdx.x, (unsigned)cp.nblocks, cp.status);     // grid barrier 2
        ct_sync();                                                                  // Sg: z and the first half_b of h are in LDS
        CT_STAMP(8);
        if (wc < G) fold_group(cp.partial2, G, wc, cp.gn2_w, cp.gn2_b, (2 * G + wc) * 32, cp.ss2_out, cp.st2_out, URNN_STATUS_CAND, false);
        ct_sync();                                                                  // Sh
        CT_STAMP(9);
        float *obase = cp.h_out + ((size_t)tb * F + 4 * half_b) * P + (size_t)tt * 64 + j_b;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            if (tile_on) {
#pragma unroll
                for (int blk = 0; blk < G; ++blk) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const f32x2 st = *reinterpret_cast<const f32x2 *>(mytab + 2 * ((2 * G + blk) * 32 + row_c(r) + 4 * half_b));
                        const float z = zbuf[((blk * 2 + pb) * 16 + r) * 64 + lane_b];
                        const float hp = hbuf[(blk * 16 + r) * 64 + lane_b];
                        const float n = tanhf_fast(fmaf(acc[blk][pb][r], st.x, st.y));
                        if (okp[pb]) obase[(size_t)(blk * 32 + row_c(r)) * P + pb * 32] = gru_blend(z, n, hp);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            ct_sync();                                                              // Si, Sj
        }
    }
#ifdef URNN_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CT_STAMP(10);
#endif
}

template <int G>
__global__ __launch_bounds__(768) void coop_tiles_kernel(const CoopTilesParams cp)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave >> 2;
    if (role == 0) ct_role<G, 0>(cp, lane, wave);
    else if (role == 1) ct_role<G, 1>(cp, lane, wave);
    else ct_role<G, 2>(cp, lane, wave);
}

// ---- host side -----------------------------------------------------------------------------------------------------------------------------
static size_t ct_lds_bytes(int G) { return (size_t)CT_D * CT_SLOT + (size_t)CT_NW * CT_WBUF + (size_t)3 * G * 32 * 4; }

// Does a cell of this shape take the multi-tile cooperative launch?  p / c: the gate / candidate parameter blocks as gru_cell_impl builds them.
// Returns the number of blocks (0: no).
int urnn_coop_tiles_blocks(const ConvGemmParams &p, const ConvGemmParams &c, int B)
{
    static const int on = (int)urnn_tune("URNN_TUNE_COOP_TILES", 1);   // development knob (A/B)
    if (!on) return 0;
    const int mm = urnn_get_matrix_mode();
    if (mm != URNN_MATRIX_FP32 && mm != URNN_MATRIX_FP32_CAND) return 0;              // the f16-piece arithmetic only
    const int F = p.F, G = F / 32;
    if (G != 2 && G != 3) return 0;
    if (!p.wf16 || p.fDwords <= 0 || !p.biasf || !c.wf16 || c.fDwords <= 0) return 0;
    if (p.P % 4 != 0 || p.KT % 8 != 0 || p.kpBegin % 8 != 0) return 0;
    for (int s = 0; s < 3; ++s)
        if (p.segKp0[s] != INT_MAX && (p.segKp0[s] % 8 != 0 || p.segC[s] % 16 != 0)) return 0;
    if (c.hKp0 % 8 != 0 || (p.KT - c.hKp0) * 2 != F || c.hKp0 <= p.kpBegin) return 0;   // x | e first, at least one plain group
    const int KG = (p.KT - p.kpBegin) / 8;
    if (KG < CT_D) return 0;                                                         // (the ring's prologue assumes K >= 96)
    if (2 * G > CT_D) return 0;                                                      // the hidden state must fit the ring
    const long tiles = (long)B * ((p.P + 63) / 64);
    const int cus = urnn_device_cus();
    const long blocks = (tiles + CT_NT - 1) / CT_NT;
    if (tiles <= cus) return 0;                                                      // small planes: one tile per block (urnn_small.hip)
    if (blocks > cus) return 0;                                                      // every block must be resident, one per CU
    if (ct_lds_bytes(G) > 160 * 1024) return 0;
    return (int)blocks;
}

hipError_t urnn_launch_coop_tiles(const ConvGemmParams &p, const ConvGemmParams &c, const float *gn2_w, const float *gn2_b, float *ss2_out, float *st2_out,
                                  const float *h, float *h_out, unsigned *bar, int B, hipStream_t st)
{
    const int F = p.F, G = F / 32;
    CoopTilesParams cp = {};
    const int nseg = p.segKp0[2] != INT_MAX ? 3 : 2;
    for (int s = 0; s < 3; ++s) {
        const bool used = s < nseg;
        cp.seg[s] = used ? p.seg[s] : p.seg[0];
        cp.segC[s] = used ? p.segC[s] : 0;
        cp.segG0[s] = used ? p.segKp0[s] / 8 : INT_MAX;
        const double bytes = used ? (double)B * p.segC[s] * p.P * 4.0 : 0.0;
        cp.segBytes[s] = bytes > 4294967295.0 ? 0xffffffffu : (unsigned)bytes;
    }
    cp.kg0 = p.kpBegin / 8;
    cp.KG = (p.KT - p.kpBegin) / 8;
    cp.KGxe = c.hKp0 / 8 - cp.kg0;
    // canonical gate block cb of [z_0 .. | r_0 ..] -> its place (g, nb) in the grouped f16 slab
    for (int g = 0; g < p.NGf; ++g)
        for (int nb = 0; nb < p.NBf; ++nb) {
            const int cb = urnn_gate_cb(p.gHalves, p.gGS, G, g, nb);
            cp.wblk[cb] = p.wf16 + (size_t)g * p.fDwords + (size_t)nb * 512;
            cp.wstride[cb] = p.NBf * 512;
            cp.gbiasOff[cb] = (g * p.NBf + nb) * 32;
        }
    const int cNB = urnn_cand_nb(F);
    for (int ci = 0; ci < G; ++ci) {
        cp.wblk[2 * G + ci] = c.wf16 + (size_t)(ci / cNB) * c.fDwords + (size_t)(ci % cNB) * 512;
        cp.wstride[2 * G + ci] = cNB * 512;
    }
    cp.gbias = p.biasf;
    cp.cbias = c.bias;
    cp.gn1_w = c.gn_w; cp.gn1_b = c.gn_b; cp.gn2_w = gn2_w; cp.gn2_b = gn2_b;
    cp.eps = c.eps;
    cp.h = h; cp.h_out = h_out;
    cp.partial1 = p.partial; cp.partial2 = c.partial;
    cp.ss1_out = c.ss_out; cp.ss2_out = ss2_out;
    cp.st1_out = c.stat_out; cp.st2_out = st2_out;
    cp.bar = bar;
    cp.status = p.status;
    cp.B = B; cp.P = p.P; cp.F = F;
    cp.tilesPerSample = (p.P + 63) / 64;
    cp.totalTiles = B * cp.tilesPerSample;
    cp.nblocks = (cp.totalTiles + CT_NT - 1) / CT_NT;
    const size_t lds = ct_lds_bytes(G);
    static bool raised = false;
    if (!raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(coop_tiles_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void *>(coop_tiles_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        raised = true;
    }
    if (G == 2) hipLaunchKernelGGL(coop_tiles_kernel<2>, dim3(cp.nblocks), dim3(768), lds, st, cp);
    else hipLaunchKernelGGL(coop_tiles_kernel<3>, dim3(cp.nblocks), dim3(768), lds, st, cp);
    return hipGetLastError();
}
