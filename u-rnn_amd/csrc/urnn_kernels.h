// urnn_kernels.h -- kernel parameter blocks and internal launchers shared between the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/urnn_hip.h"

// Development knobs (URNN_TUNE_* environment variables: A/B switches, forced tile shapes, ...) exist in TUNING builds only
// (-DURNN_TUNING; tools build one next to the product library and select it with URNN_LIB).  The product library reads no
// environment variable: every knob is its default.
#include <stdlib.h>
static inline long urnn_tune(const char *name, long dflt)
{
#ifdef URNN_TUNING
    const char *e = getenv(name);
    return e ? atol(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Compute units of the current device (queried once): the cooperative launches need every block resident at once, one per CU, so
// their block limits are bounded by what THIS device has (256 on an MI355X; fewer on a partitioned or masked one).
static inline int urnn_device_cus()
{
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

#define URNN_FULL_RES_PIXELS 100000   // URNN_MATRIX_FP32_CAND: planes at least this large per sample count as full resolution
enum { MAP_VEC = 0, MAP_PAIR = 1, MAP_STRIDED = 2, MAP_POOL = 3, MAP_PAIR16 = 4, MAP_QUAD16 = 5 };   // pixel geometry of a wave tile (urnn_gemm.hip)
enum { EPI_LRELU = 0, EPI_POOL = 1, EPI_DECONV = 2, EPI_GRU1 = 3, EPI_CAND = 4 };   // epilogue of conv_gemm_kernel

struct ConvGemmParams {
    const float *seg[3];  // up to three channel-concatenated inputs (x | e | h)
    int segC[3];          // real channel counts
    int segKp0[3];        // first absolute k-pair of each segment in the packed weights (unused segments: INT_MAX)
    int kpBegin, KT;      // k-pair range to run: [kpBegin, KT)  (kpBegin > 0 skips an all-zero x segment)
    int hKp0;             // EPI_CAND: first k-pair of the hidden-state segment (its rows are gated by the reset gate)
    const float *gate;    // EPI_CAND: raw gates (B,2F,P); rows F..2F-1 are the reset gate
    const float *gpart;   // EPI_CAND: the gate GEMM's GroupNorm partials [B][2F/32][gtiles][2]; finalised in this kernel's prologue
    int gtiles;           //           tiles per sample of the gate GEMM
    int gtilePix;         //           pixels per gate-GEMM tile (centred partials, urnn_common.h tile_x2); 0: raw partials (strip mode)
    double gcount;        //           values per (sample, group) = 32 * P
    const float *gn_w, *gn_b;   //     GroupNorm affine of the gates [2F]
    float eps;
    float *stat_out;      // EPI_CAND: out -- per (sample, gate norm group) (mean, rstd) [B][2F/32][2] for the backward pass (block 0)
    float *ss_out;        // EPI_CAND: out -- gates' per-channel (scale, shift) [B][2F][2] for the blend kernel (written by block 0)
    int B;                // EPI_CAND: samples (size of the reset-gate scale/shift table kept in LDS)
    const float *wt;      // packed weights [NG][KT][NB][64] (group stride aFloats); lane l of row (kp, nb) holds
                          // W[k = 2*kp + (l >> 5)][n = (g*NB + nb)*32 + (l & 31)]
    const unsigned *wsplit;   // the same weights split into bf16 pieces (urnn_common.h urnn_split_slab_dwords), group stride sDwords
    int sDwords;          // dwords per n-group split slab (a multiple of 256); 0: no split form (fp32 k-loop only)
    const unsigned *wf16;     // the same weights x 2^URNN_F16_WEXP as two f16 pieces (urnn_f16_slab_dwords), group stride fDwords
    int fDwords;          // dwords per n-group f16 slab; 0: no f16 form
    int wide;             // 1: the activations are gradients (unbounded exponent range): bf16 x 6 split instead of f16 x 3
    // EPI_GRU1, f16 form only: column grouping of the gate GEMM (urnn_gate_groups): the f16 slab holds NGf groups of NBf n-blocks,
    // block nb of group g being canonical block cb = urnn_gate_cb(...) of [z_0 .. z_{G-1} | r_0 .. r_{G-1}]; biasf = the bias in
    // that packed order.  The fp32 / bf16 slabs (wt, wsplit, bias) keep F/32 groups of (z_i | r_i).
    int NGf, NBf, gHalves, gGS;
    const float *biasf;
    const float *bias;    // bias per packed column [NG*NB*32]
    int aFloats;          // floats per n-group slab, a multiple of 256 (one LDS-DMA instruction moves 256 floats)
    int NG;               // number of n-groups (blocks are specialised per group)
    int P, W, P2, W2;     // input plane size / width; pooled plane size / width (EPI_POOL)
    int tilesPerSample, totalTiles;
    float invFull, invTail;   // EPI_GRU1 / EPI_CAND: 1 / (32 * pixels) of a full tile / of the sample's last tile (tile means of the partials)
    int Cout, F;
    float slope;
    float *out0;
    int stagger;          // 8-wave blocks: start offset of the second wave of each SIMD in naps of s_sleep 64 (conv_gemm_kernel; the launcher's rule) / in 16-k groups (cand_fused_kernel)
    float *partial;       // EPI_GRU1: [B][2F/32][tiles][2]; EPI_CAND: [B][F/32][tiles][2]
    // fused-reset-gate cell (URNN_PHASE_FUSED_R): the gate GEMM keeps the reset gate's statistics but does not store its planes
    // (zOnly), cand_fused_kernel recomputes it from its own slab: phase-1 [W1 r rows | W2 x,e columns], phase-2 W2[:, h] in the
    // accumulator-row channel order (urnn_fused_cand_layout), bias [b1 r | b2]
    int zOnly;
    int candExact;        // EPI_CAND: run the exact fp32 MFMA k-loop whatever the mode (a full-resolution cell of a rollout that cannot take the fused kernel)
    const unsigned *wfused;
    int fu1Dwords, fu2Dwords;
    const float *biasfu;
    int *status;          // the workspace's status word (urnn_common.h flag_nonfinite); may be NULL
    // EPI_LRELU to 16 channels on 128-pixel tiles (the decoder's last conv): the head's stem conv (16 x 16, row-major) applied to the
    // tile in the epilogue and the partial statistics of the head's first LayerNorm [B][tilesPerSample][2] -- head_k1's pass over the
    // feature map done where the map is made (flood_head.py:131-140); NULL: off
    const float *stemW;
    float *stemPart;
    int abl;              // tuning builds (-DURNN_TUNING): ablation mask URNN_TUNE_ABL -- phases of cand_fused_kernel skipped for timing (wrong results)
};

// Packed layout of the fused candidate slab (appended to a cell's packed buffer when ok): nXE 16-k groups of x | e with 2 NBF blocks
// (r_0 .. | c_0 ..), nH = F/16 groups of h with NBF blocks (r only), then nH groups (accumulator block rb, half q) of W2[:, h] with NBF
// candidate blocks; every block = two f16 pieces x 64 lanes x 16 B (urnn_common.h).  Built for the full-resolution cells' F = 64.
struct FusedCandLayout { int ok, NBF, nXE, nH, dw1, dw2; };
__host__ __device__ static inline FusedCandLayout urnn_fused_cand_layout(int I, int F, int skip)
{
    FusedCandLayout L = {0, F / 32, 0, 0, 0, 0};
    const int Ie = (I + 1) & ~1, kH = (Ie + (skip ? F : 0)) / 2;      // k-pairs in front of the hidden-state segment
    if (F != 64 || kH % 8 != 0) return L;
    L.ok = 1;
    L.nXE = kH / 8;
    L.nH = F / 16;
    L.dw1 = L.nXE * (2 * L.NBF * 512) + L.nH * (L.NBF * 512);
    L.dw2 = L.nH * (L.NBF * 512);
    return L;
}
int urnn_cand_fused_plan(const ConvGemmParams &p, int B);            // ring depth the fused kernel would run with; 0: not eligible
hipError_t urnn_launch_cand_fused(ConvGemmParams p, int B, hipStream_t st);

// Gate GEMM column grouping of the f16 slab.  A wave that owns more of the 2F gate columns reads the K input planes fewer times
// through the CU's load path (the gate GEMM with F/32 groups of z_i|r_i was bound by that path, not by HBM: dec1 moved 576 MB
// through the CUs for 352 MB of HBM traffic).  halves = 0: group g = the pairs (z_i, r_i), i in [g*GS, (g+1)*GS), NB = 2*GS;
// halves = 1: group 0 = all z blocks, group 1 = all r blocks, NB = F/32.  Depends on (F, KT) only: packing and launch agree.
struct GateGroups { int NB, NG, halves, GS; };
GateGroups urnn_gate_groups(int F, int KT);
__host__ __device__ static inline int urnn_gate_cb(int halves, int GS, int G, int g, int nb)
{
    return halves ? g * G + nb : (nb & 1) * G + g * GS + (nb >> 1);
}
// which gate kernel a launch will take: returns 1 (and the tile shape) when the grouped f16 kernel runs, 0 for the F/32-group one
int urnn_gate_plan(const ConvGemmParams &p, int B, int pb_legacy, int map_legacy, int *pb, int *map);

int urnn_conv_nb(int Cout);   // n-blocks per wave for a Cout-wide 1x1 conv (packing and launch must agree)
int urnn_conv_ng(int Cout);   // number of n-groups (the last one may be padded with zero columns)
hipError_t urnn_launch_conv_flat(ConvGemmParams p, int B, int PB, int map, hipStream_t st);
hipError_t urnn_launch_conv_pool(ConvGemmParams p, int B, hipStream_t st);
hipError_t urnn_launch_deconv(ConvGemmParams p, int B, int PB, int map, hipStream_t st);
hipError_t urnn_launch_gru1(ConvGemmParams p, int B, int PB, int map, hipStream_t st);   // PB / map as urnn_gate_plan returned them
// small planes (urnn_small.hip): activation-stationary gate / candidate GEMMs; same outputs and 32-pixel partial tiles as the
// regular kernels with PB = 1
bool urnn_small_ok(const ConvGemmParams &p, int nblk_total, int gated);
hipError_t urnn_launch_small_gates(ConvGemmParams p, int B, hipStream_t st);
hipError_t urnn_launch_small_cand(ConvGemmParams p, int B, hipStream_t st);
// the whole cell of a small plane as one cooperative launch (urnn_small.hip coop_cell_kernel); p / c = the gate / candidate blocks
bool urnn_coop_cell_ok(const ConvGemmParams &p, const ConvGemmParams &c, int B);
hipError_t urnn_launch_coop_cell(ConvGemmParams p, const ConvGemmParams &c, const float *gn2_w, const float *gn2_b, float *ss2_out, float *h_out,
                                 unsigned *bar, int B, hipStream_t st);
// the whole cell of a HALF-RESOLUTION plane as one cooperative launch, four 64-pixel tiles per block, gates and candidate resident in the
// accumulators (urnn_coop_tiles.hip); blocks: how many blocks that launch takes (0: the shape does not qualify)
int urnn_coop_tiles_blocks(const ConvGemmParams &p, const ConvGemmParams &c, int B);
hipError_t urnn_launch_coop_tiles(const ConvGemmParams &p, const ConvGemmParams &c, const float *gn2_w, const float *gn2_b, float *ss2_out, float *st2_out,
                                  const float *h, float *h_out, unsigned *bar, int B, hipStream_t st);
int urnn_cand_nb(int F);      // n-blocks per group of the candidate GEMM
// the two-stream candidate of a half-resolution plane on 64-pixel tiles and a group-wise ring (urnn_cand_gated.hip); plan: 1 when it
// takes the launch -- the candidate's GroupNorm partials are then per 64-pixel tile
int urnn_cand_gated_plan(const ConvGemmParams &p, int B);
hipError_t urnn_launch_cand_gated(ConvGemmParams p, int B, hipStream_t st);
hipError_t urnn_launch_cand(ConvGemmParams p, int B, int PB, int map, hipStream_t st);

// ---- elementwise / reduction kernels (urnn_elem.hip) ----
hipError_t urnn_launch_gn_finalize(const float *partial, int ntiles, int tile_pix, int P, double count, const float *gamma, const float *beta,
                                   float eps, float *ss, float *stat, int B, int C, int *status, int status_bit, hipStream_t st);
hipError_t urnn_launch_blend(const float *g1, const float *c, const float *h, const float *ss1, const float *ss2, float *out,
                             int B, int F, int P, hipStream_t st);
hipError_t urnn_launch_blend_fin(const float *g1, const float *c, const float *h, const float *ss1, float *out, int B, int F, int P,
                                 const float *partial, int ntiles, int tile_pix, double count, const float *gamma, const float *beta, float eps,
                                 float *ss2, float *stat2, int *status, hipStream_t st);

// The end of a cell fused with the 1x1 conv that consumes the new state (urnn_tail.hip blend_conv_kernel)
struct TailParams {
    const float *g1, *cx, *h;          // raw gates (B,2F,P; z = the first F planes), raw candidate (B,F,P), previous state
    float *h_out;
    const float *ss1;                  // [B][2F][2] (scale, shift) of the gates, from the candidate kernel's prologue
    const float *partial2;             // candidate GroupNorm partials [B][F/32][ntiles2][2]
    int ntiles2, tile_pix2;
    double count;
    const float *gn2_w, *gn2_b;
    float eps;
    float *ss2_out, *stat2_out;
    int *status;
    int B, F, P, W;
    const unsigned *wf16;              // the consumer conv's packed f16 slab, its group stride, blocks per group, output blocks
    int fDwords, NBc, NBO, Cout, wDwords;   // wDwords: all groups' dwords
    const float *bias;
    float slope;
    int P2, W2;                        // pooled plane (TAIL_POOL)
    float *out;
    const float *head_w;               // TAIL_FLAT: the head's stem conv (16 x 16) -> LayerNorm statistics of u0 = Ws . out; may be NULL
    float *partial0;                   //            [B][blocksPerSample][2], centred per 128-pixel block
    int blocksPerSample;
    int chunk;                         // persistent blocks: consecutive tiles per block (0: tiles strided by the grid)
};
bool urnn_tail_ok(int B, int F, int H, int W, int Cin, int Cout, int pool);
hipError_t urnn_launch_tail(TailParams tp, int H, int pool, hipStream_t st);

struct HeadParams {
    const float *feat;
    const float *conv_w;        // 5 x C x C
    const float *ln_w, *ln_b;   // 5 x C x P
    const float *cls_w, *cls_b, *reg_w, *reg_b;
    float *out_masked, *out_cls, *out_raw;
    const int *frame_index;
    float *u1, *u2;             // workspace activations (B,C,P) each: cls / reg branch pre-norm values
    float *partial;             // [5][B][nblk][2]
    float *stats;               // [5][B][2] mean, rstd
    int B, C, P, nblk;
    long Pglobal;               // > 0: strip mode, LayerNorm statistics over Pglobal pixels from two pseudo-blocks of partials
    float cls_thred, eps, slope;
    int *status;                // the workspace's status word (urnn_common.h flag_nonfinite)
    // the first LayerNorm's partial statistics came from the kernel that produced feat (urnn_tail.hip): head_k1 is skipped and head_k2
    // folds them from here: [B][nblk0][2], centred per block of bpix0 pixels
    const float *partial0;
    int nblk0, bpix0;
    int *bump;                  // rollouts: where the head's first launch stores *frame_index + 1 (the next head's frame word; see urnn_head_rollout_f32)
};
hipError_t urnn_launch_head(const HeadParams &p, int phase_mask, hipStream_t st);
int urnn_head_coop_blocks(int B, int P);
hipError_t urnn_launch_head_coop(const HeadParams &p, unsigned *bar, hipStream_t st);
int urnn_head_nblk(int P);        // blocks allocated per (norm, sample) in the partial buffer
int urnn_head_nblk_used(int P);   // blocks the head kernels write
hipError_t urnn_launch_stats_reduce(const float *partial, int rows, int stride, int ntiles, int tile_pix, int P, int chans, double *sums,
                                    hipStream_t st);
int urnn_head_block_pix(int P);   // pixels per block of the head kernels (the unit of their LayerNorm partials)
hipError_t urnn_launch_stats_scatter(const double *sums, int rows, int stride, float *partial, hipStream_t st);

hipError_t urnn_launch_preprocess(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                                  const float *manhole, float dem_min, float dem_max, float *out, int t, const int *t_dev,
                                  int B, int T, int nums, int P, int spatial, float rain_max, float cumsum_max,
                                  hipStream_t st, int *bump = nullptr);
hipError_t urnn_launch_advance(int *counter, int delta, hipStream_t st);
hipError_t urnn_launch_max_abs(const float *v, long n, float *out, hipStream_t st);
hipError_t urnn_launch_stage1_static(const float *dem, const float *imperv, const float *manhole, float dem_min, float dem_max,
                                     const float *w, float *S, int B, int nums, int Cout, int P, hipStream_t st);
hipError_t urnn_launch_stage1_scalar(const float *S, const float *rain, const float *cumsum, const float *w, const float *bias,
                                     float *out, int t, const int *t_dev, int B, int T, int nums, int Cout, int P, float rain_max,
                                     float cumsum_max, float slope, hipStream_t st, int *bump = nullptr);

hipError_t urnn_launch_pack_conv(const float *w, const float *bias, float *packed, int Cin, int Cout, hipStream_t st);
hipError_t urnn_launch_pack_gru(const float *W1, const float *b1, const float *W2, const float *b2, float *packed, int I,
                                int F, int skip, hipStream_t st);
size_t urnn_packed_gru_fused_floats(int I, int F, int skip);
size_t urnn_packed_gru_total(int I, int F, int skip);   // floats of a packed cell buffer (layout: urnn_elem.hip pack_gru_kernel)
hipError_t urnn_launch_pack_deconv(const float *w, const float *bias, float *packed, int Cin, int Cout, hipStream_t st);

// ---- training building blocks (urnn_train.hip) ----
int urnn_train_nchunk(int P);
hipError_t urnn_train_chan_sums(const float *a, long a_bs, const float *v, long v_bs, const float *stat, int B, int C, int P,
                                float *partial, double *sums, hipStream_t st);
hipError_t urnn_train_gn_backward(float *dy, const float *v, const float *stat, const float *gamma, int B, int C, int P, float *partial,
                                  double *sums, float *coef, float *dgamma, float *dbeta, int accumulate, int have_partials, hipStream_t st);
hipError_t urnn_train_blend_bwd(const float *dout, const float *dout2, const float *dout3, const float *dout4, const float *g1, const float *c,
                                const float *h, const float *ss1, const float *ss2,
                                const float *st1, const float *st2, float *dy2, float *dy1, float *dh, float *part1, float *part2, int B,
                                int F, int P, hipStream_t st);
hipError_t urnn_train_reset_gate(const float *g1, const float *h, const float *ss1, float *rh, int B, int F, int P, hipStream_t st);
hipError_t urnn_train_reset_gate_bwd(const float *drh, long drh_bs, const float *g1, const float *h, const float *ss1, const float *st1,
                                     float *dy1, float *dh, float *part1, int B, int F, int P, hipStream_t st);
hipError_t urnn_train_add_slices(float *out, long out_bs, const float *a, long a_bs, const float *a2, long a2_bs, int B, int C, int P,
                                 int accumulate, hipStream_t st);
hipError_t urnn_train_zero(float *p, size_t n, hipStream_t st);
size_t urnn_train_wgrad_partial_floats(int B, int N, int K, int P);
hipError_t urnn_train_wgrad(const float *dy, const float *const seg[3], const int segC[3], int B, int N, int K, int P, float *partial,
                            float *dW, float *db, int accumulate, hipStream_t st);
hipError_t urnn_train_transpose(const float *w, float *wt, int N, int K, hipStream_t st);
hipError_t urnn_train_cell_bwd_weights(const float *W1, const float *W2, float *out, int F, int K, int rlo, int nrows, int mode, hipStream_t st);
hipError_t urnn_train_lrelu_pool_bwd(float *u, const float *dy, int B, int C, int H, int W, int pool, float slope, hipStream_t st);
hipError_t urnn_train_deconv_unshuffle(const float *dy, const float *y, float *d4, int B, int Cout, int H, int W, float slope, hipStream_t st);
hipError_t urnn_train_deconv_weight_rows(const float *w, float *rows, int Cin, int Cout, hipStream_t st);
hipError_t urnn_train_deconv_rows_weight(const float *drows, const float *dsum, float *dw, float *db, int Cin, int Cout, int accumulate,
                                         hipStream_t st);
hipError_t urnn_train_head_save(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *stats, int B,
                                int P, float *save, hipStream_t st);
hipError_t urnn_train_head_pred_bwd(const float *dout, const float *cls, const float *reg, const float *reg_w, float thr, float slope,
                                    int B, int P, float *draw, float *ds, hipStream_t st);
int urnn_train_head_ln_chunks(int P);   // blocks per channel plane of the head's LayerNorm backward (the size of its partial buffer)
hipError_t urnn_train_head_ln_bwd(float *ds, const float *u, const float *g, const float *bt, const float *stats, int B, int P, float *dg,
                                  float *dbt, int accumulate, float *partial, float *coef, hipStream_t st);
int urnn_train_loss_nblk(long n);
hipError_t urnn_train_loss(const float *reg, const float *tgt, float thr, long n, float *partial, float *scales, float *comps, float *dreg,
                           hipStream_t st);
hipError_t urnn_train_clip_coef(const float *g, long n, float max_norm, float *partial, float *out, hipStream_t st);
hipError_t urnn_train_adam(float *p, const float *g, float *m, float *v, long n, float lr, float b1, float b2, float eps, int step,
                           const int *step_dev, const float *coef, hipStream_t st);
