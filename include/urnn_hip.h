/*
 * urnn_hip.h -- C ABI of liburnn_hip.so, the MI355X (gfx950) kernels under the U-RNN rollout path.
 *
 * The reference has no FFI/operator interface: its hot path sits behind plain nn.Module calls
 * (SURVEY 8b).  Each entry point below replaces the torch.nn / ATen ops of one reference module
 * forward, cited as file:line relative to /root/reference/code/.  INTEGRATION.md shows the ctypes
 * binding a reference maintainer would add at each of those sites.
 *
 * Conventions (all entry points):
 *   - tensors are float32, NCHW, contiguous, device pointers (hipMalloc / torch storage);
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream on ROCm); calls only
 *     ENQUEUE on it, never synchronise, allocate or retain pointers => safe under hipGraph capture;
 *   - scratch memory is caller-owned: size it with the matching *_workspace_bytes();
 *   - return 0 on success, a negative URNN_E* code on argument errors, or a positive hipError_t;
 *     urnn_last_error() gives the message of the calling thread's last failure;
 *   - weights cross the ABI in a PACKED layout produced once by urnn_pack_*() from the
 *     reference's nn.Conv2d / nn.ConvTranspose2d parameter layout.
 */
#ifndef URNN_HIP_H
#define URNN_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: rounds 1-3.  2: the first 256 bytes of every cell / head workspace are status / barrier words (zero them once:
 * "Operand range of the default matrix mode" and URNN_PHASE_COOP below), the cooperative and frame-loop entry points, the cell tail.
 * 3: the status area is URNN_STATUS_AREA_BYTES = 16 KB (zero once): the cooperative launches' grid barrier moved from two words of the
 * first 256 bytes to 33 words, 256 bytes apart, from byte 1024 on (sharded arrivals, no fences: 2.7 instead of 7.7 us per barrier). */
#define URNN_ABI_VERSION 3
#define URNN_STATUS_AREA_BYTES 16384

#define URNN_OK 0
#define URNN_EINVAL (-1)   /* bad dimension / unsupported shape                     */
#define URNN_ENULL (-2)    /* required pointer is NULL                              */
#define URNN_EWORKSPACE (-3) /* workspace too small                                 */
#define URNN_EALIGN (-4)   /* pointer not 16-byte aligned                           */

int urnn_abi_version(void);
const char *urnn_last_error(void);

/* One device per process.  The library caches, once per process, the compute-unit count of the device that is current at the first
 * cooperative plan (the block limits of URNN_PHASE_COOP / urnn_head_coop_f32) and the raised dynamic-LDS limits of its large kernels:
 * drive ONE device per process (torchrun's model, main.py:76-81) and make it current before the first call.  All phases of one cell
 * (urnn_gru_cell_phases_f32 called several times for the same cell) must run under the same matrix mode: the candidate's tile size --
 * and with it the layout of its GroupNorm partials the later phases fold -- is planned from the shapes AND the mode. */

/* Arithmetic of the GEMM kernels, process-wide (one process drives one GPU; the launch functions read it when they enqueue, so a
 * captured hipGraph keeps the mode it was captured with).
 *   URNN_MATRIX_FP32 (default): the reference's fp32 semantics on the 16-bit matrix pipe.  Forward GEMMs: both fp32 operands
 *     as two scaled f16 pieces (22 significant bits; activations x 2^5, weights x 2^10, result x 2^-15 -- all exact), three
 *     v_mfma_f32_32x32x16_f16 per 16 k, fp32 accumulation (rms error 1.8e-7 of the dot product's natural scale, below the fp32
 *     fma chain's 2.7e-7; finite for |activation| < 2047 and |weight| < 64, beyond that the outputs turn NaN).  Gradient GEMMs
 *     (no lower bound on the magnitudes): three exact bf16 pieces, six v_mfma_f32_32x32x16_bf16 per 16 k.  v_mfma_f32_32x32x2_f32
 *     where the channel counts do not form whole 16-k groups.
 *   URNN_MATRIX_BF16: bf16 compute for training (BASELINE configs[3]; the reference only declares --amp, config.py:179) --
 *     GEMM operands rounded to bf16 (weights keep 16 mantissa bits in the forward / input-gradient GEMMs), fp32 accumulation;
 *     norms, statistics, states, loss and the optimizer stay fp32.  Applies to every forward GEMM, the input-gradient GEMMs AND
 *     the weight-gradient GEMMs (dY and X rounded to bf16, fp32 accumulate); gradients are stored in fp32. */
/*   URNN_MATRIX_FP32_MFMA: every GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation: the k-loop the other modes
 *     keep for channel counts that do not form 16-k groups), the cells on their three-pass kernels.  About 0.8x the default
 *     mode's frames/s.  Background: per step the 16-bit k-loops are the MORE accurate ones (cell output vs
 *     float64: 5e-8 against plain fp32 torch's 1.2e-7 rms at 500x500), but over hundreds of recurrent steps ONE product of the
 *     cell in that form -- W2[:, h] . (r * h) of the full-resolution cells -- moves the trajectory 1.3-1.8x further from the float64
 *     trajectory than plain fp32 torch's.  The default mode therefore computes that product on the fp32 instruction wherever the
 *     fused candidate kernel runs (URNN_PHASE_FUSED_R); this mode and the next exist for the cells that cannot take that kernel
 *     and for callers that compare long rollouts digit by digit. */
#define URNN_MATRIX_FP32 0
#define URNN_MATRIX_BF16 1
#define URNN_MATRIX_FP32_MFMA 2
/*   URNN_MATRIX_FP32_CAND: the default mode, except that the candidate GEMM of a cell on a plane of >= 100 000 pixels per sample (the
 *     two full-resolution cells of the published network) runs on v_mfma_f32_32x32x2_f32 as a whole and the cell takes its
 *     three-pass form (URNN_PHASE_FUSED_R is ignored): the long-rollout behaviour of URNN_MATRIX_FP32_MFMA for the one launch that
 *     needs it, whatever the plane's shape, at a few per cent of the frames/s. */
#define URNN_MATRIX_FP32_CAND 3
int urnn_set_matrix_mode(int mode);
int urnn_get_matrix_mode(void);

/* ---- weight packing (one-off, at checkpoint-load time) ------------------------------------------ */

/* Packed 1x1-conv weights: transposed, K padded to 8, N padded to 32, bias appended.
 * Source: nn.Conv2d.weight (Cout, Cin, 1, 1) + bias (Cout) -- utils.py:109-115 (make_layers 'conv'). */
size_t urnn_packed_conv_floats(int Cin, int Cout);
int urnn_pack_conv_f32(const float *weight, const float *bias, float *packed, int Cin, int Cout, void *stream);

/* Packed ConvGRU / Skip-ConvGRU weights: the two GEMMs of the cell, each as the LDS image its kernel keeps resident
 * (rows = input channels x | e | h in k-pairs, x padded to an even count): the gate GEMM conv1 (2F x K) in F/32 groups of
 * [z_i | r_i] 32-column blocks + b1, then the candidate GEMM conv2 (F x K) in groups of at most three 32-column blocks + b2.
 * K = I + F (encoder, skip=0) or I + 2F (decoder).  F must be 32, 64, 96 or 128.
 * Source: CGRU_cell.conv1[0] / conv2[0] weight+bias -- ConvRNN.py:94-104. */
size_t urnn_packed_gru_floats(int I, int F, int skip);
int urnn_pack_gru_f32(const float *W1, const float *b1, const float *W2, const float *b2, float *packed, int I, int F,
                      int skip, void *stream);

/* Packed ConvTranspose2d(k=2,s=2) weights.  Source: nn.ConvTranspose2d.weight (Cin, Cout, 2, 2) + bias
 * -- utils.py:95-101 (make_layers 'deconv'). */
size_t urnn_packed_deconv_floats(int Cin, int Cout);
int urnn_pack_deconv_f32(const float *weight, const float *bias, float *packed, int Cin, int Cout, void *stream);

/* Operand range of the default matrix mode.  The f16 pieces of URNN_MATRIX_FP32 are finite for |weight| < 64 and
 * |activation| < 2047; the published network (O(1) weights, normalised activations) sits far inside.  Two guards for a checkpoint
 * that does not:
 *   - weights: max |w[i]| of a device tensor -> *max_abs_out (device float; enqueue-only like everything else).  A layer whose
 *     weights reach 64 must be launched under URNN_MATRIX_FP32_MFMA (exact fp32 matrix instruction, no range limit); the Python
 *     host does exactly that per layer (u-rnn_amd/networks/_packing.py, ops.exact_matrix_if) -- reference checkpoint path:
 *     test.py:380-408.
 *   - activations: the FIRST URNN_STATUS_AREA_BYTES (16 KB) of every cell / head workspace are status / barrier words.  Kernels only ever atomically OR into
 *     word 0: URNN_STATUS_GATES / _CAND when the GroupNorm sums of a cell's gates / candidate are not finite, URNN_STATUS_HEAD
 *     for a LayerNorm of the head -- which is where an overflowed operand (inf out of the matrix pipe) surfaces, one norm later
 *     at most.  The owner zeroes the workspace once and reads the word wherever it synchronises anyway (RolloutEngine: once per
 *     event, raising FloatingPointError and naming the first layer with non-finite output). */
#define URNN_STATUS_GATES 1
#define URNN_STATUS_CAND 2
#define URNN_STATUS_HEAD 4
int urnn_max_abs_f32(const float *values, long n, float *max_abs_out, void *stream);

/* ---- per-module forwards -------------------------------------------------------------------------- */

/* Encoder/decoder stage conv: out = [AvgPool2d(2,2)](LeakyReLU_slope(W.in + b)).
 * Replaces Encoder.stage{1,2,3}(inputs) -- encoder.py:140-151 with specs net_params.py:80-88 -- and
 * Decoder.stage1 (plain conv) -- decoder.py:150-164 / net_params.py:117-120.
 * in (B,Cin,H,W) -> out (B,Cout,H,W) or (B,Cout,H/2,W/2) when pool != 0. */
int urnn_stage_conv_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, int pool,
                        float slope, void *stream);

/* The decoder's last stage (Decoder.stage1: Conv1x1 F -> 16 + LeakyReLU, decoder.py:150-164) with the FIRST pass of the head done in its
 * epilogue: out as urnn_stage_conv_f32(pool = 0), plus the partial statistics of the head's first LayerNorm (u0 = stems.conv . out,
 * flood_head.py:131-140) in head_partial0 (urnn_head_tail_partial_floats floats) -- urnn_head_after_tail_f32 then runs the head without
 * reading the feature map for them.  head_conv_w: the head's conv_w (its first 16 x 16 block is used).  Applies (urnn_stage_conv_stem_applies)
 * where the conv takes its 128-pixel-tile form: Cout = 16, P % 4 == 0, >= 131 072 pixels per launch, f16-piece matrix modes; URNN_EINVAL elsewhere. */
int urnn_stage_conv_stem_applies(int B, int Cin, int Cout, int H, int W);
int urnn_stage_conv_stem_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, float slope,
                             const float *head_conv_w, float *head_partial0, void *stream);

/* ConvGRU (e == NULL) / Skip-ConvGRU cell, one timestep.
 * Replaces CGRU_cell.forward(inputs, hidden_state, seq_len=1) -- ConvRNN.py:111-194 -- including the
 * torch.cat of decoder.py:130-135.  x (B,I,H,W) may be NULL: x == 0 (decoder stage 3, ConvRNN.py:143-146).
 * h_out may alias h (in-place state update).  gn*_w / gn*_b are the GroupNorm affines of conv1[1]/conv2[1]. */
size_t urnn_gru_cell_workspace_bytes(int B, int F, int H, int W);
int urnn_gru_cell_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                      const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                      size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, void *stream);

/* The same cell with a subset of its kernels enqueued (profiling / roofline measurement: bench.py times the
 * gate GEMM alone with this).  phase_mask is an OR of URNN_PHASE_*; URNN_PHASE_ALL == urnn_gru_cell_f32. */
#define URNN_PHASE_GATES 1  /* gate GEMM: raw z|r gates, GroupNorm partial sums                          */
#define URNN_PHASE_GN1 2    /* GroupNorm finalise of the gates (part of the CAND kernel when both set) */
#define URNN_PHASE_CAND 4   /* candidate GEMM W2.[x;e;sigmoid(GN(r))*h], GroupNorm partial sums       */
#define URNN_PHASE_GN2 8    /* GroupNorm finalise of the candidate                                   */
#define URNN_PHASE_BLEND 16 /* h' = (1 - z) * h + z * tanh(GN(c))                                    */
#define URNN_PHASE_ALL 31
/* Optional modifier (not part of URNN_PHASE_ALL), for callers that do not read the workspace afterwards -- inference rollouts:
 * the reset gate is RECOMPUTED inside the candidate kernel (W1[r rows].[x;e;h] next to W2.[x;e], same input stream) instead of
 * being written as F raw planes by the gate GEMM and read back; the gate GEMM keeps r's GroupNorm statistics and stores z only.
 * Same arithmetic for r (bit-identical accumulators); two plane passes of F channels less per cell (ConvRNN.py:165-180).  Applies
 * where the cell has that form (F = 64, P % 4 == 0, >= 65 536 pixels per launch, fp32 matrix mode, slabs within the LDS) and is
 * ignored elsewhere -- except that a cell on a plane of >= 100 000 pixels that cannot take this form (P % 4 != 0, F != 64) then runs its
 * candidate GEMM on the fp32 matrix instruction, so that a long rollout behaves the same whatever the grid's shape; pass the flag to
 * every phase-split call of a cell or to none.  The workspace's raw reset-gate planes are then
 * undefined: the backward pass (urnn_gru_cell_backward_f32) needs a forward WITHOUT this flag. */
#define URNN_PHASE_FUSED_R 32
/* Optional modifier (not part of URNN_PHASE_ALL), for the same callers: ONE cooperative launch for the whole cell of a small plane
 * (<= 256 blocks of 64 pixels, F <= 96, f16-piece matrix modes): gate GEMM -> grid barrier -> candidate GEMM -> grid barrier ->
 * blend, with the raw gates and the candidate kept in registers; only the partial GroupNorm statistics leave the CU between the
 * phases (ConvRNN.py:111-194).  Same arithmetic and summation orders as the three kernels: h_out is bit-identical.  Needs every
 * phase in the mask; ignored where the shape does not qualify.  The workspace's raw gate / candidate planes are then undefined.
 * Bytes 1024 .. 9727 of the workspace's status area are the launch's barrier state (zero once, never touch).  A grid barrier that does
 * not complete within ~1 s gives up and sets URNN_STATUS_BARRIER in the status word instead of hanging the GPU.
 * urnn_gru_cell_coop_blocks: how many blocks that launch would take for a cell of this shape (0: the flag would be ignored).  Every
 * block must be resident at once (one per CU): a caller that keeps SEVERAL kernel chains in flight passes the flag only for cells
 * of at most 128 blocks (any two of those fit side by side; u-rnn_amd/rollout.py), a single chain for any planned cell. */
#define URNN_PHASE_COOP 64
#define URNN_STATUS_BARRIER 8
int urnn_gru_cell_coop_blocks(int B, int I, int F, int H, int W, int skip, int has_x);
/* The END of a cell fused with the layer that consumes its new state (encoder.py:170-185, decoder.py:150-164: every cell's output
 * goes straight into a stage's 1x1 conv): the phases in phase_mask run as in urnn_gru_cell_phases_f32, except that GN2 | BLEND and
 * the conv  conv_out = [AvgPool2](LeakyReLU_slope(Wc . h_out + bc))  (conv_packed: urnn_pack_conv_f32 of a conv with Cin = F) are ONE
 * launch -- h_out is written as always, but the stage conv no longer reads it back and is no launch of its own.  h_out is
 * bit-identical to the unfused cell's, conv_out to urnn_stage_conv_f32(h_out).  With head_conv_w (the head's stem conv, 16 x 16; flat
 * Cout = 16 only) the launch also takes the statistics of the head's first LayerNorm into head_partial0
 * (urnn_head_tail_partial_floats floats) and urnn_head_after_tail_f32 runs the head without its first pass over feat
 * (flood_head.py:131-140).  Shapes: urnn_gru_cell_tail_applies (F = 64 / 96, Cout <= F, even H and W when pooling, f16-piece matrix
 * modes); other shapes return URNN_EINVAL -- call the unfused entries. */
int urnn_gru_cell_tail_applies(int B, int F, int H, int W, int Cout, int pool);
size_t urnn_head_tail_partial_floats(int B, int H, int W);
int urnn_gru_cell_tail_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                           const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                           size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask,
                           const float *conv_packed, int Cout, int pool, float slope, float *conv_out,
                           const float *head_conv_w, float *head_partial0, void *stream);
int urnn_head_after_tail_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                             const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                             float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C, int H,
                             int W, float cls_thred, float eps, float slope, const float *head_partial0, void *stream);
/* The head of a small plane as ONE cooperative launch: its four passes with three grid barriers between them, every thread keeping its
 * pixels' branch activations in registers (flood_head.py:131-177).  Bit-identical to urnn_head_f32.  urnn_head_coop_blocks_f32: the
 * blocks of that launch, or 0 when they could not all be resident at once on THIS device (one block per compute unit: the count is
 * queried, 256 on an MI355X; urnn_head_coop_f32 then returns URNN_EINVAL) -- and at most half the compute units for a caller with
 * several kernel chains in flight (the rule of URNN_PHASE_COOP, whose block limit is bounded by the queried count the same way).
 * Barrier state: words 16 and 32 of the workspace's status area. */
int urnn_head_coop_blocks_f32(int B, int H, int W);
int urnn_head_coop_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                       const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                       float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C, int H,
                       int W, float cls_thred, float eps, float slope, void *stream);
/* 1 when URNN_PHASE_FUSED_R takes effect for a cell of this shape under the current matrix mode (x present; skip: an e input of F
 * channels), else 0 -- for byte accounting (bench.py) and tests; the cell entry decides by the same rule. */
int urnn_gru_cell_fused_reset_gate_applies(int B, int I, int F, int H, int W, int skip);
int urnn_gru_cell_phases_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                             const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                             size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask,
                             void *stream);

/* Decoder up-sampling: out = LeakyReLU_slope(ConvTranspose2d(k=2,s=2,p=0)(in)).
 * Replaces Decoder.stage{3,2}(inputs) -- decoder.py:150-164 with specs net_params.py:106-116.
 * in (B,Cin,H,W) -> out (B,Cout,2H,2W). */
int urnn_deconv2x2_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, float slope,
                       void *stream);

/* Dual head + wet/dry mask.  Replaces YOLOXHead.forward + correction_depth -- flood_head.py:131-202 --
 * with BaseConv = Conv1x1(no bias) -> LayerNorm([C,H,W]) -> SiLU (network_blocks.py:74-101) and
 * finalConv (network_blocks.py:129-171).
 *   feat (B,C,H,W); conv_w 5 x (C x C) row-major in the order stems, cls_convs.0, cls_convs.1,
 *   reg_convs.0, reg_convs.1; ln_w / ln_b 5 x (C,H,W) same order; cls_w/reg_w (C), cls_b/reg_b (1).
 *   out_masked = reg * [cls >= cls_thred], out_cls = cls, out_raw (nullable) = reg before the mask;
 *   each written at  base + frame * B*H*W  where frame = *frame_index (device int, NULL => 0): a rollout
 *   graph replays the same node while the device-side frame counter advances. */
size_t urnn_head_workspace_bytes(int B, int C, int H, int W);
int urnn_head_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                  const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                  float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C, int H,
                  int W, float cls_thred, float eps, float slope, void *stream);

/* Per-frame input assembly.  Replaces preprocess_inputs / get_past_rainfall / MinMaxScaler --
 * Dynamic2DFlood.py:265-376.  out (B, 2*nums+3, H, W) = [rain(t-n+1..t)/rain_max, cumsum(..)/cumsum_max,
 * (DEM-min)/(max-min), (imp-0.05)/0.9, manhole], history left-zero-padded.
 *   rain / cumsum: (B,T) if spatial == 0 else (B,T,H,W); dem / imperv / manhole: (B,H,W);
 *   t is read from *t_dev when t_dev != NULL (device int), else from the host argument t. */
int urnn_preprocess_f32(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                        const float *manhole, float dem_min, float dem_max, float *out, int t, const int *t_dev, int B,
                        int T, int nums, int H, int W, int spatial, float rain_max, float cumsum_max, void *stream);

/* Scalar-rainfall fast path of encoder stage 1: preprocess_inputs + Encoder.stage1 fused for events whose
 * rainfall is one value per frame (Dynamic2DFlood.py:181-216 "scalar" case; encoder.py:140-151; net_params.py:80-81).
 * With scalar rain 2*nums of the 2*nums+3 input channels are spatial constants, so
 *   stage1(preprocess_inputs(t))[n][p] = LeakyReLU( S[n][p] + v_t[n] )
 *   S   = W[:, 2*nums:] . [(DEM-min)/(max-min), (imp-0.05)/0.9, manhole]      urnn_stage1_static_f32, once per event
 *   v_t = b + W[:, :2*nums] . [rain hist / rain_max, cumsum hist / cumsum_max]  computed inside urnn_stage1_scalar_rain_f32
 * weight is the nn.Conv2d weight (Cout, 2*nums+3) in its reference layout; rain / cumsum are (B,T); S and out (B,Cout,H,W).
 * The (B, 2*nums+3, H, W) input tensor never materialises (63 MB per frame at 500x500). */
int urnn_stage1_static_f32(const float *dem, const float *imperv, const float *manhole, float dem_min, float dem_max,
                           const float *weight, float *S, int B, int nums, int Cout, int H, int W, void *stream);
int urnn_stage1_scalar_rain_f32(const float *S, const float *rain, const float *cumsum, const float *weight, const float *bias,
                                float *out, int t, const int *t_dev, int B, int T, int nums, int Cout, int H, int W,
                                float rain_max, float cumsum_max, float slope, void *stream);

/* ---- training: backward of every layer, loss, optimizer step; first version --------------------- */

/* Backward of urnn_gru_cell_f32: gradients of every input and parameter of one ConvGRU / Skip-ConvGRU step
 * (autograd of CGRU_cell.forward -- ConvRNN.py:111-194) given dL/dh' = dh_out (+ dh_out2, dh_out3, dh_out4 where non-NULL: the
 * state's consumers -- the layer above, a skip connection's reader, the next timestep -- hand in their terms separately and the
 * sum is formed on the fly, left to right).  dh2 (optional, (B,F,H,W)): when non-NULL, dL/dh leaves as TWO terms, dh + dh2 (dh2 =
 * W1[:, h]^T . dgates straight out of its GEMM), for a consumer that sums its terms itself -- the previous timestep's call of this
 * function; NULL: dh holds the whole gradient.
 * Call it after the forward of the SAME x / e / h with the forward's workspace untouched (fwd_workspace: it holds the raw
 * gates, the raw candidate, the folded GroupNorm tables and the group statistics).  W1 (2F,K) / W2 (F,K) are the conv
 * weights in their reference layout, K ordered x | e | h.  dx / de must be non-NULL exactly when x / e are; dx, de, dh
 * (B,*,H,W) are overwritten; the parameter gradients (shapes of the parameters) are overwritten, or added to when
 * accumulate != 0 (BPTT over the steps of an SWP window).  Deterministic: reductions use fixed-order partial sums. */
size_t urnn_gru_cell_backward_workspace_bytes(int B, int I, int F, int skip, int H, int W);
/* bwd_packed (urnn_gru_cell_backward_packed_floats floats, caller-owned, one per cell): the packed weights of the three
 * input-gradient GEMMs.  They depend on W1 / W2 only: pass repack != 0 on the first backward call after the weights changed
 * (once per SWP window) and 0 afterwards.  When dx and de are one contiguous block (de == dx + I*H*W, B == 1) they are written
 * in place by a single GEMM over the contraction [dgates; dcandidate]. */
size_t urnn_gru_cell_backward_packed_floats(int I, int F, int skip);
int urnn_gru_cell_backward_f32(const float *x, const float *e, const float *h, const float *W1, const float *W2, const float *gn1_w,
                               const float *gn2_w, const void *fwd_workspace, const float *dh_out, const float *dh_out2,
                               const float *dh_out3, const float *dh_out4, float *dx, float *de, float *dh, float *dh2,
                               float *dW1, float *db1, float *dgn1_w, float *dgn1_b, float *dW2, float *db2, float *dgn2_w,
                               float *dgn2_b, float *bwd_packed, int repack, void *workspace, size_t workspace_bytes, int B, int I,
                               int F, int H, int W, int accumulate, void *stream);

/* Weight gradient of a 1x1 convolution over the channel concatenation of up to three inputs (the GEMM every layer backward above
 * runs; autograd of nn.Conv2d's weight / bias -- utils.py:90-94, ConvRNN.py:94-104):  dW[n][k] (+)= sum_{b,p} dy[b][n][p] * x[b][k][p],
 * db[n] (+)= sum_{b,p} dy[b][n][p].  dy (B,N,P); seg0 / seg1 / seg2 (B,C_i,P) with C0 + C1 + C2 = K (unused segments: NULL, 0); dW
 * (N,K) row-major, db (N) or NULL.  Deterministic (fixed-order pixel chunks).  Exposed for measurement (bench.py --mode train
 * prices this kernel against the HBM roof) and for callers that differentiate a layer of their own. */
size_t urnn_weight_gradient_workspace_bytes(int B, int N, int K, int H, int W);
int urnn_weight_gradient_f32(const float *dy, const float *seg0, int C0, const float *seg1, int C1, const float *seg2, int C2, float *dW,
                             float *db, void *workspace, size_t workspace_bytes, int B, int N, int H, int W, int accumulate, void *stream);

/* Backward of urnn_stage_conv_f32 (conv1x1 + LeakyReLU [+ AvgPool2d(2,2)]): weight (Cout,Cin), bias (Cout) in the reference
 * layout; dout has the forward output's shape; din (B,Cin,H,W) is overwritten, dweight / dbias overwritten or accumulated. */
size_t urnn_stage_conv_backward_workspace_bytes(int B, int Cin, int Cout, int H, int W);
/* bwd_packed (urnn_stage_conv_backward_packed_floats floats, 16-byte aligned, caller-owned): the layer's packed weights for the
 * backward GEMMs, filled when repack != 0 and reused by later calls until the parameters change (once per SWP window, like
 * urnn_gru_cell_backward_f32); NULL: packed into the workspace on every call. */
size_t urnn_stage_conv_backward_packed_floats(int Cin, int Cout);
int urnn_stage_conv_backward_f32(const float *in, const float *weight, const float *bias, const float *dout, float *din, float *dweight,
                                 float *dbias, float *bwd_packed, int repack, void *workspace, size_t workspace_bytes, int B, int Cin,
                                 int Cout, int H, int W, int pool, float slope, int accumulate, void *stream);

/* Backward of urnn_deconv2x2_f32: weight (Cin,Cout,2,2); out = the forward output (B,Cout,2H,2W) (LeakyReLU keeps the sign
 * of its input), dout the same shape; din (B,Cin,H,W) overwritten, dweight / dbias overwritten or accumulated. */
size_t urnn_deconv2x2_backward_workspace_bytes(int B, int Cin, int Cout, int H, int W);
size_t urnn_deconv2x2_backward_packed_floats(int Cin, int Cout);      /* bwd_packed / repack: as for the stage conv */
int urnn_deconv2x2_backward_f32(const float *in, const float *weight, const float *out, const float *dout, float *din, float *dweight,
                                float *dbias, float *bwd_packed, int repack, void *workspace, size_t workspace_bytes, int B, int Cin,
                                int Cout, int H, int W, float slope, int accumulate, void *stream);

/* Backward of urnn_head_f32 w.r.t. its first output (the masked depth; the loss never sees the probability map).  Call it
 * after the forward on the same feat with the forward's workspace untouched (fwd_workspace: the five LayerNorm statistics) and
 * the forward's out_raw / out_cls.  Only the regression branch carries gradient: the wet/dry mask is a comparison
 * (flood_head.py:179-202), so the classification blocks' gradients are zero.  dconv_w 5 x (C x C), dln_w / dln_b 5 x (C,H,W),
 * dreg_w (C), dreg_b (1) in the forward's parameter order; overwritten, or added to when accumulate != 0. */
size_t urnn_head_backward_workspace_bytes(int B, int H, int W);
int urnn_head_backward_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *reg_w,
                           const void *fwd_workspace, const float *out_raw, const float *out_cls, const float *dout, float *dfeat,
                           float *dconv_w, float *dln_w, float *dln_b, float *dreg_w, float *dreg_b, void *workspace,
                           size_t workspace_bytes, int B, int C, int H, int W, float cls_thred, float slope, int accumulate,
                           void *stream);

/* Training loss FocalBCE_and_WMSE (losses.py:44-249) on a window's concatenated outputs reg / targets (n values each), as the SWP
 * loop forms it (main.py:489-539: cls = [reg >= cls_thred], a comparison).  components (device, 5 floats): loss, loss_reg,
 * wet-cell MSE, dry-cell MSE, loss_cls.  dreg (n, may be NULL): d loss / d reg.  Deterministic. */
size_t urnn_loss_workspace_bytes(long n);
int urnn_loss_f32(const float *reg, const float *target, float cls_thred, float *components, float *dreg, void *workspace,
                  size_t workspace_bytes, long n, void *stream);

/* One optimizer step on flat buffers of n floats: gradients are scaled by min(1, max_grad_norm / (||g||_2 + 1e-6))
 * (torch.nn.utils.clip_grad_norm_, main.py:760-761; max_grad_norm <= 0: no clipping), then Adam with torch.optim.Adam's
 * defaults' arithmetic (main.py:306).  step counts from 1; step_dev (device int, may be NULL) overrides it for hipGraph replay.
 * clip_out (device, 2 floats): the coefficient and the norm. */
size_t urnn_adam_workspace_bytes(long n);
int urnn_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long n, float lr, float beta1, float beta2,
                       float eps, int step, const int *step_dev, float max_grad_norm, float *clip_out, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ---- Spatial strips (one event's plane split over ranks, single-event latency) ------------------------------
 * Every conv of the network is 1x1 and the pool / transposed-conv blocks are 2x2, so a horizontal strip whose height is a multiple
 * of 4 needs no halo: the only cross-strip quantities are the GroupNorm (ConvRNN.py:85,99) and LayerNorm (network_blocks.py:93)
 * statistics.  A strip run is the phase-split cell / head with an exchange between the phase that accumulates a norm's partial
 * sums and the one that finalizes it:
 *     urnn_gru_cell_strip_f32(GATES)  -> stats(which 1, direction 0) -> all-reduce(sum) -> stats(which 1, direction 1)
 *     urnn_gru_cell_strip_f32(CAND)   -> stats(which 2, 0)           -> all-reduce      -> stats(which 2, 1)
 *     urnn_gru_cell_strip_f32(GN2 | BLEND)
 * and likewise K1 | F1 | K2 | F2 | K3 | F3 | K4 for the head (levels 0..2).  H, W are the STRIP's; global_pixels the whole plane's
 * pixel count at this stage's resolution.  sums: device doubles, (sum, sum of squares) per (sample, norm group) [B][groups][2]
 * (cell) or per (norm of the level, sample) [norms][B][2] (head); the caller all-reduces them (RCCL) between direction 0 and 1.
 * With one rank the result equals the unsplit call up to the double -> float pair round trip of the sums (~1e-14 relative). */
int urnn_gru_cell_strip_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w, const float *gn1_b,
                            const float *gn2_w, const float *gn2_b, float *h_out, void *workspace, size_t workspace_bytes, int B, int I,
                            int F, int H, int W, float eps, int phase_mask, long global_pixels, void *stream);
int urnn_gru_cell_strip_stats_f32(void *workspace, size_t workspace_bytes, int B, int F, int H, int W, int which, int direction,
                                  double *sums, void *stream);
#define URNN_HEAD_K1 1   /* stems conv, LayerNorm partial sums (level 0)                 */
#define URNN_HEAD_F1 2   /* finalize level 0                                             */
#define URNN_HEAD_K2 4   /* stems norm + SiLU, cls_convs.0 / reg_convs.0 convs (level 1) */
#define URNN_HEAD_F2 8
#define URNN_HEAD_K3 16  /* cls_convs.1 / reg_convs.1 (level 2)                          */
#define URNN_HEAD_F3 32
#define URNN_HEAD_K4 64  /* prediction layers, mask                                      */
#define URNN_HEAD_ALL 127
int urnn_head_strip_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                        const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls, float *out_raw,
                        const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C, int H, int W, float cls_thred,
                        float eps, float slope, int phase_mask, long global_pixels, void *stream);
int urnn_head_strip_stats_f32(void *workspace, size_t workspace_bytes, int B, int C, int H, int W, int level, int direction, double *sums,
                              void *stream);

/* Device-side frame counter helper for graph-captured rollouts: *counter += delta. */
int urnn_advance_counter(int *counter, int delta, void *stream);

/* Frame-loop forms (same call sites as urnn_head_f32 / urnn_preprocess_f32 / urnn_stage1_scalar_rain_f32 -- test.py:326-377, one loop
 * iteration -- for a CAPTURED loop, where the frame index must live on the device): the launch reads its frame from *frame_index /
 * *t_dev and stores that value + 1 to *frame_next / *t_next, the word the same entry's NEXT call will be given as its index.  Two
 * words used alternately (even frames read word 0 and write word 1, odd frames the reverse) thus form a frame counter that costs no
 * kernel of its own (urnn_advance_counter: ~2 us + a kernel boundary per frame and counter: 3 % of a 64x64 frame), and each kernel
 * family keeps its own pair -- the families run on different streams at different frames.  A launch must not write the word it
 * reads (its other blocks may not have read it yet): frame_next == frame_index / t_next == t_dev is URNN_EINVAL.  NULL = no store.
 * urnn_head_rollout_f32: coop != 0 = urnn_head_coop_f32's launch form, head_partial0 != NULL = urnn_head_after_tail_f32's. */
int urnn_head_rollout_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                          const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                          float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C, int H, int W,
                          float cls_thred, float eps, float slope, int coop, const float *head_partial0, int *frame_next,
                          void *stream);
int urnn_preprocess_rollout_f32(const float *rain, const float *cumsum, const float *dem, const float *imperv, const float *manhole,
                                float dem_min, float dem_max, float *out, const int *t_dev, int *t_next, int B, int T, int nums,
                                int H, int W, int spatial, float rain_max, float cumsum_max, void *stream);
int urnn_stage1_scalar_rain_rollout_f32(const float *S, const float *rain, const float *cumsum, const float *weight,
                                        const float *bias, float *out, const int *t_dev, int *t_next, int B, int T, int nums,
                                        int Cout, int H, int W, float rain_max, float cumsum_max, float slope, void *stream);

/* One whole inference timestep behind one call -- ED.forward, model.py:65-121 inside test.py:326-377's loop -- for a host that is not
 * Python: the launches the per-module entries above make for a frame, in their order, on ONE stream (enqueue-only: capture the call in a
 * hipGraph and replay it per frame -- what u-rnn_amd/rollout.py's one-chain engine does with the same launches; its three-chain schedule,
 * DESIGN.md 4.3, adds ~17 % on top and stays host-side).
 *   urnn_net_f32: the network as the slabs of urnn_pack_conv_f32 / urnn_pack_deconv_f32 / urnn_pack_gru_f32 and the norms' affines, plus its
 *     channel counts (net_params.py:80-116: in 2n+3 -> 16 | 64 | 64(pool) | 96 | 96(pool) | 96 ; decoder 96 (x = 0, dec_zero_input_channels
 *     = 96) -> 96 | 96 -> 96 | 64 -> 16; head width 16).  Index 0 of every decoder array is the DEEPEST stage (Decoder.stage3 / rnn3).
 *   x_t (B, in_channels, H, W): the frame's input (urnn_preprocess_f32); states: e1 e2 e3 d1 d2 d3 as model.py returns them (d1 the
 *     deepest), (B, F, H / s, W / s), updated IN PLACE; outputs and frame_index as urnn_head_f32.  H and W multiples of four.
 *   The cells run with URNN_PHASE_FUSED_R | URNN_PHASE_COOP and the head as urnn_head_coop_f32 wherever the shapes qualify (an inference
 *   step; the training forward needs the raw planes and uses the per-module entries).  Operand-range rules as for those entries.
 *   workspace: urnn_step_workspace_bytes; urnn_step_workspace_init zeroes its two status areas (once per workspace) and returns their
 *   addresses (word 0 of each: URNN_STATUS_* bits). */
typedef struct urnn_net_f32 {
    int in_channels;                       /* 2 * nums + 3                                                            */
    int enc_stage_out[3];                  /* Encoder.stage1..3 output channels (16, 64, 96); stages 2 and 3 pool 2x2 */
    int enc_features[3];                   /* Encoder.rnn1..3 hidden channels (64, 96, 96)                            */
    int dec_zero_input_channels;           /* Decoder.rnn3's declared (all-zero) input channels (96)                  */
    int dec_features[3];                   /* Decoder.rnn3, rnn2, rnn1 hidden channels (96, 96, 64)                   */
    int dec_stage_out[2];                  /* Decoder.stage3, stage2 (transposed convs) output channels (96, 96)      */
    int feat_channels;                     /* Decoder.stage1 output = head width (16)                                 */
    const float *enc_stage[3];             /* urnn_pack_conv_f32 slabs                                                */
    const float *enc_cell[3];              /* urnn_pack_gru_f32 slabs (skip = 0)                                      */
    const float *enc_gn1_w[3], *enc_gn1_b[3], *enc_gn2_w[3], *enc_gn2_b[3];
    const float *dec_cell[3];              /* urnn_pack_gru_f32 slabs (skip = 1), deepest first                       */
    const float *dec_gn1_w[3], *dec_gn1_b[3], *dec_gn2_w[3], *dec_gn2_b[3];
    const float *dec_stage[3];             /* [0], [1]: urnn_pack_deconv_f32 slabs of stage3, stage2; [2]: urnn_pack_conv_f32 slab of stage1 */
    const float *head_conv_w, *head_ln_w, *head_ln_b, *cls_w, *cls_b, *reg_w, *reg_b;   /* as urnn_head_f32          */
} urnn_net_f32;
size_t urnn_step_workspace_bytes(const urnn_net_f32 *net, int B, int H, int W);
int urnn_step_workspace_init(const urnn_net_f32 *net, void *workspace, size_t workspace_bytes, int B, int H, int W, int **cell_status,
                             int **head_status, void *stream);
int urnn_step_f32(const urnn_net_f32 *net, const float *x_t, float *const states[6], float *out_masked, float *out_cls, float *out_raw,
                  const int *frame_index, void *workspace, size_t workspace_bytes, int B, int H, int W, float cls_thred, float eps,
                  float slope, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* URNN_HIP_H */
