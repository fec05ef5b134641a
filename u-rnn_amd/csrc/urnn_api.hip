// urnn_api.hip -- the C ABI of liburnn_hip.so (declared in include/urnn_hip.h): argument checking, workspace carving and
// kernel sequencing.  Enqueue-only: nothing here synchronises, allocates or keeps state besides the thread-local error text.
#include "urnn_common.h"
#include "urnn_kernels.h"
#include "../../include/urnn_hip.h"

#include <stdarg.h>
#include <stdio.h>
#include <limits.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

#define CHECK_HIP(expr, what)                      \
    do {                                           \
        hipError_t e_ = (expr);                    \
        if (e_ != hipSuccess) return hip_fail(e_, what); \
    } while (0)

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// The activation DMA addresses one input of one sample through a buffer descriptor with 32-bit offsets: channels * P * 4
// bytes must stay below 4 GiB (P < 8.3 M pixels at 128 channels).  Larger planes need the spatial tiling of SURVEY 8(e).
static inline bool plane_fits(long P, int channels) { return P > 0 && (double)P * 4.0 * (double)(channels + 2) < 4294967296.0; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t slab_floats(size_t KT, size_t NB) { return (KT * NB * 64 + 255) / 256 * 256; }

extern "C" int urnn_abi_version(void) { return URNN_ABI_VERSION; }
extern "C" const char *urnn_last_error(void) { return g_err; }

// Wave-tile shape.  Big planes use 128-pixel tiles (PB = 4, 16-byte DMA); small planes shrink the tile so that the launch
// still covers the chip (1024 SIMDs; the deep stages are 15 625 pixels at 500x500).  Planes that are not 16-byte / 8-byte
// aligned fall back to dword DMA (MAP_PAIR / MAP_STRIDED).
static int tune_env(const char *name)
{
    return (int)urnn_tune(name, 0);
}

static void pick_tile(long pixels_total, int waves_per_tile, long P, int *pb_out, int *map_out, const char *tune = nullptr,
                      long min_items = 2048)
{
    int pb = 4;
    const int forced = tune ? tune_env(tune) : 0;   // development knob: URNN_TUNE_PB_* = 1 | 2 | 4
    if (forced == 1 || forced == 2 || forced == 4) pb = forced;
    else
        while (pb > 1 && (pixels_total / (32 * pb)) * waves_per_tile < min_items) pb >>= 1;
    if (pb == 4 && P % 4 != 0) pb = 2;
    *pb_out = pb;
    *map_out = pb == 4 ? MAP_VEC : (pb == 2 && P % 4 == 0 ? MAP_PAIR16 : (pb == 2 && P % 2 == 0 ? MAP_PAIR : MAP_STRIDED));
}

extern "C" int urnn_max_abs_f32(const float *values, long n, float *max_abs_out, void *stream)
{
    if (!values || !max_abs_out) return fail(URNN_ENULL, "urnn_max_abs_f32: NULL argument");
    if (n < 1) return fail(URNN_EINVAL, "urnn_max_abs_f32: n=%ld", n);
    CHECK_HIP(urnn_launch_max_abs(values, n, max_abs_out, (hipStream_t)stream), "max_abs");
    return URNN_OK;
}

// ---- packing ---------------------------------------------------------------------------------------------------------
extern "C" size_t urnn_packed_conv_floats(int Cin, int Cout)
{
    const size_t NB = urnn_conv_nb(Cout), NG = urnn_conv_ng(Cout), KT = (Cin + 1) / 2;
    return NG * slab_floats(KT, NB) + NG * NB * 32 + NG * (size_t)urnn_split_slab_dwords((int)KT, (int)NB) +
           NG * (size_t)urnn_f16_slab_dwords((int)KT, (int)NB);
}

extern "C" int urnn_pack_conv_f32(const float *weight, const float *bias, float *packed, int Cin, int Cout, void *stream)
{
    if (!weight || !packed) return fail(URNN_ENULL, "urnn_pack_conv_f32: weight/packed is NULL");
    if (Cin < 1 || Cout < 1) return fail(URNN_EINVAL, "urnn_pack_conv_f32: Cin=%d Cout=%d", Cin, Cout);
    CHECK_HIP(urnn_launch_pack_conv(weight, bias, packed, Cin, Cout, (hipStream_t)stream), "pack_conv");
    return URNN_OK;
}

extern "C" size_t urnn_packed_gru_floats(int I, int F, int skip)
{
    if (I < 1 || F < 32 || F % 32 != 0 || F > 128) return 0;
    return urnn_packed_gru_total(I, F, skip);
}

extern "C" int urnn_pack_gru_f32(const float *W1, const float *b1, const float *W2, const float *b2, float *packed, int I, int F,
                                 int skip, void *stream)
{
    if (!W1 || !b1 || !W2 || !b2 || !packed) return fail(URNN_ENULL, "urnn_pack_gru_f32: NULL argument");
    if (I < 1 || F < 32 || F % 32 != 0 || F > 128)
        return fail(URNN_EINVAL, "urnn_pack_gru_f32: need I>=1 and F in {32,64,96,128} (got I=%d F=%d)", I, F);
    CHECK_HIP(urnn_launch_pack_gru(W1, b1, W2, b2, packed, I, F, skip ? 1 : 0, (hipStream_t)stream), "pack_gru");
    return URNN_OK;
}

extern "C" size_t urnn_packed_deconv_floats(int Cin, int Cout)
{
    const size_t NB = 2 * (size_t)((Cout + 31) / 32), KT = (Cin + 1) / 2;
    return 2 * slab_floats(KT, NB) + 2 * NB * 32 + 2 * (size_t)urnn_split_slab_dwords((int)KT, (int)NB) +
           2 * (size_t)urnn_f16_slab_dwords((int)KT, (int)NB);
}

extern "C" int urnn_pack_deconv_f32(const float *weight, const float *bias, float *packed, int Cin, int Cout, void *stream)
{
    if (!weight || !packed) return fail(URNN_ENULL, "urnn_pack_deconv_f32: weight/packed is NULL");
    if (Cin < 1 || Cout < 1 || Cout > 96) return fail(URNN_EINVAL, "urnn_pack_deconv_f32: need 1<=Cout<=96 (got Cin=%d Cout=%d)", Cin, Cout);
    CHECK_HIP(urnn_launch_pack_deconv(weight, bias, packed, Cin, Cout, (hipStream_t)stream), "pack_deconv");
    return URNN_OK;
}

// ---- stage conv ------------------------------------------------------------------------------------------------------
static int stage_conv_impl(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, int pool, float slope,
                           int wide, void *stream, const float *stem_w = nullptr, float *stem_partial = nullptr);

extern "C" int urnn_stage_conv_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W,
                                   int pool, float slope, void *stream)
{
    return stage_conv_impl(in, packed, out, B, Cin, Cout, H, W, pool, slope, 0, stream);
}

// wide = 1: `in` holds gradients (the input-gradient GEMMs of the backward pass): bf16 x 6 split instead of f16 x 3
// The flat conv to the head's 16 channels with the head's first LayerNorm statistics taken in its epilogue (conv_gemm_kernel, stemW): the
// 128-pixel-tile form of the kernel only -- planes of >= 131 072 pixels per launch with P % 4 == 0, the f16-piece matrix modes
static bool stage_conv_stem_ok(int B, int Cin, int Cout, int H, int W)
{
    if (B < 1 || Cin < 1 || Cout != 16 || H < 1 || W < 1) return false;
    const int mm = urnn_get_matrix_mode();
    if (mm != URNN_MATRIX_FP32 && mm != URNN_MATRIX_FP32_CAND) return false;
    int pb, map;
    pick_tile((long)B * H * W, urnn_conv_ng(Cout), (long)H * W, &pb, &map, "URNN_TUNE_PB_CONV", 1024);
    return pb == 4 && map == MAP_VEC && urnn_conv_nb(Cout) == 1;
}

extern "C" int urnn_stage_conv_stem_applies(int B, int Cin, int Cout, int H, int W) { return stage_conv_stem_ok(B, Cin, Cout, H, W) ? 1 : 0; }

extern "C" int urnn_stage_conv_stem_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, float slope,
                                        const float *head_conv_w, float *head_partial0, void *stream)
{
    if (!head_conv_w || !head_partial0) return fail(URNN_ENULL, "urnn_stage_conv_stem_f32: NULL head_conv_w / head_partial0");
    if (!stage_conv_stem_ok(B, Cin, Cout, H, W))
        return fail(URNN_EINVAL, "urnn_stage_conv_stem_f32: this conv does not take the 128-pixel-tile form to 16 channels (urnn_stage_conv_stem_applies)");
    return stage_conv_impl(in, packed, out, B, Cin, Cout, H, W, 0, slope, 0, stream, head_conv_w, head_partial0);
}

static int stage_conv_impl(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W, int pool, float slope,
                           int wide, void *stream, const float *stem_w, float *stem_partial)
{
    if (!in || !packed || !out) return fail(URNN_ENULL, "urnn_stage_conv_f32: NULL argument");
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_stage_conv_f32: bad dims");
    if ((Cout + 31) / 32 > 12) return fail(URNN_EINVAL, "urnn_stage_conv_f32: Cout=%d > 384 unsupported", Cout);
    if (pool && (H < 2 || W < 2)) return fail(URNN_EINVAL, "urnn_stage_conv_f32: pool needs H,W >= 2");
    if (!aligned16(in) || !aligned16(out) || !aligned16(packed)) return fail(URNN_EALIGN, "urnn_stage_conv_f32: pointers must be 16-byte aligned");
    const long P = (long)H * W;
    if (!plane_fits(P, Cin > Cout ? Cin : Cout)) return fail(URNN_EINVAL, "urnn_stage_conv_f32: %d x %d plane with %d channels exceeds the 4-GiB segment limit", H, W, Cin > Cout ? Cin : Cout);
    const int NB = urnn_conv_nb(Cout), NG = urnn_conv_ng(Cout);
    ConvGemmParams p = {};
    p.seg[0] = p.seg[1] = p.seg[2] = in;
    p.segC[0] = p.segC[1] = p.segC[2] = Cin;
    p.segKp0[0] = 0;
    p.segKp0[1] = p.segKp0[2] = INT_MAX;
    p.kpBegin = 0;
    p.KT = (Cin + 1) / 2;
    p.hKp0 = INT_MAX;
    p.wt = packed;
    p.aFloats = (int)slab_floats(p.KT, NB);
    p.NG = NG;
    p.bias = packed + (size_t)NG * p.aFloats;
    p.wsplit = reinterpret_cast<const unsigned *>(p.bias + (size_t)NG * NB * 32);
    p.sDwords = urnn_split_slab_dwords(p.KT, NB);
    p.wf16 = p.wsplit + (size_t)NG * p.sDwords;
    p.fDwords = urnn_f16_slab_dwords(p.KT, NB);
    p.P = (int)P;
    p.W = W;
    p.Cout = Cout;
    p.slope = slope;
    p.out0 = out;
    p.wide = wide;
    p.stemW = stem_w;
    p.stemPart = stem_partial;
    hipStream_t st = (hipStream_t)stream;
    if (pool) {
        p.W2 = W / 2;
        p.P2 = (H / 2) * (W / 2);
        CHECK_HIP(urnn_launch_conv_pool(p, B, st), "stage_conv(pool)");
    } else {
        int pb, map;
        pick_tile((long)B * P, NG, P, &pb, &map, "URNN_TUNE_PB_CONV", 1024);   // thin-N conv: fewer, fuller DMAs win (19 vs 27 us)
        CHECK_HIP(urnn_launch_conv_flat(p, B, pb, map, st), "stage_conv");
    }
    return URNN_OK;
}

// ---- GRU cell --------------------------------------------------------------------------------------------------------
struct GruWs {
    int *status;        // first URNN_STATUS_BYTES of every cell / head workspace (urnn_common.h): the same word whatever the cell's shape
    float *g1, *cx, *part1, *part2, *ss1, *ss2;
    float *st1, *st2;   // per (sample, norm group) (mean, rstd) of the gates / the candidate: read by the backward pass
    size_t bytes;
};

static GruWs carve_gru(void *base, int B, int F, long P)
{
    // partial buffers are sized for the smallest tile (32 pixels) so any PB choice fits
    const size_t tiles = (size_t)((P + 31) / 32) < 2 ? 2 : (size_t)((P + 31) / 32);   // >= 2: the strip mode's two pseudo-tiles
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float *p = base ? reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) : nullptr;
        off += align_up(nfloats * sizeof(float), 256);
        return p;
    };
    GruWs w;
    w.status = reinterpret_cast<int *>(take(URNN_STATUS_BYTES / sizeof(float)));
    w.g1 = take((size_t)B * 2 * F * P);
    w.cx = take((size_t)B * F * P);
    w.part1 = take((size_t)B * (2 * F / 32) * tiles * 2);
    w.part2 = take((size_t)B * (F / 32) * tiles * 2);
    w.ss1 = take((size_t)B * 2 * F * 2);
    w.ss2 = take((size_t)B * F * 2);
    w.st1 = take((size_t)B * (2 * F / 32) * 2);
    w.st2 = take((size_t)B * (F / 32) * 2);
    w.bytes = off;
    return w;
}

extern "C" size_t urnn_gru_cell_workspace_bytes(int B, int F, int H, int W)
{
    if (B < 1 || F < 1 || H < 1 || W < 1) return 0;
    return carve_gru(nullptr, B, F, (long)H * W).bytes;
}

// tiles per sample of the gate GEMM (which = 1) / the candidate GEMM (which = 2): the layout of their GroupNorm partials
static int gru_tiles(int B, int F, long P, int which, int *pb_out = nullptr, int *map_out = nullptr)
{
    int pb, map;
    if (which == 1) pick_tile((long)B * P, F / 32, P, &pb, &map, "URNN_TUNE_PB_GATES", 1024);
    else pick_tile((long)B * P, (F / 32) / urnn_cand_nb(F), P, &pb, &map, "URNN_TUNE_PB_CAND", 1024);
    if (pb_out) *pb_out = pb;
    if (map_out) *map_out = map;
    return (int)((P + 32 * pb - 1) / (32 * pb));
}

// 32-pixel tiles (small planes: the quarter-resolution cells, the F = 96 candidates at half resolution) take the activation-
// stationary kernels of urnn_small.hip up to URNN_TUNE_SMALL pixels per launch (default 24 000; 0 disables)
static bool small_on_gates(int B, long P)
{
    static const long small_max = urnn_tune("URNN_TUNE_SMALL", 24000);
    return (long)B * P <= small_max;
}

// Does URNN_PHASE_FUSED_R take effect for a cell of this shape (skip: Skip-ConvGRU with an e input of F channels; x present)?
extern "C" int urnn_gru_cell_fused_reset_gate_applies(int B, int I, int F, int H, int W, int skip)
{
    if (B < 1 || I < 1 || F < 32 || H < 1 || W < 1) return 0;
    const FusedCandLayout fu = urnn_fused_cand_layout(I, F, skip);
    const long P = (long)H * W;
    if (!fu.ok || small_on_gates(B, P)) return 0;
    static const unsigned dummy = 0;
    ConvGemmParams p = {};
    const int Ie = (I + 1) & ~1;
    p.F = F;
    p.P = (int)P;
    p.KT = (Ie + (skip ? F : 0) + F) / 2;
    p.kpBegin = 0;
    p.hKp0 = (Ie + (skip ? F : 0)) / 2;
    p.wfused = &dummy;
    p.fu1Dwords = fu.dw1;
    p.fu2Dwords = fu.dw2;
    return urnn_cand_fused_plan(p, B) != 0;
}

// the consumer of the cell's new state, fused with the cell's last kernel (urnn_gru_cell_tail_f32)
struct TailArgs {
    const float *conv_packed;
    int Cout, pool;
    float slope;
    float *conv_out;
    const float *head_w;
    float *head_partial0;
};

// forward declaration: the cell entry with a "plan only" switch (coop_blocks != nullptr: report, launch nothing)
static int gru_cell_impl(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                         const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                         size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask, long global_pixels,
                         void *stream, int *coop_blocks = nullptr, const TailArgs *tail = nullptr);

// Blocks of the cooperative launch a cell of this shape would take under URNN_PHASE_COOP (0: it would run its three kernels).  More than
// 128 blocks means the launch needs more than half of the chip's CUs to itself: a caller that keeps SEVERAL kernel chains in flight
// passes the flag only up to 128 blocks (two larger ones waiting for CUs at their grid barriers could starve each other; up to 128
// blocks any two fit side by side).
extern "C" int urnn_gru_cell_coop_blocks(int B, int I, int F, int H, int W, int skip, int has_x)
{
    if (B < 1 || I < 1 || F < 32 || F % 32 != 0 || F > 128 || H < 1 || W < 1) return 0;
    int blocks = 0;
    alignas(16) static float dummy[4];
    const float *d = dummy;
    gru_cell_impl(has_x ? d : nullptr, skip ? d : nullptr, d, d, d, d, d, d, dummy, dummy, (size_t)1 << 62, B, I, F, H, W, 1e-5f,
                  URNN_PHASE_ALL | URNN_PHASE_COOP, 0, nullptr, &blocks);
    return blocks;
}

// global_pixels > 0: this call computes one horizontal STRIP of a plane of global_pixels pixels that is split over ranks
// (SURVEY 8e); the GroupNorm partials have been replaced by the all-reduced totals (two pseudo-tiles: hi + lo floats of
// the double sums, urnn_gru_cell_strip_stats_f32) and the statistics are over the whole plane
static int gru_cell_impl(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                         const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                         size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask, long global_pixels,
                         void *stream, int *coop_blocks, const TailArgs *tail)
{
    if (!h || !packed || !gn1_w || !gn1_b || !gn2_w || !gn2_b || !h_out || !workspace)
        return fail(URNN_ENULL, "urnn_gru_cell_f32: NULL argument");
    if (B < 1 || I < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_gru_cell_f32: bad dims");
    if (F < 32 || F % 32 != 0 || F > 128)
        return fail(URNN_EINVAL, "urnn_gru_cell_f32: hidden channels F=%d must be a multiple of 32 in [32,128]", F);
    if (!aligned16(h) || !aligned16(h_out) || !aligned16(packed) || !aligned16(workspace) || (x && !aligned16(x)) || (e && !aligned16(e)))
        return fail(URNN_EALIGN, "urnn_gru_cell_f32: pointers must be 16-byte aligned");
    const long P = (long)H * W;
    if (!plane_fits(P, 2 * F > I ? 2 * F : I)) return fail(URNN_EINVAL, "urnn_gru_cell_f32: %d x %d plane exceeds the 4-GiB segment limit", H, W);
    const GruWs ws = carve_gru(workspace, B, F, P);
    if (workspace_bytes < ws.bytes)
        return fail(URNN_EWORKSPACE, "urnn_gru_cell_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int skip = e != nullptr;
    const int Ie = (I + 1) & ~1;
    const int KT = (Ie + (skip ? F : 0) + F) / 2;
    const int NW = F / 32;

    // K1: raw gates z | r = W1 . [x; e; h] + b1, GroupNorm partials
    ConvGemmParams p = {};
    p.seg[0] = x ? x : h;  // x == nullptr: the segment is skipped via kpBegin, the pointer is never dereferenced
    p.segC[0] = I;
    p.segKp0[0] = 0;
    if (skip) {
        p.seg[1] = e; p.segC[1] = F; p.segKp0[1] = Ie / 2;
        p.seg[2] = h; p.segC[2] = F; p.segKp0[2] = Ie / 2 + F / 2;
    } else {
        p.seg[1] = h; p.segC[1] = F; p.segKp0[1] = Ie / 2;
        p.seg[2] = h; p.segC[2] = F; p.segKp0[2] = INT_MAX;
    }
    p.hKp0 = INT_MAX;
    p.kpBegin = x ? 0 : Ie / 2;
    p.KT = KT;
    p.wt = packed;
    p.aFloats = (int)slab_floats(KT, 2);
    p.NG = NW;
    p.bias = packed + (size_t)NW * p.aFloats;
    const int NB2s = urnn_cand_nb(F);
    const float *split0 = packed + (size_t)NW * p.aFloats + 2 * F + (size_t)(NW / NB2s) * slab_floats(KT, NB2s) + F;
    p.wsplit = reinterpret_cast<const unsigned *>(split0);
    p.sDwords = urnn_split_slab_dwords(KT, 2);
    // f16 forms: the gate slab in the grouping of urnn_gate_groups (wide groups: the K input planes pass through the CUs fewer times)
    const GateGroups gg = urnn_gate_groups(F, KT);
    p.wf16 = p.wsplit + (size_t)NW * p.sDwords + (size_t)(NW / NB2s) * urnn_split_slab_dwords(KT, NB2s);
    p.fDwords = urnn_f16_slab_dwords(KT, gg.NB);
    p.NGf = gg.NG; p.NBf = gg.NB; p.gHalves = gg.halves; p.gGS = gg.GS;
    p.biasf = reinterpret_cast<const float *>(p.wf16 + (size_t)gg.NG * p.fDwords + (size_t)(NW / NB2s) * urnn_f16_slab_dwords(KT, NB2s));
    p.P = (int)P;
    p.W = W;
    p.F = F;
    p.Cout = 2 * F;
    p.out0 = ws.g1;
    p.partial = ws.part1;
    p.status = ws.status;
    int pb1, map1;
    // the gate GEMM prefers 128-pixel tiles (one 1-KiB DMA per 8 MFMAs) even when they only fill half the wave slots; the grouped
    // f16 kernel (4 or 3 n-blocks per wave) takes 64-pixel ones.  Strips keep the F/32-group kernel (their statistics exchange
    // is written against its tile layout).
    int tiles1 = gru_tiles(B, F, P, 1, &pb1, &map1);
    const bool small_on = small_on_gates(B, P);
    const bool small_gates = small_on && pb1 == 1 && urnn_small_ok(p, 2 * NW, 0);
    if (global_pixels > 0 && gg.NB != 2) { p.wf16 = nullptr; p.fDwords = 0; }        // strips: no f16 form in the F/32 grouping
    if (!small_gates && global_pixels <= 0) {
        urnn_gate_plan(p, B, pb1, map1, &pb1, &map1);
        tiles1 = (int)((P + 32 * pb1 - 1) / (32 * pb1));
    }
    const int ftiles1 = global_pixels > 0 ? 2 : tiles1;                              // tiles the finalizes read
    // URNN_PHASE_FUSED_R: the reset gate is recomputed inside the candidate kernel and its raw planes are never stored -- when the
    // cell has that form (F = 64 on a plane of >= 65 536 pixels with P % 4 == 0, f16 arithmetic, slabs + rings within the LDS);
    // otherwise the flag is ignored and the three-pass cell runs.  Both phase-split calls of one cell take the same decision.
    ConvGemmParams fz = p;
    bool fused_r = false;
    {
        const FusedCandLayout fu = urnn_fused_cand_layout(I, F, skip);
        if ((phase_mask & URNN_PHASE_FUSED_R) && fu.ok && global_pixels <= 0 && !small_gates && p.biasf) {
            fz.hKp0 = Ie / 2 + (skip ? F / 2 : 0);
            fz.wfused = reinterpret_cast<const unsigned *>(p.biasf + 2 * F);
            fz.fu1Dwords = fu.dw1;
            fz.fu2Dwords = fu.dw2;
            fz.biasfu = reinterpret_cast<const float *>(fz.wfused + fu.dw1 + fu.dw2);
            fused_r = urnn_cand_fused_plan(fz, B) != 0;
        }
    }
    p.zOnly = fused_r ? 1 : 0;
    const double count = 32.0 * (double)(global_pixels > 0 ? global_pixels : P);     // values per (sample, norm group)
    // 32-pixel tiles (small planes: the quarter-resolution cells, the F = 96 candidates at half resolution) take the
    // activation-stationary kernels of urnn_small.hip: same outputs, same partial layout (development knob URNN_TUNE_SMALL=0)
    // up to URNN_TUNE_SMALL pixels per launch (default 24 000; 0 disables): at 62 500 pixels the per-block weight stream and
    // prologue cost more than they save (candidate GEMMs 38 -> 49 and 56 -> 110 us)

    // K2: candidate (pre-norm) = W2 . [x; e; sigmoid(GN(r)) * h] + b2, GroupNorm partials.  The hidden-state rows are gated
    // on the fly from the raw reset gate and K1's folded (scale, shift).
    const int NB2 = urnn_cand_nb(F), NG2 = NW / NB2;
    ConvGemmParams c = p;
    if (!skip) { c.segKp0[1] = INT_MAX; }
    c.segKp0[2] = INT_MAX;                       // h only enters through the gated slots
    c.hKp0 = Ie / 2 + (skip ? F / 2 : 0);
    c.gate = ws.g1;
    c.gpart = ws.part1;
    c.gtiles = ftiles1;
    c.gtilePix = global_pixels > 0 ? 0 : 32 * pb1;     // strip mode: raw all-reduced totals
    c.gcount = count;
    c.gn_w = gn1_w;
    c.gn_b = gn1_b;
    c.eps = eps;
    c.ss_out = ws.ss1;
    c.stat_out = ws.st1;
    c.wt = packed + (size_t)NW * p.aFloats + 2 * F;
    c.aFloats = (int)slab_floats(KT, NB2);
    c.NG = NG2;
    c.bias = c.wt + (size_t)NG2 * c.aFloats;
    c.wsplit = p.wsplit + (size_t)NW * p.sDwords;
    c.sDwords = urnn_split_slab_dwords(KT, NB2);
    c.wf16 = p.wsplit + (size_t)NW * p.sDwords + (size_t)(NW / NB2s) * urnn_split_slab_dwords(KT, NB2s) + (size_t)gg.NG * urnn_f16_slab_dwords(KT, gg.NB);
    c.fDwords = urnn_f16_slab_dwords(KT, NB2);
    c.biasf = nullptr;
    c.Cout = F;
    c.out0 = ws.cx;
    c.partial = ws.part2;
    // a rollout's full-resolution cell that cannot take the fused kernel (P % 4 != 0, too few tiles, F != 64): the candidate GEMM on the
    // fp32 instruction, so that the long-rollout behaviour does not depend on the grid's shape
    if ((phase_mask & URNN_PHASE_FUSED_R) && !fused_r && global_pixels <= 0 && P >= URNN_FULL_RES_PIXELS &&
        urnn_get_matrix_mode() == URNN_MATRIX_FP32)
        c.candExact = 1;
    // ... and a STRIP of a full-resolution plane (inference only): the same arithmetic for that product as the single-chip
    // rollout of the same grid, so that the multi-chip and the single-chip trajectories agree over a long event
    if (global_pixels >= URNN_FULL_RES_PIXELS && urnn_get_matrix_mode() == URNN_MATRIX_FP32) c.candExact = 1;
    int pb2, map2;
    int tiles2 = gru_tiles(B, F, P, 2, &pb2, &map2);
    // half-resolution planes: the two-stream candidate on 64-pixel tiles (urnn_cand_gated.hip).  Decided from the shapes and the process-wide
    // matrix mode (urnn_cand_gated_plan), so phase-split callers see the same tile size in the candidate and in the blend's fold AS LONG AS
    // every phase of the cell runs under the same mode (include/urnn_hip.h "One device per process ...")
    const bool gated2 = !fused_r && global_pixels <= 0 && !small_on && urnn_cand_gated_plan(c, B) != 0;
    if (gated2) {
        pb2 = 2;
        tiles2 = (int)((P + 63) / 64);
    }
    // URNN_PHASE_COOP on a half-resolution plane (more 64-pixel tiles than CUs, at most four per CU): ONE cooperative launch with the gates
    // and the candidate resident in the accumulators of persistent blocks (urnn_coop_tiles.hip)
    if ((phase_mask & URNN_PHASE_COOP) && !tail && (phase_mask & URNN_PHASE_ALL) == URNN_PHASE_ALL && global_pixels <= 0 && !fused_r) {
        const int nb = urnn_coop_tiles_blocks(p, c, B);
        if (nb > 0) {
            if (coop_blocks) {                                      // plan only (urnn_gru_cell_coop_blocks)
                *coop_blocks = nb;
                return URNN_OK;
            }
            CHECK_HIP(urnn_launch_coop_tiles(p, c, gn2_w, gn2_b, ws.ss2, ws.st2, h, h_out, reinterpret_cast<unsigned *>(ws.status) + 256, B, st),
                      "gru cell (one cooperative launch, four tiles per block)");
            return URNN_OK;
        }
    }
    // URNN_PHASE_COOP: the whole cell of a small plane as ONE cooperative launch (urnn_small.hip coop_cell_kernel) -- when the caller
    // asked for every phase and the shape qualifies; otherwise the flag is ignored and the three kernels run
    if ((phase_mask & URNN_PHASE_COOP) && !tail && (phase_mask & URNN_PHASE_ALL) == URNN_PHASE_ALL && global_pixels <= 0 && !fused_r && small_gates &&
        small_on && pb2 == 1 && urnn_coop_cell_ok(p, c, B)) {
        if (coop_blocks) {                                          // plan only (urnn_gru_cell_coop_blocks)
            *coop_blocks = B * (int)((P + 63) / 64);
            return URNN_OK;
        }
        CHECK_HIP(urnn_launch_coop_cell(p, c, gn2_w, gn2_b, ws.ss2, h_out, reinterpret_cast<unsigned *>(ws.status) + 256, B, st), "gru cell (one cooperative launch)");
        return URNN_OK;
    }
    if (coop_blocks) return URNN_OK;                                // plan only: not a cooperative launch (*coop_blocks stays 0)
    if (phase_mask & URNN_PHASE_GATES) {
        if (small_gates) CHECK_HIP(urnn_launch_small_gates(p, B, st), "gru gates (small plane)");
        else CHECK_HIP(urnn_launch_gru1(p, B, pb1, map1, st), "gru gates");
    }
    // GroupNorm finalise of the gates: folded into the candidate GEMM's prologue; launched on its own only when asked for
    // without the candidate phase (profiling)
    if ((phase_mask & URNN_PHASE_GN1) && !(phase_mask & URNN_PHASE_CAND))
        CHECK_HIP(urnn_launch_gn_finalize(ws.part1, ftiles1, global_pixels > 0 ? 0 : 32 * pb1, (int)P, count, gn1_w, gn1_b, eps, ws.ss1, ws.st1, B,
                                          2 * F, ws.status, URNN_STATUS_GATES, st), "gn finalize 1");
    if (fused_r) {
        pb2 = 2;                                                    // the fused kernel's 64-pixel tiles carry the candidate's partials
        tiles2 = (int)((P + 63) / 64);
        fz.zOnly = 0;
        fz.gpart = c.gpart; fz.gtiles = c.gtiles; fz.gtilePix = c.gtilePix; fz.gcount = c.gcount;
        fz.gn_w = c.gn_w; fz.gn_b = c.gn_b; fz.eps = c.eps; fz.ss_out = c.ss_out; fz.stat_out = c.stat_out;
        fz.Cout = F; fz.out0 = ws.cx; fz.partial = ws.part2;
        if (phase_mask & URNN_PHASE_CAND) CHECK_HIP(urnn_launch_cand_fused(fz, B, st), "gru candidate (reset gate recomputed)");
    } else
    if (phase_mask & URNN_PHASE_CAND) {
        if (gated2) CHECK_HIP(urnn_launch_cand_gated(c, B, st), "gru candidate (64-pixel tiles, group-wise ring)");
        else if (small_on && pb2 == 1 && urnn_small_ok(c, NW, 1)) CHECK_HIP(urnn_launch_small_cand(c, B, st), "gru candidate (small plane)");
        else CHECK_HIP(urnn_launch_cand(c, B, pb2, map2, st), "gru candidate");
    }
    // K3: GroupNorm finalize of the candidate + blend.  One launch when both are asked for (the product path); separate
    // launches for phase-split callers (profiling, strips: the statistics are exchanged in between)
    static const bool fuse_on = urnn_tune("URNN_TUNE_FUSE_BLEND", 1) != 0;   // development knob
    const bool fused = fuse_on && (phase_mask & URNN_PHASE_GN2) && (phase_mask & URNN_PHASE_BLEND) && global_pixels <= 0;
    if (tail && (phase_mask & URNN_PHASE_GN2) && (phase_mask & URNN_PHASE_BLEND)) {
        // finalize + blend + the consumer's 1x1 conv in one launch (urnn_tail.hip); the caller checked urnn_gru_cell_tail_applies
        const int Cout = tail->Cout;
        const int NBc = urnn_conv_nb(Cout), NGc = urnn_conv_ng(Cout), KTc = (F + 1) / 2;
        const float *cb = tail->conv_packed + (size_t)NGc * slab_floats(KTc, NBc);
        TailParams tp = {};
        tp.g1 = ws.g1; tp.cx = ws.cx; tp.h = h; tp.h_out = h_out; tp.ss1 = ws.ss1;
        tp.partial2 = ws.part2; tp.ntiles2 = tiles2; tp.tile_pix2 = 32 * pb2; tp.count = count;
        tp.gn2_w = gn2_w; tp.gn2_b = gn2_b; tp.eps = eps; tp.ss2_out = ws.ss2; tp.stat2_out = ws.st2; tp.status = ws.status;
        tp.B = B; tp.F = F; tp.P = (int)P; tp.W = W;
        tp.wf16 = reinterpret_cast<const unsigned *>(cb + (size_t)NGc * NBc * 32) + (size_t)NGc * urnn_split_slab_dwords(KTc, NBc);
        tp.fDwords = urnn_f16_slab_dwords(KTc, NBc);
        tp.wDwords = NGc * tp.fDwords;
        tp.NBc = NBc; tp.Cout = Cout; tp.bias = cb; tp.slope = tail->slope; tp.out = tail->conv_out;
        tp.head_w = tail->pool ? nullptr : tail->head_w;
        tp.partial0 = tail->head_partial0;
        CHECK_HIP(urnn_launch_tail(tp, H, tail->pool, st), "gru finalize + blend + consumer conv");
        return URNN_OK;
    }
    if (fused) {
        CHECK_HIP(urnn_launch_blend_fin(ws.g1, ws.cx, h, ws.ss1, h_out, B, F, (int)P, ws.part2, tiles2, 32 * pb2, count, gn2_w, gn2_b, eps, ws.ss2,
                                        ws.st2, ws.status, st), "gru finalize + blend");
        return URNN_OK;
    }
    if (phase_mask & URNN_PHASE_GN2)
        CHECK_HIP(urnn_launch_gn_finalize(ws.part2, global_pixels > 0 ? 2 : tiles2, global_pixels > 0 ? 0 : 32 * pb2, (int)P, count, gn2_w, gn2_b,
                                          eps, ws.ss2, ws.st2, B, F, ws.status, URNN_STATUS_CAND, st), "gn finalize 2");
    if (phase_mask & URNN_PHASE_BLEND) CHECK_HIP(urnn_launch_blend(ws.g1, ws.cx, h, ws.ss1, ws.ss2, h_out, B, F, (int)P, st), "gru blend");
    return URNN_OK;
}

extern "C" int urnn_gru_cell_phases_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                                        const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                                        size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask,
                                        void *stream)
{
    return gru_cell_impl(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, h_out, workspace, workspace_bytes, B, I, F, H, W, eps, phase_mask, 0,
                         stream);
}

// 1 when the end of a cell on (B, F, H, W) can be fused with a 1x1 conv F -> Cout (pool: + AvgPool2) under the current matrix mode
extern "C" int urnn_gru_cell_tail_applies(int B, int F, int H, int W, int Cout, int pool)
{
    if (B < 1 || F < 32 || H < 1 || W < 1 || Cout < 1) return 0;
    return urnn_tail_ok(B, F, H, W, F, Cout, pool) ? 1 : 0;
}

extern "C" size_t urnn_head_tail_partial_floats(int B, int H, int W)
{
    if (B < 1 || H < 1 || W < 1) return 0;
    return (size_t)B * (((size_t)H * W + 127) / 128) * 2;
}

extern "C" int urnn_gru_cell_tail_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                                      const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                                      size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask,
                                      const float *conv_packed, int Cout, int pool, float slope, float *conv_out,
                                      const float *head_conv_w, float *head_partial0, void *stream)
{
    if (!conv_packed || !conv_out) return fail(URNN_ENULL, "urnn_gru_cell_tail_f32: NULL consumer conv");
    if ((head_conv_w != nullptr) != (head_partial0 != nullptr)) return fail(URNN_EINVAL, "urnn_gru_cell_tail_f32: head_conv_w and head_partial0 go together");
    if (head_conv_w && (pool || Cout != 16)) return fail(URNN_EINVAL, "urnn_gru_cell_tail_f32: the head statistics need the flat 16-channel feature map");
    if (!aligned16(conv_packed)) return fail(URNN_EALIGN, "urnn_gru_cell_tail_f32: conv_packed must be 16-byte aligned");
    if (!urnn_tail_ok(B, F, H, W, F, Cout, pool))
        return fail(URNN_EINVAL, "urnn_gru_cell_tail_f32: no fused form for F=%d -> %d (pool=%d) on %dx%d in this matrix mode (urnn_gru_cell_tail_applies)", F, Cout, pool, H, W);
    const TailArgs t = {conv_packed, Cout, pool, slope, conv_out, head_conv_w, head_partial0};
    return gru_cell_impl(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, h_out, workspace, workspace_bytes, B, I, F, H, W, eps,
                         phase_mask & ~URNN_PHASE_COOP, 0, stream, nullptr, &t);
}

extern "C" int urnn_gru_cell_strip_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                                       const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                                       size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, int phase_mask,
                                       long global_pixels, void *stream)
{
    if (global_pixels < (long)H * W) return fail(URNN_EINVAL, "urnn_gru_cell_strip_f32: global_pixels %ld < the strip's %ld", global_pixels, (long)H * W);
    if ((phase_mask & URNN_PHASE_GATES) && (phase_mask & (URNN_PHASE_CAND | URNN_PHASE_GN1)))
        return fail(URNN_EINVAL, "urnn_gru_cell_strip_f32: the gates' statistics must be exchanged between the gate and the candidate phase");
    if ((phase_mask & URNN_PHASE_CAND) && (phase_mask & URNN_PHASE_GN2))
        return fail(URNN_EINVAL, "urnn_gru_cell_strip_f32: the candidate's statistics must be exchanged before its finalize");
    return gru_cell_impl(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, h_out, workspace, workspace_bytes, B, I, F, H, W, eps, phase_mask,
                         global_pixels, stream);
}

// direction 0: local tile partials -> sums[B][groups][2] (double: sum, sum of squares; fixed order);
// direction 1: (all-reduced) sums -> the two pseudo-tiles the strip finalizes read
extern "C" int urnn_gru_cell_strip_stats_f32(void *workspace, size_t workspace_bytes, int B, int F, int H, int W, int which, int direction,
                                             double *sums, void *stream)
{
    if (!workspace || !sums) return fail(URNN_ENULL, "urnn_gru_cell_strip_stats_f32: NULL argument");
    if (B < 1 || H < 1 || W < 1 || F < 32 || F % 32 != 0 || F > 128 || (which != 1 && which != 2))
        return fail(URNN_EINVAL, "urnn_gru_cell_strip_stats_f32: bad arguments");
    const long P = (long)H * W;
    const GruWs ws = carve_gru(workspace, B, F, P);
    if (workspace_bytes < ws.bytes) return fail(URNN_EWORKSPACE, "urnn_gru_cell_strip_stats_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    const int rows = B * (which == 1 ? 2 * F / 32 : F / 32);
    float *part = which == 1 ? ws.part1 : ws.part2;
    int pb;
    const int tiles = gru_tiles(B, F, P, which, &pb);
    if (direction == 0) CHECK_HIP(urnn_launch_stats_reduce(part, rows, tiles, tiles, 32 * pb, (int)P, 32, sums, (hipStream_t)stream), "strip stats");
    else CHECK_HIP(urnn_launch_stats_scatter(sums, rows, 2, part, (hipStream_t)stream), "strip stats");
    return URNN_OK;
}

extern "C" int urnn_gru_cell_f32(const float *x, const float *e, const float *h, const float *packed, const float *gn1_w,
                                 const float *gn1_b, const float *gn2_w, const float *gn2_b, float *h_out, void *workspace,
                                 size_t workspace_bytes, int B, int I, int F, int H, int W, float eps, void *stream)
{
    return urnn_gru_cell_phases_f32(x, e, h, packed, gn1_w, gn1_b, gn2_w, gn2_b, h_out, workspace, workspace_bytes, B, I, F, H, W,
                                    eps, URNN_PHASE_ALL, stream);
}

// ---- GRU cell backward (training building block) ----------------------------------------------------------------------
struct GruBwdWs {
    float *dy1, *dy2, *rh, *tmpF, *dxe, *wv, *chpart, *chpart2, *coef, *wpart;
    double *sums;
    size_t bytes;
};

// packed "weights" of the three input-gradient GEMMs (cell_bwd_weights_kernel): they only change with the optimizer step,
// so the caller keeps them per cell and asks for a re-pack once per window
struct GruBwdPacks {
    size_t h2, xe, h1, total;     // float offsets
};
static GruBwdPacks gru_bwd_packs(int I, int F, int skip)
{
    GruBwdPacks p;
    p.h2 = 0;
    p.xe = p.h2 + urnn_packed_conv_floats(F, F);
    p.h1 = p.xe + urnn_packed_conv_floats(3 * F, I + (skip ? F : 0));
    p.total = p.h1 + urnn_packed_conv_floats(2 * F, F);
    return p;
}

extern "C" size_t urnn_gru_cell_backward_packed_floats(int I, int F, int skip)
{
    if (I < 1 || F < 32 || F % 32 != 0 || F > 128) return 0;
    return gru_bwd_packs(I, F, skip ? 1 : 0).total;
}

static GruBwdWs carve_gru_bwd(void *base, int B, int I, int F, int skip, long P)
{
    const int K = I + (skip ? F : 0) + F;
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        char *p = base ? reinterpret_cast<char *>(base) + off : nullptr;
        off += align_up(nbytes, 256);
        return p;
    };
    auto takef = [&](size_t nfloats) { return reinterpret_cast<float *>(take(nfloats * sizeof(float))); };
    GruBwdWs w;
    w.dy1 = takef((size_t)B * 2 * F * P);
    w.dy2 = takef((size_t)B * F * P);
    w.rh = takef((size_t)B * F * P);
    w.tmpF = takef((size_t)B * F * P);
    w.dxe = takef((size_t)B * (K - F) * P);
    w.wv = takef((size_t)3 * F * (K - F > F ? K - F : F));
    w.chpart = takef((size_t)B * 2 * F * 64 * 2);      // per-plane partial sums: at most 64 blocks per plane
    w.chpart2 = takef((size_t)B * F * 64 * 2);
    w.coef = takef((size_t)B * (2 * F / 32) * 2);
    {
        const size_t a = urnn_train_wgrad_partial_floats(B, 2 * F, K, (int)P), b = urnn_train_wgrad_partial_floats(B, F, K, (int)P);
        w.wpart = takef(a > b ? a : b);
    }
    w.sums = reinterpret_cast<double *>(take((size_t)B * 2 * F * 2 * sizeof(double)));
    w.bytes = off;
    return w;
}

extern "C" size_t urnn_gru_cell_backward_workspace_bytes(int B, int I, int F, int skip, int H, int W)
{
    if (B < 1 || I < 1 || F < 32 || F % 32 != 0 || F > 128 || H < 1 || W < 1) return 0;
    return carve_gru_bwd(nullptr, B, I, F, skip ? 1 : 0, (long)H * W).bytes;
}

// out (B,Cout,P) = packed 1x1 conv (no bias, identity) of the channel concatenation [in0 (C0) ; in1 (C1)], C0 even
static int conv_2seg(const float *in0, int C0, const float *in1, int C1, const float *packed, float *out, int B, int Cout, int H, int W,
                     hipStream_t st, const char *what)
{
    const long P = (long)H * W;
    const int Cin = C0 + C1;
    const int NB = urnn_conv_nb(Cout), NG = urnn_conv_ng(Cout);
    ConvGemmParams p = {};
    p.seg[0] = in0; p.segC[0] = C0; p.segKp0[0] = 0;
    p.seg[1] = in1 ? in1 : in0; p.segC[1] = in1 ? C1 : C0; p.segKp0[1] = in1 ? C0 / 2 : INT_MAX;
    p.seg[2] = in0; p.segC[2] = C0; p.segKp0[2] = INT_MAX;
    p.kpBegin = 0;
    p.KT = (Cin + 1) / 2;
    p.hKp0 = INT_MAX;
    p.wt = packed;
    p.aFloats = (int)slab_floats(p.KT, NB);
    p.NG = NG;
    p.bias = packed + (size_t)NG * p.aFloats;
    p.wsplit = reinterpret_cast<const unsigned *>(p.bias + (size_t)NG * NB * 32);
    p.sDwords = urnn_split_slab_dwords(p.KT, NB);
    p.wf16 = p.wsplit + (size_t)NG * p.sDwords;
    p.fDwords = urnn_f16_slab_dwords(p.KT, NB);
    p.P = (int)P;
    p.W = W;
    p.Cout = Cout;
    p.slope = 1.0f;
    p.out0 = out;
    p.wide = 1;                                    // gradients: no lower bound on the magnitudes -> bf16 x 6, not f16 x 3
    int pb, map;
    pick_tile((long)B * P, NG, P, &pb, &map, "URNN_TUNE_PB_CONV", 1024);
    CHECK_HIP(urnn_launch_conv_flat(p, B, pb, map, st), what);
    return URNN_OK;
}

// dX = W^T . dY through the forward GEMM kernel: "weight" = W^T (K x N), identity epilogue, no bias.  out (B,K,P).
// pack_transposed builds the packed W^T (once per parameter update), dx_gemm runs the GEMM.
static int pack_transposed(const float *w, float *wt, float *packed, int N, int K, hipStream_t st, const char *what)
{
    CHECK_HIP(urnn_train_transpose(w, wt, N, K, st), what);
    CHECK_HIP(urnn_launch_pack_conv(wt, nullptr, packed, N, K, st), what);
    return URNN_OK;
}

static int dx_gemm(const float *dy, const float *packed, float *out, int B, int N, int K, int H, int W, hipStream_t st)
{
    return stage_conv_impl(dy, packed, out, B, N, K, H, W, 0, 1.0f, 1, st);
}

extern "C" int urnn_gru_cell_backward_f32(const float *x, const float *e, const float *h, const float *W1, const float *W2,
                                          const float *gn1_w, const float *gn2_w, const void *fwd_workspace, const float *dh_out,
                                          const float *dh_out2, const float *dh_out3, const float *dh_out4, float *dx, float *de, float *dh,
                                          float *dh2, float *dW1, float *db1, float *dgn1_w, float *dgn1_b,
                                          float *dW2, float *db2, float *dgn2_w, float *dgn2_b, float *bwd_packed, int repack,
                                          void *workspace, size_t workspace_bytes, int B, int I, int F, int H, int W, int accumulate,
                                          void *stream)
{
    if (!h || !W1 || !W2 || !gn1_w || !gn2_w || !fwd_workspace || !dh_out || !dh || !dW1 || !db1 || !dgn1_w || !dgn1_b || !dW2 || !db2 ||
        !dgn2_w || !dgn2_b || !bwd_packed || !workspace)
        return fail(URNN_ENULL, "urnn_gru_cell_backward_f32: NULL argument");
    if ((x != nullptr) != (dx != nullptr) || (e != nullptr) != (de != nullptr))
        return fail(URNN_EINVAL, "urnn_gru_cell_backward_f32: dx / de must be given exactly when x / e are");
    if (B < 1 || I < 1 || H < 1 || W < 1 || F < 32 || F % 32 != 0 || F > 128)
        return fail(URNN_EINVAL, "urnn_gru_cell_backward_f32: bad dims (F=%d must be a multiple of 32 in [32,128])", F);
    const long P = (long)H * W;
    if (!plane_fits(P, 3 * F > I + F ? 3 * F : I + F)) return fail(URNN_EINVAL, "urnn_gru_cell_backward_f32: plane exceeds the 4-GiB segment limit");
    const int skip = e != nullptr;
    const int K = I + (skip ? F : 0) + F;
    const GruWs fw = carve_gru(const_cast<void *>(fwd_workspace), B, F, P);
    const GruBwdWs ws = carve_gru_bwd(workspace, B, I, F, skip, P);
    if (workspace_bytes < ws.bytes)
        return fail(URNN_EWORKSPACE, "urnn_gru_cell_backward_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    if (!aligned16(bwd_packed)) return fail(URNN_EALIGN, "urnn_gru_cell_backward_f32: bwd_packed must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int Pi = (int)P;
    // rows of [x; e] that need a gradient: a missing x (decoder stage 3) still owns columns of W1 / W2 but gets none
    const int rlo = x ? 0 : I, nrows = (x ? I : 0) + (skip ? F : 0);
    const GruBwdPacks pk = gru_bwd_packs(I, F, skip);
    if (repack) {
        CHECK_HIP(urnn_train_cell_bwd_weights(W1, W2, ws.wv, F, K, 0, 0, 0, st), "input-gradient weights");
        CHECK_HIP(urnn_launch_pack_conv(ws.wv, nullptr, bwd_packed + pk.h2, F, F, st), "input-gradient weights");
        if (nrows > 0) {
            CHECK_HIP(urnn_train_cell_bwd_weights(W1, W2, ws.wv, F, K, rlo, nrows, 1, st), "input-gradient weights");
            CHECK_HIP(urnn_launch_pack_conv(ws.wv, nullptr, bwd_packed + pk.xe, 3 * F, nrows, st), "input-gradient weights");
        }
        CHECK_HIP(urnn_train_cell_bwd_weights(W1, W2, ws.wv, F, K, 0, 0, 2, st), "input-gradient weights");
        CHECK_HIP(urnn_launch_pack_conv(ws.wv, nullptr, bwd_packed + pk.h1, 2 * F, F, st), "input-gradient weights");
    }

    // 1. blend: dy2 (normalised candidate), dy1[:, :F] (normalised update gate), dh = dout * (1 - z)
    CHECK_HIP(urnn_train_blend_bwd(dh_out, dh_out2, dh_out3, dh_out4, fw.g1, fw.cx, h, fw.ss1, fw.ss2, fw.st1, fw.st2, ws.dy2, ws.dy1, dh, ws.chpart, ws.chpart2, B, F,
                                   Pi, st), "blend backward");
    // 2. GroupNorm of the candidate: dy2 -> dc (in place), dgamma2 / dbeta2
    CHECK_HIP(urnn_train_gn_backward(ws.dy2, fw.cx, fw.st2, gn2_w, B, F, Pi, ws.chpart2, ws.sums, ws.coef, dgn2_w, dgn2_b, accumulate, 1, st),
              "GroupNorm 2 backward");
    // 3. conv2: dW2 / db2 = dc . [x; e; r*h]^T;  d(r*h) = W2[:, h]^T . dc  (the x / e rows of W2^T . dc join conv1's in step 7)
    CHECK_HIP(urnn_train_reset_gate(fw.g1, h, fw.ss1, ws.rh, B, F, Pi, st), "reset gate");
    {
        const float *seg[3] = {x, e, ws.rh};
        const int segC[3] = {I, skip ? F : 0, F};
        CHECK_HIP(urnn_train_wgrad(ws.dy2, seg, segC, B, F, K, Pi, ws.wpart, dW2, db2, accumulate, st), "conv2 weight gradient");
    }
    int rc = conv_2seg(ws.dy2, F, nullptr, 0, bwd_packed + pk.h2, ws.tmpF, B, F, H, W, st, "conv2 hidden-state gradient");
    if (rc) return rc;
    // 4. reset gate: d(r*h) -> dy1[:, F:], dh += d(r*h) * r
    CHECK_HIP(urnn_train_reset_gate_bwd(ws.tmpF, (long)F * P, fw.g1, h, fw.ss1, fw.st1, ws.dy1, dh, ws.chpart, B, F, Pi, st),
              "reset gate backward");
    // 5. GroupNorm of the gates: dy1 -> dg (in place), dgamma1 / dbeta1
    CHECK_HIP(urnn_train_gn_backward(ws.dy1, fw.g1, fw.st1, gn1_w, B, 2 * F, Pi, ws.chpart, ws.sums, ws.coef, dgn1_w, dgn1_b, accumulate, 1, st),
              "GroupNorm 1 backward");
    // 6. conv1: dW1 / db1 = dg . [x; e; h]^T
    {
        const float *seg[3] = {x, e, h};
        const int segC[3] = {I, skip ? F : 0, F};
        CHECK_HIP(urnn_train_wgrad(ws.dy1, seg, segC, B, 2 * F, K, Pi, ws.wpart, dW1, db1, accumulate, st), "conv1 weight gradient");
    }
    // 7. d[x; e] = [W1^T | W2^T][xe rows] . [dg; dc] in ONE GEMM (contraction over 3F), written straight into dx | de when they
    //    are one contiguous (B, rows, P) block (always the case for B = 1 when the caller allocates them together)
    if (nrows > 0) {
        float *first = dx ? dx : de;
        const bool direct = !(dx && de) || (B == 1 && de == dx + (size_t)I * P);
        rc = conv_2seg(ws.dy1, 2 * F, ws.dy2, F, bwd_packed + pk.xe, direct ? first : ws.dxe, B, nrows, H, W, st, "input gradient");
        if (rc) return rc;
        if (!direct) {
            const long bs = (long)nrows * P;
            CHECK_HIP(urnn_train_add_slices(dx, (long)I * P, ws.dxe, bs, nullptr, 0, B, I, Pi, 0, st), "dx");
            CHECK_HIP(urnn_train_add_slices(de, (long)F * P, ws.dxe + (size_t)I * P, bs, nullptr, 0, B, F, Pi, 0, st), "de");
        }
    }
    // 8. dh += W1[:, h]^T . dg -- or, when the caller takes dL/dh as two terms (dh2 != NULL: the previous timestep's cell backward
    //    sums its dh_out terms on the fly), the GEMM writes the second term straight into dh2 and the adding pass is gone
    rc = conv_2seg(ws.dy1, 2 * F, nullptr, 0, bwd_packed + pk.h1, dh2 ? dh2 : ws.tmpF, B, F, H, W, st, "conv1 hidden-state gradient");
    if (rc) return rc;
    if (!dh2) CHECK_HIP(urnn_train_add_slices(dh, (long)F * P, ws.tmpF, (long)F * P, nullptr, 0, B, F, Pi, 1, st), "dh");
    return URNN_OK;
}

// ---- stage conv / deconv backward (training building blocks) ----------------------------------------------------------
struct ConvBwdWs {
    float *u, *pk, *wt, *pkt, *wpart, *rows, *drows, *dsum;
    size_t bytes;
};

// N = output channels of the equivalent 1x1 conv (Cout, or 4*Cout for the transposed conv), planes of P pixels
static ConvBwdWs carve_conv_bwd(void *base, int B, int Cin, int N, long P, int deconv)
{
    size_t off = 0;
    auto takef = [&](size_t nfloats) {
        float *p = base ? reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) : nullptr;
        off += align_up(nfloats * sizeof(float), 256);
        return p;
    };
    ConvBwdWs w;
    w.u = takef((size_t)B * N * P);
    w.pk = takef(urnn_packed_conv_floats(Cin, N));
    w.wt = takef((size_t)N * Cin);
    w.pkt = takef(urnn_packed_conv_floats(N, Cin));
    w.wpart = takef(urnn_train_wgrad_partial_floats(B, N, Cin, (int)P));
    w.rows = takef(deconv ? (size_t)N * Cin : 0);
    w.drows = takef(deconv ? (size_t)N * Cin : 0);
    w.dsum = takef(deconv ? (size_t)N : 0);
    w.bytes = off;
    return w;
}

extern "C" size_t urnn_stage_conv_backward_workspace_bytes(int B, int Cin, int Cout, int H, int W)
{
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    return carve_conv_bwd(nullptr, B, Cin, Cout, (long)H * W, 0).bytes;
}

// packed weights of the backward pass, kept by the caller across calls until the parameters change: [forward pack | W^T pack]
static size_t conv_bwd_packed_split(int Cin, int N) { return align_up(urnn_packed_conv_floats(Cin, N) * sizeof(float), 256) / sizeof(float); }

extern "C" size_t urnn_stage_conv_backward_packed_floats(int Cin, int Cout)
{
    if (Cin < 1 || Cout < 1) return 0;
    return conv_bwd_packed_split(Cin, Cout) + urnn_packed_conv_floats(Cout, Cin);
}

extern "C" int urnn_stage_conv_backward_f32(const float *in, const float *weight, const float *bias, const float *dout, float *din,
                                            float *dweight, float *dbias, float *bwd_packed, int repack, void *workspace,
                                            size_t workspace_bytes, int B, int Cin, int Cout, int H, int W, int pool, float slope,
                                            int accumulate, void *stream)
{
    if (!in || !weight || !bias || !dout || !din || !dweight || !dbias || !workspace)
        return fail(URNN_ENULL, "urnn_stage_conv_backward_f32: NULL argument");
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || (pool && (H < 2 || W < 2)))
        return fail(URNN_EINVAL, "urnn_stage_conv_backward_f32: bad dims");
    if (bwd_packed && !aligned16(bwd_packed)) return fail(URNN_EALIGN, "urnn_stage_conv_backward_f32: bwd_packed must be 16-byte aligned");
    const long P = (long)H * W;
    const ConvBwdWs ws = carve_conv_bwd(workspace, B, Cin, Cout, P, 0);
    if (workspace_bytes < ws.bytes)
        return fail(URNN_EWORKSPACE, "urnn_stage_conv_backward_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    // bwd_packed == NULL: packs live in the workspace and are rebuilt by every call
    float *pk = bwd_packed ? bwd_packed : ws.pk, *pkt = bwd_packed ? bwd_packed + conv_bwd_packed_split(Cin, Cout) : ws.pkt;
    if (repack || !bwd_packed) {
        CHECK_HIP(urnn_launch_pack_conv(weight, bias, pk, Cin, Cout, st), "pack");
        const int rc = pack_transposed(weight, ws.wt, pkt, Cout, Cin, st, "input-gradient weights");
        if (rc) return rc;
    }
    // pre-activation u = W.x + b (the forward kernel with an identity epilogue), then du = (un-pooled) dout * lrelu'(u)
    int rc = urnn_stage_conv_f32(in, pk, ws.u, B, Cin, Cout, H, W, 0, 1.0f, stream);
    if (rc) return rc;
    CHECK_HIP(urnn_train_lrelu_pool_bwd(ws.u, dout, B, Cout, H, W, pool, slope, st), "lrelu backward");
    const float *seg[3] = {in, nullptr, nullptr};
    const int segC[3] = {Cin, 0, 0};
    CHECK_HIP(urnn_train_wgrad(ws.u, seg, segC, B, Cout, Cin, (int)P, ws.wpart, dweight, dbias, accumulate, st), "weight gradient");
    return dx_gemm(ws.u, pkt, din, B, Cout, Cin, H, W, st);
}

extern "C" size_t urnn_deconv2x2_backward_workspace_bytes(int B, int Cin, int Cout, int H, int W)
{
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return 0;
    return carve_conv_bwd(nullptr, B, Cin, 4 * Cout, (long)H * W, 1).bytes;
}

extern "C" size_t urnn_deconv2x2_backward_packed_floats(int Cin, int Cout)
{
    if (Cin < 1 || Cout < 1) return 0;
    return urnn_packed_conv_floats(4 * Cout, Cin);
}

extern "C" int urnn_deconv2x2_backward_f32(const float *in, const float *weight, const float *out, const float *dout, float *din,
                                           float *dweight, float *dbias, float *bwd_packed, int repack, void *workspace,
                                           size_t workspace_bytes, int B, int Cin, int Cout, int H, int W, float slope, int accumulate,
                                           void *stream)
{
    if (!in || !weight || !out || !dout || !din || !dweight || !dbias || !workspace)
        return fail(URNN_ENULL, "urnn_deconv2x2_backward_f32: NULL argument");
    if (B < 1 || Cin < 1 || Cout < 1 || Cout > 96 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_deconv2x2_backward_f32: bad dims");
    if (bwd_packed && !aligned16(bwd_packed)) return fail(URNN_EALIGN, "urnn_deconv2x2_backward_f32: bwd_packed must be 16-byte aligned");
    const long P = (long)H * W;
    const int N = 4 * Cout;
    const ConvBwdWs ws = carve_conv_bwd(workspace, B, Cin, N, P, 1);
    if (workspace_bytes < ws.bytes)
        return fail(URNN_EWORKSPACE, "urnn_deconv2x2_backward_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    // the transposed conv is a 1x1 conv to 4*Cout channels (one per output parity) followed by a pixel shuffle
    float *pkt = bwd_packed ? bwd_packed : ws.pkt;
    if (repack || !bwd_packed) {
        CHECK_HIP(urnn_train_deconv_weight_rows(weight, ws.rows, Cin, Cout, st), "deconv weight rows");
        const int rc = pack_transposed(ws.rows, ws.wt, pkt, N, Cin, st, "input-gradient weights");
        if (rc) return rc;
    }
    CHECK_HIP(urnn_train_deconv_unshuffle(dout, out, ws.u, B, Cout, H, W, slope, st), "deconv un-shuffle");
    const float *seg[3] = {in, nullptr, nullptr};
    const int segC[3] = {Cin, 0, 0};
    CHECK_HIP(urnn_train_wgrad(ws.u, seg, segC, B, N, Cin, (int)P, ws.wpart, ws.drows, ws.dsum, 0, st), "weight gradient");
    CHECK_HIP(urnn_train_deconv_rows_weight(ws.drows, ws.dsum, dweight, dbias, Cin, Cout, accumulate, st), "weight gradient layout");
    return dx_gemm(ws.u, pkt, din, B, N, Cin, H, W, st);
}

// ---- deconv ----------------------------------------------------------------------------------------------------------
extern "C" int urnn_deconv2x2_f32(const float *in, const float *packed, float *out, int B, int Cin, int Cout, int H, int W,
                                  float slope, void *stream)
{
    if (!in || !packed || !out) return fail(URNN_ENULL, "urnn_deconv2x2_f32: NULL argument");
    if (B < 1 || Cin < 1 || Cout < 1 || Cout > 96 || H < 1 || W < 1)
        return fail(URNN_EINVAL, "urnn_deconv2x2_f32: bad dims (Cout must be <= 96, got %d)", Cout);
    if (!aligned16(in) || !aligned16(out) || !aligned16(packed)) return fail(URNN_EALIGN, "urnn_deconv2x2_f32: pointers must be 16-byte aligned");
    const long P = (long)H * W;
    if (!plane_fits(4 * P, Cin > Cout ? Cin : Cout))
        return fail(URNN_EINVAL, "urnn_deconv2x2_f32: %d x %d output plane exceeds the 4-GiB segment limit", 2 * H, 2 * W);
    const int NB = 2 * ((Cout + 31) / 32);
    ConvGemmParams p = {};
    p.seg[0] = p.seg[1] = p.seg[2] = in;
    p.segC[0] = p.segC[1] = p.segC[2] = Cin;
    p.segKp0[0] = 0;
    p.segKp0[1] = p.segKp0[2] = INT_MAX;
    p.kpBegin = 0;
    p.KT = (Cin + 1) / 2;
    p.hKp0 = INT_MAX;
    p.wt = packed;
    p.aFloats = (int)slab_floats(p.KT, NB);
    p.NG = 2;
    p.bias = packed + (size_t)2 * p.aFloats;
    p.wsplit = reinterpret_cast<const unsigned *>(p.bias + (size_t)2 * NB * 32);
    p.sDwords = urnn_split_slab_dwords(p.KT, NB);
    p.wf16 = p.wsplit + (size_t)2 * p.sDwords;
    p.fDwords = urnn_f16_slab_dwords(p.KT, NB);
    p.P = (int)P;
    p.W = W;
    p.Cout = Cout;
    p.slope = slope;
    p.out0 = out;
    // pairs of horizontally adjacent pixels need an even width; tiny planes use 32-pixel strided tiles to fill the chip
    const bool big = ((long)B * P / 64) * 2 >= 1024;
    const bool pair = big && (W % 2) == 0;
    CHECK_HIP(urnn_launch_deconv(p, B, pair ? 2 : 1, pair ? (P % 4 == 0 ? MAP_PAIR16 : MAP_PAIR) : MAP_STRIDED, (hipStream_t)stream),
              "deconv2x2");
    return URNN_OK;
}

// ---- weight gradient of a 1x1 conv (the building block of every layer backward) ----------------------------------------------------
extern "C" size_t urnn_weight_gradient_workspace_bytes(int B, int N, int K, int H, int W)
{
    if (B < 1 || N < 1 || K < 1 || H < 1 || W < 1) return 0;
    return align_up(urnn_train_wgrad_partial_floats(B, N, K, H * W) * sizeof(float), 256);
}

extern "C" int urnn_weight_gradient_f32(const float *dy, const float *seg0, int C0, const float *seg1, int C1, const float *seg2, int C2,
                                        float *dW, float *db, void *workspace, size_t workspace_bytes, int B, int N, int H, int W,
                                        int accumulate, void *stream)
{
    if (!dy || !seg0 || !dW || !workspace) return fail(URNN_ENULL, "urnn_weight_gradient_f32: NULL argument");
    if (B < 1 || N < 1 || C0 < 1 || C1 < 0 || C2 < 0 || H < 1 || W < 1 || (C1 > 0) != (seg1 != nullptr) || (C2 > 0) != (seg2 != nullptr))
        return fail(URNN_EINVAL, "urnn_weight_gradient_f32: bad dims");
    const int K = C0 + C1 + C2;
    const long P = (long)H * W;
    if (!plane_fits(P, N > K ? N : K)) return fail(URNN_EINVAL, "urnn_weight_gradient_f32: plane exceeds the 4-GiB segment limit");
    if (workspace_bytes < urnn_weight_gradient_workspace_bytes(B, N, K, H, W))
        return fail(URNN_EWORKSPACE, "urnn_weight_gradient_f32: workspace %zu < %zu bytes", workspace_bytes, urnn_weight_gradient_workspace_bytes(B, N, K, H, W));
    const float *seg[3] = {seg0, seg1 ? seg1 : seg0, seg2 ? seg2 : seg0};
    const int segC[3] = {C0, C1, C2};
    CHECK_HIP(urnn_train_wgrad(dy, seg, segC, B, N, K, (int)P, reinterpret_cast<float *>(workspace), dW, db, accumulate, (hipStream_t)stream),
              "weight gradient");
    return URNN_OK;
}

// ---- head ------------------------------------------------------------------------------------------------------------
struct HeadWs {
    int *status;
    float *u1, *u2, *partial, *stats;
    size_t bytes;
};

static HeadWs carve_head(void *base, int B, int C, long P)
{
    size_t off = 0;
    auto take = [&](size_t nfloats) {
        float *p = base ? reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) : nullptr;
        off += align_up(nfloats * sizeof(float), 256);
        return p;
    };
    HeadWs w;
    w.status = reinterpret_cast<int *>(take(URNN_STATUS_BYTES / sizeof(float)));
    w.u1 = take((size_t)B * C * P);
    w.u2 = take((size_t)B * C * P);
    w.partial = take((size_t)5 * B * urnn_head_nblk((int)P) * 2);
    w.stats = take((size_t)5 * B * 2);
    w.bytes = off;
    return w;
}

extern "C" size_t urnn_head_workspace_bytes(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < 1 || W < 1) return 0;
    return carve_head(nullptr, B, C, (long)H * W).bytes;
}

static int head_impl(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                     const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                     float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                     int H, int W, float cls_thred, float eps, float slope, int phase_mask, long global_pixels, void *stream,
                     const float *partial0 = nullptr, int coop = 0, int *bump = nullptr)
{
    if (!feat || !conv_w || !ln_w || !ln_b || !cls_w || !cls_b || !reg_w || !reg_b || !out_masked || !out_cls || !workspace)
        return fail(URNN_ENULL, "urnn_head_f32: NULL argument");
    if (C != 16) return fail(URNN_EINVAL, "urnn_head_f32: head width C=%d unsupported (kernels are built for 16)", C);
    if (B < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_head_f32: bad dims");
    if (!aligned16(feat) || !aligned16(ln_w) || !aligned16(ln_b) || !aligned16(workspace))
        return fail(URNN_EALIGN, "urnn_head_f32: pointers must be 16-byte aligned");
    const long P = (long)H * W;
    const HeadWs ws = carve_head(workspace, B, C, P);
    if (workspace_bytes < ws.bytes) return fail(URNN_EWORKSPACE, "urnn_head_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    HeadParams p = {};
    p.feat = feat;
    p.conv_w = conv_w;
    p.ln_w = ln_w;
    p.ln_b = ln_b;
    p.cls_w = cls_w;
    p.cls_b = cls_b;
    p.reg_w = reg_w;
    p.reg_b = reg_b;
    p.out_masked = out_masked;
    p.out_cls = out_cls;
    p.out_raw = out_raw;
    p.frame_index = frame_index;
    p.u1 = ws.u1;
    p.u2 = ws.u2;
    p.partial = ws.partial;
    p.stats = ws.stats;
    p.status = ws.status;
    p.B = B;
    p.C = C;
    p.P = (int)P;
    p.nblk = urnn_head_nblk((int)P);
    p.cls_thred = cls_thred;
    p.eps = eps;
    p.slope = slope;
    p.Pglobal = global_pixels;
    p.partial0 = partial0;
    p.nblk0 = (int)((P + 127) / 128);
    p.bpix0 = 128;
    p.bump = bump;
    if (bump && (bump == frame_index || phase_mask != URNN_HEAD_ALL))
        return fail(URNN_EINVAL, "urnn_head_rollout_f32: frame_next must not be the head's own frame_index");
    if (coop) {
        if (urnn_head_coop_blocks(B, (int)P) <= 0) return fail(URNN_EINVAL, "urnn_head_coop_f32: the launch's blocks cannot all be resident on this device (urnn_head_coop_blocks_f32 returned 0)");
        CHECK_HIP(urnn_launch_head_coop(p, reinterpret_cast<unsigned *>(ws.status) + 256, (hipStream_t)stream), "head (one cooperative launch)");
        return URNN_OK;
    }
    CHECK_HIP(urnn_launch_head(p, phase_mask, (hipStream_t)stream), "head");
    return URNN_OK;
}

extern "C" int urnn_head_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                             const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                             float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                             int H, int W, float cls_thred, float eps, float slope, void *stream)
{
    return head_impl(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw, frame_index, workspace,
                     workspace_bytes, B, C, H, W, cls_thred, eps, slope, URNN_HEAD_ALL, 0, stream);
}

// The head of a small plane as ONE cooperative launch (urnn_elem.hip head_coop_kernel): the four passes with grid barriers between them
extern "C" int urnn_head_coop_blocks_f32(int B, int H, int W)
{
    if (B < 1 || H < 1 || W < 1) return 0;
    return urnn_head_coop_blocks(B, H * W);
}

extern "C" int urnn_head_coop_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                                  const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                                  float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                                  int H, int W, float cls_thred, float eps, float slope, void *stream)
{
    return head_impl(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw, frame_index, workspace,
                     workspace_bytes, B, C, H, W, cls_thred, eps, slope, URNN_HEAD_ALL, 0, stream, nullptr, 1);
}

// The head behind urnn_gru_cell_tail_f32(..., head_conv_w, head_partial0): its first pass (the statistics of the stem's LayerNorm) was
// taken where feat was produced; three passes remain.
extern "C" int urnn_head_after_tail_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                                        const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                                        float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                                        int H, int W, float cls_thred, float eps, float slope, const float *head_partial0, void *stream)
{
    if (!head_partial0) return fail(URNN_ENULL, "urnn_head_after_tail_f32: NULL head_partial0");
    return head_impl(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw, frame_index, workspace,
                     workspace_bytes, B, C, H, W, cls_thred, eps, slope, URNN_HEAD_ALL, 0, stream, head_partial0);
}

// The head inside a captured frame loop: it writes frame *frame_index, and its first launch stores *frame_index + 1 to *frame_next --
// the word the NEXT head reads.  Two words used alternately (frame parity) are a device frame counter that needs no kernel of its own;
// a launch never writes the word it reads (other blocks may not have read it yet), hence frame_next != frame_index.
// coop / head_partial0 select urnn_head_coop_f32 / urnn_head_after_tail_f32 (at most one of them).
extern "C" int urnn_head_rollout_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                                     const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                                     float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                                     int H, int W, float cls_thred, float eps, float slope, int coop, const float *head_partial0,
                                     int *frame_next, void *stream)
{
    if (coop && head_partial0) return fail(URNN_EINVAL, "urnn_head_rollout_f32: coop and head_partial0 exclude each other");
    if (frame_next && !frame_index) return fail(URNN_ENULL, "urnn_head_rollout_f32: frame_next without frame_index");
    return head_impl(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw, frame_index, workspace,
                     workspace_bytes, B, C, H, W, cls_thred, eps, slope, URNN_HEAD_ALL, 0, stream, head_partial0, coop ? 1 : 0, frame_next);
}

extern "C" int urnn_head_strip_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *cls_w,
                                   const float *cls_b, const float *reg_w, const float *reg_b, float *out_masked, float *out_cls,
                                   float *out_raw, const int *frame_index, void *workspace, size_t workspace_bytes, int B, int C,
                                   int H, int W, float cls_thred, float eps, float slope, int phase_mask, long global_pixels,
                                   void *stream)
{
    if (global_pixels < (long)H * W) return fail(URNN_EINVAL, "urnn_head_strip_f32: global_pixels %ld < the strip's %ld", global_pixels, (long)H * W);
    for (int l = 0; l < 3; ++l)
        if ((phase_mask & (URNN_HEAD_K1 << (2 * l))) && (phase_mask & (URNN_HEAD_F1 << (2 * l))))
            return fail(URNN_EINVAL, "urnn_head_strip_f32: the LayerNorm statistics of level %d must be exchanged before their finalize", l);
    return head_impl(feat, conv_w, ln_w, ln_b, cls_w, cls_b, reg_w, reg_b, out_masked, out_cls, out_raw, frame_index, workspace,
                     workspace_bytes, B, C, H, W, cls_thred, eps, slope, phase_mask, global_pixels, stream);
}

// LayerNorm statistics of level 0 (stems: 1 norm), 1 (cls_convs.0 + reg_convs.0), 2 (cls_convs.1 + reg_convs.1):
// sums[norms of the level][B][2] doubles; directions as in urnn_gru_cell_strip_stats_f32
extern "C" int urnn_head_strip_stats_f32(void *workspace, size_t workspace_bytes, int B, int C, int H, int W, int level, int direction,
                                         double *sums, void *stream)
{
    if (!workspace || !sums) return fail(URNN_ENULL, "urnn_head_strip_stats_f32: NULL argument");
    if (B < 1 || C != 16 || H < 1 || W < 1 || level < 0 || level > 2) return fail(URNN_EINVAL, "urnn_head_strip_stats_f32: bad arguments");
    const long P = (long)H * W;
    const HeadWs ws = carve_head(workspace, B, C, P);
    if (workspace_bytes < ws.bytes) return fail(URNN_EWORKSPACE, "urnn_head_strip_stats_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    const int nblk = urnn_head_nblk((int)P), used = urnn_head_nblk_used((int)P);
    const int first = level == 0 ? 0 : level, n = level == 0 ? 1 : 2;      // norm indices first, first + 2
    for (int i = 0; i < n; ++i) {
        float *part = ws.partial + (size_t)(first + 2 * i) * B * nblk * 2;
        if (direction == 0)
            CHECK_HIP(urnn_launch_stats_reduce(part, B, nblk, used, urnn_head_block_pix((int)P), (int)P, 16, sums + (size_t)i * B * 2, (hipStream_t)stream),
                      "strip stats");
        else CHECK_HIP(urnn_launch_stats_scatter(sums + (size_t)i * B * 2, B, nblk, part, (hipStream_t)stream), "strip stats");
    }
    return URNN_OK;
}

// ---- head backward (training building block) -----------------------------------------------------------------------------
struct HeadBwdWs {
    float *save, *ds, *draw, *partial, *coef, *wpart, *wt, *pkt;
    size_t bytes;
};

static HeadBwdWs carve_head_bwd(void *base, int B, long P)
{
    size_t off = 0;
    auto takef = [&](size_t nfloats) {
        float *p = base ? reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) : nullptr;
        off += align_up(nfloats * sizeof(float), 256);
        return p;
    };
    HeadBwdWs w;
    w.save = takef((size_t)6 * B * 16 * P);
    w.ds = takef((size_t)B * 16 * P);
    w.draw = takef((size_t)B * P);
    w.partial = takef((size_t)B * 16 * urnn_train_head_ln_chunks((int)P) * 2);      // head_ln_bwd_a_kernel: one pair per (sample, channel, 1024-pixel chunk)
    w.coef = takef((size_t)B * 2);
    {
        const size_t a = urnn_train_wgrad_partial_floats(B, 16, 16, (int)P), b = urnn_train_wgrad_partial_floats(B, 1, 16, (int)P);
        w.wpart = takef(a > b ? a : b);
    }
    w.wt = takef(256);
    w.pkt = takef(urnn_packed_conv_floats(16, 16));
    w.bytes = off;
    return w;
}

extern "C" size_t urnn_head_backward_workspace_bytes(int B, int H, int W)
{
    if (B < 1 || H < 1 || W < 1) return 0;
    return carve_head_bwd(nullptr, B, (long)H * W).bytes;
}

extern "C" int urnn_head_backward_f32(const float *feat, const float *conv_w, const float *ln_w, const float *ln_b, const float *reg_w,
                                      const void *fwd_workspace, const float *out_raw, const float *out_cls, const float *dout,
                                      float *dfeat, float *dconv_w, float *dln_w, float *dln_b, float *dreg_w, float *dreg_b,
                                      void *workspace, size_t workspace_bytes, int B, int C, int H, int W, float cls_thred, float slope,
                                      int accumulate, void *stream)
{
    if (!feat || !conv_w || !ln_w || !ln_b || !reg_w || !fwd_workspace || !out_raw || !out_cls || !dout || !dfeat || !dconv_w || !dln_w ||
        !dln_b || !dreg_w || !dreg_b || !workspace)
        return fail(URNN_ENULL, "urnn_head_backward_f32: NULL argument");
    if (C != 16) return fail(URNN_EINVAL, "urnn_head_backward_f32: head width C=%d unsupported (kernels are built for 16)", C);
    if (B < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_head_backward_f32: bad dims");
    const long P = (long)H * W;
    const HeadWs fw = carve_head(const_cast<void *>(fwd_workspace), B, C, P);
    const HeadBwdWs ws = carve_head_bwd(workspace, B, P);
    if (workspace_bytes < ws.bytes)
        return fail(URNN_EWORKSPACE, "urnn_head_backward_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    hipStream_t st = (hipStream_t)stream;
    const int Pi = (int)P;
    const size_t CP = (size_t)16 * P, plane = (size_t)B * CP;
    CHECK_HIP(urnn_train_head_save(feat, conv_w, ln_w, ln_b, fw.stats, B, Pi, ws.save, st), "head: recompute the regression branch");
    CHECK_HIP(urnn_train_head_pred_bwd(dout, out_cls, out_raw, reg_w, cls_thred, slope, B, Pi, ws.draw, ws.ds, st), "head: prediction layer");
    {   // d(reg_preds): w_r[c] += sum draw * q2[c], b_r += sum draw
        const float *seg[3] = {ws.save + 5 * plane, nullptr, nullptr};
        const int segC[3] = {16, 0, 0};
        CHECK_HIP(urnn_train_wgrad(ws.draw, seg, segC, B, 1, 16, Pi, ws.wpart, dreg_w, dreg_b, accumulate, st), "head: prediction weights");
    }
    const int blk[3] = {4, 3, 0};            // reg_convs[1], reg_convs[0], stems
    const int uslot[3] = {2, 1, 0};          // pre-norm activations v2, v1, u0
    // The gradient ping-pongs between ws.ds and dfeat (a GEMM's output may not alias its input): ds -> dfeat -> ds -> dfeat,
    // so the third layer's input gradient lands in dfeat, where the caller wants it.
    float *cur = ws.ds, *other = dfeat;
    for (int l = 0; l < 3; ++l) {
        const int k = blk[l];
        CHECK_HIP(urnn_train_head_ln_bwd(cur, ws.save + uslot[l] * plane, ln_w + k * CP, ln_b + k * CP, fw.stats + (size_t)k * B * 2, B, Pi,
                                         dln_w + k * CP, dln_b + k * CP, accumulate, ws.partial, ws.coef, st), "head: LayerNorm backward");
        const float *in = l == 2 ? feat : ws.save + (size_t)(3 + (1 - l)) * plane;      // layer input: q1, t, feat
        const float *seg[3] = {in, nullptr, nullptr};
        const int segC[3] = {16, 0, 0};
        CHECK_HIP(urnn_train_wgrad(cur, seg, segC, B, 16, 16, Pi, ws.wpart, dconv_w + k * 256, nullptr, accumulate, st), "head: conv weights");
        int rc = pack_transposed(conv_w + k * 256, ws.wt, ws.pkt, 16, 16, st, "head: input-gradient weights");
        if (rc) return rc;
        rc = dx_gemm(cur, ws.pkt, other, B, 16, 16, H, W, st);
        if (rc) return rc;
        float *t = cur;
        cur = other;
        other = t;
    }
    if (!accumulate) {   // the classification branch receives no gradient
        for (int k = 1; k <= 2; ++k) {
            CHECK_HIP(urnn_train_zero(dconv_w + k * 256, 256, st), "head: zero cls grads");
            CHECK_HIP(urnn_train_zero(dln_w + k * CP, CP, st), "head: zero cls grads");
            CHECK_HIP(urnn_train_zero(dln_b + k * CP, CP, st), "head: zero cls grads");
        }
    }
    return URNN_OK;
}

// ---- training loss -----------------------------------------------------------------------------------------------------
extern "C" size_t urnn_loss_workspace_bytes(long n)
{
    if (n < 1) return 0;
    return align_up((size_t)urnn_train_loss_nblk(n) * 5 * sizeof(float), 256) + 256;
}

extern "C" int urnn_loss_f32(const float *reg, const float *target, float cls_thred, float *components, float *dreg, void *workspace,
                             size_t workspace_bytes, long n, void *stream)
{
    if (!reg || !target || !components || !workspace) return fail(URNN_ENULL, "urnn_loss_f32: NULL argument");
    if (n < 1) return fail(URNN_EINVAL, "urnn_loss_f32: n=%ld", n);
    if (workspace_bytes < urnn_loss_workspace_bytes(n)) return fail(URNN_EWORKSPACE, "urnn_loss_f32: workspace too small");
    float *partial = reinterpret_cast<float *>(workspace);
    float *scales = reinterpret_cast<float *>(reinterpret_cast<char *>(workspace) + align_up((size_t)urnn_train_loss_nblk(n) * 5 * sizeof(float), 256));
    CHECK_HIP(urnn_train_loss(reg, target, cls_thred, n, partial, scales, components, dreg, (hipStream_t)stream), "loss");
    return URNN_OK;
}

// ---- optimizer -------------------------------------------------------------------------------------------------------
extern "C" size_t urnn_adam_workspace_bytes(long n)
{
    if (n < 1) return 0;
    return align_up((size_t)urnn_train_loss_nblk(n) * sizeof(float), 256);
}

extern "C" int urnn_adam_step_f32(float *params, const float *grads, float *exp_avg, float *exp_avg_sq, long n, float lr, float beta1,
                                  float beta2, float eps, int step, const int *step_dev, float max_grad_norm, float *clip_out,
                                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (!params || !grads || !exp_avg || !exp_avg_sq || !clip_out || !workspace) return fail(URNN_ENULL, "urnn_adam_step_f32: NULL argument");
    if (n < 1 || (!step_dev && step < 1)) return fail(URNN_EINVAL, "urnn_adam_step_f32: n=%ld step=%d", n, step);
    if (workspace_bytes < urnn_adam_workspace_bytes(n)) return fail(URNN_EWORKSPACE, "urnn_adam_step_f32: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    CHECK_HIP(urnn_train_clip_coef(grads, n, max_grad_norm, reinterpret_cast<float *>(workspace), clip_out, st), "gradient norm");
    CHECK_HIP(urnn_train_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, step_dev, clip_out, st), "adam");
    return URNN_OK;
}

// ---- input assembly --------------------------------------------------------------------------------------------------
static int preprocess_impl(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                           const float *manhole, float dem_min, float dem_max, float *out, int t, const int *t_dev,
                           int B, int T, int nums, int H, int W, int spatial, float rain_max, float cumsum_max,
                           void *stream, int *bump)
{
    if (!rain || !cumsum || !dem || !imperv || !manhole || !out) return fail(URNN_ENULL, "urnn_preprocess_f32: NULL argument");
    if (B < 1 || T < 1 || nums < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_preprocess_f32: bad dims");
    if (!t_dev && t < 0) return fail(URNN_EINVAL, "urnn_preprocess_f32: t=%d", t);
    CHECK_HIP(urnn_launch_preprocess(rain, cumsum, dem, imperv, manhole, dem_min, dem_max, out, t, t_dev, B, T, nums, H * W,
                                     spatial ? 1 : 0, rain_max, cumsum_max, (hipStream_t)stream, bump),
              "preprocess");
    return URNN_OK;
}

extern "C" int urnn_preprocess_f32(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                                   const float *manhole, float dem_min, float dem_max, float *out, int t, const int *t_dev,
                                   int B, int T, int nums, int H, int W, int spatial, float rain_max, float cumsum_max,
                                   void *stream)
{
    return preprocess_impl(rain, cumsum, dem, imperv, manhole, dem_min, dem_max, out, t, t_dev, B, T, nums, H, W, spatial, rain_max,
                           cumsum_max, stream, nullptr);
}

// Frame-loop forms of the two input-assembly entries (see urnn_head_rollout_f32): the frame comes from *t_dev and the launch stores
// *t_dev + 1 to *t_next, the word the next frame's input assembly reads (never t_dev itself).
extern "C" int urnn_preprocess_rollout_f32(const float *rain, const float *cumsum, const float *dem, const float *imperv,
                                           const float *manhole, float dem_min, float dem_max, float *out, const int *t_dev,
                                           int *t_next, int B, int T, int nums, int H, int W, int spatial, float rain_max,
                                           float cumsum_max, void *stream)
{
    if (!t_dev) return fail(URNN_ENULL, "urnn_preprocess_rollout_f32: NULL t_dev");
    if (t_next == t_dev) return fail(URNN_EINVAL, "urnn_preprocess_rollout_f32: t_next must not be t_dev");
    return preprocess_impl(rain, cumsum, dem, imperv, manhole, dem_min, dem_max, out, 0, t_dev, B, T, nums, H, W, spatial, rain_max,
                           cumsum_max, stream, t_next);
}

extern "C" int urnn_stage1_static_f32(const float *dem, const float *imperv, const float *manhole, float dem_min, float dem_max,
                                      const float *weight, float *S, int B, int nums, int Cout, int H, int W, void *stream)
{
    if (!dem || !imperv || !manhole || !weight || !S) return fail(URNN_ENULL, "urnn_stage1_static_f32: NULL argument");
    if (B < 1 || nums < 1 || Cout < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_stage1_static_f32: bad dims");
    CHECK_HIP(urnn_launch_stage1_static(dem, imperv, manhole, dem_min, dem_max, weight, S, B, nums, Cout, H * W, (hipStream_t)stream),
              "stage1_static");
    return URNN_OK;
}

static int stage1_scalar_impl(const float *S, const float *rain, const float *cumsum, const float *weight,
                              const float *bias, float *out, int t, const int *t_dev, int B, int T, int nums, int Cout,
                              int H, int W, float rain_max, float cumsum_max, float slope, void *stream, int *bump)
{
    if (!S || !rain || !cumsum || !weight || !bias || !out) return fail(URNN_ENULL, "urnn_stage1_scalar_rain_f32: NULL argument");
    if (B < 1 || T < 1 || nums < 1 || Cout < 1 || H < 1 || W < 1) return fail(URNN_EINVAL, "urnn_stage1_scalar_rain_f32: bad dims");
    if (!aligned16(S) || !aligned16(out)) return fail(URNN_EALIGN, "urnn_stage1_scalar_rain_f32: pointers must be 16-byte aligned");
    CHECK_HIP(urnn_launch_stage1_scalar(S, rain, cumsum, weight, bias, out, t, t_dev, B, T, nums, Cout, H * W, rain_max, cumsum_max,
                                        slope, (hipStream_t)stream, bump),
              "stage1_scalar_rain");
    return URNN_OK;
}

extern "C" int urnn_stage1_scalar_rain_f32(const float *S, const float *rain, const float *cumsum, const float *weight,
                                           const float *bias, float *out, int t, const int *t_dev, int B, int T, int nums, int Cout,
                                           int H, int W, float rain_max, float cumsum_max, float slope, void *stream)
{
    return stage1_scalar_impl(S, rain, cumsum, weight, bias, out, t, t_dev, B, T, nums, Cout, H, W, rain_max, cumsum_max, slope, stream, nullptr);
}

extern "C" int urnn_stage1_scalar_rain_rollout_f32(const float *S, const float *rain, const float *cumsum, const float *weight,
                                                   const float *bias, float *out, const int *t_dev, int *t_next, int B, int T,
                                                   int nums, int Cout, int H, int W, float rain_max, float cumsum_max, float slope,
                                                   void *stream)
{
    if (!t_dev) return fail(URNN_ENULL, "urnn_stage1_scalar_rain_rollout_f32: NULL t_dev");
    if (t_next == t_dev) return fail(URNN_EINVAL, "urnn_stage1_scalar_rain_rollout_f32: t_next must not be t_dev");
    return stage1_scalar_impl(S, rain, cumsum, weight, bias, out, 0, t_dev, B, T, nums, Cout, H, W, rain_max, cumsum_max, slope, stream, t_next);
}

extern "C" int urnn_advance_counter(int *counter, int delta, void *stream)
{
    if (!counter) return fail(URNN_ENULL, "urnn_advance_counter: NULL counter");
    CHECK_HIP(urnn_launch_advance(counter, delta, (hipStream_t)stream), "advance_counter");
    return URNN_OK;
}

// ---- one whole timestep behind one call ---------------------------------------------------------------------------------------
// ED.forward (model.py:65-121): encoder (stage conv -> cell) x 3, decoder (cell -> transposed conv) x 2 + cell + conv, head -- the
// launches a Python host makes through the per-module entries, in their order, on ONE stream: enqueue-only like them, so the caller can
// capture the call in a hipGraph and replay it per frame (what RolloutEngine(overlap=False) does with the same launches).  An INFERENCE
// step: the cells take URNN_PHASE_FUSED_R | URNN_PHASE_COOP and the head its cooperative form wherever the shapes qualify (one stream:
// a cooperative launch's blocks are all resident once its predecessor drains), so the workspaces' raw planes are undefined afterwards.
struct StepWs {
    float *a1, *a2, *a3, *u3, *u2, *feat, *part0;
    void *cell, *head;
    size_t cell_bytes, head_bytes, bytes;
};

static StepWs carve_step(void *base, const urnn_net_f32 *n, int B, int H, int W)
{
    size_t off = 0;
    auto take = [&](size_t nbytes) {
        void *p = base ? reinterpret_cast<char *>(base) + off : nullptr;
        off += align_up(nbytes, 256);
        return p;
    };
    const int H2 = H / 2, W2 = W / 2, H4 = H2 / 2, W4 = W2 / 2;
    const size_t P1 = (size_t)H * W, P2 = (size_t)H2 * W2, P4 = (size_t)H4 * W4;
    StepWs w;
    w.cell_bytes = 0;
    const int F[6] = {n->enc_features[0], n->enc_features[1], n->enc_features[2], n->dec_features[0], n->dec_features[1], n->dec_features[2]};
    const int hh[6] = {H, H2, H4, H4, H2, H}, ww[6] = {W, W2, W4, W4, W2, W};
    for (int k = 0; k < 6; ++k) {
        const size_t b = urnn_gru_cell_workspace_bytes(B, F[k], hh[k], ww[k]);
        w.cell_bytes = b > w.cell_bytes ? b : w.cell_bytes;
    }
    w.head_bytes = urnn_head_workspace_bytes(B, n->feat_channels, H, W);
    w.cell = take(w.cell_bytes);           // (first: its status words are the call's status words)
    w.head = take(w.head_bytes);
    w.a1 = reinterpret_cast<float *>(take((size_t)B * n->enc_stage_out[0] * P1 * 4));
    w.a2 = reinterpret_cast<float *>(take((size_t)B * n->enc_stage_out[1] * P2 * 4));
    w.a3 = reinterpret_cast<float *>(take((size_t)B * n->enc_stage_out[2] * P4 * 4));
    w.u3 = reinterpret_cast<float *>(take((size_t)B * n->dec_stage_out[0] * P2 * 4));
    w.u2 = reinterpret_cast<float *>(take((size_t)B * n->dec_stage_out[1] * P1 * 4));
    w.feat = reinterpret_cast<float *>(take((size_t)B * n->feat_channels * P1 * 4));
    w.part0 = reinterpret_cast<float *>(take(urnn_head_tail_partial_floats(B, H, W) * 4));
    w.bytes = off;
    return w;
}

static int step_net_ok(const urnn_net_f32 *n)
{
    if (!n) return 0;
    for (int k = 0; k < 3; ++k)
        if (!n->enc_stage[k] || !n->enc_cell[k] || !n->dec_cell[k] || !n->enc_gn1_w[k] || !n->enc_gn1_b[k] || !n->enc_gn2_w[k] || !n->enc_gn2_b[k] ||
            !n->dec_gn1_w[k] || !n->dec_gn1_b[k] || !n->dec_gn2_w[k] || !n->dec_gn2_b[k] || n->enc_stage_out[k] < 1 || n->enc_features[k] < 32 || n->dec_features[k] < 32)
            return 0;
    return n->dec_stage[0] && n->dec_stage[1] && n->dec_stage[2] && n->head_conv_w && n->head_ln_w && n->head_ln_b && n->cls_w && n->cls_b && n->reg_w &&
           n->reg_b && n->in_channels >= 1 && n->dec_stage_out[0] >= 1 && n->dec_stage_out[1] >= 1 && n->feat_channels == 16;
}

extern "C" size_t urnn_step_workspace_bytes(const urnn_net_f32 *net, int B, int H, int W)
{
    if (!step_net_ok(net) || B < 1 || H < 4 || W < 4) return 0;
    return carve_step(nullptr, net, B, H, W).bytes;
}

extern "C" int urnn_step_f32(const urnn_net_f32 *net, const float *x_t, float *const states[6], float *out_masked, float *out_cls, float *out_raw,
                             const int *frame_index, void *workspace, size_t workspace_bytes, int B, int H, int W, float cls_thred,
                             float eps, float slope, void *stream)
{
    if (!step_net_ok(net)) return fail(URNN_EINVAL, "urnn_step_f32: incomplete network description (NULL slab or bad channel count)");
    if (!x_t || !states || !out_masked || !out_cls || !workspace) return fail(URNN_ENULL, "urnn_step_f32: NULL argument");
    for (int k = 0; k < 6; ++k)
        if (!states[k]) return fail(URNN_ENULL, "urnn_step_f32: NULL state %d", k);
    if (B < 1 || H < 4 || W < 4 || (H & 3) || (W & 3)) return fail(URNN_EINVAL, "urnn_step_f32: H = %d, W = %d must be multiples of four (two 2x2 poolings, two 2x2 transposed convolutions)", H, W);
    const StepWs ws = carve_step(workspace, net, B, H, W);
    if (workspace_bytes < ws.bytes) return fail(URNN_EWORKSPACE, "urnn_step_f32: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    const int H2 = H / 2, W2 = W / 2, H4 = H / 4, W4 = W / 4;
    float *e1 = states[0], *e2 = states[1], *e3 = states[2], *d1 = states[3], *d2 = states[4], *d3 = states[5];
    const int *so = net->enc_stage_out, *ef = net->enc_features, *df = net->dec_features, *uo = net->dec_stage_out;
    int rc;
    const int CELL_MASK = URNN_PHASE_ALL | URNN_PHASE_FUSED_R | URNN_PHASE_COOP;
#define URNN_STEP(call) do { rc = (call); if (rc) return rc; } while (0)
    // encoder (encoder.py:140-185): stage conv (+ pool) -> ConvGRU cell, three scales; states updated in place
    URNN_STEP(urnn_stage_conv_f32(x_t, net->enc_stage[0], ws.a1, B, net->in_channels, so[0], H, W, 0, slope, stream));
    URNN_STEP(urnn_gru_cell_phases_f32(ws.a1, nullptr, e1, net->enc_cell[0], net->enc_gn1_w[0], net->enc_gn1_b[0], net->enc_gn2_w[0], net->enc_gn2_b[0], e1,
                                ws.cell, ws.cell_bytes, B, so[0], ef[0], H, W, eps, CELL_MASK, stream));
    URNN_STEP(urnn_stage_conv_f32(e1, net->enc_stage[1], ws.a2, B, ef[0], so[1], H, W, 1, slope, stream));
    URNN_STEP(urnn_gru_cell_phases_f32(ws.a2, nullptr, e2, net->enc_cell[1], net->enc_gn1_w[1], net->enc_gn1_b[1], net->enc_gn2_w[1], net->enc_gn2_b[1], e2,
                                ws.cell, ws.cell_bytes, B, so[1], ef[1], H2, W2, eps, CELL_MASK, stream));
    URNN_STEP(urnn_stage_conv_f32(e2, net->enc_stage[2], ws.a3, B, ef[1], so[2], H2, W2, 1, slope, stream));
    URNN_STEP(urnn_gru_cell_phases_f32(ws.a3, nullptr, e3, net->enc_cell[2], net->enc_gn1_w[2], net->enc_gn1_b[2], net->enc_gn2_w[2], net->enc_gn2_b[2], e3,
                                ws.cell, ws.cell_bytes, B, so[2], ef[2], H4, W4, eps, CELL_MASK, stream));
    // decoder (decoder.py:130-203): Skip-ConvGRU cell -> transposed conv, deepest scale first (its x is zero: ConvRNN.py:143-146)
    URNN_STEP(urnn_gru_cell_phases_f32(nullptr, e3, d1, net->dec_cell[0], net->dec_gn1_w[0], net->dec_gn1_b[0], net->dec_gn2_w[0], net->dec_gn2_b[0], d1,
                                ws.cell, ws.cell_bytes, B, net->dec_zero_input_channels, df[0], H4, W4, eps, CELL_MASK, stream));
    URNN_STEP(urnn_deconv2x2_f32(d1, net->dec_stage[0], ws.u3, B, df[0], uo[0], H4, W4, slope, stream));
    URNN_STEP(urnn_gru_cell_phases_f32(ws.u3, e2, d2, net->dec_cell[1], net->dec_gn1_w[1], net->dec_gn1_b[1], net->dec_gn2_w[1], net->dec_gn2_b[1], d2,
                                ws.cell, ws.cell_bytes, B, uo[0], df[1], H2, W2, eps, CELL_MASK, stream));
    URNN_STEP(urnn_deconv2x2_f32(d2, net->dec_stage[1], ws.u2, B, df[1], uo[1], H2, W2, slope, stream));
    URNN_STEP(urnn_gru_cell_phases_f32(ws.u2, e1, d3, net->dec_cell[2], net->dec_gn1_w[2], net->dec_gn1_b[2], net->dec_gn2_w[2], net->dec_gn2_b[2], d3,
                                ws.cell, ws.cell_bytes, B, uo[1], df[2], H, W, eps, CELL_MASK, stream));
    const bool stem = stage_conv_stem_ok(B, df[2], net->feat_channels, H, W);   // the head's first statistics in the last conv's epilogue
    if (stem) URNN_STEP(urnn_stage_conv_stem_f32(d3, net->dec_stage[2], ws.feat, B, df[2], net->feat_channels, H, W, slope, net->head_conv_w, ws.part0, stream));
    else URNN_STEP(urnn_stage_conv_f32(d3, net->dec_stage[2], ws.feat, B, df[2], net->feat_channels, H, W, 0, slope, stream));
    // head + wet / dry mask (flood_head.py:131-202)
    if (stem)
        URNN_STEP(urnn_head_after_tail_f32(ws.feat, net->head_conv_w, net->head_ln_w, net->head_ln_b, net->cls_w, net->cls_b, net->reg_w, net->reg_b,
                                           out_masked, out_cls, out_raw, frame_index, ws.head, ws.head_bytes, B, net->feat_channels, H, W, cls_thred, eps,
                                           slope, ws.part0, stream));
    else if (urnn_head_coop_blocks_f32(B, H, W) > 0)
        URNN_STEP(urnn_head_coop_f32(ws.feat, net->head_conv_w, net->head_ln_w, net->head_ln_b, net->cls_w, net->cls_b, net->reg_w, net->reg_b, out_masked,
                                     out_cls, out_raw, frame_index, ws.head, ws.head_bytes, B, net->feat_channels, H, W, cls_thred, eps, slope, stream));
    else
        URNN_STEP(urnn_head_f32(ws.feat, net->head_conv_w, net->head_ln_w, net->head_ln_b, net->cls_w, net->cls_b, net->reg_w, net->reg_b, out_masked,
                                out_cls, out_raw, frame_index, ws.head, ws.head_bytes, B, net->feat_channels, H, W, cls_thred, eps, slope, stream));
#undef URNN_STEP
    return URNN_OK;
}

// The status / barrier words of the step's cell and head scratch (include/urnn_hip.h "status words"): zero them once per workspace,
// read them at a synchronisation point of the caller's choice.
extern "C" int urnn_step_workspace_init(const urnn_net_f32 *net, void *workspace, size_t workspace_bytes, int B, int H, int W, int **cell_status,
                                        int **head_status, void *stream)
{
    if (!step_net_ok(net) || !workspace) return fail(URNN_ENULL, "urnn_step_workspace_init: NULL argument or incomplete network description");
    if (B < 1 || H < 4 || W < 4) return fail(URNN_EINVAL, "urnn_step_workspace_init: bad dims");
    const StepWs ws = carve_step(workspace, net, B, H, W);
    if (workspace_bytes < ws.bytes) return fail(URNN_EWORKSPACE, "urnn_step_workspace_init: workspace %zu < %zu bytes", workspace_bytes, ws.bytes);
    CHECK_HIP(urnn_train_zero(reinterpret_cast<float *>(ws.cell), URNN_STATUS_BYTES / sizeof(float), (hipStream_t)stream), "zero the cell status words");
    CHECK_HIP(urnn_train_zero(reinterpret_cast<float *>(ws.head), URNN_STATUS_BYTES / sizeof(float), (hipStream_t)stream), "zero the head status words");
    if (cell_status) *cell_status = reinterpret_cast<int *>(ws.cell);
    if (head_status) *head_status = reinterpret_cast<int *>(ws.head);
    return URNN_OK;
}
