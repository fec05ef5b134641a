// urnn_gemm.hip -- stage convolutions (flat and pooled 1x1 convs) on conv_gemm_kernel (urnn_gemm.h) + the process-wide matrix mode
#define URNN_TU urnn_gemm
#include "urnn_gemm.h"

// Process-wide arithmetic of the GEMMs (urnn_set_matrix_mode): 0 fp32-exact (bf16 x 6 split / fp32 MFMA), 1 bf16 compute.
std::atomic<int> g_matrix_mode{0};
extern "C" int urnn_set_matrix_mode(int mode)
{
    if (mode != URNN_MATRIX_FP32 && mode != URNN_MATRIX_BF16 && mode != URNN_MATRIX_FP32_MFMA && mode != URNN_MATRIX_FP32_CAND) return URNN_EINVAL;
    g_matrix_mode.store(mode, std::memory_order_relaxed);
    return URNN_OK;
}
extern "C" int urnn_get_matrix_mode(void) { return g_matrix_mode.load(std::memory_order_relaxed); }

int urnn_conv_nb(int Cout)
{
    // n-blocks per group: all of them up to three; else 3 or 2 when that divides the count; else 3 with the last group
    // padded by zero columns (7 blocks -> 3 groups instead of 7 groups of one block that would each re-read the input)
    const int nblk = (Cout + 31) / 32;
    // 96 output channels: three groups of one block.  A 3-block wave tile (192 accumulators with 128-pixel / pooled tiles) cannot
    // take the split k-loop; three 1-block groups can, and the second and third read of the input hit the XCD's L2 (pooled
    // 96 -> 96 conv at 250x250: 37 -> 22 us).  Development knob URNN_TUNE_CONV_NB3=3 restores one group.
    static const int nb3 = (int)urnn_tune("URNN_TUNE_CONV_NB3", 1);
    if (nblk == 3) return nb3 == 3 ? 3 : 1;
    // wider outputs (the backward pass's input-gradient GEMMs: 160 / 192 columns): groups of two blocks, the last one padded -- a
    // 3-block wave tile of 128 pixels (192 accumulators) would fall back to the fp32 MFMA k-loop (190 us per launch at 500x500)
    return nblk <= 3 ? nblk : 2;
}

int urnn_conv_ng(int Cout)
{
    const int nblk = (Cout + 31) / 32, NB = urnn_conv_nb(Cout);
    return (nblk + NB - 1) / NB;
}

// Flat 1x1 conv + LeakyReLU.  NB n-blocks per group chosen from Cout; NG groups cover all columns.
hipError_t urnn_launch_conv_flat(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int NB = urnn_conv_nb(p.Cout);
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.totalTiles = B * p.tilesPerSample;
    if (NB == 1) return launch_flat<1, EPI_LRELU>(p, PB, map, st);
    if (NB == 2) return launch_flat<2, EPI_LRELU>(p, PB, map, st);
    return launch_flat<3, EPI_LRELU>(p, PB, map, st);
}

hipError_t urnn_launch_conv_pool(ConvGemmParams p, int B, hipStream_t st)
{
    const int NB = urnn_conv_nb(p.Cout);
    p.tilesPerSample = (p.P2 + 31) / 32;
    p.totalTiles = B * p.tilesPerSample;
    if (NB == 1) return launch_conv<1, 4, MAP_POOL, EPI_POOL>(p, st);
    if (NB == 2) return launch_conv<2, 4, MAP_POOL, EPI_POOL>(p, st);
    return launch_conv<3, 4, MAP_POOL, EPI_POOL>(p, st);
}

