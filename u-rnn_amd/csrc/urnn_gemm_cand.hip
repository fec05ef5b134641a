// urnn_gemm_cand.hip -- the two-stream candidate GEMM (conv_gemm_kernel<..., EPI_CAND>, urnn_gemm.h)
#define URNN_TU urnn_gemm_cand
#include "urnn_gemm.h"

// Candidate GEMM: C = W2 . [x; e; sigmoid(GN(r)) * h] + b2, NG groups of NB n-blocks (urnn_cand_nb).
int urnn_cand_nb(int F)
{
    static const int forced = (int)urnn_tune("URNN_TUNE_CAND_NB", 0);   // development knob
    const int nblk = F / 32;
    if (forced > 0 && nblk % forced == 0) return forced;
    return nblk <= 3 ? nblk : 2;
}

hipError_t urnn_launch_cand(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int NB = urnn_cand_nb(p.F);
    if (p.F % 32 != 0 || (p.F / 32) % NB != 0) return hipErrorInvalidValue;
    p.B = B;
    p.NG = (p.F / 32) / NB;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.totalTiles = B * p.tilesPerSample;
    set_tile_means(p, 32 * PB);
    if (NB == 1) return launch_flat<1, EPI_CAND>(p, PB, map, st);
    if (NB == 2) return launch_flat<2, EPI_CAND>(p, PB, map, st);
    return launch_flat<3, EPI_CAND>(p, PB, map, st);
}

