// urnn_gemm_deconv.hip -- ConvTranspose 2x2 stride 2 as a 4*Cout-column GEMM with a scatter epilogue (conv_gemm_kernel, urnn_gemm.h)
#define URNN_TU urnn_gemm_deconv
#include "urnn_gemm.h"

// Deconv: two groups (output row parity), each 2 * ceil(Cout/32) n-blocks, PB = 2 pairs (1 strided for tiny/odd planes).
hipError_t urnn_launch_deconv(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    const int nbc = (p.Cout + 31) / 32;
    if (nbc < 1 || nbc > 3) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.totalTiles = B * p.tilesPerSample;
    if (PB == 2 && map == MAP_PAIR16 && nbc == 3 && quad_ok<6, 2, EPI_DECONV>(p)) return launch_conv<6, 2, MAP_QUAD16, EPI_DECONV>(p, st);
    if (PB == 2 && map == MAP_PAIR16) {
        if (nbc == 1) return launch_conv<2, 2, MAP_PAIR16, EPI_DECONV>(p, st);
        if (nbc == 2) return launch_conv<4, 2, MAP_PAIR16, EPI_DECONV>(p, st);
        return launch_conv<6, 2, MAP_PAIR16, EPI_DECONV>(p, st);
    }
    if (PB == 2 && map == MAP_PAIR) {
        if (nbc == 1) return launch_conv<2, 2, MAP_PAIR, EPI_DECONV>(p, st);
        if (nbc == 2) return launch_conv<4, 2, MAP_PAIR, EPI_DECONV>(p, st);
        return launch_conv<6, 2, MAP_PAIR, EPI_DECONV>(p, st);
    }
    if (PB != 1 || map != MAP_STRIDED) return hipErrorInvalidValue;
    if (nbc == 1) return launch_conv<2, 1, MAP_STRIDED, EPI_DECONV>(p, st);
    if (nbc == 2) return launch_conv<4, 1, MAP_STRIDED, EPI_DECONV>(p, st);
    return launch_conv<6, 1, MAP_STRIDED, EPI_DECONV>(p, st);
}

