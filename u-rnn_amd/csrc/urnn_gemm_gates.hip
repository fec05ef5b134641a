// urnn_gemm_gates.hip -- the ConvGRU gate GEMM (conv_gemm_kernel<..., EPI_GRU1>, urnn_gemm.h)
#define URNN_TU urnn_gemm_gates
#include "urnn_gemm.h"

// GRU gate GEMM.  Column grouping of the f16 slab: as many of the 2F columns per wave as the accumulators (128 registers at two
// waves per SIMD: 4 n-blocks x 64 pixels) and the LDS (slab + 8 rings of 64-pixel slots) allow -- F = 64: one group of all four
// blocks (z0 r0 z1 r1); F = 96: the z half and the r half (3 blocks each); F = 128: two groups of two pairs.  Development knob
// URNN_TUNE_GATE_ALLN=0 keeps the F/32 groups of (z_i | r_i).
GateGroups urnn_gate_groups(int F, int KT)
{
    static const int alln = (int)urnn_tune("URNN_TUNE_GATE_ALLN", 1);
    const int G = F / 32;
    auto fits = [&](int NB) { return (size_t)urnn_f16_slab_dwords(KT, NB) * 4 + 8 * 5 * 1024 + NB * 128 + 2048 <= LDS_PER_CU; };   // 8 rings of 4 (+1) 1-KB slots
    if (alln && KT % 8 == 0) {
        if (G == 2 && fits(4)) return GateGroups{4, 1, 0, 2};
        if (G == 4 && fits(4)) return GateGroups{4, 2, 0, 2};
        if (G == 3 && fits(3)) return GateGroups{3, 2, 1, 0};
    }
    return GateGroups{2, G, 0, 1};
}


// f16-eligible launch with a wide grouping -> the grouped kernel on 64-pixel tiles (pair maps need an even plane)
static bool gate_grouped(const ConvGemmParams &p)
{
    if (p.NBf == 2 || !p.wf16 || p.fDwords <= 0 || !p.biasf) return false;
    { const int mm_ = g_matrix_mode.load(std::memory_order_relaxed); if ((mm_ != URNN_MATRIX_FP32 && mm_ != URNN_MATRIX_FP32_CAND) || !tune_split() || !tune_f16()) return false; }
    if (p.KT % 8 != 0 || p.kpBegin % 8 != 0 || p.KT <= p.kpBegin) return false;
    return true;
}

int urnn_gate_plan(const ConvGemmParams &p, int B, int pb_legacy, int map_legacy, int *pb, int *map)
{
    (void)B;
    if (!gate_grouped(p)) { *pb = pb_legacy; *map = map_legacy; return 0; }
    if (pb_legacy >= 2) { *pb = 2; *map = p.P % 4 == 0 ? MAP_PAIR16 : (p.P % 2 == 0 ? MAP_PAIR : MAP_STRIDED); }
    else { *pb = 1; *map = MAP_STRIDED; }
    return 1;
}

hipError_t urnn_launch_gru1(ConvGemmParams p, int B, int PB, int map, hipStream_t st)
{
    if (p.NG < 1 || p.NG > 4) return hipErrorInvalidValue;
    p.tilesPerSample = (p.P + 32 * PB - 1) / (32 * PB);
    p.totalTiles = B * p.tilesPerSample;
    set_tile_means(p, 32 * PB);
    if (gate_grouped(p)) {
        if (PB > 2) return hipErrorInvalidValue;                  // the caller plans with urnn_gate_plan
        p.NG = p.NGf;
        if (p.NBf == 4) return launch_flat<4, EPI_GRU1>(p, PB, map, st);
        if (p.NBf == 3) return launch_flat<3, EPI_GRU1>(p, PB, map, st);
        return hipErrorInvalidValue;
    }
    if (p.NBf != 2) { p.wf16 = nullptr; p.fDwords = 0; p.biasf = nullptr; }     // the f16 slab is grouped for the other kernel
    return launch_flat<2, EPI_GRU1>(p, PB, map, st);
}

