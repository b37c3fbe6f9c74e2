// Tile geometry shared by the kernels of the two single-channel layers (stencil_c1.hip, c1_mfma.hip): a tile is TW
// consecutive pixels of one row of the half-resolution grid (LH, LW) = (HH/2, WW/2); its 16 taps per pixel come from
// 4 image rows x (2*TW+2) columns:  patch(ly, lx)[kh*4+kw] = img[2ly-1+kh][2lx-1+kw].
#pragma once
#include "nsg_common.h"

namespace {

constexpr int TW = 64;             // low-res pixels of one image row per tile
constexpr int PP = 2 * TW + 4;     // patch row pitch (2*TW + 2 used)

struct C1Geom {
    int B, LH, LW, HH, WW, C;
    int segs;                      // tiles per low-res row
    int ntiles;
    FastDiv div_segs, div_lh;
};
constexpr int PATCH_VALS = 4 * (2 * TW + 2);

__device__ __forceinline__ void tile_coords(const C1Geom &g, int tile, int &b, int &ly, int &ox0)
{
    const int row = nsg_div(tile, g.div_segs);
    ox0 = (tile - row * g.segs) * TW;
    b = nsg_div(row, g.div_lh);
    ly = row - b * g.LH;
}

inline C1Geom make_geom(int B, int LH, int LW, int HH, int WW, int C)
{
    C1Geom g;
    g.B = B; g.LH = LH; g.LW = LW; g.HH = HH; g.WW = WW; g.C = C;
    g.segs = (LW + TW - 1) / TW;
    g.ntiles = B * LH * g.segs;
    g.div_segs = nsg_fastdiv((uint32_t)g.segs);
    g.div_lh = nsg_fastdiv((uint32_t)LH);
    return g;
}

}  // namespace
