// Argument block, vector types and workgroup size shared by the cross-spectral kernels (csd_kernel.h, csd3m_kernel.h).
#pragma once

#include "spy_intrinsics.h"

namespace spycsd {

constexpr int CSD_THREADS = 512;   // 8 waves = 2 per SIMD: one wave's LDS waits hide behind the other's MFMAs

struct CsdArgs {
    const float2* spec;   // (nrows, F, C) complex64
    long long nrows;
    int F, C;
    float2* acc;          // (F, C, C) complex64
    int nt;               // channel tiles = ceil(C/32)
    int ntiles;           // nt*(nt+1)/2
    long long nitems;     // F*ntiles
    long long item_base;  // this launch covers items [item_base, item_end)
    long long item_end;
    int cpad;             // nt*32
    int kb;               // rows per LDS chunk (multiple of 4)
    // row split (tail re-cut): blockIdx.y = s works on rows [s*rows_per_split, (s+1)*rows_per_split);
    // split 0 adds into acc, split s > 0 stores its partial tile sums into part[s-1][f - part_f0]
    long long rows_per_split;   // 0 = no split
    float2* part;
    int part_f0, part_nf;
    int blocked;                // spec = (nrows, ceil(C/4), F, 4): channel quads contiguous in frequency
    int fast_per;               // FAST path: items per workgroup = (frequencies per 256-element LDS row) * ntiles
    int fast_nwgf;              // FAST == 3: workgroups per frequency (each owns <= fast_per of its ntiles tiles)
    // channel sub-ranges of the 3-multiplication kernel (more than 512 channels; 0 = the whole rows): rows are `ctot`
    // channels wide, the launch works on n0 channels from ch0 (and, for a rectangle, n1 channels from ch1)
    int ctot, ch0, n0, ch1, n1;
    // csd3m_kernel<256, 8> as the float32 stand-in of csdh_kernel (csdh_kernel.h): only frequencies f with only_flagged[f] != 0
    const int* only_flagged;
};

}  // namespace spycsd
