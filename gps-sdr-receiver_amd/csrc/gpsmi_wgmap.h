// Workgroup -> (block, channel group) of the code-phase correlation launch, as ONE function the
// host (grid size), the kernel (its unit) and a CPU test (gpsmi_trk_corr_wg_map) share.
//
// Batches: a block's rows are read once per channel group; the groups of a block must sit on ONE
// XCD so that the second and third reads hit that XCD's L2 (each of the 8 XCDs has its own;
// workgroup w of a launch goes to XCD w % 8).  Slot s = w / 8 of XCD x = w % 8 therefore serves
// block (s / ng) * 8 + x, group s % ng: consecutive slots of an XCD are the groups of one block.
// The grid is padded to a multiple of 8 blocks (workgroups beyond nblocks exit).
// Fewer than 8 blocks (the closed loop): the padding would be most of the launch; the grid is
// exactly nblocks * ng and maps linearly (a single block's groups then spread over the XCDs,
// which is what a latency-bound launch wants anyway).
// Round 3 inferred the mode from gridDim.x == nblocks * ng inside the kernel, which is also true
// for every batch whose block count is a multiple of 8: the 1024-block batch ran linear and
// fetched its centre rows three times from HBM (FETCH_SIZE x 2 = 411 MB instead of 139 MB).
#pragma once

#if defined(__HIPCC__)
#define GPSMI_HD __host__ __device__
#else
#define GPSMI_HD
#endif

namespace gpsmi {

struct CorrWg { int block, group; };

GPSMI_HD constexpr bool corr_linear(int nblocks) { return nblocks < 8; }

GPSMI_HD constexpr int corr_grid(int nblocks, int ng) {
    return corr_linear(nblocks) ? nblocks * ng : ((nblocks + 7) / 8) * 8 * ng;
}

// block >= nblocks: a padding workgroup (nothing to do)
GPSMI_HD constexpr CorrWg corr_wg_map(int wg, int nblocks, int ng) {
    if (corr_linear(nblocks)) return CorrWg{wg / ng, wg % ng};
    const int xcd = wg & 7, slot = wg >> 3;
    return CorrWg{(slot / ng) * 8 + xcd, slot % ng};
}

}  // namespace gpsmi
