// The tracking correlator, span form (default for CS = 2048, N_CYC = 32).
//
// Same mathematics as gpsmi_trk_stream_mfma.h -- prompt correlate-and-dump of a 32-ms
// block, y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed per
// code-period window (reference src/gpslib.py:1400-1420) -- re-cut so that
//   (1) every load instruction reads 512 consecutive bytes of a row (tiles of 32 rows x 64
//       positions; 256-byte segments top out at 5.2 TB/s, 512-byte ones reach 5.7-6.0,
//       tools/probe/tile_read.hip),
//   (2) a batch of 1024 blocks is two full rounds of workgroups (two workgroups of four
//       waves per CU, 18 KiB of LDS per wave) with no one-wave-per-SIMD tail, and the waves
//       of a workgroup never wait for each other (no barrier: each writes its own sums),
//   (3) the order of the float32 sums is defined by the DATA, not by the launch: a block
//       is cut into 32 SPANS of 64 positions; a span is summed position by position, the
//       spans of a quarter (512 positions) are added in order, then the four quarters.
//       Any assignment of spans to waves that follows this order gives the same bits:
//       the batch form gives a wave a quarter (eight spans), the single-block form gives
//       every span its own wave (32 waves on up to 32 CUs instead of 4 waves on one).
//
//   v_mfma_f32_16x16x4_f32, D[16 x 16] += A[16 x 4] B[4 x 16] for one PAIR of positions:
//     M = 16 code periods (two M tiles = the 32 rows of the block),
//     K = (position parity pi, re/im kappa) of the samples at positions 2 q + pi,
//     N = (channel, re/im) for 8 channels; two N tiles = up to 16 channels (12 are staged):
//       A[r][(pi, kappa)]      = x[r][2 q + pi].{re, im}
//       B[(pi, 0)][(c, re)] =  B_re,  B[(pi, 1)][(c, re)] = -B_im,
//       B[(pi, 0)][(c, im)] =  B_im,  B[(pi, 1)][(c, im)] =  B_re,
//       B_c(m) = replica_c[(m - d_c) mod 2048] * exp(-j theta_c(m)),  m = 2 q + pi.
//   The fp32 MFMA runs on the SIMD's own fp32 lanes: no other VALU instruction issues
//   meanwhile, so every VALU / LDS instruction between two MFMAs adds to the SIMD's time.
//   B is therefore built once per pair for all 32 rows: a lane holds one real component u
//   of the carrier phasor for its two channels (one per N tile) at positions m, m + 2 in a
//   packed register pair and advances both by four positions with the coupled recurrence
//   dl -= kappa u, u += dl (kappa = 4 sin^2(2 phi)), twelve packed instructions per four
//   pairs in one asm block (no wait states before the MFMAs that read B).  The exact phasor
//   is taken at the start of every quarter only (128 steps, as far as the recurrence stays
//   accurate); a wave that starts at a later span of the quarter runs the same recurrence
//   from the quarter start without the MFMAs, so it forms the same bits.  The replica
//   samples come from a table split by index parity (a lane's positions all have the same
//   parity), staged through LDS 32 positions at a time.
//
// Window q of the reference = positions m >= d of row q ("hi") plus m < d of row q+1
// ("lo").  Where the boundary d_c falls inside a wave's range the four lanes of channel c
// close the lo sum (lo = sum so far) and restart; a pair of positions that straddles an odd
// d_c is issued twice with B masked.  Only the group of four pairs that holds a boundary
// takes that path; everything else runs 16 MFMAs per group straight.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace gpsmi {

constexpr int kSpCh = 12;                      // channels per workgroup (staged through LDS)
constexpr int kSpTile = 64;                    // positions per tile = per span (unit of the summation order)
constexpr int kSpQuarter = 512;                // positions per quarter (eight spans)
constexpr int kSpRowDw = 2 * kSpTile + 2;      // dwords per tile row: lane = (row, k) reads are conflict-free
constexpr int kSpTileFloats = 32 * kSpRowDw;
constexpr int kSpWin = 32;                     // positions per replica window
constexpr int kSpCodePitch = kSpWin / 2 + 4;   // floats per (channel, parity) row of a window
constexpr int kSpCodeFloats = kSpCh * 2 * kSpCodePitch;
constexpr int kSpWaveFloats = kSpTileFloats + kSpCodeFloats;     // 4640 floats = 18,560 B per wave
constexpr int kSpInf = 1 << 20;


typedef float sp4 __attribute__((ext_vector_type(4)));
typedef float sp2 __attribute__((ext_vector_type(2)));

// This header is compiled with floating-point contraction OFF (see gpsmi_trk.hip): the batch
// form, the single-block form and the collect step must round identically, so every fused
// multiply-add below is written as one.
// (cos, -sin) of 2*pi*rev as phasor_rev (gpsmi_trk_stream.h), with a fixed operation order
__device__ __forceinline__ float2 sp_phasor_rev(float rev) {
    const float f = rev - rintf(rev);                  // [-0.5, 0.5]
    const float qf = rintf(4.0f * f);                  // -2 .. 2
    const float z = fmaf(-0.25f, qf, f) * 6.28318530717958647692f;   // |z| <= pi/4
    const float z2 = z * z;
    const float sn = fmaf(fmaf(fmaf(-1.9515295891e-4f, z2, 8.3321608736e-3f), z2, -1.6666654611e-1f),
                          z2 * z, z);
    const float co = fmaf(fmaf(fmaf(2.443315711809948e-5f, z2, -1.388731625493765e-3f), z2,
                               4.166664568298827e-2f), z2 * z2, fmaf(-0.5f, z2, 1.0f));
    const int q = (int)qf & 3;
    const float c = (q == 0) ? co : (q == 1) ? -sn : (q == 2) ? -co : sn;
    const float sg = (q == 0) ? sn : (q == 1) ? co : (q == 2) ? -sn : -co;
    return make_float2(c, -sg);
}
__device__ __forceinline__ float2 sp_cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
// the row factor U[q] = exp(-j om q T) of a channel (angle reduced in double)
__device__ __forceinline__ float2 sp_row_factor(float om, int q) {
    const double fr = (double)om * 0.15915494309189533576888376337251 * 1.0e-3;
    const double r = fr * (double)q;
    return sp_phasor_rev((float)(r - rint(r)));
}
// partial[q + 1] = U[q] hi[q] + U[q+1] lo[q+1]
__device__ __forceinline__ float2 sp_window(float hx, float hy, float lx, float ly, float2 u0, float2 u1) {
    const float re = fmaf(hx, u0.x, -(hy * u0.y)) + fmaf(lx, u1.x, -(ly * u1.y));
    const float im = fmaf(hy, u0.x, hx * u0.y) + fmaf(ly, u1.x, lx * u1.y);
    return make_float2(re, im);
}

// B of four pairs for both N tiles and two steps of the recurrence, in program order:
// every B value is written at least four instructions before the block ends, so the MFMAs
// behind it need no wait state.  nk = (-kappa of tile 0, -kappa of tile 1).
__device__ __forceinline__ void sp_b4(sp4 c0, sp4 c1, sp2 nk, sp2& u0, sp2& dl0, sp2& u1, sp2& dl1,
                                      sp2& b00, sp2& b01, sp2& b10, sp2& b11) {
    const sp2 c0a = {c0.x, c0.y}, c0b = {c0.z, c0.w}, c1a = {c1.x, c1.y}, c1b = {c1.z, c1.w};
    asm("v_pk_mul_f32 %0, %8, %4\n\t"
        "v_pk_mul_f32 %2, %10, %6\n\t"
        "v_pk_fma_f32 %5, %12, %4, %5 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %7, %12, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_add_f32 %4, %4, %5\n\t"
        "v_pk_add_f32 %6, %6, %7\n\t"
        "v_pk_mul_f32 %1, %9, %4\n\t"
        "v_pk_mul_f32 %3, %11, %6\n\t"
        "v_pk_fma_f32 %5, %12, %4, %5 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %7, %12, %6, %7 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_add_f32 %4, %4, %5\n\t"
        "v_pk_add_f32 %6, %6, %7"
        : "=&v"(b00), "=&v"(b01), "=&v"(b10), "=&v"(b11), "+v"(u0), "+v"(dl0), "+v"(u1), "+v"(dl1)
        : "v"(c0a), "v"(c0b), "v"(c1a), "v"(c1b), "v"(nk));
}
// the recurrence alone (the positions before a range that starts inside a quarter)
__device__ __forceinline__ void sp_rec(sp2 nk, sp2& u0, sp2& dl0, sp2& u1, sp2& dl1) {
    asm("v_pk_fma_f32 %1, %4, %0, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %4, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_add_f32 %2, %2, %3"
        : "+v"(u0), "+v"(dl0), "+v"(u1), "+v"(dl1)
        : "v"(nk));
}

__device__ __forceinline__ int sp_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// what a wave needs of a job descriptor
struct SpDesc {
    int delay_used, active, prn;
    float om, ph;
};
__device__ __forceinline__ SpDesc sp_desc(const JobMid* __restrict__ midrow, int c, bool have) {
    const JobMid m = midrow[have ? c : 0];
    SpDesc d;
    d.delay_used = m.delay_used; d.active = have && m.active; d.prn = m.prn; d.om = m.om; d.ph = m.ph;
    return d;
}
// the descriptors of a wave's lane roles: its two channels (one per N tile) and, for the lanes
// that stage the replica (l < 48), the channel l / 4
__device__ __forceinline__ void sp_wave_descs(const JobMid* __restrict__ midrow, int nch_g, int lane,
                                              SpDesc (&md)[2], SpDesc& smd) {
    const int j = lane & 15;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 8 * n + (j >> 1);
        md[n] = sp_desc(midrow, c, c < kSpCh && c < nch_g);
    }
    const int sc = lane >> 2;
    smd = sp_desc(midrow, sc, lane < 4 * kSpCh && sc < nch_g);
}

// ---- tile staging: 16 x b128 per lane; instruction i covers rows 2 i, 2 i + 1 (lane / 32) and
// 512 bytes of each.  Global address = scalar base + one lane offset (buffer loads; a tile
// position past the block reads nothing and returns zeros).  The rows are read once:
// non-temporal loads (3 % faster than the default policy).
// FMT 1: the block is raw uint16 (Q << 8 | I) samples as the recorder writes them
// (gpsrecv.py:168-173), 2 bytes per sample: four instructions per tile, instruction i covers
// rows 8 i + lane / 8 and 128 bytes (64 samples) of each; sp_store_tile decodes them.
template <int AUX = 2, int FMT = 0>
__device__ __forceinline__ void sp_load_tile(const void* blk, int pos, int lane, sp4 (&st)[16]) {
    constexpr int CS = kFftN, NC = 32;
    if (FMT == 0) {
        const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(blk), 0, CS * NC * (int)sizeof(float2), kMfRsrcFlags);
        const int ld_off = ((lane >> 5) * CS + 2 * (lane & 31)) * (int)sizeof(float2);
        const int tb = pos * (int)sizeof(float2);
#pragma unroll
        for (int i = 0; i < 16; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (2 * CS * (int)sizeof(float2)), AUX));
    } else {
        const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(blk), 0, CS * NC * 2, kMfRsrcFlags);
        const int ld_off = ((lane >> 3) * CS + 8 * (lane & 7)) * 2;
        const int tb = pos * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (8 * CS * 2), AUX));
    }
}
// the staged tile into LDS (interleaved re / im, row pitch kSpRowDw); FMT 1 decodes the raw
// samples exactly as gpsmi_dev_unpack_u8iq does: fl32(byte) * fl32(1 / 127.5) - 1, two roundings
template <int FMT>
__device__ __forceinline__ void sp_store_tile(float* tl, int lane, const sp4 (&st)[16]) {
    if (FMT == 0) {
        // rows are 8-byte aligned (pitch 130 dwords, which the transposed reads need): two b64 writes
        // per 16-byte piece.  Lanes c and c + 8 of a 16-lane store group would meet on one bank (their
        // pieces are 32 and 64 dwords apart: measured, tools/probe/lds_conflict_probe.hip, conflict /
        // active = 0.50) -- so the pieces with bit 3 of their index set keep their two samples in
        // SWAPPED order in LDS: each store instruction then covers all 32 banks once.  sp_tile_swz()
        // tells the readers.
        const int swz = ((lane & 31) >> 3) & 1;
        float* st_dst = tl + (lane >> 5) * kSpRowDw + 4 * (lane & 31);
        float* d0 = st_dst + 2 * swz;                  // where the piece's first sample goes
        float* d1 = st_dst + 2 * (1 - swz);            // ... and its second
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            *reinterpret_cast<sp2*>(d0 + i * 2 * kSpRowDw) = sp2{st[i].x, st[i].y};
            *reinterpret_cast<sp2*>(d1 + i * 2 * kSpRowDw) = sp2{st[i].z, st[i].w};
        }
    } else {
        float* st_dst = tl + (lane >> 3) * kSpRowDw + 16 * (lane & 7);
        const float scl = 1.0f / 127.5f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            // (the whole vector is reinterpreted at once: with hipcc 7.2 a per-element
            // __builtin_bit_cast(unsigned, st[i][w]) made the compiler narrow the 16-byte load
            // to its first dword and use that for all four)
            typedef unsigned sp_u4 __attribute__((ext_vector_type(4)));
            const sp_u4 q = __builtin_bit_cast(sp_u4, st[i]);
            const unsigned vv[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int w = 0; w < 4; ++w) {              // one dword = two samples
                const unsigned v = vv[w];
                const sp2 s0 = {(float)(v & 0xFF) * scl - 1.0f, (float)((v >> 8) & 0xFF) * scl - 1.0f};
                const sp2 s1 = {(float)((v >> 16) & 0xFF) * scl - 1.0f, (float)(v >> 24) * scl - 1.0f};
                *reinterpret_cast<sp2*>(st_dst + i * 8 * kSpRowDw + 4 * w) = s0;
                *reinterpret_cast<sp2*>(st_dst + i * 8 * kSpRowDw + 4 * w + 2) = s1;
            }
        }
    }
}

// One wave: NSPANS consecutive spans starting at position `pos0` (a multiple of 64) of a
// block.  The caller has requested the first tile into `st` (sp_load_tile) and fetched the
// descriptors; on return `st` holds the request for the tile at `next_pos` of `next_blk` (the
// first tile of the wave's next range, or a position past the block: nothing).  Returns the
// sums of the range for [M tile][N tile] in (tot, lo_fin): lo_fin = what was summed below the
// delay when the boundary lies inside the range, tot = the rest.
// (DIAG: tools/probe/span_prof.hip only -- 1 no MFMAs, 2 no row loads after the first tile, 8 default
// cache policy; the library instantiates DIAG = 0 alone, where every test of it folds away)
template <int NSPANS, int FMT = 0, int DIAG = 0>
__device__ __forceinline__ void span_wave(const void* __restrict__ blk, const void* next_blk,
                                          int next_pos, float* tl, float* cd, const SpDesc (&mdd)[2],
                                          const SpDesc& smd, const float* __restrict__ code_eo,
                                          int pos0, int lane, sp4 (&st)[16], sp4 (&tot)[2][2],
                                          sp4 (&lo_fin)[2][2], bool (&all_lo)[2], int (&pb)[2]) {
    constexpr int CS = kFftN;
    const int j = lane & 15, k = lane >> 4, pi = k >> 1, kap = k & 1, part = j & 1;
    const int q0 = pos0 & ~(kSpQuarter - 1);                      // start of the quarter
    constexpr int kTiles = NSPANS;
    auto store_tile = [&]() { sp_store_tile<FMT>(tl, lane, st); };

    // ---- lane roles: two channels (one per N tile); the recurrence of both, seeded with the
    // exact phasor of the lane's first two positions of the QUARTER
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float inv_fs = 1.0f / (1000.0f * (float)CS);
    const float sx = ((kap == 0) == (part == 0)) ? 1.f : 0.f;          // (0,re) and (1,im): +z.x
    const float sy = (sx != 0.f) ? 0.f : (part == 0 ? -1.f : 1.f);     // (1,re): -z.y, (0,im): +z.y
    // pb: boundary relative to the quarter start, kSpInf if outside (0, 512)
    const float* crow[2];             // LDS: the lane's replica rows of the current window
    sp2 u2[2], dl2[2], nk;
    float nkv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 8 * n + (j >> 1);
        const SpDesc& md = mdd[n];
        const bool active = md.active;
        const float f_eff = active ? (float)((double)md.om * inv_2pi) : 0.f;
        const float ph_rev = active ? md.ph * (float)inv_2pi : 0.f;
        const float2 w2 = sp_phasor_rev(2.0f * (f_eff * inv_fs));      // exp(-j 2 phi)
        const float sh = -w2.y, chh = w2.x;
        nkv[n] = -4.0f * sh * sh;                                      // -4 sin^2(2 phi): step of four positions
        const float2 omw4 = make_float2(2.0f * sh * sh, -2.0f * sh * chh);   // 1 - exp(+j 4 phi)
        const float2 z0 = sp_phasor_rev(fmaf(f_eff, (float)(q0 + pi + 1) * inv_fs, ph_rev));
        const float2 z1 = sp_cmul(z0, w2);
        const float2 dz0 = sp_cmul(z0, omw4), dz1 = sp_cmul(z1, omw4);   // z(m) - z(m - 4)
        u2[n] = sp2{active ? fmaf(sx, z0.x, sy * z0.y) : 0.f, active ? fmaf(sx, z1.x, sy * z1.y) : 0.f};
        dl2[n] = sp2{active ? fmaf(sx, dz0.x, sy * dz0.y) : 0.f,
                     active ? fmaf(sx, dz1.x, sy * dz1.y) : 0.f};
        const int d = active ? md.delay_used : 0;
        const int rel = d - q0;                                        // boundary relative to the quarter
        all_lo[n] = active && rel >= kSpQuarter;
        pb[n] = (active && rel > 0 && rel < kSpQuarter) ? rel : kSpInf;
        crow[n] = cd + ((c < kSpCh ? c : 0) * 2 + pi) * kSpCodePitch;
    }
    nk = sp2{nkv[0], nkv[1]};                                          // -kappa of both tiles in one pair
    const int rel0 = pos0 - q0;                                        // range start within the quarter
    // a range that starts inside the quarter: the recurrence steps of the positions before it
#pragma unroll 1
    for (int s = 0; s < rel0 / 4; ++s) sp_rec(nk, u2[0], dl2[0], u2[1], dl2[1]);

    // ---- replica staging: lane l < 48 fetches 8 consecutive entries of one (channel, parity)
    // row of a 32-position window: channel l / 4, parity (l / 2) & 1, half l & 1.  Tile
    // position m = 2 q + pi, rolled index r = (m - d) mod 2048 has parity e = (pi - d) & 1 and
    // half index (r - e) / 2, which advances by one per pair: a contiguous run of the table
    // plane e (doubled, so the run never wraps).  A closed channel reads PRN slot 0 (zeros).
    const int sc = lane >> 2, spl = (lane >> 1) & 1, shf = lane & 1;
    const bool s_act = smd.active;
    const __amdgpu_buffer_rsrc_t code_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(code_eo), 0, (GPSMI_MAX_PRN + 1) * 2 * CS * (int)sizeof(float), kMfRsrcFlags);
    int sc_off;
    {
        const int d = s_act ? smd.delay_used : 0;
        const int e = (spl - d) & 1;
        const int hidx = ((pos0 + spl - d - e) & (CS - 1)) >> 1;          // 0 .. 1023
        sc_off = ((s_act ? smd.prn : 0) * (2 * CS) + e * CS + hidx + 8 * shf) * (int)sizeof(float);
    }
    float* sdst = cd + ((lane < 4 * kSpCh ? sc : 0) * 2 + spl) * kSpCodePitch + 8 * shf;
    sp4 cst[2];
    auto load_code = [&](int win) {                    // window index from pos0, 16 entries each
#pragma unroll
        for (int i = 0; i < 2; ++i)
            cst[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                code_rs, sc_off, (win * (kSpWin / 2) + 4 * i) * (int)sizeof(float), 0));
    };
    auto store_code = [&]() {
        if (lane < 4 * kSpCh) {
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<sp4*>(sdst + 4 * i) = cst[i];
        }
    };
    load_code(0);

    const sp4 zero4 = sp4{0.f, 0.f, 0.f, 0.f};
    sp4 acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int n = 0; n < 2; ++n) { acc[mt][n] = zero4; tot[mt][n] = zero4; lo_fin[mt][n] = zero4; }
    // the four lanes of a channel close its lo sum where the boundary passes
    auto close_lo = [&](int n) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            lo_fin[mt][n] = tot[mt][n] + acc[mt][n];
            tot[mt][n] = zero4;
            acc[mt][n] = zero4;
        }
    };
    // the boundaries of the group's channels as scalars (channel c sits in lane 2 (c % 8) of N
    // tile c / 8): the next one is a dozen scalar compares, no cross-lane reduction in the loop
    int pbs[kSpCh];
#pragma unroll
    for (int c = 0; c < kSpCh; ++c) pbs[c] = __builtin_amdgcn_readlane(pb[c >> 3], 2 * (c & 7));
    auto next_boundary = [&](int after) {                            // first boundary position > after
        int v = kSpInf;
#pragma unroll
        for (int c = 0; c < kSpCh; ++c)
            if (pbs[c] > after && pbs[c] < rel0 + NSPANS * kSpTile) v = pbs[c] < v ? pbs[c] : v;
        return v;
    };
    int nb = next_boundary(rel0);        // (a boundary AT the range start needs no action: all hi)
    const float* ap0 = tl + (lane & 15) * kSpRowDw + k;              // lane = (row, k), rows 0 .. 15
    const float* ap1 = ap0 + 16 * kSpRowDw;                          // rows 16 .. 31
    // pairs 8 .. 15 and 24 .. 31 of a complex64 tile sit with their two positions swapped in LDS
    // (sp_store_tile): the lane of parity pi reads the other half there, k ^ 2
    const int swz_ofs = FMT == 0 ? (2 - 4 * pi) : 0;                 // (k ^ 2) - k

#pragma unroll 1
    for (int tix = 0; tix < kTiles; ++tix) {
        // the tile that waited in registers goes to LDS (the reads of the previous one are
        // behind us: LDS serves a wave in order), then the tile after it is requested
        // (unconditional: behind the last tile of the range it is the first one of the wave's
        // next range, or a position past the block, which costs no memory traffic)
        store_code();
        store_tile();
        load_code(2 * tix + 1);
        if (!(DIAG & 2)) {
            const bool more = tix + 1 < kTiles;            // else: the first tile of the wave's next range
            sp_load_tile<(DIAG & 8) ? 0 : 2, FMT>(more ? blk : next_blk,
                                                  more ? pos0 + (tix + 1) * kSpTile : next_pos, lane, st);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int tpos = rel0 + tix * kSpTile;                       // tile start within the quarter

        // operands of four pairs (eight positions): the lane's component of the samples of
        // both row halves and the replica entries of its two channels
        struct Ops { sp4 a0, a1, c0, c1; };
        auto read_ops = [&](int q4) {                                // q4: pair index in the tile, multiple of 4
            Ops o;
            const int sw = ((q4 >> 3) & 1) ? swz_ofs : 0;            // (four pairs never straddle a multiple of 8)
            const float* a0p = ap0 + sw;
            const float* a1p = ap1 + sw;
            o.a0 = sp4{a0p[4 * q4], a0p[4 * q4 + 4], a0p[4 * q4 + 8], a0p[4 * q4 + 12]};
            o.a1 = sp4{a1p[4 * q4], a1p[4 * q4 + 4], a1p[4 * q4 + 8], a1p[4 * q4 + 12]};
            o.c0 = *reinterpret_cast<const sp4*>(crow[0] + (q4 & 15));
            o.c1 = *reinterpret_cast<const sp4*>(crow[1] + (q4 & 15));
            return o;
        };
        auto mfma4 = [&](float a0, float a1, float b0, float b1) {
            if (DIAG & 1) {
                acc[0][0][0] = fmaf(a0, b0, acc[0][0][0]);
                acc[1][1][0] = fmaf(a1, b1, acc[1][1][0]);
            } else {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        };
        // the first pair of a span: C = 0 (an inline constant), no zeroing of the accumulators
        auto mfma4_first = [&](float a0, float a1, float b0, float b1) {
            if (DIAG & 1) {
                acc[0][0] = zero4; acc[0][1] = zero4; acc[1][0] = zero4; acc[1][1] = zero4;
                acc[0][0][0] = a0 * b0;
                acc[1][1][0] = a1 * b1;
            } else {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, zero4, 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, zero4, 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, zero4, 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, zero4, 0, 0, 0);
            }
        };
        auto four = [&](const Ops& o, int q4, auto check, auto first) {
            sp2 b00, b01, b10, b11;
            sp_b4(o.c0, o.c1, nk, u2[0], dl2[0], u2[1], dl2[1], b00, b01, b10, b11);
            const float b0v[4] = {b00.x, b00.y, b01.x, b01.y};
            const float b1v[4] = {b10.x, b10.y, b11.x, b11.y};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float bv[2] = {b0v[i], b1v[i]};
                if (decltype(check)::value) {
                    const int P = tpos + 2 * (q4 + i);               // positions P, P + 1
                    if (nb <= P + 1) {                               // some channel's boundary is here
                        bool odd[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            odd[n] = pb[n] == P + 1;
                            if (pb[n] == P) close_lo(n);             // even boundary: close lo before the pair
                        }
                        if (__builtin_amdgcn_ballot_w64(odd[0] || odd[1]) != 0) {
                            // position P alone for the boundary lanes, both positions elsewhere
                            mfma4(o.a0[i], o.a1[i], odd[0] ? (pi == 0 ? bv[0] : 0.f) : bv[0],
                                  odd[1] ? (pi == 0 ? bv[1] : 0.f) : bv[1]);
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                if (odd[n]) close_lo(n);
                                bv[n] = odd[n] ? (pi == 1 ? bv[n] : 0.f) : 0.f;   // then position P + 1 of those lanes
                            }
                        }
                        nb = next_boundary(P + 1);
                    }
                }
                if (decltype(first)::value && i == 0) mfma4_first(o.a0[i], o.a1[i], bv[0], bv[1]);
                else mfma4(o.a0[i], o.a1[i], bv[0], bv[1]);
            }
        };
        // one replica window = 16 pairs = four groups of four pairs
        auto half_tile = [&](int hw, auto first) {
            if (nb >= tpos + (hw + 1) * kSpWin) {
                Ops o[2];
                o[0] = read_ops(16 * hw);
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    if (gq + 1 < 4) o[(gq + 1) & 1] = read_ops(16 * hw + 4 * (gq + 1));
                    __builtin_amdgcn_sched_barrier(0);
                    if (gq == 0) four(o[0], 16 * hw, std::false_type{}, first);
                    else four(o[gq & 1], 16 * hw + 4 * gq, std::false_type{}, std::false_type{});
                }
            } else {
                Ops cur = read_ops(16 * hw);
                if (decltype(first)::value) {           // (the accumulators are read where a boundary closes)
                    acc[0][0] = zero4; acc[0][1] = zero4; acc[1][0] = zero4; acc[1][1] = zero4;
                }
#pragma unroll 1
                for (int q4 = 0; q4 < 16; q4 += 4) {
                    const Ops nxt = read_ops(16 * hw + (q4 + 4 < 16 ? q4 + 4 : q4));   // last: harmless re-read
                    __builtin_amdgcn_sched_barrier(0);
                    if (nb >= tpos + 2 * (16 * hw + q4 + 4))
                        four(cur, 16 * hw + q4, std::false_type{}, std::false_type{});
                    else four(cur, 16 * hw + q4, std::true_type{}, std::false_type{});
                    cur = nxt;
                }
            }
        };
        half_tile(0, std::true_type{});
        // second window of the tile: its entries were requested before the rows of the next
        // tile, so waiting for them never waits for rows from HBM
        store_code();
        load_code(2 * tix + 2);
        half_tile(1, std::false_type{});
        // a span ends (its accumulators restart from C = 0 at the next span)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int n = 0; n < 2; ++n) tot[mt][n] = tot[mt][n] + acc[mt][n];
    }
}

// ---- the kernels.  18.1 KiB of LDS per wave = eight waves per CU.
//   batch form:        range = quarter; the four quarters of a block are the four waves of
//                      one workgroup (so the four 4 KiB pieces of every 16 KiB row are
//                      requested together).  The waves leave their hi / lo row sums in LDS;
//                      behind a barrier all threads add the quarters in their fixed order and
//                      write partial[job][.] (3 KiB per block).  The workgroups are
//                      persistent (two per CU, each takes every gridDim-th block): the
//                      first rows and the descriptors of the next block are requested before
//                      the combine step of the current one.
//                      Measured instead, per 1024-block launch: the wave that arrives last
//                      combines alone, the others exit without a barrier +6 us (the
//                      workgroup's LDS is held until that wave is done); one-wave workgroups
//                      that write 4 KiB records of raw sums +6 us, four-wave ones +9 us.
//   single-block form: range = span, 32 one-wave workgroups per (block, channel group), for
//                      a launch too small to fill the CUs with quarters (the closed loop).
//                      A wave writes the raw sums of its span (tot always, lo_fin by the
//                      lanes whose boundary lay inside it) to `rec`, 2048 floats per wave:
//                        rec[unit = b * ngroups + g][span][tot | lo_fin][M tile][N tile][v][lane]
//                      and span_collect (called by the epilogue kernel) adds the spans of a
//                      block up in the order the batch form uses: same bits.
constexpr int kSpRecFloats = 2 * 16 * 64;              // one wave's record
constexpr int kSpLoOfs = 16 * 64;                      // lo_fin within it

// (FMT 1: iq holds raw uint16 samples, 2 bytes each, decoded on the way into LDS)
template <int NSPANS, int WAVES, int FMT = 0, int DIAG = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void trk_span_kernel(
    const void* __restrict__ iq_v, const JobMid* __restrict__ mid,
    const float* __restrict__ code_eo, TrkParams P, int ngroups, int nblocks,
    float* __restrict__ rec, float2* __restrict__ partial) {
    constexpr int NC = 32, CS = kFftN;
    constexpr int kRanges = CS / (NSPANS * kSpTile);    // per block: 4 quarters or 32 spans
    constexpr bool kWholeBlock = kRanges == WAVES;      // the workgroup holds all ranges of its block
    static_assert(kRanges % WAVES == 0, "the waves of a workgroup share a block");
    constexpr int kNowhere = CS * NC;                   // a tile position past the block: reads nothing
    constexpr size_t kBlkBytes = (size_t)CS * NC * (FMT == 0 ? sizeof(float2) : 2);
    const char* iq = static_cast<const char*>(iq_v);
    __shared__ __attribute__((aligned(16))) float lds[WAVES][kSpWaveFloats];
    __shared__ float4 ufac[kWholeBlock ? kSpCh * (NC + 1) : 1];      // (U[q], U[q+1]); .w = NaN: channel closed
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nunits = nblocks * ngroups;
    float* tl = &lds[wave][0];
    sp4 st[16], tot[2][2], lo_fin[2][2];
    bool all_lo[2];
    int pb[2];
    SpDesc md[2], smd;

    if (!kWholeBlock) {
        // ---- single-block form: one span per wave, raw sums to `rec`
        const int widx = blockIdx.x * WAVES + wave;
        const int unit = widx / kRanges, range = widx % kRanges;
        if (unit >= nunits) return;
        const int g = unit % ngroups, b = unit / ngroups;
        const char* blk = iq + (size_t)b * kBlkBytes;
        const int pos0 = range * NSPANS * kSpTile;
        sp_load_tile<(DIAG & 8) ? 0 : 2, FMT>(blk, pos0, lane, st);   // before anything that depends on the descriptors
        sp_wave_descs(mid + (size_t)b * P.nch + g * kSpCh, P.nch - g * kSpCh, lane, md, smd);
        span_wave<NSPANS, FMT, DIAG>(blk, blk, kNowhere, tl, tl + kSpTileFloats, md, smd, code_eo, pos0, lane, st,
                                     tot, lo_fin, all_lo, pb);
        float* o = rec + ((size_t)unit * kRanges + range) * kSpRecFloats + lane;
        const int rel0 = pos0 & (kSpQuarter - 1);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const bool closed = pb[n] > rel0 && pb[n] < rel0 + NSPANS * kSpTile;   // lo was closed in this range
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    o[((mt * 2 + n) * 4 + v) * 64] = tot[mt][n][v];
                    if (closed) o[kSpLoOfs + ((mt * 2 + n) * 4 + v) * 64] = lo_fin[mt][n][v];
                }
        }
        return;
    }

    // ---- batch form: the workgroup takes units blockIdx.x, blockIdx.x + gridDim.x, ...; the
    // first rows and the descriptors of the next unit are requested before the combine step of
    // the current one, so only the first unit of a workgroup waits for memory at its start
    constexpr int kSumFloats = kSpCh * NC * 2;          // [channel][row][re, im]
    static_assert(2 * kSumFloats <= kSpTileFloats, "row sums must fit the tile area");
    constexpr int kItems = (kSpCh * (NC + 1) + 64 * WAVES - 1) / (64 * WAVES);
    int unit = blockIdx.x;
    if (unit >= nunits) return;
    const int pos0 = wave * kSpQuarter;
    // for the combine step: which partial[q + 1] a thread writes, whether its channel is open
    // and the row factors U[q], U[q+1].  The descriptor is fetched with the unit's others, the
    // factors (double arithmetic) are formed at the start of the unit and wait in LDS: nothing
    // of this stays in registers across the tile loop or is left for the tail of the block
    bool on[kItems];
    float om_item[kItems];
    auto fetch_items = [&](int u) {
        const int g = u % ngroups, b = u / ngroups;
#pragma unroll
        for (int e = 0; e < kItems; ++e) {
            const int item = (int)threadIdx.x + 64 * WAVES * e;
            const int ci = g * kSpCh + item / (NC + 1);
            on[e] = item < kSpCh * (NC + 1) && ci < P.nch;
            const JobMid m2 = mid[b * P.nch + (on[e] ? ci : 0)];
            on[e] = on[e] && m2.active;
            om_item[e] = m2.om;
        }
    };
    auto park_items = [&]() {
#pragma unroll
        for (int e = 0; e < kItems; ++e) {
            const int item = (int)threadIdx.x + 64 * WAVES * e;
            if (item >= kSpCh * (NC + 1)) continue;
            const int q = item % (NC + 1) - 1;
            const float2 a0 = sp_row_factor(om_item[e], q), a1 = sp_row_factor(om_item[e], q + 1);
            ufac[item] = make_float4(a0.x, a0.y, a1.x, on[e] ? a1.y : __builtin_nanf(""));
        }
    };
    {
        const int g = unit % ngroups, b = unit / ngroups;
        sp_load_tile<(DIAG & 8) ? 0 : 2, FMT>(iq + (size_t)b * kBlkBytes, pos0, lane, st);
        sp_wave_descs(mid + (size_t)b * P.nch + g * kSpCh, P.nch - g * kSpCh, lane, md, smd);
        fetch_items(unit);
    }
#pragma unroll 1
    for (;;) {
        const int g = unit % ngroups, b = unit / ngroups;
        park_items();
        // (the lane-derived constants of a wave's set-up are recomputed per unit instead of
        // being carried through the tile loop in registers the loop needs)
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));
        const char* blk = iq + (size_t)b * kBlkBytes;
        const int next = unit + (int)gridDim.x;
        const bool has_next = next < nunits;
        const char* next_blk = has_next ? iq + (size_t)(next / ngroups) * kBlkBytes : blk;
        span_wave<NSPANS, FMT, DIAG>(blk, next_blk, has_next ? pos0 : kNowhere, tl, tl + kSpTileFloats, md, smd, code_eo,
                                pos0, lane_u, st, tot, lo_fin, all_lo, pb);
        // the hi / lo sums of every row of the quarter into LDS (the wave's own tile area),
        // D[i = 4 (lane / 16) + v][j = lane % 16]
        if (!(DIAG & 4)) {
            float* hi = tl;
            float* lo = tl + kSumFloats;
            const int j = lane & 15, part = j & 1;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int c = 8 * n + (j >> 1);
                const sp4 z = sp4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const sp4 w_hi = all_lo[n] ? z : tot[mt][n];
                    const sp4 w_lo = all_lo[n] ? tot[mt][n] : lo_fin[mt][n];
                    if (c < kSpCh) {
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const int row = 16 * mt + 4 * (lane >> 4) + v;
                            hi[(c * NC + row) * 2 + part] = w_hi[v];
                            lo[(c * NC + row) * 2 + part] = w_lo[v];
                        }
                    }
                }
            }
        }
        // the descriptors of the next unit on their way while the quarters are combined
        if (has_next) {
            sp_wave_descs(mid + (size_t)(next / ngroups) * P.nch + (next % ngroups) * kSpCh,
                          P.nch - (next % ngroups) * kSpCh, lane, md, smd);
            fetch_items(next);
        }
        if (!(DIAG & 4)) {
            // behind one barrier all threads add the quarters in their fixed order, apply U and
            // write partial[q + 1], q = -1 .. 31
            __syncthreads();
#pragma unroll
            for (int e = 0; e < kItems; ++e) {
                const int item = (int)threadIdx.x + 64 * WAVES * e;
                if (item >= kSpCh * (NC + 1)) continue;
                const float4 uf = ufac[item];
                if (uf.w != uf.w) continue;                 // closed channel
                const int cc = item / (NC + 1), o = item % (NC + 1), q = o - 1;
                float hx = 0.f, hy = 0.f, lx = 0.f, ly = 0.f;
#pragma unroll
                for (int w = 0; w < WAVES; ++w) {
                    const float* hi = &lds[w][0];
                    const float* lo = hi + kSumFloats;
                    if (q >= 0) { hx += hi[(cc * NC + q) * 2]; hy += hi[(cc * NC + q) * 2 + 1]; }
                    if (q + 1 < NC) { lx += lo[(cc * NC + q + 1) * 2]; ly += lo[(cc * NC + q + 1) * 2 + 1]; }
                }
                partial[((size_t)b * P.nch + g * kSpCh + cc) * (NC + 1) + o] =
                    sp_window(hx, hy, lx, ly, make_float2(uf.x, uf.y), make_float2(uf.z, uf.w));
            }
        }
        if (DIAG & 4) {                                  // (probe: keep the sums alive without the combine step)
            if (tot[0][0][0] + tot[0][1][1] + tot[1][0][2] + tot[1][1][3] + lo_fin[0][0][0] + lo_fin[1][1][1] == 123.456f)
                partial[(size_t)unit * 64 + lane] = make_float2(tot[0][0][0], lo_fin[1][1][1]);
        }
        if (!has_next) break;
        if (!(DIAG & 4)) __syncthreads();               // the tile areas are free again
        unit = next;
    }
}

// The sums of quarter Q of channel `cidx` of block `b`, for lane = (row, re/im): what the
// quarter contributes to the hi and to the lo side of the delay `d`.  NSPANS = spans per record,
// as the kernel that wrote them.  (Loads first, then the adds in the fixed order.)
template <int NSPANS>
__device__ __forceinline__ void span_collect_quarter(const float* __restrict__ rec, int ngroups, int b,
                                                     int cidx, int d, int lane, int Q, float& w_hi,
                                                     float& w_lo) {
    constexpr int kPerQ = kSpQuarter / (NSPANS * kSpTile);   // records per quarter: 1 or 8
    constexpr int kLen = NSPANS * kSpTile;                   // positions per record
    const int g = cidx / kSpCh, cc = cidx % kSpCh;
    const int r = lane >> 1, part = lane & 1;
    const int n = cc >> 3, j = 2 * (cc & 7) + part, mt = r >> 4, rg = (r & 15) >> 2, v = r & 3;
    const float* src = rec + ((size_t)(b * ngroups + g) * (4 * kPerQ) + (size_t)kPerQ * Q) * kSpRecFloats
                       + ((mt * 2 + n) * 4 + v) * 64 + 16 * rg + j;
    const int rel = d - Q * kSpQuarter;
    const bool all_lo = rel >= kSpQuarter;
    const int pb = (rel > 0 && rel < kSpQuarter) ? rel : kSpInf;
    float t[kPerQ], lf = 0.f;
#pragma unroll
    for (int s = 0; s < kPerQ; ++s) t[s] = src[(size_t)s * kSpRecFloats];
    if (pb != kSpInf && pb % kLen != 0) lf = src[(size_t)(pb / kLen) * kSpRecFloats + kSpLoOfs];
    float T = 0.f, L = 0.f;
    if (kPerQ == 1) {
        // the wave that summed the quarter closed lo itself where the boundary passed
        T = t[0];
        L = (pb != kSpInf) ? lf : 0.f;
    } else {
#pragma unroll
        for (int s = 0; s < kPerQ; ++s) {
            if (pb >= s * kLen && pb < (s + 1) * kLen) {       // lo closes in (or exactly at the start of) this span
                L = T + (pb == s * kLen ? 0.f : lf);
                T = 0.f;
            }
            T = T + t[s];
        }
    }
    w_hi = all_lo ? 0.f : T;
    w_lo = all_lo ? T : L;
}

// hi / lo (64 floats of LDS each, lane = (row, re/im)) -> S[0 .. 32] = what the other
// correlators write to partial[job][.]; one wave
__device__ __forceinline__ void span_windows(const float* hi, const float* lo, float om, int lane, float2* S) {
    constexpr int NC = 32;
    if (lane <= NC) {
        const int q = lane - 1;
        const float hx = q >= 0 ? hi[2 * q] : 0.f, hy = q >= 0 ? hi[2 * q + 1] : 0.f;
        const float lx = q + 1 < NC ? lo[2 * (q + 1)] : 0.f, ly = q + 1 < NC ? lo[2 * (q + 1) + 1] : 0.f;
        S[lane] = sp_window(hx, hy, lx, ly, sp_row_factor(om, q), sp_row_factor(om, q + 1));
    }
}

// One wave per job: the four quarters in turn, then the windows.
template <int NSPANS>
__device__ __forceinline__ void span_collect(const float* __restrict__ rec, int ngroups, int b, int cidx,
                                             int d, float om, int lane, float* hi, float* lo, float2* S) {
    float h = 0.f, l = 0.f;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        float wh, wl;
        span_collect_quarter<NSPANS>(rec, ngroups, b, cidx, d, lane, Q, wh, wl);
        h += wh;
        l += wl;
    }
    hi[lane] = h;
    lo[lane] = l;
    __builtin_amdgcn_wave_barrier();
    span_windows(hi, lo, om, lane, S);
    __builtin_amdgcn_wave_barrier();
}

}  // namespace gpsmi
