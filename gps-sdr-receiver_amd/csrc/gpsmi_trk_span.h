// The tracking correlator, span form (default for CS = 2048; N_CYC = 32, and -- round 4 -- 16 and 8:
// template parameter NC, one M tile of 16 rows instead of two, see "Other block lengths" below).
//
// Same mathematics as gpsmi_trk_stream_mfma.h -- prompt correlate-and-dump of a 32-ms
// block, y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed per
// code-period window (reference src/gpslib.py:1400-1420) -- re-cut so that
//   (1) every load instruction reads 512 consecutive bytes of a row (tiles of 32 rows x 64
//       positions; 256-byte segments top out at 5.2 TB/s, 512-byte ones reach 5.7-6.0,
//       tools/probe/tile_read.hip),
//   (2) a batch of 1024 blocks is two full rounds of workgroups (two workgroups of four
//       waves per CU, 16.3 KiB of LDS per wave) with no one-wave-per-SIMD tail,
//   (3) the order of the float32 sums is defined by the DATA, not by the launch: a block
//       is cut into 32 SPANS of 64 positions; a span is summed K-step by K-step (four
//       positions), the spans of a quarter (512 positions) are added in order, then the four
//       quarters.  Any assignment of spans to waves that follows this order gives the same
//       bits: the batch form gives a wave a quarter (eight spans), the single-block form gives
//       every span its own wave (32 waves on up to 32 CUs instead of 4 waves on one).
//
// THREE real products per complex one (round 3; the round-2 form spent four).  With
// x = a + j b and B = c + j d,
//     P1 = (a + b) c,   P2 = a (d - c),   P3 = b (c + d):   re = P1 - P3,  im = P1 + P2,
// so a K-step of FOUR positions is three real matrix products on v_mfma_f32_16x16x4_f32,
// D[16 x 16] += A[16 x 4] B[4 x 16], with M = 16 code periods (two M tiles = the 32 rows of the
// block), K = four positions, N = channel (12 of 16 columns; the four-product form had
// N = (channel, re/im) and K = (position pair, re/im)): 96 MFMAs per tile instead of 128.
//     A[r][k]  = (a + b | a | b) of x[r][4 s + k],        lane = (row r = lane % 16, k = lane / 16)
//     B[k][c]  = replica_c[(m - d_c) mod 2048] * (u1 | u2 | u3)_c(m), m = 4 s + k, lane = (c, k)
//     D[r][c]  = three accumulators k1, k2, k3 per M tile, lane = (c, row group)
// The fp32 MFMA runs on the SIMD's own fp32 lanes: no other VALU instruction issues
// meanwhile, so every VALU instruction between two MFMAs adds to the SIMD's time.  The three
// B components (z.re, z.im - z.re, z.re + z.im of the carrier phasor z = exp(-j theta)) are
// linear in (cos, sin), so each follows the same coupled recurrence dl -= kappa u, u += dl:
// a lane holds them for ITS channel at positions m, m + 4 in packed register pairs and advances
// eight positions per step (kappa = 4 sin^2(4 phi)): nine packed instructions per two K-steps
// (twelve MFMAs) in one asm block, plus one v_add_f32 per A operand for a + b.  The exact
// phasor is taken at the start of every quarter only (64 steps, as far as the recurrence stays
// accurate); a wave that starts at a later span of the quarter runs the same recurrence from
// the quarter start without the MFMAs, so it forms the same bits.  The replica comes straight
// from a table split by index mod 4 (a lane's positions are 4 apart): 16 consecutive entries
// per lane and tile, no LDS stage.
//
// Window q of the reference = positions m >= d of row q ("hi") plus m < d of row q+1
// ("lo").  Where the boundary d_c falls inside a wave's range the lanes of channel c close the
// lo sum (lo = sum so far) and restart; a K-step that holds a boundary is issued in pieces with
// B masked to the positions of the piece.  Only the half tile that holds a boundary takes that
// (rolled) path; everything else runs 48 MFMAs per half tile straight.
//
// What bounds the batch form now is the memory system (16 KiB in flight per wave at ~6 us of
// latency under load), so nothing outside the tile loop may wait for a global load: see the
// comments at sp_swap_tile, in span_wave (where the replica entries are requested) and at the
// batch form of the kernel (scalar descriptor fetch, LDS-only barriers, lo sums parked in LDS).
//
// Other block lengths (gpsglob.py:122-124: N_CYC "currently possible are (32,16,8)").  NC = 16 is ONE M
// tile (MT = 1) of the same scheme: tiles of 16 rows x 64 positions (8 KiB, eight loads per lane), half the
// MFMAs per tile and the same B arithmetic, 155 VGPRs -> three workgroups per CU.  NC = 8 stacks the M
// tile: rows 0..7 = Re x, rows 8..15 = Im x of the block's eight rows, TWO real products per K-step
// (B = replica * z.re, replica * z.im), re / im paired across the two lane halves once per span (kStack in
// span_wave), 122 VGPRs -> four workgroups per CU.  The order of the sums within a form is a property
// of the data, so the single-block form and the batch form stay bytewise equal for every NC.
#pragma once
#include <hip/hip_runtime.h>


#include <type_traits>

namespace gpsmi {

constexpr int kSpCh = 12;                      // channels per workgroup (of the 16 columns of an MFMA)
constexpr int kSpTile = 64;                    // positions per tile = per span (unit of the summation order)
constexpr int kSpQuarter = 512;                // positions per quarter (eight spans)
constexpr int kSpRowDw = 2 * kSpTile + 2;      // dwords per tile row: lane = (row, k) reads are conflict-free
constexpr int kSpTileFloats = 32 * kSpRowDw;
constexpr int kSpWaveFloats = kSpTileFloats;                     // 4160 floats = 16,640 B per wave
constexpr int kSpInf = 1 << 20;
constexpr int kRsrcFlags = 0x00020000;         // raw buffer descriptor, 32-bit data format (gfx9)


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
// ---- tile staging: 16 x b128 per lane; instruction i covers rows 2 i, 2 i + 1 (lane / 32) and
// 512 bytes of each.  Global address = scalar base + one lane offset (buffer loads; a tile
// position past the block reads nothing and returns zeros).  The rows are read once:
// non-temporal loads (3 % faster than the default policy).
// FMT 1: the block is raw uint16 (Q << 8 | I) samples as the recorder writes them
// (gpsrecv.py:168-173), 2 bytes per sample: four instructions per tile, instruction i covers
// rows 8 i + lane / 8 and 128 bytes (64 samples) of each; sp_store_tile decodes them.
template <int AUX = 2, int FMT = 0, int NC = 32>
__device__ __forceinline__ void sp_load_tile(const void* blk, int pos, int lane, sp4 (&st)[16]) {
    constexpr int CS = kFftN;
    if (FMT == 0) {
        const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(blk), 0, CS * NC * (int)sizeof(float2), kRsrcFlags);
        const int ld_off = ((lane >> 5) * CS + 2 * (lane & 31)) * (int)sizeof(float2);
        const int tb = pos * (int)sizeof(float2);
#pragma unroll
        for (int i = 0; i < NC / 2; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (2 * CS * (int)sizeof(float2)), AUX));
    } else {
        const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(blk), 0, CS * NC * 2, kRsrcFlags);
        const int ld_off = ((lane >> 3) * CS + 8 * (lane & 7)) * 2;
        const int tb = pos * 2;
#pragma unroll
        for (int i = 0; i < NC / 8; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (8 * CS * 2), AUX));
    }
}
// the staged tile into LDS (interleaved re / im, row pitch kSpRowDw); FMT 1 decodes the raw
// samples exactly as gpsmi_dev_unpack_u8iq does: fl32(byte) * fl32(1 / 127.5) - 1, two roundings
template <int FMT, int NC = 32>
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
        for (int i = 0; i < NC / 2; ++i) {
            *reinterpret_cast<sp2*>(d0 + i * 2 * kSpRowDw) = sp2{st[i].x, st[i].y};
            *reinterpret_cast<sp2*>(d1 + i * 2 * kSpRowDw) = sp2{st[i].z, st[i].w};
        }
    } else {
        float* st_dst = tl + (lane >> 3) * kSpRowDw + 16 * (lane & 7);
        const float scl = 1.0f / 127.5f;
#pragma unroll
        for (int i = 0; i < NC / 8; ++i) {
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

// The staged tile into LDS and the request for the next one, piece by piece: a register quad is
// asked for again as soon as it has been written to LDS, so a wave's requests never drain to zero
// while it waits for the last rows of a tile and stores them (loads return in order: the wait in
// front of piece i lets the 15 requests behind it stay in flight).
template <int AUX, int FMT, int NC = 32>
__device__ __forceinline__ void sp_swap_tile(float* tl, int lane, sp4 (&st)[16], const void* blk, int pos) {
    constexpr int CS = kFftN;
    if (FMT == 0) {
        const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<void*>(blk), 0, CS * NC * (int)sizeof(float2), kRsrcFlags);
        const int ld_off = ((lane >> 5) * CS + 2 * (lane & 31)) * (int)sizeof(float2);
        const int tb = pos * (int)sizeof(float2);
        const int swz = ((lane & 31) >> 3) & 1;              // (see sp_store_tile)
        float* st_dst = tl + (lane >> 5) * kSpRowDw + 4 * (lane & 31);
        float* d0 = st_dst + 2 * swz;
        float* d1 = st_dst + 2 * (1 - swz);
#pragma unroll
        for (int i = 0; i < NC / 2; ++i) {
            *reinterpret_cast<sp2*>(d0 + i * 2 * kSpRowDw) = sp2{st[i].x, st[i].y};
            *reinterpret_cast<sp2*>(d1 + i * 2 * kSpRowDw) = sp2{st[i].z, st[i].w};
            __builtin_amdgcn_sched_barrier(0);
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (2 * CS * (int)sizeof(float2)), AUX));
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        sp_store_tile<FMT, NC>(tl, lane, st);
        sp_load_tile<AUX, FMT, NC>(blk, pos, lane, st);
    }
}

// ======================================================================================
// Three real products per complex one (round 3).  y = sum x B with x = a + j b, B = c + j d:
//     P1 = (a + b) c,   P2 = a (d - c),   P3 = b (c + d):   re = P1 - P3,  im = P1 + P2,
// so a K-step of FOUR positions is three real matrix products with N = channel (12 of 16
// columns) instead of four products per PAIR with N = (channel, re/im): 96 MFMAs per tile of
// 32 rows x 64 positions where the four-product form needs 128.  The three B components
// (z.re, z.im - z.re, z.re + z.im of the carrier phasor z) are linear in (cos, sin): each
// follows the same coupled recurrence, one lane = one channel at positions m, m + 4 in a
// packed pair advanced eight positions per step (kappa = 4 sin^2(4 phi)).  a + b costs one
// v_add_f32 per A operand.  The replica comes straight from a table split by index mod 4
// (a lane's positions are 4 apart): 16 consecutive entries per lane and tile, no LDS stage.
//   A[r][k]  = (a + b | a | b) of x[r][4 s + k],        lane = (row r = lane % 16, k = lane / 16)
//   B[k][c]  = replica_c[(m - d_c) mod 2048] * (u1 | u2 | u3)_c(m), m = 4 s + k, lane = (c, k)
//   D[r][c]  = three accumulators k1, k2, k3 per M tile, lane = (c, row group)
// ======================================================================================
// the descriptor of a wave's lane role: channel lane % 16
__device__ __forceinline__ void sp_wave_desc(const JobMid* __restrict__ midrow, int nch_g, int lane, SpDesc& cmd) {
    const int c = lane & 15;
    cmd = sp_desc(midrow, c, c < kSpCh && c < nch_g);
}

// B of two K-steps (positions m and m + 4 of the lane) for the three products, then one step of
// the recurrences; every B value is written six instructions before the block ends
__device__ __forceinline__ void sp_b3(sp2 cd, sp2 nk, sp2& u1, sp2& d1, sp2& u2, sp2& d2, sp2& u3, sp2& d3,
                                      sp2& b1, sp2& b2, sp2& b3) {
    asm("v_pk_mul_f32 %0, %9, %3\n\t"
        "v_pk_mul_f32 %1, %9, %5\n\t"
        "v_pk_mul_f32 %2, %9, %7\n\t"
        "v_pk_fma_f32 %4, %10, %3, %4\n\t"
        "v_pk_fma_f32 %6, %10, %5, %6\n\t"
        "v_pk_fma_f32 %8, %10, %7, %8\n\t"
        "v_pk_add_f32 %3, %3, %4\n\t"
        "v_pk_add_f32 %5, %5, %6\n\t"
        "v_pk_add_f32 %7, %7, %8"
        : "=&v"(b1), "=&v"(b2), "=&v"(b3), "+v"(u1), "+v"(d1), "+v"(u2), "+v"(d2), "+v"(u3), "+v"(d3)
        : "v"(cd), "v"(nk));
}
// the recurrences alone (the positions before a range that starts inside a quarter)
__device__ __forceinline__ void sp_rec3(sp2 nk, sp2& u1, sp2& d1, sp2& u2, sp2& d2, sp2& u3, sp2& d3) {
    asm("v_pk_fma_f32 %1, %6, %0, %1\n\t"
        "v_pk_fma_f32 %3, %6, %2, %3\n\t"
        "v_pk_fma_f32 %5, %6, %4, %5\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_add_f32 %2, %2, %3\n\t"
        "v_pk_add_f32 %4, %4, %5"
        : "+v"(u1), "+v"(d1), "+v"(u2), "+v"(d2), "+v"(u3), "+v"(d3)
        : "v"(nk));
}

// The same for the stacked form of eight-row blocks (below): TWO components, B = replica * (z.re | z.im)
__device__ __forceinline__ void sp_b2(sp2 cd, sp2 nk, sp2& u1, sp2& d1, sp2& u2, sp2& d2, sp2& b1, sp2& b2) {
    asm("v_pk_mul_f32 %0, %6, %2\n\t"
        "v_pk_mul_f32 %1, %6, %4\n\t"
        "v_pk_fma_f32 %3, %7, %2, %3\n\t"
        "v_pk_fma_f32 %5, %7, %4, %5\n\t"
        "v_pk_add_f32 %2, %2, %3\n\t"
        "v_pk_add_f32 %4, %4, %5"
        : "=&v"(b1), "=&v"(b2), "+v"(u1), "+v"(d1), "+v"(u2), "+v"(d2)
        : "v"(cd), "v"(nk));
}
__device__ __forceinline__ void sp_rec2(sp2 nk, sp2& u1, sp2& d1, sp2& u2, sp2& d2) {
    asm("v_pk_fma_f32 %1, %4, %0, %1\n\t"
        "v_pk_fma_f32 %3, %4, %2, %3\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "v_pk_add_f32 %2, %2, %3"
        : "+v"(u1), "+v"(d1), "+v"(u2), "+v"(d2)
        : "v"(nk));
}
// the value the lane 32 away holds (the other half of the M tile's row groups)
__device__ __forceinline__ sp4 sp_swap32(sp4 v) {
    return sp4{__shfl_xor(v[0], 32, 64), __shfl_xor(v[1], 32, 64), __shfl_xor(v[2], 32, 64), __shfl_xor(v[3], 32, 64)};
}

// One wave: NSPANS consecutive spans starting at position `pos0` (a multiple of 64) of a
// block.  The caller has requested the first tile into `st` (sp_load_tile) and fetched the
// descriptor; on return `st` holds the request for the tile at `next_pos` of `next_blk` (the
// first tile of the wave's next range, or a position past the block: nothing).  Returns in
// `tot` the sums of the range for [M tile][re, im] of the lane's channel that lie above the
// delay (everything, if the boundary is not inside the range); what was summed below a
// boundary inside the range is handed to `close(mt, re, im)` at the moment the boundary passes,
// for the lanes of that channel only (the caller keeps it in registers or in LDS: 16 registers
// held through the tile loop for an event that happens once per channel and block were the
// difference between spilling and not).  `after_first_swap()` runs once, behind the first
// tile's store / request sequence: the place for loads that must not wait behind rows.
// (DIAG: tools/probe/span_prof.hip only -- 1 no MFMAs, 2 no row loads after the first tile, 8 default
// cache policy; the library instantiates DIAG = 0 alone, where every test of it folds away)
template <int NSPANS, int FMT = 0, int DIAG = 0, int NC = 32, class Close, class Hook>
__device__ __forceinline__ void span_wave(const void* __restrict__ blk, const void* next_blk,
                                          int next_pos, float* tl, const SpDesc& cmd,
                                          const float* __restrict__ code_q4, int pos0, int lane,
                                          sp4 (&st)[16], sp4 (&tot)[(NC + 15) / 16][2], bool& all_lo, int& pb,
                                          Close&& close, Hook&& after_first_swap) {
    constexpr int CS = kFftN;
    constexpr int MT = (NC + 15) / 16;                           // M tiles of 16 rows
    // EIGHT rows: the M tile is stacked, rows 0..7 = Re x of the block's rows, rows 8..15 = Im x, and a
    // K-step is TWO real products with B = replica * z.re and replica * z.im (x = a + j b, B = c + j d:
    // the first gives (a c ; b c), the second (a d ; b d)); re = a c - b d and im = a d + b c pair a
    // value of lanes 0..31 (row groups 0, 1: the a rows) with one of lanes 32..63 (the b rows): one
    // exchange per span.  Every row of M works (the three-product form would leave half of them empty:
    // 0.182 ms per 512 MiB, 37 % of the HBM peak), two recurrences instead of three, no a + b.
    constexpr bool kStack = NC == 8;
    const int k = lane >> 4;
    const int q0 = pos0 & ~(kSpQuarter - 1);                      // start of the quarter
    constexpr int kTiles = NSPANS;

    // ---- lane role: one channel; the three recurrences, seeded with the exact phasor of the
    // lane's first two positions (m, m + 4) of the QUARTER
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float inv_fs = 1.0f / (1000.0f * (float)CS);
    const bool active = cmd.active;
    sp2 u1, u2, u3, d1, d2, d3, nk;
    {
        const float f_eff = active ? (float)((double)cmd.om * inv_2pi) : 0.f;
        const float ph_rev = active ? cmd.ph * (float)inv_2pi : 0.f;
        const float2 w4 = sp_phasor_rev(4.0f * (f_eff * inv_fs));      // exp(-j 4 phi)
        const float sh = -w4.y, chh = w4.x;
        const float nkv = -4.0f * sh * sh;                             // -4 sin^2(4 phi): step of eight positions
        const float2 omw8 = make_float2(2.0f * sh * sh, -2.0f * sh * chh);   // 1 - exp(+j 8 phi)
        const float2 z0 = sp_phasor_rev(fmaf(f_eff, (float)(q0 + k + 1) * inv_fs, ph_rev));
        const float2 z1 = sp_cmul(z0, w4);
        const float2 dz0 = sp_cmul(z0, omw8), dz1 = sp_cmul(z1, omw8);   // z(m) - z(m - 8)
        const float on = active ? 1.f : 0.f;
        u1 = sp2{on * z0.x, on * z1.x};
        d1 = sp2{on * dz0.x, on * dz1.x};
        if (kStack) {
            u2 = sp2{on * z0.y, on * z1.y};
            d2 = sp2{on * dz0.y, on * dz1.y};
            u3 = sp2{0.f, 0.f}; d3 = u3;
        } else {
            u2 = sp2{on * (z0.y - z0.x), on * (z1.y - z1.x)};
            u3 = sp2{on * (z0.x + z0.y), on * (z1.x + z1.y)};
            d2 = sp2{on * (dz0.y - dz0.x), on * (dz1.y - dz1.x)};
            d3 = sp2{on * (dz0.x + dz0.y), on * (dz1.x + dz1.y)};
        }
        nk = sp2{nkv, nkv};
    }
    const int d_c = active ? cmd.delay_used : 0;
    {
        const int rel = d_c - q0;                                      // boundary relative to the quarter
        all_lo = active && rel >= kSpQuarter;
        pb = (active && rel > 0 && rel < kSpQuarter) ? rel : kSpInf;   // kSpInf if outside (0, 512)
    }
    const int rel0 = pos0 - q0;                                        // range start within the quarter
    // a range that starts inside the quarter: the recurrence steps of the positions before it
#pragma unroll 1
    for (int s = 0; s < rel0 / 8; ++s) {
        if (kStack) sp_rec2(nk, u1, d1, u2, d2);
        else sp_rec3(nk, u1, d1, u2, d2, u3, d3);
    }

    // ---- replica: position m = 4 s + k of the tile, rolled index r = (m - d) mod 2048 = 4 h + e
    // with e = (pos0 + k - d) & 3 fixed for the lane and h advancing by one per K-step: a
    // contiguous run of plane e of the table (each plane stored twice over, so the run never
    // wraps).  A closed channel reads PRN slot 0 (zeros).  Every request for entries is issued
    // at the top of a tile, IN FRONT of the row requests: loads return in order, so a wait for
    // entries (or for the registers they land in) never waits for rows.  The second half of the
    // tile goes straight to its place, the first half of the NEXT tile to a side buffer.
    // (The entries are float32, not signs: GPSCacode interpolates between chips, gpslib.py:62-77 --
    // 489 distinct values per replica.  A table of sign bits, tried in round 4, would take these loads
    // out of the tile loop -- 64 lanes in 64 different cache lines, four times per tile -- but not
    // reproduce the reference's replica.)
    const __amdgpu_buffer_rsrc_t code_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(code_q4), 0, (GPSMI_MAX_PRN + 1) * 2 * CS * (int)sizeof(float), kRsrcFlags);
    int cd_off;
    {
        const int r0 = (pos0 + k - d_c) & (CS - 1);
        cd_off = ((active ? cmd.prn : 0) * (2 * CS) + (r0 & 3) * (CS / 2) + (r0 >> 2)) * (int)sizeof(float);
    }
    sp4 cd[4], cdn[2];                                   // K-steps 4 i .. 4 i + 3 of the tile; next tile's first half
    cdn[0] = sp4{0.f, 0.f, 0.f, 0.f}; cdn[1] = cdn[0];
    auto load_half = [&](int half, sp4& r0, sp4& r1) {   // half-tile index from pos0, 8 entries
        r0 = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
            code_rs, cd_off, (half * 8) * (int)sizeof(float), 0));
        r1 = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
            code_rs, cd_off, (half * 8 + 4) * (int)sizeof(float), 0));
    };
    load_half(0, cd[0], cd[1]);

    const sp4 zero4 = sp4{0.f, 0.f, 0.f, 0.f};
    sp4 k1[MT], k2[MT], k3[MT];                          // the three products per M tile
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        k1[mt] = zero4; k2[mt] = zero4; k3[mt] = zero4;
        tot[mt][0] = zero4; tot[mt][1] = zero4;
    }
    // (re, im) of what the accumulators hold
    auto acc_re = [&](int mt) -> sp4 { return kStack ? k1[mt] - sp_swap32(k2[mt]) : k1[mt] - k3[mt]; };
    auto acc_im = [&](int mt) -> sp4 { return kStack ? k2[mt] + sp_swap32(k1[mt]) : k1[mt] + k2[mt]; };
    // the lanes of a channel close its lo sum where the boundary passes
    auto close_lo = [&]() {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            close(mt, tot[mt][0] + acc_re(mt), tot[mt][1] + acc_im(mt));
            tot[mt][0] = zero4; tot[mt][1] = zero4;
            k1[mt] = zero4; k2[mt] = zero4; k3[mt] = zero4;
        }
    };
    // the boundaries of the group's channels as scalars (channel c sits in lane c): the next one
    // is a dozen scalar compares, no cross-lane reduction in the loop
    int pbs[kSpCh];
#pragma unroll
    for (int c = 0; c < kSpCh; ++c) pbs[c] = __builtin_amdgcn_readlane(pb, c);
    auto next_boundary = [&](int after) {                            // first boundary position > after
        int v = kSpInf;
#pragma unroll
        for (int c = 0; c < kSpCh; ++c)
            if (pbs[c] > after && pbs[c] < rel0 + NSPANS * kSpTile) v = pbs[c] < v ? pbs[c] : v;
        return v;
    };
    int nb = next_boundary(rel0);        // (a boundary AT the range start needs no action: all hi)
    // lane = (row, k) reads the sample at position 4 s + k of its row: 8 bytes.  Pairs 8 .. 15 and
    // 24 .. 31 of a complex64 tile sit with their two positions swapped in LDS (sp_store_tile):
    // K-steps 4 .. 7 and 12 .. 15 read the other half of their 16-byte piece
    // (stacked form: row = lane % 8, component = (lane / 8) % 2: one float)
    const float* ap = kStack ? tl + (lane & 7) * kSpRowDw + 2 * k + ((lane >> 3) & 1)
                             : tl + (lane & 15) * kSpRowDw + 2 * k;
    const int swz_ofs = FMT == 0 ? ((k & 1) ? -2 : 2) : 0;

    // operands of one K-step (four positions): the lane's sample of both row halves and re + im
    struct Ops { sp2 x[MT]; float xs[MT]; };
    auto read_ops = [&](int s) {                                     // s: K-step index in the tile
        Ops o;
        const float* p = ap + (((s >> 2) & 1) ? swz_ofs : 0) + 8 * s;
        if (kStack) {
            o.xs[0] = *p;
            o.x[0] = sp2{0.f, 0.f};
            return o;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o.x[mt] = *reinterpret_cast<const sp2*>(p + mt * 16 * kSpRowDw);
        return o;
    };
    auto sums = [&](Ops& o) {
        if (kStack) return;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) o.xs[mt] = o.x[mt].x + o.x[mt].y;
    };
    auto mfma = [&](float a, float b, sp4 c) -> sp4 {
        if (DIAG & 1) { c[0] = fmaf(a, b, c[0]); return c; }
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    };
    // one K-step: six MFMAs (first: C = 0, an inline constant, no zeroing of the accumulators)
    auto kstep = [&](const Ops& o, float b1, float b2, float b3, auto first) {
        if (kStack) {                                    // two products on the stacked tile
            k1[0] = mfma(o.xs[0], b1, decltype(first)::value ? zero4 : k1[0]);
            k2[0] = mfma(o.xs[0], b2, decltype(first)::value ? zero4 : k2[0]);
            return;
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            k1[mt] = mfma(o.xs[mt], b1, decltype(first)::value ? zero4 : k1[mt]);
            k2[mt] = mfma(o.x[mt].x, b2, decltype(first)::value ? zero4 : k2[mt]);
            k3[mt] = mfma(o.x[mt].y, b3, decltype(first)::value ? zero4 : k3[mt]);
        }
    };

    // Tile `t` moves from its registers to LDS: first the entries that tile still needs and the first
    // ones of the tile after it, then the tile itself piece by piece (the reads of the previous
    // one are behind us: LDS serves a wave in order) while tile t + 1 is requested (unconditional:
    // behind the last tile of the range it is the first one of the wave's next range, or a
    // position past the block, which costs no memory traffic)
    auto enter_tile = [&](int t) {
        load_half(2 * t + 1, cd[2], cd[3]);
        if (t + 1 < kTiles) load_half(2 * t + 2, cdn[0], cdn[1]);    // (the next RANGE fetches its own)
        if (!(DIAG & 2)) {
            const bool more = t + 1 < kTiles;              // else: the first tile of the wave's next range
            sp_swap_tile<(DIAG & 8) ? 0 : 2, FMT, NC>(tl, lane, st, more ? blk : next_blk,
                                                      more ? pos0 + (t + 1) * kSpTile : next_pos);
        } else {
            sp_store_tile<FMT, NC>(tl, lane, st);
        }
    };
    enter_tile(0);
    after_first_swap();                // (outside the loop: what it holds in registers is dead from here on)

#pragma unroll 1
    for (int tix = 0; tix < kTiles; ++tix) {
        __builtin_amdgcn_sched_barrier(0);
        const int tpos = rel0 + tix * kSpTile;                       // tile start within the quarter

        // a K-step that holds boundaries is issued in pieces, B masked to the positions of the
        // piece, lo closed between them
        auto kstep_checked = [&](const Ops& o, int s, float b1, float b2, float b3) {
            const int P = tpos + 4 * s;                              // positions P .. P + 3
            int done = 0;                                            // positions of the K-step already summed
            while (nb <= P + 3) {
                const int t = nb - P;
                if (t > done) {
                    const bool in = k >= done && k < t;
                    kstep(o, in ? b1 : 0.f, in ? b2 : 0.f, in ? b3 : 0.f, std::false_type{});
                    done = t;
                }
                if (pb == nb) close_lo();
                nb = next_boundary(nb);
            }
            const bool in = k >= done;
            kstep(o, in ? b1 : 0.f, in ? b2 : 0.f, in ? b3 : 0.f, std::false_type{});
        };
        auto cd_pair = [&](int s2) {                                 // replica entries of K-steps s2, s2 + 1
            const sp4 v = cd[(s2 >> 2) & 3];
            return (s2 & 2) ? sp2{v.z, v.w} : sp2{v.x, v.y};
        };
        // half a tile = eight K-steps.  The operands of K-step s + 1 are requested from LDS before
        // the MFMAs of K-step s are issued; B is formed for two K-steps at a time.
        auto half_tile = [&](int hw, auto first) {
            const bool fast = nb >= tpos + (hw + 1) * (kSpTile / 2);
            if (!fast && decltype(first)::value) {     // (the accumulators are read where a boundary closes)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { k1[mt] = zero4; k2[mt] = zero4; k3[mt] = zero4; }
            }
            Ops o[2];
            o[0] = read_ops(8 * hw);
            sp2 b1, b2, b3 = sp2{0.f, 0.f};
            if (fast) {
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    if (s + 1 < 8) o[(s + 1) & 1] = read_ops(8 * hw + s + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    sums(o[s & 1]);
                    if (!(s & 1)) {
                        if (kStack) sp_b2(cd_pair(8 * hw + s), nk, u1, d1, u2, d2, b1, b2);
                        else sp_b3(cd_pair(8 * hw + s), nk, u1, d1, u2, d2, u3, d3, b1, b2, b3);
                    }
                    const float c1 = (s & 1) ? b1.y : b1.x, c2 = (s & 1) ? b2.y : b2.x, c3 = (s & 1) ? b3.y : b3.x;
                    if (s == 0) kstep(o[0], c1, c2, c3, first);
                    else kstep(o[s & 1], c1, c2, c3, std::false_type{});
                }
            } else {
                // (a boundary somewhere in this half tile: the same pipeline, every K-step asking
                // whether it is the one; rolled over the two groups of four K-steps -- eight inlined
                // copies of the checked path cost registers the combine step then reloads from
                // scratch, behind the prefetched rows)
#pragma unroll 1
                for (int g = 0; g < 2; ++g) {
                    const sp4 v = g ? cd[2 * hw + 1] : cd[2 * hw];          // (a select, not indexed registers)
                    const int s0 = 8 * hw + 4 * g;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j + 1 < 4 || g == 0) o[(j + 1) & 1] = read_ops(s0 + j + 1);
                        __builtin_amdgcn_sched_barrier(0);
                        sums(o[j & 1]);
                        if (!(j & 1)) {
                            const sp2 cdv = j == 0 ? sp2{v.x, v.y} : sp2{v.z, v.w};
                            if (kStack) sp_b2(cdv, nk, u1, d1, u2, d2, b1, b2);
                            else sp_b3(cdv, nk, u1, d1, u2, d2, u3, d3, b1, b2, b3);
                        }
                        const float c1 = (j & 1) ? b1.y : b1.x, c2 = (j & 1) ? b2.y : b2.x, c3 = (j & 1) ? b3.y : b3.x;
                        if (nb >= tpos + 4 * (s0 + j + 1)) kstep(o[j & 1], c1, c2, c3, std::false_type{});
                        else kstep_checked(o[j & 1], s0 + j, c1, c2, c3);
                    }
                }
            }
        };
        half_tile(0, std::true_type{});
        half_tile(1, std::false_type{});
        // a span ends (its accumulators restart from C = 0 at the next span)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            tot[mt][0] = tot[mt][0] + acc_re(mt);
            tot[mt][1] = tot[mt][1] + acc_im(mt);
        }
        cd[0] = cdn[0]; cd[1] = cdn[1];
        if (tix + 1 < kTiles) enter_tile(tix + 1);
    }
}

// probe builds only (tools/probe/span_prof.hip): the 100 MHz clock at the phase boundaries of every
// wave of the batch form, into a buffer nothing else reads
#ifdef GPSMI_SPAN_STAMPS
__device__ unsigned long long* g_span_stamps;
#define SPAN_STAMP(i) do { if (lane == 0 && iter < 4 && blockIdx.x < 1024) g_span_stamps[(((size_t)blockIdx.x * 4 + iter) * 4 + wave) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define SPAN_STAMP(i) do {} while (0)
#endif

// ---- the kernels.  16.3 KiB of LDS per wave (+ 10 KiB per workgroup of the batch form) = eight waves per CU.
//   batch form:        range = quarter; the four quarters of a block are the four waves of
//                      one workgroup (so the four 4 KiB pieces of every 16 KiB row are
//                      requested together).  The waves leave their hi / lo row sums in LDS;
//                      behind an (LDS-only) barrier all threads add the quarters in their fixed
//                      order and write partial[job][.] (3 KiB per block).  The workgroups are
//                      persistent (two per CU, each takes every gridDim-th block): the first
//                      rows of the next block are requested by the last tile of this one, its
//                      descriptors wait in LDS since the start of this one.
//                      Measured instead, per 1024-block launch: the wave that arrives last
//                      combines alone, the others exit without a barrier +6 us (the
//                      workgroup's LDS is held until that wave is done); one-wave workgroups
//                      that write 4 KiB records of raw sums +6 us, four-wave ones +9 us.
//   single-block form: range = span, 32 one-wave workgroups per (block, channel group), for
//                      a launch too small to fill the CUs with quarters (the closed loop).
//                      A wave writes the raw sums of its span (tot always, lo_fin by the
//                      lanes whose boundary lay inside it) to `rec`, 2048 floats per wave:
//                        rec[unit = b * ngroups + g][span][tot | lo_fin][M tile][re, im][v][lane = 16 (row group) + channel]
//                      and span_collect (called by the epilogue kernel) adds the spans of a
//                      block up in the order the batch form uses: same bits.
constexpr int kSpRecFloats = 2 * 16 * 64;              // one wave's record
constexpr int kSpLoOfs = 16 * 64;                      // lo_fin within it

// (FMT 1: iq holds raw uint16 samples, 2 bytes each, decoded on the way into LDS)
// workgroups per CU the registers admit: two M tiles = 250 VGPRs, one = 168
// (eight rows, stacked form: 122, four workgroups)
constexpr __host__ __device__ int sp_wg_per_cu(int nc) { return nc == 32 ? 2 : (nc == 16 ? 3 : 4); }
// a wave's tile area: NC rows (16 at least: the A operand's lanes address rows 0..15)
template <int NC> constexpr int kSpAreaFloats = (NC < 16 ? 16 : NC) * kSpRowDw;

template <int NSPANS, int WAVES, int FMT = 0, int DIAG = 0, int NC = 32>
__global__ __launch_bounds__(64 * WAVES, sp_wg_per_cu(NC)) void trk_span_kernel(
    const void* __restrict__ iq_v, const JobMid* __restrict__ mid,
    const float* __restrict__ code_eo, TrkParams P, int ngroups, int nblocks,
    float* __restrict__ rec, float2* __restrict__ partial) {
    constexpr int CS = kFftN;
    constexpr int MT = (NC + 15) / 16;
    constexpr int kRanges = CS / (NSPANS * kSpTile);    // per block: 4 quarters or 32 spans
    constexpr bool kWholeBlock = kRanges == WAVES;      // the workgroup holds all ranges of its block
    static_assert(kRanges % WAVES == 0, "the waves of a workgroup share a block");
    constexpr int kNowhere = CS * NC;                   // a tile position past the block: reads nothing
    constexpr size_t kBlkBytes = (size_t)CS * NC * (FMT == 0 ? sizeof(float2) : 2);
    const char* iq = static_cast<const char*>(iq_v);
    __shared__ __attribute__((aligned(16))) float lds[WAVES][kSpAreaFloats<NC>];
    __shared__ float4 ufac[kWholeBlock ? kSpCh * (NC + 1) : 1];      // (U[q], U[q+1]); .w = NaN: channel closed
    __shared__ __attribute__((aligned(16))) float lo_close[kWholeBlock ? kSpCh * NC * 2 : 2];   // [channel][row][re, im]
    __shared__ JobMid mids[2][kWholeBlock ? kSpCh : 1];             // descriptors of this unit and the next
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nunits = nblocks * ngroups;
    float* tl = &lds[wave][0];
    sp4 st[16], tot[MT][2];          // [M tile][re, im] of channel lane % 16
    bool all_lo;
    int pb;
    SpDesc cmd;

    if (!kWholeBlock) {
        // ---- single-block form: one span per wave, raw sums to `rec`
        const int widx = blockIdx.x * WAVES + wave;
        const int unit = widx / kRanges, range = widx % kRanges;
        if (unit >= nunits) return;
        const int g = unit % ngroups, b = unit / ngroups;
        const char* blk = iq + (size_t)b * kBlkBytes;
        const int pos0 = range * NSPANS * kSpTile;
        sp_load_tile<(DIAG & 8) ? 0 : 2, FMT, NC>(blk, pos0, lane, st);   // before anything that depends on the descriptors
        float* o = rec + ((size_t)unit * kRanges + range) * kSpRecFloats + lane;
        const int rel0 = pos0 & (kSpQuarter - 1);
        sp_wave_desc(mid + (size_t)b * P.nch + g * kSpCh, P.nch - g * kSpCh, lane, cmd);
        sp4 lo_fin[MT][2];               // what was summed below a boundary inside the span
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { lo_fin[mt][0] = sp4{0.f, 0.f, 0.f, 0.f}; lo_fin[mt][1] = lo_fin[mt][0]; }
        span_wave<NSPANS, FMT, DIAG, NC>(blk, blk, kNowhere, tl, cmd, code_eo, pos0, lane, st, tot, all_lo, pb,
                                         [&](int mt, sp4 re, sp4 im) { lo_fin[mt][0] = re; lo_fin[mt][1] = im; }, [] {});
        // rec[unit][span][tot | lo_fin][M tile][re, im][v][lane = 16 (row group) + channel]
        const bool closed = pb > rel0 && pb < rel0 + NSPANS * kSpTile;   // lo was closed in this range
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    o[((mt * 2 + p) * 4 + v) * 64] = tot[mt][p][v];
                    if (closed) o[kSpLoOfs + ((mt * 2 + p) * 4 + v) * 64] = lo_fin[mt][p][v];
                }
        return;
    }

    // ---- batch form: the workgroup takes units blockIdx.x, blockIdx.x + gridDim.x, ...
    // Between the tile loops of two units nothing may wait for a global load: the rows of the next
    // unit's first tile are requested by the last tile of this one, loads return in order, and a
    // wait for anything younger is a wait for those 16 KiB at the latency of a loaded memory system
    // (~6 us: span_prof -DGPSMI_SPAN_STAMPS showed 13 us per unit of set-up, barrier and combine
    // phases made of such waits -- the descriptor fetch, spill reloads, a register of the replica
    // entries reused as a temporary).  So: the descriptors of the next unit are fetched through the
    // scalar cache at the START of this unit and parked in LDS behind the first tile's requests; the sums
    // below a boundary go to LDS where the boundary passes (no registers held for them: no spills);
    // the two barriers of the combine step order LDS traffic only.
    constexpr int kSumFloats = kSpCh * NC * 2;          // [channel][row][re, im]
    static_assert(2 * kSumFloats <= kSpAreaFloats<NC>, "row sums must fit the tile area");
    constexpr int kItems = (kSpCh * (NC + 1) + 64 * WAVES - 1) / (64 * WAVES);
    int unit = blockIdx.x;
    if (unit >= nunits) return;
    __builtin_amdgcn_s_setprio(3);
    const int pos0 = wave * kSpQuarter;
    // a unit's descriptor of channel `cc` of the group (past the last channel: some valid one, to be
    // marked closed by the caller once it has arrived -- nothing here may wait for the load)
    auto mid_index = [&](int u, int cc) {
        const int g = u % ngroups, b = u / ngroups;
        return (size_t)b * P.nch + (g * kSpCh + cc < P.nch ? g * kSpCh + cc : 0);
    };
    auto mid_beyond = [&](int u, int cc) { return (u % ngroups) * kSpCh + cc >= P.nch; };
    // the descriptors of unit `u` into mids[slot], through the SCALAR cache, three channels per wave:
    // scalar loads have a counter of their own and wait for no row
    constexpr int kPerWave = kSpCh / WAVES;
    static_assert(kPerWave * WAVES == kSpCh, "the waves share the descriptor fetch evenly");
    auto park_mids = [&](int u, int slot, int lane_w) {
#pragma unroll
        for (int i = 0; i < kPerWave; ++i) {
            const int cc = wave * kPerWave + i;
            const JobMid& src = mid[__builtin_amdgcn_readfirstlane((int)mid_index(u, cc))];
            // (the fields as scalars: a uniform address alone did not keep the compiler from
            // moving the loads under the one-lane store below as vector loads)
            int f[5] = {src.delay_used, mid_beyond(u, cc) ? 0 : src.active, src.prn,
                        __float_as_int(src.om), __float_as_int(src.ph)};
#pragma unroll
            for (int w = 0; w < 5; ++w) asm volatile("" : "+s"(f[w]));
            if (lane_w == 0) {
                JobMid m{};
                m.delay_used = f[0]; m.active = f[1]; m.prn = f[2];
                m.om = __int_as_float(f[3]); m.ph = __int_as_float(f[4]);
                mids[slot][cc] = m;
            }
        }
    };
    {
        sp_load_tile<(DIAG & 8) ? 0 : 2, FMT, NC>(iq + (size_t)(unit / ngroups) * kBlkBytes, pos0, lane, st);
        for (int i = threadIdx.x; i < kSumFloats; i += 64 * WAVES) lo_close[i] = 0.f;
        park_mids(unit, 0, lane);
        __syncthreads();
    }
#pragma unroll 1
    for (int iter = 0;; ++iter) {
        const int g = unit % ngroups, b = unit / ngroups;
        const int cur = iter & 1;
        SPAN_STAMP(0);
        // (the lane-derived constants of a wave's set-up are recomputed per unit instead of
        // being carried through the tile loop in registers the loop needs)
        int lane_u = lane;
        asm volatile("" : "+v"(lane_u));
        const int next = unit + (int)gridDim.x;
        const bool has_next = next < nunits;
        // the row factors U[q], U[q+1] of the combine step (double arithmetic) are formed now and
        // wait in LDS: nothing of this stays in registers across the tile loop or is left for the
        // tail of the block
#pragma unroll
        for (int e = 0; e < kItems; ++e) {
            const int item = (int)threadIdx.x + 64 * WAVES * e;
            if (item >= kSpCh * (NC + 1)) continue;
            const JobMid& m2 = mids[cur][item / (NC + 1)];
            const int q = item % (NC + 1) - 1;
            const float2 a0 = sp_row_factor(m2.om, q), a1 = sp_row_factor(m2.om, q + 1);
            ufac[item] = make_float4(a0.x, a0.y, a1.x, m2.active ? a1.y : __builtin_nanf(""));
        }
        {
            const int c = lane_u & 15;
            const JobMid& m = mids[cur][c < kSpCh ? c : 0];
            cmd.delay_used = m.delay_used; cmd.active = c < kSpCh && m.active; cmd.prn = m.prn; cmd.om = m.om; cmd.ph = m.ph;
        }
        SPAN_STAMP(1);
        // Everything outside the tile loop runs at raised priority: a wave in its set-up or combine
        // phase shares its SIMD with a wave that issues one fp32 MFMA after the other
        __builtin_amdgcn_s_setprio(0);
        const char* blk = iq + (size_t)b * kBlkBytes;
        const char* next_blk = has_next ? iq + (size_t)(next / ngroups) * kBlkBytes : blk;
        span_wave<NSPANS, FMT, DIAG, NC>(
            blk, next_blk, has_next ? pos0 : kNowhere, tl, cmd, code_eo, pos0, lane_u, st, tot, all_lo, pb,
            [&](int mt, sp4 re, sp4 im) {              // below a boundary: rows 16 mt + 4 (lane / 16) + v of the channel
                if (NC < 16 && 4 * (lane_u >> 4) >= NC) return;        // (rows the block does not have)
                float* dst = lo_close + ((lane_u & 15) * NC + 16 * mt + 4 * (lane_u >> 4)) * 2;
#pragma unroll
                for (int v = 0; v < 4; ++v) *reinterpret_cast<sp2*>(dst + 2 * v) = sp2{re[v], im[v]};
            },
            [&] { if (has_next) park_mids(next, cur ^ 1, lane_u); });   // behind the first tile's requests
        __builtin_amdgcn_s_setprio(3);
        SPAN_STAMP(2);
        // the hi / lo sums of every row of the quarter into LDS (the wave's own tile area),
        // D[i = 4 (lane / 16) + v][j = lane % 16]; a quarter that lies below the delay altogether
        // is lo, one that holds the boundary has left its lo part in lo_close
        if (!(DIAG & 4)) {
            float* hi = tl;
            float* lo = tl + kSumFloats;
            const int c = lane & 15;
            if (c < kSpCh && (NC >= 16 || 4 * (lane >> 4) < NC)) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int row = 16 * mt + 4 * (lane >> 4) + v;
                        const sp2 t = sp2{tot[mt][0][v], tot[mt][1][v]}, z = sp2{0.f, 0.f};
                        *reinterpret_cast<sp2*>(hi + (c * NC + row) * 2) = all_lo ? z : t;
                        *reinterpret_cast<sp2*>(lo + (c * NC + row) * 2) = all_lo ? t : z;
                    }
            }
            // behind one barrier all threads add the quarters in their fixed order (the lo part of
            // the quarter with the boundary comes after the quarters below it and before zeros: at
            // the end), apply U and write partial[q + 1], q = -1 .. 31
            SPAN_STAMP(3);
            lds_barrier();
            SPAN_STAMP(4);
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
                    const float* hi_w = &lds[w][0];
                    const float* lo_w = hi_w + kSumFloats;
                    if (q >= 0) { hx += hi_w[(cc * NC + q) * 2]; hy += hi_w[(cc * NC + q) * 2 + 1]; }
                    if (q + 1 < NC) { lx += lo_w[(cc * NC + q + 1) * 2]; ly += lo_w[(cc * NC + q + 1) * 2 + 1]; }
                }
                if (q + 1 < NC) {
                    sp2* lc = reinterpret_cast<sp2*>(lo_close + (cc * NC + q + 1) * 2);
                    const sp2 v = *lc;
                    lx += v.x; ly += v.y;
                    *lc = sp2{0.f, 0.f};                    // (this thread is its only reader: free for the next unit)
                }
                partial[((size_t)b * P.nch + g * kSpCh + cc) * (NC + 1) + o] =
                    sp_window(hx, hy, lx, ly, make_float2(uf.x, uf.y), make_float2(uf.z, uf.w));
            }
        }
        if (DIAG & 4) {                                  // (probe: keep the sums alive without the combine step)
            if (tot[0][0][0] + tot[0][1][1] + tot[MT - 1][0][2] + tot[MT - 1][1][3] == 123.456f)
                partial[(size_t)unit * 64 + lane] = make_float2(tot[0][0][0], tot[MT - 1][1][1]);
        }
        SPAN_STAMP(5);
        if (!has_next) break;
        if (!(DIAG & 4)) lds_barrier();              // the tile areas, ufac and mids[cur] are free again
        SPAN_STAMP(6);
        unit = next;
    }
}

// The sums of quarter Q of channel `cidx` of block `b`, for lane = (row, re/im): what the
// quarter contributes to the hi and to the lo side of the delay `d`.  NSPANS = spans per record,
// as the kernel that wrote them.  (Loads first, then the adds in the fixed order.)
template <int NSPANS, int NC = 32>
__device__ __forceinline__ void span_collect_quarter(const float* __restrict__ rec, int ngroups, int b,
                                                     int cidx, int d, int lane, int Q, float& w_hi,
                                                     float& w_lo) {
    if (lane >= 2 * NC) {                 // (row, re/im) pairs the block does not have
        w_hi = w_lo = 0.f;
        return;
    }
    constexpr int kPerQ = kSpQuarter / (NSPANS * kSpTile);   // records per quarter: 1 or 8
    constexpr int kLen = NSPANS * kSpTile;                   // positions per record
    const int g = cidx / kSpCh, cc = cidx % kSpCh;
    const int r = lane >> 1, part = lane & 1;
    const int mt = r >> 4, rg = (r & 15) >> 2, v = r & 3;
    const int n = part, j = cc;                        // [M tile][re, im][v][16 (row group) + channel]
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
template <int NC = 32>
__device__ __forceinline__ void span_windows(const float* hi, const float* lo, float om, int lane, float2* S) {
    if (lane <= NC) {
        const int q = lane - 1;
        const float hx = q >= 0 ? hi[2 * q] : 0.f, hy = q >= 0 ? hi[2 * q + 1] : 0.f;
        const float lx = q + 1 < NC ? lo[2 * (q + 1)] : 0.f, ly = q + 1 < NC ? lo[2 * (q + 1) + 1] : 0.f;
        S[lane] = sp_window(hx, hy, lx, ly, sp_row_factor(om, q), sp_row_factor(om, q + 1));
    }
}

// One wave per job: the four quarters in turn, then the windows.
template <int NSPANS, int NC = 32>
__device__ __forceinline__ void span_collect(const float* __restrict__ rec, int ngroups, int b, int cidx,
                                             int d, float om, int lane, float* hi, float* lo, float2* S) {
    float h = 0.f, l = 0.f;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        float wh, wl;
        span_collect_quarter<NSPANS, NC>(rec, ngroups, b, cidx, d, lane, Q, wh, wl);
        h += wh;
        l += wl;
    }
    hi[lane] = h;
    lo[lane] = l;
    __builtin_amdgcn_wave_barrier();
    span_windows<NC>(hi, lo, om, lane, S);
    __builtin_amdgcn_wave_barrier();
}

}  // namespace gpsmi
