// The tracking correlator for BASELINE configs[4]: CODE_SAMPLES = 16368, N_CYC = 8
// (16.368 Msps, 8-ms blocks).  Same mathematics as gpsmi_trk_span.h -- prompt
// correlate-and-dump, y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed
// per code-period window (reference src/gpslib.py:1400-1420) -- on the matrix pipe, re-cut for
// eight rows:
//
//   v_mfma_f32_16x16x4_f32, D[16 x 16] += A[16 x 4] B[4 x 16] for one PAIR of positions
//   m = 2 q + pi:
//     M = (code period r = 0..7, component o of the RESULT),
//     K = (position parity pi, component kappa of the replica-times-carrier factor B_c),
//     N = channel c (12 of 16 columns; ONE N tile holds all channels of a group),
//       A[(r, re)][(pi, re)] =  x_re,   A[(r, re)][(pi, im)] = -x_im,
//       A[(r, im)][(pi, re)] =  x_im,   A[(r, im)][(pi, im)] =  x_re,      x = x[r][2 q + pi],
//       B[(pi, kappa)][c]    =  component kappa of  replica_c[(m - d_c) mod CS] exp(-j theta_c(m)).
//   One MFMA per pair of positions does the 8 x 12 complex multiply-accumulates of both: the same
//   two matrix-pipe cycles per row and position as the four-product form of the 32-row kernel had
//   (its three-product form needs M = rows alone: not with eight rows), every row of M in use.  (The vector kernel this replaces rebuilt B for every group of
//   six channels and amortised it over 8 rows instead of 32: 0.29 ms per 512 MiB, 23 % of the HBM
//   peak, bound by VALU issue.)
//   A lane holds ONE real number of B per pair -- component kappa at parity pi for its channel --
//   for two consecutive pairs in a packed register, advanced four positions at a time by the
//   coupled recurrence dl -= k u, u += dl (k = 4 sin^2(2 phi)), re-seeded with the exact phasor
//   every fourth tile (one step behind the tile start, so every step advances first and uses then);
//   the sign of A's (re, im) entries is one packed multiply per two pairs.
//
// Order of the float32 sums = a property of the data: a block is cut into 31 RANGES of 528
// positions (11 tiles of 48: 16368 = 31 * 11 * 48; with 11 ranges of 31 tiles a 512-block batch is
// 22 waves per CU against the 16 the registers admit, a round and a third: 0.150 ms, with 31 ranges
// it is 62 waves in just under four rounds: 0.143 ms); a range is summed position by position (two
// interleaved accumulators: even and odd pairs) by one wave, which writes its raw sums
// (`tot`, and `lo_fin` = what lay below the delay when the boundary fell inside the range);
// trk_span8_collect_kernel adds the ranges of a block in ascending order and forms the windows.
// The closed loop and replay run the same decomposition: same bits (tests/test_gpu_trk.py).
//
// Measured and not kept: two tiles of look-ahead per wave instead of one (12 more registers):
// 0.138 against 0.135 ms -- the kernel waits for its SIMDs (per step of two pairs: two MFMAs = 64
// cycles, ~6 VALU instructions = ~27, three LDS reads; per tile another ~60 instructions of staging,
// per range ~300 of set-up), not for its rows.
//
// Window q of the reference = positions m >= d of row q ("hi") plus m < d of row q + 1 ("lo").
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

constexpr int kS8Cs = 16368, kS8Rows = 8;
constexpr int kS8Tile = 48;                          // positions per tile: 384 contiguous bytes per row
constexpr int kS8TilesPerRange = 11, kS8Ranges = 31; // 31 * 11 * 48 = 16368
constexpr int kS8RangeLen = kS8Tile * kS8TilesPerRange;
constexpr int kS8RowDw = 2 * kS8Tile + 4;            // dwords per tile row (100: lane = (row, k) reads hit 32 banks)
constexpr int kS8TileFloats = kS8Rows * kS8RowDw;
constexpr int kS8CodePitch = kS8Tile / 2 + 4;        // floats per (channel, parity) row of a tile's replica window
constexpr int kS8CodeFloats = kSpCh * 2 * kS8CodePitch;
constexpr int kS8WaveFloats = kS8TileFloats + kS8CodeFloats;
constexpr int kS8RecFloats = 2 * 4 * 64;             // one wave's record: [tot | lo_fin][acc register][lane]
constexpr int kS8Reseed = 4;                         // tiles between two exact phasors
static_assert(kS8Ranges * kS8RangeLen == kS8Cs, "ranges must tile the code period");

// One wave = one range of one (block, channel group).  iq: complex64 blocks of 8 x 16368; code_eo:
// [PRN][2][CS] the replica split by index parity, each plane stored twice (no wrap inside a tile).
__global__ __launch_bounds__(256) void trk_span8_kernel(
    const float2* __restrict__ iq, const JobMid* __restrict__ mid, const float* __restrict__ code_eo,
    TrkParams P, int ngroups, int nblocks, float* __restrict__ rec) {
    constexpr int CS = kS8Cs;
    __shared__ __attribute__((aligned(16))) float lds[4][kS8WaveFloats];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int widx = blockIdx.x * 4 + wave;
    const int unit = widx / kS8Ranges, range = widx % kS8Ranges;
    if (unit >= nblocks * ngroups) return;
    const int g = unit % ngroups, b = unit / ngroups;
    float* tl = &lds[wave][0];
    float* cd = tl + kS8TileFloats;
    const int pos0 = range * kS8RangeLen;

    // ---- lane roles
    const int m16 = lane & 15, kk = lane >> 4;          // A: row of M, column of K
    const int r_a = m16 >> 1, o_a = m16 & 1, pi = kk >> 1, kap = kk & 1;
    const int c_b = lane & 15;                           // B / D: the lane's channel
    const float a_sign = (o_a == 0 && kap == 1) ? -1.f : 1.f;
    const float* ap = tl + r_a * kS8RowDw + 2 * pi + (o_a ^ kap);     // + 4 q per pair
    const float* cp = cd + ((c_b < kSpCh ? c_b : 0) * 2 + pi) * kS8CodePitch;

    // ---- tile staging: 3 x b128 per lane; flat index L = lane + 64 i -> row L / 24, 16-byte piece L % 24
    const char* blk = reinterpret_cast<const char*>(iq) + (size_t)b * ((size_t)CS * kS8Rows * sizeof(float2));
    const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(blk), 0, CS * kS8Rows * (int)sizeof(float2), kRsrcFlags);
    int ld_off[3], st_off[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int L = lane + 64 * i, row = L / 24, piece = L % 24;
        ld_off[i] = (row * CS + 2 * piece) * (int)sizeof(float2);
        st_off[i] = row * kS8RowDw + 4 * piece;
    }
    sp4 st[3];
    auto load_tile = [&](int tix) {
        const int tb = (pos0 + tix * kS8Tile) * (int)sizeof(float2);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off[i], tix < kS8TilesPerRange ? tb : CS * kS8Rows * (int)sizeof(float2), 2));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 3; ++i) *reinterpret_cast<sp4*>(tl + st_off[i]) = st[i];
    };
    load_tile(0);                                        // before anything that depends on the descriptors

    // ---- descriptors: the lane's channel (B / D role) and the channel it stages the replica for
    const JobMid* midrow = mid + (size_t)b * P.nch + g * kSpCh;
    const int nch_g = P.nch - g * kSpCh;
    const SpDesc md = sp_desc(midrow, c_b, c_b < kSpCh && c_b < nch_g);
    const int sc = lane >> 2, spl = (lane >> 1) & 1, shf = lane & 1;   // lanes < 48: channel, parity, half
    const SpDesc smd = sp_desc(midrow, sc, lane < 4 * kSpCh && sc < nch_g);

    // replica window of a tile: rolled index r = (m - d) mod CS of position m = 2 q + pi has parity
    // e = (pi - d) & 1 and half index (r - e) / 2, which advances by one per pair: 24 consecutive
    // entries of plane e (doubled: never wraps).  A closed channel reads PRN slot 0 (zeros).
    const __amdgpu_buffer_rsrc_t code_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(code_eo), 0, (GPSMI_MAX_PRN + 1) * 2 * CS * (int)sizeof(float), kRsrcFlags);
    const int s_d = smd.active ? smd.delay_used : 0;
    const int s_e = (spl - s_d) & 1;
    const int s_plane = ((smd.active ? smd.prn : 0) * 2 + s_e) * CS;
    float* sdst = cd + ((lane < 4 * kSpCh ? sc : 0) * 2 + spl) * kS8CodePitch + 12 * shf;
    sp4 cst[3];
    auto load_code = [&](int tix) {
        int r0 = pos0 + tix * kS8Tile + spl - s_d - s_e;                // even, in (-CS, CS)
        r0 += r0 < 0 ? CS : 0;
        const int off = (s_plane + (r0 >> 1) + 12 * shf) * (int)sizeof(float);
#pragma unroll
        for (int i = 0; i < 3; ++i)
            cst[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                code_rs, tix < kS8TilesPerRange ? off : 0, 4 * i * (int)sizeof(float), 0));
    };
    auto store_code = [&]() {
        if (lane < 4 * kSpCh) {
#pragma unroll
            for (int i = 0; i < 3; ++i) *reinterpret_cast<sp4*>(sdst + 4 * i) = cst[i];
        }
    };
    load_code(0);

    // ---- the lane's carrier component: u(m) for m = m0 + pi and m0 + pi + 2 (two pairs), exact
    const bool active = md.active;
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float inv_fs = 1.0f / (1000.0f * (float)CS);
    const float f_eff = active ? (float)((double)md.om * inv_2pi) : 0.f;
    const float ph_rev = active ? md.ph * (float)inv_2pi : 0.f;
    const float2 w2 = sp_phasor_rev(2.0f * (f_eff * inv_fs));          // exp(-j 2 phi)
    const float sh = -w2.y, chh = w2.x;
    const sp2 nk = sp2{-4.0f * sh * sh, -4.0f * sh * sh};               // -4 sin^2(2 phi): a step of four positions
    const float2 omw4 = make_float2(2.0f * sh * sh, -2.0f * sh * chh);  // 1 - exp(+j 4 phi)
    sp2 u2, dl2;
    // (seeded one step -- four positions -- BEHIND the tile start: every step of the loop advances
    // first and uses then, the first one included)
    auto seed = [&](int m0) {
        const float2 z0 = sp_phasor_rev(fmaf(f_eff, (float)(m0 - 4 + pi + 1) * inv_fs, ph_rev));
        const float2 z1 = sp_cmul(z0, w2);
        const float2 dz0 = sp_cmul(z0, omw4), dz1 = sp_cmul(z1, omw4);  // z(m) - z(m - 4)
        const float a0 = kap ? z0.y : z0.x, a1 = kap ? z1.y : z1.x;
        const float e0 = kap ? dz0.y : dz0.x, e1 = kap ? dz1.y : dz1.x;
        u2 = sp2{active ? a0 : 0.f, active ? a1 : 0.f};
        dl2 = sp2{active ? e0 : 0.f, active ? e1 : 0.f};
    };

    // ---- boundary of the lane's channel inside this range (relative position), else none
    const int d = active ? md.delay_used : 0;
    const int rel = d - pos0;
    int pb = (active && rel > 0 && rel < kS8RangeLen) ? rel : kSpInf;
    int nbs = sp_wave_min(pb == kSpInf ? kSpInf : pb >> 2);            // first step (4 positions) with a boundary

    const sp4 zero4 = sp4{0.f, 0.f, 0.f, 0.f};
    sp4 acc0 = zero4, acc1 = zero4, lo_fin = zero4;
    auto close_lo = [&]() { lo_fin = acc0 + acc1; acc0 = zero4; acc1 = zero4; };

#pragma unroll 1
    for (int tix = 0; tix < kS8TilesPerRange; ++tix) {
        store_code();
        store_tile();
        load_code(tix + 1);
        load_tile(tix + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (tix % kS8Reseed == 0) seed(pos0 + tix * kS8Tile);
        const int step0 = tix * (kS8Tile / 4);                          // first step of the tile
        constexpr int kSteps = kS8Tile / 4;                             // two pairs = four positions per step
        if (nbs < step0 || nbs >= step0 + kSteps) {
            // ---- no window boundary in this tile (30 of 31 tiles at least): straight-line code, every
            // LDS read of the tile requested before the first MFMA
            // (a third of a tile at a time: 16 registers of operands, six waves per SIMD; the reads are
            // pinned above the arithmetic, hipcc otherwise sinks each to its use and waits there)
            constexpr int kChunk = 4;
#pragma unroll
            for (int h2 = 0; h2 < kSteps / kChunk; ++h2) {
                sp2 a2[kChunk], c2[kChunk];
#pragma unroll
                for (int s = 0; s < kChunk; ++s) {
                    a2[s] = sp2{ap[8 * (s + kChunk * h2)], ap[8 * (s + kChunk * h2) + 4]};
                    c2[s] = *reinterpret_cast<const sp2*>(cp + 2 * (s + kChunk * h2));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < kChunk; ++s) {
                    dl2 = __builtin_elementwise_fma(nk, u2, dl2);
                    u2 = u2 + dl2;
                    const sp2 av = a2[s] * a_sign, b2 = c2[s] * u2;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b2.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b2.y, acc1, 0, 0, 0);
                }
            }
        } else {
            // ---- some channel's window boundary lies in this tile: the lanes of that channel close
            // their lo sum where it passes; a pair that straddles an odd boundary is issued twice
            // with B masked
#pragma unroll 1
            for (int s = 0; s < kSteps; ++s) {
                sp2 a2 = sp2{ap[8 * s], ap[8 * s + 4]};
                const sp2 c2 = *reinterpret_cast<const sp2*>(cp + 2 * s);
                a2 = a2 * a_sign;
                dl2 = __builtin_elementwise_fma(nk, u2, dl2);
                u2 = u2 + dl2;
                const sp2 b2 = c2 * u2;
                if (step0 + s == nbs) {
                    const int Pq = 4 * (step0 + s);                    // relative position of the step
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int Pi = Pq + 2 * i;
                        float bv = i ? b2.y : b2.x;
                        const float av = i ? a2.y : a2.x;
                        if (pb == Pi) close_lo();
                        const bool odd = pb == Pi + 1;
                        if (__builtin_amdgcn_ballot_w64(odd) != 0) {
                            const float b_first = odd ? (pi == 0 ? bv : 0.f) : bv;
                            if (i) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_first, acc1, 0, 0, 0);
                            else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b_first, acc0, 0, 0, 0);
                            if (odd) close_lo();
                            bv = odd ? (pi == 1 ? bv : 0.f) : 0.f;      // then position P + 1 of those lanes
                        }
                        if (i) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc0, 0, 0, 0);
                    }
                    if (pb >= Pq && pb < Pq + 4) pb = kSpInf;           // this lane's boundary is behind it
                    nbs = sp_wave_min(pb == kSpInf ? kSpInf : pb >> 2);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.x, b2.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a2.y, b2.y, acc1, 0, 0, 0);
                }
            }
        }
    }
    // ---- the record: D[row = 4 (lane / 16) + v][channel = lane % 16]
    const sp4 tot = acc0 + acc1;
    float* o = rec + (size_t)widx * kS8RecFloats + lane;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        o[v * 64] = tot[v];
        o[(4 + v) * 64] = lo_fin[v];
    }
}

// One wave: the ranges of a job's block in ascending order -> hi / lo row sums -> the windows
// S[q + 1] = U[q] hi[q] + U[q+1] lo[q+1], q = -1 .. 7 (what the other correlators write to
// partial[job][.]).  s_hi / s_lo: 16 floats of LDS each, S: 9 entries.
__device__ __forceinline__ void span8_collect(const float* __restrict__ rec, int ngroups, int b, int cidx,
                                              int d, float om, int lane, float* s_hi, float* s_lo, float2* S) {
    const int g = cidx / kSpCh, c = cidx % kSpCh;
    if (lane < 16) {                                      // lane = row of M = (code period, component)
        const int src_lane = (lane >> 2) * 16 + c, v = lane & 3;
        const float* base = rec + ((size_t)(b * ngroups + g) * kS8Ranges) * kS8RecFloats + src_lane;
        float t[kS8Ranges], lf = 0.f;
#pragma unroll
        for (int w = 0; w < kS8Ranges; ++w) t[w] = base[(size_t)w * kS8RecFloats + v * 64];
        const int wb = d / kS8RangeLen;                   // the range the boundary lies in (or starts)
        const bool inside = d % kS8RangeLen != 0;
        if (inside) lf = base[(size_t)wb * kS8RecFloats + (4 + v) * 64];
        float hi = 0.f, lo = 0.f;
#pragma unroll
        for (int w = 0; w < kS8Ranges; ++w) {
            if (w < wb) lo += t[w];
            else if (w == wb && inside) { lo += lf; hi += t[w]; }
            else hi += t[w];
        }
        s_hi[lane] = hi;
        s_lo[lane] = lo;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane <= kS8Rows) {
        const int q = lane - 1;
        const float hx = q >= 0 ? s_hi[2 * q] : 0.f, hy = q >= 0 ? s_hi[2 * q + 1] : 0.f;
        const float lx = q + 1 < kS8Rows ? s_lo[2 * (q + 1)] : 0.f, ly = q + 1 < kS8Rows ? s_lo[2 * (q + 1) + 1] : 0.f;
        S[lane] = sp_window(hx, hy, lx, ly, sp_row_factor(om, q), sp_row_factor(om, q + 1));
    }
    __builtin_amdgcn_wave_barrier();
}

}  // namespace gpsmi
