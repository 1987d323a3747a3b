// The tracking correlator, span form (default for CS = 2048, N_CYC = 32).
//
// Same mathematics as gpsmi_trk_stream_mfma.h -- prompt correlate-and-dump of a 32-ms
// block, y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed per
// code-period window (reference src/gpslib.py:1400-1420) -- re-cut so that
//   (1) every load instruction reads 512 consecutive bytes of a row (tiles of 16 rows x 64
//       positions; 256-byte segments top out at 5.2 TB/s, 512-byte ones reach 5.7-6.0,
//       tools/probe/tile_read.hip),
//   (2) a wave needs < 128 VGPRs and 10 KiB of LDS, so 16 waves live on a CU and a batch
//       of 1024 blocks is two full rounds of workgroups with no one-wave-per-SIMD tail,
//   (3) the order of the float32 sums is defined by the DATA, not by the launch: a block
//       is cut into 16 SPANS of 128 positions; a span is summed position by position, the
//       spans of a quarter (512 positions) are added in order, then the four quarters.
//       Any assignment of spans to waves that follows this order gives the same bits:
//       the batch form gives a wave a quarter (four spans), the single-block form gives
//       every span its own wave (32 waves on 8 workgroups instead of 4 waves on one CU).
//
//   v_mfma_f32_16x16x4_f32, D[16 x 16] += A[16 x 4] B[4 x 16] for one PAIR of positions:
//     M = 16 code periods (rows 16 h .. 16 h + 15 of the block: a wave owns one row half h),
//     K = (position parity pi, re/im kappa) of the samples at positions 2 q + pi,
//     N = (channel, re/im) for 8 channels; two N tiles = up to 16 channels (12 are staged):
//       A[r][(pi, kappa)]      = x[r][2 q + pi].{re, im}
//       B[(pi, 0)][(c, re)] =  B_re,  B[(pi, 1)][(c, re)] = -B_im,
//       B[(pi, 0)][(c, im)] =  B_im,  B[(pi, 1)][(c, im)] =  B_re,
//       B_c(m) = replica_c[(m - d_c) mod 2048] * exp(-j theta_c(m)),  m = 2 q + pi.
//   A lane holds one real component u of the carrier phasor for its two channels (one per
//   N tile) at positions m, m + 2 in a packed register pair and advances both by four
//   positions with the coupled recurrence dl -= kappa u, u += dl (kappa = 4 sin^2(2 phi));
//   the exact phasor is taken at the start of every quarter only (128 steps, as far as the
//   recurrence stays accurate); a wave that starts at a later span of the quarter runs the
//   same recurrence from the quarter start without the MFMAs, so it forms the same bits.  The replica samples come from a table split by index parity (a lane's
//   positions all have the same parity), staged through LDS 32 positions at a time.
//
// Window q of the reference = positions m >= d of row q ("hi") plus m < d of row q+1
// ("lo").  Where the boundary d_c falls inside a wave's range the four lanes of channel c
// close the lo sum (lo = sum so far) and restart; a pair of positions that straddles an odd
// d_c is issued twice with B masked.  Tiles without a boundary run 64 MFMAs straight.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace gpsmi {

constexpr int kSpCh = 12;                      // channels per workgroup (staged through LDS)
constexpr int kSpTile = 64;                    // positions per tile
constexpr int kSpSpan = 128;                   // positions per span (unit of the summation order)
constexpr int kSpQuarter = 512;                // positions per quarter (four spans)
constexpr int kSpRowDw = 2 * kSpTile + 2;      // dwords per tile row: lane = (row, k) reads are conflict-free
constexpr int kSpTileFloats = 16 * kSpRowDw;
constexpr int kSpWin = 32;                     // positions per replica window
constexpr int kSpCodePitch = kSpWin / 2 + 4;   // floats per (channel, parity) row of a window
constexpr int kSpCodeFloats = kSpCh * 2 * kSpCodePitch;
constexpr int kSpWaveFloats = kSpTileFloats + kSpCodeFloats;     // 2560 floats = 10 KiB per wave
constexpr int kSpInf = 1 << 20;
constexpr int kSpParkFloats = 8 * 64;                // parked lo sums of one wave

#ifdef GPSMI_SP_PROF       // tools/probe/span_prof.hip only: per-wave cycle stamps
__device__ unsigned long long* g_sp_prof;
__device__ unsigned long long g_sp_wait[1];
#define SP_STAMP() clock64()
#else
#define SP_STAMP() 0ull
#endif

typedef float sp4 __attribute__((ext_vector_type(4)));
typedef float sp2 __attribute__((ext_vector_type(2)));

// Per-lane constants of one N tile (one channel per lane and tile)
struct SpChan {
    bool active;
    int pb;            // boundary position relative to the quarter start, kSpInf if outside (0, 512)
    bool all_lo;       // the whole quarter lies below the delay
    const float* crow; // LDS: this lane's replica row of the current window
};

// dl + k u on a packed pair, k = the low (tile 0) or the high (tile 1) half of `nk`
__device__ __forceinline__ sp2 sp_rec_lo(sp2 nk, sp2 u, sp2 dl) {
    sp2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(nk), "v"(u), "v"(dl));
    return r;
}
__device__ __forceinline__ sp2 sp_rec_hi(sp2 nk, sp2 u, sp2 dl) {
    sp2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(nk), "v"(u), "v"(dl));
    return r;
}

__device__ __forceinline__ int sp_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

// One wave: NSPANS consecutive spans starting at position `pos0` (a multiple of 128) of row
// half `h` of a block.  Returns the sums of the range in (tot, lo_fin): lo_fin = what was
// summed below the delay when the boundary lies inside the range, tot = the rest.
// (DIAG, probes only: 1 no MFMAs, 2 no row loads after the first tile, 4 no barrier / combine)
template <int NSPANS, int DIAG = 0>
__device__ __forceinline__ void span_wave(const float2* __restrict__ blk, float* tl, float* cd,
                                          const JobMid* __restrict__ midrow, int nch_g,
                                          const float* __restrict__ code_eo, float* park, int h,
                                          int pos0, int lane, sp4 (&tot)[2], sp4 (&lo_fin)[2],
                                          bool (&all_lo)[2]) {
    constexpr int CS = kFftN, NC = 32;
    const int j = lane & 15, k = lane >> 4, pi = k >> 1, kap = k & 1, part = j & 1;
    const int q0 = pos0 & ~(kSpQuarter - 1);                      // start of the quarter
    constexpr int kTiles = NSPANS * (kSpSpan / kSpTile);

    // ---- tile staging: 8 x b128 per lane; instruction i covers rows 2 i, 2 i + 1 (lane / 32)
    // and 512 bytes of each.  Global address = scalar base + one lane offset (buffer loads).
    const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(blk), 0, CS * NC * (int)sizeof(float2), kMfRsrcFlags);
    const int ld_off = ((lane >> 5) * CS + 2 * (lane & 31)) * (int)sizeof(float2);
    const int row0 = 16 * h;
    float* st_dst = tl + (lane >> 5) * kSpRowDw + 4 * (lane & 31);
    sp4 st[8];
    auto load_tile = [&](int tix) {
        const int tb = (row0 * CS + pos0 + tix * kSpTile) * (int)sizeof(float2);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st[i] = __builtin_bit_cast(sp4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (2 * CS * (int)sizeof(float2)), 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < 8; ++i) {                  // rows are 8-byte aligned: two b64 writes
            *reinterpret_cast<sp2*>(st_dst + i * 2 * kSpRowDw) = sp2{st[i].x, st[i].y};
            *reinterpret_cast<sp2*>(st_dst + i * 2 * kSpRowDw + 2) = sp2{st[i].z, st[i].w};
        }
    };
    load_tile(0);                     // requested before anything that depends on the descriptors

    // ---- lane roles: two channels (one per N tile); the recurrence of both, seeded with the
    // exact phasor of the lane's first two positions of the QUARTER
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float inv_fs = 1.0f / (1000.0f * (float)CS);
    const float sx = ((kap == 0) == (part == 0)) ? 1.f : 0.f;          // (0,re) and (1,im): +z.x
    const float sy = (sx != 0.f) ? 0.f : (part == 0 ? -1.f : 1.f);     // (1,re): -z.y, (0,im): +z.y
    SpChan ch[2];
    sp2 u2[2], dl2[2], nk01;
    float nkv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int c = 8 * n + (j >> 1);
        const bool col = c < kSpCh && c < nch_g;
        const JobMid md = midrow[col ? c : 0];
        SpChan& s = ch[n];
        s.active = col && md.active;
        const float f_eff = s.active ? (float)((double)md.om * inv_2pi) : 0.f;
        const float ph_rev = s.active ? md.ph * (float)inv_2pi : 0.f;
        const float2 w2 = phasor_rev(2.0f * (f_eff * inv_fs));         // exp(-j 2 phi)
        const float sh = -w2.y, chh = w2.x;
        const float kappa = 4.0f * sh * sh;                            // 4 sin^2(2 phi): step of four positions
        const float2 omw4 = make_float2(2.0f * sh * sh, -2.0f * sh * chh);   // 1 - exp(+j 4 phi)
        const float2 z0 = phasor_rev(fmaf(f_eff, (float)(q0 + pi + 1) * inv_fs, ph_rev));
        const float2 z1 = cmulf(z0, w2);
        const float2 dz0 = cmulf(z0, omw4), dz1 = cmulf(z1, omw4);     // z(m) - z(m - 4)
        u2[n] = sp2{s.active ? fmaf(sx, z0.x, sy * z0.y) : 0.f, s.active ? fmaf(sx, z1.x, sy * z1.y) : 0.f};
        dl2[n] = sp2{s.active ? fmaf(sx, dz0.x, sy * dz0.y) : 0.f,
                     s.active ? fmaf(sx, dz1.x, sy * dz1.y) : 0.f};
        nkv[n] = -kappa;
        const int d = s.active ? md.delay_used : 0;
        const int rel = d - q0;                                        // boundary relative to the quarter
        s.all_lo = s.active && rel >= kSpQuarter;
        s.pb = (s.active && rel > 0 && rel < kSpQuarter) ? rel : kSpInf;
        s.crow = cd + ((c < kSpCh ? c : 0) * 2 + pi) * kSpCodePitch;
        all_lo[n] = s.all_lo;
    }
    nk01 = sp2{nkv[0], nkv[1]};                                        // -kappa of both tiles in one pair
    const int rel0 = pos0 - q0;                                        // range start within the quarter
    // a range that starts inside the quarter: the recurrence steps of the positions before it
#pragma unroll 1
    for (int s = 0; s < rel0 / 4; ++s) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            dl2[n] = n == 0 ? sp_rec_lo(nk01, u2[n], dl2[n]) : sp_rec_hi(nk01, u2[n], dl2[n]);
            u2[n] = mf_pk_add(u2[n], dl2[n]);
        }
    }

    // ---- replica staging: lane l < 48 fetches 8 consecutive entries of one (channel, parity)
    // row of a 32-position window: channel l / 4, parity (l / 2) & 1, half l & 1.  Tile
    // position m = 2 q + pi, rolled index r = (m - d) mod 2048 has parity e = (pi - d) & 1 and
    // half index (r - e) / 2, which advances by one per pair: a contiguous run of the table
    // plane e (doubled, so the run never wraps).  A closed channel reads PRN slot 0 (zeros).
    const int sc = lane >> 2, spl = (lane >> 1) & 1, shf = lane & 1;
    const bool s_on = lane < 4 * kSpCh && sc < nch_g;
    const JobMid smd = midrow[s_on ? sc : 0];
    const bool s_act = s_on && smd.active;
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

    sp4 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        acc[n] = sp4{0.f, 0.f, 0.f, 0.f};
        tot[n] = sp4{0.f, 0.f, 0.f, 0.f};
    }
    // the four lanes of a channel close its lo sum where the boundary passes: the sum is
    // parked in global memory (a dozen events per block) instead of a second register set
    // and read back after the loop
    auto close_lo = [&](int n) {
        const sp4 v = tot[n] + acc[n];
#pragma unroll
        for (int e = 0; e < 4; ++e) park[(n * 4 + e) * 64 + lane] = v[e];
        tot[n] = sp4{0.f, 0.f, 0.f, 0.f};
        acc[n] = sp4{0.f, 0.f, 0.f, 0.f};
    };
    auto next_boundary = [&](int after) {                            // first boundary position > after
        int v = kSpInf;
#pragma unroll
        for (int n = 0; n < 2; ++n)
            if (ch[n].pb > after && ch[n].pb < rel0 + NSPANS * kSpSpan) v = ch[n].pb < v ? ch[n].pb : v;
        return sp_wave_min(v);
    };
    int nb = next_boundary(rel0);        // (a boundary AT the range start needs no action: all hi)
    const float* ap = tl + (lane & 15) * kSpRowDw + k;               // lane = (row, k)

    [[maybe_unused]] unsigned long long ts_wait = 0;
#pragma unroll 1
    for (int tix = 0; tix < kTiles; ++tix) {
#ifdef GPSMI_SP_PROF
        {
            const unsigned long long ta = clock64();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ts_wait += clock64() - ta;
        }
#endif
        // the tile that waited in registers goes to LDS (the reads of the previous one are
        // behind us: LDS serves a wave in order), then the tile after it is requested
        // (unconditional: past the end the last tile is fetched again and never used)
        store_code();
        store_tile();
        load_code(2 * tix + 1);
        if (!(DIAG & 2)) load_tile(tix + 1 < kTiles ? tix + 1 : kTiles - 1);
        __builtin_amdgcn_sched_barrier(0);
        const int tpos = rel0 + tix * kSpTile;                       // tile start within the quarter

        // operands of two pairs (four positions): the lane's component of the samples and
        // the replica entries of its two channels
        struct Ops { sp2 a, c0, c1; };
        auto read_ops = [&](int q2) {                                // q2: pair index in the tile, even
            Ops o;
            o.a = sp2{ap[4 * q2], ap[4 * q2 + 4]};
            o.c0 = *reinterpret_cast<const sp2*>(ch[0].crow + (q2 & 15));
            o.c1 = *reinterpret_cast<const sp2*>(ch[1].crow + (q2 & 15));
            return o;
        };
        auto two = [&](const Ops& o, int q2, auto check) {
            const sp2 b0 = mf_pk_mul(o.c0, u2[0]);
            dl2[0] = sp_rec_lo(nk01, u2[0], dl2[0]);
            u2[0] = mf_pk_add(u2[0], dl2[0]);
            const sp2 b1 = mf_pk_mul(o.c1, u2[1]);
            dl2[1] = sp_rec_hi(nk01, u2[1], dl2[1]);
            u2[1] = mf_pk_add(u2[1], dl2[1]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float bv[2] = {b0[i], b1[i]};
                if (decltype(check)::value) {
                    const int P = tpos + 2 * (q2 + i);               // positions P, P + 1
                    if (nb <= P + 1) {                               // some channel's boundary is here
                        bool odd[2];
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            odd[n] = ch[n].pb == P + 1;
                            if (ch[n].pb == P) close_lo(n);          // even boundary: close lo before the pair
                        }
                        if (__builtin_amdgcn_ballot_w64(odd[0] || odd[1]) != 0) {
                            // position P alone for the boundary lanes, both positions elsewhere
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                const float b1st = odd[n] ? (pi == 0 ? bv[n] : 0.f) : bv[n];
                                acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i], b1st, acc[n], 0, 0, 0);
                            }
#pragma unroll
                            for (int n = 0; n < 2; ++n) {
                                if (odd[n]) close_lo(n);
                                bv[n] = odd[n] ? (pi == 1 ? bv[n] : 0.f) : 0.f;   // then position P + 1 of those lanes
                            }
                        }
                        nb = next_boundary(P + 1);
                    }
                }
                if (DIAG & 1) {
                    acc[0][0] = fmaf(o.a[i], bv[0], acc[0][0]);
                    acc[1][0] = fmaf(o.a[i], bv[1], acc[1][0]);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i], bv[0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a[i], bv[1], acc[1], 0, 0, 0);
                }
            }
        };
        // one replica window = 16 pairs = eight steps of two pairs
        auto half_tile = [&](int hw) {
            Ops cur = read_ops(16 * hw);
            if (nb >= tpos + (hw + 1) * kSpWin) {
#pragma unroll
                for (int q2 = 0; q2 < 16; q2 += 2) {
                    Ops nxt = cur;
                    if (q2 + 2 < 16) nxt = read_ops(16 * hw + q2 + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    two(cur, 16 * hw + q2, std::false_type{});
                    cur = nxt;
                }
            } else {
#pragma unroll 1
                for (int q2 = 0; q2 < 16; q2 += 2) {
                    const Ops nxt = read_ops(16 * hw + (q2 + 2 < 16 ? q2 + 2 : q2));   // last: harmless re-read
                    __builtin_amdgcn_sched_barrier(0);
                    two(cur, 16 * hw + q2, std::true_type{});
                    cur = nxt;
                }
            }
        };
        half_tile(0);
        // second window of the tile: its entries were requested before the rows of the next
        // tile, so waiting for them never waits for rows from HBM
        store_code();
        load_code(2 * tix + 2);
        half_tile(1);
        if (tix & 1) {                                               // a span ends
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                tot[n] = tot[n] + acc[n];
                acc[n] = sp4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
#ifdef GPSMI_SP_PROF
    if (lane == 0) atomicAdd(&g_sp_wait[0], ts_wait);
#endif
    // lo sums of the lanes whose boundary lay inside the range (own stores: wait, bypass L1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        lo_fin[n] = sp4{0.f, 0.f, 0.f, 0.f};
        if (ch[n].pb > rel0 && ch[n].pb < rel0 + NSPANS * kSpSpan) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                lo_fin[n][e] = __builtin_nontemporal_load(&park[(n * 4 + e) * 64 + lane]);
        }
    }
}

// what a range contributes to the two sides of the delay (all_lo: the whole QUARTER is lo)
__device__ __forceinline__ void span_sides(const sp4& tot, const sp4& lo_fin, bool all_lo, sp4& w_hi,
                                           sp4& w_lo) {
    const sp4 z = sp4{0.f, 0.f, 0.f, 0.f};
    w_hi = all_lo ? z : tot;
    w_lo = all_lo ? tot : lo_fin;
}

// ---- batch form: one workgroup = one block x up to 12 channels, eight waves = two row
// halves x four quarters; two workgroups per CU.  Output as trk_stream_mfma_kernel:
// partial[job][q + 1] = U[q] hi[q] + U[q+1] lo[q+1].
template <int DIAG = 0>
__global__ __launch_bounds__(512, 4) void trk_span_kernel(
    const float2* __restrict__ iq, const JobMid* __restrict__ mid,
    const float* __restrict__ code_eo, TrkParams P, int ngroups, int nblocks,
    float* __restrict__ lo_park, float2* __restrict__ partial) {
    constexpr int NC = 32, CS = kFftN;
    __shared__ __attribute__((aligned(16))) float lds[8][kSpWaveFloats];
    constexpr int kSumFloats = kSpCh * 16 * 2;          // [channel][row of the half][re, im]
    static_assert(2 * kSumFloats <= kSpTileFloats, "row sums must fit the tile area");

    [[maybe_unused]] const unsigned long long ts0 = SP_STAMP();
    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int g = slot % ngroups;
    const int b = (slot / ngroups) * 8 + xcd;
    if (b >= nblocks) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int h = wave >> 2, qd = wave & 3;
    const float2* blk = iq + (size_t)b * ((size_t)CS * NC);
    const int nch_g = P.nch - g * kSpCh;
    float* tl = &lds[wave][0];
    float* cd = tl + kSpTileFloats;

    sp4 tot[2], lo_fin[2];
    bool all_lo[2];
    span_wave<kSpQuarter / kSpSpan, DIAG>(blk, tl, cd, mid + (size_t)b * P.nch + g * kSpCh, nch_g, code_eo,
                                    lo_park + ((size_t)wg * 8 + wave) * kSpParkFloats, h,
                                    qd * kSpQuarter, lane, tot, lo_fin, all_lo);

    if (DIAG & 4) {
        if (tot[0][0] + tot[1][1] + lo_fin[0][2] + lo_fin[1][3] == 123.456f)
            partial[(size_t)b * P.nch * (NC + 1) + t] = make_float2(tot[0][0], tot[1][0]);
        return;
    }
    [[maybe_unused]] const unsigned long long ts1 = SP_STAMP();
    // ---- per wave: hi / lo sums of its 16 rows into LDS.  D[i = 4 (lane / 16) + v][j = lane % 16]
    {
        float* hi = tl;
        float* lo = tl + kSumFloats;
        const int j = lane & 15, part = j & 1;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int c = 8 * n + (j >> 1);
            sp4 w_hi, w_lo;
            span_sides(tot[n], lo_fin[n], all_lo[n], w_hi, w_lo);
            if (c < kSpCh) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int row = 4 * (lane >> 4) + v;
                    hi[(c * 16 + row) * 2 + part] = w_hi[v];
                    lo[(c * 16 + row) * 2 + part] = w_lo[v];
                }
            }
        }
    }
    __syncthreads();
    [[maybe_unused]] const unsigned long long ts2 = SP_STAMP();
    // ---- combine the quarters (fixed order), apply U, write partial[q + 1], q = -1 .. 31
    for (int item = t; item < kSpCh * (NC + 1); item += 512) {
        const int cc = item / (NC + 1), o = item % (NC + 1), q = o - 1;
        const int ci = g * kSpCh + cc;
        if (ci >= P.nch) continue;
        const JobMid m2 = mid[b * P.nch + ci];
        if (!m2.active) continue;
        float hx = 0.f, hy = 0.f, lx = 0.f, ly = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (q >= 0) {
                const float* hi = &lds[(q >> 4) * 4 + w][0];
                hx += hi[(cc * 16 + (q & 15)) * 2];
                hy += hi[(cc * 16 + (q & 15)) * 2 + 1];
            }
            if (q + 1 < NC) {
                const float* lo = &lds[((q + 1) >> 4) * 4 + w][0] + kSumFloats;
                lx += lo[(cc * 16 + ((q + 1) & 15)) * 2];
                ly += lo[(cc * 16 + ((q + 1) & 15)) * 2 + 1];
            }
        }
        const double fr = (double)m2.om * 0.15915494309189533576888376337251 * 1.0e-3;
        const double r0 = fr * (double)q, r1 = fr * (double)(q + 1);
        const float2 u0 = phasor_rev((float)(r0 - rint(r0)));          // U[q]
        const float2 u1 = phasor_rev((float)(r1 - rint(r1)));          // U[q+1]
        const float re = (hx * u0.x - hy * u0.y) + (lx * u1.x - ly * u1.y);
        const float im = (hy * u0.x + hx * u0.y) + (ly * u1.x + lx * u1.y);
        partial[((size_t)b * P.nch + ci) * (NC + 1) + o] = make_float2(re, im);
    }
#ifdef GPSMI_SP_PROF
    if (lane == 0 && g_sp_prof) {
        unsigned long long* o = g_sp_prof + ((size_t)wg * 8 + wave) * 4;
        o[0] = ts0; o[1] = ts1; o[2] = ts2; o[3] = clock64();
    }
#endif
}

}  // namespace gpsmi
