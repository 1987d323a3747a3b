// The tracking correlator on the matrix instructions (default for CS = 2048, N_CYC = 32;
// GPSMI_STREAM_MFMA=0 selects the vector kernel of gpsmi_trk_stream.h).
//
// Same mathematics as gpsmi_trk_stream.h -- prompt correlate-and-dump of a 32-ms block,
// y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed per code-period
// window (reference src/gpslib.py:1400-1420) -- but the 12 complex MACs per sample are
// issued as one v_mfma_f32_32x32x2_f32 per code POSITION instead of 24 packed FMAs per
// lane plus cross-lane sums: the vector kernel is bound by VALU issue at 12 channels
// (DESIGN.md 4.3).
//
//   D[32 x 32] += A[32 x 2] * B[2 x 32] for one position m of the code period:
//   M = the 32 code periods (rows) of the block, K = (re, im) of the sample,
//   N = (channel, re/im) for up to 12 channels:
//       A[r][0..1]      = (x_re[r][m], x_im[r][m])
//       B[.][(c, re)]   = ( B_re, -B_im ),  B[.][(c, im)] = ( B_im, B_re ),
//       B_c(m) = replica_c[(m - d_c) mod 2048] * exp(-j theta_c(m))
//   so D[r][(c, .)] accumulates sum_m B_c(m) x[r][m] over the positions a wave owns --
//   no cross-lane reduction, every sample fetched once for all 12 channels.
//
// What bounds it (tools/probe/mfma_rate.hip, mfma_prof.hip): the fp32 MFMA runs on the
// SIMD's own fp32 lanes -- 64 cycles each, and NO other VALU instruction of ANY wave of
// that SIMD issues meanwhile.  The kernel is therefore bound by 64 cycles x positions plus
// every other VALU instruction, and a VALU instruction outside the loop waits up to a
// whole MFMA of a neighbouring wave.  Hence: WAVES = 4 waves per workgroup (wave w owns
// positions [512 w, 512 w + 512)), three independent workgroups per CU, whose set-up,
// barrier and combine phases fill each other's gaps; 1.5 packed VALU instructions per
// position in the loop; a per-tile preamble that is almost free of VALU work (rows go
// global -> registers -> LDS unchanged, addresses are SGPR base + one lane offset, the
// exact phasor is recomputed every eighth tile only).  WAVES = 8 (two waves per SIMD of
// one workgroup, 256 positions each) has the shorter dependent chain and serves a launch
// too small to fill the CUs.
//
// The rows arrive by coalesced 256-byte row segments (tiles of 32 rows x 32 positions),
// wait in registers for one tile, are written to a wave-private LDS tile as they are
// (interleaved re / im, row pitch 66 dwords) and read back transposed: lane = (row, k),
// two positions of its component per ds_read2_b32 (no workgroup barrier in the loop).  B
// is one real phasor component per lane: two positions ride in a packed register pair
// and advance by a coupled recurrence, times the rolled replica samples (doubled table,
// staged through LDS per tile).
//
// Window q of the reference = positions m >= d of row q plus m < d of row q+1.  A wave
// keeps ONE accumulator; at m = d_c the lanes of channel c move their columns into a
// second register set and start from zero, so at the end save = sum over m < d (the "lo"
// part of every row), acc = the "hi" part.  A tile without a boundary (all but one per
// channel and block) runs 32 MFMAs straight; the others compare every position.  The
// combine step adds the waves in fixed order and forms
// partial[q+1] = U[q] hi[q] + U[q+1] lo[q+1]  (U[r] = exp(-j w r T), the row factor).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace gpsmi {

constexpr int kMfCh = 12;                 // channels per workgroup
constexpr int kMfTile = 32;               // positions per tile
constexpr int kMfRowDw = 2 * kMfTile + 2; // dwords per tile row: lane = row reads hit every bank twice
constexpr int kMfTileFloats = 32 * kMfRowDw;
constexpr int kMfCodePitch = kMfTile + 4; // floats per replica row: b128 reads conflict-free
constexpr int kMfReseed = 8;              // tiles between exact phasors
constexpr int kMfRsrcFlags = kRsrcFlags;

typedef float mf16 __attribute__((ext_vector_type(16)));
typedef float mf2 __attribute__((ext_vector_type(2)));
typedef float mf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ mf2 mf_pk_mul(mf2 a, mf2 b) {
    mf2 r;
    // (its result feeds an MFMA: a VALU write needs two wait states before the MFMA reads
    // it, and the compiler does not count them for inline asm)
    asm("v_pk_mul_f32 %0, %1, %2\n\ts_nop 1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ mf2 mf_pk_add(mf2 a, mf2 b) {
    mf2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ mf2 mf_pk_fma(mf2 a, mf2 b, mf2 c) {
    mf2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ int mf_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

#ifdef GPSMI_MF_PROF      // tools/probe/mfma_prof.hip only: per-wave cycle stamps
__device__ unsigned long long* g_mf_prof;
#define MF_STAMP() clock64()
#else
#define MF_STAMP() 0ull
#endif

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 4 ? 3 : 1) void trk_stream_mfma_kernel(
    const float2* __restrict__ iq, const JobMid* __restrict__ mid,
    const float* __restrict__ code2, TrkParams P, int ngroups, int nblocks,
    float2* __restrict__ partial) {
    constexpr int NC = 32, CS = kFftN;
    __shared__ __attribute__((aligned(16))) float tiles[WAVES][kMfTileFloats];         // per wave, one tile
    __shared__ __attribute__((aligned(16))) float codes[WAVES][kMfCh][kMfCodePitch];   // per wave: the
                                                                    // tile's rolled replica samples
    // after the loop the first 6 KiB of a wave's tile area hold its row sums:
    // [hi | lo][channel][row] complex
    constexpr int kSumFloats = kMfCh * NC * 2;
    static_assert(2 * kSumFloats <= kMfTileFloats, "row sums must fit the tile area");

    [[maybe_unused]] const unsigned long long ts0 = MF_STAMP();
    [[maybe_unused]] unsigned long long ts_wait = 0, ts_pre = 0, ts_in = 0;
    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int g = slot % ngroups;
    const int b = (slot / ngroups) * 8 + xcd;
    if (b >= nblocks) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);   // uniform: addresses stay scalar
    const float2* blk = iq + (size_t)b * ((size_t)CS * NC);
    constexpr int kWavePos = CS / WAVES;                       // positions per wave
    const int w0 = kWavePos * wave;

    // ---- tile staging: a lane's share of a tile is 8 float4 (row = 4 i + lane / 16, positions
    // 2 (lane % 16) and the next); the next tile waits in registers while the current one
    // is read from LDS.  Global address = scalar base (block, wave, tile, i) + one lane offset.
    // (buffer loads: descriptor = the block, lane offset in a VGPR, everything else in an
    // SGPR -- no per-tile address arithmetic on the VALU; out of range reads return zero)
    float* tl = &tiles[wave][0];
    const __amdgpu_buffer_rsrc_t blk_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float2*>(blk), 0, CS * NC * (int)sizeof(float2), kMfRsrcFlags);
    constexpr int kLd = 32 * kMfTile / 2 / 64;                 // float4 per lane and tile
    const int ld_off = ((lane >> 4) * CS + 2 * (lane & 15)) * (int)sizeof(float2);
    float* st_dst = tl + (lane >> 4) * kMfRowDw + 4 * (lane & 15);
    mf4 st[kLd];
    auto load_tile = [&](int tix) {
        const int tb = (w0 + tix * kMfTile) * (int)sizeof(float2);
#pragma unroll
        for (int i = 0; i < kLd; ++i)
            st[i] = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(
                blk_rs, ld_off, tb + i * (4 * CS * (int)sizeof(float2)), 0));
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < kLd; ++i) {                  // rows are 8-byte aligned: two b64 writes
            *reinterpret_cast<mf2*>(st_dst + i * 4 * kMfRowDw) = mf2{st[i].x, st[i].y};
            *reinterpret_cast<mf2*>(st_dst + i * 4 * kMfRowDw + 2) = mf2{st[i].z, st[i].w};
        }
    };
    load_tile(0);                          // requested before anything that depends on the job
                                           // descriptors: HBM latency hides behind the set-up

    // ---- lane roles: column j = (channel, re/im), k = which half of the complex sample
    const int j = lane & 31, kk = lane >> 5;
    const int c = j >> 1, part = j & 1;
    const int cidx = g * kMfCh + c;
    const bool col = j < 2 * kMfCh && cidx < P.nch;
    const JobMid md = mid[b * P.nch + (col ? cidx : 0)];
    const bool active = col && md.active;
    const float sx = ((kk == 0) == (part == 0)) ? 1.f : 0.f;          // (0,re) and (1,im): +z.x
    const float sy = (sx != 0.f) ? 0.f : (part == 0 ? -1.f : 1.f);    // (1,re): -z.y, (0,im): +z.y
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float inv_fs = 1.0f / (1000.0f * (float)CS);
    const float f_eff = active ? (float)((double)md.om * inv_2pi) : 0.f;
    const float ph_rev = active ? md.ph * (float)inv_2pi : 0.f;
    // B needs one real component of the carrier phasor per lane: u(p) = sx Re z + sy Im z,
    // z(p+1) = z(p) exp(-j phi).  Two positions (p, p+1) ride in one packed register pair and
    // advance two positions at a time by the coupled (Reinsch) recurrence
    //   dl(p+2) = dl(p) - kappa u(p),  u(p+2) = u(p) + dl(p+2),  kappa = 4 sin^2(phi),
    // (one v_pk_fma_f32 + one v_pk_add_f32 per pair; accurate for the small phi of a Doppler,
    // where 2 cos(2 phi) u - u' is not), re-seeded exactly every kMfReseed tiles.
    const float2 wph = phasor_rev(f_eff * inv_fs);                    // exp(-j phi) = (cos phi, -sin phi)
    const float sh = -wph.y, ch = wph.x;
    const float kappa = 4.0f * sh * sh;
    const float2 omw = make_float2(2.0f * sh * sh, -2.0f * sh * ch);  // 1 - conj(exp(-j 2 phi))
    const int d = active ? md.delay_used : 0;
    // where this wave's positions change from "lo" (m < d) to "hi": 0 = all hi, kWavePos = all lo
    int pb = d <= w0 ? 0 : (d >= w0 + kWavePos ? kWavePos : d - w0);
    if (!active) pb = 0;

    // The replica samples of a tile (12 channels x 32 positions) go through LDS as well:
    // lane l < 48 fetches 8 consecutive samples of channel l / 4 one tile ahead (before
    // the row loads of the tile after next, so that waiting for them -- loads return in
    // order -- never waits for rows from HBM) and every lane reads its own channel's row.
    // A closed channel reads replica slot 0 (zeros).
    const int sc = lane >> 2, sq = lane & 3;                  // staging role: channel, quarter of the tile
    const int sidx = g * kMfCh + sc;
    const bool s_on = lane < 4 * kMfCh && sidx < P.nch;
    const JobMid smd = mid[b * P.nch + (s_on ? sidx : 0)];
    const bool s_act = s_on && smd.active;
    const __amdgpu_buffer_rsrc_t code_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(code2), 0, (GPSMI_MAX_PRN + 1) * 2 * CS * (int)sizeof(float), kMfRsrcFlags);
    const int sc_off = ((s_act ? smd.prn : 0) * (2 * CS) + ((w0 - (s_act ? smd.delay_used : 0)) & (CS - 1))
                        + (kMfTile / 4) * sq) * (int)sizeof(float);
    float* sdst = &codes[wave][lane < 4 * kMfCh ? sc : 0][(kMfTile / 4) * sq];
    constexpr int kCst = kMfTile / 16;
    mf4 cst[kCst];
    auto load_code = [&](int tix) {
#pragma unroll
        for (int i = 0; i < kCst; ++i) {
            cst[i] = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(
                code_rs, sc_off, (tix * kMfTile + 4 * i) * (int)sizeof(float), 0));
        }
    };
    auto store_code = [&]() {
        if (lane < 4 * kMfCh) {
#pragma unroll
            for (int i = 0; i < kCst; ++i) *reinterpret_cast<mf4*>(sdst + 4 * i) = cst[i];
        }
    };
    constexpr int kTiles = kWavePos / kMfTile;
    load_code(0);

    mf16 acc, save;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc[v] = 0.f; save[v] = 0.f; }
    int nb = mf_wave_min(pb > 0 && pb < kWavePos ? pb : 1 << 20);     // next boundary position
    const float* ap = tl + (lane & 31) * kMfRowDw + kk;               // lane = (row, component)
    const float* crow = &codes[wave][j < 2 * kMfCh ? c : 0][0];
    const mf2 nkappa2 = {-kappa, -kappa};
    mf2 u2 = {0.f, 0.f}, dl2 = {0.f, 0.f};

    [[maybe_unused]] const unsigned long long ts1 = MF_STAMP();
#pragma unroll 1
    for (int tix = 0; tix < kTiles; ++tix) {
#ifdef GPSMI_MF_PROF
        const unsigned long long ta = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tb = clock64();
        ts_wait += tb - ta;
#endif
        // the tile that waited in registers goes to LDS (the reads of the previous one are
        // behind us: LDS serves a wave in order), then the tile after it is requested
        // (unconditional: past the end the last tile is fetched again and never used)
        store_code();
        store_tile();
        load_code(tix + 1 < kTiles ? tix + 1 : kTiles - 1);
        load_tile(tix + 1 < kTiles ? tix + 1 : kTiles - 1);
        __builtin_amdgcn_sched_barrier(0);
        // phasor of the first two positions, exact range reduction
        if ((tix & (kMfReseed - 1)) == 0) {
            const int m0 = w0 + tix * kMfTile;
            const float2 z0 = phasor_rev(fmaf(f_eff, (float)(m0 + 1) * inv_fs, ph_rev));
            const float2 z1 = cmulf(z0, wph);
            const float2 dz0 = cmulf(z0, omw), dz1 = cmulf(z1, omw);  // z(m) - z(m - 2)
            u2 = mf2{active ? fmaf(sx, z0.x, sy * z0.y) : 0.f, active ? fmaf(sx, z1.x, sy * z1.y) : 0.f};
            dl2 = mf2{active ? fmaf(sx, dz0.x, sy * dz0.y) : 0.f,
                      active ? fmaf(sx, dz1.x, sy * dz1.y) : 0.f};
        }
        // operands of four positions: the lane's component of the samples, the replica
        struct Ops { mf4 a, c; };
        auto read_ops = [&](int p4) {
            Ops o;
            o.a = mf4{ap[2 * p4], ap[2 * p4 + 2], ap[2 * p4 + 4], ap[2 * p4 + 6]};
            o.c = *reinterpret_cast<const mf4*>(crow + p4);
            return o;
        };
        // four positions: B of two position pairs (packed), the recurrence, four MFMAs
        auto four = [&](const Ops& o, int p0, auto check) {
            const mf2 b01 = mf_pk_mul(mf2{o.c.x, o.c.y}, u2);
            dl2 = mf_pk_fma(nkappa2, u2, dl2);
            u2 = mf_pk_add(u2, dl2);
            const mf2 b23 = mf_pk_mul(mf2{o.c.z, o.c.w}, u2);
            dl2 = mf_pk_fma(nkappa2, u2, dl2);
            u2 = mf_pk_add(u2, dl2);
            const float av[4] = {o.a.x, o.a.y, o.a.z, o.a.w};
            const float bv[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (decltype(check)::value && p0 + q == nb) {          // some channel's boundary
                    const bool mine = pb == p0 + q;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        save[v] = mine ? acc[v] : save[v];
                        acc[v] = mine ? 0.f : acc[v];
                    }
                    nb = mf_wave_min(pb > p0 + q && pb < kWavePos ? pb : 1 << 20);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc, 0, 0, 0);
            }
        };
        Ops cur = read_ops(0);
#ifdef GPSMI_MF_PROF
        const unsigned long long tc = clock64();
        ts_pre += tc - tb;
#endif
        if (nb >= (tix + 1) * kMfTile) {
            // no boundary in this tile: 32 MFMAs straight, LDS offsets are immediates; the
            // operands of the next four positions are read while the current four compute
#pragma unroll
            for (int p4 = 0; p4 < kMfTile; p4 += 4) {
                Ops nxt = cur;
                if (p4 + 4 < kMfTile) nxt = read_ops(p4 + 4);
                __builtin_amdgcn_sched_barrier(0);
                four(cur, tix * kMfTile + p4, std::false_type{});
                cur = nxt;
            }
        } else {
#pragma unroll 1
            for (int p4 = 0; p4 < kMfTile; p4 += 4) {
                const Ops nxt = read_ops(p4 + 4 < kMfTile ? p4 + 4 : p4);   // last: harmless re-read
                __builtin_amdgcn_sched_barrier(0);
                four(cur, tix * kMfTile + p4, std::true_type{});
                cur = nxt;
            }
        }
#ifdef GPSMI_MF_PROF
        ts_in += clock64() - tc;
#endif
    }
    [[maybe_unused]] const unsigned long long ts5 = MF_STAMP();

    // ---- per wave: lo / hi sums of every row into LDS (column pairs -> complex)
    {
        // D[i = 8 (v/4) + 4 (lane/32) + v%4][j = lane%32]
        const bool all_lo = pb == kWavePos;
        float* hi = &tiles[wave][0];
        float* lo = hi + kSumFloats;
        if (j < 2 * kMfCh) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = 8 * (v >> 2) + 4 * kk + (v & 3);
                hi[(c * NC + row) * 2 + part] = all_lo ? 0.f : acc[v];
                lo[(c * NC + row) * 2 + part] = all_lo ? acc[v] : save[v];
            }
        }
    }
    __syncthreads();
    [[maybe_unused]] const unsigned long long ts6 = MF_STAMP();
    // ---- combine the waves (fixed order), apply U, write partial[q + 1], q = -1 .. 31
    for (int item = t; item < kMfCh * (NC + 1); item += 64 * WAVES) {
        const int cc = item / (NC + 1), o = item % (NC + 1), q = o - 1;
        const int ci = g * kMfCh + cc;
        if (ci >= P.nch) continue;
        const JobMid m2 = mid[b * P.nch + ci];
        if (!m2.active) continue;
        float hx = 0.f, hy = 0.f, lx = 0.f, ly = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) {
            const float* hi = &tiles[w][0];
            const float* lo = hi + kSumFloats;
            if (q >= 0) { hx += hi[(cc * NC + q) * 2]; hy += hi[(cc * NC + q) * 2 + 1]; }
            if (q + 1 < NC) { lx += lo[(cc * NC + q + 1) * 2]; ly += lo[(cc * NC + q + 1) * 2 + 1]; }
        }
        const double fr = (double)m2.om * 0.15915494309189533576888376337251 * 1.0e-3;
        const double r0 = fr * (double)q, r1 = fr * (double)(q + 1);
        const float2 u0 = phasor_rev((float)(r0 - rint(r0)));          // U[q]
        const float2 u1 = phasor_rev((float)(r1 - rint(r1)));          // U[q+1]
        const float re = (hx * u0.x - hy * u0.y) + (lx * u1.x - ly * u1.y);
        const float im = (hy * u0.x + hx * u0.y) + (ly * u1.x + lx * u1.y);
        partial[((size_t)b * P.nch + ci) * (NC + 1) + o] = make_float2(re, im);
    }
#ifdef GPSMI_MF_PROF
    if (lane == 0 && g_mf_prof) {
        unsigned long long* o = g_mf_prof + ((size_t)wg * WAVES + wave) * 8;
        o[0] = ts0; o[1] = ts1; o[2] = ts_wait; o[3] = ts_pre; o[4] = ts_in; o[5] = ts5;
        o[6] = ts6; o[7] = clock64();
    }
#endif
}

}  // namespace gpsmi
