// The tracking correlator on the matrix pipe (default for CS = 2048, N_CYC = 32;
// GPSMI_STREAM_MFMA=0 selects the vector kernel of gpsmi_trk_stream.h).
//
// Same mathematics as gpsmi_trk_stream.h -- prompt correlate-and-dump of a 32-ms block,
// y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))) summed per code-period
// window (reference src/gpslib.py:1400-1420) -- but the 12 complex MACs per sample are
// issued as one v_mfma_f32_32x32x2_f32 per code POSITION instead of 24 packed FMAs per
// lane: the vector kernel is bound by VALU issue at 12 channels, the
// fp32 matrix pipe has the same peak rate and leaves the VALU free.
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
// A workgroup = one block x 12 channels, eight waves (two per SIMD: one wave's VALU / LDS
// work runs beside the other's MFMA), wave w owns positions [256 w, 256 w + 256).  The
// rows arrive by coalesced 256-byte row segments (tiles of 32 rows x 32 positions), wait
// in registers for one tile, are written to a wave-private LDS tile as two planes (re /
// im, row pitch 36 floats) and read back transposed, lane = row, four positions per
// ds_read_b128 (no workgroup barrier in the loop).  B is generated on the VALU in the
// shadow of the MFMA: the one real phasor component a lane needs, advanced by a coupled
// two-FMA recurrence (re-seeded exactly at every tile), times the rolled replica sample
// (doubled table, staged through LDS per tile).
//
// Window q of the reference = positions m >= d of row q plus m < d of row q+1.  A wave
// keeps ONE accumulator; at m = d_c (a scalar compare per position against the next
// boundary) the lanes of channel c move their columns into a second register set and
// start from zero, so at the end save = sum over m < d (the "lo" part of every row), acc =
// the "hi" part.  The combine step adds the four waves in fixed order and forms
// partial[q+1] = U[q] hi[q] + U[q+1] lo[q+1]  (U[r] = exp(-j w r T), the row factor).
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

constexpr int kMfCh = 12;                 // channels per workgroup
constexpr int kMfTile = 32;               // positions per tile
constexpr int kMfWaves = 8;               // waves per workgroup: two per SIMD, so that one
                                          // wave's VALU / LDS work runs beside the other's MFMA
constexpr int kMfPitch = kMfTile + 4;     // floats per tile row: 16-byte rows, b128 reads conflict-free
constexpr int kMfPlane = 32 * kMfPitch;   // floats per plane
constexpr int kMfTileFloats = 2 * kMfPlane;

typedef float mf16 __attribute__((ext_vector_type(16)));
typedef float mf4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned b128 load

__device__ __forceinline__ int mf_wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return __builtin_amdgcn_readfirstlane(v);
}

__global__ __launch_bounds__(64 * kMfWaves, 1) void trk_stream_mfma_kernel(
    const float2* __restrict__ iq, const JobMid* __restrict__ mid,
    const float* __restrict__ code2, TrkParams P, int ngroups, int nblocks,
    float2* __restrict__ partial) {
    constexpr int NC = 32, CS = kFftN;
    __shared__ __attribute__((aligned(16))) float tiles[kMfWaves][kMfTileFloats];      // per wave, one tile
    __shared__ __attribute__((aligned(16))) float codes[kMfWaves][kMfCh][kMfPitch];   // per wave: the
                                                                    // tile's rolled replica samples
    // after the loop the first 6 KiB of a wave's tile area hold its row sums:
    // [hi | lo][channel][row] complex
    constexpr int kSumFloats = kMfCh * NC * 2;

    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int g = slot % ngroups;
    const int b = (slot / ngroups) * 8 + xcd;
    if (b >= nblocks) return;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const float2* blk = iq + (size_t)b * ((size_t)CS * NC);
    constexpr int kWavePos = CS / kMfWaves;                    // positions per wave
    const int w0 = kWavePos * wave;

    // ---- tile staging: a lane's share of a tile is 8 float4 (row = idx / 16, two positions each);
    // the next tile waits in registers while the current one is read from LDS
    float* tl = &tiles[wave][0];
    const float2* src = blk + w0;
    constexpr int kLd = 32 * kMfTile / 2 / 64;                 // float4 per lane and tile
    float4 st[kLd];
    auto load_tile = [&](int tix) {
#pragma unroll
        for (int i = 0; i < kLd; ++i) {
            const int idx = i * 64 + lane, r = idx / (kMfTile / 2), c4 = idx % (kMfTile / 2);
            st[i] = *reinterpret_cast<const float4*>(src + (size_t)r * CS + tix * kMfTile + 2 * c4);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < kLd; ++i) {
            const int idx = i * 64 + lane, r = idx / (kMfTile / 2), c4 = idx % (kMfTile / 2);
            float* q = tl + r * kMfPitch + 2 * c4;
            *reinterpret_cast<float2*>(q) = make_float2(st[i].x, st[i].z);              // re plane
            *reinterpret_cast<float2*>(q + kMfPlane) = make_float2(st[i].y, st[i].w);   // im plane
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
    // z(p+1) = z(p) exp(-j phi).  Advanced by the coupled (Reinsch) recurrence
    //   dl(p+1) = dl(p) - kappa u(p),  u(p+1) = u(p) + dl(p+1),  kappa = 4 sin^2(phi/2),
    // two FMAs per position, accurate for the small phi of a Doppler (2 cos(phi) u - u'
    // is not), re-seeded exactly at every tile.
    const float2 hph = phasor_rev(0.5f * f_eff * inv_fs);             // (cos(phi/2), -sin(phi/2))
    const float sh = -hph.y, ch = hph.x;
    const float kappa = 4.0f * sh * sh;
    const float2 omw = make_float2(2.0f * sh * sh, -2.0f * sh * ch);  // 1 - conj(exp(-j phi))
    const int d = active ? md.delay_used : 0;
    const float* cp = code2 + (size_t)(active ? md.prn : 0) * (2 * CS) + ((w0 - d) & (CS - 1));
    // where this wave's positions change from "lo" (m < d) to "hi": 0 = all hi, kWavePos = all lo
    int pb = d <= w0 ? 0 : (d >= w0 + kWavePos ? kWavePos : d - w0);
    if (!active) pb = 0;

    // The replica samples of a tile (12 channels x 32 positions) go through LDS as well:
    // lane l < 48 fetches 8 consecutive samples of channel l / 4 one tile ahead (before
    // the row loads of the tile after next, so that waiting for them -- loads return in
    // order -- never waits for rows from HBM) and every lane reads its own channel's row.
    const int sc = lane >> 2, sq = lane & 3;                  // staging role: channel, quarter of the tile
    const int sidx = g * kMfCh + sc;
    const bool s_on = lane < 4 * kMfCh && sidx < P.nch;
    const JobMid smd = mid[b * P.nch + (s_on ? sidx : 0)];
    const bool s_act = s_on && smd.active;
    const float* scp = code2 + (size_t)(s_act ? smd.prn : 0) * (2 * CS)
                       + ((w0 - (s_act ? smd.delay_used : 0)) & (CS - 1)) + (kMfTile / 4) * sq;
    float* sdst = &codes[wave][lane < 4 * kMfCh ? sc : 0][(kMfTile / 4) * sq];
    constexpr int kCst = kMfTile / 16;
    float4 cst[kCst];
    auto load_code = [&](int tix) {
#pragma unroll
        for (int i = 0; i < kCst; ++i) {
            const mf4u q = *reinterpret_cast<const mf4u*>(scp + tix * kMfTile + 4 * i);
            cst[i] = make_float4(q.x, q.y, q.z, q.w);
        }
    };
    auto store_code = [&]() {
        if (lane < 4 * kMfCh) {
#pragma unroll
            for (int i = 0; i < kCst; ++i)
                *reinterpret_cast<float4*>(sdst + 4 * i) =
                    s_act ? cst[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    constexpr int kTiles = kWavePos / kMfTile;
    load_code(0);

    mf16 acc, save;
#pragma unroll
    for (int v = 0; v < 16; ++v) { acc[v] = 0.f; save[v] = 0.f; }
    int nb = mf_wave_min(pb > 0 && pb < kWavePos ? pb : 1 << 20);     // next boundary position
    const float* ap = tl + kk * kMfPlane + (lane & 31) * kMfPitch;    // lane = row, plane = k
    const float* crow = &codes[wave][j < 2 * kMfCh ? c : 0][0];

#pragma unroll 1
    for (int tix = 0; tix < kTiles; ++tix) {
        // the tile that waited in registers goes to LDS (the reads of the previous one are
        // behind us: LDS serves a wave in order), then the tile after it is requested
        // (unconditional: past the end the last tile is fetched again and never used)
        store_code();
        store_tile();
        load_code(tix + 1 < kTiles ? tix + 1 : kTiles - 1);
        load_tile(tix + 1 < kTiles ? tix + 1 : kTiles - 1);
        __builtin_amdgcn_sched_barrier(0);
        // phasor of the tile's first position, exact range reduction
        const int m0 = w0 + tix * kMfTile;
        const float2 z = phasor_rev(fmaf(f_eff, (float)(m0 + 1) * inv_fs, ph_rev));
        const float2 dz = cmulf(z, omw);                              // z(m0) - z(m0 - 1)
        float u = active ? fmaf(sx, z.x, sy * z.y) : 0.f;
        float dl = active ? fmaf(sx, dz.x, sy * dz.y) : 0.f;
        // the operands of the next four positions are read from LDS while the current four
        // are in the matrix pipe
        float4 a4n = *reinterpret_cast<const float4*>(ap);
        float4 c4n = *reinterpret_cast<const float4*>(crow);
#pragma unroll 1
        for (int p4 = 0; p4 < kMfTile; p4 += 4) {
            const float av[4] = {a4n.x, a4n.y, a4n.z, a4n.w};
            const float cv[4] = {c4n.x, c4n.y, c4n.z, c4n.w};
            const int pn = p4 + 4 < kMfTile ? p4 + 4 : p4;            // last group: harmless re-read
            a4n = *reinterpret_cast<const float4*>(ap + pn);
            c4n = *reinterpret_cast<const float4*>(crow + pn);
            const int p0 = tix * kMfTile + p4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (p0 + q == nb) {                                   // some channel's boundary
                    const bool mine = pb == p0 + q;
#pragma unroll
                    for (int v = 0; v < 16; ++v) {
                        save[v] = mine ? acc[v] : save[v];
                        acc[v] = mine ? 0.f : acc[v];
                    }
                    nb = mf_wave_min(pb > p0 + q && pb < kWavePos ? pb : 1 << 20);
                }
                const float bval = cv[q] * u;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bval, acc, 0, 0, 0);
                dl = fmaf(-kappa, u, dl);
                u += dl;
            }
        }
    }

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
    // ---- combine the waves (fixed order), apply U, write partial[q + 1], q = -1 .. 31
    for (int item = t; item < kMfCh * (NC + 1); item += 64 * kMfWaves) {
        const int cc = item / (NC + 1), o = item % (NC + 1), q = o - 1;
        const int ci = g * kMfCh + cc;
        if (ci >= P.nch) continue;
        const JobMid m2 = mid[b * P.nch + ci];
        if (!m2.active) continue;
        float hx = 0.f, hy = 0.f, lx = 0.f, ly = 0.f;
#pragma unroll
        for (int w = 0; w < kMfWaves; ++w) {
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
}

}  // namespace gpsmi
