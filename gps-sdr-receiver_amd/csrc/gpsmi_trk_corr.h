// Code-phase correlation of the tracking loop: cacodeCorr + findCodePhase +
// fitCodePhase (reference src/gpslib.py:1315-1327, :1293-1304, :1268-1290) for
// up to CG (1, 2, 4 or 6) channels of one block per workgroup.
//
// The reference wipes the carrier off the centre corr_avg code periods, sums
// their FFTs, multiplies by conj(FFT(replica)) and takes |ifft|.  Here the sum
// of FFTs is the FFT of the sum, and the wipe-off separates exactly as in the
// correlator (gpsmi_trk_span.h): for sample k = i*CS + m
//     exp(-j(phase + w (k+1)/fs)) = V(m) * U[i],  U[i] = exp(-j w i T),
// so fold[m] = V(m) * sum_i U[i] x[i][m]: one complex multiply-accumulate per
// channel-sample with a wave-uniform U, every sample loaded once for the CG
// channels; V is applied once per position.  Then, per channel, two passes of
// the LDS-resident 2048-point FFT (the second one as the inverse), |.|, mean /
// population std / first-index argmax and the two neighbours of the peak; at the
// end one lane per channel applies the CORR_MIN threshold and the peak fit and
// chooses the DELAY the block is decoded with (gpslib.py:1181-1182).
// DESIGN.md section 4.4 has the measurements behind the structure (what the
// prologue requests at once, the three-deep row pipeline, the LDS tables of V).
#pragma once
#include <type_traits>
#include "gpsmi_fft.h"
#include "gpsmi_stats.h"
#include "gpsmi_wgmap.h"

namespace gpsmi {

// rows of the fold in flight per workgroup: three beside four channels' accumulators; all eight
// when a workgroup serves one channel of one block (the closed loop: latency is all that counts)
template <int CG> constexpr int kFoldDepthOf = CG == 1 ? 8 : 3;
constexpr int kFoldChunk = 8;      // the row count the pipelined fold is written for (CORR_AVG of the reference)

// One channel's correlation result (thread 0 parks it in LDS), and its way into the output
// record and the correlator's job entry.
struct CorrFin { int bi; float bv, mean, sd, elo, ehi; };

__device__ __forceinline__ void corr_finish(const StreamChan& s, const CorrFin& f, const TrkParams& P,
                                            const int* __restrict__ delay_forced,
                                            gpsmi_trk_out* __restrict__ out, JobMid* __restrict__ mid) {
    const int bi = f.bi;
    const float bv = f.bv, mean = f.mean, sd = f.sd, elo = f.elo, ehi = f.ehi;
    const float norm = (bv - mean) / sd;
    gpsmi_trk_out& o = out[s.job];
    o.prn = s.prn;
    o.mx = bi;
    o.epl[0] = elo; o.epl[1] = bv; o.epl[2] = ehi;
    o.corr_mean = mean; o.corr_std = sd;
    o.norm_max_corr = norm;
    int delay = -1;
    double cp = -1.0;
    if (norm > P.corr_min) {
        delay = bi;
        cp = fit_code_phase((double)elo, (double)bv, (double)ehi, bi);
    }
    o.delay = delay;
    o.reserved0 = 0;
    o.code_phase = cp;
    int used = delay >= 0 ? delay : s.d;
    if (delay_forced && delay_forced[s.job] >= 0) used = delay_forced[s.job];
    o.delay_used = used;
    JobMid md;
    md.delay_used = used; md.active = 1; md.prn = s.prn; md.om = s.om; md.ph = s.ph;
    md.pad[0] = md.pad[1] = md.pad[2] = 0;
    mid[s.job] = md;
}

// (FMT 1: iq holds raw uint16 (Q << 8 | I) samples, decoded on load exactly as
// gpsmi_dev_unpack_u8iq does)
template <int CG, int FMT = 0>
__global__ __launch_bounds__(256, (CG > 4 || CG == 1) ? 2 : 3) void trk_corr_kernel(
    const void* __restrict__ iq, const gpsmi_trk_state* __restrict__ st_in,
    const int* __restrict__ delay_forced, const float2* __restrict__ rep,
    const float2* __restrict__ tw, TrkParams P, int ngroups, int nblocks,
    gpsmi_trk_out* __restrict__ out, JobMid* __restrict__ mid) {
    __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
    __shared__ __attribute__((aligned(16))) float lds_tw[kFftTwFloats];
    __shared__ float red[kStatsRedFloats];
    // Two small arrays live inside the second FFT buffer, which a transform leaves free when it
    // returns and does not write before its first barrier: the magnitudes of the statistics
    // (8 KiB) and the row factors of the fold (1 KiB, used before any transform).  39.3 KiB of
    // LDS per workgroup, three workgroups per CU at CG = 4 by the registers (160 VGPRs, capped
    // at 168 by the launch bounds); CG = 2 (four workgroups per CU, the fold's rows read six
    // times per block instead of three) measured 10 % slower per batch; CG = 4 compiled for four
    // workgroups per CU (128 registers: 68-128 bytes of scratch, the per-thread twiddles of the
    // transforms reloaded in front of every pass) 25 % slower (round 4: 0.162-0.165 against 0.131 ms).
    float* magbuf = lds + 2 * kFftPlane;
    float2 (*urow)[32] = reinterpret_cast<float2 (*)[32]>(lds + 2 * kFftPlane);   // U[c][i], i = row
    // (round 4: the factors of V(t) and the 256-position step live there too -- V is applied to all
    // channels' folds right behind the fold, before the first transform touches the buffer)
    float2 (*vtab)[32] = urow + CG;              // the factors of V(t), see the prologue
    float2* step = reinterpret_cast<float2*>(urow + 2 * CG);     // exp(-j w 256/fs)
    static_assert(kFftN <= 2 * kFftPlane1 && (CG * 64 + CG) * 2 <= 2 * kFftPlane1, "aliases must fit buffer 1");
    __shared__ StreamChan schan[CG];
    __shared__ CorrFin fin[CG];

    // batches: the channel groups of a block are neighbours on one XCD (they share its rows through
    // that L2); fewer than 8 blocks (the closed loop) map linearly: gpsmi_wgmap.h
    const CorrWg unit = corr_wg_map((int)blockIdx.x, nblocks, ngroups);
    const int g = unit.group, b = unit.block;
    if (b >= nblocks) return;
    const int t = threadIdx.x;
    const int cs = kFftN;
    constexpr int kFoldDepth = kFoldDepthOf<CG>;
    const float2* blk = static_cast<const float2*>(iq) + (size_t)b * ((size_t)cs * P.n_cyc);
    const uint16_t* rblk = static_cast<const uint16_t*>(iq) + (size_t)b * ((size_t)cs * P.n_cyc);
    const double inv_2pi = 0.15915494309189533576888376337251;
    const int first = (P.n_cyc - P.corr_avg) / 2;
    // The rows of the fold stream through kFoldDepth x 8 registers per thread: the first rows are
    // requested here, ahead of the per-channel constants, and as soon as a position of row i has
    // gone into the accumulators its register is reloaded with the same position of row
    // i + kFoldDepth: that many rows stay in flight instead of one round trip to memory per row
    // (the registers are free: the transforms that follow need more than the fold does).
    // The channel state of the threads that set up the per-channel constants below: every field
    // is requested here, with the first rows and the twiddles, so that the prologue makes one
    // trip to memory instead of one per dependent load.
    const int pc_c = min(t / 33, CG - 1);
    const gpsmi_trk_state& st = st_in[b * P.nch + min(g * CG + pc_c, P.nch - 1)];
    int st_prn = st.prn, st_delay = st.delay;
    float st_om0 = st.omega0, st_freq = st.freq, st_phase = st.phase;
    using Raw = typename std::conditional<FMT == 0, v2f, unsigned>::type;
    auto fetch = [&](int i, int r) -> Raw {
        if constexpr (FMT == 0) {
            const float2 v = blk[(size_t)i * cs + t + 256 * r];
            return v2f{v.x, v.y};
        } else {
            return (unsigned)rblk[(size_t)i * cs + t + 256 * r];
        }
    };
    auto decode = [&](Raw v) -> v2f {
        if constexpr (FMT == 0) {
            return v;
        } else {
            const float scl = 1.0f / 127.5f;
            return v2f{sub_rn(mul_rn((float)(v & 0xFF), scl), 1.0f), sub_rn(mul_rn((float)(v >> 8), scl), 1.0f)};
        }
    };
    Raw x[kFoldDepth][8];
    const bool nofold = diag_flag(P, 32);                     // diagnostics: no fold at all
    const bool piped = P.corr_avg == kFoldChunk && !nofold;   // any other row count: one row at a time
    if (piped) {
#pragma unroll
        for (int k = 0; k < kFoldDepth; ++k)
#pragma unroll
            for (int r = 0; r < 8; ++r) x[k][r] = fetch(first + k, r);
    }
    const FftTw ftw = fft_setup(lds_tw, tw, t);
    asm volatile("" : "+v"(st_prn), "+v"(st_delay), "+v"(st_om0), "+v"(st_freq), "+v"(st_phase));   // (pins the loads above this line)

    // ---- per-channel constants (one thread each), then one barrier
    if (t < CG * 33) {
        const int c = t / 33, i = t % 33;
        const int cidx = g * CG + c;
        const int job = b * P.nch + cidx;
        StreamChan s;
        s.job = job; s.active = 0; s.om = 0.f; s.ph = 0.f; s.d = 0; s.prn = 0;
        if (cidx < P.nch && st_prn > 0) {
            s.active = 1;
            s.om = st_om0 != 0.f ? st_om0 : omega_of(st_freq);
            s.ph = st_phase;
            s.d = st_delay;
            s.prn = st_prn;
        }
        const double f_eff = (double)s.om * inv_2pi;
        if (i < 32) {
            const double rev = f_eff * (double)i * 1.0e-3;            // w i T / 2 pi
            urow[c][i] = phasor_rev((float)(rev - rint(rev)));
            // V at position t = 16 h + k is vtab[k] * vtab[16 + h]: the phase of the channel and
            // position k + 1 in the first factor, 16 h positions in the second
            const double per = f_eff / (1000.0 * (double)cs);         // revolutions per position
            const double rv = i < 16 ? per * (double)(i + 1) + (double)s.ph * inv_2pi
                                     : per * (double)(16 * (i - 16));
            vtab[c][i] = phasor_rev((float)(rv - rint(rv)));
        } else {
            const double rev = f_eff * 256.0 / (1000.0 * (double)cs);
            step[c] = phasor_rev((float)(rev - rint(rev)));
            schan[c] = s;
            if (cidx < P.nch && !s.active) {                          // closed channel
                JobMid z;
                z.delay_used = 0; z.active = 0; z.prn = 0; z.om = 0.f; z.ph = 0.f;
                z.pad[0] = z.pad[1] = z.pad[2] = 0;
                mid[job] = z;
            }
        }
    }
    lds_barrier();                     // (the rows requested above stay in flight)

    // ---- fold: acc[c][r] = sum_i U[c][i] x[i][t + 256 r]
    v2f acc[CG][8];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[c][r] = v2f{0.f, 0.f};
    auto fold_in = [&](const v2f (&u)[CG], int r, v2f xr, v2f xprev) {
        if constexpr (CG == 1) {                 // (cmac2 works on pairs: two positions at once)
            if (r & 1) cmac2(acc[0][r - 1], acc[0][r], u[0], xprev, u[0], xr);
        } else {
#pragma unroll
            for (int c = 0; c + 1 < CG; c += 2)
                cmac2(acc[c][r], acc[c + 1][r], u[c], xr, u[c + 1], xr);
        }
    };
    auto fold_row = [&](int i, Raw (&xk)[8], auto reload, int inext) {
        v2f u[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) u[c] = v2f{urow[c][i].x, urow[c][i].y};
        v2f xprev = v2f{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const v2f xr = decode(xk[r]);
            fold_in(u, r, xr, xprev);
            xprev = xr;
            // (reloaded behind its last use: requested ahead of it, the old value needs a copy)
            if constexpr (decltype(reload)::value) {
                if (CG > 1 || (r & 1)) {
                    xk[r] = fetch(inext, r);
                    if (CG == 1) xk[r - 1] = fetch(inext, r - 1);
                }
            }
        }
    };
    if (piped) {
#pragma unroll
        for (int j = 0; j < kFoldChunk; ++j) {            // straight-line code: exact wait counts
            if (j + kFoldDepth < kFoldChunk)
                fold_row(first + j, x[j % kFoldDepth], std::true_type{}, first + j + kFoldDepth);
            else
                fold_row(first + j, x[j % kFoldDepth], std::false_type{}, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else if (!nofold) {
        for (int i = first; i < first + P.corr_avg; ++i) {
#pragma unroll
            for (int r = 0; r < 8; ++r) x[0][r] = fetch(i, r);
            fold_row(i, x[0], std::false_type{}, 0);
        }
    }

    // ---- per channel: apply V, FFT, x conj(R), FFT, statistics
    // (1 / N of the inverse transform rides on the same factor: a power of two, so the
    // magnitudes carry the same bits as when they are scaled at the end)
    const float sc = (1.0f / (float)P.corr_avg) * (1.0f / kFftN);
    // V(m) for m = t + 256 r (base phasor and seven steps of 256 positions) applied to the fold;
    // the replica spectrum is fetched at the same time so that its latency hides behind the FFT
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const float2 e1 = vtab[c][t & 15], e2 = vtab[c][16 + (t >> 4)], s2 = step[c];
        fft_c vm = cmulp(fft_c{e1.x, e1.y}, fft_c{e2.x, e2.y});
        const fft_c st256 = fft_c{s2.x, s2.y};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            acc[c][r] = cmulp(acc[c][r] * sc, vm);                    // (packed: two instructions each)
            vm = cmulp(vm, st256);
        }
    }
    auto prepare = [&](int c, const StreamChan& s, float2 (&v)[8], float2 (&rs)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = make_float2(acc[c][r].x, acc[c][r].y);
        const float2* R = rep + (size_t)s.prn * kFftN;
#pragma unroll
        for (int q = 0; q < 8; ++q) rs[q] = R[t + 256 * q];
    };
    auto times_conj = [&](float2 (&v)[8], const float2 (&rs)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const fft_c o = cmulp_conj(fft_c{v[q].x, v[q].y}, fft_c{rs[q].x, rs[q].y});   // conj(x) * r
            v[q] = make_float2(o.x, o.y);
        }
    };
    auto magnitudes = [&](const float2 (&v)[8], float (&mag)[8]) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            mag[q] = __builtin_amdgcn_sqrtf(v[q].x * v[q].x + v[q].y * v[q].y);   // v_sqrt_f32, 1 ulp
    };
    // one channel through the transforms (the two barriers of the previous statistics already
    // separate the last FFT reads from the writes of the next transform)
    auto single = [&](int c) {
        const StreamChan s = schan[c];
        float2 v[8], rs[8];
        prepare(c, s, v, rs);
        fft2048(v, lds, ftw, t);
        times_conj(v, rs);
        lds_barrier();
        fft2048(v, lds, ftw, t);
        float mag[8];
        magnitudes(v, mag);
        // mean / std / first-index argmax over the 2048 lags, neighbours of the peak
        int bi; float bv, mean, sd, elo, ehi;
        corr_stats8(mag, t, magbuf, red, bi, bv, mean, sd, elo, ehi);
        if (t == 0) fin[c] = CorrFin{bi, bv, mean, sd, elo, ehi};
    };
    if (diag_flag(P, 64)) {                          // diagnostics: no transforms
        float sm = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c)
#pragma unroll
            for (int r = 0; r < 8; ++r) sm += acc[c][r].x + acc[c][r].y;
        if (t < CG && schan[t].active)
            corr_finish(schan[t], CorrFin{0, sm, 0.f, 1.f, 0.f, 0.f}, P, delay_forced, out, mid);
        return;
    }
#pragma unroll
    for (int c = 0; c < CG; ++c)
        if (schan[c].active) single(c);                                   // uniform over the workgroup
    // the output records, one lane per channel (thread 0 wrote fin; the double-precision peak
    // fit runs once per workgroup instead of once per channel on the first wave)
    __syncthreads();
    if (t < CG && schan[t].active) corr_finish(schan[t], fin[t], P, delay_forced, out, mid);
}

}  // namespace gpsmi
