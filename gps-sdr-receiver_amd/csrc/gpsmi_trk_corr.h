// Code-phase correlation of the tracking loop: cacodeCorr + findCodePhase +
// fitCodePhase (reference src/gpslib.py:1315-1327, :1293-1304, :1268-1290) for
// up to CG (1, 2, 4 or 6) channels of one block per workgroup.
//
// The reference wipes the carrier off the centre corr_avg code periods, sums
// their FFTs, multiplies by conj(FFT(replica)) and takes |ifft|.  Here the sum
// of FFTs is the FFT of the sum, and the wipe-off separates exactly as in the
// correlator (gpsmi_trk_stream.h): for sample k = i*CS + m
//     exp(-j(phase + w (k+1)/fs)) = V(m) * U[i],  U[i] = exp(-j w i T),
// so fold[m] = V(m) * sum_i U[i] x[i][m]: one complex multiply-accumulate per
// channel-sample with a wave-uniform U, every sample loaded once for the six
// channels; V is applied once per position.  Then, per channel, two passes of
// the LDS-resident 2048-point FFT (the second one as the inverse), |.|, mean /
// population std / first-index argmax, the two neighbours of the peak, and
// thread 0 applies the CORR_MIN threshold, the peak fit and chooses the DELAY
// the block is decoded with (gpslib.py:1181-1182).
#pragma once
#include "gpsmi_fft.h"
#include "gpsmi_stats.h"

namespace gpsmi {

// (FMT 1: iq holds raw uint16 (Q << 8 | I) samples, decoded on load exactly as
// gpsmi_dev_unpack_u8iq does)
template <int CG, int FMT = 0>
__global__ __launch_bounds__(256) void trk_corr_kernel(
    const void* __restrict__ iq, const gpsmi_trk_state* __restrict__ st_in,
    const int* __restrict__ delay_forced, const float2* __restrict__ rep,
    const float2* __restrict__ tw, TrkParams P, int ngroups, int nblocks,
    gpsmi_trk_out* __restrict__ out, JobMid* __restrict__ mid) {
    __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
    __shared__ __attribute__((aligned(16))) float lds_tw[kFftTwFloats];
    __shared__ float red[kStatsRedFloats];
    // Two small arrays live inside the second FFT buffer, which a transform leaves free when it
    // returns and does not write before its first barrier: the magnitudes of the statistics
    // (8 KiB) and the row factors of the fold (1 KiB, used before any transform).  40.1 KiB of
    // LDS per workgroup instead of 49.3, so LDS no longer limits the kernel to three workgroups
    // per CU; the registers still do for CG = 4 (166 VGPRs), and CG = 2 (126 VGPRs, four
    // workgroups per CU, the fold's rows read six times per block instead of three) measured
    // 3-4 % slower per batch.
    float* magbuf = lds + 2 * kFftPlane;
    float2 (*urow)[32] = reinterpret_cast<float2 (*)[32]>(lds + 2 * kFftPlane);   // U[c][i], i = row
    static_assert(kFftN <= 2 * kFftPlane1 && CG * 32 * 2 <= 2 * kFftPlane1, "aliases must fit buffer 1");
    __shared__ float2 step[CG];                  // exp(-j w 256/fs)
    __shared__ StreamChan schan[CG];

    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int g = slot % ngroups;
    const int b = (slot / ngroups) * 8 + xcd;
    if (b >= nblocks) return;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int cs = kFftN;
    const float2* blk = static_cast<const float2*>(iq) + (size_t)b * ((size_t)cs * P.n_cyc);
    const uint16_t* rblk = static_cast<const uint16_t*>(iq) + (size_t)b * ((size_t)cs * P.n_cyc);
    const double inv_2pi = 0.15915494309189533576888376337251;
    const int first = (P.n_cyc - P.corr_avg) / 2;
    const FftTw ftw = fft_setup(lds_tw, tw, t);

    // ---- per-channel constants (one thread each), then one barrier
    if (t < CG * 33) {
        const int c = t / 33, i = t % 33;
        const int cidx = g * CG + c;
        const int job = b * P.nch + cidx;
        StreamChan s;
        s.job = job; s.active = 0; s.om = 0.f; s.ph = 0.f; s.d = 0; s.prn = 0;
        if (cidx < P.nch && st_in[job].prn > 0) {
            const gpsmi_trk_state& st = st_in[job];
            s.active = 1;
            s.om = st.omega0 != 0.f ? st.omega0 : omega_of(st.freq);
            s.ph = st.phase;
            s.d = st.delay;
            s.prn = st.prn;
        }
        const double f_eff = (double)s.om * inv_2pi;
        if (i < 32) {
            const double rev = f_eff * (double)i * 1.0e-3;            // w i T / 2 pi
            urow[c][i] = phasor_rev((float)(rev - rint(rev)));
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
    __syncthreads();

    // ---- fold: acc[c][r] = sum_i U[c][i] x[i][t + 256 r]
    v2f acc[CG][8];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[c][r] = v2f{0.f, 0.f};
    for (int i = first; i < first + P.corr_avg; ++i) {
        v2f x[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (FMT == 0) {
                const float2 v = blk[(size_t)i * cs + t + 256 * r];
                x[r] = v2f{v.x, v.y};
            } else {
                const unsigned v = rblk[(size_t)i * cs + t + 256 * r];
                const float scl = 1.0f / 127.5f;
                x[r] = v2f{sub_rn(mul_rn((float)(v & 0xFF), scl), 1.0f), sub_rn(mul_rn((float)(v >> 8), scl), 1.0f)};
            }
        }
        v2f u[CG];
#pragma unroll
        for (int c = 0; c < CG; ++c) u[c] = v2f{urow[c][i].x, urow[c][i].y};
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (CG == 1) {                       // (cmac2 works on pairs: two positions at once)
                if (r & 1) cmac2(acc[0][r - 1], acc[0][r], u[0], x[r - 1], u[0], x[r]);
            } else {
#pragma unroll
                for (int c = 0; c + 1 < CG; c += 2)
                    cmac2(acc[c][r], acc[c + 1][r], u[c], x[r], u[c + 1], x[r]);
            }
        }
    }

    // ---- per channel: apply V, FFT, x conj(R), FFT, statistics
    const float inv_fs = 1.0f / (1000.0f * (float)cs);
    const float sc = 1.0f / (float)P.corr_avg;
#pragma unroll
    for (int c = 0; c < CG; ++c) {
        const StreamChan s = schan[c];
        if (!s.active) continue;                                      // uniform over the workgroup
        // V(m) for m = t + 256 r: base phasor and seven steps of 256 positions
        const float f_eff = (float)((double)s.om * inv_2pi);
        const float rev0 = fmaf(f_eff, (float)(t + 1) * inv_fs, s.ph * (float)inv_2pi);
        float2 vm = phasor_rev(rev0);
        const float2 st256 = step[c];
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const float2 a = make_float2(acc[c][r].x * sc, acc[c][r].y * sc);
            v[r] = cmulf(a, vm);
            vm = cmulf(vm, st256);
        }
        // the replica spectrum is fetched now so that its latency hides behind the FFT
        const float2* R = rep + (size_t)s.prn * kFftN;
        float2 rs[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) rs[q] = R[t + 256 * q];
        // (the two barriers of the previous channel's statistics already separate its
        // last FFT reads from the writes below)
        fft2048(v, lds, ftw, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 x = v[q], r = rs[q];
            v[q] = make_float2(x.x * r.x + x.y * r.y, x.x * r.y - x.y * r.x);   // conj(x) * r
        }
        __syncthreads();
        fft2048(v, lds, ftw, t);
        float mag[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            mag[q] = sqrtf(v[q].x * v[q].x + v[q].y * v[q].y) * (1.0f / kFftN);

        // mean / std / first-index argmax over the 2048 lags, neighbours of the peak
        int bi; float bv, mean, sd, elo, ehi;
        corr_stats8(mag, t, magbuf, red, bi, bv, mean, sd, elo, ehi);
        if (t == 0) {
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
    }
}

}  // namespace gpsmi
