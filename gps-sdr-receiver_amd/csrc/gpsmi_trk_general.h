// Tracking at code lengths other than 2048 samples (BASELINE config 5:
// CODE_SAMPLES = 16368, N_CYC = 8).  The code-phase correlation of cacodeCorr
// (reference src/gpslib.py:1315-1327) is done in the time domain with the kernels
// of gpsmi_direct.h; these are the pieces around them:
//
//   trk_fold_general_kernel   carrier wipe-off (float32 phase argument as in
//                             demodDoppler :1343-1346) and mean of the centre
//                             corr_avg code periods; writes the job descriptor.
//   trk_decide_kernel         findCodePhase threshold, fitCodePhase and the DELAY
//                             the block is decoded with (:1293-1304, :1268-1290,
//                             :1181-1182), one thread per job.
//   trk_partial_reduce_kernel adds the per-chunk partial sums of the chunked
//                             correlator (gpsmi_trk_stream.h, GEN = true).
#pragma once
#include "gpsmi_direct.h"

namespace gpsmi {

// One workgroup per (256 positions, block).  The wipe-off separates as in gpsmi_trk_corr.h:
// for sample k = i cs + m, exp(-j(phase + w t[k])) = U[i] V(m) with U[i] = exp(-j w i T) uniform
// over a row and V(m) = exp(-j(phase + w (m + 1) / fs)), so a sample costs one complex
// multiply-accumulate per channel (instead of a sine and a cosine of a float32 argument of up
// to ~1000 rad) and V is applied once per position and channel.  The rows of a position are
// loaded ONCE, into registers, and serve every channel of the block (round 2 re-read them per
// group of four channels: 1.5 GB of reads for 0.5 GB of samples made the kernel HBM-bound at
// 0.45 ms per 512-block batch); the row factors of up to kGenFoldCh channels wait in LDS.
constexpr int kGenFoldCh = 16;       // channels per pass over the row-factor tables
constexpr int kGenFoldRows = 8;      // rows held in registers (CORR_AVG of the reference)

__global__ __launch_bounds__(256) void trk_fold_general_kernel(
    const float2* __restrict__ iq, const float* __restrict__ t32,
    const gpsmi_trk_state* __restrict__ st_in, TrkParams P, float2* __restrict__ fold,
    int* __restrict__ xsel, int* __restrict__ rsel, JobMid* __restrict__ mid, int b0) {
    // (b0: first block of this launch; the folded samples are written chunk-local, fold[(b - b0) ...],
    // so that a batch can go through the fold and the correlation in pieces that stay in the cache)
    (void)t32;
    __shared__ float2 urow[kGenFoldCh][32];              // corr_avg <= n_cyc <= 32
    __shared__ float s_om[kGenFoldCh], s_ph[kGenFoldCh];
    __shared__ int s_active[kGenFoldCh];
    const int b = b0 + blockIdx.y, t = threadIdx.x, m = blockIdx.x * 256 + t;
    const int cs = P.cs, nch = P.nch;
    const int first = (P.n_cyc - P.corr_avg) / 2;
    const double inv_2pi = 0.15915494309189533576888376337251;
    const float2* blk = iq + (size_t)b * ((size_t)cs * P.n_cyc) + (size_t)first * cs + (m < cs ? m : 0);
    const float sc = 1.0f / (float)P.corr_avg;
    const float tm = (float)(m + 1) / (1000.0f * (float)cs);              // (m + 1) / fs
    const bool in_regs = P.corr_avg == kGenFoldRows;
    float2 xr[kGenFoldRows];
    if (in_regs) {
#pragma unroll
        for (int i = 0; i < kGenFoldRows; ++i) xr[i] = blk[(size_t)i * cs];
    }
    for (int c0 = 0; c0 < nch; c0 += kGenFoldCh) {
        // ---- the pass's constants and job descriptors
        if (t < kGenFoldCh) {
            const int ch = c0 + t;
            int active = 0;
            float om = 0.f, ph = 0.f;
            if (ch < nch) {
                const int job = b * nch + ch;
                const gpsmi_trk_state& st = st_in[job];
                active = st.prn > 0;
                om = active ? (st.omega0 != 0.f ? st.omega0 : omega_of(st.freq)) : 0.f;
                ph = active ? st.phase : 0.f;
                if (blockIdx.x == 0) {
                    JobMid md;
                    md.delay_used = active ? st.delay : 0; md.active = active; md.prn = active ? st.prn : 0;
                    md.om = om; md.ph = ph;
                    md.pad[0] = md.pad[1] = md.pad[2] = 0;
                    mid[job] = md;
                    xsel[job] = job - b0 * nch;
                    rsel[job] = md.prn;
                }
            }
            s_om[t] = om; s_ph[t] = ph; s_active[t] = active;
        }
        __syncthreads();
        for (int e = t; e < kGenFoldCh * P.corr_avg; e += 256) {
            const int c = e / P.corr_avg, i = e % P.corr_avg;
            const double rev = (double)s_om[c] * inv_2pi * (double)(first + i) * 1.0e-3;   // w i T / 2 pi
            urow[c][i] = phasor_rev((float)(rev - rint(rev)));
        }
        __syncthreads();
        if (m < cs) {
            const int npass = nch - c0 < kGenFoldCh ? nch - c0 : kGenFoldCh;
            for (int c = 0; c < npass; ++c) {
                float2 acc = make_float2(0.f, 0.f);
                if (in_regs) {
#pragma unroll
                    for (int i = 0; i < kGenFoldRows; ++i) {
                        const float2 u = urow[c][i], x = xr[i];
                        acc.x = fmaf(u.x, x.x, fmaf(-u.y, x.y, acc.x));
                        acc.y = fmaf(u.x, x.y, fmaf(u.y, x.x, acc.y));
                    }
                } else {
                    for (int i = 0; i < P.corr_avg; ++i) {
                        const float2 u = urow[c][i], x = blk[(size_t)i * cs];
                        acc.x = fmaf(u.x, x.x, fmaf(-u.y, x.y, acc.x));
                        acc.y = fmaf(u.x, x.y, fmaf(u.y, x.x, acc.y));
                    }
                }
                float2 r = make_float2(0.f, 0.f);
                if (s_active[c]) {
                    const float f_eff = (float)((double)s_om[c] * inv_2pi);
                    const float2 v = phasor_rev(fmaf(f_eff, tm, s_ph[c] * (float)inv_2pi));
                    r = cmulf(make_float2(acc.x * sc, acc.y * sc), v);
                }
                fold[(size_t)((b - b0) * nch + c0 + c) * cs + m] = r;
            }
        }
        __syncthreads();                                   // (the tables are rewritten for the next pass)
    }
}

__global__ void trk_decide_kernel(const DirStats* __restrict__ stats,
                                  const int* __restrict__ delay_forced, TrkParams P, int njobs,
                                  gpsmi_trk_out* __restrict__ out, JobMid* __restrict__ mid) {
    const int job = blockIdx.x * blockDim.x + threadIdx.x;
    if (job >= njobs) return;
    JobMid md = mid[job];
    if (!md.active) return;
    const DirStats s = stats[job];
    const float norm = (s.peak - s.mean) / s.std;
    gpsmi_trk_out& o = out[job];
    o.prn = md.prn;
    o.mx = s.argmax;
    o.epl[0] = s.lo; o.epl[1] = s.peak; o.epl[2] = s.hi;
    o.corr_mean = s.mean; o.corr_std = s.std;
    o.norm_max_corr = norm;
    int delay = -1;
    double cp = -1.0;
    if (norm > P.corr_min) {
        delay = s.argmax;
        cp = fit_code_phase((double)s.lo, (double)s.peak, (double)s.hi, s.argmax);
    }
    o.delay = delay;
    o.reserved0 = 0;
    o.code_phase = cp;
    int used = delay >= 0 ? delay : md.delay_used;       // the state's DELAY otherwise
    if (delay_forced && delay_forced[job] >= 0) used = delay_forced[job];
    o.delay_used = used;
    md.delay_used = used;
    mid[job] = md;
}

__global__ void trk_partial_reduce_kernel(const float2* __restrict__ pg, int nchunks, int per_job,
                                          int njobs, const JobMid* __restrict__ mid,
                                          float2* __restrict__ partial) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= njobs * per_job) return;
    const int job = i / per_job, o = i % per_job;
    if (!mid[job].active) return;
    float sx = 0.f, sy = 0.f;
    for (int c = 0; c < nchunks; ++c) {
        const float2 v = pg[((size_t)job * nchunks + c) * per_job + o];
        sx += v.x; sy += v.y;
    }
    partial[i] = make_float2(sx, sy);
}

}  // namespace gpsmi
