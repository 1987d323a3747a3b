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

__global__ __launch_bounds__(256) void trk_fold_general_kernel(
    const float2* __restrict__ iq, const float* __restrict__ t32,
    const gpsmi_trk_state* __restrict__ st_in, TrkParams P, float2* __restrict__ fold,
    int* __restrict__ xsel, int* __restrict__ rsel, JobMid* __restrict__ mid) {
    const int job = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
    const int cs = P.cs, b = job / P.nch;
    const gpsmi_trk_state& st = st_in[job];
    const int active = st.prn > 0;
    const float om = active ? (st.omega0 != 0.f ? st.omega0 : omega_of(st.freq)) : 0.f;
    const float ph = st.phase;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        JobMid md;
        md.delay_used = active ? st.delay : 0; md.active = active; md.prn = active ? st.prn : 0;
        md.om = om; md.ph = active ? ph : 0.f;
        md.pad[0] = md.pad[1] = md.pad[2] = 0;
        mid[job] = md;
        xsel[job] = job;
        rsel[job] = md.prn;
    }
    if (m >= cs) return;
    float2 acc = make_float2(0.f, 0.f);
    if (active) {
        const float2* blk = iq + (size_t)b * ((size_t)cs * P.n_cyc);
        const int first = (P.n_cyc - P.corr_avg) / 2;
        for (int i = first; i < first + P.corr_avg; ++i) {
            const int k = i * cs + m;
            const float2 y = wipe(blk[k], ph, om, t32[k]);
            acc.x += y.x; acc.y += y.y;
        }
        const float sc = 1.0f / (float)P.corr_avg;
        acc.x *= sc; acc.y *= sc;
    }
    fold[(size_t)job * cs + m] = acc;
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
