// Acquisition: batched per-SV x Doppler parallel code-phase search on gfx950.
//
// Replaces the array arithmetic of reference src/gpsrecv.py:241-274.
//
//   acq_spectrum_kernel   one workgroup per Doppler bin: carrier wipe-off with
//                         the reference's float32 phase argument
//                         (gpsrecv.py:232-235), fold of the n_avg code periods
//                         (sum of FFTs = FFT of the sum, :250-254), 2048-point
//                         FFT in LDS, spectrum to a small L2-resident scratch.
//   acq_corr_kernel       one workgroup per (SV, bin): conj(X) * R from
//                         coalesced reads of the replica spectra, the same FFT
//                         as the inverse (|ifft(Y)| = |fft(conj Y)| / N, :258),
//                         |.|, then mean / population std / first-index argmax
//                         (findCodePhase, :217-223) by wave64 shuffles.  The
//                         nbins x nsv x 2048 correlation surface never reaches
//                         HBM; 16 bytes per cell do.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gpsmi_common.h"
#include "gpsmi_bigfft.h"
#include "gpsmi_pfa.h"
#include "gpsmi_direct.h"
#include "gpsmi_fft.h"
#include "gpsmi_stats.h"

namespace gpsmi {

// ---- kernels ---------------------------------------------------------------
// G = 1: 256 threads, the code periods folded one after the other.  G = 4 (long coherent
// searches): 1024 threads, the periods dealt round-robin to four groups of 256 whose partial
// folds meet in LDS (the sine / cosine per sample is what this kernel spends its time on);
// group 0 then adds them in group order and transforms.
// (FMT 1: iq holds the recorder's raw uint16 samples, decoded on load: gpsmi_acq_set_input_format)
template <int G, int FMT = 0>
__global__ __launch_bounds__(256 * G) void acq_spectrum_kernel(
    const void* __restrict__ iq, const float* __restrict__ t32,
    const float* __restrict__ omega, int n_avg, float2* __restrict__ spectra,
    const float2* __restrict__ tw) {
    __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
    __shared__ __attribute__((aligned(16))) float lds_tw[kFftTwFloats];
    __shared__ float2 part[G > 1 ? G - 1 : 1][G > 1 ? kFftN : 1];
    const int t = threadIdx.x & 255, grp = threadIdx.x >> 8, bin = blockIdx.x;
    const FftTw ftw = fft_setup(lds_tw, tw, t);           // (every group writes the same tables)
    const float om = omega[bin];
    float2 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = make_float2(0.f, 0.f);
    for (int i = grp; i < n_avg; i += G) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int k = i * kFftN + t + 256 * r;
            float2 x = load_iq<FMT>(iq, k);
            float p = mul_rn(om, t32[k]);      // float32 phase argument, phase0 = 0
            float s, c;
            sincosf(p, &s, &c);
            // factor = (c, -s); factor * x as numpy multiplies complex64
            v[r].x += c * x.x + s * x.y;
            v[r].y += c * x.y - s * x.x;
        }
    }
    if (G > 1) {
        if (grp > 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) part[grp - 1][t + 256 * r] = v[r];
        }
        __syncthreads();
        if (grp > 0) return;                   // (a wave that has ended no longer counts at a barrier)
#pragma unroll
        for (int g = 1; g < G; ++g)
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const float2 o = part[g - 1][t + 256 * r];
                v[r].x += o.x; v[r].y += o.y;
            }
    }
    __syncthreads();
    fft2048(v, lds, ftw, t);
    const float sc = 1.0f / (float)n_avg;
    float2* out = spectra + (size_t)bin * kFftN;
#pragma unroll
    for (int q = 0; q < 8; ++q) out[t + 256 * q] = make_float2(v[q].x * sc, v[q].y * sc);
}

__global__ __launch_bounds__(256) void acq_corr_kernel(
    const float2* __restrict__ spectra, const float2* __restrict__ rep,
    const int* __restrict__ slot, gpsmi_peak* __restrict__ out, int nsv,
    const float2* __restrict__ tw, float2* __restrict__ nbr) {
    __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
    __shared__ float red[kStatsRedFloats];
    __shared__ __attribute__((aligned(16))) float lds_tw[kFftTwFloats];
    // the statistics' copy of the magnitudes lives in the second FFT buffer, which the transform
    // leaves free when it returns: 40.4 KiB of LDS, four workgroups per CU instead of three
    float* magbuf = lds + 2 * kFftPlane;
    static_assert(kFftN <= 2 * kFftPlane1, "the alias must fit buffer 1");
    const int t = threadIdx.x, sv = blockIdx.x, bin = blockIdx.y;
    const FftTw ftw = fft_setup(lds_tw, tw, t);
    const float2* X = spectra + (size_t)bin * kFftN;
    const float2* R = rep + (size_t)slot[sv] * kFftN;
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float2 x = X[t + 256 * q], r = R[t + 256 * q];
        v[q] = make_float2(x.x * r.x + x.y * r.y, x.x * r.y - x.y * r.x);   // conj(x) * r
    }
    __syncthreads();
    fft2048(v, lds, ftw, t);
    float mag[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mag[q] = __builtin_amdgcn_sqrtf(v[q].x * v[q].x + v[q].y * v[q].y) * (1.0f / kFftN);   // v_sqrt_f32, 1 ulp
    int amax; float peak, mean, sd, lo, hi;
    corr_stats8(mag, t, magbuf, red, amax, peak, mean, sd, lo, hi);
    if (t == 0) {
        // fft(conj Y)[n] = conj(N ifft(Y)[n]): same lag index, no reversal
        gpsmi_peak p; p.argmax = amax; p.peak = peak; p.mean = mean; p.std = sd;
        out[(size_t)bin * nsv + sv] = p;
        if (nbr) nbr[(size_t)bin * nsv + sv] = make_float2(lo, hi);   // circular neighbours
    }
}

// ---- general code length: wipe-off + fold in the time domain ----------------
// x[bin][m] = (1/n_avg) sum_i iq[i L + m] exp(-j fl32(om t32[i L + m]))
template <int FMT = 0>
__global__ __launch_bounds__(256) void acq_fold_kernel(
    const void* __restrict__ iq, const float* __restrict__ t32,
    const float* __restrict__ omega, int n_avg, int L, float2* __restrict__ xout) {
    const int m = blockIdx.x * 256 + threadIdx.x, bin = blockIdx.y;
    if (m >= L) return;
    const float om = omega[bin];
    float ar = 0.f, ai = 0.f;
    for (int i = 0; i < n_avg; ++i) {
        const int k = i * L + m;
        const float2 v = load_iq<FMT>(iq, k);
        float sn, co;
        sincosf(mul_rn(om, t32[k]), &sn, &co);
        ar += co * v.x + sn * v.y;
        ai += co * v.y - sn * v.x;
    }
    const float sc = 1.0f / (float)n_avg;
    xout[(size_t)bin * L + m] = make_float2(ar * sc, ai * sc);
}

__global__ void acq_cells_kernel(int* __restrict__ xsel, int* __restrict__ rsel,
                                 const int* __restrict__ slot, int nsv, int ncell) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    xsel[c] = c / nsv;                       // the bin's folded block
    rsel[c] = slot[c % nsv];                 // the SV's replica
}

__global__ void acq_peaks_kernel(const DirStats* __restrict__ st, gpsmi_peak* __restrict__ out,
                                 float2* __restrict__ nbr, int ncell) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    gpsmi_peak p;
    p.argmax = st[c].argmax; p.peak = st[c].peak; p.mean = st[c].mean; p.std = st[c].std;
    out[c] = p;
    if (nbr) nbr[c] = make_float2(st[c].lo, st[c].hi);
}

}  // namespace gpsmi

using namespace gpsmi;

struct gpsmi_acq;
namespace gpsmi { HandleSync trk_sync(gpsmi_trk* h); }

struct gpsmi_acq {
    gpsmi_cfg cfg;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t order = nullptr;        // orders other handles' streams behind this one
    float2* d_tw = nullptr;
    float* d_t32 = nullptr;
    float2* d_rep = nullptr;          // [GPSMI_MAX_PRN + 1][cs]
    bool have_rep[GPSMI_MAX_PRN + 1] = {};
    float2* d_iq = nullptr;  size_t iq_cap = 0;
    float2* d_spec = nullptr; size_t spec_cap = 0;   // bins
    float* d_omega = nullptr; int* d_slot = nullptr; gpsmi_peak* d_peaks = nullptr;
    size_t cell_cap = 0;
    float2* d_nbr = nullptr;
    // page-locked staging of the per-call parameters (two sets: a search can be enqueued
    // while the previous one still runs; no blocking copy, no use of the null stream)
    float* h_om[2] = {nullptr, nullptr};
    int32_t* h_slot[2] = {nullptr, nullptr};
    hipEvent_t staged[2] = {nullptr, nullptr};
    bool staged_used[2] = {false, false};
    int stage = 0;
    // direct (time-domain) path for code_samples != 2048
    bool direct = false;
    float* d_rep_time = nullptr;            // [GPSMI_MAX_PRN + 1][cs]
    bool have_time[GPSMI_MAX_PRN + 1] = {};
    float2* d_fold = nullptr; float* d_mag = nullptr; DirStats* d_stats = nullptr;
    int* d_xsel = nullptr; int* d_rsel = nullptr;
    size_t dir_bins = 0, dir_cells = 0;
    // ... through one 32768-point FFT pair when the code period fits (gpsmi_bigfft.h)
    bool big = false;
    float2* d_twN = nullptr; float2* d_RS = nullptr; float2* d_S = nullptr;
    // ... or natively in LDS when the code period is 16368 = 16 * 3 * 11 * 31 samples (gpsmi_pfa.h)
    bool pfa = false;
    float2* d_RSp = nullptr;                // [GPSMI_MAX_PRN + 1][16368] replica spectra, P3's order
    int iq_fmt = GPSMI_IQ_C64;              // what the iq pointers of the search calls point to
    float last_ms = 0.f;
    bool pending = false;
};

extern "C" int gpsmi_acq_wait(gpsmi_acq* h);

namespace gpsmi {
HandleSync acq_sync(gpsmi_acq* h) { return HandleSync{h->stream, h->order, h->cfg.device, nullptr}; }
}  // namespace gpsmi

static int acq_reserve(gpsmi_acq* h, int nbins, int nsv) {
    if ((size_t)nbins > h->spec_cap) {
        if (h->d_spec) GPSMI_HIP(hipFree(h->d_spec));
        if (h->d_omega) GPSMI_HIP(hipFree(h->d_omega));
        h->d_spec = nullptr; h->d_omega = nullptr; h->spec_cap = 0;
        GPSMI_HIP(hipMalloc((void**)&h->d_spec, (size_t)nbins * kFftN * sizeof(float2)));
        GPSMI_HIP(hipMalloc((void**)&h->d_omega, (size_t)nbins * sizeof(float)));
        h->spec_cap = nbins;
    }
    size_t cells = (size_t)nbins * nsv;
    if (cells > h->cell_cap) {
        if (h->d_peaks) GPSMI_HIP(hipFree(h->d_peaks));
        if (h->d_nbr) GPSMI_HIP(hipFree(h->d_nbr));
        h->d_peaks = nullptr; h->d_nbr = nullptr; h->cell_cap = 0;
        GPSMI_HIP(hipMalloc((void**)&h->d_peaks, cells * sizeof(gpsmi_peak)));
        GPSMI_HIP(hipMalloc((void**)&h->d_nbr, cells * sizeof(float2)));
        h->cell_cap = cells;
    }
    return GPSMI_OK;
}

extern "C" {

static int acq_build(const gpsmi_cfg* cfg, gpsmi_acq* h);

int gpsmi_acq_create(const gpsmi_cfg* cfg, gpsmi_acq** out) {
    GPSMI_REQUIRE(cfg && out, "null argument");
    *out = nullptr;
    GPSMI_REQUIRE(cfg->code_samples >= 1024 && cfg->code_samples <= 65536 &&
                      cfg->code_samples % 16 == 0,
                  "code_samples must be a multiple of 16 in 1024..65536");
    GPSMI_REQUIRE(cfg->n_cyc >= 1 && cfg->n_cyc <= 64, "n_cyc out of range");
    GPSMI_HIP(hipSetDevice(cfg->device));
    gpsmi_acq* h = new (std::nothrow) gpsmi_acq();
    if (!h) return fail(GPSMI_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    const int rc = acq_build(cfg, h);
    if (rc) {                       // nothing half-built leaves this function
        (void)gpsmi_acq_destroy(h);
        return rc;
    }
    *out = h;
    return GPSMI_OK;
}

}  // extern "C"

static int acq_build(const gpsmi_cfg* cfg, gpsmi_acq* h) {
    h->direct = cfg->code_samples != kFftN;
    GPSMI_HIP(hipStreamCreate(&h->stream));
    GPSMI_HIP(hipEventCreateWithFlags(&h->order, hipEventDisableTiming));
    GPSMI_HIP(hipEventCreate(&h->ev0));
    GPSMI_HIP(hipEventCreate(&h->ev1));
    std::vector<float2> tw;
    make_twiddles(tw);
    GPSMI_HIP(hipMalloc((void**)&h->d_tw, tw.size() * sizeof(float2)));
    GPSMI_HIP(hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    // SEC_TIME (gpsrecv.py:32-33): float32(k+1) / SAMPLE_RATE in float32
    const int ngps = cfg->n_cyc * cfg->code_samples;
    const float fs = (float)(1000 * cfg->code_samples);
    std::vector<float> t32(ngps);
    for (int k = 0; k < ngps; ++k) t32[k] = (float)(k + 1) / fs;
    GPSMI_HIP(hipMalloc((void**)&h->d_t32, ngps * sizeof(float)));
    GPSMI_HIP(hipMemcpy(h->d_t32, t32.data(), ngps * sizeof(float), hipMemcpyHostToDevice));
    GPSMI_HIP(hipMalloc((void**)&h->d_rep, (size_t)(GPSMI_MAX_PRN + 1) * kFftN * sizeof(float2)));
    GPSMI_HIP(hipMalloc((void**)&h->d_slot, (GPSMI_MAX_PRN + 1) * sizeof(int)));
    for (int k = 0; k < 2; ++k) {
        GPSMI_HIP(hipHostMalloc((void**)&h->h_om[k], 65536 * sizeof(float), hipHostMallocDefault));
        GPSMI_HIP(hipHostMalloc((void**)&h->h_slot[k], (GPSMI_MAX_PRN + 1) * sizeof(int32_t),
                                hipHostMallocDefault));
        GPSMI_HIP(hipEventCreateWithFlags(&h->staged[k], hipEventDisableTiming));
    }
    if (h->direct) {
        GPSMI_HIP(hipMalloc((void**)&h->d_rep_time,
                            (size_t)(GPSMI_MAX_PRN + 1) * cfg->code_samples * sizeof(float)));
        long long forced = 0;                    // option "codephase": 1 keeps the time-domain kernel,
        default_opt("codephase", &forced, 0);    // 2 the zero-padded 32768-point pair
        h->pfa = cfg->code_samples == kPfaL && forced == 0;
        h->big = !h->pfa && 2 * cfg->code_samples - 1 <= kBigN && forced != 1;
        if (h->pfa) {
            const size_t b = (size_t)(GPSMI_MAX_PRN + 1) * kPfaL * sizeof(float2);
            GPSMI_HIP(hipMalloc((void**)&h->d_RSp, b));
            GPSMI_HIP(hipMemset(h->d_RSp, 0, b));
        }
        if (h->big) {
            std::vector<float2> twn(kBigN);
            for (int k = 0; k < kBigN; ++k) {
                const double a = -2.0 * M_PI * (double)k / (double)kBigN;
                twn[k] = make_float2((float)cos(a), (float)sin(a));
            }
            GPSMI_HIP(hipMalloc((void**)&h->d_twN, kBigN * sizeof(float2)));
            GPSMI_HIP(hipMemcpy(h->d_twN, twn.data(), kBigN * sizeof(float2), hipMemcpyHostToDevice));
            GPSMI_HIP(hipMalloc((void**)&h->d_RS, (size_t)(GPSMI_MAX_PRN + 1) * kBigN * sizeof(float2)));
            GPSMI_HIP(hipMalloc((void**)&h->d_S, (size_t)kBigChunkCells * kBigN * sizeof(float2)));
        }
    }
    return GPSMI_OK;
}

extern "C" {

int gpsmi_acq_destroy(gpsmi_acq* h) {
    if (!h) return GPSMI_OK;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* bufs[] = {h->d_tw, h->d_t32, h->d_rep, h->d_iq, h->d_spec, h->d_omega, h->d_slot,
                    h->d_peaks, h->d_nbr, h->d_rep_time, h->d_fold, h->d_mag, h->d_stats,
                    h->d_xsel, h->d_rsel, h->d_twN, h->d_RS, h->d_S, h->d_RSp};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (int k = 0; k < 2; ++k) {
        if (h->h_om[k]) (void)hipHostFree(h->h_om[k]);
        if (h->h_slot[k]) (void)hipHostFree(h->h_slot[k]);
        if (h->staged[k]) (void)hipEventDestroy(h->staged[k]);
    }
    if (h->order) (void)hipEventDestroy(h->order);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GPSMI_OK;
}

int gpsmi_acq_set_replica_time(gpsmi_acq* h, int prn, const float* replica) {
    GPSMI_REQUIRE(h && replica, "null argument");
    GPSMI_REQUIRE(prn >= 1 && prn <= GPSMI_MAX_PRN, "prn out of range 1..37");
    if (!h->direct) return GPSMI_OK;         // the FFT path has no use for it
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    GPSMI_HIP(hipMemcpy(h->d_rep_time + (size_t)prn * h->cfg.code_samples, replica,
                        (size_t)h->cfg.code_samples * sizeof(float), hipMemcpyHostToDevice));
    if (h->big)
        big_replica_launch(h->stream, h->d_rep_time, prn, h->cfg.code_samples, h->d_RS, h->d_tw,
                           h->d_twN);
    if (h->pfa) pfa_replica_launch(h->stream, h->d_rep_time, prn, h->d_RSp);
    if (h->big || h->pfa) {
        GPSMI_HIP(hipGetLastError());
        GPSMI_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_time[prn] = true;
    return GPSMI_OK;
}

int gpsmi_acq_set_replica(gpsmi_acq* h, int prn, const float* spectrum) {
    GPSMI_REQUIRE(h && spectrum, "null argument");
    GPSMI_REQUIRE(prn >= 1 && prn <= GPSMI_MAX_PRN, "prn out of range 1..37");
    if (h->direct) return fail(GPSMI_E_STATE, "code_samples != 2048: use gpsmi_acq_set_replica_time");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    GPSMI_HIP(hipMemcpy(h->d_rep + (size_t)prn * kFftN, spectrum, kFftN * sizeof(float2),
                        hipMemcpyHostToDevice));
    h->have_rep[prn] = true;
    return GPSMI_OK;
}

static int acq_search_impl(gpsmi_acq* h, const void* d_iq, size_t n, const int32_t* prn, int nsv,
                           const double* freqs, int nbins, int n_avg, gpsmi_peak* out,
                           void* out_dev, float* nbr, bool wait = true) {
    GPSMI_REQUIRE(h && d_iq && prn && freqs, "null argument");
    GPSMI_REQUIRE(out || out_dev, "no output requested");
    GPSMI_REQUIRE(nsv >= 0 && nsv <= GPSMI_MAX_PRN, "nsv out of range");
    GPSMI_REQUIRE(nbins >= 0 && nbins <= 65535, "nbins out of range");
    GPSMI_REQUIRE(n_avg >= 1 && n_avg <= h->cfg.n_cyc, "n_avg out of range 1..n_cyc");
    const int cs = h->cfg.code_samples;
    GPSMI_REQUIRE(n >= (size_t)n_avg * cs, "iq shorter than n_avg code periods");
    for (int i = 0; i < nsv; ++i) {
        GPSMI_REQUIRE(prn[i] >= 1 && prn[i] <= GPSMI_MAX_PRN, "prn out of range 1..37");
        if (!(h->direct ? h->have_time[prn[i]] : h->have_rep[prn[i]]))
            return fail(GPSMI_E_STATE, "no replica set for PRN %d", (int)prn[i]);
    }
    h->last_ms = 0.f;
    if (nsv == 0 || nbins == 0) return GPSMI_OK;            // empty search: nothing to do
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = acq_reserve(h, nbins, nsv);
    if (rc) return rc;
    // parameter uploads: the caller's arrays are consumed before this function returns
    // (copied into page-locked staging), the device copies are ordered on the stream
    const int sg = h->stage ^= 1;
    if (h->staged_used[sg]) GPSMI_HIP(hipEventSynchronize(h->staged[sg]));
    for (int b = 0; b < nbins; ++b) h->h_om[sg][b] = (float)(2.0 * M_PI * freqs[b]);
    for (int i = 0; i < nsv; ++i) h->h_slot[sg][i] = prn[i];
    GPSMI_HIP(hipMemcpyAsync(h->d_omega, h->h_om[sg], nbins * sizeof(float), hipMemcpyHostToDevice,
                             h->stream));
    GPSMI_HIP(hipMemcpyAsync(h->d_slot, h->h_slot[sg], nsv * sizeof(int), hipMemcpyHostToDevice,
                             h->stream));
    GPSMI_HIP(hipEventRecord(h->staged[sg], h->stream));
    h->staged_used[sg] = true;
    if (h->direct) {
        const size_t cells = (size_t)nbins * nsv;
        if ((size_t)nbins > h->dir_bins) {
            if (h->d_fold) GPSMI_HIP(hipFree(h->d_fold));
            h->d_fold = nullptr; h->dir_bins = 0;
            GPSMI_HIP(hipMalloc((void**)&h->d_fold, (size_t)nbins * cs * sizeof(float2)));
            h->dir_bins = nbins;
        }
        if (cells > h->dir_cells) {
            void* olds[] = {h->d_mag, h->d_stats, h->d_xsel, h->d_rsel};
            for (void* p : olds)
                if (p) GPSMI_HIP(hipFree(p));
            h->d_mag = nullptr; h->d_stats = nullptr; h->d_xsel = h->d_rsel = nullptr;
            h->dir_cells = 0;
            if (!h->pfa) GPSMI_HIP(hipMalloc((void**)&h->d_mag, cells * cs * sizeof(float)));
            GPSMI_HIP(hipMalloc((void**)&h->d_stats, cells * sizeof(DirStats)));
            GPSMI_HIP(hipMalloc((void**)&h->d_xsel, cells * sizeof(int)));
            GPSMI_HIP(hipMalloc((void**)&h->d_rsel, cells * sizeof(int)));
            h->dir_cells = cells;
        }
    }
    GPSMI_HIP(hipEventRecord(h->ev0, h->stream));
    if (h->direct) {
        const int ncell = nbins * nsv;
        if (h->iq_fmt == GPSMI_IQ_U8)
            hipLaunchKernelGGL(acq_fold_kernel<1>, dim3((cs + 255) / 256, nbins), dim3(256), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, cs, h->d_fold);
        else
            hipLaunchKernelGGL(acq_fold_kernel<0>, dim3((cs + 255) / 256, nbins), dim3(256), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, cs, h->d_fold);
        hipLaunchKernelGGL(acq_cells_kernel, dim3((ncell + 255) / 256), dim3(256), 0, h->stream,
                           h->d_xsel, h->d_rsel, h->d_slot, nsv, ncell);
        if (h->pfa)              // transform, product, transform and statistics in one launch
            pfa_corr_launch(h->stream, h->d_fold, h->d_xsel, h->d_rsel, ncell, h->d_RSp, h->d_stats);
        else if (h->big)
            big_corr_launch(h->stream, h->d_fold, h->d_xsel, h->d_rsel, ncell, cs, h->d_RS, h->d_S,
                            h->d_tw, h->d_twN, h->d_mag);
        else
            hipLaunchKernelGGL(circ_corr_direct_kernel,
                               dim3((cs + kDirLagsPerWg - 1) / kDirLagsPerWg, ncell), dim3(256), 0,
                               h->stream, h->d_fold, h->d_rep_time, h->d_xsel, h->d_rsel, cs,
                               h->d_mag);
        if (!h->pfa)
            hipLaunchKernelGGL(corr_stats_kernel, dim3(ncell), dim3(256), 0, h->stream, h->d_mag, cs,
                               h->d_stats);
        hipLaunchKernelGGL(acq_peaks_kernel, dim3((ncell + 255) / 256), dim3(256), 0, h->stream,
                           h->d_stats, h->d_peaks, nbr ? h->d_nbr : nullptr, ncell);
    } else {
        const bool u8 = h->iq_fmt == GPSMI_IQ_U8;
        if (n_avg >= 4 && u8)
            hipLaunchKernelGGL((acq_spectrum_kernel<4, 1>), dim3(nbins), dim3(1024), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, h->d_spec, h->d_tw);
        else if (n_avg >= 4)
            hipLaunchKernelGGL((acq_spectrum_kernel<4, 0>), dim3(nbins), dim3(1024), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, h->d_spec, h->d_tw);
        else if (u8)
            hipLaunchKernelGGL((acq_spectrum_kernel<1, 1>), dim3(nbins), dim3(256), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, h->d_spec, h->d_tw);
        else
            hipLaunchKernelGGL((acq_spectrum_kernel<1, 0>), dim3(nbins), dim3(256), 0, h->stream,
                               d_iq, h->d_t32, h->d_omega, n_avg, h->d_spec, h->d_tw);
        hipLaunchKernelGGL(acq_corr_kernel, dim3(nsv, nbins), dim3(256), 0, h->stream, h->d_spec,
                           h->d_rep, h->d_slot, h->d_peaks, nsv, h->d_tw,
                           nbr ? h->d_nbr : nullptr);
    }
    GPSMI_HIP(hipGetLastError());
    GPSMI_HIP(hipEventRecord(h->ev1, h->stream));
    size_t bytes = (size_t)nbins * nsv * sizeof(gpsmi_peak);
    if (out_dev)
        GPSMI_HIP(hipMemcpyAsync(out_dev, h->d_peaks, bytes, hipMemcpyDeviceToDevice, h->stream));
    if (out)
        GPSMI_HIP(hipMemcpyAsync(out, h->d_peaks, bytes, hipMemcpyDeviceToHost, h->stream));
    if (nbr)
        GPSMI_HIP(hipMemcpyAsync(nbr, h->d_nbr, (size_t)nbins * nsv * sizeof(float2),
                                 hipMemcpyDeviceToHost, h->stream));
    h->pending = true;
    if (!wait) return GPSMI_OK;
    return gpsmi_acq_wait(h);
}

int gpsmi_acq_wait(gpsmi_acq* h) {
    GPSMI_REQUIRE(h, "null handle");
    if (!h->pending) return GPSMI_OK;
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    GPSMI_HIP(hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    h->pending = false;
    return GPSMI_OK;
}

int gpsmi_acq_search_dev_async(gpsmi_acq* h, const void* d_iq, size_t n, const int32_t* prn,
                               int nsv, const double* freqs, int nbins, int n_avg,
                               gpsmi_peak* out, void* out_dev) {
    return acq_search_impl(h, d_iq, n, prn, nsv, freqs, nbins, n_avg, out, out_dev, nullptr, false);
}

int gpsmi_acq_search_dev(gpsmi_acq* h, const void* d_iq, size_t n, const int32_t* prn, int nsv,
                         const double* freqs, int nbins, int n_avg, gpsmi_peak* out,
                         void* out_dev) {
    return acq_search_impl(h, d_iq, n, prn, nsv, freqs, nbins, n_avg, out, out_dev, nullptr);
}

int gpsmi_acq_search(gpsmi_acq* h, const float* iq, size_t n, const int32_t* prn, int nsv,
                     const double* freqs, int nbins, int n_avg, gpsmi_peak* out) {
    return gpsmi_acq_search_ex(h, iq, n, prn, nsv, freqs, nbins, n_avg, out, nullptr);
}

int gpsmi_acq_search_ex(gpsmi_acq* h, const float* iq, size_t n, const int32_t* prn, int nsv,
                        const double* freqs, int nbins, int n_avg, gpsmi_peak* out, float* nbr) {
    GPSMI_REQUIRE(h && iq && out, "null argument");
    GPSMI_REQUIRE(n_avg >= 1 && n_avg <= h->cfg.n_cyc, "n_avg out of range 1..n_cyc");
    GPSMI_REQUIRE(n >= (size_t)n_avg * h->cfg.code_samples, "iq shorter than n_avg code periods");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    size_t need = (size_t)n_avg * h->cfg.code_samples;
    if (need > h->iq_cap) {
        if (h->d_iq) GPSMI_HIP(hipFree(h->d_iq));
        h->d_iq = nullptr; h->iq_cap = 0;
        GPSMI_HIP(hipMalloc((void**)&h->d_iq, need * sizeof(float2)));
        h->iq_cap = need;
    }
    // (the staging buffer is sized for complex64; raw input uploads a quarter of it)
    GPSMI_HIP(hipMemcpyAsync(h->d_iq, iq, need * (h->iq_fmt == GPSMI_IQ_U8 ? 2 : sizeof(float2)),
                             hipMemcpyHostToDevice, h->stream));
    return acq_search_impl(h, h->d_iq, need, prn, nsv, freqs, nbins, n_avg, out, nullptr, nbr);
}

int gpsmi_acq_set_input_format(gpsmi_acq* h, int fmt) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_REQUIRE(fmt == GPSMI_IQ_C64 || fmt == GPSMI_IQ_U8, "unknown input format");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    h->iq_fmt = fmt;
    return GPSMI_OK;
}

int gpsmi_acq_last_ms(gpsmi_acq* h, float* ms) {
    GPSMI_REQUIRE(h && ms, "null argument");
    *ms = h->last_ms;
    return GPSMI_OK;
}

int gpsmi_acq_after_trk(gpsmi_acq* later, gpsmi_trk* earlier) {
    GPSMI_REQUIRE(later && earlier, "null handle");
    const HandleSync e = trk_sync(earlier);
    GPSMI_REQUIRE(e.device == later->cfg.device, "handles on different devices");
    GPSMI_HIP(hipSetDevice(e.device));
    if (e.tail) {                         // (an event record is a barrier packet in the queue: reuse one)
        GPSMI_HIP(hipStreamWaitEvent(later->stream, e.tail, 0));
        return GPSMI_OK;
    }
    GPSMI_HIP(hipEventRecord(e.order, e.stream));
    GPSMI_HIP(hipStreamWaitEvent(later->stream, e.order, 0));
    return GPSMI_OK;
}

}  // extern "C"
