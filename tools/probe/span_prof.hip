// Checks and times trk_span_kernel (gpsmi_trk_span.h) on fabricated descriptors, next to
// trk_stream_mfma_kernel<4> on the same data: both against a float64 reference kernel, then
// HIP-event times of back-to-back launches.  Tuning aid only (not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igps-sdr-receiver_amd/csrc \
//         tools/probe/span_prof.hip gps-sdr-receiver_amd/csrc/gpsmi_core.hip \
//         gps-sdr-receiver_amd/csrc/gpsmi_acq.hip -o tools/probe/span_prof
//   tools/probe/span_prof [blocks = 1024] [workgroup slots = 512]
// With -DPROBE_NC=16 or 8: the batch form for blocks of that many rows (no checks: the tests have them;
// timings, diagnostics and stamps only; slots = 768 / 1024).
// With -DGPSMI_SPAN_STAMPS (-o tools/probe/span_stamps): the checks, then the 100 MHz phase stamps of
// one launch of the batch form (where a unit's time goes besides its tile loop) and nothing else.
#include "../../gps-sdr-receiver_amd/csrc/gpsmi_trk.hip"
#pragma clang fp contract(fast)
#include "round1_stream_mfma.h"      // the round-1 correlator, retired from the library in round 4
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

using namespace gpsmi;
#ifndef PROBE_NC
#define PROBE_NC 32
#endif

// naive reference of partial[(b, c, o)], o = q + 1: window q = positions m >= d of row q plus
// m < d of row q + 1 of  replica[(m - d) mod CS] x[r][m] exp(-j (ph + om (r CS + m + 1) / fs))
__global__ void ref_kernel(const float2* iq, const JobMid* mid, const float* code2, int nch,
                           int nblocks, double* out) {
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nblocks * nch * 33) return;
    const int o = id % 33, c = (id / 33) % nch, b = id / (33 * nch), q = o - 1;
    const JobMid m = mid[b * nch + c];
    const int d = m.delay_used;
    double re = 0, im = 0;
    if (m.active)
        for (int part = 0; part < 2; ++part) {
            const int r = q + part;
            if (r < 0 || r > 31) continue;
            for (int p = part ? 0 : d; p < (part ? d : 2048); ++p) {
                const float2 x = iq[(size_t)b * 65536 + r * 2048 + p];
                const double cv = code2[m.prn * 4096 + ((p - d) & 2047)];
                const double th = (double)m.ph + (double)m.om * (double)(r * 2048 + p + 1) / 2048000.0;
                const double cs = cos(th), sn = -sin(th);
                re += cv * (x.x * cs - x.y * sn);
                im += cv * (x.x * sn + x.y * cs);
            }
        }
    out[2 * id] = re; out[2 * id + 1] = im;
}

struct Bufs {
    float2* iq; JobMid* mid; float* code2; float* code_eo; float2* partial; float* rec; double* ref;
};

static int g_slots = 512;          // workgroups of the batch form the chip holds (2 per CU)

// what the epilogue kernel does with the records: one wave per job -> partial[job][0 .. 32]
template <int NSP>
__global__ void collect_kernel(const float* rec, const JobMid* mid, int ngroups, int nch, int njobs,
                               float2* partial) {
    __shared__ float hi[4][64], lo[4][64];
    __shared__ float2 S[4][GPSMI_MAX_DUMPS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, job = blockIdx.x * 4 + wave;
    if (job >= njobs || !mid[job].active) return;
    span_collect<NSP>(rec, ngroups, job / nch, job % nch, mid[job].delay_used, mid[job].om, lane, hi[wave],
                      lo[wave], S[wave]);
    if (lane <= 32) partial[(size_t)job * 33 + lane] = S[wave][lane];
}

#ifdef GPSMI_SPAN_STAMPS
static unsigned long long* g_stamp_buf;
#endif
template <int NSP, int WAVES, int DIAG>
static void launch_span_t(const Bufs& B, int nblocks, int nch, bool collect) {
    TrkParams P{};
    P.cs = 2048; P.n_cyc = PROBE_NC; P.nch = nch;
    const int ng = (nch + kSpCh - 1) / kSpCh;
    int grid = nblocks * ng * (32 / NSP) / WAVES;
    if (NSP * WAVES == 32) {                       // batch form: persistent workgroups, two per CU
        const int per = (grid + g_slots - 1) / g_slots;
        grid = (grid + per - 1) / per;
    }
    hipLaunchKernelGGL((trk_span_kernel<NSP, WAVES, 0, DIAG, PROBE_NC>), dim3(grid), dim3(64 * WAVES), 0, 0, B.iq, B.mid,
                       B.code_eo, P, ng, nblocks, B.rec, B.partial);
    if (collect && NSP * WAVES != 32)
        hipLaunchKernelGGL(collect_kernel<NSP>, dim3((nblocks * nch + 3) / 4), dim3(256), 0, 0, B.rec, B.mid, ng,
                           nch, nblocks * nch, B.partial);
}
template <int DIAG>
static void launch_span_diag(const Bufs& B, int nblocks, int nch) { launch_span_t<8, 4, DIAG>(B, nblocks, nch, false); }
static void launch_span(const Bufs& B, int nblocks, int nch) { launch_span_t<8, 4, 0>(B, nblocks, nch, false); }
static void launch_span_c(const Bufs& B, int nblocks, int nch) { launch_span_t<8, 4, 0>(B, nblocks, nch, true); }
static void launch_single_c(const Bufs& B, int nblocks, int nch) { launch_span_t<1, 1, 0>(B, nblocks, nch, true); }
static void launch_single(const Bufs& B, int nblocks, int nch) { launch_span_t<1, 1, 0>(B, nblocks, nch, false); }
static void launch_single4(const Bufs& B, int nblocks, int nch) { launch_span_t<1, 4, 0>(B, nblocks, nch, false); }
static void launch_mfma(const Bufs& B, int nblocks, int nch) {
    TrkParams P{};
    P.cs = 2048; P.n_cyc = 32; P.nch = nch;
    const int ng = (nch + kMfCh - 1) / kMfCh;
    hipLaunchKernelGGL(trk_stream_mfma_kernel<4>, dim3(((nblocks + 7) / 8) * 8 * ng), dim3(256), 0, 0, B.iq,
                       B.mid, B.code2, P, ng, nblocks, B.partial);
}

template <class F>
static void check(const char* name, F launch, const Bufs& B, int nblocks, int nch) {
    const int n = nblocks * nch * 33;
    hipLaunchKernelGGL(ref_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, B.iq, B.mid, B.code2, nch,
                       nblocks, B.ref);
    hipMemset(B.partial, 0xff, (size_t)n * 8);
    launch(B, nblocks, nch);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { printf("%s: %s\n", name, hipGetErrorString(e)); exit(1); }
    std::vector<double> r(2 * n); std::vector<float2> g(n);
    hipMemcpy(r.data(), B.ref, (size_t)n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), B.partial, (size_t)n * 8, hipMemcpyDeviceToHost);
    std::vector<JobMid> mid((size_t)nblocks * nch);
    hipMemcpy(mid.data(), B.mid, mid.size() * sizeof(JobMid), hipMemcpyDeviceToHost);
    double worst = 0, scale = 0; int wi = 0, nbad = 0;
    for (int i = 0; i < n; ++i) {
        if (!mid[i / 33].active) continue;
        const double e2 = std::max(fabs(g[i].x - r[2 * i]), fabs(g[i].y - r[2 * i + 1]));
        scale = std::max(scale, std::max(fabs(r[2 * i]), fabs(r[2 * i + 1])));
        if (!(e2 <= worst)) { worst = e2; wi = i; }
        if (!(e2 <= 1e-3)) ++nbad;
    }
    printf("check %-12s %4d blocks x %2d ch: max |kernel - reference| = %.3g (largest value %.3g), %d bad, "
           "worst at block %d ch %d (delay %d) o %d: (%g, %g) vs (%g, %g)\n", name, nblocks, nch, worst,
           scale, nbad, wi / (33 * nch), (wi / 33) % nch, mid[wi / 33].delay_used, wi % 33, g[wi].x, g[wi].y,
           r[2 * wi], r[2 * wi + 1]);
}

template <class F>
static void timeit(const char* name, F launch, const Bufs& B, int nblocks, int nch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch(B, nblocks, nch);
    float best = 1e9f, sum = 0;
    const int reps = 5, per = 10;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < per; ++i) launch(B, nblocks, nch);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= per;
        best = std::min(best, ms); sum += ms;
    }
    const double gb = (double)nblocks * 2048 * PROBE_NC * 8 / 1e9;
    printf("time  %-12s %4d blocks x %2d ch: %.4f ms mean, %.4f best of %d x %d back-to-back launches: "
           "%.0f GB/s (%.1f %% of 8 TB/s)\n", name, nblocks, nch, sum / reps, best, reps, per,
           gb / (sum / reps) * 1e3, gb / (sum / reps) * 1e3 / 80.0);
}

// a VALU-bound kernel of ~150 us in front of every timed launch: what the code-phase
// correlation is to the correlator in a replay step (clock / power state, cache contents)
__global__ void heater_kernel(float* out, int n) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < n; ++i) {
        a = fmaf(a, b, c); c = fmaf(c, b, d); d = fmaf(d, b, a); b = fmaf(b, 0.99999f, 1e-6f);
    }
    if (a + c + d == 123.f) out[0] = a;
}
// reads another 512 MiB (what lies in the 256 MiB Infinity Cache afterwards is not ours)
__global__ void flush_kernel(const float4* src, size_t n, float* out) {
    float a = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        a += v.x + v.y + v.z + v.w;
    }
    if (a == 123.456f) out[0] = a;
}
static float4* g_flush = nullptr;
template <class F>
static void time_cold(const char* name, F launch, const Bufs& B, int nblocks, int nch) {
    const size_t n = (size_t)512 << 20 >> 4;
    if (!g_flush) { hipMalloc((void**)&g_flush, n * 16); hipMemset(g_flush, 0, n * 16); }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float sum = 0; const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
        hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, 0, g_flush, n, (float*)B.ref);
        hipEventRecord(e0, 0);
        launch(B, nblocks, nch);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) sum += ms;
    }
    printf("cold  %-12s %4d blocks x %2d ch: %.4f ms (event pair around one launch) behind a read of another 512 MiB\n",
           name, nblocks, nch, sum / reps);
}
template <class F>
static void time_hot(const char* name, F launch, const Bufs& B, int nblocks, int nch, int heat) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float sum = 0; const int reps = 10;
    for (int r = 0; r < reps + 2; ++r) {
        hipLaunchKernelGGL(heater_kernel, dim3(2048), dim3(256), 0, 0, (float*)B.partial, heat);
        hipEventRecord(e0, 0);
        launch(B, nblocks, nch);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r >= 2) sum += ms;
    }
    printf("hot   %-12s %4d blocks x %2d ch: %.4f ms (event pair around one launch) behind a VALU kernel of %d iterations\n",
           name, nblocks, nch, sum / reps, heat);
}

static void set_delays(Bufs& B, int nblocks, int nch, int mode) {
    static const int edge[12] = {0, 1, 2, 127, 128, 129, 511, 512, 513, 1023, 2046, 2047};
    std::vector<JobMid> mid((size_t)nblocks * nch);
    for (int b = 0; b < nblocks; ++b)
        for (int c = 0; c < nch; ++c) {
            JobMid& m = mid[(size_t)b * nch + c];
            m = JobMid{};
            m.delay_used = mode == 0 ? (1137 * c + 11) % 2048
                         : mode == 1 ? edge[(c + b) % 12]
                         : mode == 2 ? (40 * c + 20 + b) % 2048          // all boundaries in one quarter
                         : (int)(((unsigned)(b * 131 + c * 977) * 2654435761u) >> 21);
            m.active = (mode == 3 && (b + c) % 7 == 0) ? 0 : 1;
            m.prn = 2 + c;
            m.om = 6.2831853f * (-4000.f + 700.f * c); m.ph = 0.1f * b + c;
        }
    hipMemcpy(B.mid, mid.data(), mid.size() * sizeof(JobMid), hipMemcpyHostToDevice);
}


int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 1024, nch = 12;
    if (argc > 2) g_slots = atoi(argv[2]);
    const size_t blk = (size_t)2048 * PROBE_NC;
    Bufs B{};
    hipMalloc((void**)&B.iq, nblocks * blk * sizeof(float2));
    {
        std::vector<float2> h(blk * 16);
        unsigned s = 12345;
        // full-mantissa values (sums of uniforms): the chip's clock under load depends on the
        // operand bits, 8-bit values ran the kernels 10 % faster than real samples do
        auto rnd = [&]() {
            float a = 0;
            for (int i = 0; i < 4; ++i) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
            return a * 0.43f;
        };
        for (auto& v : h) { v.x = rnd(); v.y = rnd(); }
        for (int b = 0; b < nblocks; b += 16)
            hipMemcpy(B.iq + b * blk, h.data(), std::min(16, nblocks - b) * blk * sizeof(float2),
                      hipMemcpyHostToDevice);
    }
    hipMalloc((void**)&B.mid, (size_t)nblocks * nch * sizeof(JobMid));
    // replica: doubled table [prn][2][2048] for the old kernel and the reference, parity planes
    // [prn][e][2048] (entry s = code[2 (s mod 1024) + e]) for the span kernel
    std::vector<float> code2((size_t)(GPSMI_MAX_PRN + 1) * 4096), eo(code2.size());
    for (int p = 0; p <= GPSMI_MAX_PRN; ++p)
        for (int i = 0; i < 2048; ++i) {
            const unsigned hsh = (unsigned)(p * 2048 + i) * 2654435761u;
            const float v = p == 0 ? 0.f : ((hsh >> 13) & 1 ? 1.f : -1.f) * (((hsh >> 20) & 3) ? 1.f : 0.3f + (hsh >> 24) * 0.002f);
            code2[(size_t)p * 4096 + i] = code2[(size_t)p * 4096 + 2048 + i] = v;
        }
    for (int p = 0; p <= GPSMI_MAX_PRN; ++p) {
        for (int e = 0; e < 4; ++e)
            for (int s = 0; s < 1024; ++s)
                eo[(size_t)p * 4096 + e * 1024 + s] = code2[(size_t)p * 4096 + (4 * s + e) % 2048];
    }
    hipMalloc((void**)&B.code2, code2.size() * 4);
    hipMemcpy(B.code2, code2.data(), code2.size() * 4, hipMemcpyHostToDevice);
    hipMalloc((void**)&B.code_eo, eo.size() * 4);
    hipMemcpy(B.code_eo, eo.data(), eo.size() * 4, hipMemcpyHostToDevice);
    hipMalloc((void**)&B.partial, (size_t)nblocks * nch * 33 * sizeof(float2));
    hipMalloc((void**)&B.ref, (size_t)64 * nch * 33 * 16);
    hipMalloc((void**)&B.rec, (size_t)std::max(nblocks * 4, 32 * 32) * kSpRecFloats * sizeof(float));
#ifdef GPSMI_SPAN_STAMPS
    {   // every launch of the batch form writes its stamps: the buffer exists before the first one
        // (up to 1024 workgroups x 4 units x 4 waves x 8 stamps)
        hipMalloc((void**)&g_stamp_buf, (size_t)1024 * 4 * 4 * 8 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(g_span_stamps), &g_stamp_buf, sizeof(g_stamp_buf));
    }
#endif

    const int ncheck = std::min(nblocks, 24);   // (<= 32: the record buffer holds 32 x 32 records)
    for (int mode = 0; mode < (PROBE_NC == 32 ? 4 : 0); ++mode) {
        set_delays(B, nblocks, nch, mode);
        printf("-- delays: %s\n", mode == 0 ? "spread" : mode == 1 ? "edges" : mode == 2 ? "one quarter" : "random, some closed");
        check("mfma<4>", launch_mfma, B, ncheck, nch);
        check("span", launch_span_c, B, ncheck, nch);
        check("span single", launch_single_c, B, ncheck, nch);
        if (mode == 3) check("span 5ch", launch_span_c, B, ncheck, 5);
        {   // the two forms must agree bit for bit
            std::vector<float2> a((size_t)ncheck * nch * 33), c(a.size());
            launch_span_c(B, ncheck, nch); hipDeviceSynchronize();
            hipMemcpy(a.data(), B.partial, a.size() * 8, hipMemcpyDeviceToHost);
            launch_single_c(B, ncheck, nch); hipDeviceSynchronize();
            hipMemcpy(c.data(), B.partial, c.size() * 8, hipMemcpyDeviceToHost);
            printf("batch form == single-block form: %s\n", memcmp(a.data(), c.data(), a.size() * 8) ? "NO" : "yes");
        }
    }
#ifdef GPSMI_SPAN_STAMPS
    {   // phase stamps of the batch form (build with -DGPSMI_SPAN_STAMPS): per wave and unit,
        // 0 unit start, 1 row factors parked, 2 tiles done, 3 sums in LDS, 4 past the first barrier,
        // 5 combined, 6 past the second barrier; 100 MHz ticks
        set_delays(B, nblocks, nch, 0);
        const int grid = std::min(nblocks, g_slots);
        const size_t nst = (size_t)grid * 4 * 4 * 8;
        unsigned long long* d_st = g_stamp_buf;
        for (int i = 0; i < 20; ++i) launch_span(B, nblocks, nch);
        hipDeviceSynchronize();
        hipMemset(d_st, 0, nst * 8);
        launch_span(B, nblocks, nch);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(nst);
        hipMemcpy(st.data(), d_st, nst * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        for (auto v : st) if (v) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
        printf("-- stamps: kernel spans %.2f us (first to last stamp), %d workgroups\n", (t1 - t0) * 0.01, grid);
        const int iters = (nblocks + grid - 1) / grid;
        for (int it = 0; it < iters && it < 4; ++it) {
            double ph[7] = {0}, wait1 = 0, spread = 0, start_min = 1e30, start_max = 0, end_min = 1e30, end_max = 0;
            int n = 0;
            for (int wg = 0; wg < grid; ++wg) {
                const unsigned long long* w0 = &st[(((size_t)wg * 4 + it) * 4) * 8];
                if (!w0[0]) continue;
                unsigned long long tmin = ~0ull, tmax = 0;
                for (int w = 0; w < 4; ++w) {
                    const unsigned long long* q = w0 + w * 8;
                    for (int i = 0; i < 6; ++i) if (q[i + 1]) ph[i] += (double)(q[i + 1] - q[i]);
                    tmin = std::min(tmin, q[2]); tmax = std::max(tmax, q[2]);
                    wait1 += (double)(q[4] - q[3]);
                    start_min = std::min(start_min, (double)(q[0] - t0)); start_max = std::max(start_max, (double)(q[0] - t0));
                    end_min = std::min(end_min, (double)(q[5] - t0)); end_max = std::max(end_max, (double)(q[5] - t0));
                }
                spread += (double)(tmax - tmin);
                ++n;
            }
            if (!n) continue;
            printf("   unit %d of a workgroup (%d workgroups): park %.2f  tiles %.2f  sums->LDS %.2f  barrier wait %.2f  combine %.2f  second barrier %.2f us per wave;"
                   "  spread of 'tiles done' inside a workgroup %.2f us;  starts %.1f..%.1f  ends %.1f..%.1f us\n",
                   it, n, ph[0] / (4 * n) * 0.01, ph[1] / (4 * n) * 0.01, ph[2] / (4 * n) * 0.01, ph[3] / (4 * n) * 0.01,
                   ph[4] / (4 * n) * 0.01, ph[5] / (4 * n) * 0.01, spread / n * 0.01, start_min * 0.01, start_max * 0.01,
                   end_min * 0.01, end_max * 0.01);
        }
        // how many waves are inside the tile loop at each microsecond
        std::vector<int> busy((size_t)((t1 - t0) / 100 + 2), 0);
        for (size_t r = 0; r < nst / 8; ++r) {
            const unsigned long long* q = &st[r * 8];
            if (!q[0] || !q[2]) continue;
            for (unsigned long long t = (q[1] - t0) / 100; t <= (q[2] - t0) / 100; ++t) busy[t]++;
        }
        printf("   waves inside the tile loop per us:");
        for (size_t t = 0; t < busy.size(); ++t) printf(" %d", busy[t]);
        printf("\n");
        return 0;
    }
#endif
    if (PROBE_NC != 32) {
        set_delays(B, nblocks, nch, 0);
        printf("-- %d rows per block; diagnostics (delays spread): 1 no MFMAs, 2 no row loads after the first tile, "
               "4 no barrier / combine\n", PROBE_NC);
        for (int r = 0; r < 2; ++r) {
            timeit("span", launch_span, B, nblocks, nch);
            timeit("span diag 1", launch_span_diag<1>, B, nblocks, nch);
            timeit("span diag 2", launch_span_diag<2>, B, nblocks, nch);
            timeit("span diag 3", launch_span_diag<3>, B, nblocks, nch);
            timeit("span diag 4", launch_span_diag<4>, B, nblocks, nch);
            timeit("span diag 5", launch_span_diag<5>, B, nblocks, nch);
            timeit("span diag 6", launch_span_diag<6>, B, nblocks, nch);
            timeit("span diag 7", launch_span_diag<7>, B, nblocks, nch);
        }
        std::vector<JobMid> mid((size_t)nblocks * nch);
        hipMemcpy(mid.data(), B.mid, mid.size() * sizeof(JobMid), hipMemcpyDeviceToHost);
        for (auto& m : mid) m.delay_used = 0;
        hipMemcpy(B.mid, mid.data(), mid.size() * sizeof(JobMid), hipMemcpyHostToDevice);
        printf("-- every delay 0 (no boundary tiles)\n");
        timeit("span", launch_span, B, nblocks, nch);
        timeit("span diag 1", launch_span_diag<1>, B, nblocks, nch);
        timeit("span diag 2", launch_span_diag<2>, B, nblocks, nch);
        timeit("span diag 3", launch_span_diag<3>, B, nblocks, nch);
        timeit("span diag 7", launch_span_diag<7>, B, nblocks, nch);
        return 0;
    }
    for (int mode : {0, 2}) {
        set_delays(B, nblocks, nch, mode);
        printf("-- delays: %s\n", mode == 0 ? "spread" : "one quarter");
        timeit("mfma<4>", launch_mfma, B, nblocks, nch);
        timeit("span", launch_span, B, nblocks, nch);
        timeit("mfma<4>", launch_mfma, B, nblocks, nch);
        timeit("span", launch_span, B, nblocks, nch);
    }
    set_delays(B, nblocks, nch, 0);
    printf("-- diagnostics (delays spread): 1 no MFMAs, 2 no row loads after the first tile, 4 no barrier / combine\n");
    timeit("span", launch_span, B, nblocks, nch);
    timeit("span no nt (8)", launch_span_diag<8>, B, nblocks, nch);
    timeit("span", launch_span, B, nblocks, nch);
    timeit("span no nt (8)", launch_span_diag<8>, B, nblocks, nch);
    for (int heat : {0, 20000}) {
        time_hot("span", launch_span, B, nblocks, nch, heat);
        time_hot("mfma<4>", launch_mfma, B, nblocks, nch, heat);
    }
    time_cold("span", launch_span, B, nblocks, nch);
    time_cold("mfma<4>", launch_mfma, B, nblocks, nch);
    time_cold("span", launch_span, B, nblocks, nch);
    time_cold("span no nt", launch_span_diag<8>, B, nblocks, nch);
    time_cold("span diag 4", launch_span_diag<4>, B, nblocks, nch);
    time_cold("span diag 5", launch_span_diag<5>, B, nblocks, nch);
    time_cold("span diag 6", launch_span_diag<6>, B, nblocks, nch);
    timeit("span diag 4", launch_span_diag<4>, B, nblocks, nch);
    timeit("span diag 1", launch_span_diag<1>, B, nblocks, nch);
    timeit("span diag 2", launch_span_diag<2>, B, nblocks, nch);
    timeit("span diag 5", launch_span_diag<5>, B, nblocks, nch);
    timeit("span diag 6", launch_span_diag<6>, B, nblocks, nch);
    timeit("span diag 7", launch_span_diag<7>, B, nblocks, nch);
    printf("-- single-block form, few blocks (latency)\n");
    timeit("single", launch_single, B, 1, nch);
    timeit("span", launch_span, B, 1, nch);
    timeit("single", launch_single, B, 8, nch);
    timeit("single 4w", launch_single4, B, 1, nch);
    timeit("single 4w", launch_single4, B, 8, nch);
    {   // no boundary anywhere: every delay 0
        std::vector<JobMid> mid((size_t)nblocks * nch);
        hipMemcpy(mid.data(), B.mid, mid.size() * sizeof(JobMid), hipMemcpyDeviceToHost);
        for (auto& m : mid) m.delay_used = 0;
        hipMemcpy(B.mid, mid.data(), mid.size() * sizeof(JobMid), hipMemcpyHostToDevice);
        printf("-- every delay 0 (no boundary tiles)\n");
        timeit("span", launch_span, B, nblocks, nch);
        timeit("mfma<4>", launch_mfma, B, nblocks, nch);
    }
    return 0;
}
