// Where does a wave of trk_stream_mfma_kernel spend its cycles?  Compiles the product
// translation unit with GPSMI_MF_PROF (per-wave clock64 stamps, see gpsmi_trk_stream_mfma.h)
// and launches the kernel directly on fabricated descriptors.  Tuning aid only.
// Also checks both forms of the kernel against a float64 reference kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igps-sdr-receiver_amd/csrc \
//         tools/probe/mfma_prof.hip gps-sdr-receiver_amd/csrc/gpsmi_core.hip \
//         gps-sdr-receiver_amd/csrc/gpsmi_acq.hip -o tools/probe/mfma_prof
//   tools/probe/mfma_prof 1024 x      (blocks; any second argument adds the 8-block runs)
#define GPSMI_MF_PROF 1
#include "../../gps-sdr-receiver_amd/csrc/gpsmi_trk.hip"
#pragma clang fp contract(fast)
#include "round1_stream_mfma.h"      // the round-1 correlator, retired from the library in round 4
#pragma clang fp contract(off)

#include <algorithm>
#include <cmath>
#include <cstdio>

using namespace gpsmi;

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

template <int WAVES>
static void check(int nblocks, int nch, const float2* d_iq, const JobMid* d_mid, const float* d_code2,
                  float2* d_partial) {
    TrkParams P{};
    P.cs = 2048; P.n_cyc = 32; P.nch = nch;
    const int n = nblocks * nch * 33;
    double* d_ref; hipMalloc((void**)&d_ref, n * 16);
    hipLaunchKernelGGL(ref_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_iq, d_mid, d_code2, nch,
                       nblocks, d_ref);
    hipMemset(d_partial, 0, n * 8);
    hipLaunchKernelGGL(trk_stream_mfma_kernel<WAVES>, dim3(((nblocks + 7) / 8) * 8), dim3(64 * WAVES), 0, 0,
                       d_iq, d_mid, d_code2, P, 1, nblocks, d_partial);
    std::vector<double> r(2 * n); std::vector<float2> g(n);
    hipMemcpy(r.data(), d_ref, n * 16, hipMemcpyDeviceToHost);
    hipMemcpy(g.data(), d_partial, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0; int wi = 0;
    for (int i = 0; i < n; ++i) {
        const double e = std::max(fabs(g[i].x - r[2 * i]), fabs(g[i].y - r[2 * i + 1]));
        scale = std::max(scale, std::max(fabs(r[2 * i]), fabs(r[2 * i + 1])));
        if (!(e <= worst)) { worst = e; wi = i; }
    }
    printf("check %d waves: max |kernel - reference| = %.3g (largest value %.3g) at block %d ch %d o %d: "
           "(%g, %g) vs (%g, %g)\n", WAVES, worst, scale, wi / (33 * nch), (wi / 33) % nch, wi % 33,
           g[wi].x, g[wi].y, r[2 * wi], r[2 * wi + 1]);
    hipFree(d_ref);
}

template <int WAVES>
static void run(const char* name, int nblocks, int nch, const float2* d_iq, const JobMid* d_mid,
                const float* d_code2, float2* d_partial, unsigned long long* d_prof, int dyn_lds = 0) {
    TrkParams P{};
    P.cs = 2048; P.n_cyc = 32; P.nch = nch;
    const int ng12 = (nch + kMfCh - 1) / kMfCh;
    const dim3 grid(((nblocks + 7) / 8) * 8 * ng12), block(64 * WAVES);
    const size_t nrec = (size_t)grid.x * WAVES * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int it = 0; it < 4; ++it) {
        hipMemset(d_prof, 0, nrec * 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(trk_stream_mfma_kernel<WAVES>, grid, block, dyn_lds, 0, d_iq, d_mid, d_code2, P,
                           ng12, nblocks, d_partial);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<unsigned long long> r(nrec);
    hipMemcpy(r.data(), d_prof, nrec * 8, hipMemcpyDeviceToHost);
    const size_t nw = nrec / 8;
    double setup = 0, wait = 0, pre = 0, in = 0, tail = 0, comb = 0, tot = 0;
    unsigned long long tmin = ~0ull, tmax = 0;
    std::vector<double> tots;
    for (size_t w = 0; w < nw; ++w) {
        const unsigned long long* o = &r[w * 8];
        if (!o[7]) continue;
        setup += o[1] - o[0]; wait += o[2]; pre += o[3]; in += o[4];
        tail += o[6] - o[5]; comb += o[7] - o[6]; tot += o[7] - o[0];
        tots.push_back((double)(o[7] - o[0]));
        tmin = std::min(tmin, o[0]); tmax = std::max(tmax, o[7]);
    }
    const double n = (double)tots.size();
    std::sort(tots.begin(), tots.end());
    printf("%s: %d blocks x %d ch, event %.1f us, first stamp -> last stamp %llu ticks\n", name, nblocks,
           nch, ms * 1e3, tmax - tmin);
    printf("  per wave (ticks): total %.0f (min %.0f med %.0f max %.0f)\n", tot / n, tots.front(),
           tots[tots.size() / 2], tots.back());
    printf("    set-up %.0f | tile wait (vmcnt) %.0f | tile preamble %.0f | inner loop %.0f | "
           "to barrier %.0f | combine %.0f\n", setup / n, wait / n, pre / n, in / n, tail / n,
           comb / n);
    printf("    inner loop per position %.1f ticks, preamble per tile %.1f, wait per tile %.1f\n",
           in / n / (2048.0 / WAVES), pre / n / (64.0 / WAVES), wait / n / (64.0 / WAVES));
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 1024, nch = 12;
    const size_t blk = (size_t)2048 * 32;
    float2* d_iq; JobMid* d_mid; float* d_code2; float2* d_partial; unsigned long long* d_prof;
    hipMalloc((void**)&d_iq, nblocks * blk * sizeof(float2));
    {
        std::vector<float2> h(blk * 16);
        unsigned s = 12345;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u; v.x = ((int)(s >> 16) % 256 - 128) / 512.f;
            s = s * 1664525u + 1013904223u; v.y = ((int)(s >> 16) % 256 - 128) / 512.f;
        }
        for (int b = 0; b < nblocks; b += 16)
            hipMemcpy(d_iq + b * blk, h.data(), std::min(16, nblocks - b) * blk * sizeof(float2),
                      hipMemcpyHostToDevice);
    }
    std::vector<JobMid> mid((size_t)nblocks * nch);
    for (int b = 0; b < nblocks; ++b)
        for (int c = 0; c < nch; ++c) {
            JobMid& m = mid[(size_t)b * nch + c];
            m = JobMid{};
            m.delay_used = (1137 * c + 11) % 2048; m.active = 1; m.prn = 2 + c;
            m.om = 6.2831853f * (-4000.f + 700.f * c); m.ph = 0.1f * b + c;
        }
    hipMalloc((void**)&d_mid, mid.size() * sizeof(JobMid));
    hipMemcpy(d_mid, mid.data(), mid.size() * sizeof(JobMid), hipMemcpyHostToDevice);
    std::vector<float> code((size_t)(GPSMI_MAX_PRN + 1) * 4096);
    for (size_t i = 0; i < code.size(); ++i)      // doubled table: [prn][2][2048]
        code[i] = (((i / 4096) * 2048 + i % 2048) * 2654435761u >> 13) & 1 ? 1.f : -1.f;
    hipMalloc((void**)&d_code2, code.size() * 4);
    hipMemcpy(d_code2, code.data(), code.size() * 4, hipMemcpyHostToDevice);
    hipMalloc((void**)&d_partial, (size_t)nblocks * nch * 33 * sizeof(float2));
    const size_t nrec = (size_t)((nblocks + 7) / 8) * 8 * 8 * 8;
    hipMalloc((void**)&d_prof, nrec * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_mf_prof), &d_prof, sizeof(d_prof));
    check<8>(8, nch, d_iq, d_mid, d_code2, d_partial);
    check<4>(8, nch, d_iq, d_mid, d_code2, d_partial);
    run<8>("8 waves, 1 workgroup/CU", nblocks, nch, d_iq, d_mid, d_code2, d_partial, d_prof);
    run<4>("4 waves, 3 workgroups/CU", nblocks, nch, d_iq, d_mid, d_code2, d_partial, d_prof);
    run<4>("4 waves, 2 workgroups/CU (16 KiB of unused dynamic LDS)", nblocks, nch, d_iq, d_mid, d_code2, d_partial, d_prof, 16384);
    run<4>("4 waves, 1 workgroup/CU (48 KiB of unused dynamic LDS)", nblocks, nch, d_iq, d_mid, d_code2, d_partial, d_prof, 49152);
    if (argc > 2) {     // one block per XCD: no contention
        run<8>("8 waves, 8 blocks", 8, nch, d_iq, d_mid, d_code2, d_partial, d_prof);
        run<4>("4 waves, 8 blocks", 8, nch, d_iq, d_mid, d_code2, d_partial, d_prof);
    }
    return 0;
}
