// Times pfa_corr_kernel (gpsmi_pfa.h: the 16368-point code-phase correlation of BASELINE
// configs[4]) on random data, checks one cell against a float64 time-domain correlation on the
// host, and prints where a workgroup spends its cycles (shader-clock stamps at the phase
// boundaries, and inside the statistics phase: probe build only, -DGPSMI_PFA_STAMPS).  Tuning aid,
// not part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGPSMI_PFA_STAMPS -Iinclude \
//         -Igps-sdr-receiver_amd/csrc tools/probe/pfa_prof.hip -o tools/probe/pfa_prof
//   tools/probe/pfa_prof [cells = 6144]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gpsmi_pfa.h"

using namespace gpsmi;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int ncell = argc > 1 ? atoi(argv[1]) : 6144;
    const int L = kPfaL, nvec = std::min(ncell, 1024);
    std::vector<float2> x((size_t)nvec * L);
    std::vector<float> rep(2 * (size_t)L);
    srand(5);
    auto rnd = [] { return (float)rand() / RAND_MAX - 0.5f; };
    for (auto& v : x) v = make_float2(rnd(), rnd());
    for (int i = 0; i < L; ++i) { rep[i] = 0.f; rep[L + i] = rnd() > 0 ? 1.f : -1.f; }
    // a planted peak in cell 0: x += 0.2 * replica shifted by 1234
    for (int m = 0; m < L; ++m) x[m].x += 0.2f * rep[L + (m - 1234 + L) % L];
    std::vector<int> xsel(ncell), rsel(ncell, 1);
    for (int i = 0; i < ncell; ++i) xsel[i] = i % nvec;
    float2 *d_x, *d_RS; float* d_rep; int *d_xs, *d_rs; DirStats* d_st; unsigned long long* d_stamp;
    CK(hipMalloc(&d_x, x.size() * sizeof(float2)));
    CK(hipMalloc(&d_RS, 2 * (size_t)L * sizeof(float2)));
    CK(hipMalloc(&d_rep, rep.size() * sizeof(float)));
    CK(hipMalloc(&d_xs, ncell * sizeof(int)));
    CK(hipMalloc(&d_rs, ncell * sizeof(int)));
    CK(hipMalloc(&d_st, ncell * sizeof(DirStats)));
    CK(hipMalloc(&d_stamp, (size_t)ncell * 16 * sizeof(unsigned long long)));
    CK(hipMemcpy(d_x, x.data(), x.size() * sizeof(float2), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rep, rep.data(), rep.size() * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_xs, xsel.data(), ncell * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_rs, rsel.data(), ncell * sizeof(int), hipMemcpyHostToDevice));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pfa_stamps), &d_stamp, sizeof(d_stamp)));
    pfa_replica_launch(0, d_rep, 1, d_RS);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) pfa_corr_launch(0, d_x, d_xs, d_rs, ncell, d_RS, d_st);
    CK(hipDeviceSynchronize());
    const int reps = 10;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) pfa_corr_launch(0, d_x, d_xs, d_rs, ncell, d_RS, d_st);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("pfa_corr_kernel<0>: %d cells, %.1f us per launch, %.2f us per cell and CU (256 CUs)\n", ncell,
           ms / reps * 1e3, ms / reps * 1e3 / (ncell / 256.0));
    // ---- check cell 0 against float64 on the host
    std::vector<DirStats> st(ncell);
    CK(hipMemcpy(st.data(), d_st, ncell * sizeof(DirStats), hipMemcpyDeviceToHost));
    {
        std::vector<double> mag(L);
        double sum = 0;
        int am = 0;
        for (int n = 0; n < L; ++n) {
            double re = 0, im = 0;
            for (int m = 0; m < L; ++m) {
                const double r = rep[L + (m - n + L) % L];
                re += x[m].x * r; im += x[m].y * r;
            }
            mag[n] = std::sqrt(re * re + im * im);
            sum += mag[n];
            if (mag[n] > mag[am]) am = n;
        }
        const double mean = sum / L;
        double d2 = 0;
        for (int n = 0; n < L; ++n) d2 += (mag[n] - mean) * (mag[n] - mean);
        const double sd = std::sqrt(d2 / L);
        printf("cell 0: argmax %d (ref %d)  peak %.6f (%.6f)  mean %.6f (%.6f)  std %.6f (%.6f)  lo %.6f (%.6f)  hi %.6f (%.6f)\n",
               st[0].argmax, am, st[0].peak, mag[am], st[0].mean, mean, st[0].std, sd, st[0].lo,
               mag[(am + L - 1) % L], st[0].hi, mag[(am + 1) % L]);
        const bool ok = st[0].argmax == am && std::fabs(st[0].peak - mag[am]) < 1e-4 * mag[am] &&
                        std::fabs(st[0].mean - mean) < 1e-4 * mean && std::fabs(st[0].std - sd) < 1e-4 * sd;
        printf("%s\n", ok ? "check ok" : "CHECK FAILED");
    }
    // ---- phase stamps of the last launch: median over workgroups
    std::vector<unsigned long long> sp((size_t)ncell * 16);
    CK(hipMemcpy(sp.data(), d_stamp, sp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const char* names[6] = {"P1 load + FFT-16", "P2 3 x 11", "P3 31, x R, 31", "P4 3 x 11", "P5 FFT-16 + |.|", "statistics"};
    double tot = 0;
    for (int ph = 0; ph < 6; ++ph) {
        std::vector<double> d(ncell);
        for (int c = 0; c < ncell; ++c) d[c] = (double)(sp[(size_t)c * 16 + ph + 1] - sp[(size_t)c * 16 + ph]);
        std::nth_element(d.begin(), d.begin() + ncell / 2, d.end());
        printf("  %-18s %8.0f cycles (median over workgroups)\n", names[ph], d[ncell / 2]);
        tot += d[ncell / 2];
    }
    printf("  %-18s %8.0f cycles\n", "sum", tot);
    // inside the statistics: request issued | wave reductions + LDS | first barrier | thread 0's part | second barrier
    const int sub[6] = {5, 8, 9, 10, 11, 6};
    const char* sn[5] = {"  request", "  wave reductions", "  first barrier", "  thread 0", "  second barrier"};
    for (int k = 0; k < 5; ++k) {
        std::vector<double> d(ncell);
        for (int c = 0; c < ncell; ++c) d[c] = (double)(sp[(size_t)c * 16 + sub[k + 1]] - sp[(size_t)c * 16 + sub[k]]);
        std::nth_element(d.begin(), d.begin() + ncell / 2, d.end());
        printf("  %-18s %8.0f cycles\n", sn[k], d[ncell / 2]);
    }
    return 0;
}
