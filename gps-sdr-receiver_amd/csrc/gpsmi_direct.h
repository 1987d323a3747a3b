// Circular code correlation for code lengths that are not 2048: the exact
// time-domain form of |ifft(fft(x) * conj(fft(replica)))| (reference
// src/gpsrecv.py:258, src/gpslib.py:1324-1325),
//     corr[n] = | sum_m x[m] * replica[(m - n) mod L] |      (replica is real),
// for every lag n, plus the statistics of findCodePhase.  BASELINE config 5
// samples the code at 16.368 Msps: L = 16368 = 16*3*11*31, for which the
// workgroup-resident power-of-two FFT of gpsmi_fft.h does not exist.  L^2 real-
// by-complex multiply-accumulates (268 M for L = 16368) as packed FMAs cost about
// 8 us per correlation chip-wide: far above real time, though LDS/VALU bound.
//
//   circ_corr_direct_kernel  grid (lag tile, cell).  256 threads x 4 consecutive
//       lags = 1024 lags per workgroup; x and the replica window are staged
//       through LDS in steps of 1024 positions; per 4 positions a thread reads
//       two broadcast b128 of x and two b128 of the (reversed-index-free) replica
//       window for 16 packed FMAs.
//   corr_stats_kernel        one workgroup per cell over the L magnitudes in
//       global memory: mean, population std (two passes), first-index argmax and
//       the two circular neighbours of the peak.
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

constexpr int kDirLagsPerWg = 1024;
constexpr int kDirStep = 1024;

struct DirStats {            // per cell
    int argmax;
    float peak, mean, std;
    float lo, hi;            // corr[argmax-1], corr[argmax+1] (circular)
};

// x: [nvec][L] complex; rep: [slots][L] real; cell c uses x[xsel[c]] and
// rep[rsel[c]]; mag: [ncell][L]
static __global__ __launch_bounds__(256) void circ_corr_direct_kernel(
    const float2* __restrict__ x, const float* __restrict__ rep, const int* __restrict__ xsel,
    const int* __restrict__ rsel, int L, float* __restrict__ mag) {
    __shared__ __attribute__((aligned(16))) float2 sx[kDirStep];
    __shared__ __attribute__((aligned(16))) float sr[kDirStep + kDirLagsPerWg + 8];
    const int t = threadIdx.x, cell = blockIdx.y;
    const int n0 = blockIdx.x * kDirLagsPerWg;
    const float2* xv = x + (size_t)xsel[cell] * L;
    const float* rv = rep + (size_t)rsel[cell] * L;
    float2 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = make_float2(0.f, 0.f);
    for (int m0 = 0; m0 < L; m0 += kDirStep) {
        __syncthreads();
        // x[m0 .. m0+step), zero beyond L
        for (int i = t; i < kDirStep; i += 256) {
            const int m = m0 + i;
            sx[i] = m < L ? xv[m] : make_float2(0.f, 0.f);
        }
        // window of the replica: sr[k] = rep[(m0 - n0 - 1024 + k) mod L]
        for (int k = t; k < kDirStep + kDirLagsPerWg + 8; k += 256) {
            int idx = (m0 - n0 - kDirLagsPerWg + k) % L;
            if (idx < 0) idx += L;
            sr[k] = rv[idx];
        }
        __syncthreads();
        // lag n_q = n0 + 4 t + q needs rep[m - n_q] = sr[mm - 4 t - q + 1024]
        const float* rb = sr + kDirLagsPerWg - 4 * t - 4;       // k0 = mm - 4t + 1020
#pragma unroll 2
        for (int mm = 0; mm < kDirStep; mm += 4) {
            const float4 xa = *reinterpret_cast<const float4*>(&sx[mm]);
            const float4 xb = *reinterpret_cast<const float4*>(&sx[mm + 2]);
            const float4 r0 = *reinterpret_cast<const float4*>(rb + mm);
            const float4 r1 = *reinterpret_cast<const float4*>(rb + mm + 4);
            const float rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            const float2 xs[4] = {make_float2(xa.x, xa.y), make_float2(xa.z, xa.w),
                                  make_float2(xb.x, xb.y), make_float2(xb.z, xb.w)};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float r = rr[4 + i - q];              // k = k0 + 4 + i - q
                    acc[q].x = fmaf(xs[i].x, r, acc[q].x);
                    acc[q].y = fmaf(xs[i].y, r, acc[q].y);
                }
        }
    }
    float* out = mag + (size_t)cell * L;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = n0 + 4 * t + q;
        if (n < L) out[n] = sqrtf(acc[q].x * acc[q].x + acc[q].y * acc[q].y);
    }
}

__device__ __forceinline__ float dir_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

static __global__ __launch_bounds__(256) void corr_stats_kernel(const float* __restrict__ mag, int L,
                                                         DirStats* __restrict__ out) {
    __shared__ float red[16];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, cell = blockIdx.x;
    const float* m = mag + (size_t)cell * L;
    float s = 0.f, bv = -1.f;
    int bi = 0x7fffffff;
    for (int i = t; i < L; i += 256) {
        const float v = m[i];
        s += v;
        if (v > bv) { bv = v; bi = i; }                         // ascending i: strict >
    }
    s = dir_wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_down(bv, o, 64);
        const int oi = __shfl_down(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = s; red[4 + wave] = bv; ((int*)red)[8 + wave] = bi; }
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    bv = red[4]; bi = ((int*)red)[8];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ov = red[4 + w];
        const int oi = ((int*)red)[8 + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const float mean = s / (float)L;
    float d2 = 0.f;
    for (int i = t; i < L; i += 256) { const float d = m[i] - mean; d2 += d * d; }
    d2 = dir_wave_sum(d2);
    if (lane == 0) red[12 + wave] = d2;
    __syncthreads();
    if (t == 0) {
        d2 = (red[12] + red[13]) + (red[14] + red[15]);
        DirStats r;
        r.argmax = bi;
        r.peak = bv;
        r.mean = mean;
        r.std = sqrtf(d2 / (float)L);
        r.lo = m[bi > 0 ? bi - 1 : L - 1];
        r.hi = m[bi < L - 1 ? bi + 1 : 0];
        out[cell] = r;
    }
}

}  // namespace gpsmi
