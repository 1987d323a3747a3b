// Circular code correlation for code lengths up to 16384 samples through one
// 32768-point FFT pair (BASELINE config 5: L = 16368 = 16*3*11*31 has no
// workgroup-resident power-of-two transform of its own length).
//
//     corr[n] = | sum_m x[m] replica[(m - n) mod L] | = | lin[n] + lin[n - L] |,
//     lin[k]  = sum_m x[m] replica[m - k]   (linear cross-correlation, |k| < L),
//
// and lin = IFFT_N( FFT_N(x 0-padded) * conj(FFT_N(replica 0-padded)) ) without aliasing
// for N = 32768 >= 2L - 1.  The reference does the same product with scipy's length-L
// transforms (src/gpsrecv.py:258, src/gpslib.py:1324-1325); the time-domain kernel of
// gpsmi_direct.h is exact but costs L^2 = 268 M MACs per correlation, this costs ~3 M.
//
// N = 16 x 2048, four-step: index n = 2048 n1 + n2, k = k1 + 16 k2,
//   X[k1 + 16 k2] = sum_n2 W_N^(n2 k1) [ sum_n1 x[2048 n1 + n2] W_16^(n1 k1) ] W_2048^(n2 k2):
//   big_cols_kernel : the 16-point DFTs down the columns and the W_N twiddles,
//   big_rows_kernel : per row k1 the LDS-resident fft2048, the product with the replica
//                     spectrum (same layout), and the row transform of the way back,
//   big_cols_kernel : again for the way back (everything is kept conjugated, so the
//                     same forward kernels serve; only magnitudes are needed),
//   big_fold_kernel : mag[n] = |Y[n] + Y[n + N - L]| / N.
// Scratch: 256 KiB per correlation between the steps; the host runs the cells in chunks
// that stay inside the 256 MiB Infinity Cache.
#pragma once
#include <hip/hip_runtime.h>

#include "gpsmi_fft.h"

namespace gpsmi {

constexpr int kBigN = 32768;
constexpr int kBigRows = 16;                 // N = kBigRows * kFftN
constexpr int kBigChunkCells = 512;          // 128 MiB of scratch per chunk

__device__ __forceinline__ float2 big_cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// 16-point DFT down a column of up to `rows_in` non-zero rows, times W_N^(n2 k1).
// FIRST = true: input is the zero-padded signal of cell c (complex x[xsel[c]][L], or
// the real replica rep[c][L] when REAL), output S[c][k1][n2].  FIRST = false: input is
// S[c][k][n2] itself (16 rows), transformed in place and not twiddled.
template <bool FIRST, bool REAL>
static __global__ __launch_bounds__(256) void big_cols_kernel(
    const void* __restrict__ xin, const int* __restrict__ xsel, int L, float2* __restrict__ S,
    const float2* __restrict__ twN, int cell0) {
    const int n2 = blockIdx.x * 256 + threadIdx.x, cell = blockIdx.y;
    float2* Sc = S + (size_t)cell * kBigN;
    // w16[j] = exp(-2 pi i j / 16) = W_N^(2048 j)
    float2 in[kBigRows];
    int rows = kBigRows;
    if (FIRST) {
        rows = (L + kFftN - 1) / kFftN;                        // <= 8: the rest is padding
        const size_t base = (size_t)(xsel ? xsel[cell0 + cell] : cell0 + cell) * L;
#pragma unroll
        for (int r = 0; r < kBigRows / 2; ++r) {
            const int n = kFftN * r + n2;
            float2 v = make_float2(0.f, 0.f);
            if (r < rows && n < L) {
                if (REAL) v.x = reinterpret_cast<const float*>(xin)[base + n];
                else v = reinterpret_cast<const float2*>(xin)[base + n];
            }
            in[r] = v;
        }
#pragma unroll
        for (int r = kBigRows / 2; r < kBigRows; ++r) in[r] = make_float2(0.f, 0.f);
    } else {
#pragma unroll
        for (int r = 0; r < kBigRows; ++r) in[r] = Sc[(size_t)r * kFftN + n2];
    }
    float2 w16[kBigRows];
#pragma unroll
    for (int j = 0; j < kBigRows; ++j) w16[j] = twN[kFftN * j];
    constexpr int NIN = FIRST ? kBigRows / 2 : kBigRows;
#pragma unroll
    for (int k1 = 0; k1 < kBigRows; ++k1) {
        float2 a = in[0];
#pragma unroll
        for (int r = 1; r < NIN; ++r) {
            const float2 p = big_cmul(in[r], w16[(r * k1) & (kBigRows - 1)]);
            a.x += p.x; a.y += p.y;
        }
        if (FIRST) a = big_cmul(a, twN[n2 * k1]);
        Sc[(size_t)k1 * kFftN + n2] = a;
    }
}

// One row (k1) of one cell: fft2048.  MODE 0: store the spectrum (replica set-up).
// MODE 1: X -> conj(X) * R, fft2048 again, times W_N^(k1 n2) for the column step back.
template <int MODE>
static __global__ __launch_bounds__(256) void big_rows_kernel(
    float2* __restrict__ S, const float2* __restrict__ RS, const int* __restrict__ rsel,
    const float2* __restrict__ tw, const float2* __restrict__ twN, int cell0) {
    __shared__ __attribute__((aligned(16))) float lds[kFftLdsFloats];
    __shared__ __attribute__((aligned(16))) float lds_tw[kFftTwFloats];
    const int t = threadIdx.x, k1 = blockIdx.x, cell = blockIdx.y;
    const FftTw ftw = fft_setup(lds_tw, tw, t);
    float2* row = S + (size_t)cell * kBigN + (size_t)k1 * kFftN;
    float2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = row[t + 256 * q];
    __syncthreads();
    fft2048(v, lds, ftw, t);
    if (MODE == 1) {
        const float2* R = RS + (size_t)rsel[cell0 + cell] * kBigN + (size_t)k1 * kFftN;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float2 x = v[q], r = R[t + 256 * q];
            v[q] = make_float2(x.x * r.x + x.y * r.y, x.x * r.y - x.y * r.x);   // conj(x) * r
        }
        __syncthreads();
        fft2048(v, lds, ftw, t);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = big_cmul(v[q], twN[(t + 256 * q) * k1]);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) row[t + 256 * q] = v[q];
}

// mag[cell][n] = |Y[n] + Y[n + N - L]| / N for n < L (Y conjugated throughout: same
// magnitude).  Y[2048 n1 + n2] lives at S[cell][n1][n2].
static __global__ __launch_bounds__(256) void big_fold_kernel(const float2* __restrict__ S, int L,
                                                              float* __restrict__ mag) {
    const int n = blockIdx.x * 256 + threadIdx.x, cell = blockIdx.y;
    if (n >= L) return;
    const float2* Y = S + (size_t)cell * kBigN;
    const float2 a = Y[n], b = Y[n + kBigN - L];
    const float re = a.x + b.x, im = a.y + b.y;
    mag[(size_t)cell * L + n] = sqrtf(re * re + im * im) * (1.0f / kBigN);
}

// the whole correlation of `ncell` cells on `stream`: x [nvec][L] complex, xsel/rsel per
// cell (device), RS replica spectra [slots][N], S scratch [min(ncell, chunk)][N],
// mag [ncell][L]
inline void big_corr_launch(hipStream_t stream, const float2* x, const int* xsel, const int* rsel,
                            int ncell, int L, const float2* RS, float2* S, const float2* tw,
                            const float2* twN, float* mag) {
    for (int c0 = 0; c0 < ncell; c0 += kBigChunkCells) {
        const int nc = ncell - c0 < kBigChunkCells ? ncell - c0 : kBigChunkCells;
        hipLaunchKernelGGL((big_cols_kernel<true, false>), dim3(kFftN / 256, nc), dim3(256), 0,
                           stream, (const void*)x, xsel, L, S, twN, c0);
        hipLaunchKernelGGL(big_rows_kernel<1>, dim3(kBigRows, nc), dim3(256), 0, stream, S, RS,
                           rsel, tw, twN, c0);
        hipLaunchKernelGGL((big_cols_kernel<false, false>), dim3(kFftN / 256, nc), dim3(256), 0,
                           stream, (const void*)nullptr, (const int*)nullptr, L, S, twN, c0);
        hipLaunchKernelGGL(big_fold_kernel, dim3((L + 255) / 256, nc), dim3(256), 0, stream, S, L,
                           mag + (size_t)c0 * L);
    }
}

// spectrum of one replica (real, length L) into RS[slot]
inline void big_replica_launch(hipStream_t stream, const float* rep_slot0, int slot, int L,
                               float2* RS, const float2* tw, const float2* twN) {
    float2* dst = RS + (size_t)slot * kBigN;
    hipLaunchKernelGGL((big_cols_kernel<true, true>), dim3(kFftN / 256, 1), dim3(256), 0, stream,
                       (const void*)rep_slot0, (const int*)nullptr, L, dst, twN, slot);
    hipLaunchKernelGGL(big_rows_kernel<0>, dim3(kBigRows, 1), dim3(256), 0, stream, dst,
                       (const float2*)nullptr, (const int*)nullptr, tw, twN, 0);
}

}  // namespace gpsmi
