// Workgroup-resident 2048-point complex FFT for gfx950, staged through LDS.
//
// 256 threads (4 wave64), 8 points per thread held in registers between passes,
// Stockham radix 8-8-8-4.  Thread t enters with x[t + 256 r] in v[r] and leaves
// with X[t + 256 q] in v[q], the same layout, so spectra can be multiplied
// element-wise and transformed again without any reshuffle.  Three exchanges go
// through two LDS buffers (ping-pong, so one barrier per exchange); each buffer
// is a pair of float planes (re, im) with one pad word per 32, which makes the
// stride-8 scatter of pass 1 and every gather conflict-free for ds_*_b32.
//
// Only the forward transform exists: ifft(Y) = conj(fft(conj(Y)))/N, and the
// callers need |ifft| only.
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

constexpr int kFftN = 2048;
constexpr int kFftThreads = 256;
constexpr int kFftPlane = kFftN + kFftN / 32;         // buffer 0 plane, floats
constexpr int kFftPlane1 = kFftN + 8 * (kFftN / 64);  // buffer 1 plane, floats
constexpr int kFftLdsFloats = 2 * kFftPlane + 2 * kFftPlane1;   // two buffers x (re, im)

// Buffer 0 takes the stride-8 scatter of pass 1 (8 t + r) and the unit-stride
// scatter of pass 3: one pad word per 32 makes both and the gathers
// conflict-free.  Buffer 1 takes the scatter of pass 2 (64 (t/8) + t%8 + 8 r):
// eight pad words per 64 spread the four 64-blocks of a half-wave over all 32
// banks, and the unit-stride gather stays conflict-free.
__device__ __forceinline__ int fft_pad(int i) { return i + (i >> 5); }
__device__ __forceinline__ int fft_pad1(int i) { return i + 8 * (i >> 6); }

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) {
    return make_float2(a.x + b.x, a.y + b.y);
}
__device__ __forceinline__ float2 csub(float2 a, float2 b) {
    return make_float2(a.x - b.x, a.y - b.y);
}
// a * (-i)
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

// 4-point DFT, Y[k] = sum_n b[n] (-i)^(nk), in place, natural order.
__device__ __forceinline__ void dft4(float2& b0, float2& b1, float2& b2, float2& b3) {
    float2 c0 = cadd(b0, b2), c1 = csub(b0, b2);
    float2 c2 = cadd(b1, b3), c3 = cmul_mi(csub(b1, b3));
    b0 = cadd(c0, c2);
    b2 = csub(c0, c2);
    b1 = cadd(c1, c3);
    b3 = csub(c1, c3);
}

// 8-point DFT in place, natural order in and out.
__device__ __forceinline__ void dft8(float2* v) {
    const float h = 0.70710678118654752440f;
    float2 a0 = cadd(v[0], v[4]), a4 = csub(v[0], v[4]);
    float2 a1 = cadd(v[1], v[5]), a5 = csub(v[1], v[5]);
    float2 a2 = cadd(v[2], v[6]), a6 = csub(v[2], v[6]);
    float2 a3 = cadd(v[3], v[7]), a7 = csub(v[3], v[7]);
    // odd branch twiddles w8^1, w8^2 = -i, w8^3
    a5 = make_float2(h * (a5.x + a5.y), h * (a5.y - a5.x));
    a6 = cmul_mi(a6);
    a7 = make_float2(h * (a7.y - a7.x), -h * (a7.x + a7.y));
    dft4(a0, a1, a2, a3);      // X0 X2 X4 X6
    dft4(a4, a5, a6, a7);      // X1 X3 X5 X7
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = a4; v[3] = a5; v[5] = a6; v[7] = a7;
}

// Twiddles of one thread.  Passes 2 and 3 need 64th / 512th roots indexed by
// r*(t mod 8) and r*(t mod 64): two compact LDS tables (64 + 512 entries, reads
// at small strides, no global-memory latency inside a transform).  Pass 4 needs
// six values that depend on the thread only: registers, loaded once per kernel.
constexpr int kFftTwFloats = 2 * (64 + 512);          // LDS floats for the two tables
struct FftTw {
    const float2* t64;       // LDS: exp(-2 pi i k / 64),  k = 0..63
    const float2* t512;      // LDS: exp(-2 pi i k / 512), k = 0..511
    float2 w4[2][3];         // exp(-2 pi i r (t + 256 b) / 2048), r = 1..3
};

// tw[k] = exp(-2 pi i k / 2048) in global memory; lds_tw: kFftTwFloats floats.
// Call once per kernel by all 256 threads, then __syncthreads() before the
// first transform.
__device__ __forceinline__ FftTw fft_setup(float* lds_tw, const float2* __restrict__ tw, int t) {
    float2* t64 = reinterpret_cast<float2*>(lds_tw);
    float2* t512 = t64 + 64;
    if (t < 64) t64[t] = tw[32 * t];
    t512[t] = tw[4 * t];
    t512[t + 256] = tw[4 * (t + 256)];
    FftTw f;
    f.t64 = t64;
    f.t512 = t512;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 1; r <= 3; ++r) f.w4[b][r - 1] = tw[r * (t + 256 * b)];
    return f;
}

__device__ __forceinline__ void fft2048(float2* v, float* lds, const FftTw& tw, int t) {
    float* re0 = lds;
    float* im0 = lds + kFftPlane;
    float* re1 = lds + 2 * kFftPlane;
    float* im1 = lds + 2 * kFftPlane + kFftPlane1;

    // pass 1: Ns = 1, no twiddles; out index 8 t + r
    dft8(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int p = fft_pad(8 * t + r);
        re0[p] = v[r].x; im0[p] = v[r].y;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int p = fft_pad(t + 256 * r);
        v[r] = make_float2(re0[p], im0[p]);
    }
    // pass 2: Ns = 8, twiddle exp(-2 pi i r k / 64), out (t/8)*64 + k + 8 r
    {
        int k = t & 7;
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw.t64[r * k]);
        dft8(v);
        int base = (t >> 3) * 64 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int p = fft_pad1(base + 8 * r);
            re1[p] = v[r].x; im1[p] = v[r].y;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int p = fft_pad1(t + 256 * r);
        v[r] = make_float2(re1[p], im1[p]);
    }
    // pass 3: Ns = 64, twiddle exp(-2 pi i r k / 512), out (t/64)*512 + k + 64 r
    {
        int k = t & 63;
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmul(v[r], tw.t512[r * k]);
        dft8(v);
        int base = (t >> 6) * 512 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            int p = fft_pad(base + 64 * r);
            re0[p] = v[r].x; im0[p] = v[r].y;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        int p = fft_pad(t + 256 * r);
        v[r] = make_float2(re0[p], im0[p]);
    }
    // pass 4: Ns = 512, radix 4, two butterflies per thread (j = t, t + 256);
    // inputs z[j + 512 r] = v[b + 2 r], outputs X[j + 512 r] -> same slots.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        float2 z1 = cmul(v[b + 2], tw.w4[b][0]);
        float2 z2 = cmul(v[b + 4], tw.w4[b][1]);
        float2 z3 = cmul(v[b + 6], tw.w4[b][2]);
        float2 z0 = v[b];
        dft4(z0, z1, z2, z3);
        v[b] = z0; v[b + 2] = z1; v[b + 4] = z2; v[b + 6] = z3;
    }
    // the next user of `lds` must barrier before overwriting buffer 0
}

}  // namespace gpsmi
