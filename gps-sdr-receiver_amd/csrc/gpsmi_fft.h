// Workgroup-resident 2048-point complex FFT for gfx950, staged through LDS.
//
// 256 threads (4 wave64), 8 points per thread held in registers between passes,
// Stockham radix 8-8-8-4.  Thread t enters with x[t + 256 r] in v[r] and leaves
// with X[t + 256 q] in v[q], the same layout, so spectra can be multiplied
// element-wise and transformed again without any reshuffle.  Three exchanges go
// through two LDS buffers (ping-pong, so one barrier per exchange) of interleaved
// (re, im) pairs moved with ds_*_b64; one pad element per 32 (buffer 0) or eight per
// 64 (buffer 1) makes every scatter and gather conflict-free.
//
// The arithmetic is packed: a complex number is one even-aligned VGPR pair, an
// add/subtract is one v_pk_add_f32, a multiplication by -i or +i is folded into the
// following add through op_sel/neg modifiers, and a twiddle multiplication is
// v_pk_mul_f32 + v_pk_fma_f32.  (The scalar version spent 550 VALU instructions per
// thread and transform; the correlation kernel built on it was VALU-bound.)
//
// Only the forward transform exists: ifft(Y) = conj(fft(conj(Y)))/N, and the
// callers need |ifft| only.
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

// A workgroup barrier that orders LDS traffic only.  __syncthreads() is a full workgroup fence: it
// also waits for every outstanding global load (s_waitcnt vmcnt(0)), so a load issued ahead of a
// transform to hide its latency behind it -- the replica spectrum of the code-phase correlation, the
// rows of a pipelined fold, the first tile of a correlator's next unit -- would be waited for at the
// transform's first barrier.  What the threads of these kernels exchange lives in LDS: only the LDS
// counter has to drain.  (Values loaded from global memory are still waited for where they are
// used: the compiler tracks that per register.)
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


constexpr int kFftN = 2048;
constexpr int kFftThreads = 256;
constexpr int kFftPlane = kFftN + kFftN / 32;         // buffer 0, complex elements
constexpr int kFftPlane1 = kFftN + 8 * (kFftN / 64);  // buffer 1, complex elements
constexpr int kFftLdsFloats = 2 * kFftPlane + 2 * kFftPlane1;   // two buffers of (re, im) pairs

// Buffer 0 takes the stride-8 scatter of pass 1 (8 t + r) and the unit-stride
// scatter of pass 3: one pad element per 32 makes both and the gathers
// conflict-free.  Buffer 1 takes the scatter of pass 2 (64 (t/8) + t%8 + 8 r):
// eight pad elements per 64 spread the four 64-blocks of a half-wave over all
// banks, and the unit-stride gather stays conflict-free.
__device__ __forceinline__ int fft_pad(int i) { return i + (i >> 5); }
__device__ __forceinline__ int fft_pad1(int i) { return i + 8 * (i >> 6); }

typedef float fft_c __attribute__((ext_vector_type(2)));   // (re, im) in one register pair

// a + (-i) b and a + (+i) b in one packed add: the halves of b swapped, one negated
__device__ __forceinline__ fft_c cadd_mi(fft_c a, fft_c b) {
    fft_c d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ fft_c cadd_pi(fft_c a, fft_c b) {
    fft_c d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// -a + (-i) a  =  (-1 - i) a   (for the eighth root w8^3)
__device__ __forceinline__ fft_c cneg_add_mi(fft_c a) {
    fft_c d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,1]" : "=v"(d) : "v"(a));
    return d;
}
// a * w: (a.x w.x - a.y w.y, a.x w.y + a.y w.x)
__device__ __forceinline__ fft_c cmulp(fft_c a, fft_c w) {
    fft_c d;
    asm("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
        : "=&v"(d) : "v"(a), "v"(w));
    return d;
}

// conj(a) * w, packed the same way
__device__ __forceinline__ fft_c cmulp_conj(fft_c a, fft_c w) {
    fft_c d;
    asm("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]"
        : "=&v"(d) : "v"(a), "v"(w));
    return d;
}

// 4-point DFT, Y[k] = sum_n b[n] (-i)^(nk), in place, natural order.  With B2MI the
// input b2 still lacks its factor -i (folded into the first two adds).
template <bool B2MI>
__device__ __forceinline__ void dft4p(fft_c& b0, fft_c& b1, fft_c& b2, fft_c& b3) {
    const fft_c c0 = B2MI ? cadd_mi(b0, b2) : b0 + b2;
    const fft_c c1 = B2MI ? cadd_pi(b0, b2) : b0 - b2;
    const fft_c c2 = b1 + b3, d = b1 - b3;
    b0 = c0 + c2;
    b2 = c0 - c2;
    b1 = cadd_mi(c1, d);
    b3 = cadd_pi(c1, d);
}

// 8-point DFT in place, natural order in and out.
__device__ __forceinline__ void dft8p(fft_c* v) {
    const float h = 0.70710678118654752440f;
    fft_c a0 = v[0] + v[4], a4 = v[0] - v[4];
    fft_c a1 = v[1] + v[5], a5 = v[1] - v[5];
    fft_c a2 = v[2] + v[6], a6 = v[2] - v[6];
    fft_c a3 = v[3] + v[7], a7 = v[3] - v[7];
    // odd branch twiddles: w8^1 = h (1 - i), w8^2 = -i (folded into dft4p), w8^3 = h (-1 - i)
    a5 = cadd_mi(a5, a5) * h;
    a7 = cneg_add_mi(a7) * h;
    dft4p<false>(a0, a1, a2, a3);     // X0 X2 X4 X6
    dft4p<true>(a4, a5, a6, a7);      // X1 X3 X5 X7
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = a4; v[3] = a5; v[5] = a6; v[7] = a7;
}

// Twiddles of one thread.  Passes 2 and 3 need 64th / 512th roots indexed by
// r*(t mod 8) and r*(t mod 64): two compact LDS tables (64 + 512 entries, reads
// at small strides, no global-memory latency inside a transform).  Pass 4 needs
// six values that depend on the thread only: registers, loaded once per kernel.
constexpr int kFftTwFloats = 2 * (64 + 512);          // LDS floats for the two tables
struct FftTw {
    const fft_c* t64;        // LDS: exp(-2 pi i k / 64),  k = 0..63
    const fft_c* t512;       // LDS: exp(-2 pi i k / 512), k = 0..511
    fft_c w4[2][3];          // exp(-2 pi i r (t + 256 b) / 2048), r = 1..3
};

// tw[k] = exp(-2 pi i k / 2048) in global memory; lds_tw: kFftTwFloats floats.
// Call once per kernel by all 256 threads, then __syncthreads() before the
// first transform.
__device__ __forceinline__ FftTw fft_setup(float* lds_tw, const float2* __restrict__ tw, int t) {
    float2* t64 = reinterpret_cast<float2*>(lds_tw);
    float2* t512 = t64 + 64;
    t64[t & 63] = tw[32 * (t & 63)];                // (every wave writes the same 64 entries: no branch to wait in)
    t512[t] = tw[4 * t];
    t512[t + 256] = tw[4 * (t + 256)];
    FftTw f;
    f.t64 = reinterpret_cast<const fft_c*>(t64);
    f.t512 = reinterpret_cast<const fft_c*>(t512);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 1; r <= 3; ++r) {
            const float2 w = tw[r * (t + 256 * b)];
            f.w4[b][r - 1] = fft_c{w.x, w.y};
        }
    return f;
}

__device__ __forceinline__ void fft2048(float2* vio, float* lds, const FftTw& tw, int t) {
    fft_c* buf0 = reinterpret_cast<fft_c*>(lds);
    fft_c* buf1 = buf0 + kFftPlane;
    fft_c v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = fft_c{vio[r].x, vio[r].y};

    // pass 1: Ns = 1, no twiddles; out index 8 t + r
    dft8p(v);
#pragma unroll
    for (int r = 0; r < 8; ++r) buf0[fft_pad(8 * t + r)] = v[r];
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf0[fft_pad(t + 256 * r)];
    // pass 2: Ns = 8, twiddle exp(-2 pi i r k / 64), out (t/8)*64 + k + 8 r
    {
        const int k = t & 7;
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulp(v[r], tw.t64[r * k]);
        dft8p(v);
        const int base = (t >> 3) * 64 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) buf1[fft_pad1(base + 8 * r)] = v[r];
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf1[fft_pad1(t + 256 * r)];
    // pass 3: Ns = 64, twiddle exp(-2 pi i r k / 512), out (t/64)*512 + k + 64 r
    {
        const int k = t & 63;
#pragma unroll
        for (int r = 1; r < 8; ++r) v[r] = cmulp(v[r], tw.t512[r * k]);
        dft8p(v);
        const int base = (t >> 6) * 512 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r) buf0[fft_pad(base + 64 * r)] = v[r];
    }
    lds_barrier();
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = buf0[fft_pad(t + 256 * r)];
    // pass 4: Ns = 512, radix 4, two butterflies per thread (j = t, t + 256);
    // inputs z[j + 512 r] = v[b + 2 r], outputs X[j + 512 r] -> same slots.
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        fft_c z1 = cmulp(v[b + 2], tw.w4[b][0]);
        fft_c z2 = cmulp(v[b + 4], tw.w4[b][1]);
        fft_c z3 = cmulp(v[b + 6], tw.w4[b][2]);
        fft_c z0 = v[b];
        dft4p<false>(z0, z1, z2, z3);
        v[b] = z0; v[b + 2] = z1; v[b + 4] = z2; v[b + 6] = z3;
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) vio[r] = make_float2(v[r].x, v[r].y);
    // the next user of `lds` must barrier before overwriting buffer 0
}

}  // namespace gpsmi
