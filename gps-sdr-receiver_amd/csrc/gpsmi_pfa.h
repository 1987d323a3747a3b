// Circular code correlation at the native code length of BASELINE config 5,
// L = 16368 = 16 * 3 * 11 * 31 samples (16.368 Msps): one workgroup works on one correlation at a
// time (persistent, one workgroup per CU), everything between the folded samples and the
// findCodePhase statistics in LDS.
//
// The reference does abs(ifft(fft(x) * conj(fft(replica)))) with scipy's length-L
// transforms (src/gpslib.py:1315-1327, src/gpsrecv.py:250-258) and then mean / std /
// argmax over the L lags (src/gpslib.py:1293-1304).  The four factors of L are
// pairwise coprime, so Z_L = Z_16 x Z_3 x Z_11 x Z_31 (Chinese remainder theorem) and
// a circular correlation over Z_L IS the four-dimensional circular correlation over
// the product group: the transform is a separable 4-D DFT with NO twiddle factors
// between the dimensions (Good-Thomas), and because only a correlation is wanted the
// order in which the 4-D spectrum comes out does not matter (signal and replica
// spectra are permuted alike).  What the layout has to respect is the coordinates:
// sample n sits at (n mod 16, n mod 3, n mod 11, n mod 31).
//
//   P1  thread c (= n mod 1023) loads x[c + 1023 j], j = 0..15 (coalesced over c): the 16
//       samples that share (n mod 3, n mod 11, n mod 31).  Their Z_16 coordinate is
//       (c - j) mod 16, so a plain FFT-16 over j followed by exp(+2 pi i c k / 16) is the
//       DFT along that axis; the 16 results go to LDS[k][sigma(c)],
//       sigma(c) = (c mod 3) 352 + (c mod 11) 32 + c mod 31 (the LDS image, see kPfaPitch).
//   P2  496 threads, one (k, i31) each: DFT-3 x DFT-11 over the 33 elements at stride 31.
//   P3  528 threads, one (k, i3, i11) each: DFT-31 of 31 contiguous elements, times the
//       replica spectrum (conjugated: the way back runs as a forward transform of the
//       conjugate, only magnitudes are needed), DFT-31 again.
//   P4  as P2.   P5  as P1 backwards: exp(+2 pi i c k / 16), FFT-16, |.| / L at lag
//       n = c + 1023 j, then mean, population std (from the sum and the sum of squares, combined in
//       double), first-index argmax and the two circular neighbours of the peak for all 16368 lags.
//
// The DFTs of prime length p are dense but use the symmetry of the roots: with
// a_n = x_n + x_(p-n), b_n = x_n - x_(p-n):  X_k, X_(p-k) = (x_0 + sum a_n cos) -/+ i (sum b_n sin),
// (p-1)^2 real multiply-adds per transform instead of 4 (p-1)^2: 29 per point at p = 31.
// No FFT of this length exists among the power-of-two kernels; the round-2 path zero-padded
// to 32768 points in four passes through memory (gpsmi_bigfft.h, still used for other lengths).
#pragma once
#include <hip/hip_runtime.h>

#include "gpsmi_direct.h"
#include "gpsmi_fft.h"
#include "gpsmi_stats.h"

namespace gpsmi {

constexpr int kPfaL = 16368;
constexpr int kPfaC = 1023;          // 3 * 11 * 31
// LDS image: element (k, i3, i11, i31) at k * kPfaPitch + i3 * kPfaS3 + i11 * kPfaS11 + i31.  The strides
// of i3 and i11 are multiples of 32 and the pitch is 31 mod 32, so that the bank of an element is
// (i31 - k) mod 32: the scatter of P1 / gather of P5 (64 consecutive c per wave: i31 runs, i3 and i11
// wrap) meets one repeated bank per 32 lanes instead of the eight of the dense order, P2 / P4 (lanes
// along i31) are conflict-free, and P3's lanes run along k first (tools: a search over the padded
// mixed-radix orders).
constexpr int kPfaS11 = 32, kPfaS3 = 11 * 32;
constexpr int kPfaPitch = 1087;      // complex elements per Z_16 row in LDS (>= 3 * kPfaS3 = 1056)
constexpr int kPfaThreads = 1024;
constexpr int kPfaLines31 = 528;     // 16 * 33 lines along Z_31
constexpr int kPfaSlabs33 = 496;     // 16 * 31 slabs Z_3 x Z_11

template <int P> struct PfaTrig;
template <> struct PfaTrig<3> {
    static constexpr double c[3] = {1.0, -0.4999999999999998, -0.5000000000000004};
    static constexpr double s[3] = {0.0, 0.8660254037844387, -0.8660254037844384};
};
template <> struct PfaTrig<11> {
    static constexpr double c[11] = {1.0, 0.8412535328311812, 0.41541501300188644, -0.142314838273285,
                                     -0.654860733945285, -0.9594929736144974, -0.9594929736144975,
                                     -0.6548607339452852, -0.14231483827328523, 0.41541501300188605,
                                     0.8412535328311812};
    static constexpr double s[11] = {0.0, 0.5406408174555976, 0.9096319953545183, 0.9898214418809328,
                                     0.7557495743542583, 0.28173255684142967, -0.2817325568414294,
                                     -0.7557495743542582, -0.9898214418809327, -0.9096319953545186,
                                     -0.5406408174555974};
};
template <> struct PfaTrig<31> {
    static constexpr double c[31] = {
        1.0, 0.9795299412524945, 0.9189578116202306, 0.8207634412072763, 0.6889669190756866,
        0.5289640103269624, 0.3473052528448203, 0.1514277775045767, -0.05064916883871264,
        -0.2506525322587204, -0.4403941515576344, -0.6121059825476626, -0.7587581226927909,
        -0.8743466161445821, -0.9541392564000488, -0.994869323391895, -0.9948693233918952,
        -0.9541392564000488, -0.8743466161445822, -0.7587581226927911, -0.6121059825476627,
        -0.44039415155763423, -0.2506525322587213, -0.05064916883871355, 0.15142777750457667,
        0.3473052528448203, 0.5289640103269624, 0.6889669190756865, 0.8207634412072763,
        0.9189578116202306, 0.9795299412524943};
    static constexpr double s[31] = {
        0.0, 0.20129852008866006, 0.39435585511331855, 0.5712682150947923, 0.7247927872291199,
        0.8486442574947509, 0.9377521321470804, 0.9884683243281114, 0.9987165071710528,
        0.9680771188662043, 0.8978045395707416, 0.7907757369376989, 0.6513724827222223,
        0.48530196253108104, 0.29936312297335804, 0.10116832198743272, -0.10116832198743204,
        -0.2993631229733582, -0.4853019625310808, -0.651372482722222, -0.7907757369376986,
        -0.8978045395707417, -0.9680771188662041, -0.9987165071710528, -0.9884683243281114,
        -0.9377521321470804, -0.848644257494751, -0.72479278722912, -0.5712682150947924,
        -0.3943558551133187, -0.20129852008866114};
};

// cos / sin(2 pi m / 16)
__device__ __forceinline__ fft_c pfa_w16(int m, bool minus) {
    constexpr float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    constexpr float cs[16] = {1.f, c1, h, s1, 0.f, -s1, -h, -c1, -1.f, -c1, -h, -s1, 0.f, s1, h, c1};
    constexpr float sn[16] = {0.f, s1, h, c1, 1.f, c1, h, s1, 0.f, -s1, -h, -c1, -1.f, -c1, -h, -s1};
    return fft_c{cs[m & 15], minus ? -sn[m & 15] : sn[m & 15]};
}

// a constant operand for the packed multiplies, made on the spot: without this hipcc keeps the nine
// twiddle pairs of pfa_fft16 in registers across the whole persistent loop (and spills them)
__device__ __forceinline__ fft_c pfa_fresh(fft_c w) {
    asm volatile("" : "+v"(w));
    return w;
}

// 16-point DFT in registers, natural order in and out, kernel exp(-2 pi i j k / 16):
// 4 x 4 Cooley-Tukey (j = 4 a + b, k = ka + 4 kb).
__device__ __forceinline__ void pfa_fft16(fft_c* v) {
    fft_c y[4][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        fft_c z0 = v[b], z1 = v[4 + b], z2 = v[8 + b], z3 = v[12 + b];
        dft4p<false>(z0, z1, z2, z3);
        y[b][0] = z0; y[b][1] = z1; y[b][2] = z2; y[b][3] = z3;
    }
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
        fft_c z0 = y[0][ka];
        fft_c z1 = ka == 0 ? y[1][0] : cmulp(y[1][ka], pfa_fresh(pfa_w16(ka, true)));
        fft_c z2 = ka == 0 ? y[2][0] : cmulp(y[2][ka], pfa_fresh(pfa_w16(2 * ka, true)));
        fft_c z3 = ka == 0 ? y[3][0] : cmulp(y[3][ka], pfa_fresh(pfa_w16(3 * ka, true)));
        dft4p<false>(z0, z1, z2, z3);
        v[ka] = z0; v[ka + 4] = z1; v[ka + 8] = z2; v[ka + 12] = z3;
    }
}

// DFT of odd prime length P, kernel exp(-2 pi i n k / P), of x[0..P-1] (destroyed).
// pre(k) is called before the pair (k, P - k) is formed (a place to issue loads for a later
// pair), out0(X_0) once, out(k, X_k, X_(P-k)) for k = 1..(P-1)/2.
template <int P, class Pre, class Out0, class Out>
__device__ __forceinline__ void pfa_dft_prime(fft_c* x, Pre pre, Out0 out0, Out out) {
    constexpr int H = (P - 1) / 2;
    fft_c sum = x[0];
#pragma unroll
    for (int n = 1; n <= H; ++n) {
        const fft_c a = x[n] + x[P - n], b = x[n] - x[P - n];
        x[n] = a;
        x[P - n] = b;
        sum += a;
    }
    out0(sum);
#pragma unroll
    for (int k = 1; k <= H; ++k) {
        pre(k);
        fft_c A = x[0], B = fft_c{0.f, 0.f};
#pragma unroll
        for (int n = 1; n <= H; ++n) {
            const float c = (float)PfaTrig<P>::c[(n * k) % P];
            const float s = (float)PfaTrig<P>::s[(n * k) % P];
            A += x[n] * c;
            B += x[P - n] * s;
        }
        // X_k = A - i B, X_(P-k) = A + i B
        out(k, cadd_mi(A, B), cadd_pi(A, B));
    }
}

// probe builds only (tools/probe/pfa_prof.hip): the shader clock at the phase boundaries of
// every workgroup, into a buffer nothing else reads
#ifdef GPSMI_PFA_STAMPS
__device__ unsigned long long* g_pfa_stamps;
#define PFA_STAMP(i) do { if (t == 0) g_pfa_stamps[(size_t)cell * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PFA_STAMP(i) do {} while (0)
#endif

struct PfaNoPre { __device__ __forceinline__ void operator()(int) const {} };

// sigma(c): position of the Z_1023 element c inside a row
__device__ __forceinline__ int pfa_sigma(int c) { return (c % 3) * kPfaS3 + (c % 11) * kPfaS11 + c % 31; }

// DFT-3 x DFT-11 in place over the 33 elements base[i3 * kPfaS3 + i11 * kPfaS11]
__device__ __forceinline__ void pfa_slab33(fft_c* base) {
    fft_c v[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) v[i] = base[(i / 11) * kPfaS3 + (i % 11) * kPfaS11];
    // Z_3 first, in registers: three elements i11 apart by 11
#pragma unroll
    for (int i11 = 0; i11 < 11; ++i11) {
        fft_c t[3] = {v[i11], v[11 + i11], v[22 + i11]};
        pfa_dft_prime<3>(
            t, PfaNoPre{}, [&](fft_c X0) { v[i11] = X0; },
            [&](int, fft_c Xk, fft_c Xpk) { v[11 + i11] = Xk; v[22 + i11] = Xpk; });
    }
    // Z_11, results straight to LDS
#pragma unroll
    for (int i3 = 0; i3 < 3; ++i3) {
        fft_c* col = base + i3 * kPfaS3;
        pfa_dft_prime<11>(
            v + i3 * 11, PfaNoPre{}, [&](fft_c X0) { col[0] = X0; },
            [&](int k, fft_c Xk, fft_c Xpk) { col[k * kPfaS11] = Xk; col[(11 - k) * kPfaS11] = Xpk; });
    }
}

// MODE 0: ncell correlations, persistent workgroups.  x: [nvec][L] complex (wiped-off, folded samples),
//         cell c uses x[xsel[cell0 + c]] and the spectrum RS[rsel[cell0 + c]]; out[cell0 + c].
// MODE 1: spectrum of the real replica rep[slot][L] into RS[slot] (P3's thread order:
//         RS[slot][q * 528 + line], q the Z_31 frequency).
template <int MODE>
__global__ __launch_bounds__(kPfaThreads) void pfa_corr_kernel(
    const float2* __restrict__ x, const float* __restrict__ rep, const int* __restrict__ xsel,
    const int* __restrict__ rsel, float2* __restrict__ RS, int cell0, int ncell, DirStats* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) fft_c data[16 * kPfaPitch];
    __shared__ float red_s[16], red_v[16], red_d[16];
    __shared__ int red_i[16];
    const int t = threadIdx.x;
    // MODE 0: the workgroups are persistent (one per CU: the LDS image admits no second one) and
    // take every gridDim-th cell; the samples of the next cell are requested before the statistics
    // of the current one, so only a workgroup's first cell waits for memory at its start
    int cell = MODE == 0 ? cell0 + (int)blockIdx.x : cell0;
    const int cell_end = MODE == 0 ? cell0 + ncell : cell0 + 1;
    __shared__ fft_c tw16[16];                    // exp(+2 pi i m / 16)
    __shared__ fft_c trig31[31];                  // (cos, sin)(2 pi m / 31) for P3's split lines
    if (t < 16) tw16[t] = pfa_w16(t, false);
    if (MODE == 0 && t >= 32 && t < 63) {
        float sn, cs;
        sincospif(2.0f * (float)(t - 32) / 31.0f, &sn, &cs);
        trig31[t - 32] = fft_c{cs, sn};
    }
    lds_barrier();
    fft_c nx[16];                                 // MODE 0: the samples of `cell`, requested one cell ahead
    // (buffer loads: one scalar base per cell, ONE lane offset for the sixteen loads, the row
    // distance in the instruction's scalar offset -- sixteen 64-bit lane addresses would not fit
    // beside the magnitudes the request is issued next to; a lane past the 1023rd reads nothing)
    // (unconditional: past the last cell the last one is requested again and never used -- under a
    // condition the sixteen pairs would have to survive the whole loop body for the case that no
    // request overwrites them, and spill)
    auto request = [&](int c) {
        if (MODE == 0) {
            c = c < cell_end ? c : cell_end - 1;
            const int sel = __builtin_amdgcn_readfirstlane(xsel[c]);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float2*>(x) + (size_t)sel * kPfaL, 0, kPfaL * (int)sizeof(float2), 0x00020000);
            const int voff = (t < kPfaC ? t : kPfaL) * (int)sizeof(float2);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                nx[j] = __builtin_bit_cast(fft_c, __builtin_amdgcn_raw_buffer_load_b64(
                    rs, voff, j * kPfaC * (int)sizeof(float2), 0));
            }
        }
    };
    // P1: FFT-16 along Z_16 of the samples in `nx` (MODE 1: of the real replica), coordinate
    // twiddle, scatter to LDS
    auto p1 = [&]() {
        // (an opaque copy of the thread index per call: the two inlined copies of this phase must not
        // share their fifteen twiddle indices and addresses across the whole loop through registers)
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int sig = pfa_sigma(t < kPfaC ? t : 0);
        if (t < kPfaC) {
            if (MODE == 1) {
                const float* rv = rep + (size_t)cell * kPfaL + t;
#pragma unroll
                for (int j = 0; j < 16; ++j) nx[j] = fft_c{rv[kPfaC * j], 0.f};
            }
            pfa_fft16(nx);
            data[sig] = nx[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                data[k * kPfaPitch + sig] = cmulp(nx[k], tw16[(t * k) & 15]);
                if ((k & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // (the twiddle reads four at a time)
            }
        }
    };
    request(cell);
    PFA_STAMP(0);
    p1();
    // (the loop body starts behind P1: the samples requested for the next cell are live through P5
    // and the statistics only, not through the register-hungry middle phases)
#pragma unroll 1
    for (;;) {
    // (the thread index is made opaque per cell: hipcc otherwise hoists every address of the five
    // phases out of the loop and spills 66 registers to keep them)
    int t_opaque = threadIdx.x;
    asm volatile("" : "+v"(t_opaque));
    const int t = t_opaque, wave = t >> 6, lane = t & 63;
    const int sig = pfa_sigma(t < kPfaC ? t : 0);
    lds_barrier();
    PFA_STAMP(1);

    // ---- P2: Z_3 x Z_11
    if (t < kPfaSlabs33) pfa_slab33(data + (t / 31) * kPfaPitch + (t % 31));
    lds_barrier();
    PFA_STAMP(2);

    // ---- P3: Z_31, x conj(replica spectrum), Z_31 again
    // 528 lines = eight full waves + 16 lines.  A ninth wave for those 16 would give one SIMD three
    // waves of ~1400 instructions against two on the others (18 K of the kernel's 46 K cycles were this
    // phase).  The 16 lines left over go to waves 8..11 instead, 16 lanes per line: lane s forms the
    // output pair (s, 31 - s) (s = 0: X_0) with the roots read from an LDS table by (n s) mod 31 --
    // ~450 instructions per wave, one such wave per SIMD.  (MODE 1, one workgroup per replica, keeps
    // the plain form.)
    if (MODE == 1) {
        if (t < kPfaLines31) {
            fft_c* line = data + (t % 16) * kPfaPitch + ((t / 16) % 3) * kPfaS3 + (t / 48) * kPfaS11;
            fft_c v[31];
#pragma unroll
            for (int i = 0; i < 31; ++i) v[i] = line[i];
            float2* dst = RS + (size_t)cell * kPfaL + t;
            pfa_dft_prime<31>(
                v, PfaNoPre{}, [&](fft_c X0) { dst[0] = make_float2(X0.x, X0.y); },
                [&](int k, fft_c Xk, fft_c Xpk) {
                    dst[(size_t)k * kPfaLines31] = make_float2(Xk.x, Xk.y);
                    dst[(size_t)(31 - k) * kPfaLines31] = make_float2(Xpk.x, Xpk.y);
                });
        }
        return;
    } else if (t < 512) {
        fft_c* line = data + (t % 16) * kPfaPitch + ((t / 16) % 3) * kPfaS3 + (t / 48) * kPfaS11;
        fft_c v[31];
#pragma unroll
        for (int i = 0; i < 31; ++i) v[i] = line[i];
        // the replica spectrum of this line, two pairs ahead of their use
        const float2* R = RS + (size_t)rsel[cell] * kPfaL + t;
        float2 r0 = R[0];
        float2 ra[3], rb[3];
        ra[1] = R[1 * kPfaLines31]; rb[1] = R[30 * kPfaLines31];
        ra[2] = R[2 * kPfaLines31]; rb[2] = R[29 * kPfaLines31];
        pfa_dft_prime<31>(
            v,
            [&](int k) {
                if (k + 2 <= 15) {
                    ra[(k + 2) % 3] = R[(size_t)(k + 2) * kPfaLines31];
                    rb[(k + 2) % 3] = R[(size_t)(31 - k - 2) * kPfaLines31];
                }
            },
            [&](fft_c X0) { line[0] = cmulp_conj(X0, fft_c{r0.x, r0.y}); },
            [&](int k, fft_c Xk, fft_c Xpk) {
                line[k] = cmulp_conj(Xk, fft_c{ra[k % 3].x, ra[k % 3].y});
                line[31 - k] = cmulp_conj(Xpk, fft_c{rb[k % 3].x, rb[k % 3].y});
            });
        // (the line belongs to this thread alone: no barrier, the LDS queue is in order)
#pragma unroll
        for (int i = 0; i < 31; ++i) v[i] = line[i];
        pfa_dft_prime<31>(
            v, PfaNoPre{}, [&](fft_c X0) { line[0] = X0; },
            [&](int k, fft_c Xk, fft_c Xpk) { line[k] = Xk; line[31 - k] = Xpk; });
    } else if (t < 768) {
        const int s = t & 15, lid = 512 + ((t - 512) >> 4);        // output pair, line (as thread `lid` of the plain form)
        fft_c* line = data + (lid % 16) * kPfaPitch + ((lid / 16) % 3) * kPfaS3 + (lid / 48) * kPfaS11;
        const float2* R = RS + (size_t)rsel[cell] * kPfaL + lid;
        const float2 rk = R[(size_t)s * kPfaLines31], rpk = R[(size_t)(s ? 31 - s : 0) * kPfaLines31];
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            fft_c v[31];
#pragma unroll
            for (int i = 0; i < 31; ++i) v[i] = line[i];            // (16 lanes read one line: broadcasts)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // every lane has the line before any writes it
            __builtin_amdgcn_wave_barrier();
            fft_c A = v[0], B = fft_c{0.f, 0.f};
            int idx = 0;
#pragma unroll
            for (int n = 1; n <= 15; ++n) {
                idx += s;
                idx = idx >= 31 ? idx - 31 : idx;
                const fft_c w = trig31[idx];                         // (cos, sin)(2 pi n s / 31)
                A += (v[n] + v[31 - n]) * w.x;
                B += (v[n] - v[31 - n]) * w.y;
            }
            fft_c Xk = cadd_mi(A, B), Xpk = cadd_pi(A, B);           // s = 0: B = 0, both are X_0
            if (pass == 0) {
                Xk = cmulp_conj(Xk, fft_c{rk.x, rk.y});
                Xpk = cmulp_conj(Xpk, fft_c{rpk.x, rpk.y});
            }
            line[s] = Xk;
            if (s) line[31 - s] = Xpk;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
        }
    }
    lds_barrier();
    PFA_STAMP(3);

    // ---- P4: Z_3 x Z_11 on the way back
    if (t < kPfaSlabs33) pfa_slab33(data + (t / 31) * kPfaPitch + (t % 31));
    lds_barrier();
    PFA_STAMP(4);

    // ---- P5: coordinate twiddle, FFT-16, magnitudes at lag n = t + 1023 j, statistics.
    // Sum, sum of squares and first-index maximum are formed as the magnitudes appear; each
    // magnitude goes back into the LDS slot its spectrum value came from (the thread's own 16
    // slots: no other thread touches them in this phase), from where the peak's two neighbours are
    // picked up later.  Nothing of the transform stays in registers: the next cell's 16 samples
    // per thread are requested right here and arrive under the reductions and barriers.
    float sm = 0.f, s2 = 0.f, bv = -1.f;
    int bi = 0x7fffffff;
    if (t < kPfaC) {
        fft_c v[16];
        v[0] = data[sig];
#pragma unroll
        for (int k = 1; k < 16; ++k) v[k] = cmulp(data[k * kPfaPitch + sig], tw16[(t * k) & 15]);
        pfa_fft16(v);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float p2 = v[j].x * v[j].x + v[j].y * v[j].y;
            const float m = __builtin_amdgcn_sqrtf(p2) * (1.0f / kPfaL);
            sm += m;
            s2 += p2;
            if (m > bv) { bv = m; bi = t + kPfaC * j; }               // ascending lag: strict >
            data[j * kPfaPitch + sig].x = m;
        }
    }
    PFA_STAMP(5);
    request(cell + (MODE == 0 ? (int)gridDim.x : 1));
    PFA_STAMP(8);
    sm = wave_sum_dpp(sm);
    s2 = wave_sum_dpp(s2);
    wave_argmax_dpp(bv, bi);
    if (lane == 0) { red_s[wave] = sm; red_d[wave] = s2; red_v[wave] = bv; red_i[wave] = bi; }
    PFA_STAMP(9);
    lds_barrier();
    PFA_STAMP(10);
    // The sixteen partial results are combined by the first sixteen lanes of wave 0 together (the
    // serial form on one thread -- 32 dependent double additions, two double divisions and a double
    // square root -- was 4.2 K of a cell's 38 K cycles with fifteen waves waiting at the barrier
    // behind it): a four-step DPP butterfly in double for the two sums, the wave argmax on the partial
    // maxima, then mean and variance with the reciprocals of L as double constants and the square
    // root in float (the rounding of the float results absorbs the difference).
    if (wave == 0) {
        const bool in = lane < 16;
        double dsm = in ? (double)red_s[lane & 15] : 0.0, ds2 = in ? (double)red_d[lane & 15] : 0.0;
        float pv = in ? red_v[lane & 15] : -1.f;
        int pi = in ? red_i[lane & 15] : 0x7fffffff;
        dsm = row_sum_f64_dpp(dsm);              // (lanes 0..15 are one DPP row)
        ds2 = row_sum_f64_dpp(ds2);
        wave_argmax_dpp(pv, pi);
        if (lane < 2) {                          // lane 0: the neighbour below the peak, lane 1: the one above
            const int nb = lane == 0 ? (pi > 0 ? pi - 1 : kPfaL - 1) : (pi < kPfaL - 1 ? pi + 1 : 0);
            const float nv = data[(nb / kPfaC) * kPfaPitch + pfa_sigma(nb % kPfaC)].x;
            const float hi = __shfl(nv, 1, 64);
            if (lane == 0) {
                const double inv_l = 1.0 / (double)kPfaL;
                // population variance from the two sums (the squares were summed unscaled)
                const double mean = dsm * inv_l;
                const double var = ds2 * (inv_l * inv_l * inv_l) - mean * mean;
                DirStats r;
                r.argmax = pi;
                r.peak = pv;
                r.mean = (float)mean;
                r.std = sqrtf((float)(var > 0.0 ? var : 0.0));
                r.lo = nv;
                r.hi = hi;
                out[cell] = r;
            }
        }
    }
    PFA_STAMP(11);
    lds_barrier();                              // (the neighbours are read: the next cell may write `data`)
    PFA_STAMP(6);
    cell += MODE == 0 ? (int)gridDim.x : 1;
    if (cell >= cell_end) break;
    PFA_STAMP(0);
    p1();
    }
}

// all correlations of `ncell` cells on `stream`, statistics included
inline void pfa_corr_launch(hipStream_t stream, const float2* x, const int* xsel, const int* rsel,
                            int ncell, float2* RS, DirStats* stats, int cell0 = 0) {
    if (ncell <= 0) return;
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) == hipSuccess)
        (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = ncell < n_cu ? ncell : n_cu;
    hipLaunchKernelGGL(pfa_corr_kernel<0>, dim3(grid), dim3(kPfaThreads), 0, stream, x,
                       (const float*)nullptr, xsel, rsel, RS, cell0, ncell, stats);
}

// spectrum of the replica in slot `slot` (rep_slot0 = table of real replicas [slots][L])
inline void pfa_replica_launch(hipStream_t stream, const float* rep_slot0, int slot, float2* RS) {
    hipLaunchKernelGGL(pfa_corr_kernel<1>, dim3(1), dim3(kPfaThreads), 0, stream,
                       (const float2*)nullptr, rep_slot0, (const int*)nullptr, (const int*)nullptr, RS,
                       slot, 1, (DirStats*)nullptr);
}

}  // namespace gpsmi
