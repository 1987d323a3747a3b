// The tracking correlator: carrier-NCO mix and prompt correlate-and-dump of a
// whole 32-ms block for up to six channels per workgroup, every IQ sample
// loaded once per workgroup and shared by its channels.
//
// Replaces decodeData's array arithmetic (reference src/gpslib.py:1400-1420):
// y = roll(replica, delay) * (data * exp(-j(phase + 2 pi f t))), then sums of y
// over code-period windows.  What makes it a streaming kernel:
//
//  * The carrier separates.  For sample k = r*CS + m (row r = code period,
//    m = position in it):  phase + w (k+1)/fs = [phase + w (m+1)/fs] + w r T,
//    T = CS/fs = 1 ms.  So  B[ch][m] = replica[(m - d) mod CS] * exp(-j theta_m)
//    is computed once per (block, channel, m) and kept in registers, the inner
//    loop is one complex multiply-accumulate per channel-sample, and the row
//    factor U[r] = exp(-j w r T) is applied once per dump at the end.
//  * Window q of the reference covers samples with m >= d of row q ("hi") and
//    m < d of row q+1 ("lo").  B of a lo element carries one extra row rotation
//    (theta_m + w T), so both parts of a window share U[q].
//  * Lane mapping: wave w owns the 512 consecutive positions [512 w, 512 w+512),
//    lane l the pairs m = 512 w + 128 i + 2 l + e (i < 4, e < 2): every load
//    instruction of a wave reads 1 KiB contiguous, and for each channel all
//    waves but the one containing d are purely hi or purely lo.  Pure waves
//    accumulate acc[row] += B*x[row] (a lo wave's row r is window r-1, fixed up
//    when the waves are combined).  The one mixed wave per channel does the same
//    and then moves the lo part of the row into the accumulator of the row above
//    (window r-1): it sums the shorter side of the boundary again (one or two
//    packed complex MACs with ballot masks) and subtracts / adds.  Nothing in the
//    row loop depends on the next row, so the three-row prefetch distance holds.
//  * Rows are processed in passes of 4 so that 6 channels x 4 accumulators fit
//    in registers next to B (6 x 8 complex); after each pass the 64 lanes of a
//    wave are summed through an LDS transpose (b128 stores, fixed order:
//    deterministic).  The kernel is bound by VALU issue, not by HBM: the 96 packed
//    FMAs per wave-row are 60 % of its instruction stream.
//
// Output: partial[job][0] = head (lo part of row 0, joins the carry from the
// previous block), partial[job][q+1] = window q (q = 0..NC-2), partial[job][NC]
// = tail (hi part of the last row: a full window when d == 0, else the carry),
// all multiplied by their U.  Algorithmic HBM traffic: 8 bytes per sample per
// channel group.
#pragma once
#include <hip/hip_runtime.h>

namespace gpsmi {

constexpr int kGroupCh = 6;        // channels per workgroup
constexpr int kPassRows = 4;       // rows per accumulation pass
constexpr int kStreamThreads = 256;
constexpr int kJ = 8;              // code positions per lane (LDS-ring variant; the default kernel takes J)
constexpr int kTrVals = 2 * kGroupCh * kPassRows;            // 48 floats per lane and pass
constexpr int kTrCols = kTrVals + 2 * kGroupCh;              // + the six carry sums into the previous row
// transpose scratch: J = 8 (2 workgroups per CU) sums all 60 columns in one round,
// J = 4 (3 per CU) in two rounds of 32; row strides of 68 / 36 floats keep the b128
// stores and the b32 column reads free of bank conflicts

struct StreamChan {
    float om, ph;
    int d, prn, job, active;
};

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cmac(float2& a, float2 b, float2 x) {
    a.x = fmaf(b.x, x.x, a.x);
    a.x = fmaf(-b.y, x.y, a.x);
    a.y = fmaf(b.x, x.y, a.y);
    a.y = fmaf(b.y, x.x, a.y);
}

// Two complex multiply-accumulates a0 += b0*x0, a1 += b1*x1 as four
// v_pk_fma_f32 on (re, im) register pairs: op_sel picks the halves, neg_lo
// supplies the minus sign of re -= b.im*x.im, so b stays one register pair
// (hipcc's own packing keeps (b.re, b.re) and (-b.im, b.im) copies: 4 VGPRs per
// element).  Dependent packed FMAs need no wait state on gfx950 (checked with
// tools/probe/hazard_probe.hip); the two chains are interleaved anyway.
__device__ __forceinline__ void cmac2(v2f& a0, v2f& a1, v2f b0, v2f x0, v2f b1, v2f x1) {
    asm("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a0), "+v"(a1)
        : "v"(b0), "v"(x0), "v"(b1), "v"(x1));
}

// One sample against six channels: a[c] += b[c] * x, twelve packed FMAs with the two
// updates of an accumulator six instructions apart (4.9 cycles per instruction
// measured, 126 TFLOP/s chip-wide).
__device__ __forceinline__ void cmac6(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5, v2f b0,
                                      v2f b1, v2f b2, v2f b3, v2f b4, v2f b5, v2f x) {
    asm("v_pk_fma_f32 %0, %6, %12, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %7, %12, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %2, %8, %12, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %4, %10, %12, %4 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %5, %11, %12, %5 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %6, %12, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %7, %12, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %2, %8, %12, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %4, %10, %12, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %5, %11, %12, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(x));
}

// The same with a[c] = b[c] * x (first sample of a row: no zeroing of the accumulators)
__device__ __forceinline__ void cmac6_init(v2f& a0, v2f& a1, v2f& a2, v2f& a3, v2f& a4, v2f& a5,
                                           v2f b0, v2f b1, v2f b2, v2f b3, v2f b4, v2f b5, v2f x) {
    asm("v_pk_mul_f32 %0, %6, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %1, %7, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %2, %8, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %3, %9, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %4, %10, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_mul_f32 %5, %11, %12 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %6, %12, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %7, %12, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %2, %8, %12, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %4, %10, %12, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %5, %11, %12, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5)
        : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(b4), "v"(b5), "v"(x));
}

// (cos, -sin) of 2*pi*rev, i.e. exp(-j 2 pi rev), |error| ~ 1e-7.  Exact range
// reduction in revolutions, octant polynomials (Cephes sinf/cosf coefficients).
__device__ __forceinline__ float2 phasor_rev(float rev) {
    const float f = rev - rintf(rev);                  // [-0.5, 0.5]
    const float qf = rintf(4.0f * f);                  // -2 .. 2
    const float z = (f - 0.25f * qf) * 6.28318530717958647692f;   // |z| <= pi/4
    const float z2 = z * z;
    float sn = fmaf(fmaf(fmaf(-1.9515295891e-4f, z2, 8.3321608736e-3f), z2, -1.6666654611e-1f),
                    z2 * z, z);
    float co = fmaf(fmaf(fmaf(2.443315711809948e-5f, z2, -1.388731625493765e-3f), z2,
                         4.166664568298827e-2f), z2 * z2, fmaf(-0.5f, z2, 1.0f));
    const int q = (int)qf & 3;
    // rotate by q quarter turns: (c, s) -> (c cos - s sin ...)
    float c = (q == 0) ? co : (q == 1) ? -sn : (q == 2) ? -co : sn;
    float sg = (q == 0) ? sn : (q == 1) ? co : (q == 2) ? -sn : -co;
    return make_float2(c, -sg);
}

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}

// d = lane-mask ? x : 0 on a register pair, the 64-bit mask held in SGPRs
__device__ __forceinline__ v2f mask_pair(v2f x, unsigned long long m) {
    v2f d;
    asm("v_cndmask_b32_e64 %0, 0, %2, %4\n\t"
        "v_cndmask_b32_e64 %1, 0, %3, %4"
        : "=&v"(d.x), "=&v"(d.y)
        : "v"(x.x), "v"(x.y), "s"(m));
    return d;
}

// Sum over the shorter side of the delay boundary of one row, for the one mixed wave
// of a channel.  A wave owns J/2 chunks of 128 positions (two elements per lane and
// chunk); IS is the chunk holding the boundary.  Boundary in the lower half of the
// chunks: the lo side is summed (chunks below IS whole, chunk IS where the ballot
// masks say so); in the upper half: the hi side (chunk IS masked, chunks above whole).
// b * x as a packed pair: the first term of a chain (no zeroed accumulator)
__device__ __forceinline__ v2f cmul_pk(v2f b, v2f x) {
    v2f d;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "=&v"(d) : "v"(b), "v"(x));
    return d;
}
__device__ __forceinline__ void cmac1(v2f& a, v2f b, v2f x) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
        : "+v"(a) : "v"(b), "v"(x));
}

template <int J, int IS>
__device__ __forceinline__ v2f side_sum(const v2f* B, const v2f* x, unsigned long long m0,
                                        unsigned long long m1) {
    constexpr int NCH = J / 2;
    constexpr bool HI = IS >= (NCH + 1) / 2;
    // two chains, each started by a multiplication
    v2f t0 = cmul_pk(B[2 * IS], mask_pair(x[2 * IS], m0));
    v2f t1 = cmul_pk(B[2 * IS + 1], mask_pair(x[2 * IS + 1], m1));
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (HI ? i > IS : i < IS) cmac2(t0, t1, B[2 * i], x[2 * i], B[2 * i + 1], x[2 * i + 1]);
    return t0 + t1;
}
// moves the lo part of the row from acc (window r) to prev (window r-1)
template <int J, int IS>
__device__ __forceinline__ void side_apply(v2f& acc, v2f& prev, const v2f* B, const v2f* x,
                                           unsigned long long m0, unsigned long long m1) {
    constexpr bool HI = IS >= (J / 2 + 1) / 2;
    const v2f sd = side_sum<J, IS>(B, x, m0, m1);
    if (HI) {                       // sd = hi part: the rest of the row total is lo
        prev += acc - sd;
        acc = sd;
    } else {                        // sd = lo part
        acc -= sd;
        prev += sd;
    }
}
template <int J>
__device__ __forceinline__ bool side_is_hi(int is) { return is >= (J / 2 + 1) / 2; }

// (kept for the LDS-ring variant) sum of the lo elements: chunks below IS whole,
// chunk IS where the masks say so
template <int IS>
__device__ __forceinline__ v2f lo_sum(const v2f* B, const v2f* x, unsigned long long m0,
                                      unsigned long long m1) {
    v2f t0 = v2f{0.f, 0.f}, t1 = v2f{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < IS; ++i) cmac2(t0, t1, B[2 * i], x[2 * i], B[2 * i + 1], x[2 * i + 1]);
    cmac2(t0, t1, B[2 * IS], mask_pair(x[2 * IS], m0), B[2 * IS + 1], mask_pair(x[2 * IS + 1], m1));
    return t0 + t1;
}

// J = code positions per lane (8 or 4).  A workgroup of 256 threads covers a span of
// 256 J positions of the code period: with J = 8 and CS = 2048 the whole period, else
// blockIdx.y selects the span ("chunk"), positions beyond CS are masked (B = 0, their
// loads redirected to position 0 of the row) and each span writes its own partial
// sums (partial[job][span][NC+1], added up by trk_partial_reduce_kernel).  J = 4
// halves B and the row ring: 3 workgroups per CU instead of 2.
// POW2: CS is 2048 (replica index by mask); otherwise any multiple of 16.
template <int NC, bool POW2, int J>
__global__ __launch_bounds__(kStreamThreads, J == 8 ? 2 : 3) void trk_stream_kernel(
    const float2* __restrict__ iq, const gpsmi_trk_state* __restrict__ st_in,
    const JobMid* __restrict__ mid, const float* __restrict__ code, TrkParams P,
    int ngroups, int nblocks, float2* __restrict__ partial) {
    static_assert(NC % kPassRows == 0 && kPassRows == 4, "the row ring assumes passes of four rows");
    static_assert(J == 8 || J == 4, "positions per lane");
    constexpr int kSpan = kStreamThreads * J;                 // positions per workgroup
    constexpr int kWaveSpan = 64 * J;                         // positions per wave
    constexpr int NLD = J / 2;                                // float4 loads per lane and row
    constexpr int kRounds = J == 8 ? 1 : 2, kTrHalf = 64 / kRounds, kTrStride = J == 8 ? 68 : 36;
    __shared__ __attribute__((aligned(16))) float tr[4][64][kTrStride];   // per-wave transpose scratch
    __shared__ float2 sw[4][kGroupCh][NC];                    // per-quarter row sums
    __shared__ int cls[4][kGroupCh];                          // 0 hi, 1 lo, 2 mixed
    __shared__ float2 hd[4][kGroupCh];                        // mixed quarters: lo part of row 0 (head)
    __shared__ StreamChan schan[kGroupCh];
    __shared__ float2 rot[kGroupCh][J + 1];                   // exp(-j w off_j/fs), [J]: exp(-j w T)

    // XCD-aware decode: the groups of one block run on one XCD, back to back
    const int wg = blockIdx.x;
    const int xcd = wg & 7, slot = wg >> 3;
    const int g = slot % ngroups;
    const int b = (slot / ngroups) * 8 + xcd;
    if (b >= nblocks) return;

    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int cs = POW2 ? kFftN : P.cs;
    const int chunk = blockIdx.y, nchunks = gridDim.y;
    const float2* blk = iq + (size_t)b * ((size_t)cs * NC);
    const float inv_fs = 1.0f / (1000.0f * (float)cs);

    // ---- per-channel set-up
    // (a) wave-uniform rotation constants, one thread each, angles in double
    const double inv_2pi = 0.15915494309189533576888376337251;
    if (t < kGroupCh * (J + 1)) {
        const int c = t / (J + 1), k = t % (J + 1);
        const int cidx = g * kGroupCh + c;
        float2 r = make_float2(1.f, 0.f);
        const JobMid md = mid[b * P.nch + (cidx < P.nch ? cidx : P.nch - 1)];
        if (cidx < P.nch && md.active) {
            const int off = (k == J) ? cs : 128 * (k >> 1) + (k & 1);
            const double rev = (double)md.om * inv_2pi * (double)off / (1000.0 * (double)cs);
            r = phasor_rev((float)(rev - rint(rev)));
        }
        rot[c][k] = r;
    }
    __syncthreads();
    // (b) B in registers: one base phasor per channel and lane, the other positions by
    // rotation; lo elements carry one extra row rotation
    v2f B[kGroupCh][J];
    int kcls[kGroupCh], istar[kGroupCh];
    unsigned long long lm0[kGroupCh], lm1[kGroupCh];
    // the quarter of the span a wave owns rotates with the block, so that a wave slot
    // (= SIMD) does not get the delay boundaries of every block
    const int wq = (wave + b) & 3;
    const int w0 = kSpan * chunk + kWaveSpan * wq;            // first position of the wave
    const int mbase = w0 + 2 * lane;                          // m = mbase + 128 i + e
    int xo[NLD];                                              // load offsets in the row
#pragma unroll
    for (int i = 0; i < NLD; ++i) xo[i] = (mbase + 128 * i < cs) ? mbase + 128 * i : 0;
    // ---- the first three rows are requested before B is built: their latency hides
    // behind the set-up arithmetic
    // ring of four row buffers with static roles: row r lives in xb[r & 3]; while
    // row r is processed, row r+3 is loaded into the buffer row r-1 just left
    // (prefetch distance three rows, no register copies)
    v2f xb[4][J];
    // Loads are unconditional (rows past the end re-read the last row; their data is
    // never used): a load inside a branch gives the paths different numbers of
    // outstanding loads and hipcc then falls back to s_waitcnt vmcnt(0).
    auto load_row = [&](v2f* dst, int r) {
        const float2* row = blk + (size_t)(r < NC ? r : NC - 1) * cs;
        const float4* p = reinterpret_cast<const float4*>(row + mbase);
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            float4 v;
            if (POW2) v = p[i * 64];                           // 128 samples = 64 float4 apart
            else v = *reinterpret_cast<const float4*>(row + xo[i]);
            dst[2 * i] = v2f{v.x, v.y};
            dst[2 * i + 1] = v2f{v.z, v.w};
        }
        __builtin_amdgcn_sched_barrier(0);                     // keep the loads up here
    };
    load_row(xb[0], 0);
    load_row(xb[1], 1);
    load_row(xb[2], 2);

#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) {
        const int cidx = g * kGroupCh + c;
        const JobMid md = mid[b * P.nch + (cidx < P.nch ? cidx : P.nch - 1)];
        StreamChan s;
        s.job = b * P.nch + cidx;
        s.active = (cidx < P.nch) && md.active;
        s.om = md.om; s.ph = md.ph; s.d = md.delay_used; s.prn = md.prn;
        if (t == 0) schan[c] = s;
        // theta(mbase) in revolutions: ph/2pi + (om/2pi) (mbase+1)/fs
        const float f_eff = (float)((double)s.om * inv_2pi);
        const float rev0 = fmaf(f_eff, (float)(mbase + 1) * inv_fs, s.ph * (float)inv_2pi);
        float2 z0 = phasor_rev(rev0);
        const float2 rT = rot[c][J];
        // wave-uniform class: 0 all hi (m >= d), 1 all lo, 2 mixed
        int k = (s.d <= w0) ? 0 : (s.d >= w0 + kWaveSpan ? 1 : 2);
        if (!s.active) k = 0;
        k = __builtin_amdgcn_readfirstlane(k);
        kcls[c] = k;
        if (lane == 0) cls[wq][c] = k;
        if (k == 1) z0 = cmulf(z0, rT);                       // every element is lo
        const float* cv = code + (size_t)s.prn * cs;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int m = mbase + 128 * (j >> 1) + (j & 1);
            float2 z = (j == 0) ? z0 : cmulf(z0, rot[c][j]);
            if (k == 2) {                                     // only the mixed wave decides per element
                const float2 zl = cmulf(z, rT);
                const bool lo = m < s.d;
                z = make_float2(lo ? zl.x : z.x, lo ? zl.y : z.y);
            }
            float v;
            if (POW2) {
                v = s.active ? cv[(m - s.d) & (cs - 1)] : 0.f;
            } else {
                int idx = m - s.d;
                if (idx < 0) idx += cs;
                v = (s.active && m < cs) ? cv[idx] : 0.f;
            }
            B[c][j] = v2f{v * z.x, v * z.y};
        }
        // mixed wave: boundary chunk and the lane masks of its two elements, for the
        // shorter side of the boundary (see side_sum)
        int is = 0;
        unsigned long long b0 = 0, b1 = 0;
        if (k == 2) {
            is = __builtin_amdgcn_readfirstlane((s.d - w0 - 1) >> 7);   // chunk holding m = d-1
            const bool hi_side = side_is_hi<J>(is);
            b0 = __ballot((w0 + 128 * is + 2 * lane < s.d) != hi_side);
            b1 = __ballot((w0 + 128 * is + 2 * lane + 1 < s.d) != hi_side);
        }
        istar[c] = __builtin_amdgcn_readfirstlane(is);
        lm0[c] = b0;
        lm1[c] = b1;
    }
#pragma unroll 1
    for (int pass = 0; pass < NC / kPassRows; ++pass) {
        v2f acc[kGroupCh][kPassRows];
        v2f carry[kGroupCh];                // lo part of the pass's first row: belongs to the row above
#pragma unroll
        for (int c = 0; c < kGroupCh; ++c) carry[c] = v2f{0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < kPassRows; ++rr) {
            const int r = pass * kPassRows + rr;
            v2f* xc = xb[rr & 3];
            load_row(xb[(rr + 3) & 3], r + 3);
            // every channel, every element: acc[row] += B * x[row]
            if (!diag_flag(P, 1)) {
                cmac6_init(acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr], acc[4][rr], acc[5][rr],
                           B[0][0], B[1][0], B[2][0], B[3][0], B[4][0], B[5][0], xc[0]);
#pragma unroll
                for (int j = 1; j < J; ++j)
                    cmac6(acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr], acc[4][rr], acc[5][rr],
                          B[0][j], B[1][j], B[2][j], B[3][j], B[4][j], B[5][j], xc[j]);
            } else {
#pragma unroll
                for (int j = 0; j < J; ++j) asm volatile("" ::"v"(xc[j]));
#pragma unroll
                for (int c = 0; c < kGroupCh; ++c) acc[c][rr] = c == 0 ? xc[0] : v2f{0.f, 0.f};
            }
            // the one mixed wave of a channel: the lo elements of the row belong to the
            // window of the row above
            if (!diag_flag(P, 4)) {
#pragma unroll
                for (int c = 0; c < kGroupCh; ++c) {
                    if (kcls[c] == 2) {
                        v2f& prev = rr == 0 ? carry[c] : acc[c][rr == 0 ? 0 : rr - 1];
                        if (J == 8) {
                            switch (istar[c]) {
                                case 0: side_apply<J, 0>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]); break;
                                case 1: side_apply<J, 1>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]); break;
                                case 2: side_apply<J, J == 8 ? 2 : 0>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]); break;
                                default: side_apply<J, J == 8 ? 3 : 0>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]); break;
                            }
                        } else {
                            if (istar[c] == 0) side_apply<J, 0>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]);
                            else side_apply<J, 1>(acc[c][rr], prev, B[c], xc, lm0[c], lm1[c]);
                        }
                    }
                }
            }
        }
        // ---- sum over the 64 lanes of the wave: transpose through LDS in two rounds of
        // up to 32 columns (6 channels x 4 rows x re/im = 48 values per lane + 12
        // carries), fixed order
        if (diag_flag(P, 2)) {
            if (lane < kTrVals) {
                const int c = lane / (2 * kPassRows), rr = (lane % (2 * kPassRows)) / 2;
                float sacc = 0.f;
#pragma unroll
                for (int cc = 0; cc < kGroupCh; ++cc)
#pragma unroll
                    for (int r2 = 0; r2 < kPassRows; ++r2) sacc += acc[cc][r2].x + acc[cc][r2].y;
                reinterpret_cast<float*>(&sw[wq][c][pass * kPassRows + rr])[lane & 1] = sacc;
            }
            continue;
        }
        // The scratch is private to the wave and the LDS unit executes one wave's
        // instructions in order, so no workgroup barrier is needed here -- and none
        // is wanted: __syncthreads() would also wait for vmcnt(0) and drain the
        // three rows of loads in flight.
        float4* row4 = reinterpret_cast<float4*>(&tr[wave][lane][0]);
#pragma unroll
        for (int round = 0; round < kRounds; ++round) {
            __builtin_amdgcn_wave_barrier();
            // columns 0..31: channels 0..3; columns 32..59: channels 4, 5 and the carries
            const int o4 = (kRounds == 2 && round == 1) ? -8 : 0;
            if (kRounds == 1 || round == 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    row4[2 * c] = make_float4(acc[c][0].x, acc[c][0].y, acc[c][1].x, acc[c][1].y);
                    row4[2 * c + 1] = make_float4(acc[c][2].x, acc[c][2].y, acc[c][3].x, acc[c][3].y);
                }
            }
            if (kRounds == 1 || round == 1) {
#pragma unroll
                for (int c = 4; c < kGroupCh; ++c) {
                    row4[o4 + 2 * c] = make_float4(acc[c][0].x, acc[c][0].y, acc[c][1].x, acc[c][1].y);
                    row4[o4 + 2 * c + 1] =
                        make_float4(acc[c][2].x, acc[c][2].y, acc[c][3].x, acc[c][3].y);
                }
#pragma unroll
                for (int c = 0; c < kGroupCh; c += 2)
                    row4[o4 + 12 + c / 2] =
                        make_float4(carry[c].x, carry[c].y, carry[c + 1].x, carry[c + 1].y);
            }
            __builtin_amdgcn_wave_barrier();
            const int col = kTrHalf * round + lane;            // each lane sums one column
            if (lane < kTrHalf && col < kTrCols) {
                v2f s2 = v2f{0.f, 0.f}, s3 = v2f{0.f, 0.f};
#pragma unroll
                for (int l = 0; l < 64; l += 4) {                  // two chains
                    s2 += v2f{tr[wave][l][lane], tr[wave][l + 1][lane]};
                    s3 += v2f{tr[wave][l + 2][lane], tr[wave][l + 3][lane]};
                }
                s2 += s3;
                const float s = s2.x + s2.y;
                if (col < kTrVals) {
                    const int c = col / (2 * kPassRows), rr = (col % (2 * kPassRows)) / 2;
                    float* dst = reinterpret_cast<float*>(&sw[wq][c][pass * kPassRows + rr]);
                    dst[col & 1] = s;
                } else {                    // carries: into the last row of the pass before
                    const int c = (col - kTrVals) >> 1;
                    float* dst = pass == 0 ? reinterpret_cast<float*>(&hd[wq][c])
                                           : reinterpret_cast<float*>(&sw[wq][c][pass * kPassRows - 1]);
                    if (pass == 0) dst[col & 1] = s;
                    else dst[col & 1] += s;
                }
            }
        }
    }
    __syncthreads();

    // ---- combine the four quarters (fixed order, whichever wave summed them), apply U,
    // write the partial sums
    // out index o = q + 1, q = -1 .. NC-1.  A lo wave's row r is window r-1.
    for (int item = t; item < kGroupCh * (NC + 1); item += kStreamThreads) {
        const int c = item / (NC + 1), o = item % (NC + 1), q = o - 1;
        const StreamChan s = schan[c];
        if (!s.active) continue;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int k = cls[w][c];
            const int r = (k == 1) ? q + 1 : q;                // lo wave: row q+1 feeds window q
            if (r >= 0 && r < NC) { sx += sw[w][c][r].x; sy += sw[w][c][r].y; }
            if (q == -1 && k == 2) { sx += hd[w][c].x; sy += hd[w][c].y; }
        }
        // U[q] = exp(-j om q T), angle reduced in double (in revolutions)
        const double rev = (double)s.om * 0.15915494309189533576888376337251 * (double)q * 1.0e-3;
        const float2 u = phasor_rev((float)(rev - rint(rev)));          // (cos a, -sin a)
        partial[((size_t)s.job * nchunks + chunk) * (NC + 1) + o] =
            make_float2(sx * u.x - sy * u.y, sy * u.x + sx * u.y);
    }
}

}  // namespace gpsmi
