// findCodePhase statistics (reference src/gpsrecv.py:222-238, src/gpslib.py:1293-1304)
// of a 2048-lag correlation held 8 magnitudes per thread by a 256-thread workgroup:
// mean, population standard deviation, first-index argmax and the two circular
// neighbours of the peak.  Wave reductions use DPP (row-local steps plus the two
// row broadcasts), not ds_bpermute; two workgroup barriers per call.
#pragma once
#include <hip/hip_runtime.h>

#include "gpsmi_fft.h"

namespace gpsmi {

template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_add0(float v) {      // v + partner, 0 where no partner
    const float o = __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
    return v + o;
}
// sum over the 64 lanes, broadcast to all of them
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = dpp_add0<0xB1, 0xF>(v);        // quad_perm [1,0,3,2]
    v = dpp_add0<0x4E, 0xF>(v);        // quad_perm [2,3,0,1]
    v = dpp_add0<0x141, 0xF>(v);       // row_half_mirror: 8 lanes
    v = dpp_add0<0x140, 0xF>(v);       // row_mirror: 16 lanes
    v = dpp_add0<0x142, 0xA>(v);       // row_bcast15 into rows 1 and 3
    v = dpp_add0<0x143, 0xC>(v);       // row_bcast31 into rows 2 and 3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// v + partner for a double, the partner's halves fetched by two DPP moves
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// sum over the 16 lanes of a DPP row, in every lane of the row (the adds commute: the partners of a
// step hold the same bits)
__device__ __forceinline__ double row_sum_f64_dpp(double v) {
    v = dpp_add_f64<0xB1>(v);          // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E>(v);          // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x141>(v);         // row_half_mirror
    v = dpp_add_f64<0x140>(v);         // row_mirror
    return v;
}

template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_max_own(float v) {   // max(v, partner), v where no partner
    const int vi = __builtin_bit_cast(int, v);
    return fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(vi, vi, CTRL, ROWS, 0xF, false)));
}
template <int CTRL, int ROWS>
__device__ __forceinline__ int dpp_min_own(int v) {
    return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROWS, 0xF, false));
}
// maximum and its smallest index over the 64 lanes, broadcast to all of them: the maximum
// first, then the smallest index among the lanes that hold it (12 DPP steps, no branches)
__device__ __forceinline__ void wave_argmax_dpp(float& v, int& i) {
    float m = v;
    m = dpp_max_own<0xB1, 0xF>(m);
    m = dpp_max_own<0x4E, 0xF>(m);
    m = dpp_max_own<0x141, 0xF>(m);
    m = dpp_max_own<0x140, 0xF>(m);
    m = dpp_max_own<0x142, 0xA>(m);
    m = dpp_max_own<0x143, 0xC>(m);
    m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m), 63));
    int k = v == m ? i : 0x7fffffff;
    k = dpp_min_own<0xB1, 0xF>(k);
    k = dpp_min_own<0x4E, 0xF>(k);
    k = dpp_min_own<0x141, 0xF>(k);
    k = dpp_min_own<0x140, 0xF>(k);
    k = dpp_min_own<0x142, 0xA>(k);
    k = dpp_min_own<0x143, 0xC>(k);
    v = m;
    i = __builtin_amdgcn_readlane(k, 63);
}

constexpr int kStatsRedFloats = 16;

// mag[q] = correlation magnitude at lag t + 256 q.  magbuf: 2048 floats of LDS of its
// own (not the FFT buffers), red: kStatsRedFloats floats.  Every thread returns the
// same values.  Safe to call again after any later workgroup barrier.
__device__ __forceinline__ void corr_stats8(const float* mag, int t, float* magbuf, float* red,
                                            int& amax, float& peak, float& mean, float& sd,
                                            float& lo, float& hi) {
    constexpr int N = 2048;
    const int wave = t >> 6, lane = t & 63;
    float sm = 0.f, bv = mag[0];
    int bi = t;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sm += mag[q];
        if (mag[q] > bv) { bv = mag[q]; bi = t + 256 * q; }   // ascending lag: strict >
        magbuf[t + 256 * q] = mag[q];
    }
    sm = wave_sum_dpp(sm);
    wave_argmax_dpp(bv, bi);
    if (lane == 0) { red[wave] = sm; red[4 + wave] = bv; ((int*)red)[8 + wave] = bi; }
    lds_barrier();
    sm = (red[0] + red[1]) + (red[2] + red[3]);
    bv = red[4]; bi = ((int*)red)[8];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float ov = red[4 + w];
        const int oi = ((int*)red)[8 + w];
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    mean = sm * (1.0f / N);
    float d2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float d = mag[q] - mean; d2 += d * d; }
    d2 = wave_sum_dpp(d2);
    if (lane == 0) red[12 + wave] = d2;
    lds_barrier();
    d2 = (red[12] + red[13]) + (red[14] + red[15]);
    sd = sqrtf(d2 * (1.0f / N));
    amax = bi;
    peak = bv;
    lo = magbuf[(bi + N - 1) & (N - 1)];
    hi = magbuf[(bi + 1) & (N - 1)];
}

}  // namespace gpsmi
