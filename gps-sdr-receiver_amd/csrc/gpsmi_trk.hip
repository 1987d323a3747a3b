// Tracking: all channels of one device per call, closed loop or replay.
//
// Replaces the numeric part of gpslib.SatStream.process (reference
// src/gpslib.py:1141-1210).  A "job" is one (channel, block) pair; the closed
// loop runs the open channels of one block, replay runs nb x nch jobs at once
// from a recorded state table.  Three kernels per call, in stream order:
//
//   trk_corr_kernel   (gpsmi_trk_corr.h) one workgroup per (block, six
//                     channels): carrier wipe-off and fold of the centre
//                     corr_avg code periods (demodDoppler :1343-1346),
//                     2048-point FFT in LDS, x conj(replica spectrum), FFT
//                     again as the inverse, |.|, mean / std / first argmax and
//                     the neighbours of the peak (cacodeCorr :1315-1327,
//                     findCodePhase :1293-1304); CORR_MIN threshold,
//                     fitCodePhase (:1268-1290) and the DELAY the block is
//                     decoded with (:1181-1182).
//   trk_stream_kernel (gpsmi_trk_stream.h) the correlator: carrier-NCO mix of
//                     the whole block times the replica rolled by DELAY
//                     (decodeData :1400-1401), summed per code-period window
//                     (:1408-1420) including the partial first window and the
//                     carry into the next block (:1403-1405, :1440).
//   trk_epilogue_kernel  one wave per job, lane = prompt dump: window means,
//                     amplitude statistics (:1186-1188), phaseLockedLoop
//                     (:1215-1262) and the state update (:1178, :1205-1208).
//
// Loop-carried state lives in device memory; the closed loop needs no host
// round trip between blocks.
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include <hip/hip_ext.h>

#include "gpsmi_common.h"
#include "gpsmi_fft.h"

// Code that restates the reference's float32 arithmetic step by step (the phase
// argument of the carrier, the PLL, numpy's summation order) must not be fused into
// multiply-adds (hipcc contracts by default; see mul_rn/add_rn in gpsmi_common.h).  The streaming kernels (included below) keep the default.
#pragma clang fp contract(off)

namespace gpsmi {

constexpr float kTwoPiF = 6.28318530717958647692f;   // float32(2*pi), numpy's weak-scalar cast
constexpr float kPiF = 3.14159265358979323846f;

struct TrkParams {
    int cs;            // code samples (2048)
    int n_cyc;
    int corr_avg;
    float corr_min;
    float min_freq, max_freq;
    int nch;           // jobs per block
    int df_no;         // 1024 / n_cyc
    float t_last;      // SEC_TIME[NGPS-1]
    float om_min, om_max;   // float32(2*pi*MIN_FREQ), float32(2*pi*MAX_FREQ) from float64
    int flags;              // diagnostics, -DGPSMI_DIAG builds only (`make diag`, GPSMI_DEBUG_FLAGS): 1 no MAC,
                            // 2 no lane sums, 4 no mixed fix (vector correlator), 32 no fold, 64 no
                            // transforms (code-phase correlation); the shipped library ignores them
};

// per-job descriptor handed from the correlation kernel to the correlator and
// the epilogue: everything the correlator's set-up needs in one 32-byte load
struct __attribute__((aligned(32))) JobMid {
    int delay_used;    // DELAY the block is decoded with
    int active;
    int prn;
    float om;          // 2*pi*FREQ as the reference forms it
    float ph;          // PHASE at the start of the block
    int pad[3];
};

// a diagnostics bit of TrkParams::flags: constant false in the shipped library
__device__ __forceinline__ bool diag_flag(const TrkParams& P, int bit) {
#ifdef GPSMI_DIAG
    return (P.flags & bit) != 0;
#else
    (void)P; (void)bit;
    return false;
#endif
}

__device__ __forceinline__ float wave_sum_t(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ void wave_argmax_t(float& v, int& i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_down(v, o, 64);
        int oi = __shfl_down(i, o, 64);
        if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
    }
}

// omega as the reference forms 2*pi*freq for a float32 FREQ (numpy >= 2):
// float32(2*pi) * freq in float32.
__device__ __host__ __forceinline__ float omega_of(float freq) {
    return mul_rn(kTwoPiF, freq);
}

// carrier factor exp(-j(phase + omega t[k])) with the float32 phase argument
__device__ __forceinline__ float2 wipe(float2 x, float phase, float om, float tk) {
    float p = add_rn(phase, mul_rn(om, tk));
    float s, c;
    sincosf(p, &s, &c);
    return make_float2(c * x.x + s * x.y, c * x.y - s * x.x);
}

// fitCodePhase (gpslib.py:1268-1290) in double on the float32 correlation values
__device__ inline double fit_code_phase(double lo, double pk, double hi, int mx) {
    double tri = (lo > hi) ? 0.5 * (hi - lo) / (pk - hi) : 0.5 * (hi - lo) / (pk - lo);
    double par = 0.5 * (hi - lo) / (2.0 * pk - hi - lo);
    return (double)mx + 0.5 * (tri + par);
}

}  // namespace gpsmi

#pragma clang fp contract(fast)
#include "gpsmi_trk_stream.h"
#include "gpsmi_trk_corr.h"
#include "gpsmi_bigfft.h"
#include "gpsmi_pfa.h"
#include "gpsmi_trk_general.h"
#pragma clang fp contract(off)
#include "gpsmi_trk_span.h"
#include "gpsmi_trk_span8.h"

namespace gpsmi {

// np.mean of a float32 array of n <= 128 elements: numpy's pairwise kernel
// (eight strided accumulators, tree-combined, tail added in order).
__device__ inline float np_sum_f32(const float* a, int n) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r = add_rn(r, a[i]);
        return r;
    }
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int j = 0; j < 8; ++j) r[j] = add_rn(r[j], a[i + j]);
    float res = add_rn(add_rn(add_rn(r[0], r[1]), add_rn(r[2], r[3])),
                          add_rn(add_rn(r[4], r[5]), add_rn(r[6], r[7])));
    for (; i < n; ++i) res = add_rn(res, a[i]);
    return res;
}

// The same sum by one wave, every lane returning it: lanes 0..7 each carry one of numpy's eight
// strided accumulators (n / 8 - 1 dependent adds instead of n), the tree ((r0+r1)+(r2+r3)) +
// ((r4+r5)+(r6+r7)) is three DPP steps (the adds commute, so the partners of a step hold the same
// bits), the tail is added in order.  a: LDS, n <= 128.
__device__ __forceinline__ float np_sum_f32_wave(const float* a, int n, int lane) {
    if (n < 8) return np_sum_f32(a, n);
    const int j = lane & 7, body = n - (n % 8);
    float r = a[j];
    for (int i = 8; i < body; i += 8) r = add_rn(r, a[i + j]);
    r = dpp_add0<0xB1, 0xF>(r);          // quad_perm [1,0,3,2]: r0+r1 | r2+r3 | r4+r5 | r6+r7
    r = dpp_add0<0x4E, 0xF>(r);          // quad_perm [2,3,0,1]: (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
    r = dpp_add0<0x141, 0xF>(r);         // row_half_mirror: lane j with lane 7 - j
    for (int i = body; i < n; ++i) r = add_rn(r, a[i]);
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, r)));
}

// The per-job part of a block behind the correlator, one wave, lane i = prompt dump i.
// Element-wise work (windows, |g|, atan, phase unwrapping by a lane prefix sum) is spread
// over the lanes; the few float32 sums are evaluated redundantly by every lane over small
// LDS arrays in numpy's own order (np_sum_f32), so the result does not depend on how lanes
// are scheduled.  S[0] head, S[q+1] window q, S[nc] tail (global memory or LDS).
__device__ __forceinline__ void epilogue_job(const gpsmi_trk_state& si, gpsmi_trk_state& so, int d,
                                             const float2* S, const TrkParams& P, gpsmi_trk_out& o,
                                             int lane, float* s_mag_w, float* s_dev_w, float* s_real_w,
                                             float* s_df_w) {
    const int cs = P.cs, nc = P.n_cyc;
    // scalar state (same address in every lane: one broadcast load each)
    const int nps = si.nps, df_len = si.df_len, was_locked = si.phase_locked;
    const float freq0 = si.freq, phase0 = si.phase, omega0 = si.omega0;
    const float prev_r = si.prev_sum_re, prev_i = si.prev_sum_im;
    const int edge_state0 = si.edge_state;
    const float prev_signal0 = si.prev_signal, std_dev0 = si.std_dev;
    for (int i = lane; i < df_len; i += 64) s_df_w[i] = si.df[i];

    // ---- prompt dumps: windows of decodeData (gpslib.py:1403-1420, :1440)
    const int n1 = nps + d;
    int nd;
    float gr = 0.f, gi = 0.f, car_r = 0.f, car_i = 0.f;
    int nps_new = 0;
    if (n1 == 0) {                       // no carry, delay 0: the rows are the windows
        nd = nc;
        if (lane < nd) { gr = S[lane + 1].x / (float)cs; gi = S[lane + 1].y / (float)cs; }
    } else {
        nd = (d == 0) ? nc + 1 : nc;     // delay 0: the last code period is complete
        if (lane == 0) {
            gr = (prev_r + S[0].x) / (float)n1;
            gi = (prev_i + S[0].y) / (float)n1;
        } else if (lane < nd) {
            gr = S[lane].x / (float)cs;
            gi = S[lane].y / (float)cs;
        }
        if (d != 0) { car_r = S[nc].x; car_i = S[nc].y; nps_new = cs - d; }
    }
    if (lane < GPSMI_MAX_DUMPS) {
        o.dumps[2 * lane] = lane < nd ? gr : 0.f;
        o.dumps[2 * lane + 1] = lane < nd ? gi : 0.f;
    }

    // ---- edge scan of decodeData (gpslib.py:1394-1398, :1421-1436): while PHASE_LOCKED (the flag
    // before this block's PLL), a dump is an edge when its sign differs from prevSign, the dump
    // before it carried prevSign's sign (prevSign * PREV_SIGNAL > 0) and the step between the two
    // exceeds MIN_EDGE_AMP = 3 * STD_DEV (of the block before).  prevSign only changes at an edge,
    // so the scan jumps from edge to edge over three wave-wide bit masks (scalar arithmetic, the
    // same in every lane): P / N = dump positive / negative, B = step large enough.
    unsigned long long edge_mask = 0;
    int edge_sign0 = 0, edge_state = edge_state0, ms_count = 0;
    float prev_signal = prev_signal0;
    if (was_locked) {
        const unsigned long long V = (1ull << nd) - 1;             // nd <= 33
        const float thr = mul_rn(3.0f, std_dev0);
        const float re_up = __shfl_up(gr, 1, 64);
        const float re_prev = lane == 0 ? prev_signal0 : re_up;
        const unsigned long long Pm = __ballot(gr > 0.f) & V, Nm = __ballot(gr < 0.f) & V;
        const unsigned long long Bm = __ballot(fabsf(sub_rn(gr, re_prev)) > thr) & V;
        const unsigned long long Pp = (Pm << 1) | (prev_signal0 > 0.f ? 1ull : 0ull);   // the dump before
        const unsigned long long Np = (Nm << 1) | (prev_signal0 < 0.f ? 1ull : 0ull);
        unsigned long long todo = V;
        int p = edge_state0 == 2 ? 0 : edge_state0;
        if (edge_state0 == 0) {            // EDGES[0] == 0: every dump stores its sign until one is non-zero
            const unsigned long long nz = Pm | Nm;
            if (nz == 0) {
                todo = 0;
            } else {
                const int j = __builtin_ctzll(nz);
                p = ((Pm >> j) & 1) ? 1 : -1;
                edge_sign0 = p;
                todo = V & ~((2ull << j) - 1);
            }
        }
        while (p != 0 && todo != 0) {
            const unsigned long long cand = (p > 0 ? (Pp & ~Pm) : (Np & ~Nm)) & Bm & todo;
            if (cand == 0) break;
            const int i = __builtin_ctzll(cand);
            edge_mask |= 1ull << i;
            p = ((Pm >> i) & 1) ? 1 : (((Nm >> i) & 1) ? -1 : 0);
            todo &= ~((2ull << i) - 1);
        }
        if (edge_state0 != 0 || edge_sign0 != 0) edge_state = p == 0 ? 2 : p;
        prev_signal = __shfl(gr, nd - 1, 64);                      // PREV_SIGNAL = m.real of the last dump
        ms_count = nd;
    }

    // ---- amplitude statistics (gpslib.py:1186-1187), float32 like numpy
    const float mag = hypotf(gr, gi);
    if (lane < nd) s_mag_w[lane] = mag;
    __builtin_amdgcn_wave_barrier();
    const float mmean = np_sum_f32_wave(s_mag_w, nd, lane) / (float)nd;
    {
        const float e = sub_rn(mag, mmean);
        if (lane < nd) s_dev_w[lane] = mul_rn(e, e);
    }
    __builtin_amdgcn_wave_barrier();
    const float sdev = sqrtf(np_sum_f32_wave(s_dev_w, nd, lane) / (float)nd);

    // ---- phaseLockedLoop (gpslib.py:1215-1262): unwrap by a lane prefix sum
    const float ph = atanf(gi / gr);
    const float ph_prev = __shfl_up(ph, 1, 64);
    float jump = 0.f;
    if (lane >= 1 && lane < nd) {
        const float delta = sub_rn(ph, ph_prev);
        if (fabsf(delta) > 2.0f) jump = (delta > 0.f) ? -1.f : 1.f;
    }
#pragma unroll
    for (int o2 = 1; o2 < 64; o2 <<= 1) {            // inclusive scan (small integers: exact)
        const float v = __shfl_up(jump, o2, 64);
        if (lane >= o2) jump += v;
    }
    const float real = (lane == 0) ? ph : add_rn(ph, mul_rn(jump, kPiF));
    if (lane < nd) s_real_w[lane] = real;
    __builtin_amdgcn_wave_barrier();
    const float offset = np_sum_f32(s_real_w + (nd - 4), 4) / 4.0f;
    const float pdev = np_sum_f32_wave(s_real_w, nd, lane) / (float)nd;
    const float max_df = 20.0f / (float)P.df_no;
    int locked = was_locked;
    int new_len;
    float df;
    if (locked) {
        const float mean_df = np_sum_f32_wave(s_df_w, df_len, lane) / (float)df_len;
        df = add_rn(pdev, mean_df);                 // DF_GAIN2 = 1
        if (fabsf(df) > max_df) df = (df > 0.f ? 1.f : -1.f) * max_df;
        const int shift = df_len >= P.df_no ? 1 : 0;   // drop the oldest entry
        new_len = df_len - shift + 1;
        for (int i = lane; i < new_len - 1; i += 64) so.df[i] = s_df_w[i + shift];
        if (lane == 0) so.df[new_len - 1] = df;
    } else {
        df = mul_rn(10.0f, pdev);                   // DF_GAIN1 = 10
        new_len = 1;
        if (lane == 0) so.df[0] = df;
    }
    if (fabsf(pdev) < 0.1f) locked = 1;

    // ---- state update (gpslib.py:1178 via :1345-1346, then :1205-1208)
    const float om = omega0 != 0.f ? omega0 : omega_of(freq0);
    float phase = add_rn(phase0, mul_rn(om, P.t_last));
    float mod = fmodf(phase, kTwoPiF);                 // np.remainder(phase, 2*pi)
    if (mod != 0.f && mod < 0.f) mod = add_rn(mod, kTwoPiF);
    phase = add_rn(mod, offset);
    float freq = add_rn(freq0, df);
    float om_new = 0.f;                                // FREQ is float32 from here on ...
    if (freq > P.max_freq) { freq = P.max_freq; om_new = P.om_max; }   // ... unless clamped to
    else if (freq < P.min_freq) { freq = P.min_freq; om_new = P.om_min; }  // a Python float

    if (lane == 0) {
        so.prn = si.prn;
        so.delay = d;
        so.freq = freq;
        so.phase = phase;
        so.phase_locked = locked;
        so.nps = nps_new;
        so.prev_sum_re = car_r;
        so.prev_sum_im = car_i;
        so.df_len = new_len;
        so.omega0 = om_new;
        so.edge_state = edge_state;
        so.prev_signal = prev_signal;
        so.std_dev = sdev;
        so.reserved = 0;
        o.n_dumps = nd;
        o.first_len = (n1 == 0) ? cs : n1;
        o.std_dev = sdev;
        o.amplitude = mmean / sdev;
        o.df = df;
        o.phase_shift = offset;
        o.freq = freq;
        o.phase = phase;
        o.phase_locked = locked;
        o.nps = nps_new;
        o.edge_mask = (uint32_t)edge_mask;
        o.edge_mask_hi = (uint32_t)(edge_mask >> 32);
        o.edge_sign0 = edge_sign0;
        o.ms_count = ms_count;
        o.reserved1 = 0;
    }
}

// a closed channel: its state row is copied through, its record is all-zero (prn = 0); every
// byte of an open channel's record is written by the correlation kernel and epilogue_job, so
// the result buffer needs no memset per launch
__device__ __forceinline__ void epilogue_closed(const gpsmi_trk_state& si, gpsmi_trk_state& so,
                                                gpsmi_trk_out& out, bool copy, int lane, int nlanes = 64) {
    if (copy) {
        const int* a = reinterpret_cast<const int*>(&si);
        int* b = reinterpret_cast<int*>(&so);
        for (int i = lane; i < (int)(sizeof(gpsmi_trk_state) / 4); i += nlanes) b[i] = a[i];
    }
    int* z = reinterpret_cast<int*>(&out);
    for (int i = lane; i < (int)(sizeof(gpsmi_trk_out) / 4); i += nlanes) z[i] = 0;
}

// One wave per job (four jobs per workgroup): the batch form.
__global__ __launch_bounds__(256) void trk_epilogue_kernel(
    const gpsmi_trk_state* __restrict__ st_in, gpsmi_trk_state* __restrict__ st_out,
    const JobMid* __restrict__ mid, const float2* __restrict__ partial, TrkParams P,
    int njobs, gpsmi_trk_out* __restrict__ out) {
    __shared__ float s_mag[4][40], s_dev[4][40], s_real[4][40], s_df[4][GPSMI_MAX_DF];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wave;
    if (job >= njobs) return;
    if (!mid[job].active) {
        epilogue_closed(st_in[job], st_out[job], out[job], st_out != st_in, lane);
        return;
    }
    epilogue_job(st_in[job], st_out[job], mid[job].delay_used, partial + (size_t)job * (P.n_cyc + 1), P,
                 out[job], lane, s_mag[wave], s_dev[wave], s_real[wave], s_df[wave]);
}

// ---- The batch form with EIGHT LANES per job (round 4; option "epilogue_form" = 1, the default).
// The wave-per-job kernel above spends a whole wave -- and, beside the code-phase correlation of the
// next batch, a wave slot with its registers on every SIMD of a CU for the ~10 us of its dependent
// chain -- on 33 dumps: 12288 jobs = 3072 workgroups that each keep one of that kernel's workgroups
// off a CU while they wait for their own loads.  Here lane j of a job's eight carries the dumps
// i = j, j + 8, j + 16, ... (slot k = i / 8; the (N_CYC + 1)-th dump is slot N_CYC / 8 of lane 0), a
// wave is a workgroup and takes eight jobs: 1536 one-wave workgroups instead of 3072 four-wave ones,
// an eighth of the wave slots.  The arithmetic is epilogue_job's, operation by operation:
//   * numpy's pairwise sum IS this layout: its eight strided accumulators r[j] = a[j] + a[j + 8] + ...
//     are the lanes' own slots added in slot order, the tree is the same three DPP steps (they stay
//     inside a group of eight lanes), the tail element comes from lane 0's last slot;
//   * the masks of the edge scan (dump positive / negative / step large) and the +-1 jumps of the phase
//     unwrapping are bits i of per-job 64-bit words, OR-ed over the eight lanes by DPP; the scan is the
//     same arithmetic on them in every lane of the group, the unwrap count of dump i is two popcounts
//     (small integers: exact, like the float prefix sum of the wave form);
//   * the neighbour dump (i - 1) is lane j - 1's slot k, or lane 7's slot k - 1 for j = 0: one
//     rotation inside the group per slot.
// Same bits as the wave form (tests/test_gpu_trk.py: both forms against each other and replay ==
// closed loop, whose single-block epilogue is the wave form).
template <int CTRL>
__device__ __forceinline__ unsigned dpp_or(unsigned v) {
    return v | (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// OR over the eight lanes of a group, in every lane of it
__device__ __forceinline__ unsigned long long or8(unsigned long long m) {
    unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
    lo = dpp_or<0xB1>(lo); hi = dpp_or<0xB1>(hi);       // quad_perm [1,0,3,2]
    lo = dpp_or<0x4E>(lo); hi = dpp_or<0x4E>(hi);       // quad_perm [2,3,0,1]
    lo = dpp_or<0x141>(lo); hi = dpp_or<0x141>(hi);     // row_half_mirror
    return ((unsigned long long)hi << 32) | lo;
}
// lane `l` of this lane's group of eight
__device__ __forceinline__ float grp_get(float v, int lane, int l) { return __shfl(v, (lane & ~7) | l, 64); }
// np.sum of a[0 .. n), element i in slot i / 8 of lane i % 8; n = 8 (K - 1) or 8 (K - 1) + 1
template <int K>
__device__ __forceinline__ float np_sum8(const float (&a)[K], int n, int lane) {
    float r = a[0];
#pragma unroll
    for (int k = 1; k < K - 1; ++k) r = add_rn(r, a[k]);
    r = dpp_add0<0xB1, 0xF>(r);
    r = dpp_add0<0x4E, 0xF>(r);
    r = dpp_add0<0x141, 0xF>(r);
    const float t = grp_get(a[K - 1], lane, 0);
    if (n > 8 * (K - 1)) r = add_rn(r, t);
    return r;
}
// np_sum_f32_wave for a group of eight lanes (a: LDS, any n <= 128), every lane of the group returning it
__device__ __forceinline__ float np_sum_f32_grp(const float* a, int n, int j) {
    if (n < 8) return np_sum_f32(a, n);
    const int body = n - (n % 8);
    float r = a[j];
    for (int i = 8; i < body; i += 8) r = add_rn(r, a[i + j]);
    r = dpp_add0<0xB1, 0xF>(r);
    r = dpp_add0<0x4E, 0xF>(r);
    r = dpp_add0<0x141, 0xF>(r);
    for (int i = body; i < n; ++i) r = add_rn(r, a[i]);
    return r;
}

template <int NCV>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(6, 6))) void trk_epilogue8_kernel(
    const gpsmi_trk_state* __restrict__ st_in, gpsmi_trk_state* __restrict__ st_out,
    const JobMid* __restrict__ mid, const float2* __restrict__ partial, TrkParams P,
    int njobs, gpsmi_trk_out* __restrict__ out) {
    constexpr int nc = NCV, K = NCV / 8 + 1, KD = (GPSMI_MAX_DUMPS + 7) / 8;
    static_assert(NCV % 8 == 0 && NCV + 1 <= GPSMI_MAX_DUMPS, "slots of eight dumps");
    __shared__ float s_real[8][NCV + 8], s_df[8][GPSMI_MAX_DF];
    const int lane = threadIdx.x & 63, j = lane & 7, jw = lane >> 3;
    const int job = blockIdx.x * 8 + jw;
    if (job >= njobs) return;                       // (whole groups: nothing below leaves a group)
    const gpsmi_trk_state& si = st_in[job];
    gpsmi_trk_state& so = st_out[job];
    gpsmi_trk_out& o = out[job];
    const JobMid md = mid[job];
    if (!md.active) {
        epilogue_closed(si, so, o, st_out != st_in, j, 8);
        return;
    }
    const float2* S = partial + (size_t)job * (nc + 1);
    const int d = md.delay_used, cs = P.cs;
    float* s_real_w = s_real[jw];
    float* s_df_w = s_df[jw];
    const int nps = si.nps, df_len = si.df_len, was_locked = si.phase_locked;
    const float freq0 = si.freq, phase0 = si.phase, omega0 = si.omega0;
    const float prev_r = si.prev_sum_re, prev_i = si.prev_sum_im;
    const int edge_state0 = si.edge_state;
    const float prev_signal0 = si.prev_signal, std_dev0 = si.std_dev;
    for (int i = j; i < df_len; i += 8) s_df_w[i] = si.df[i];

    // ---- prompt dumps (epilogue_job: gpslib.py:1403-1420, :1440)
    const int n1 = nps + d;
    int nd;
    float gr[K], gi[K], car_r = 0.f, car_i = 0.f;
    int nps_new = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) gr[k] = gi[k] = 0.f;
    if (n1 == 0) {
        nd = nc;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = j + 8 * k;
            if (i < nd) { gr[k] = S[i + 1].x / (float)cs; gi[k] = S[i + 1].y / (float)cs; }
        }
    } else {
        nd = (d == 0) ? nc + 1 : nc;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = j + 8 * k;
            if (i == 0) {
                gr[k] = (prev_r + S[0].x) / (float)n1;
                gi[k] = (prev_i + S[0].y) / (float)n1;
            } else if (i < nd) {
                gr[k] = S[i].x / (float)cs;
                gi[k] = S[i].y / (float)cs;
            }
        }
        if (d != 0) { car_r = S[nc].x; car_i = S[nc].y; nps_new = cs - d; }
    }
#pragma unroll
    for (int k = 0; k < KD; ++k) {
        const int i = j + 8 * k;
        if (i < GPSMI_MAX_DUMPS) {
            o.dumps[2 * i] = (k < K && i < nd) ? gr[k < K ? k : 0] : 0.f;
            o.dumps[2 * i + 1] = (k < K && i < nd) ? gi[k < K ? k : 0] : 0.f;
        }
    }
    // the dump before dump i: slot k of lane j - 1, for j = 0 slot k - 1 of lane 7
    const int rot = (lane & ~7) | ((j + 7) & 7);

    // ---- edge scan (epilogue_job: gpslib.py:1394-1398, :1421-1436)
    unsigned long long edge_mask = 0;
    int edge_sign0 = 0, edge_state = edge_state0, ms_count = 0;
    float prev_signal = prev_signal0;
    if (was_locked) {
        const unsigned long long V = (1ull << nd) - 1;
        const float thr = mul_rn(3.0f, std_dev0);
        unsigned long long Pm = 0, Nm = 0, Bm = 0;
        float before = prev_signal0;                   // lane 0: slot k - 1 of lane 7
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = j + 8 * k;
            const float up = __shfl(gr[k], rot, 64);
            const float re_prev = j == 0 ? before : up;
            before = up;
            if (i < nd) {
                if (gr[k] > 0.f) Pm |= 1ull << i;
                if (gr[k] < 0.f) Nm |= 1ull << i;
                if (fabsf(sub_rn(gr[k], re_prev)) > thr) Bm |= 1ull << i;
            }
        }
        Pm = or8(Pm); Nm = or8(Nm); Bm = or8(Bm);
        const unsigned long long Pp = (Pm << 1) | (prev_signal0 > 0.f ? 1ull : 0ull);
        const unsigned long long Np = (Nm << 1) | (prev_signal0 < 0.f ? 1ull : 0ull);
        unsigned long long todo = V;
        int p = edge_state0 == 2 ? 0 : edge_state0;
        if (edge_state0 == 0) {
            const unsigned long long nz = Pm | Nm;
            if (nz == 0) {
                todo = 0;
            } else {
                const int q = __builtin_ctzll(nz);
                p = ((Pm >> q) & 1) ? 1 : -1;
                edge_sign0 = p;
                todo = V & ~((2ull << q) - 1);
            }
        }
        while (p != 0 && todo != 0) {
            const unsigned long long cand = (p > 0 ? (Pp & ~Pm) : (Np & ~Nm)) & Bm & todo;
            if (cand == 0) break;
            const int i = __builtin_ctzll(cand);
            edge_mask |= 1ull << i;
            p = ((Pm >> i) & 1) ? 1 : (((Nm >> i) & 1) ? -1 : 0);
            todo &= ~((2ull << i) - 1);
        }
        if (edge_state0 != 0 || edge_sign0 != 0) edge_state = p == 0 ? 2 : p;
        const float last_full = grp_get(gr[K - 2], lane, 7), last_extra = grp_get(gr[K - 1], lane, 0);
        prev_signal = nd == nc + 1 ? last_extra : last_full;       // the real part of dump nd - 1
        ms_count = nd;
    }

    // ---- amplitude statistics (epilogue_job: gpslib.py:1186-1187)
    float mag[K], dev[K];
#pragma unroll
    for (int k = 0; k < K; ++k) mag[k] = hypotf(gr[k], gi[k]);
    const float mmean = np_sum8<K>(mag, nd, lane) / (float)nd;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float e = sub_rn(mag[k], mmean);
        dev[k] = mul_rn(e, e);
    }
    const float sdev = sqrtf(np_sum8<K>(dev, nd, lane) / (float)nd);

    // ---- phaseLockedLoop (epilogue_job: gpslib.py:1215-1262)
    float ph[K], real[K];
    unsigned long long Jp = 0, Jn = 0;                 // dumps whose unwrap step is +1 / -1
    {
        float before = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = j + 8 * k;
            ph[k] = atanf(gi[k] / gr[k]);
            const float up = __shfl(ph[k], rot, 64);
            const float ph_prev = j == 0 ? before : up;
            before = up;
            if (i >= 1 && i < nd) {
                const float delta = sub_rn(ph[k], ph_prev);
                if (fabsf(delta) > 2.0f) {
                    if (delta > 0.f) Jn |= 1ull << i;
                    else Jp |= 1ull << i;
                }
            }
        }
    }
    Jp = or8(Jp); Jn = or8(Jn);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = j + 8 * k;
        const unsigned long long upto = (2ull << i) - 1;               // dumps 0 .. i
        const float jump = (float)(__builtin_popcountll(Jp & upto) - __builtin_popcountll(Jn & upto));
        real[k] = (i == 0) ? ph[k] : add_rn(ph[k], mul_rn(jump, kPiF));
        if (i < nd) s_real_w[i] = real[k];
    }
    __builtin_amdgcn_wave_barrier();
    const float offset = np_sum_f32(s_real_w + (nd - 4), 4) / 4.0f;
    const float pdev = np_sum8<K>(real, nd, lane) / (float)nd;
    const float max_df = 20.0f / (float)P.df_no;
    int locked = was_locked;
    int new_len;
    float df;
    if (locked) {
        const float mean_df = np_sum_f32_grp(s_df_w, df_len, j) / (float)df_len;
        df = add_rn(pdev, mean_df);
        if (fabsf(df) > max_df) df = (df > 0.f ? 1.f : -1.f) * max_df;
        const int shift = df_len >= P.df_no ? 1 : 0;
        new_len = df_len - shift + 1;
        for (int i = j; i < new_len - 1; i += 8) so.df[i] = s_df_w[i + shift];
        if (j == 0) so.df[new_len - 1] = df;
    } else {
        df = mul_rn(10.0f, pdev);
        new_len = 1;
        if (j == 0) so.df[0] = df;
    }
    if (fabsf(pdev) < 0.1f) locked = 1;

    // ---- state update (epilogue_job: gpslib.py:1178 via :1345-1346, then :1205-1208)
    const float om = omega0 != 0.f ? omega0 : omega_of(freq0);
    float phase = add_rn(phase0, mul_rn(om, P.t_last));
    float mod = fmodf(phase, kTwoPiF);
    if (mod != 0.f && mod < 0.f) mod = add_rn(mod, kTwoPiF);
    phase = add_rn(mod, offset);
    float freq = add_rn(freq0, df);
    float om_new = 0.f;
    if (freq > P.max_freq) { freq = P.max_freq; om_new = P.om_max; }
    else if (freq < P.min_freq) { freq = P.min_freq; om_new = P.om_min; }

    if (j == 0) {
        so.prn = si.prn;
        so.delay = d;
        so.freq = freq;
        so.phase = phase;
        so.phase_locked = locked;
        so.nps = nps_new;
        so.prev_sum_re = car_r;
        so.prev_sum_im = car_i;
        so.df_len = new_len;
        so.omega0 = om_new;
        so.edge_state = edge_state;
        so.prev_signal = prev_signal;
        so.std_dev = sdev;
        so.reserved = 0;
        o.n_dumps = nd;
        o.first_len = (n1 == 0) ? cs : n1;
        o.std_dev = sdev;
        o.amplitude = mmean / sdev;
        o.df = df;
        o.phase_shift = offset;
        o.freq = freq;
        o.phase = phase;
        o.phase_locked = locked;
        o.nps = nps_new;
        o.edge_mask = (uint32_t)edge_mask;
        o.edge_mask_hi = (uint32_t)(edge_mask >> 32);
        o.edge_sign0 = edge_sign0;
        o.ms_count = ms_count;
        o.reserved1 = 0;
    }
}

// One workgroup per job: behind the single-block form of the span correlator (the closed
// loop).  The four waves add up the spans of one quarter each (independent loads, issued
// together), wave 0 adds the quarters in their fixed order, forms the windows and runs the
// per-job part.
template <int NC>
__global__ __launch_bounds__(256) void trk_epilogue_span_kernel(
    const gpsmi_trk_state* __restrict__ st_in, gpsmi_trk_state* __restrict__ st_out,
    const JobMid* __restrict__ mid, const float* __restrict__ rec, int ng_span, TrkParams P,
    int njobs, gpsmi_trk_out* __restrict__ out) {
    __shared__ float s_mag[40], s_dev[40], s_real[40], s_df[GPSMI_MAX_DF];
    __shared__ float q_hi[4][64], q_lo[4][64], s_hi[64], s_lo[64];
    __shared__ float2 s_S[GPSMI_MAX_DUMPS];
    __shared__ gpsmi_trk_state s_si;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x;
    // the state row is requested with the job entry (one trip to memory for both) and waits in
    // LDS for the per-job part
    if (wave == 0) {
        const int* a = reinterpret_cast<const int*>(&st_in[job]);
        int* sa = reinterpret_cast<int*>(&s_si);
        for (int i = lane; i < (int)(sizeof(gpsmi_trk_state) / 4); i += 64) sa[i] = a[i];
    }
    const JobMid md = mid[job];
    if (!md.active) {
        if (wave == 0) epilogue_closed(st_in[job], st_out[job], out[job], st_out != st_in, lane);
        return;
    }
    span_collect_quarter<1, NC>(rec, ng_span, job / P.nch, job % P.nch, md.delay_used, lane, wave,
                                q_hi[wave][lane], q_lo[wave][lane]);
    __syncthreads();
    if (wave != 0) return;
    float h = 0.f, l = 0.f;
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) { h += q_hi[Q][lane]; l += q_lo[Q][lane]; }
    s_hi[lane] = h;
    s_lo[lane] = l;
    __builtin_amdgcn_wave_barrier();
    span_windows<NC>(s_hi, s_lo, md.om, lane, s_S);
    __builtin_amdgcn_wave_barrier();
    epilogue_job(s_si, st_out[job], md.delay_used, s_S, P, out[job], lane, s_mag, s_dev, s_real, s_df);
}

// One wave per job (four jobs per workgroup) behind trk_span8_kernel: the wave adds up the 11 range
// records of its job (span8_collect) and runs the per-job part on the windows: no separate collect
// launch, no round trip of the window sums through memory.
__global__ __launch_bounds__(256) void trk_epilogue_span8_kernel(
    const gpsmi_trk_state* __restrict__ st_in, gpsmi_trk_state* __restrict__ st_out,
    const JobMid* __restrict__ mid, const float* __restrict__ rec, int ngroups, TrkParams P, int njobs,
    gpsmi_trk_out* __restrict__ out) {
    __shared__ float s_mag[4][40], s_dev[4][40], s_real[4][40], s_df[4][GPSMI_MAX_DF];
    __shared__ float s_hi[4][16], s_lo[4][16];
    __shared__ float2 s_S[4][GPSMI_MAX_DUMPS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int job = blockIdx.x * 4 + wave;
    if (job >= njobs) return;
    const JobMid md = mid[job];
    if (!md.active) {
        epilogue_closed(st_in[job], st_out[job], out[job], st_out != st_in, lane);
        return;
    }
    span8_collect(rec, ngroups, job / P.nch, job % P.nch, md.delay_used, md.om, lane, s_hi[wave], s_lo[wave],
                  s_S[wave]);
    epilogue_job(st_in[job], st_out[job], md.delay_used, s_S[wave], P, out[job], lane, s_mag[wave], s_dev[wave],
                 s_real[wave], s_df[wave]);
}

}  // namespace gpsmi

namespace gpsmi {
// Upload of a streamed block by the GPU itself: the workgroups read the page-locked host block
// over PCIe (16 bytes per lane, everything requested at once) and write the staging block in
// HBM.  On its own stream it runs beside the tracking kernels of the previous block; the copy
// engines' own path (hipMemcpyAsync) measured anywhere between 40 and 240 us per step for the
// same 128 KiB - 1 MiB blocks from one run to the next.
typedef unsigned stage_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void stage_copy_kernel(stage_u4* __restrict__ dst,
                                                         const stage_u4* __restrict__ src, size_t n16) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n16; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
}
}  // namespace gpsmi

using namespace gpsmi;

struct gpsmi_trk {
    gpsmi_cfg cfg;
    int max_ch = 0;
    int n_streams = 1;                   // independent receivers (IQ streams) of the closed loop; a state
                                         // row is stream * max_ch + channel
    int rows() const { return n_streams * max_ch; }
    int span_single_max = 80;            // (block, channel group) units up to which the span correlator runs
                                         // its single-block form (GPSMI_SPAN_SINGLE_MAX; measured per step with
                                         // the round's final kernels: 50 against 56 us at 64 units, 69 against 63
                                         // at 96, 81 against 66 at 128)
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;   // result read-back, overlaps the next replay run (the epilogue's
    bool own_copy_stream = false;        // stream unless option "copy_stream" = 1)
    hipStream_t alt_stream = nullptr;    // "corr_overlap": the runs of result slot 1 (slot 0 keeps `stream`)
    hipStream_t epi_stream = nullptr;    // replay: the epilogue of run k beside the code-phase
                                         // correlation of run k + 1 (the other slot's buffers)
    // streaming from host memory (gpsmi_trk_process_stream): two staging blocks filled on a stream
    // of their own, so that the upload of block k + 1 runs under the kernels of block k
    hipStream_t up_stream = nullptr;
    void* d_stage[2] = {nullptr, nullptr};
    size_t stage_bytes = 0;
    hipEvent_t up_done[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};
    hipEvent_t in_done[2] = {nullptr, nullptr};   // end of a streamed step (its iq read, its out written)
    bool in_pending[2] = {false, false};
    bool stage_used[2] = {false, false};
    int stage_idx = 0;
    size_t stream_inline_max = 8u << 20;     // bytes up to which a streamed block is copied on the main stream
    size_t stream_direct_max = 0;            // bytes up to which the kernels of a streamed step read a page-locked
                                             // block where it lies (no staging copy): option "stream_direct_max"
                                             // (GPSMI_STREAM_INLINE_MAX)
    hipEvent_t order = nullptr;          // orders other handles' streams behind this one
    hipEvent_t main_tail = nullptr;      // the event recorded behind the last work on `stream`, if any
    // two result slots: a replay run writes one while the other is still being copied out
    struct Slot {
        gpsmi_trk_out* d_out = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // start, corr done, correlator done, end
        hipEvent_t ready = nullptr, copied = nullptr;
        hipEvent_t corr_stop = nullptr;      // the event that carries the end stamp of the slot's timed correlator
        hipEvent_t corr_done = nullptr, epi_done = nullptr;   // correlator / epilogue of the slot's run
        bool copy_pending = false, epi_pending = false;
        int timing_pending = 0;              // the timing mode of the slot's run, until its times are taken
        JobMid* d_mid = nullptr;             // per-job descriptors and window sums of the slot's run
        float2* d_partial = nullptr;
        hipStream_t run_stream = nullptr;    // "corr_overlap": the stream the slot's runs are enqueued on (null: h->stream)
        float* d_rec = nullptr;              // raw sums of the single-block span correlator (gpsmi_trk_span.h):
                                             // per slot, the epilogue of run k reads them on its own stream
                                             // while run k + 1 writes the other slot's
    } slot[2];
    int cur = 0;                         // slot of the latest launch
    int timing = 1;                      // 1: record the four kernel-timing events per launch; 2: only the
                                         // begin / end stamps of the batch correlator's own dispatch (no
                                         // packet in the queue); 0: none
    float2* d_tw = nullptr;
    float* d_t32 = nullptr;
    float2* d_rep = nullptr;         // [GPSMI_MAX_PRN + 1][cs] spectra
    float* d_code = nullptr;         // [GPSMI_MAX_PRN + 1][cs] replica
    bool have_rep[GPSMI_MAX_PRN + 1] = {};
    float2* d_block = nullptr;       // staging for host blocks
    // closed loop (max_ch jobs)
    gpsmi_trk_state* d_state = nullptr;
    std::vector<gpsmi_trk_state> h_state;
    bool state_dirty_host = false;   // host copy newer than device
    // job buffers, sized for njobs_cap
    size_t njobs_cap = 0;
    gpsmi_trk_state* d_tab_in = nullptr;
    gpsmi_trk_state* d_tab_out = nullptr;
    int* d_forced = nullptr;
    float last_total_ms = 0.f, last_corr_ms = 0.f, last_cp_ms = 0.f;
    int replay_nb = 0;
    bool replay_forced = false;
    int corr_cg = 4;
    int corr_small1 = 384, corr_small2 = 1536;   // jobs per launch up to which 1 / 2 channels per correlation workgroup
    int done_by_dispatch = 1;          // option "done_by_dispatch" = 0: an event record behind the correlator instead
    // code_samples != 2048: time-domain correlation + chunked correlator
    bool general = false;
    static constexpr int stream_j = 8;   // code positions per lane of the vector correlator
    int mfma = 0;                    // 4: the span form of the matrix-pipe correlator (gpsmi_trk_span.h),
                                     // the default where it applies; 0: another correlator
    int codephase = 0;               // option "codephase" as taken at create time
    int corr_overlap = 0;            // option "corr_overlap": see gpsmi_trk_replay_run_async
    int fold_chunk = 0;              // option "fold_chunk": blocks per fold -> correlation piece at CS = 16368
    int epilogue_form = 1;           // option "epilogue_form": 1 = eight lanes per job in the batch epilogue
                                     // (trk_epilogue8_kernel), 0 = a wave per job (trk_epilogue_kernel); same bits
    float* d_code_eo = nullptr;      // [GPSMI_MAX_PRN + 1][...]: the replica re-cut for the matrix correlators.  Span form
                                     // (2048): four planes by index mod 4, entry h of plane e = replica[(4 h + e) mod 2048],
                                     // 1024 entries each (a lane's run never wraps); span8 form: two planes by index
                                     // parity, each twice over
    int n_cu = 256;                  // compute units of the device
    int iq_fmt = GPSMI_IQ_C64;       // what the iq pointers of process / replay point to
    int nchunks = 1;                 // spans of 256 * stream_j positions per code period
    float2* d_fold = nullptr; float* d_mag = nullptr; DirStats* d_stats = nullptr;
    int* d_xsel = nullptr; int* d_rsel = nullptr; float2* d_partial_g = nullptr;
    bool big = false;                // correlation through the 32768-point FFT pair
    float2* d_twN = nullptr; float2* d_RS = nullptr; float2* d_S = nullptr;
    bool pfa = false;                // ... or natively in LDS at 16368 samples (gpsmi_pfa.h)
    float2* d_RSp = nullptr;
    bool span8 = false;              // the matrix-pipe correlator for CS = 16368, N_CYC = 8 (gpsmi_trk_span8.h)
    TrkParams P;
    // gpsmi_trk_process_stream's submission thread (option "stream_thread", default on): the four
    // launches and the event record of a streamed step cost ~25 us of runtime calls on the host --
    // as long as the step runs on the GPU -- so they are made by a thread of the handle's own while
    // the caller prepares its next block.  Every other entry point first waits for this thread to
    // have nothing queued (trk_quiesce), so at any time only one thread works on the handle.
    struct StreamJob { const void* iq; size_t n; gpsmi_trk_out* out; };
    struct StreamWorker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv_job, cv_done;
        std::deque<StreamJob> q;
        bool stop = false;
        long long submitted = 0;     // jobs handed over
        long long cleared = 0;       // jobs whose step before last has been seen complete (the caller may go on)
        long long finished = 0;      // jobs fully enqueued on the device
        // the same three counters for the other thread to POLL before it goes to sleep on a condition
        // variable: being woken from a futex measured 50-100 us on these hosts, three hand-overs per
        // report block made gpsmi_trk_wait 290 us where the work outstanding was 60 us
        std::atomic<long long> a_submitted{0}, a_cleared{0}, a_finished{0};
        int err = 0;
        char errmsg[512] = "";
    };
    StreamWorker* worker = nullptr;
    int stream_thread = 1;
    int stream_depth = 2;            // calls a streamed step's buffers stay in use: 2 (gpsmi.h) or 3
    long long stat_backlog = 0;
    long long stat_quiesce_ns = 0, stat_evwait_ns = 0, stat_waits = 0;   // gpsmi_trk_wait behind streamed steps
    long long stat_wait_ns = 0, stat_launch_ns = 0, stat_steps = 0;   // streamed steps: host time waiting for the
                                                                      // step before last / making the runtime calls
};

constexpr int kSpanUnitsMax = 256;   // (block, channel group) units the single-block span form can serve (records)

static int trk_reserve(gpsmi_trk* h, size_t njobs) {
    if (njobs <= h->njobs_cap) return GPSMI_OK;
    void* olds[] = {h->d_tab_in, h->d_tab_out, h->d_forced, h->slot[0].d_mid, h->slot[1].d_mid,
                    h->slot[0].d_partial, h->slot[1].d_partial, h->slot[0].d_out, h->slot[1].d_out,
                    h->d_fold, h->d_mag, h->d_stats, h->d_xsel, h->d_rsel, h->d_partial_g,
                    h->slot[0].d_rec, h->slot[1].d_rec};
    for (void* p : olds)
        if (p) GPSMI_HIP(hipFree(p));
    h->d_tab_in = h->d_tab_out = nullptr; h->d_forced = nullptr;
    for (auto& sl : h->slot) { sl.d_mid = nullptr; sl.d_partial = nullptr; sl.d_out = nullptr; sl.d_rec = nullptr; }
    h->njobs_cap = 0;
    h->d_fold = nullptr; h->d_mag = nullptr; h->d_stats = nullptr; h->d_xsel = h->d_rsel = nullptr;
    h->d_partial_g = nullptr;
    if (h->mfma == 4)       // 32 records per (block, channel group) of the single-block span form
        for (auto& sl : h->slot)
            GPSMI_HIP(hipMalloc((void**)&sl.d_rec, (size_t)kSpanUnitsMax * 32 * kSpRecFloats * sizeof(float)));
    if (h->general) {
        const size_t cs = h->cfg.code_samples;
        GPSMI_HIP(hipMalloc((void**)&h->d_fold, njobs * cs * sizeof(float2)));
        if (!h->pfa) GPSMI_HIP(hipMalloc((void**)&h->d_mag, njobs * cs * sizeof(float)));
        GPSMI_HIP(hipMalloc((void**)&h->d_stats, njobs * sizeof(DirStats)));
        GPSMI_HIP(hipMalloc((void**)&h->d_xsel, njobs * sizeof(int)));
        GPSMI_HIP(hipMalloc((void**)&h->d_rsel, njobs * sizeof(int)));
    }
    if (h->span8) {         // one 2 KiB record per range of every (block, channel group)
        const size_t units = ((njobs + h->max_ch - 1) / h->max_ch) * ((h->max_ch + kSpCh - 1) / kSpCh);
        for (auto& sl : h->slot)
            GPSMI_HIP(hipMalloc((void**)&sl.d_rec, units * kS8Ranges * kS8RecFloats * sizeof(float)));
    }
    if (h->nchunks > 1 && !h->span8)
        GPSMI_HIP(hipMalloc((void**)&h->d_partial_g,
                            njobs * h->nchunks * (h->cfg.n_cyc + 1) * sizeof(float2)));
    GPSMI_HIP(hipMalloc((void**)&h->d_tab_in, njobs * sizeof(gpsmi_trk_state)));
    GPSMI_HIP(hipMalloc((void**)&h->d_tab_out, njobs * sizeof(gpsmi_trk_state)));
    GPSMI_HIP(hipMalloc((void**)&h->d_forced, njobs * sizeof(int)));
    for (auto& sl : h->slot) {
        GPSMI_HIP(hipMalloc((void**)&sl.d_mid, njobs * sizeof(JobMid)));
        GPSMI_HIP(hipMalloc((void**)&sl.d_partial, njobs * (h->cfg.n_cyc + 1) * sizeof(float2)));
        GPSMI_HIP(hipMalloc((void**)&sl.d_out, njobs * sizeof(gpsmi_trk_out)));
    }
    h->njobs_cap = njobs;
    return GPSMI_OK;
}

// the three kernels over njobs jobs on the handle's stream, events around them
static int trk_launch(gpsmi_trk* h, gpsmi_trk::Slot& sl, const void* d_iq_v,
                      const gpsmi_trk_state* st_in, gpsmi_trk_state* st_out, const int* forced,
                      int njobs, int nch, bool side_epilogue = false, hipEvent_t tail_stop = nullptr,
                      bool* tail_stop_used = nullptr) {
    // tail_stop: an event to carry the completion signal of the launch's LAST kernel (the epilogue),
    // instead of an event record behind it -- a record is a barrier packet, ~5 us of idle queue
    // between this step and the next (the streamed closed loop: 32.5 -> 27 us per block);
    // *tail_stop_used says whether the epilogue form at hand could take it
    TrkParams P = h->P;
    P.nch = nch;
    // the stream this launch goes to: the handle's, or ("corr_overlap") the slot's own, so that the
    // code-phase correlation of this batch may run beside the correlator of the batch before
    hipStream_t rs = sl.run_stream ? sl.run_stream : h->stream;
    h->main_tail = nullptr;
    const float2* d_iq = static_cast<const float2*>(d_iq_v);       // (raw uint16 when iq_fmt says so)
    const bool u8 = h->iq_fmt == GPSMI_IQ_U8;
    sl.corr_stop = sl.ev[2];
    const bool timed = h->timing == 1;   // each event record is a barrier packet (~5 us of bubble)
    const bool corr_stamps = h->timing == 2;
    if (timed) GPSMI_HIP(hipEventRecord(sl.ev[0], rs));
    const int nblocks = njobs / nch;
    const int ngroups = (nch + kGroupCh - 1) / kGroupCh;
    const dim3 sgrid(((nblocks + 7) / 8) * 8 * ngroups);
    // a launch too small to fill the CUs with whole blocks (the closed loop): every span of
    // a block is a wave of its own; same bits as the batch form (gpsmi_trk_span.h)
    const int ng_span = (nch + kSpCh - 1) / kSpCh;
    const bool span_single = h->mfma == 4 && nblocks * ng_span <= h->span_single_max;
    // ---- code-phase correlation
    if (h->general) {
        const int cs = P.cs;
        // The folded samples (nch x cs complex64 per block: 805 MB for a 512-block batch at 16368) are an
        // intermediate between two kernels: with the native-length correlation they go through in
        // pieces of `fold_chunk` blocks that reuse one scratch area, so that what the fold writes is
        // still in the memory-side cache when the correlation reads it (option "fold_chunk"; 0: the
        // whole launch at once)
        const int chunk = (h->pfa && h->fold_chunk > 0 && h->fold_chunk < nblocks) ? h->fold_chunk : nblocks;
        for (int b0 = 0; b0 < nblocks; b0 += chunk) {
            const int nbc = nblocks - b0 < chunk ? nblocks - b0 : chunk;
            hipLaunchKernelGGL(trk_fold_general_kernel, dim3((cs + 255) / 256, nbc), dim3(256), 0,
                               rs, d_iq, h->d_t32, st_in, P, h->d_fold, h->d_xsel, h->d_rsel,
                               sl.d_mid, b0);
            if (h->pfa)              // transform, product, transform and statistics in one launch
                pfa_corr_launch(rs, h->d_fold, h->d_xsel, h->d_rsel, nbc * nch, h->d_RSp, h->d_stats, b0 * nch);
        }
        if (h->pfa) {
        } else if (h->big)
            big_corr_launch(rs, h->d_fold, h->d_xsel, h->d_rsel, njobs, cs, h->d_RS, h->d_S,
                            h->d_tw, h->d_twN, h->d_mag);
        else
            hipLaunchKernelGGL(circ_corr_direct_kernel,
                               dim3((cs + kDirLagsPerWg - 1) / kDirLagsPerWg, njobs), dim3(256), 0,
                               rs, h->d_fold, h->d_code, h->d_xsel, h->d_rsel, cs,
                               h->d_mag);
        if (!h->pfa)
            hipLaunchKernelGGL(corr_stats_kernel, dim3(njobs), dim3(256), 0, rs, h->d_mag, cs,
                               h->d_stats);
        hipLaunchKernelGGL(trk_decide_kernel, dim3((njobs + 255) / 256), dim3(256), 0, rs,
                           h->d_stats, forced, P, njobs, sl.d_out, sl.d_mid);
    } else {
        // channels per correlation workgroup: fewer channels = fewer live accumulators
        // = more workgroups per CU for the barrier-heavy FFT phase (GPSMI_CORR_CG to tune)
        // (a single block, the closed loop, is latency-bound: spread it over more CUs)
        // (GPSMI_CORR_SMALL: job counts up to which one / two channels per workgroup are taken; measured
        // with R batched receivers of 12 channels, tools/batched_bench.py)
        const int jobs_all = nblocks * nch;
        const int cg = jobs_all <= h->corr_small1 ? 1 : (jobs_all <= h->corr_small2 ? 2 : h->corr_cg);
        const int ng = (nch + cg - 1) / cg;
        const dim3 cgrid(corr_grid(nblocks, ng));
#define GPSMI_LAUNCH_CORR(CGV)                                                                          \
    do {                                                                                              \
        if (u8)                                                                                       \
            hipLaunchKernelGGL((trk_corr_kernel<CGV, 1>), cgrid, dim3(256), 0, rs, d_iq_v,     \
                               st_in, forced, h->d_rep, h->d_tw, P, ng, nblocks, sl.d_out, sl.d_mid); \
        else                                                                                          \
            hipLaunchKernelGGL((trk_corr_kernel<CGV, 0>), cgrid, dim3(256), 0, rs, d_iq_v,     \
                               st_in, forced, h->d_rep, h->d_tw, P, ng, nblocks, sl.d_out, sl.d_mid); \
    } while (0)
        if (cg == 6) GPSMI_LAUNCH_CORR(6);
        else if (cg == 4) GPSMI_LAUNCH_CORR(4);
        else if (cg == 2) GPSMI_LAUNCH_CORR(2);
        else GPSMI_LAUNCH_CORR(1);
#undef GPSMI_LAUNCH_CORR
    }
    // ---- the correlator.  When a launch is timed, the two events of the batch form of the span
    // correlator are the dispatch's own begin / end stamps (hipExtLaunchKernel: what a kernel
    // trace reports), not event records around it: no barrier packets next to the kernel and
    // no launch gap inside the pair.
    const bool ext_timed = (timed || corr_stamps) && h->mfma == 4 && !span_single;
    bool corr_done_recorded = false;
    // batch form of the span correlator: persistent workgroups, two per CU, an equal number of
    // (block, channel group) units each
    const int span_units = nblocks * ng_span, span_slots = sp_wg_per_cu(P.n_cyc) * h->n_cu;
    const int span_per = (span_units + span_slots - 1) / span_slots;
    const dim3 span_grid((span_units + span_per - 1) / (span_per > 0 ? span_per : 1));
    if (timed && !ext_timed) GPSMI_HIP(hipEventRecord(sl.ev[1], rs));
    if (h->mfma) {                         // the correlator on the matrix pipe (span form, N_CYC = 32 / 16 / 8)
        const int ng12 = (nch + kSpCh - 1) / kSpCh;
        const JobMid* cmid = sl.d_mid;
        const float* ceo = h->d_code_eo;
        // one launch of trk_span_kernel<NSPANS, WAVES, FMT, 0, NC>: with the dispatch's own begin / end
        // stamps (ext, both events), with its completion signal as `stop` alone, or plainly
#define GPSMI_LAUNCH_SPAN(NSP, WV, FMTV, NCV, GRID, BLOCK)                                                     \
    do {                                                                                                      \
        if (ext_timed && (NSP) == 8)                                                                          \
            hipExtLaunchKernelGGL((trk_span_kernel<NSP, WV, FMTV, 0, NCV>), GRID, BLOCK, 0, rs, sl.ev[1],     \
                                  sl.corr_stop, 0, d_iq_v, cmid, ceo, P, ng12, nblocks, sl.d_rec,             \
                                  sl.d_partial);                                                              \
        else if (by_dispatch && (NSP) == 8)                                                                   \
            hipExtLaunchKernelGGL((trk_span_kernel<NSP, WV, FMTV, 0, NCV>), GRID, BLOCK, 0, rs, nullptr,      \
                                  sl.corr_done, 0, d_iq_v, cmid, ceo, P, ng12, nblocks, sl.d_rec,             \
                                  sl.d_partial);                                                              \
        else                                                                                                  \
            hipLaunchKernelGGL((trk_span_kernel<NSP, WV, FMTV, 0, NCV>), GRID, BLOCK, 0, rs, d_iq_v, cmid,    \
                               ceo, P, ng12, nblocks, sl.d_rec, sl.d_partial);                                \
    } while (0)
#define GPSMI_LAUNCH_SPAN_NC(NCV)                                                                       \
    do {                                                                                               \
        if (span_single && u8) GPSMI_LAUNCH_SPAN(1, 1, 1, NCV, dim3(nblocks * ng12 * 32), dim3(64));   \
        else if (span_single) GPSMI_LAUNCH_SPAN(1, 1, 0, NCV, dim3(nblocks * ng12 * 32), dim3(64));    \
        else if (u8) GPSMI_LAUNCH_SPAN(8, 4, 1, NCV, span_grid, dim3(256));                            \
        else GPSMI_LAUNCH_SPAN(8, 4, 0, NCV, span_grid, dim3(256));                                    \
    } while (0)
        // replay: the event the epilogue stream (and a search) waits for is the completion signal of
        // this very dispatch, not a record behind it - a record is one more barrier packet between
        // this kernel and the next batch's first one
        const bool by_dispatch = !span_single && side_epilogue && h->done_by_dispatch;
        // (a timed launch that is also the one the epilogue stream waits for: ONE stop event serves
        // both -- the dispatch's completion signal -- instead of a record packet behind the kernel)
        sl.corr_stop = (ext_timed && by_dispatch) ? sl.corr_done : sl.ev[2];
        if (P.n_cyc == 32) GPSMI_LAUNCH_SPAN_NC(32);
        else if (P.n_cyc == 16) GPSMI_LAUNCH_SPAN_NC(16);
        else GPSMI_LAUNCH_SPAN_NC(8);
#undef GPSMI_LAUNCH_SPAN_NC
#undef GPSMI_LAUNCH_SPAN
        corr_done_recorded = by_dispatch;
    } else if (h->span8) {                 // CS = 16368, N_CYC = 8 on the matrix pipe
        const int ng12 = (nch + kSpCh - 1) / kSpCh;
        const int nwaves = nblocks * ng12 * kS8Ranges;
        hipLaunchKernelGGL(trk_span8_kernel, dim3((nwaves + 3) / 4), dim3(256), 0, rs, d_iq,
                           sl.d_mid, h->d_code_eo, P, ng12, nblocks, sl.d_rec);
    } else {                               // the vector correlator (other block / code lengths)
        const dim3 grid(sgrid.x, h->nchunks), block(kStreamThreads);
        float2* pdst = h->nchunks > 1 ? h->d_partial_g : sl.d_partial;
#define GPSMI_LAUNCH_STREAM(NC, POW2, J)                                                        \
    hipLaunchKernelGGL((trk_stream_kernel<NC, POW2, J>), grid, block, 0, rs, d_iq, st_in, \
                       sl.d_mid, h->d_code, P, ngroups, nblocks, pdst)
#define GPSMI_LAUNCH_STREAM_NC(POW2, J)                 \
    do {                                                \
        if (P.n_cyc == 32) GPSMI_LAUNCH_STREAM(32, POW2, J);      \
        else if (P.n_cyc == 16) GPSMI_LAUNCH_STREAM(16, POW2, J); \
        else GPSMI_LAUNCH_STREAM(8, POW2, J);           \
    } while (0)
        if (h->general) GPSMI_LAUNCH_STREAM_NC(false, 8);
        else GPSMI_LAUNCH_STREAM_NC(true, 8);
#undef GPSMI_LAUNCH_STREAM_NC
#undef GPSMI_LAUNCH_STREAM
        if (h->nchunks > 1) {
            const int per_job = P.n_cyc + 1;
            hipLaunchKernelGGL(trk_partial_reduce_kernel, dim3((njobs * per_job + 255) / 256),
                               dim3(256), 0, rs, h->d_partial_g, h->nchunks, per_job, njobs,
                               sl.d_mid, sl.d_partial);
        }
    }
    if (timed && !ext_timed) GPSMI_HIP(hipEventRecord(sl.ev[2], rs));
    // replay: the epilogue goes to a stream of its own behind the correlator, so that the
    // code-phase correlation of the next run (other slot, main stream) starts at once
    hipStream_t es = rs;
    if (side_epilogue) {
        es = h->epi_stream;
        if (!corr_done_recorded) GPSMI_HIP(hipEventRecord(sl.corr_done, rs));
        h->main_tail = sl.corr_done;          // (gpsmi_acq_after_trk orders the search behind this one)
        GPSMI_HIP(hipStreamWaitEvent(es, sl.corr_done, 0));
    }
    const gpsmi_trk_state* c_in = st_in;
    const JobMid* c_mid = sl.d_mid;
    const float* c_rec = sl.d_rec;
    const float2* c_partial = sl.d_partial;
    bool stop_used = false;
#define GPSMI_LAUNCH_EPI_SPAN(NCV)                                                                         \
    do {                                                                                                  \
        if (tail_stop) {                                                                                  \
            hipExtLaunchKernelGGL(trk_epilogue_span_kernel<NCV>, dim3(njobs), dim3(256), 0, es, nullptr,  \
                                  tail_stop, 0, c_in, st_out, c_mid, c_rec, ng_span, P, njobs, sl.d_out); \
            stop_used = true;                                                                             \
        } else {                                                                                          \
            hipLaunchKernelGGL(trk_epilogue_span_kernel<NCV>, dim3(njobs), dim3(256), 0, es, c_in,        \
                               st_out, c_mid, c_rec, ng_span, P, njobs, sl.d_out);                        \
        }                                                                                                 \
    } while (0)
    if (h->span8)
        hipLaunchKernelGGL(trk_epilogue_span8_kernel, dim3((njobs + 3) / 4), dim3(256), 0, es, st_in,
                           st_out, sl.d_mid, sl.d_rec, ng_span, P, njobs, sl.d_out);
    else if (span_single && P.n_cyc == 32)
        GPSMI_LAUNCH_EPI_SPAN(32);
    else if (span_single && P.n_cyc == 16)
        GPSMI_LAUNCH_EPI_SPAN(16);
    else if (span_single)
        GPSMI_LAUNCH_EPI_SPAN(8);
    else if (h->epilogue_form == 0) {        // a wave per job
        if (tail_stop) {
            hipExtLaunchKernelGGL(trk_epilogue_kernel, dim3((njobs + 3) / 4), dim3(256), 0, es, nullptr,
                                  tail_stop, 0, c_in, st_out, c_mid, c_partial, P, njobs, sl.d_out);
            stop_used = true;
        } else
            hipLaunchKernelGGL(trk_epilogue_kernel, dim3((njobs + 3) / 4), dim3(256), 0, es, st_in,
                               st_out, sl.d_mid, sl.d_partial, P, njobs, sl.d_out);
    } else {                                 // eight lanes per job (the default)
#define GPSMI_LAUNCH_EPI8(NCV)                                                                           \
    do {                                                                                                \
        if (tail_stop) {                                                                                \
            hipExtLaunchKernelGGL(trk_epilogue8_kernel<NCV>, dim3((njobs + 7) / 8), dim3(64), 0, es,    \
                                  nullptr, tail_stop, 0, c_in, st_out, c_mid, c_partial, P, njobs,      \
                                  sl.d_out);                                                            \
            stop_used = true;                                                                           \
        } else {                                                                                        \
            hipLaunchKernelGGL(trk_epilogue8_kernel<NCV>, dim3((njobs + 7) / 8), dim3(64), 0, es, c_in, \
                               st_out, c_mid, c_partial, P, njobs, sl.d_out);                           \
        }                                                                                               \
    } while (0)
        if (P.n_cyc == 32) GPSMI_LAUNCH_EPI8(32);
        else if (P.n_cyc == 16) GPSMI_LAUNCH_EPI8(16);
        else GPSMI_LAUNCH_EPI8(8);
#undef GPSMI_LAUNCH_EPI8
    }
#undef GPSMI_LAUNCH_EPI_SPAN
    if (tail_stop_used) *tail_stop_used = stop_used;
    GPSMI_HIP(hipGetLastError());
    if (timed) GPSMI_HIP(hipEventRecord(sl.ev[3], es));
    if (side_epilogue) {
        GPSMI_HIP(hipEventRecord(sl.epi_done, es));
        sl.epi_pending = true;
    }
    return GPSMI_OK;
}

namespace gpsmi {
HandleSync acq_sync(gpsmi_acq* h);
HandleSync trk_sync(gpsmi_trk* h);
}  // namespace gpsmi

// kernel times of a finished launch into last_*_ms
static int trk_take_timing(gpsmi_trk* h, gpsmi_trk::Slot& sl) {
    if (!sl.timing_pending) return GPSMI_OK;
    if (sl.timing_pending == 1) {
        GPSMI_HIP(hipEventElapsedTime(&h->last_total_ms, sl.ev[0], sl.ev[3]));
        GPSMI_HIP(hipEventElapsedTime(&h->last_cp_ms, sl.ev[0], sl.ev[1]));
    }
    GPSMI_HIP(hipEventElapsedTime(&h->last_corr_ms, sl.ev[1], sl.corr_stop ? sl.corr_stop : sl.ev[2]));
    sl.timing_pending = 0;
    return GPSMI_OK;
}

// everything enqueued so far (kernels and read-backs) has finished
static int trk_settle(gpsmi_trk* h) {
    if (h->up_stream && (h->stage_used[0] || h->stage_used[1])) GPSMI_HIP(hipStreamSynchronize(h->up_stream));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    if (h->alt_stream) GPSMI_HIP(hipStreamSynchronize(h->alt_stream));
    GPSMI_HIP(hipStreamSynchronize(h->epi_stream));
    h->slot[0].epi_pending = h->slot[1].epi_pending = false;
    h->in_pending[0] = h->in_pending[1] = false;
    if (h->slot[0].copy_pending || h->slot[1].copy_pending)
        GPSMI_HIP(hipStreamSynchronize(h->copy_stream));
    for (int k = 0; k < 2; ++k) {          // older slot first: last_*_ms end up with the latest run
        gpsmi_trk::Slot& sl = h->slot[k == 0 ? (h->cur ^ 1) : h->cur];
        sl.copy_pending = false;
        int rc = trk_take_timing(h, sl);
        if (rc) return rc;
    }
    return GPSMI_OK;
}

static int trk_push_state(gpsmi_trk* h) {
    if (!h->state_dirty_host) return GPSMI_OK;
    h->main_tail = nullptr;              // (new work on `stream`: the recorded tail no longer covers it)
    GPSMI_HIP(hipMemcpyAsync(h->d_state, h->h_state.data(), h->rows() * sizeof(gpsmi_trk_state),
                             hipMemcpyHostToDevice, h->stream));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    h->state_dirty_host = false;
    return GPSMI_OK;
}

static int trk_pull_state(gpsmi_trk* h) {
    if (h->state_dirty_host) return GPSMI_OK;        // host copy is the newest
    GPSMI_HIP(hipMemcpyAsync(h->h_state.data(), h->d_state, h->rows() * sizeof(gpsmi_trk_state),
                             hipMemcpyDeviceToHost, h->stream));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    return GPSMI_OK;
}

// Is [p, p + bytes) page-locked host memory a kernel may address?  -> its device pointer.
// Wait for an event the GPU is about to signal by POLLING it: hipEventSynchronize may put the thread
// to sleep, and being woken by the driver measured ~250 us where the GPU had ~100 us of work left (the
// drop-in path waits like this once a second of signal: 353 -> ~110 us per report block).  After 2 ms
// the thread gives in and sleeps.
// Poll `done()` for up to `us` microseconds (then the caller sleeps on its condition variable).
template <class F>
static void trk_spin_until(F&& done, int us) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!done()) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return;
        __builtin_ia32_pause();
    }
}

static hipError_t trk_spin_wait(hipEvent_t ev) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(ev);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        __builtin_ia32_pause();
    }
    (void)hipGetLastError();                // (hipErrorNotReady is sticky in the thread's last-error slot)
    return hipEventSynchronize(ev);
}

// (memory from gpsmi_host_alloc is known without asking the runtime: hipPointerGetAttributes costs
// ~2 us a call, twice per step)
static bool trk_pinned_dev(gpsmi_trk* h, const void* p, size_t bytes, void** dev) {
    (void)h;
    if (bytes % 16 != 0 || ((uintptr_t)p & 15) != 0) return false;
    if (host_alloc_lookup(p, bytes, dev)) return true;
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, p) != hipSuccess) {
        (void)hipGetLastError();            // an unregistered pointer is not an error here
        return false;
    }
    if (at.type != hipMemoryTypeHost || at.devicePointer == nullptr) return false;
    hipPointerAttribute_t at_end{};
    const char* last = static_cast<const char*>(p) + bytes - 1;
    if (hipPointerGetAttributes(&at_end, last) != hipSuccess || at_end.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return false;
    }
    *dev = at.devicePointer;
    return true;
}

static int trk_stream_step(gpsmi_trk* h, const void* iq, size_t n, gpsmi_trk_out* out,
                           void (*cleared)(gpsmi_trk*) = nullptr);

static void trk_worker_cleared(gpsmi_trk* h) {
    gpsmi_trk::StreamWorker& w = *h->worker;
    {
        std::lock_guard<std::mutex> lock(w.m);
        w.cleared = w.finished + 1;             // (the job in hand)
        w.a_cleared.store(w.cleared, std::memory_order_release);
    }
    w.cv_done.notify_all();
}

static void trk_worker_main(gpsmi_trk* h) {
    gpsmi_trk::StreamWorker& w = *h->worker;
    (void)hipSetDevice(h->cfg.device);
    for (;;) {
        gpsmi_trk::StreamJob job;
        trk_spin_until([&] { return w.a_submitted.load(std::memory_order_acquire) > w.a_finished.load(std::memory_order_relaxed); }, 300);
        {
            std::unique_lock<std::mutex> lock(w.m);
            w.cv_job.wait(lock, [&] { return w.stop || !w.q.empty(); });
            if (w.q.empty()) return;            // (stop, and nothing left to enqueue)
            job = w.q.front();
        }
        int rc = w.err ? w.err : trk_stream_step(h, job.iq, job.n, job.out, trk_worker_cleared);
        {
            std::lock_guard<std::mutex> lock(w.m);
            if (rc && !w.err) {                 // the first failure is kept for the caller
                w.err = rc;
                snprintf(w.errmsg, sizeof(w.errmsg), "%s", last_error_buf());
            }
            w.q.pop_front();
            w.finished += 1;
            if (w.cleared < w.finished) w.cleared = w.finished;
            w.a_cleared.store(w.cleared, std::memory_order_release);
            w.a_finished.store(w.finished, std::memory_order_release);
        }
        w.cv_done.notify_all();
    }
}

// Nothing is queued on the submission thread and it is idle: from here on the calling thread is the
// only one working on the handle.  A failure of a streamed step surfaces here (once).
static int trk_quiesce(gpsmi_trk* h) {
    if (!h || !h->worker) return GPSMI_OK;
    gpsmi_trk::StreamWorker& w = *h->worker;
    trk_spin_until([&] { return w.a_finished.load(std::memory_order_acquire) == w.a_submitted.load(std::memory_order_relaxed); }, 1000);
    std::unique_lock<std::mutex> lock(w.m);
    w.cv_done.wait(lock, [&] { return w.finished == w.submitted; });
    if (w.err) {
        const int rc = w.err;
        w.err = 0;
        return fail(rc, "a streamed step failed: %s", w.errmsg);
    }
    return GPSMI_OK;
}

#define GPSMI_QUIESCE(h)                 \
    do {                                 \
        int rc_q__ = trk_quiesce(h);     \
        if (rc_q__) return rc_q__;       \
    } while (0)

namespace gpsmi {
HandleSync trk_sync(gpsmi_trk* h) {
    (void)trk_quiesce(h);                  // (a streamed step still being enqueued belongs in front)
    hipStream_t latest = h->slot[h->cur].run_stream ? h->slot[h->cur].run_stream : h->stream;
    return HandleSync{latest, h->order, h->cfg.device, h->main_tail};
}
}  // namespace gpsmi

extern "C" {

static int trk_build(const gpsmi_cfg* cfg, int max_ch, gpsmi_trk* h);

int gpsmi_trk_create(const gpsmi_cfg* cfg, int max_ch, gpsmi_trk** out) {
    GPSMI_REQUIRE(cfg && out, "null argument");
    *out = nullptr;
    GPSMI_REQUIRE(max_ch >= 1 && max_ch <= 4096, "max_ch out of range");
    GPSMI_REQUIRE(cfg->code_samples >= 1024 && cfg->code_samples <= 65536 &&
                      cfg->code_samples % 16 == 0,
                  "code_samples must be a multiple of 16 in 1024..65536");
    GPSMI_REQUIRE(cfg->n_cyc == 8 || cfg->n_cyc == 16 || cfg->n_cyc == 32,
                  "n_cyc must be 8, 16 or 32 (gpsglob.py:122)");
    GPSMI_REQUIRE(cfg->corr_avg >= 1, "corr_avg must be >= 1");
    GPSMI_REQUIRE(1024 / cfg->n_cyc <= GPSMI_MAX_DF, "n_cyc too small for the DF list");
    GPSMI_HIP(hipSetDevice(cfg->device));
    gpsmi_trk* h = new (std::nothrow) gpsmi_trk();
    if (!h) return fail(GPSMI_E_NOMEM, "out of host memory");
    h->cfg = *cfg;
    h->max_ch = max_ch;
    h->general = cfg->code_samples != kFftN;
    h->nchunks = (cfg->code_samples + 256 * h->stream_j - 1) / (256 * h->stream_j);
    const int rc = trk_build(cfg, max_ch, h);
    if (rc) {                       // nothing half-built leaves this function
        (void)gpsmi_trk_destroy(h);
        return rc;
    }
    *out = h;
    return GPSMI_OK;
}

}  // extern "C"

static int trk_build(const gpsmi_cfg* cfg, int max_ch, gpsmi_trk* h) {
    GPSMI_HIP(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device));
    GPSMI_HIP(hipStreamCreate(&h->stream));
    GPSMI_HIP(hipStreamCreate(&h->epi_stream));
    // The read-back of a replay run follows the run's epilogue anyway and is over long before the next
    // epilogue is due, so by default it shares the epilogue's stream: the HIP runtime maps streams onto
    // four hardware queues (one per pipe of the command processor), and a process with this handle's
    // compute and epilogue streams, an acquisition handle's stream and the null stream has four.  A
    // fifth stream shares a queue with one of them -- which one differs from run to run -- and a step
    // whose epilogue or search queues behind the compute stream's kernels takes 0.33 ms instead of
    // 0.23 (GPU_MAX_HW_QUEUES=3 forces it; more queues than pipes cost as much: DESIGN.md 4.6).
    // Option "copy_stream" = 1 (create time) gives the read-back a stream of its own again.
    long long own_copy = 0;
    default_opt("copy_stream", &own_copy, 0);
    h->own_copy_stream = own_copy != 0;
    if (h->own_copy_stream) GPSMI_HIP(hipStreamCreate(&h->copy_stream));
    else h->copy_stream = h->epi_stream;
    GPSMI_HIP(hipEventCreateWithFlags(&h->order, hipEventDisableTiming));
    for (auto& sl : h->slot) {
        for (auto& e : sl.ev) GPSMI_HIP(hipEventCreate(&e));
        GPSMI_HIP(hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
        GPSMI_HIP(hipEventCreateWithFlags(&sl.copied, hipEventDisableTiming));
        GPSMI_HIP(hipEventCreate(&sl.corr_done));   // (also used as a dispatch's stop event)
        GPSMI_HIP(hipEventCreateWithFlags(&sl.epi_done, hipEventDisableTiming));
    }
    std::vector<float2> tw;
    make_twiddles(tw);
    GPSMI_HIP(hipMalloc((void**)&h->d_tw, tw.size() * sizeof(float2)));
    GPSMI_HIP(hipMemcpy(h->d_tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    const int ngps = cfg->n_cyc * cfg->code_samples;
    const float fs = (float)(1000 * cfg->code_samples);
    std::vector<float> t32(ngps);
    for (int k = 0; k < ngps; ++k) t32[k] = (float)(k + 1) / fs;   // gpslib.py:1053-1054
    GPSMI_HIP(hipMalloc((void**)&h->d_t32, ngps * sizeof(float)));
    GPSMI_HIP(hipMemcpy(h->d_t32, t32.data(), ngps * sizeof(float), hipMemcpyHostToDevice));
    GPSMI_HIP(hipMalloc((void**)&h->d_rep, (size_t)(GPSMI_MAX_PRN + 1) * kFftN * sizeof(float2)));
    const size_t code_bytes = (size_t)(GPSMI_MAX_PRN + 1) * cfg->code_samples * sizeof(float);
    GPSMI_HIP(hipMalloc((void**)&h->d_code, code_bytes));
    GPSMI_HIP(hipMemset(h->d_code, 0, code_bytes));          // slot 0: closed channels
    long long want_matrix = 1;           // option "correlator": 0 keeps the vector kernel (gpsmi_trk_stream.h)
    default_opt("correlator", &want_matrix, 1);
    {
        // default for CS = 2048, N_CYC = 32: the span form of the MFMA correlator.  One form per handle:
        // the closed loop and the replay of a handle sum in the same order (bytewise equal results).
        h->mfma = (!h->general && want_matrix != 0) ? 4 : 0;          // (N_CYC = 32, 16 and 8: template parameter NC)
        if (h->mfma) {
            const size_t b2 = (size_t)(GPSMI_MAX_PRN + 1) * 2 * kFftN * sizeof(float);
            GPSMI_HIP(hipMalloc((void**)&h->d_code_eo, b2));
            GPSMI_HIP(hipMemset(h->d_code_eo, 0, b2));
        }
    }
    if (h->general) {
        long long forced = 0;                    // option "codephase": 1 keeps the time-domain kernel,
        default_opt("codephase", &forced, 0);    // 2 the zero-padded 32768-point pair
        h->codephase = (int)forced;
        h->pfa = cfg->code_samples == kPfaL && forced == 0;
        h->big = !h->pfa && 2 * cfg->code_samples - 1 <= kBigN && forced != 1;
    }
    if (h->pfa) {
        const size_t b = (size_t)(GPSMI_MAX_PRN + 1) * kPfaL * sizeof(float2);
        GPSMI_HIP(hipMalloc((void**)&h->d_RSp, b));
        GPSMI_HIP(hipMemset(h->d_RSp, 0, b));                // slot 0: closed channels
    }
    {
        h->span8 = h->general && cfg->code_samples == kS8Cs && cfg->n_cyc == kS8Rows && want_matrix != 0;
        if (h->span8) {
            const size_t b2 = (size_t)(GPSMI_MAX_PRN + 1) * 2 * kS8Cs * sizeof(float);
            GPSMI_HIP(hipMalloc((void**)&h->d_code_eo, b2));
            GPSMI_HIP(hipMemset(h->d_code_eo, 0, b2));
        }
    }
    if (h->big) {
        std::vector<float2> twn(kBigN);
        for (int k = 0; k < kBigN; ++k) {
            const double a = -2.0 * M_PI * (double)k / (double)kBigN;
            twn[k] = make_float2((float)cos(a), (float)sin(a));
        }
        GPSMI_HIP(hipMalloc((void**)&h->d_twN, kBigN * sizeof(float2)));
        GPSMI_HIP(hipMemcpy(h->d_twN, twn.data(), kBigN * sizeof(float2), hipMemcpyHostToDevice));
        const size_t rs_bytes = (size_t)(GPSMI_MAX_PRN + 1) * kBigN * sizeof(float2);
        GPSMI_HIP(hipMalloc((void**)&h->d_RS, rs_bytes));
        GPSMI_HIP(hipMemset(h->d_RS, 0, rs_bytes));          // slot 0: closed channels
        GPSMI_HIP(hipMalloc((void**)&h->d_S, (size_t)kBigChunkCells * kBigN * sizeof(float2)));
    }
    GPSMI_HIP(hipMalloc((void**)&h->d_block, (size_t)ngps * sizeof(float2)));
    GPSMI_HIP(hipMalloc((void**)&h->d_state, max_ch * sizeof(gpsmi_trk_state)));
    h->h_state.assign(max_ch, gpsmi_trk_state{});
    GPSMI_HIP(hipMemset(h->d_state, 0, max_ch * sizeof(gpsmi_trk_state)));
    TrkParams& P = h->P;
    P.cs = cfg->code_samples; P.n_cyc = cfg->n_cyc;
    P.corr_avg = cfg->corr_avg < cfg->n_cyc ? cfg->corr_avg : cfg->n_cyc;   // gpslib.py:1071
    P.corr_min = cfg->corr_min; P.min_freq = cfg->min_freq; P.max_freq = cfg->max_freq;
    P.nch = max_ch; P.df_no = 1024 / cfg->n_cyc; P.t_last = t32[ngps - 1];
    P.om_min = (float)(2.0 * M_PI * (double)cfg->min_freq);
    P.om_max = (float)(2.0 * M_PI * (double)cfg->max_freq);
    // the tuning options (gpsmi_trk_set_option changes them on a live handle); their defaults come
    // from gpsmi_set_default, else from the environment, else from the measurements quoted at the fields
    long long v = 0;
    default_opt("debug_flags", &v, 0);
    P.flags = (int)v;
    const struct { const char* key; long long fallback; } tun[] = {
        {"corr_cg", h->corr_cg}, {"span_single_max", h->span_single_max}, {"stream_inline_max", (long long)h->stream_inline_max},
        {"stream_direct_max", (long long)h->stream_direct_max},
        {"done_by_dispatch", h->done_by_dispatch}, {"corr_overlap", h->corr_overlap},
        {"stream_thread", h->stream_thread}, {"stream_depth", h->stream_depth}, {"fold_chunk", h->fold_chunk},
        {"epilogue_form", h->epilogue_form}};
    for (const auto& t : tun) {
        default_opt(t.key, &v, t.fallback);
        if (v != t.fallback) (void)gpsmi_trk_set_option(h, t.key, v);   // (an out-of-range default is ignored)
    }
    long long s1 = h->corr_small1, s2 = h->corr_small2;                 // (a pair: taken together)
    default_opt("corr_small1", &s1, s1);
    default_opt("corr_small2", &s2, s2);
    if (s1 >= 0 && s2 >= s1 && s2 <= (1 << 24)) { h->corr_small1 = (int)s1; h->corr_small2 = (int)s2; }
    return trk_reserve(h, max_ch);
}

extern "C" {

int gpsmi_trk_destroy(gpsmi_trk* h) {
    if (!h) return GPSMI_OK;
    if (h->worker) {
        (void)trk_quiesce(h);
        {
            std::lock_guard<std::mutex> lock(h->worker->m);
            h->worker->stop = true;
        }
        h->worker->cv_job.notify_all();
        h->worker->th.join();
        delete h->worker;
        h->worker = nullptr;
    }
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (h->copy_stream && h->own_copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    if (h->epi_stream) (void)hipStreamSynchronize(h->epi_stream);
    void* bufs[] = {h->d_tw, h->d_t32, h->d_rep, h->d_code, h->d_block, h->d_state, h->d_tab_in,
                    h->d_tab_out, h->d_forced, h->slot[0].d_mid, h->slot[1].d_mid, h->slot[0].d_partial,
                    h->slot[1].d_partial, h->slot[0].d_out,
                    h->slot[1].d_out, h->d_fold,
                    h->d_mag, h->d_stats, h->d_xsel, h->d_rsel, h->d_partial_g, h->d_twN, h->d_RS,
                    h->d_S, h->d_code_eo, h->slot[0].d_rec, h->slot[1].d_rec, h->d_RSp};
    for (void* b : bufs)
        if (b) (void)hipFree(b);
    for (auto& sl : h->slot) {
        for (auto e : sl.ev)
            if (e) (void)hipEventDestroy(e);
        if (sl.ready) (void)hipEventDestroy(sl.ready);
        if (sl.copied) (void)hipEventDestroy(sl.copied);
        if (sl.corr_done) (void)hipEventDestroy(sl.corr_done);
        if (sl.epi_done) (void)hipEventDestroy(sl.epi_done);
    }
    if (h->order) (void)hipEventDestroy(h->order);
    if (h->up_stream) (void)hipStreamSynchronize(h->up_stream);
    for (int k = 0; k < 2; ++k) {
        if (h->d_stage[k]) (void)hipFree(h->d_stage[k]);
        if (h->up_done[k]) (void)hipEventDestroy(h->up_done[k]);
        if (h->stage_free[k]) (void)hipEventDestroy(h->stage_free[k]);
        if (h->in_done[k]) (void)hipEventDestroy(h->in_done[k]);
    }
    if (h->up_stream) (void)hipStreamDestroy(h->up_stream);
    if (h->copy_stream && h->own_copy_stream) (void)hipStreamDestroy(h->copy_stream);
    if (h->alt_stream) { (void)hipStreamSynchronize(h->alt_stream); (void)hipStreamDestroy(h->alt_stream); }
    if (h->epi_stream) (void)hipStreamDestroy(h->epi_stream);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GPSMI_OK;
}

int gpsmi_trk_set_replica(gpsmi_trk* h, int prn, const float* replica, const float* spectrum) {
    GPSMI_REQUIRE(h && replica && (spectrum || h->general), "null argument");
    GPSMI_REQUIRE(prn >= 1 && prn <= GPSMI_MAX_PRN, "prn out of range 1..37");
    GPSMI_QUIESCE(h);
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    const size_t cs = h->cfg.code_samples;
    GPSMI_HIP(hipMemcpy(h->d_code + (size_t)prn * cs, replica, cs * sizeof(float),
                        hipMemcpyHostToDevice));
    if (h->mfma) {
        std::vector<float> eo(2 * kFftN);          // plane e of four, entry h = replica[(4 h + e) mod 2048], 1024 entries
        for (int e = 0; e < 4; ++e)
            for (int i = 0; i < kFftN / 2; ++i) eo[e * (kFftN / 2) + i] = replica[(4 * i + e) % kFftN];
        GPSMI_HIP(hipMemcpy(h->d_code_eo + (size_t)prn * 2 * kFftN, eo.data(), eo.size() * sizeof(float),
                            hipMemcpyHostToDevice));
    }
    if (h->span8) {                        // plane e, entry s = replica[2 (s mod cs / 2) + e], each plane twice
        std::vector<float> eo(2 * cs);
        for (int e = 0; e < 2; ++e)
            for (size_t i = 0; i < cs; ++i) eo[e * cs + i] = replica[2 * (i % (cs / 2)) + e];
        GPSMI_HIP(hipMemcpy(h->d_code_eo + (size_t)prn * 2 * cs, eo.data(), eo.size() * sizeof(float),
                            hipMemcpyHostToDevice));
    }
    if (!h->general)                       // the other path needs no 2048-point spectrum
        GPSMI_HIP(hipMemcpy(h->d_rep + (size_t)prn * kFftN, spectrum, kFftN * sizeof(float2),
                            hipMemcpyHostToDevice));
    if (h->big) big_replica_launch(h->stream, h->d_code, prn, (int)cs, h->d_RS, h->d_tw, h->d_twN);
    if (h->pfa) pfa_replica_launch(h->stream, h->d_code, prn, h->d_RSp);
    if (h->big || h->pfa) {
        GPSMI_HIP(hipGetLastError());
        GPSMI_HIP(hipStreamSynchronize(h->stream));
    }
    h->have_rep[prn] = true;
    return GPSMI_OK;
}

int gpsmi_trk_open(gpsmi_trk* h, int ch, int prn, float freq_hz, int delay) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(ch >= 0 && ch < h->rows(), "channel out of range");
    GPSMI_REQUIRE(prn >= 1 && prn <= GPSMI_MAX_PRN, "prn out of range 1..37");
    GPSMI_REQUIRE(delay >= 0 && delay < h->cfg.code_samples, "delay out of range");
    if (!h->have_rep[prn]) return fail(GPSMI_E_STATE, "no replica set for PRN %d", prn);
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_pull_state(h);
    if (rc) return rc;
    gpsmi_trk_state st{};                 // SatStream.__init__ (gpslib.py:1050-1091)
    st.prn = prn; st.delay = delay; st.freq = freq_hz; st.phase = 0.f;
    st.omega0 = (float)(2.0 * M_PI * (double)freq_hz);    // FREQ is a Python float here
    st.df_len = 1; st.df[0] = 0.f;
    st.std_dev = 0.005f;                  // STD_DEV "overwritten by 1st stream" (gpslib.py:1074)
    h->h_state[ch] = st;
    h->state_dirty_host = true;
    return GPSMI_OK;
}

int gpsmi_trk_close(gpsmi_trk* h, int ch) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(ch >= 0 && ch < h->rows(), "channel out of range");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_pull_state(h);
    if (rc) return rc;
    if (h->h_state[ch].prn == 0) return fail(GPSMI_E_STATE, "channel %d is not open", ch);
    h->h_state[ch] = gpsmi_trk_state{};
    h->state_dirty_host = true;
    return GPSMI_OK;
}

int gpsmi_trk_get_state(gpsmi_trk* h, int ch, gpsmi_trk_state* st) {
    GPSMI_REQUIRE(h && st, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(ch >= 0 && ch < h->rows(), "channel out of range");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_pull_state(h);
    if (rc) return rc;
    *st = h->h_state[ch];
    return GPSMI_OK;
}

int gpsmi_trk_set_state(gpsmi_trk* h, int ch, const gpsmi_trk_state* st) {
    GPSMI_REQUIRE(h && st, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(ch >= 0 && ch < h->rows(), "channel out of range");
    GPSMI_REQUIRE(st->prn >= 0 && st->prn <= GPSMI_MAX_PRN, "prn out of range");
    GPSMI_REQUIRE(st->delay >= 0 && st->delay < h->cfg.code_samples, "delay out of range");
    GPSMI_REQUIRE(st->nps >= 0 && st->nps <= h->cfg.code_samples, "nps out of range");
    GPSMI_REQUIRE(st->df_len >= 1 && st->df_len <= h->P.df_no, "df_len out of range");
    GPSMI_REQUIRE(st->edge_state >= -1 && st->edge_state <= 2, "edge_state out of range");
    if (st->prn && !h->have_rep[st->prn])
        return fail(GPSMI_E_STATE, "no replica set for PRN %d", st->prn);
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_pull_state(h);
    if (rc) return rc;
    h->h_state[ch] = *st;
    h->state_dirty_host = true;
    return GPSMI_OK;
}

int gpsmi_trk_erase_prev(gpsmi_trk* h, int ch) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(ch >= 0 && ch < h->rows(), "channel out of range");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_pull_state(h);
    if (rc) return rc;
    h->h_state[ch].nps = 0;               // PREV_SAMPLES = [] (gpslib.py:1095-1099)
    h->h_state[ch].prev_sum_re = h->h_state[ch].prev_sum_im = 0.f;
    h->h_state[ch].edge_state = 0;        // EDGES = [0]
    h->state_dirty_host = true;
    return GPSMI_OK;
}

int gpsmi_trk_process_dev(gpsmi_trk* h, const void* d_iq, size_t n, gpsmi_trk_out* out) {
    GPSMI_REQUIRE(h && d_iq, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(n == (size_t)h->n_streams * h->cfg.n_cyc * h->cfg.code_samples,
                  "input must hold one block of NGPS samples per stream");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_push_state(h);
    if (rc) return rc;
    if (h->slot[0].copy_pending || h->slot[1].copy_pending || h->slot[0].timing_pending ||
        h->slot[1].timing_pending || h->slot[0].epi_pending ||
        h->slot[1].epi_pending) {          // a replay still in flight owns the slots
        rc = trk_settle(h);
        if (rc) return rc;
    }
    gpsmi_trk::Slot& sl = h->slot[0];      // the closed loop needs one slot only
    h->cur = 0;
    sl.run_stream = nullptr;
    // the streams of a handle are the "blocks" of one launch: stream r reads block r of d_iq and
    // owns the state rows r * max_ch ..., updated in place
    rc = trk_launch(h, sl, d_iq, h->d_state, h->d_state, nullptr, h->rows(), h->max_ch);
    if (rc) return rc;
    if (out)
        GPSMI_HIP(hipMemcpyAsync(out, sl.d_out, h->rows() * sizeof(gpsmi_trk_out),
                                 hipMemcpyDeviceToHost, h->stream));
    // nothing to hand back: the block is enqueued, the state stays on the device and the
    // next call queues behind it (get_state / wait / a call with `out` synchronise)
    if (!out && h->timing != 1) return GPSMI_OK;
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    if (h->timing == 1) {
        GPSMI_HIP(hipEventElapsedTime(&h->last_total_ms, sl.ev[0], sl.ev[3]));
        GPSMI_HIP(hipEventElapsedTime(&h->last_corr_ms, sl.ev[1], sl.ev[2]));
        GPSMI_HIP(hipEventElapsedTime(&h->last_cp_ms, sl.ev[0], sl.ev[1]));
    }
    return GPSMI_OK;
}

int gpsmi_trk_process(gpsmi_trk* h, const float* iq, size_t n, gpsmi_trk_out* out) {
    GPSMI_REQUIRE(h && iq && out, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(n == (size_t)h->n_streams * h->cfg.n_cyc * h->cfg.code_samples,
                  "input must hold one block of NGPS samples per stream");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    h->main_tail = nullptr;
    GPSMI_HIP(hipMemcpyAsync(h->d_block, iq, n * (h->iq_fmt == GPSMI_IQ_U8 ? 2 : sizeof(float2)),
                             hipMemcpyHostToDevice, h->stream));
    return gpsmi_trk_process_dev(h, h->d_block, n, out);
}

// One streamed step: everything gpsmi_trk_process_stream promises, made by whichever thread works
// on the handle (the caller, or the handle's submission thread).  `cleared`, if given, is told as soon
// as the step before last is known to be complete.
static int trk_stream_step(gpsmi_trk* h, const void* iq, size_t n, gpsmi_trk_out* out,
                           void (*cleared)(gpsmi_trk*)) {
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    const size_t bytes = n * (h->iq_fmt == GPSMI_IQ_U8 ? 2 : sizeof(float2));
    if (!h->up_stream) {
        GPSMI_HIP(hipStreamCreate(&h->up_stream));
        for (int k = 0; k < 2; ++k) {
            GPSMI_HIP(hipEventCreateWithFlags(&h->up_done[k], hipEventDisableTiming));
            GPSMI_HIP(hipEventCreateWithFlags(&h->stage_free[k], hipEventDisableTiming));
            GPSMI_HIP(hipEventCreate(&h->in_done[k]));       // (also a dispatch's stop event)
        }
    }
    if (bytes > h->stage_bytes) {
        int rc = trk_settle(h);
        if (rc) return rc;
        for (int k = 0; k < 2; ++k) {
            if (h->d_stage[k]) GPSMI_HIP(hipFree(h->d_stage[k]));
            h->d_stage[k] = nullptr;
            h->stage_used[k] = false;
        }
        h->stage_bytes = 0;
        for (int k = 0; k < 2; ++k) GPSMI_HIP(hipMalloc(&h->d_stage[k], bytes));
        h->stage_bytes = bytes;
    }
    int rc = trk_push_state(h);
    if (rc) return rc;
    if (h->slot[0].copy_pending || h->slot[1].copy_pending || h->slot[0].timing_pending ||
        h->slot[1].timing_pending || h->slot[0].epi_pending || h->slot[1].epi_pending) {
        rc = trk_settle(h);                 // a replay still in flight owns the slots
        if (rc) return rc;
    }
    const int s = h->stage_idx;
    h->stage_idx ^= 1;
    // back-pressure: the step of the call before last has finished when this call returns (its iq may
    // be rewritten, its out is filled) -- the contract of gpsmi.h; the host therefore runs at most two
    // steps ahead of the device, which keeps one whole step queued behind the one that is running
    const auto t_w0 = std::chrono::steady_clock::now();
    if (h->in_pending[s]) {
        GPSMI_HIP(trk_spin_wait(h->in_done[s]));
        h->in_pending[s] = false;
    }
    const auto t_w1 = std::chrono::steady_clock::now();
    h->stat_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_w1 - t_w0).count();
    if (cleared) cleared(h);
    // page-locked memory is read by a kernel (see stage_copy_kernel); anything else -- pageable
    // memory would fault under a kernel -- goes through the runtime's copy
    void* iq_dev = nullptr;
    const bool pinned = trk_pinned_dev(h, iq, bytes, &iq_dev);
    // the records of a step go straight into a page-locked `out` (the two kernels that fill a record
    // store over PCIe: 376 bytes per channel) instead of through d_out and a copy command behind them
    void* out_dev = nullptr;
    const bool out_direct = out && trk_pinned_dev(h, out, (size_t)h->rows() * sizeof(gpsmi_trk_out), &out_dev);
    // Up to 8 MiB per step (measured: one receiver's 128 KiB to 64 receivers' 8 MiB of raw samples)
    // the block goes up IN FRONT of its own kernels on the main stream: the two event packets that
    // order an upload stream against the main one cost ~7 us per step, and a copy kernel beside the
    // tracking kernels takes from them what it hides (27.7 against 34.6 us per block for one receiver,
    // 50 against 58 us for eight; equal at 64).  Beyond that it goes up on the upload stream under the
    // previous step's kernels, as soon as the kernels that read this staging block two calls ago have
    // finished.  Either way the host never waits.
    // Small steps need no staging at all: the tracking kernels read the page-locked block over PCIe where it
    // lies (each sample is read once by the correlator and one row of it by the code-phase correlation), which
    // takes the copy kernel, its launch and the kernel boundary behind it out of the step.  The block
    // stays the caller's until the step is complete, as the contract of gpsmi.h says anyway.
    const bool direct_in = pinned && bytes <= h->stream_direct_max;
    const bool in_line = pinned && (direct_in || bytes <= h->stream_inline_max);
    hipStream_t us = in_line ? h->stream : h->up_stream;
    if (!in_line && h->stage_used[s]) GPSMI_HIP(hipStreamWaitEvent(h->up_stream, h->stage_free[s], 0));
    if (direct_in) {
        // (nothing to move)
    } else if (pinned) {
        const size_t n16 = bytes / 16;
        const unsigned grid = (unsigned)((n16 + 255) / 256 < 512 ? (n16 + 255) / 256 : 512);
        hipLaunchKernelGGL(stage_copy_kernel, dim3(grid), dim3(256), 0, us,
                           static_cast<stage_u4*>(h->d_stage[s]), static_cast<const stage_u4*>(iq_dev), n16);
    } else {
        GPSMI_HIP(hipMemcpyAsync(h->d_stage[s], iq, bytes, hipMemcpyHostToDevice, us));
    }
    if (!in_line) {
        GPSMI_HIP(hipEventRecord(h->up_done[s], h->up_stream));
        GPSMI_HIP(hipStreamWaitEvent(h->stream, h->up_done[s], 0));
    }
    gpsmi_trk::Slot& sl = h->slot[0];
    h->cur = 0;
    sl.run_stream = nullptr;
    const int timing = h->timing;
    h->timing = 0;                          // (no kernel-timing events in a streaming loop)
    gpsmi_trk_out* const d_out_keep = sl.d_out;
    if (out_direct) sl.d_out = static_cast<gpsmi_trk_out*>(out_dev);
    // the step's completion event = the completion signal of its last kernel, when nothing is
    // queued behind that kernel (no record copy, no upload-stream bookkeeping)
    const bool want_tail = in_line && (!out || out_direct);
    bool tail_used = false;
    rc = trk_launch(h, sl, direct_in ? iq_dev : h->d_stage[s], h->d_state, h->d_state, nullptr, h->rows(), h->max_ch,
                    /*side_epilogue=*/false, want_tail ? h->in_done[s] : nullptr, &tail_used);
    sl.d_out = d_out_keep;
    h->timing = timing;
    if (rc) return rc;
    if (!in_line) {
        GPSMI_HIP(hipEventRecord(h->stage_free[s], h->stream));
        h->stage_used[s] = true;
    } else {
        h->stage_used[s] = false;           // (same stream: the next writer of this block queues behind its readers)
    }
    if (out && !out_direct)
        GPSMI_HIP(hipMemcpyAsync(out, sl.d_out, h->rows() * sizeof(gpsmi_trk_out),
                                 hipMemcpyDeviceToHost, h->stream));
    if (!tail_used) GPSMI_HIP(hipEventRecord(h->in_done[s], h->stream));
    h->in_pending[s] = true;
    h->stat_launch_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_w1).count();
    h->stat_steps += 1;
    return GPSMI_OK;
}


int gpsmi_trk_process_stream(gpsmi_trk* h, const void* iq, size_t n, gpsmi_trk_out* out) {
    GPSMI_REQUIRE(h && iq, "null argument");
    GPSMI_REQUIRE(n == (size_t)h->n_streams * h->cfg.n_cyc * h->cfg.code_samples,
                  "input must hold one block of NGPS samples per stream");
    if (!h->stream_thread) {
        GPSMI_QUIESCE(h);
        return trk_stream_step(h, iq, n, out);
    }
    if (!h->worker) {
        h->worker = new (std::nothrow) gpsmi_trk::StreamWorker();
        if (!h->worker) return fail(GPSMI_E_NOMEM, "out of host memory");
        try {
            h->worker->th = std::thread(trk_worker_main, h);
        } catch (...) {                     // (no thread to be had: nothing may be thrown across the ABI;
            delete h->worker;               // the caller's thread makes the runtime calls itself)
            h->worker = nullptr;
            h->stream_thread = 0;
            return trk_stream_step(h, iq, n, out);
        }
    }
    gpsmi_trk::StreamWorker& w = *h->worker;
    std::unique_lock<std::mutex> lock(w.m);
    if (w.err) {                                // an earlier step failed: report it instead of queueing more
        w.cv_done.wait(lock, [&] { return w.finished == w.submitted; });
        const int rc = w.err;
        w.err = 0;
        return fail(rc, "a streamed step failed: %s", w.errmsg);
    }
    w.q.push_back({iq, n, out});
    const long long k = ++w.submitted;          // this job's ordinal, from 1
    w.a_submitted.store(k, std::memory_order_release);
    w.cv_job.notify_one();
    // the contract of gpsmi.h: return once the step of the call before last is complete -- the
    // submission thread says so when it has waited for that step on its way into this one
    // ("stream_depth" = 3: one step more -- the call returns when the step three calls back is
    // complete, so the caller can hand over its next block while this one is still being enqueued)
    const long long need = k - (h->stream_depth - 2);
    if (w.cleared < need) {
        lock.unlock();
        trk_spin_until([&] { return w.a_cleared.load(std::memory_order_acquire) >= need; }, 300);
        lock.lock();
    }
    w.cv_done.wait(lock, [&] { return w.cleared >= need; });
    return GPSMI_OK;
}

int gpsmi_trk_replay_load(gpsmi_trk* h, int nb, const gpsmi_trk_state* table,
                          const int32_t* delay_used) {
    GPSMI_REQUIRE(h && table, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(nb >= 1, "block count must be >= 1");
    if (h->n_streams != 1) return fail(GPSMI_E_STATE, "replay takes a handle with one stream");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    const int nch = h->max_ch;
    const size_t njobs = (size_t)nb * nch;
    for (size_t j = 0; j < njobs; ++j) {
        const gpsmi_trk_state& s = table[j];
        if (s.prn < 0 || s.prn > GPSMI_MAX_PRN || (s.prn && !h->have_rep[s.prn]))
            return fail(GPSMI_E_ARG, "replay table row %zu: bad PRN %d", j, s.prn);
        if (s.prn && (s.delay < 0 || s.delay >= h->cfg.code_samples || s.nps < 0 ||
                      s.nps > h->cfg.code_samples || s.df_len < 1 || s.df_len > h->P.df_no ||
                      s.edge_state < -1 || s.edge_state > 2))
            return fail(GPSMI_E_ARG, "replay table row %zu: state out of range", j);
        if (delay_used && delay_used[j] >= h->cfg.code_samples)
            return fail(GPSMI_E_ARG, "replay table row %zu: delay_used out of range", j);
    }
    // the epilogue of a run in flight reads d_tab_in and writes d_tab_out on its own stream: a new
    // table may only land once every outstanding run has finished (gpsmi.h states the rule)
    int rc = GPSMI_OK;
    if (h->slot[0].epi_pending || h->slot[1].epi_pending || h->slot[0].copy_pending ||
        h->slot[1].copy_pending) {
        rc = trk_settle(h);
        if (rc) return rc;
    }
    rc = trk_reserve(h, njobs);
    if (rc) return rc;
    h->main_tail = nullptr;
    GPSMI_HIP(hipMemcpyAsync(h->d_tab_in, table, njobs * sizeof(gpsmi_trk_state),
                             hipMemcpyHostToDevice, h->stream));
    if (delay_used)
        GPSMI_HIP(hipMemcpyAsync(h->d_forced, delay_used, njobs * sizeof(int),
                                 hipMemcpyHostToDevice, h->stream));
    GPSMI_HIP(hipStreamSynchronize(h->stream));
    h->replay_nb = nb;
    h->replay_forced = delay_used != nullptr;
    return GPSMI_OK;
}

int gpsmi_trk_replay_run_async(gpsmi_trk* h, const void* d_iq, int nb) {
    GPSMI_REQUIRE(h && d_iq, "null argument");
    GPSMI_QUIESCE(h);
    if (nb != h->replay_nb || nb < 1)
        return fail(GPSMI_E_STATE, "replay_run(nb=%d) without a matching replay_load (nb=%d)", nb,
                    h->replay_nb);
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    const int nch = h->max_ch;
    h->cur ^= 1;                           // the other slot may still be on its way to the host
    gpsmi_trk::Slot& sl = h->slot[h->cur];
    // "corr_overlap": consecutive runs are independent (two result slots), so they alternate between
    // two streams: the code-phase correlation of run k + 1 is then eligible while the correlator of
    // run k still runs, and the chip is never left to one kernel's tail.  Every kernel then shares
    // CUs with another one: throughput mode; the default keeps each kernel alone (its duration is
    // what the roofline is quoted on).
    if (h->corr_overlap && h->cur == 1 && !h->alt_stream) GPSMI_HIP(hipStreamCreate(&h->alt_stream));
    sl.run_stream = (h->corr_overlap && h->cur == 1) ? h->alt_stream : nullptr;
    hipStream_t rs = sl.run_stream ? sl.run_stream : h->stream;
    if (sl.copy_pending) {                 // its previous results must have left first
        // (that copy was queued behind the slot's epilogue -- replay_fetch_async -- so it stands for
        // both: one barrier packet between this run's first kernel and the previous run's last
        // instead of two, ~2 us of every step)
        GPSMI_HIP(hipStreamWaitEvent(rs, sl.copied, 0));
        sl.copy_pending = false;
        sl.epi_pending = false;
    }
    if (sl.epi_pending) {                  // ... or the epilogue that read its buffers be done
        GPSMI_HIP(hipStreamWaitEvent(rs, sl.epi_done, 0));
        sl.epi_pending = false;
    }
    // (mode 2 needs the batch form of the span correlator: its dispatch carries the stamps)
    sl.timing_pending = (h->timing == 2 && !(h->mfma == 4 && nb * ((nch + kSpCh - 1) / kSpCh) > h->span_single_max))
                            ? 0 : h->timing;
    return trk_launch(h, sl, d_iq, h->d_tab_in, h->d_tab_out,
                      h->replay_forced ? h->d_forced : nullptr, nb * nch, nch, /*side_epilogue=*/true);
}

int gpsmi_trk_wait(gpsmi_trk* h) {
    GPSMI_REQUIRE(h, "null handle");
    const auto t_q0 = std::chrono::steady_clock::now();
    if (h->worker) h->stat_backlog += h->worker->a_submitted.load() - h->worker->a_finished.load();
    GPSMI_QUIESCE(h);
    const auto t_q1 = std::chrono::steady_clock::now();
    h->stat_quiesce_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(t_q1 - t_q0).count();
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    // only streamed steps outstanding (in line on the main stream): the event behind the latest one
    // covers everything, and waiting for an event returns ~100 us sooner than the three stream
    // synchronisations of the general case
    const bool replay_busy = h->slot[0].copy_pending || h->slot[1].copy_pending || h->slot[0].timing_pending ||
                             h->slot[1].timing_pending || h->slot[0].epi_pending || h->slot[1].epi_pending;
    if (!replay_busy && !h->stage_used[0] && !h->stage_used[1] && (h->in_pending[0] || h->in_pending[1])) {
        const int latest = h->stage_idx ^ 1;               // (the slot of the step enqueued last)
        if (h->in_pending[latest]) {
            GPSMI_HIP(trk_spin_wait(h->in_done[latest]));
            h->in_pending[0] = h->in_pending[1] = false;
            h->stat_evwait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_q1).count();
            h->stat_waits += 1;
            return GPSMI_OK;
        }
    }
    return trk_settle(h);
}

int gpsmi_trk_wait_prev(gpsmi_trk* h) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    gpsmi_trk::Slot& sl = h->slot[h->cur ^ 1];
    if (sl.copy_pending) {
        GPSMI_HIP(hipEventSynchronize(sl.copied));
        sl.copy_pending = false;
        sl.epi_pending = false;            // (the copy was queued behind the slot's epilogue)
    } else if (sl.timing_pending) {
        GPSMI_HIP(hipEventSynchronize(sl.timing_pending == 1 ? sl.ev[3] : (sl.corr_stop ? sl.corr_stop : sl.ev[2])));
    }
    return trk_take_timing(h, sl);
}

int gpsmi_trk_replay_run(gpsmi_trk* h, const void* d_iq, int nb) {
    int rc = gpsmi_trk_replay_run_async(h, d_iq, nb);
    if (rc) return rc;
    return gpsmi_trk_wait(h);
}

int gpsmi_trk_replay_fetch_async(gpsmi_trk* h, gpsmi_trk_out* out, size_t n) {
    GPSMI_REQUIRE(h && out, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(n <= (size_t)h->replay_nb * h->max_ch, "more records than the last replay ran");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    gpsmi_trk::Slot& sl = h->slot[h->cur];
    // the copy runs on its own stream behind the run that produced the records, so the
    // next run (other slot) overlaps it
    if (sl.epi_pending) {                  // the records are complete when the slot's epilogue is
        GPSMI_HIP(hipStreamWaitEvent(h->copy_stream, sl.epi_done, 0));
    } else {
        GPSMI_HIP(hipEventRecord(sl.ready, sl.run_stream ? sl.run_stream : h->stream));
        GPSMI_HIP(hipStreamWaitEvent(h->copy_stream, sl.ready, 0));
    }
    GPSMI_HIP(hipMemcpyAsync(out, sl.d_out, n * sizeof(gpsmi_trk_out), hipMemcpyDeviceToHost,
                             h->copy_stream));
    GPSMI_HIP(hipEventRecord(sl.copied, h->copy_stream));
    sl.copy_pending = true;
    return GPSMI_OK;
}

int gpsmi_trk_replay_fetch(gpsmi_trk* h, gpsmi_trk_out* out, size_t n) {
    int rc = gpsmi_trk_replay_fetch_async(h, out, n);
    if (rc) return rc;
    return gpsmi_trk_wait(h);
}

int gpsmi_trk_replay(gpsmi_trk* h, const void* d_iq, int nb, const gpsmi_trk_state* table,
                     const int32_t* delay_used, gpsmi_trk_out* out) {
    GPSMI_REQUIRE(h && d_iq && table && out, "null argument");
    GPSMI_REQUIRE(nb >= 0, "negative block count");
    if (nb == 0) return GPSMI_OK;
    int rc = gpsmi_trk_replay_load(h, nb, table, delay_used);
    if (rc) return rc;
    rc = gpsmi_trk_replay_run(h, d_iq, nb);
    if (rc) return rc;
    return gpsmi_trk_replay_fetch(h, out, (size_t)nb * h->max_ch);
}

int gpsmi_trk_replay_states(gpsmi_trk* h, gpsmi_trk_state* states, size_t n) {
    GPSMI_REQUIRE(h && states, "null argument");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(n <= (size_t)h->replay_nb * h->max_ch,
                  "more states requested than the last replay produced");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    GPSMI_HIP(hipMemcpy(states, h->d_tab_out, n * sizeof(gpsmi_trk_state), hipMemcpyDeviceToHost));
    return GPSMI_OK;
}

int gpsmi_trk_last_ms(gpsmi_trk* h, float* total_ms, float* correlator_ms) {
    GPSMI_REQUIRE(h, "null handle");
    if (total_ms) *total_ms = h->last_total_ms;
    if (correlator_ms) *correlator_ms = h->last_corr_ms;
    return GPSMI_OK;
}

int gpsmi_trk_last_codephase_ms(gpsmi_trk* h, float* ms) {
    GPSMI_REQUIRE(h && ms, "null argument");
    *ms = h->last_cp_ms;
    return GPSMI_OK;
}

int gpsmi_trk_after_acq(gpsmi_trk* later, gpsmi_acq* earlier) {
    GPSMI_REQUIRE(later && earlier, "null handle");
    GPSMI_QUIESCE(later);
    const HandleSync e = acq_sync(earlier);
    GPSMI_REQUIRE(e.device == later->cfg.device, "handles on different devices");
    GPSMI_HIP(hipSetDevice(e.device));
    GPSMI_HIP(hipEventRecord(e.order, e.stream));
    GPSMI_HIP(hipStreamWaitEvent(later->stream, e.order, 0));
    return GPSMI_OK;
}

int gpsmi_trk_set_input_format(gpsmi_trk* h, int fmt) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(fmt == GPSMI_IQ_C64 || fmt == GPSMI_IQ_U8, "unknown input format");
    if (fmt == GPSMI_IQ_U8 && h->mfma != 4)
        return fail(GPSMI_E_UNSUPPORTED, "raw u8 IQ input needs CODE_SAMPLES = 2048 and the span "
                                         "correlator (option \"correlator\" = 1)");
    h->iq_fmt = fmt;
    return GPSMI_OK;
}

int gpsmi_trk_set_streams(gpsmi_trk* h, int n_streams) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    GPSMI_REQUIRE(n_streams >= 1 && (long long)n_streams * h->max_ch <= 65536,
                  "n_streams out of range (streams x channels <= 65536)");
    GPSMI_HIP(hipSetDevice(h->cfg.device));
    int rc = trk_settle(h);
    if (rc) return rc;
    const size_t rows = (size_t)n_streams * h->max_ch;
    const size_t ngps = (size_t)h->cfg.n_cyc * h->cfg.code_samples;
    // new buffers first, then the swap: a failed allocation leaves the handle as it was
    gpsmi_trk_state* d_state = nullptr;
    float2* d_block = nullptr;
    if (hipMalloc((void**)&d_state, rows * sizeof(gpsmi_trk_state)) != hipSuccess ||
        hipMalloc((void**)&d_block, (size_t)n_streams * ngps * sizeof(float2)) != hipSuccess) {
        if (d_state) (void)hipFree(d_state);
        (void)hipGetLastError();
        return fail(GPSMI_E_NOMEM, "out of device memory for %d streams", n_streams);
    }
    rc = hipMemset(d_state, 0, rows * sizeof(gpsmi_trk_state)) == hipSuccess ? trk_reserve(h, rows)
                                                                              : fail(GPSMI_E_HIP, "hipMemset of the state rows failed");
    if (rc) {
        (void)hipFree(d_state);
        (void)hipFree(d_block);
        return rc;
    }
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_block) (void)hipFree(h->d_block);
    h->d_state = d_state;
    h->d_block = d_block;
    h->h_state.assign(rows, gpsmi_trk_state{});
    h->state_dirty_host = false;
    h->n_streams = n_streams;
    h->replay_nb = 0;
    return GPSMI_OK;
}

int gpsmi_trk_set_option(gpsmi_trk* h, const char* key, long long value) {
    GPSMI_REQUIRE(h && key, "null argument");
    GPSMI_QUIESCE(h);
    const auto bad = [&]() { return fail(GPSMI_E_ARG, "gpsmi_trk_set_option: %s = %lld out of range", key, value); };
    if (!strcmp(key, "corr_cg")) {
        if (value != 2 && value != 4 && value != 6) return bad();
        h->corr_cg = (int)value;
    } else if (!strcmp(key, "corr_small1")) {
        if (value < 0 || value > h->corr_small2) return bad();
        h->corr_small1 = (int)value;
    } else if (!strcmp(key, "corr_small2")) {
        if (value < h->corr_small1 || value > (1 << 24)) return bad();
        h->corr_small2 = (int)value;
    } else if (!strcmp(key, "span_single_max")) {
        if (value < 1 || value > kSpanUnitsMax) return bad();
        h->span_single_max = (int)value;
    } else if (!strcmp(key, "stream_inline_max")) {
        if (value < 0) return bad();
        h->stream_inline_max = (size_t)value;
    } else if (!strcmp(key, "stream_direct_max")) {
        if (value < 0) return bad();
        h->stream_direct_max = (size_t)value;
    } else if (!strcmp(key, "done_by_dispatch")) {
        h->done_by_dispatch = value != 0;
    } else if (!strcmp(key, "corr_overlap")) {
        h->corr_overlap = value != 0;
    } else if (!strcmp(key, "fold_chunk")) {
        if (value < 0 || value > (1 << 20)) return bad();
        h->fold_chunk = (int)value;
    } else if (!strcmp(key, "epilogue_form")) {
        if (value != 0 && value != 1) return bad();
        h->epilogue_form = (int)value;
    } else if (!strcmp(key, "stream_thread")) {
        h->stream_thread = value != 0;
    } else if (!strcmp(key, "stream_depth")) {
        if (value < 2 || value > 64) return bad();
        h->stream_depth = (int)value;
    } else if (!strcmp(key, "correlator") || !strcmp(key, "codephase") || !strcmp(key, "debug_flags") ||
               !strcmp(key, "copy_stream")) {
        return fail(GPSMI_E_STATE, "gpsmi_trk_set_option: '%s' is taken at create time (gpsmi_set_default)", key);
    } else {
        return fail(GPSMI_E_ARG, "gpsmi_trk_set_option: unknown option '%s'", key);
    }
    return GPSMI_OK;
}

int gpsmi_trk_get_option(gpsmi_trk* h, const char* key, long long* value) {
    GPSMI_REQUIRE(h && key && value, "null argument");
    GPSMI_QUIESCE(h);
    if (!strcmp(key, "corr_cg")) *value = h->corr_cg;
    else if (!strcmp(key, "corr_small1")) *value = h->corr_small1;
    else if (!strcmp(key, "corr_small2")) *value = h->corr_small2;
    else if (!strcmp(key, "span_single_max")) *value = h->span_single_max;
    else if (!strcmp(key, "stream_inline_max")) *value = (long long)h->stream_inline_max;
    else if (!strcmp(key, "stream_direct_max")) *value = (long long)h->stream_direct_max;
    else if (!strcmp(key, "done_by_dispatch")) *value = h->done_by_dispatch;
    else if (!strcmp(key, "corr_overlap")) *value = h->corr_overlap;
    else if (!strcmp(key, "fold_chunk")) *value = h->fold_chunk;
    else if (!strcmp(key, "epilogue_form")) *value = h->epilogue_form;
    else if (!strcmp(key, "stream_thread")) *value = h->stream_thread;
    else if (!strcmp(key, "stream_depth")) *value = h->stream_depth;
    else if (!strcmp(key, "stat_stream_steps")) *value = h->stat_steps;
    else if (!strcmp(key, "stat_waits")) *value = h->stat_waits;
    else if (!strcmp(key, "stat_backlog")) *value = h->stat_backlog;

    else if (!strcmp(key, "stat_quiesce_ns")) *value = h->stat_quiesce_ns;
    else if (!strcmp(key, "stat_evwait_ns")) *value = h->stat_evwait_ns;
    else if (!strcmp(key, "stat_stream_wait_ns")) *value = h->stat_wait_ns;
    else if (!strcmp(key, "stat_stream_launch_ns")) *value = h->stat_launch_ns;
    else if (!strcmp(key, "correlator")) *value = (h->mfma == 4 || h->span8) ? 1 : 0;     // what runs, not what was asked
    else if (!strcmp(key, "codephase")) *value = h->general ? (h->pfa ? 0 : (h->big ? 2 : 1)) : 0;
    else if (!strcmp(key, "debug_flags")) *value = h->P.flags;
    else if (!strcmp(key, "copy_stream")) *value = h->own_copy_stream ? 1 : 0;
    else return fail(GPSMI_E_ARG, "gpsmi_trk_get_option: unknown option '%s'", key);
    return GPSMI_OK;
}

int gpsmi_trk_corr_grid(int nblocks, int ngroups) {
    if (nblocks < 1 || ngroups < 1) return fail(GPSMI_E_ARG, "gpsmi_trk_corr_grid: counts must be >= 1");
    return corr_grid(nblocks, ngroups);
}

int gpsmi_trk_corr_wg_map(int nblocks, int ngroups, int wg, int* block, int* group) {
    GPSMI_REQUIRE(block && group, "null argument");
    GPSMI_REQUIRE(nblocks >= 1 && ngroups >= 1 && wg >= 0 && wg < corr_grid(nblocks, ngroups), "out of range");
    const CorrWg u = corr_wg_map(wg, nblocks, ngroups);
    *block = u.block;
    *group = u.group;
    return GPSMI_OK;
}

int gpsmi_trk_set_timing(gpsmi_trk* h, int on) {
    GPSMI_REQUIRE(h, "null handle");
    GPSMI_QUIESCE(h);
    h->timing = on == 2 ? 2 : (on != 0);
    return GPSMI_OK;
}

}  // extern "C"
