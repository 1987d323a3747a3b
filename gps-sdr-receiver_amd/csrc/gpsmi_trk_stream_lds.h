// The tracking correlator, LDS-ring form: one workgroup per block computes the
// prompt correlate-and-dump of up to 12 channels while every IQ row is brought
// from HBM exactly once per CU by LDS-DMA.
//
// Same mathematics and the same per-wave inner loop as gpsmi_trk_stream.h (read
// that header first: carrier separation, B in registers, hi/lo windows, the
// mixed wave).  What changes is how rows reach the lanes:
//
//  * 256*G threads: G groups of four waves; group gi owns channels
//    [6 gi, 6 gi + 6), wave q of a group owns positions [512 q, 512 q + 512).
//  * Rows (16 KiB) are staged in a ring of kRing row slots in LDS by
//    global_load_lds_dwordx4 (1 KiB per wave-instruction, no VGPR staging);
//    kRing - 1 rows are in flight per CU.  Per row: every wave waits (counted
//    vmcnt) until its own pieces of rows r and r+1 have landed, one raw
//    s_barrier makes that true for all pieces, then the DMA for row r + kRing - 1
//    is issued into the slot row r - 1 has just left and the wave reads its
//    quarter of row r with four ds_read_b128.  The transfer of later rows
//    proceeds while the waves compute: no register dependency, no vmcnt(0).
//  * Both groups read the same LDS rows, so L2 -> CU traffic is the algorithmic
//    8 bytes per sample (the register-staged kernel reads every row once per
//    group).
//  * Lane sums after each 4-row pass: three DPP butterfly steps (quad swaps,
//    half-row mirror) bring the 64 lanes down to 8 partial sums, which cross
//    through a 1.5 KiB wave-private LDS tile; fixed order, deterministic.
//
// All LDS is one array (a second __shared__ object next to an LDS-DMA target
// makes hipcc drain vmcnt before every ds_read).
#pragma once
#include "gpsmi_trk_stream.h"

namespace gpsmi {

constexpr int kRing = 8;                 // row slots in LDS
constexpr int kDepth = kRing - 1;        // rows in flight

// Correction of the one mixed wave of a channel for one row (this variant keeps the
// x[r+1]-x[r] form): its lo elements must see x[r+1] instead of x[r]; the plain pass
// already added B*x[r], so add B*(x[r+1]-x[r]) for them.
template <int IS>
__device__ __forceinline__ void mixed_fix(v2f& acc, const v2f* B, const v2f* df,
                                          unsigned long long m0, unsigned long long m1) {
    acc += lo_sum<IS>(B, df, m0, m1);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef float v4f __attribute__((ext_vector_type(4)));

// LDS accesses inside the row loop go through inline asm: while LDS-DMA is in
// flight hipcc puts s_waitcnt vmcnt(0) in front of every ds_read it can see
// (it cannot tell the ring slots apart), which would drain the ring each row.
// The asm carries its own lgkmcnt wait; ordering against the DMA is done by the
// counted vmcnt + s_barrier in the loop.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(lds_ptr_t)p;
}
__device__ __forceinline__ void lds_read_quarter(v2f* dst, unsigned addr) {
    v4f a, b, c, d;
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(addr)
        : "memory");
    dst[0] = v2f{a.x, a.y}; dst[1] = v2f{a.z, a.w};
    dst[2] = v2f{b.x, b.y}; dst[3] = v2f{b.z, b.w};
    dst[4] = v2f{c.x, c.y}; dst[5] = v2f{c.z, c.w};
    dst[6] = v2f{d.x, d.y}; dst[7] = v2f{d.z, d.w};
}
// The same without the wait: the data are not valid before lds_wait_quarter() on
// the same registers (which also keeps the compiler from touching them earlier).
__device__ __forceinline__ void lds_read_quarter_async(v4f* q, unsigned addr) {
    asm volatile(
        "ds_read_b128 %0, %4\n\t"
        "ds_read_b128 %1, %4 offset:1024\n\t"
        "ds_read_b128 %2, %4 offset:2048\n\t"
        "ds_read_b128 %3, %4 offset:3072"
        : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
        : "v"(addr)
        : "memory");
}
__device__ __forceinline__ void lds_wait_quarter(v4f* q) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3])::"memory");
}
__device__ __forceinline__ void unpack_quarter(v2f* dst, const v4f* q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[2 * i] = v2f{q[i].x, q[i].y};
        dst[2 * i + 1] = v2f{q[i].z, q[i].w};
    }
}
// six dword stores at addr + 4*(k0 .. k0+5)
template <int K0>
__device__ __forceinline__ void lds_store6(unsigned addr, float a, float b, float c, float d,
                                           float e, float f) {
    asm volatile(
        "ds_write_b32 %0, %1 offset:%7\n\t"
        "ds_write_b32 %0, %2 offset:%8\n\t"
        "ds_write_b32 %0, %3 offset:%9\n\t"
        "ds_write_b32 %0, %4 offset:%10\n\t"
        "ds_write_b32 %0, %5 offset:%11\n\t"
        "ds_write_b32 %0, %6 offset:%12"
        :
        : "v"(addr), "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "n"(4 * K0),
          "n"(4 * K0 + 4), "n"(4 * K0 + 8), "n"(4 * K0 + 12), "n"(4 * K0 + 16), "n"(4 * K0 + 20)
        : "memory");
}
// sum of eight dwords at addr + 196*l (one per 8-lane group of the wave)
__device__ __forceinline__ float lds_sum8(unsigned addr) {
    float a, b, c, d, e, f, g, h;
    asm volatile(
        "s_waitcnt lgkmcnt(0)\n\t"
        "ds_read_b32 %0, %8\n\t"
        "ds_read_b32 %1, %8 offset:196\n\t"
        "ds_read_b32 %2, %8 offset:392\n\t"
        "ds_read_b32 %3, %8 offset:588\n\t"
        "ds_read_b32 %4, %8 offset:784\n\t"
        "ds_read_b32 %5, %8 offset:980\n\t"
        "ds_read_b32 %6, %8 offset:1176\n\t"
        "ds_read_b32 %7, %8 offset:1372\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&v"(e), "=&v"(f), "=&v"(g), "=&v"(h)
        : "v"(addr)
        : "memory");
    return ((a + b) + (c + d)) + ((e + f) + (g + h));
}
__device__ __forceinline__ void lds_store1(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int NC, int G>
__global__ __launch_bounds__(256 * G) void trk_stream_lds_kernel(
    const float2* __restrict__ iq, const gpsmi_trk_state* __restrict__ st_in,
    const JobMid* __restrict__ mid, const float* __restrict__ code, TrkParams P, int nsuper,
    int nblocks, float2* __restrict__ partial) {
    static_assert(NC % 4 == 0 && NC >= kRing, "row count");
    constexpr int NW = 4 * G;                        // waves
    constexpr int PP = 16 / NW;                      // DMA pieces per wave and row
    constexpr int kRowBytes = kFftN * 8;
    constexpr int offTr = kRing * kRowBytes;         // float tr[NW][8][49]
    constexpr int offSw = offTr + NW * 8 * 49 * 4;   // float2 sw[NW][6][NC]
    constexpr int offHd = offSw + NW * kGroupCh * NC * 8;        // float2 hd[NW][6]
    constexpr int offCls = offHd + NW * kGroupCh * 8;            // int cls[NW][6]
    constexpr int offChan = offCls + NW * kGroupCh * 4;          // StreamChan schan[6 G]
    constexpr int offRot = offChan + kGroupCh * G * (int)sizeof(StreamChan);  // float2 rot[6G][9]
    constexpr int kBytes = offRot + kGroupCh * G * (kJ + 1) * 8;
    static_assert(kBytes <= 160 * 1024, "LDS budget");
    __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes];
    float(*tr)[8][49] = reinterpret_cast<float(*)[8][49]>(smem + offTr);
    float2(*sw)[kGroupCh][NC] = reinterpret_cast<float2(*)[kGroupCh][NC]>(smem + offSw);
    float2(*hd)[kGroupCh] = reinterpret_cast<float2(*)[kGroupCh]>(smem + offHd);
    int(*cls)[kGroupCh] = reinterpret_cast<int(*)[kGroupCh]>(smem + offCls);
    StreamChan* schan = reinterpret_cast<StreamChan*>(smem + offChan);
    float2(*rot)[kJ + 1] = reinterpret_cast<float2(*)[kJ + 1]>(smem + offRot);

    const int b = blockIdx.x / nsuper, sg = blockIdx.x % nsuper;
    if (b >= nblocks) return;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int gi = wave >> 2, q = wave & 3;
    const int cs = kFftN;
    const float2* blk = iq + (size_t)b * ((size_t)cs * NC);
    const float inv_fs = 1.0f / (1000.0f * (float)cs);
    const double inv_2pi = 0.15915494309189533576888376337251;
    const int cbase = sg * kGroupCh * G;             // first channel of this workgroup

    // ---- LDS-DMA: piece p (1 KiB) of row r -> slot r % kRing
    const char* gsrc = reinterpret_cast<const char*>(blk) + (size_t)(wave * PP) * 1024 + lane * 16;
    // (A rotated visiting order, to de-phase CUs that start together on blocks a
    // power of two apart, was measured: no effect.  Rows are visited in order.)
    constexpr int off = 0;
    auto issue_row = [&](int n) {                    // n = position in the visiting order
        const int r = (n + off) % NC;
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            const char* g = gsrc + (size_t)r * kRowBytes + pp * 1024;
            unsigned char* l = smem + (n % kRing) * kRowBytes + (wave * PP + pp) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
        }
    };
    // ---- per-channel set-up (as in gpsmi_trk_stream.h); its ordinary global
    // loads come first and complete, then the DMA prologue is issued so that
    // the phasor arithmetic below overlaps the first rows' flight
    if (t < kGroupCh * G * (kJ + 1)) {
        const int c = t / (kJ + 1), k = t % (kJ + 1);
        const int cidx = cbase + c;
        const JobMid md = mid[b * P.nch + (cidx < P.nch ? cidx : P.nch - 1)];
        float2 r = make_float2(1.f, 0.f);
        if (cidx < P.nch && md.active) {
            const int off = (k == kJ) ? cs : 128 * (k >> 1) + (k & 1);
            const double rev = (double)md.om * inv_2pi * (double)off / (1000.0 * (double)cs);
            r = phasor_rev((float)(rev - rint(rev)));
        }
        rot[c][k] = r;
    }
    // plain barrier: the DMA in flight does not care, and nothing below waits on it
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    v2f B[kGroupCh][kJ];
    int kcls[kGroupCh], istar[kGroupCh];
    unsigned long long lm0[kGroupCh], lm1[kGroupCh];
    const int w0 = 512 * q;
    const int mbase = w0 + 2 * lane;                 // m = mbase + 128 i + e
    // all descriptor loads first (independent), then all replica gathers
    StreamChan chan[kGroupCh];
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) {
        const int cidx = cbase + gi * kGroupCh + c;
        const int job = b * P.nch + (cidx < P.nch ? cidx : P.nch - 1);
        const JobMid md = mid[job];
        StreamChan s;
        s.job = b * P.nch + cidx;
        s.active = (cidx < P.nch) && md.active;
        s.om = md.om; s.ph = md.ph; s.d = md.delay_used; s.prn = md.prn;
        chan[c] = s;
    }
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) {
        const StreamChan s = chan[c];
        const float* cv = code + (size_t)s.prn * cs;
#pragma unroll
        for (int j = 0; j < kJ; ++j) {               // replica samples, parked in B.x
            const int m = mbase + 128 * (j >> 1) + (j & 1);
            B[c][j] = v2f{cv[(m - s.d) & (cs - 1)], 0.f};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < kDepth; ++r) issue_row(r);
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) {
        const int cl = gi * kGroupCh + c;
        const StreamChan s = chan[c];
        if (q == 0 && lane == 0) schan[cl] = s;
        const float f_eff = (float)((double)s.om * inv_2pi);
        const float rev0 = fmaf(f_eff, (float)(mbase + 1) * inv_fs, s.ph * (float)inv_2pi);
        float2 z0 = phasor_rev(rev0);
        const float2 rT = rot[cl][kJ];
        int k = (s.d <= w0) ? 0 : (s.d >= w0 + 512 ? 1 : 2);
        if (!s.active) k = 0;
        k = __builtin_amdgcn_readfirstlane(k);
        kcls[c] = k;
        if (lane == 0) cls[wave][c] = k;
        if (k == 1) z0 = cmulf(z0, rT);
#pragma unroll
        for (int j = 0; j < kJ; ++j) {
            const int m = mbase + 128 * (j >> 1) + (j & 1);
            float2 z = (j == 0) ? z0 : cmulf(z0, rot[cl][j]);
            if (k == 2) {
                const float2 zl = cmulf(z, rT);
                const bool lo = m < s.d;
                z = make_float2(lo ? zl.x : z.x, lo ? zl.y : z.y);
            }
            const float v = s.active ? B[c][j].x : 0.f;
            B[c][j] = v2f{v * z.x, v * z.y};
        }
        int is = 0;
        unsigned long long b0 = 0, b1 = 0;
        if (k == 2) {
            is = (s.d - w0 - 1) >> 7;
            b0 = __ballot(w0 + 128 * is + 2 * lane < s.d);
            b1 = __ballot(w0 + 128 * is + 2 * lane + 1 < s.d);
        }
        istar[c] = __builtin_amdgcn_readfirstlane(is);
        lm0[c] = b0;
        lm1[c] = b1;
    }
    int anymixed = 0;
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) anymixed |= (kcls[c] == 2);

    // ---- rows
    const unsigned myq = lds_addr(smem) + q * 4096 + lane * 16;   // this lane's 16 B of each KiB
    auto read_row = [&](v2f* dst, int n, bool valid) {   // row at visiting position n
        if (valid) {
            lds_read_quarter(dst, myq + (n % kRing) * kRowBytes);
        } else {
#pragma unroll
            for (int j = 0; j < kJ; ++j) dst[j] = v2f{0.f, 0.f};
        }
    };
    const unsigned tr_w = lds_addr(&tr[wave][lane >> 3][0]);      // row of this 8-lane group
    const unsigned tr_r = lds_addr(&tr[wave][0][lane < kTrVals ? lane : 0]);
    const unsigned sw_w = lds_addr(&sw[wave][(lane < kTrVals ? lane : 0) / 8][0]) +
                          4 * (lane & 1);

    v2f headv[kGroupCh];
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) headv[c] = v2f{0.f, 0.f};
    v4f xq[2][4];                                    // row r / row r+1, roles alternate
#pragma unroll 1
    for (int pass = 0; pass < NC / 4; ++pass) {
        v2f acc[kGroupCh][4];
#pragma unroll
        for (int c = 0; c < kGroupCh; ++c)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[c][i] = v2f{0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = pass * 4 + rr;                        // visiting position
            const int row = (r + off) % NC;                     // the row it holds
            // positions r and r+1 landed (own pieces), then everybody's
            {
                const int last_issued = (r + kDepth - 1 < NC - 1) ? r + kDepth - 1 : NC - 1;
                const int need = (r + 1 < NC - 1) ? r + 1 : NC - 1;
                switch (last_issued - need) {            // rows that may stay in flight
                    case 0: wait_vmcnt<0>(); break;
                    case 1: wait_vmcnt<1 * PP>(); break;
                    case 2: wait_vmcnt<2 * PP>(); break;
                    case 3: wait_vmcnt<3 * PP>(); break;
                    case 4: wait_vmcnt<4 * PP>(); break;
                    default: wait_vmcnt<5 * PP>(); break;
                }
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (r + kDepth < NC) issue_row(r + kDepth);         // into the slot of row r-1

            // row r is already in registers (read during row r-1); start reading
            // row r+1 now, it is waited for after this row's arithmetic
            v4f* qc = xq[rr & 1];
            v4f* qn = xq[(rr + 1) & 1];
            if (r == 0) {
                lds_read_quarter_async(qc, myq);
                lds_wait_quarter(qc);
            }
            const bool has_next = r + 1 < NC;
            if (has_next) lds_read_quarter_async(qn, myq + ((r + 1) % kRing) * kRowBytes);
            v2f xc[kJ];
            unpack_quarter(xc, qc);
            if (row == 0) {
                // head: lo part of row 0 belongs to window -1 (mixed wave only; pure
                // lo waves get it by relabelling their row 0)
#pragma unroll
                for (int c = 0; c < kGroupCh; ++c) {
                    if (kcls[c] == 2) {
                        switch (istar[c]) {
                            case 0: mixed_fix<0>(headv[c], B[c], xc, lm0[c], lm1[c]); break;
                            case 1: mixed_fix<1>(headv[c], B[c], xc, lm0[c], lm1[c]); break;
                            case 2: mixed_fix<2>(headv[c], B[c], xc, lm0[c], lm1[c]); break;
                            default: mixed_fix<3>(headv[c], B[c], xc, lm0[c], lm1[c]); break;
                        }
                    }
                }
            }
            if (!(P.flags & 1)) {
#pragma unroll
                for (int j = 0; j < kJ; ++j)
#pragma unroll
                    for (int c = 0; c < kGroupCh; c += 2)
                        cmac2(acc[c][rr], acc[c + 1][rr], B[c][j], xc[j], B[c + 1][j], xc[j]);
            } else {
#pragma unroll
                for (int j = 0; j < kJ; ++j) asm volatile("" ::"v"(xc[j]));
                acc[0][rr] += xc[0];
            }
            if (has_next) lds_wait_quarter(qn);
            if (anymixed && !(P.flags & 4)) {
                v2f df[kJ];                          // the row above; nothing above the last
                if (has_next) {
                    unpack_quarter(df, qn);
                } else {
#pragma unroll
                    for (int j = 0; j < kJ; ++j) df[j] = v2f{0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < kJ; ++j) df[j] -= xc[j];
#pragma unroll
                for (int c = 0; c < kGroupCh; ++c) {
                    if (kcls[c] == 2) {
                        switch (istar[c]) {
                            case 0: mixed_fix<0>(acc[c][rr], B[c], df, lm0[c], lm1[c]); break;
                            case 1: mixed_fix<1>(acc[c][rr], B[c], df, lm0[c], lm1[c]); break;
                            case 2: mixed_fix<2>(acc[c][rr], B[c], df, lm0[c], lm1[c]); break;
                            default: mixed_fix<3>(acc[c][rr], B[c], df, lm0[c], lm1[c]); break;
                        }
                    }
                }
            }
        }
        // ---- lane sums of the pass: 64 -> 8 by DPP, 8 -> 1 through LDS
        if (P.flags & 2) {
            if (lane < kTrVals)
                lds_store1(sw_w + 8 * ((pass * 4 + off) % NC + (lane % 8) / 2), acc[0][0].x);
            continue;
        }
        float part[kTrVals];
#pragma unroll
        for (int v = 0; v < kTrVals; ++v) {
            const int c = v / 8, i = (v % 8) / 2;
            float sm = (v & 1) ? acc[c][i].y : acc[c][i].x;
            sm += dpp_get<0xB1>(sm);                            // lanes xor 1
            sm += dpp_get<0x4E>(sm);                            // lanes xor 2
            sm += dpp_get<0x141>(sm);                           // half-row mirror: 8 lanes
            part[v] = sm;
        }
        if ((lane & 7) == 0) {
            lds_store6<0>(tr_w, part[0], part[1], part[2], part[3], part[4], part[5]);
            lds_store6<6>(tr_w, part[6], part[7], part[8], part[9], part[10], part[11]);
            lds_store6<12>(tr_w, part[12], part[13], part[14], part[15], part[16], part[17]);
            lds_store6<18>(tr_w, part[18], part[19], part[20], part[21], part[22], part[23]);
            lds_store6<24>(tr_w, part[24], part[25], part[26], part[27], part[28], part[29]);
            lds_store6<30>(tr_w, part[30], part[31], part[32], part[33], part[34], part[35]);
            lds_store6<36>(tr_w, part[36], part[37], part[38], part[39], part[40], part[41]);
            lds_store6<42>(tr_w, part[42], part[43], part[44], part[45], part[46], part[47]);
        }
        if (lane < kTrVals) {
            const float sm = lds_sum8(tr_r);                    // LDS executes a wave in order
            const int i = (lane % 8) / 2;
            lds_store1(sw_w + 8 * ((pass * 4 + off) % NC + i), sm);
        }
    }
    // head sums of the mixed waves
#pragma unroll
    for (int c = 0; c < kGroupCh; ++c) {
        v2f h = headv[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            h.x += __shfl_down(h.x, o, 64);
            h.y += __shfl_down(h.y, o, 64);
        }
        if (lane == 0) hd[wave][c] = make_float2(h.x, h.y);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    // ---- combine the four waves of each group, apply U, write the partial sums
    for (int item = t; item < kGroupCh * G * (NC + 1); item += 256 * G) {
        const int cl = item / (NC + 1), o = item % (NC + 1), qq = o - 1;
        const StreamChan s = schan[cl];
        if (!s.active) continue;
        const int g2 = cl / kGroupCh, c = cl % kGroupCh;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int wv = g2 * 4 + w;
            const int k = cls[wv][c];
            const int r = (k == 1) ? qq + 1 : qq;
            if (r >= 0 && r < NC) { sx += sw[wv][c][r].x; sy += sw[wv][c][r].y; }
            if (qq == -1 && k == 2) { sx += hd[wv][c].x; sy += hd[wv][c].y; }
        }
        const double rev = (double)s.om * inv_2pi * (double)qq * 1.0e-3;
        const float2 u = phasor_rev((float)(rev - rint(rev)));
        partial[(size_t)s.job * (NC + 1) + o] = make_float2(sx * u.x - sy * u.y, sy * u.x + sx * u.y);
    }
}

}  // namespace gpsmi
