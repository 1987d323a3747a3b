// Shared host-side helpers of libgpsmi: error reporting across the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/gpsmi.h"

namespace gpsmi {

// Text of the last failure on this thread (returned by gpsmi_last_error()).
char* last_error_buf();
int fail(int code, const char* fmt, ...);

#define GPSMI_HIP(call)                                                          \
    do {                                                                         \
        hipError_t e__ = (call);                                                 \
        if (e__ != hipSuccess)                                                   \
            return ::gpsmi::fail(GPSMI_E_HIP, "%s: %s (%s:%d)", #call,           \
                                 hipGetErrorString(e__), __FILE__, __LINE__);    \
    } while (0)

#define GPSMI_REQUIRE(cond, msg)                                                 \
    do {                                                                         \
        if (!(cond)) return ::gpsmi::fail(GPSMI_E_ARG, "%s: %s", __func__, msg); \
    } while (0)

// Process-wide defaults of the options a handle takes at create time (gpsmi_set_default in gpsmi.h):
// the value set through the ABI, else the environment variable of the same meaning, else `fallback`.
// Keys: see the table in gpsmi_core.hip.  -> false for an unknown key.
bool default_opt(const char* key, long long* value, long long fallback);

// Is [p, p + bytes) inside a block handed out by gpsmi_host_alloc (and not freed since)?  -> the
// address a kernel reaches it at.
bool host_alloc_lookup(const void* p, size_t bytes, void** dev);

// exp(-2 pi i k / 2048) computed in double, rounded once.
void make_twiddles(std::vector<float2>& tw);

// stream of a handle and an event of its own to order another handle's stream behind it
// (defined next to the handle structs; used by gpsmi_trk_after_acq / gpsmi_acq_after_trk)
// `tail`: an event already recorded behind the last work on `stream` (null: record `order`)
struct HandleSync { hipStream_t stream; hipEvent_t order; int device; hipEvent_t tail; };

inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Single float32 operations that are never fused into a multiply-add, for code that
// restates the reference's numpy float32 arithmetic step by step.  (HIP's
// __fmul_rn/__fadd_rn are plain operators compiled with contraction allowed: hipcc
// fuses __fadd_rn(a, __fmul_rn(b, c)) into one v_fma.)
#pragma clang fp contract(off)
__device__ __host__ inline float mul_rn(float a, float b) { return a * b; }
__device__ __host__ inline float add_rn(float a, float b) { return a + b; }
__device__ __host__ inline float sub_rn(float a, float b) { return a - b; }
// streamData's sample decode (reference src/gpsrecv.py:168-173) of one raw uint16 (Q << 8 | I):
// float32(byte) * float32(1 / 127.5) - 1 per component, two roundings, exactly what numpy's
// portable complex64 loop and gpsmi_dev_unpack_u8iq compute
__device__ inline float2 decode_u8iq(unsigned v) {
    const float scl = 1.0f / 127.5f;
    return make_float2(sub_rn(mul_rn((float)(v & 0xFF), scl), 1.0f), sub_rn(mul_rn((float)(v >> 8), scl), 1.0f));
}
// sample k of a block in either input format (FMT 0: complex64, FMT 1: raw uint16)
template <int FMT>
__device__ __forceinline__ float2 load_iq(const void* iq, size_t k) {
    if constexpr (FMT == 0) return static_cast<const float2*>(iq)[k];
    else return decode_u8iq(static_cast<const uint16_t*>(iq)[k]);
}
#pragma clang fp contract(fast)

}  // namespace gpsmi
