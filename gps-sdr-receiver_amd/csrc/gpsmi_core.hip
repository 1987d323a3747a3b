// libgpsmi core: error text, device buffers, the u8-IQ unpack kernel.
#include <cmath>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gpsmi_common.h"

// the unpack arithmetic mirrors numpy step by step: no fused multiply-add
// (mul_rn/sub_rn of gpsmi_common.h are never contracted)
#pragma clang fp contract(off)

namespace gpsmi {

char* last_error_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// The options of gpsmi_set_default / gpsmi_trk_set_option (include/gpsmi.h) and the environment
// variable that sets each one's default when the ABI has not.
struct OptDef { const char* key; const char* env; bool set; long long value; };
static OptDef g_opts[] = {
    {"correlator", "GPSMI_STREAM_MFMA", false, 0},
    {"codephase", "GPSMI_DIRECT_CORR", false, 0},
    {"corr_cg", "GPSMI_CORR_CG", false, 0},
    {"corr_small1", nullptr, false, 0},
    {"corr_small2", nullptr, false, 0},
    {"span_single_max", "GPSMI_SPAN_SINGLE_MAX", false, 0},
    {"stream_inline_max", "GPSMI_STREAM_INLINE_MAX", false, 0},
    {"stream_direct_max", "GPSMI_STREAM_DIRECT_MAX", false, 0},
    {"done_by_dispatch", "GPSMI_DONE_BY_DISPATCH", false, 0},
    {"corr_overlap", "GPSMI_CORR_OVERLAP", false, 0},
    {"stream_thread", "GPSMI_STREAM_THREAD", false, 0},
    {"stream_depth", "GPSMI_STREAM_DEPTH", false, 0},
    {"fold_chunk", "GPSMI_FOLD_CHUNK", false, 0},
    {"epilogue_form", "GPSMI_EPILOGUE_FORM", false, 0},
    {"copy_stream", "GPSMI_COPY_STREAM", false, 0},
    {"debug_flags", "GPSMI_DEBUG_FLAGS", false, 0},
};
static std::mutex g_opts_mutex;

static OptDef* find_opt(const char* key) {
    if (!key) return nullptr;
    for (auto& o : g_opts)
        if (strcmp(o.key, key) == 0) return &o;
    return nullptr;
}

bool default_opt(const char* key, long long* value, long long fallback) {
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    OptDef* o = find_opt(key);
    if (!o) return false;
    *value = fallback;
    if (o->set) {
        *value = o->value;
    } else if (strcmp(key, "corr_small1") == 0 || strcmp(key, "corr_small2") == 0) {
        int a1 = 0, a2 = 0;                              // GPSMI_CORR_SMALL="a1,a2"
        const char* e = getenv("GPSMI_CORR_SMALL");
        if (e && sscanf(e, "%d,%d", &a1, &a2) == 2 && a1 >= 0 && a2 >= a1) *value = key[10] == '1' ? a1 : a2;
    } else if (o->env) {
        if (const char* e = getenv(o->env)) *value = atoll(e);
    }
    return true;
}

// the page-locked blocks gpsmi_host_alloc has handed out
struct HostAlloc { char* base; size_t bytes; char* dev; };
static std::vector<HostAlloc> g_host_allocs;
static std::mutex g_host_mutex;

bool host_alloc_lookup(const void* p, size_t bytes, void** dev) {
    std::lock_guard<std::mutex> lock(g_host_mutex);
    const char* c = static_cast<const char*>(p);
    for (const auto& a : g_host_allocs)
        if (c >= a.base && c + bytes <= a.base + a.bytes) {
            *dev = a.dev + (c - a.base);
            return true;
        }
    return false;
}

void make_twiddles(std::vector<float2>& tw) {
    tw.resize(2048);
    for (int k = 0; k < 2048; ++k) {
        double a = -2.0 * M_PI * (double)k / 2048.0;
        tw[k] = make_float2((float)cos(a), (float)sin(a));
    }
}

// streamData's sample decode (reference src/gpsrecv.py:170-172):
// raw = Q<<8 | I; sample = complex64(I + jQ)/127.5 - (1+1j).  numpy divides a
// complex64 by the real scalar as (a + b*0) * fl32(1/127.5) (Smith's algorithm
// with a zero imaginary divisor), i.e. a multiply by the rounded reciprocal:
// bit-exact only when done the same way (tests/test_abi.py checks all 256).
__global__ __launch_bounds__(256) void unpack_u8iq_kernel(float2* __restrict__ out,
                                                          const uint16_t* __restrict__ raw,
                                                          size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned v = raw[i];
        const float scl = 1.0f / 127.5f;
        float re = sub_rn(mul_rn((float)(v & 0xFF), scl), 1.0f);
        float im = sub_rn(mul_rn((float)(v >> 8), scl), 1.0f);
        out[i] = make_float2(re, im);
    }
}

}  // namespace gpsmi

using namespace gpsmi;

extern "C" {

const char* gpsmi_last_error(void) { return last_error_buf(); }
const char* gpsmi_version(void) { return "gpsmi 0.1 (gfx950)"; }

int gpsmi_abi_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(gpsmi_cfg);
        case 1: return (int)sizeof(gpsmi_peak);
        case 2: return (int)sizeof(gpsmi_trk_state);
        case 3: return (int)sizeof(gpsmi_trk_out);
        case 4: return (int)offsetof(gpsmi_trk_out, code_phase);
        default: return -1;
    }
}

int gpsmi_set_default(const char* key, long long value) {
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    OptDef* o = find_opt(key);
    if (!o) return fail(GPSMI_E_ARG, "gpsmi_set_default: unknown option '%s'", key ? key : "(null)");
    o->set = true;
    o->value = value;
    return GPSMI_OK;
}

int gpsmi_clear_default(const char* key) {
    std::lock_guard<std::mutex> lock(g_opts_mutex);
    OptDef* o = find_opt(key);
    if (!o) return fail(GPSMI_E_ARG, "gpsmi_clear_default: unknown option '%s'", key ? key : "(null)");
    o->set = false;
    return GPSMI_OK;
}

int gpsmi_device_count(int* n) {
    GPSMI_REQUIRE(n, "null argument");
    GPSMI_HIP(hipGetDeviceCount(n));
    return GPSMI_OK;
}

int gpsmi_device_name(int device, char* buf, size_t len) {
    GPSMI_REQUIRE(buf && len > 0, "null argument");
    hipDeviceProp_t p;
    GPSMI_HIP(hipGetDeviceProperties(&p, device));
    snprintf(buf, len, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return GPSMI_OK;
}

int gpsmi_dev_alloc(int device, size_t bytes, void** dptr) {
    GPSMI_REQUIRE(dptr && bytes > 0, "null pointer or zero size");
    GPSMI_HIP(hipSetDevice(device));
    GPSMI_HIP(hipMalloc(dptr, bytes));
    return GPSMI_OK;
}

int gpsmi_dev_free(int device, void* dptr) {
    if (!dptr) return GPSMI_OK;
    GPSMI_HIP(hipSetDevice(device));
    GPSMI_HIP(hipFree(dptr));
    return GPSMI_OK;
}

int gpsmi_dev_upload(int device, void* dptr, const void* host, size_t bytes) {
    GPSMI_REQUIRE(dptr && host, "null pointer");
    GPSMI_HIP(hipSetDevice(device));
    GPSMI_HIP(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
    return GPSMI_OK;
}

int gpsmi_dev_download(int device, void* host, const void* dptr, size_t bytes) {
    GPSMI_REQUIRE(dptr && host, "null pointer");
    GPSMI_HIP(hipSetDevice(device));
    GPSMI_HIP(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
    return GPSMI_OK;
}

int gpsmi_host_alloc(size_t bytes, void** hptr) {
    GPSMI_REQUIRE(hptr && bytes > 0, "null pointer or zero size");
    GPSMI_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    void* dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *hptr, 0) == hipSuccess && dev) {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        g_host_allocs.push_back({static_cast<char*>(*hptr), bytes, static_cast<char*>(dev)});
    } else {
        (void)hipGetLastError();
    }
    return GPSMI_OK;
}

int gpsmi_host_free(void* hptr) {
    if (!hptr) return GPSMI_OK;
    {
        std::lock_guard<std::mutex> lock(g_host_mutex);
        for (size_t i = 0; i < g_host_allocs.size(); ++i)
            if (g_host_allocs[i].base == hptr) {
                g_host_allocs.erase(g_host_allocs.begin() + i);
                break;
            }
    }
    GPSMI_HIP(hipHostFree(hptr));
    return GPSMI_OK;
}

int gpsmi_dev_sync(int device) {
    GPSMI_HIP(hipSetDevice(device));
    GPSMI_HIP(hipDeviceSynchronize());
    return GPSMI_OK;
}

int gpsmi_dev_unpack_u8iq(int device, void* d_iq, const void* d_raw, size_t n) {
    GPSMI_REQUIRE(d_iq && d_raw, "null pointer");
    if (n == 0) return GPSMI_OK;
    GPSMI_HIP(hipSetDevice(device));
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(unpack_u8iq_kernel, dim3((unsigned)blocks), dim3(256), 0, 0, (float2*)d_iq,
                       (const uint16_t*)d_raw, n);
    GPSMI_HIP(hipGetLastError());
    GPSMI_HIP(hipDeviceSynchronize());
    return GPSMI_OK;
}

}  // extern "C"
