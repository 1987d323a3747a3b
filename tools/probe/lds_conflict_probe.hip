// What SQ_LDS_BANK_CONFLICT counts on gfx950: each kernel below issues ONE kind of LDS access with an
// address pattern that is conflict-free by the bank rules of MI355X_MICROARCH.md (and one that is a
// known 2-way conflict), 4096 times per wave.  Run under
//   rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv
// A non-zero "conflict" count on the conflict-free wide accesses is what the counter adds for the
// second and later passes of 64- / 128-bit accesses; tools/lds_conflict_model.py then prices the span
// correlator's own mix of LDS instructions with these per-instruction figures.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/lds_conflict_probe.hip -o tools/probe/lds_conflict_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kIters = 4096;

#define PROBE(name, T, IDX, STORE)                                                        \
    __global__ __launch_bounds__(256) void name(float* out) {                             \
        __shared__ __attribute__((aligned(16))) float lds[16384];                         \
        const int l = threadIdx.x & 63, w = threadIdx.x >> 6;                             \
        T* p = reinterpret_cast<T*>(lds + w * 4096) + (IDX);                               \
        T acc = T{};                                                                      \
        if (STORE) {                                                                      \
            for (int i = 0; i < kIters; ++i) {                                            \
                acc += 1.0f;                                                              \
                *(volatile T*)p = acc;                                                    \
            }                                                                             \
        } else {                                                                          \
            *p = acc + 1.0f;                                                              \
            for (int i = 0; i < kIters; ++i) acc += *(volatile T*)p;                      \
        }                                                                                 \
        if (reinterpret_cast<float*>(&acc)[0] == 123.f) out[threadIdx.x] = 1.f;           \
    }

PROBE(write_b32_free, float, l, true)
PROBE(write_b64_free, f2, l, true)
PROBE(write_b128_free, f4, l, true)
PROBE(read_b32_free, float, l, false)
PROBE(read_b64_free, f2, l, false)
PROBE(read_b128_free, f4, l, false)
PROBE(read_b32_2way, float, 2 * l, false)          // lanes l and l + 16 of a half-wave share a bank
PROBE(write_b32_2way, float, 2 * l, true)
// the span correlator's own patterns: A operand (lane = (row, k): row pitch 130 dwords), tile store
// (two b64 per 16 bytes, row pitch 130 dwords)
PROBE(span_a_read_b32, float, (l & 15) * 130 + (l >> 4), false)
PROBE(span_tile_write_b64, f2, ((l >> 5) * 130 + 4 * (l & 31)) / 2, true)

int main() {
    float* d;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(write_b32_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(write_b64_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(write_b128_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(read_b32_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(read_b64_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(read_b128_free, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(read_b32_2way, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(write_b32_2way, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(span_a_read_b32, dim3(256), dim3(256), 0, 0, d);
        hipLaunchKernelGGL(span_tile_write_b64, dim3(256), dim3(256), 0, 0, d);
    }
    hipDeviceSynchronize();
    printf("done: %d LDS instructions per wave and kernel, 1024 waves\n", kIters);
    return 0;
}
