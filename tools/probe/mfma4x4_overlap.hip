// Does v_mfma_f32_4x4x1_16b_f32 overlap with packed-FMA work (a) of the same wave, (b) of other waves
// of the SIMD?  One workgroup of W waves per CU-quarter is not controllable, so: grid = 256 CUs x WG
// workgroups of 256 threads (4 waves = one per SIMD); WG = 1, 2, 3 gives 1, 2, 3 waves per SIMD.
// Modes: 0 = MFMA only (16 per iteration, 8 accumulators), 1 = packed FMA only (32 per iteration,
// 16 accumulators), 2 = both interleaved in one wave, 3 = even workgroups MFMA, odd ones packed FMA.
// Prints cycles per iteration (s_memtime of wave 0 of workgroup 0) and the kernel's wall time.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/mfma4x4_overlap tools/probe/mfma4x4_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cyc) {
    v4f acc[8];
    v2f f[16];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = v4f{a, a, a, a};
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = v2f{a + i, b};
    const v2f m = v2f{0.999f, 1.001f}, c = v2f{1e-3f, 2e-3f};
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 0);
    const bool do_f = MODE == 1 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1) == 1);
    const long long t0 = __builtin_readcyclecounter();
    if (do_m && do_f) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 7], 0, 0, 0);
                f[i] = __builtin_elementwise_fma(f[i], m, c);
                f[(i + 8) & 15] = __builtin_elementwise_fma(f[(i + 8) & 15], m, c);
            }
        }
    } else if (do_m) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i & 7] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i & 7], 0, 0, 0);
        }
    } else if (do_f) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i & 15] = __builtin_elementwise_fma(f[i & 15], m, c);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x < 2 && threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(int wg, float* out, long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d waves/SIMD %d: %.1f / %.1f counter ticks per iteration (wg 0 / wg 1), kernel %.3f ms = %.1f ns per iteration\n",
           MODE, wg, (double)h[0] / iters, (double)h[1] / iters, ms, ms * 1e6 / iters);
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float)); hipMalloc(&cyc, 16);
    printf("per iteration: 16 MFMA 4x4x1 (mode 0), 32 v_pk_fma_f32 (mode 1), both in one wave (2), split by workgroup (3)\n");
    for (int wg = 1; wg <= 3; ++wg) {
        run<0>(wg, out, cyc); run<1>(wg, out, cyc); run<2>(wg, out, cyc);
        if (wg > 1) run<3>(wg, out, cyc);
    }
    for (int wg = 4; wg <= 8; ++wg) {      // the issue rate of each kind alone at higher occupancy
        run<0>(wg, out, cyc); run<1>(wg, out, cyc);
    }
    return 0;
}
