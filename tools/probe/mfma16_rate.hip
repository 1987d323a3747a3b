// The rate of v_mfma_f32_16x16x4_f32 under load, at 1..6 waves per SIMD, alone and beside packed FMAs of
// OTHER waves (mode 1: workgroups of 512 threads, waves 0-3 matrix, waves 4-7 packed FMA: wave w and w + 4
// share a SIMD, so every SIMD holds the same number of each kind; mode 3: the same with 4x4x1): the matrix-pipe floor the eight-row correlators (gpsmi_trk_span8.h,
// trk_span_kernel<..,8>) are priced against.  Per iteration a wave issues 8 matrix instructions on two
// alternating accumulators (the correlators' pattern) or 32 packed FMAs.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/mfma16_rate tools/probe/mfma16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>      // 0: matrix only, 1 / 3: half the waves of a 512-thread workgroup matrix (16x16x4 / 4x4x1), half packed FMA, 2: four accumulators, 4: packed FMA only
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    v4f acc[4];
    v2f f[16];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = v4f{a, a, a, a};
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = v2f{a + i, b};
    const v2f m = v2f{0.999f, 1.001f}, c = v2f{1e-3f, 2e-3f};
    const bool do_f = MODE == 4 || MODE == 6 || ((MODE == 1 || MODE == 3 || MODE == 5) && threadIdx.x >= 256);
    float g[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) g[i] = a + i;
    if (!do_f) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = (MODE == 2 || MODE == 3) ? (i & 3) : (i & 1);
                if (MODE == 3) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
            }
        }
    } else if (MODE == 5 || MODE == 6) {       // 64 scalar v_fma_f32 per iteration: the flops of 32 packed ones
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[i & 31]) : "v"(m.x), "v"(c.x));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i & 15] = __builtin_elementwise_fma(f[i & 15], m, c);
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) f[i & 15].x += g[i];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
void run(int wg, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const bool mixed = MODE == 1 || MODE == 3 || MODE == 5;
    const dim3 grid(mixed ? 256 * wg / 2 : 256 * wg), block(mixed ? 512 : 256);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, 2000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e6 / iters;     // ns per iteration of the whole SIMD
    if (MODE == 5)
        printf("mode 5 (on every SIMD %d waves 16x16x4, 8 per iteration, and %d waves scalar v_fma_f32, 64 per iteration): %.1f ns per iteration\n", wg / 2, wg / 2, per);
    else if (MODE == 6)
        printf("mode 6 waves/SIMD %d: %.1f ns per iteration = %.2f ns per scalar FMA and SIMD\n", wg, per, per / (64.0 * wg));
    else if (MODE == 1 || MODE == 3)
        printf("mode %d (on every SIMD %d waves %s, 8 per iteration, and %d waves packed FMA, 32 per iteration): %.1f ns per iteration\n",
               MODE, wg / 2, MODE == 1 ? "16x16x4" : "4x4x1", wg / 2, per);
    else if (MODE == 4)
        printf("mode 4 waves/SIMD %d: %.1f ns per iteration = %.2f ns per packed FMA and SIMD\n", wg, per, per / (32.0 * wg));
    else
        printf("mode %d waves/SIMD %d: %.1f ns per iteration = %.2f ns per matrix instruction and SIMD (%s accumulators)\n",
               MODE, wg, per, per / (8.0 * wg), MODE == 2 ? "four" : "two");
}

int main() {
    float* out;
    (void)hipMalloc(&out, (size_t)256 * 8 * 512 * sizeof(float));   // blockIdx.x * 512 + threadIdx.x, up to 256 * 6 blocks
    for (int wg = 1; wg <= 6; ++wg) {
        run<0>(wg, out); run<2>(wg, out); run<4>(wg, out); run<6>(wg, out);
        if (wg % 2 == 0) { run<1>(wg, out); run<3>(wg, out); run<5>(wg, out); }
    }
    return 0;
}
