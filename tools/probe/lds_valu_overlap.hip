// Do LDS exchanges and packed-FMA work of one SIMD overlap?  Workgroups of 512 threads: waves 0-3 (one per
// SIMD) move data through LDS the way a transform pass does (8 ds_write_b64 + 8 ds_read_b64 per iteration,
// conflict-free unit-stride addresses, a wave-private region), waves 4-7 (the same SIMDs) issue 32 packed
// FMAs per iteration.  Modes: 0 = all eight waves LDS, 1 = all eight waves packed FMA, 2 = four waves of
// each kind (waves w and w + 4 share a SIMD), 3 = every wave alternates the two (8 + 8 LDS operations,
// then 32 packed FMAs), 4 = all eight waves LDS with 16-byte operations (4 ds_write_b128 + 4 ds_read_b128:
// the same 8 KiB per wave and iteration), 5 = all eight waves LDS with SINGLE 8-byte operations (the row stride is
// a kernel argument, so hipcc cannot pair them into ds_*2_b64 as it does in mode 0: what the transforms of
// gpsmi_fft.h issue for their stride-256 gathers).
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/lds_valu_overlap tools/probe/lds_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int stride) {
    __shared__ v2f buf[8][8 * 64];                 // one region of 8 x 64 elements per wave: 32 KiB
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    v2f f[16], x[8];
    const float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-6f;
#pragma unroll
    for (int i = 0; i < 16; ++i) f[i] = v2f{a + i, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = v2f{a, b + i};
    const v2f m = v2f{0.999f, 1.001f}, c = v2f{1e-3f, 2e-3f};
    const bool lds_role = MODE == 0 || MODE == 3 || (MODE == 2 && wave < 4);
    const bool fma_role = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= 4);
    v2f* mine = &buf[wave][0];
    for (int it = 0; it < iters; ++it) {
        if (lds_role) {
#pragma unroll
            for (int r = 0; r < 8; ++r) mine[64 * r + lane] = x[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = mine[64 * r + (lane ^ 1)];
            asm volatile("" ::: "memory");
        }
        if (MODE == 4) {
            v4f* mine4 = reinterpret_cast<v4f*>(mine);       // 256 elements of 16 bytes per wave
#pragma unroll
            for (int r = 0; r < 4; ++r) mine4[64 * r + lane] = v4f{x[2 * r].x, x[2 * r].y, x[2 * r + 1].x, x[2 * r + 1].y};
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const v4f q = mine4[64 * r + (lane ^ 1)];
                x[2 * r] = v2f{q.x, q.y}; x[2 * r + 1] = v2f{q.z, q.w};
            }
            asm volatile("" ::: "memory");
        }
        if (MODE == 5) {
#pragma unroll
            for (int r = 0; r < 8; ++r) mine[stride * r + lane] = x[r];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] = mine[stride * r + (lane ^ 1)];
            asm volatile("" ::: "memory");
        }
        if (fma_role) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(f[i & 15]) : "v"(m), "v"(c));
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += f[i].x + f[i].y;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 512 + threadIdx.x] = s;       // grid <= 256 * 4 workgroups of 512: see main
}

template <int MODE>
void run(int wg, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(512), 0, 0, out, 2000, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * wg), dim3(512), 0, 0, out, iters, 64);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const char* what[] = {"eight LDS waves", "eight packed-FMA waves", "four LDS waves + four packed-FMA waves",
                          "eight waves, each alternating LDS and packed FMA", "eight LDS waves, 16-byte operations", "eight LDS waves, single 8-byte operations"};
    printf("mode %d, %d workgroups of 512 per CU (%s): %.1f ns per iteration\n", MODE, wg, what[MODE], ms * 1e6 / iters);
}

int main() {
    float* out;
    (void)hipMalloc(&out, (size_t)256 * 4 * 512 * sizeof(float));
    for (int wg = 1; wg <= 3; ++wg) { run<0>(wg, out); run<1>(wg, out); run<2>(wg, out); run<3>(wg, out); run<4>(wg, out); run<5>(wg, out); }
    return 0;
}
