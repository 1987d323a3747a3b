// Standalone probe: issue cost of the packed complex MAC of gpsmi_trk_stream.h
// (four v_pk_fma_f32 + s_nop) per wave, alone and with a second wave on the SIMD.
// Diagnostic only; not part of libgpsmi.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cmac2(v2f& a0, v2f& a1, v2f b0, v2f x0, v2f b1, v2f x1) {
    asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %5, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
        "s_nop 0"
        : "+v"(a0), "+v"(a1)
        : "v"(b0), "v"(x0), "v"(b1), "v"(x1));
}
// plain: the same arithmetic as scalar fmas chosen by the compiler
__device__ __forceinline__ void cmac_plain(v2f& a, v2f b, v2f x) {
    a.x = fmaf(b.x, x.x, a.x); a.x = fmaf(-b.y, x.y, a.x);
    a.y = fmaf(b.x, x.y, a.y); a.y = fmaf(b.y, x.x, a.y);
}

template <int MODE>
__global__ void probe(float* out, long long* cyc, int iters) {
    const int t = threadIdx.x;
    v2f acc[6], B[6][8], x[8];
    for (int c = 0; c < 6; ++c) {
        acc[c] = v2f{0.f, 0.f};
        for (int j = 0; j < 8; ++j) B[c][j] = v2f{(float)(t + c) * 1e-3f, (float)(j + 1) * 1e-3f};
    }
    for (int j = 0; j < 8; ++j) x[j] = v2f{(float)t * 1e-4f, (float)j};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
#pragma unroll
                for (int c = 0; c < 6; c += 2) cmac2(acc[c], acc[c + 1], B[c][j], x[j], B[c + 1][j], x[j]);
            } else {
#pragma unroll
                for (int c = 0; c < 6; ++c) cmac_plain(acc[c], B[c][j], x[j]);
            }
        }
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < 6; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + t] = s;
    if ((t & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (t >> 6)] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 1 << 20);
    const int iters = 2000;
    long long h[64];
    for (int mode = 0; mode < 2; ++mode)
        for (int threads : {64, 256, 512, 1024}) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(long long) * (threads / 64), hipMemcpyDeviceToHost);
            // 48 complex MACs per iteration = 96 v_pk_fma (mode 0) or 192 v_fma (mode 1)
            printf("mode %d waves %2d (per SIMD %.1f): %.1f clock64 ticks per iteration of 48 cmac (wave 0)\n",
                   mode, threads / 64, threads / 256.0, (double)h[0] / iters);
        }
    // aggregate throughput: every CU filled with `wpc` waves
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wpc : {4, 8, 16, 32}) {
            const int threads = 256, wgs = 256 * (wpc / 4) * 4;   // several rounds per CU
            const int it2 = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(threads), 0, 0, out, cyc, it2);
                else hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(threads), 0, 0, out, cyc, it2);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)wgs * threads * it2 * 48 * 8;
            printf("mode %d: %d workgroups x 256 threads (~%d waves/CU if all resident): %.3f ms, %.1f TFLOP/s\n",
                   mode, wgs, wpc, ms, flop / ms * 1e-9);
        }
    return 0;
}
