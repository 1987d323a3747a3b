// Standalone probe: does a v_pk_fma_f32 that reads the result of the immediately
// preceding v_pk_fma_f32 need a wait state on gfx950?  Compares a dependent chain
// of packed FMAs (inline asm, no s_nop) with the same arithmetic done by fmaf.
// Also times cmac variants with and without trailing s_nop.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void chain(float* out, int n) {
    const int t = threadIdx.x;
    v2f a = v2f{0.1f * t, -0.2f * t}, b = v2f{1.0001f, 0.37f + 1e-3f * t}, x = v2f{0.9f, -0.11f};
    v2f r = a;
    float px = a.x, py = a.y;
    for (int i = 0; i < n; ++i) {
        // back-to-back dependent: every instruction reads what the previous one wrote
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
                     "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
                     "v_pk_fma_f32 %0, %1, %0, %0 op_sel_hi:[0,1,1]\n\t"
                     "v_pk_mul_f32 %0, %0, %3 op_sel_hi:[1,0]"
                     : "+v"(r) : "v"(b), "v"(x), "v"(v2f{0.25f, 0.25f}));
        px = fmaf(b.x, x.x, px); py = fmaf(b.x, x.y, py);
        px = fmaf(-b.y, x.y, px); py = fmaf(b.y, x.x, py);
        const float qx = fmaf(b.x, px, px), qy = fmaf(b.x, py, py);
        px = qx * 0.25f; py = qy * 0.25f;
    }
    out[4 * t] = r.x; out[4 * t + 1] = r.y; out[4 * t + 2] = px; out[4 * t + 3] = py;
}

template <int NOP>
__device__ __forceinline__ void cmac6(v2f* acc, const v2f* B, v2f x) {
    // six channels, one element: 12 packed FMAs, dependent ones six apart
    if (NOP)
        asm volatile("v_pk_fma_f32 %0, %6, %12, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %7, %12, %1 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %2, %8, %12, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %9, %12, %3 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %4, %10, %12, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %11, %12, %5 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %0, %6, %12, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %1, %7, %12, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %2, %8, %12, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %4, %10, %12, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %5, %11, %12, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\ts_nop 0"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
            : "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(x));
    else
        asm volatile("v_pk_fma_f32 %0, %6, %12, %0 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %1, %7, %12, %1 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %2, %8, %12, %2 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %3, %9, %12, %3 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %4, %10, %12, %4 op_sel_hi:[0,1,1]\n\tv_pk_fma_f32 %5, %11, %12, %5 op_sel_hi:[0,1,1]\n\t"
            "v_pk_fma_f32 %0, %6, %12, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %1, %7, %12, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %2, %8, %12, %2 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %3, %9, %12, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\t"
            "v_pk_fma_f32 %4, %10, %12, %4 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %5, %11, %12, %5 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]"
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5])
            : "v"(B[0]), "v"(B[1]), "v"(B[2]), "v"(B[3]), "v"(B[4]), "v"(B[5]), "v"(x));
}

template <int NOP>
__global__ void timeit(float* out, long long* cyc, int iters) {
    const int t = threadIdx.x;
    v2f acc[6], B[8][6], x[8];
    for (int c = 0; c < 6; ++c) {
        acc[c] = v2f{0.f, 0.f};
        for (int j = 0; j < 8; ++j) B[j][c] = v2f{(float)(t + c) * 1e-3f, (float)(j + 1) * 1e-3f};
    }
    for (int j = 0; j < 8; ++j) x[j] = v2f{(float)t * 1e-4f, (float)j};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cmac6<NOP>(acc, B[j], x[j]);
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
    }
    const long long t1 = clock64();
    float s = 0.f;
    for (int c = 0; c < 6; ++c) s += acc[c].x + acc[c].y;
    out[blockIdx.x * blockDim.x + t] = s;
    if (t == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 64 << 20); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, 1000);
    float h[256];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; ++t)
        if (memcmp(&h[4 * t], &h[4 * t + 2], 8)) ++bad;
    printf("dependent packed chain without wait states: %d of 64 lanes differ from the fmaf chain (lane 5: %g %g vs %g %g)\n",
           bad, h[20], h[21], h[22], h[23]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int nop = 0; nop < 2; ++nop) {
        long long c; float ms = 0;
        const int iters = 20000, wgs = 4096;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            if (nop) hipLaunchKernelGGL(timeit<1>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(timeit<0>, dim3(wgs), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("cmac6 %s: wave 0 %.1f cycles per 96 v_pk_fma; chip %.1f TFLOP/s\n", nop ? "with s_nop" : "no s_nop  ",
               (double)c / iters, (double)wgs * 256 * iters * 48 * 8 / ms * 1e-9);
    }
    return 0;
}
