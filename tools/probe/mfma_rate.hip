// Probe: rate of v_mfma_f32_32x32x2_f32 on gfx950 as a function of waves per SIMD, of the
// number of accumulators a wave alternates between (1 = every MFMA depends on the one
// before it) and of the VALU instructions between two MFMAs.  Prints ticks (clock64) per
// MFMA per SIMD; the pipe itself needs 64 cycles.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int VALU, int NACC>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    float a = 1.f + l, b = 0.5f, z = 0.25f;
    f16v c[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) c[i] = f16v{0};
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (VALU) {
#pragma unroll
                for (int v = 0; v < VALU; ++v) z = fmaf(z, 1.0001f, 0.001f);
                b = z;
            }
            c[q % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[q % NACC], 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        for (int v = 0; v < 16; ++v) s += c[i][v];
    out[blockIdx.x * blockDim.x + l] = s;
    if ((l & 63) == 0 && blockIdx.x == 0) {      // all waves of the workgroup: first start, last end
        atomicMin((unsigned long long*)&cyc[1], (unsigned long long)t0);
        atomicMax((unsigned long long*)&cyc[2], (unsigned long long)t1);
    }
}
template <int VALU, int NACC>
void go(float* d, long long* c) {
    for (int waves : {4, 8, 16}) {
        long long h, init[3] = {0, (long long)(~0ull >> 1), 0}, got[3];
        hipMemcpy(c, init, 24, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<VALU, NACC>), dim3(256), dim3(64 * waves), 0, 0, d, c, 1000);
        hipDeviceSynchronize();
        hipMemcpy(got, c, 24, hipMemcpyDeviceToHost);
        h = got[2] - got[1];
        printf("valu %d, acc %d, %d wave(s) per SIMD: %.1f ticks per MFMA per wave, %.1f per SIMD\n", VALU,
               NACC, waves / 4, (double)h / 4000.0, (double)h / 4000.0 / (waves / 4));
    }
}
int main() {
    float* d; long long* c; hipMalloc(&d, 1 << 24); hipMalloc(&c, 64);
    go<0, 1>(d, c); go<3, 1>(d, c); go<8, 1>(d, c);
    go<0, 2>(d, c); go<3, 2>(d, c); go<8, 2>(d, c);
    go<3, 4>(d, c);
    return 0;
}
