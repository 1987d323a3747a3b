// Probe: issue interval of dependent v_mfma_f32_32x32x2_f32 (same accumulator) on gfx950,
// alone and with ~8 VALU instructions in between.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
template <int VALU>
__global__ void k(float* out, long long* cyc, int n) {
    const int l = threadIdx.x;
    float a = 1.f + l, b = 0.5f, z = 0.25f;
    f16v c = {0};
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (VALU) {
#pragma unroll
                for (int v = 0; v < 8; ++v) z = fmaf(z, 1.0001f, 0.001f);
                b = z;
            }
            c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    float s = 0; for (int v = 0; v < 16; ++v) s += c[v];
    out[blockIdx.x * blockDim.x + l] = s;
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* d; long long* c; hipMalloc(&d, 1 << 22); hipMalloc(&c, 64);
    long long h;
    for (int valu = 0; valu < 2; ++valu)
        for (int waves : {1, 4}) {
            if (valu) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, d, c, 1000);
            else hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, d, c, 1000);
            hipDeviceSynchronize();
            hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
            printf("valu %d, %d wave(s) per workgroup (1 per SIMD): %.1f cycles per MFMA\n", valu, waves, (double)h / 4000.0);
        }
    return 0;
}
