// Probe: register layout of v_mfma_f32_32x32x2_f32 on gfx950 (A 32x2, B 2x32, D 32x32).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A[i][k] = 100*i + k + 1 in lane (i = l%32, k = l/32); B[k][j] = (k == 0) ? 1 : 1000 for all j... use j too
    const float a = 100.f * (l % 32) + (l / 32) + 1.f;          // A[i][k]
    const float b = (l / 32 == 0 ? 1.f : 0.001f) * (1.f + 0.0f) + 0.000001f * (l % 32) * 0;   // B[k][j]
    const float bj = (l / 32 == 0) ? (float)((l % 32) + 1) : 0.f;   // B[0][j] = j+1, B[1][j] = 0
    f16v c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bj, c, 0, 0, 0);
    for (int v = 0; v < 16; ++v) out[l * 16 + v] = c[v];
    (void)b;
}
int main() {
    float* d; hipMalloc(&d, 64 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[1024]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D[i][j] = A[i][0] * B[0][j] = (100 i + 1) * (j + 1)  ->  recover (i, j) of every (lane, vgpr)
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int v = 0; v < 16; ++v) {
            const int j = l % 32, i = 8 * (v / 4) + 4 * (l / 32) + v % 4;
            const float want = (100.f * i + 1.f) * (j + 1);
            if (h[l * 16 + v] != want) { if (bad < 5) printf("lane %d v %d: got %g want %g\n", l, v, h[l * 16 + v], want); ++bad; }
        }
    printf("layout D[i = 8*(v/4) + 4*(lane/32) + v%%4][j = lane%%32]: %d mismatches\n", bad);
    return 0;
}
