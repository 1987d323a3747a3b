// Standalone probe: how fast can one workgroup per CU stream 512 KiB blocks into
// an LDS ring with global_load_lds_dwordx4, as a function of waves, ring depth
// and synchronisation.  Diagnostic only; not part of libgpsmi.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE 0: wait + barrier per row; 1: wait only (no barrier); 2: plain register loads (dwordx4), no LDS
template <int NW, int RING, int MODE>
__global__ __launch_bounds__(NW * 64) void probe(const char* __restrict__ src, float* out, int rows) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * 16384];
    constexpr int PP = 16 / NW;
    constexpr int D = RING - 1;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const char* g0 = src + (size_t)blockIdx.x * 524288 + (size_t)(wave * PP) * 1024 + lane * 16;
    float acc = 0.f;
    if (MODE == 2) {
        float4 buf[4][PP];
        for (int r = 0; r < 3; ++r)
            for (int pp = 0; pp < PP; ++pp) buf[r][pp] = *(const float4*)(g0 + (size_t)r * 16384 + pp * 1024);
        for (int r = 0; r < rows; r += 4) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (r + rr + 3 < rows)
                    for (int pp = 0; pp < PP; ++pp)
                        buf[(rr + 3) & 3][pp] = *(const float4*)(g0 + (size_t)(r + rr + 3) * 16384 + pp * 1024);
                for (int pp = 0; pp < PP; ++pp) acc += buf[rr][pp].x;
            }
        }
    } else {
        auto issue = [&](int r) {
#pragma unroll
            for (int pp = 0; pp < PP; ++pp)
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(g0 + (size_t)r * 16384 + pp * 1024),
                                                 (lds_ptr_t)(smem + (r % RING) * 16384 + (wave * PP + pp) * 1024), 16, 0, 0);
        };
        for (int r = 0; r < D; ++r) issue(r);
        for (int r = 0; r < rows; ++r) {
            int left = rows - 1 - r; if (left > D - 1) left = D - 1;      // rows that may stay in flight
            switch (left) {
                case 0: wait_vm<0>(); break;
                case 1: wait_vm<1 * PP>(); break;
                case 2: wait_vm<2 * PP>(); break;
                case 3: wait_vm<3 * PP>(); break;
                case 4: wait_vm<4 * PP>(); break;
                case 5: wait_vm<5 * PP>(); break;
                case 6: wait_vm<6 * PP>(); break;
                default: wait_vm<(7 * PP > 63 ? 63 : 7 * PP)>(); break;
            }
            if (MODE == 0) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (r + D < rows) issue(r + D);
            float v;
            unsigned a = (unsigned)(size_t)(lds_ptr_t)(smem + (r % RING) * 16384 + t * 16);
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v;
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int NW, int RING, int MODE>
void run(const char* name, const char* d, float* o, int nb) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<NW, RING, MODE>), dim3(nb), dim3(NW * 64), 0, 0, d, o, 32);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<NW, RING, MODE>), dim3(nb), dim3(NW * 64), 0, 0, d, o, 32);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-34s waves %d ring %d : %.4f ms  %.0f GB/s\n", name, NW, RING, ms, nb * 524288.0 / ms / 1e6);
}

int main() {
    const int nb = 1024;
    char* d; float* o;
    hipMalloc(&d, (size_t)nb * 524288); hipMalloc(&o, 64);
    hipMemset(d, 1, (size_t)nb * 524288);
    run<8, 8, 0>("dma wait+barrier", d, o, nb);
    run<8, 8, 1>("dma wait only", d, o, nb);
    run<4, 8, 0>("dma wait+barrier", d, o, nb);
    run<4, 8, 1>("dma wait only", d, o, nb);
    run<4, 4, 0>("dma wait+barrier", d, o, nb);
    run<4, 4, 1>("dma wait only", d, o, nb);
    run<8, 4, 1>("dma wait only", d, o, nb);
    run<16, 8, 1>("dma wait only", d, o, nb);
    run<4, 2, 1>("dma wait only", d, o, nb);
    run<8, 8, 2>("register loads", d, o, nb);
    run<4, 8, 2>("register loads", d, o, nb);
    run<16, 8, 2>("register loads", d, o, nb);
    return 0;
}
