// Probe: what does the correlator's READ PATTERN alone reach?  A workgroup of four waves
// streams one 512 KiB block (32 rows x 2048 complex64) tile by tile exactly as
// trk_stream_mfma_kernel<4> requests it -- eight buffer_load_dwordx4 per lane and tile, 256-byte
// row segments 16 KiB apart, the next tile in registers -- and only adds the values up.
//   PATTERN 0: wave w owns positions [512 w, 512 w + 512)         (the kernel's)
//   PATTERN 1: wave w owns tiles w, w + 4, w + 8, ...             (1 KiB of a row per workgroup step)
//   PATTERN 2: every tile is 8 KiB of consecutive bytes           (what the loads reach on a stream)
//   PATTERN 3: tile = 16 rows x 64 positions (512-byte row segments), the two row halves in turn
// Occupancy is set with unused dynamic LDS.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float mf4 __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ __launch_bounds__(256) void tile_read(const char* __restrict__ src, float* out, int spin) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* blk = src + (size_t)blockIdx.x * 524288;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blk), 0, 524288, 0x00020000);
    const int ld_off = PATTERN == 2 ? lane * 16
                     : PATTERN == 3 ? ((lane >> 5) * 2048 + 2 * (lane & 31)) * 8
                                    : ((lane >> 4) * 2048 + 2 * (lane & 15)) * 8;
    mf4 st[8];
    auto load = [&](int tix) {
        const int tb = PATTERN == 0 ? (512 * wave + 32 * tix) * 8
                     : PATTERN == 1 ? (4 * tix + wave) * 32 * 8
                     : PATTERN == 3 ? (512 * wave + 64 * (tix >> 1)) * 8 + (tix & 1) * 16 * 2048 * 8
                                    : (16 * wave + tix) * 8192;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            st[i] = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(
                rs, ld_off, tb + i * (PATTERN == 2 ? 1024 : PATTERN == 3 ? 2 * 2048 * 8 : 4 * 2048 * 8), 0));
    };
    load(0);
    float acc = 0.f;
#pragma unroll 1
    for (int tix = 0; tix < 16; ++tix) {
        mf4 cur[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] = st[i];
        load(tix + 1 < 16 ? tix + 1 : 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += cur[i].x + cur[i].y + cur[i].z + cur[i].w;
        for (int k = 0; k < spin; ++k) asm volatile("s_sleep 8");          // stand-in for the tile's compute
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int PATTERN>
void run(const char* name, const char* d, float* o, int dyn, int spin) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(tile_read<PATTERN>, dim3(1024), dim3(256), dyn, 0, d, o, spin);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(tile_read<PATTERN>, dim3(1024), dim3(256), dyn, 0, d, o, spin);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-46s dyn LDS %6d spin %2d : %.4f ms  %.0f GB/s\n", name, dyn, spin, ms, 1024 * 524288.0 / ms / 1e6);
}

int main() {
    char* d; float* o;
    hipMalloc(&d, (size_t)1024 * 524288); hipMalloc(&o, 64);
    hipMemset(d, 0, (size_t)1024 * 524288);
    hipFuncSetAttribute((const void*)tile_read<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)tile_read<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)tile_read<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)tile_read<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int dyn : {53000, 40000, 20000})          // 3, 4, 8 workgroups per CU
        for (int spin : {0, 4, 8}) {
            run<0>("wave = 512 consecutive positions (kernel)", d, o, dyn, spin);
            run<1>("waves interleave tiles (1 KiB of a row)", d, o, dyn, spin);
            run<2>("8 KiB consecutive per tile", d, o, dyn, spin);
            run<3>("16 rows x 512 B per tile", d, o, dyn, spin);
        }
    return 0;
}
