// What does trk_span8_kernel's READ PATTERN alone reach, and what would wider tiles reach?  512 blocks of
// 8 rows x 16368 complex64 (1,047,552 B each); a wave streams a range of a block tile by tile -- a tile =
// 8 row segments of SEG bytes, the rows 130,944 B apart, SEG / 16 x 8 / 64 buffer_load_dwordx4 ... nt per
// lane, the next tile requested before the present one is used -- and only adds the values up.
//   SEG  384 (the kernel's 48 positions): 31 ranges of 11 tiles per block
//   SEG  768 (96 positions):              17 ranges of 10 tiles (the last 48 positions of a row left out)
//   SEG 1408 (176 positions):             31 ranges of 3 tiles
//   SEG 4224 (528 positions = one range of the kernel in ONE tile): 31 ranges of 1 tile
// Occupancy is set with unused dynamic LDS (bytes per workgroup of four waves on the command line,
// default 23552 = the kernel's: six workgroups per CU).  Diagnostic only.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/probe/span8_read tools/probe/span8_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float mf4 __attribute__((ext_vector_type(4)));
constexpr int kBlockBytes = 8 * 16368 * 8, kRowBytes = 16368 * 8;

template <int SEG, int TILES, int RANGES>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ src, float* out, int nblocks) {
    extern __shared__ float pad_lds[];
    constexpr int NPL = SEG * 8 / 16 / 64, PPR = SEG / 16;     // loads per lane and tile, 16-byte pieces per row segment
    static_assert(NPL * 64 * 16 == SEG * 8, "a tile must be a whole number of wave loads");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int widx = blockIdx.x * 4 + wave;
    const int b = widx / RANGES, range = widx % RANGES;
    if (b >= nblocks) return;
    const char* blk = src + (size_t)b * kBlockBytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(blk), 0, kBlockBytes, 0x00020000);
    int ld_off[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int L = lane + 64 * i;
        ld_off[i] = (L / PPR) * kRowBytes + (L % PPR) * 16;
    }
    mf4 st[NPL];
    auto load = [&](int tix) {
        const int tb = (range * TILES + tix) * SEG;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            st[i] = __builtin_bit_cast(mf4, __builtin_amdgcn_raw_buffer_load_b128(rs, ld_off[i], tix < TILES ? tb : kBlockBytes, 2));
    };
    load(0);
    float acc = 0.f;
#pragma unroll 1
    for (int tix = 0; tix < TILES; ++tix) {
        mf4 cur[NPL];
#pragma unroll
        for (int i = 0; i < NPL; ++i) cur[i] = st[i];
        load(tix + 1);
#pragma unroll
        for (int i = 0; i < NPL; ++i) acc += cur[i].x + cur[i].y + cur[i].z + cur[i].w;
    }
    if (acc == 123.456f) out[widx] = acc + pad_lds[lane];      // (never: the sums stay alive)
}

template <int SEG, int TILES, int RANGES>
void run(const char* src, float* out, int nblocks, int lds) {
    const int nwaves = nblocks * RANGES;
    const dim3 grid((nwaves + 3) / 4), block(256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((rd<SEG, TILES, RANGES>), grid, block, lds, 0, src, out, nblocks);
    (void)hipDeviceSynchronize();
    const int reps = 30;
    (void)hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((rd<SEG, TILES, RANGES>), grid, block, lds, 0, src, out, nblocks);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)nblocks * RANGES * TILES * SEG * 8;
    printf("segments of %4d B, %2d ranges of %2d tiles, %5d dynamic LDS B per workgroup: %.4f ms per launch, %.0f GB/s (%.1f MB read)\n",
           SEG, RANGES, TILES, lds, ms / reps, bytes / (ms / reps) * 1e-6, bytes * 1e-6);
}

int main(int argc, char** argv) {
    const int nblocks = 512;
    char* src; float* out;
    (void)hipMalloc(&src, (size_t)nblocks * kBlockBytes);
    (void)hipMemset(src, 0, (size_t)nblocks * kBlockBytes);
    (void)hipMalloc(&out, (size_t)nblocks * 31 * sizeof(float) + 1024);
    for (int lds : {23552, 16384, 8192}) {
        if (argc > 1) lds = atoi(argv[1]);
        run<384, 11, 31>(src, out, nblocks, lds);
        run<768, 10, 17>(src, out, nblocks, lds);
        run<1408, 3, 31>(src, out, nblocks, lds);
        run<4224, 1, 31>(src, out, nblocks, lds);
        if (argc > 1) break;
    }
    return 0;
}
