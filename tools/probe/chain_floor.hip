// What the closed loop's structure costs with its arithmetic taken out: three dependent
// launches per 32-ms block (code-phase correlation -> correlator -> epilogue), each with the
// grid of the real kernel and the real kernel's chain of DEPENDENT trips to memory (a value
// loaded in one trip addresses the next), nothing else.  Evidence for DESIGN.md section 5: how
// far below the measured 24 us per block the three-launch form could go at all.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/chain_floor.hip -o tools/probe/chain_floor
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel() {}

// one trip: every thread loads one value it was not told about and stores it
__global__ void one_trip(const int* __restrict__ a, int* __restrict__ out) {
    out[blockIdx.x * blockDim.x + threadIdx.x] = a[blockIdx.x * blockDim.x + threadIdx.x];
}

// code-phase correlation: state row -> (index from it) eight rows of the block at once ->
// (index from them) the replica spectrum -> record.  12 workgroups x 256 threads.
__global__ void k_corr(const int* __restrict__ state, const int* __restrict__ blk,
                       const int* __restrict__ rep, int* __restrict__ mid) {
    const int t = threadIdx.x, ch = blockIdx.x;
    const int s = state[ch * 138];                       // trip 1 (552-byte state rows)
    int acc = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) acc += blk[(s & 1) + (12 + r) * 4096 + t * 2];   // trip 2: the centre rows
    const int v = rep[(acc & 1) + ch * 4096 + t];        // trip 3: the replica spectrum
    if (t == 0) mid[ch * 8] = v & 1;                     // the job descriptor
}
// correlator: job descriptors -> one tile of rows (16 x 16 bytes per lane) -> records.
// 32 one-wave workgroups.
__global__ void k_span(const int* __restrict__ mid, const int4* __restrict__ blk, int* __restrict__ rec) {
    const int lane = threadIdx.x, span = blockIdx.x;
    const int m = mid[(lane & 7) * 8 % 96];              // trip 1
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {                       // trip 2
        const int4 v = blk[(m & 1) + (i * 2048 + span * 64 + lane) ];
        acc += v.x + v.w;
    }
    rec[span * 2048 + lane] = acc;
}
// epilogue: descriptor + state -> the 32 span records of the job -> output record and state.
// 12 workgroups x 256 threads.
__global__ void k_epi(const int* __restrict__ mid, int* __restrict__ state, const int* __restrict__ rec,
                      int* __restrict__ out) {
    const int t = threadIdx.x, ch = blockIdx.x;
    const int m = mid[ch * 8] + state[ch * 138 + 1];     // trip 1
    int acc = 0;
#pragma unroll
    for (int s = 0; s < 8; ++s) acc += rec[(m & 1) + ((t >> 6) * 8 + s) * 2048 + (t & 63)];   // trip 2
    if ((t & 63) == 0) { out[ch * 90 + (t >> 6)] = acc; state[ch * 138] = acc & 1; }
}

int main() {
    int *state, *blk, *rep, *mid, *rec, *out;
    CK(hipMalloc(&state, 12 * 552));
    CK(hipMalloc(&blk, 65536 * 8 + 4096));
    CK(hipMalloc(&rep, 38 * 2048 * 8));
    CK(hipMalloc(&mid, 96 * 4 * 8));
    CK(hipMalloc(&rec, 32 * 2048 * 4 + 4096));
    CK(hipMalloc(&out, 12 * 360));
    CK(hipMemset(state, 0, 12 * 552)); CK(hipMemset(blk, 0, 65536 * 8 + 4096)); CK(hipMemset(rep, 0, 38 * 2048 * 8));
    CK(hipMemset(mid, 0, 96 * 4 * 8)); CK(hipMemset(rec, 0, 32 * 2048 * 4 + 4096));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 2000;
    auto timed = [&](const char* what, auto body) {
        for (int i = 0; i < 200; ++i) body();
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < n; ++i) body();
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-78s %6.2f us per block\n", what, ms / n * 1e3);
    };
    timed("three empty kernels (grids 12x256, 32x64, 12x256)", [&] {
        hipLaunchKernelGGL(empty_kernel, dim3(12), dim3(256), 0, 0);
        hipLaunchKernelGGL(empty_kernel, dim3(32), dim3(64), 0, 0);
        hipLaunchKernelGGL(empty_kernel, dim3(12), dim3(256), 0, 0);
    });
    timed("three kernels of one load and one store each", [&] {
        hipLaunchKernelGGL(one_trip, dim3(12), dim3(256), 0, 0, state, mid);
        hipLaunchKernelGGL(one_trip, dim3(32), dim3(64), 0, 0, mid, rec);
        hipLaunchKernelGGL(one_trip, dim3(12), dim3(256), 0, 0, rec, out);
    });
    timed("the closed loop's dependent trips (3 + 2 + 2), no arithmetic", [&] {
        hipLaunchKernelGGL(k_corr, dim3(12), dim3(256), 0, 0, state, blk, rep, mid);
        hipLaunchKernelGGL(k_span, dim3(32), dim3(64), 0, 0, mid, (const int4*)blk, rec);
        hipLaunchKernelGGL(k_epi, dim3(12), dim3(256), 0, 0, mid, state, rec, out);
    });
    timed("the first of them alone (three dependent trips): what a launch with its trips costs", [&] {
        hipLaunchKernelGGL(k_corr, dim3(12), dim3(256), 0, 0, state, blk, rep, mid);
    });
    return 0;
}
