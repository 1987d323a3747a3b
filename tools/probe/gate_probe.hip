// Can a kernel release a second stream before it ends?  Stream B waits (hipStreamWaitValue32) on a word
// of signal memory that kernel A, running on stream A, writes when its FIRST workgroup is done; kernel B
// behind the wait should then start while A's other workgroups still run.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/gate_probe.hip -o tools/probe/gate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void kernel_a(unsigned* sig, unsigned seq, unsigned* cnt, unsigned long long* stamps, int spin_base) {
    // workgroup i spins ~ (1 + i % 4) * spin_base clocks: the first ones finish early
    const unsigned long long t0 = wall_clock64();
    const unsigned long long want = (unsigned long long)(1 + blockIdx.x % 4) * spin_base;
    while (wall_clock64() - t0 < want) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        if (atomicAdd(cnt, 1u) == 0) {
            __hip_atomic_store(sig, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            stamps[0] = wall_clock64();
        }
        stamps[1] = wall_clock64();          // (the last writer wins: ~ the end of A)
    }
}
__global__ void kernel_b(unsigned long long* stamps) {
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[2] = wall_clock64();
}

int main() {
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) return 0;
    unsigned* sig = nullptr;
    CK(hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory));
    *sig = 0;
    unsigned* cnt; unsigned long long* stamps;
    CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&stamps, 32));
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    for (int it = 1; it <= 3; ++it) {
        CK(hipMemset(cnt, 0, 4)); CK(hipMemset(stamps, 0, 32));
        CK(hipDeviceSynchronize());
        // B first: the wait is in its queue before A is launched
        CK(hipStreamWaitValue32(sb, sig, (unsigned)it, hipStreamWaitValueGte, 0xFFFFFFFFu));
        hipLaunchKernelGGL(kernel_b, dim3(1), dim3(64), 0, sb, stamps);
        hipLaunchKernelGGL(kernel_a, dim3(512), dim3(256), 0, sa, sig, (unsigned)it, cnt, stamps, 2500);   // 100 MHz clock: 25 .. 100 us
        CK(hipDeviceSynchronize());
        unsigned long long h[4];
        CK(hipMemcpy(h, stamps, 32, hipMemcpyDeviceToHost));
        printf("run %d: signal written at 0, B started at %+.1f us, A's last workgroup at %+.1f us\n", it,
               ((double)h[2] - (double)h[0]) / 100.0, ((double)h[1] - (double)h[0]) / 100.0);
    }
    // what the call costs the host: into an idle stream, and into a stream with a kernel in flight
    {
        using clk = std::chrono::steady_clock;
        CK(hipDeviceSynchronize());
        const unsigned v = *sig;
        double idle = 0, busy = 0, launch = 0;
        for (int i = 0; i < 20; ++i) {
            auto t0 = clk::now();
            CK(hipStreamWaitValue32(sb, sig, v, hipStreamWaitValueGte, 0xFFFFFFFFu));
            idle += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            CK(hipStreamSynchronize(sb));
        }
        for (int i = 0; i < 20; ++i) {
            CK(hipMemset(cnt, 0, 4));
            CK(hipDeviceSynchronize());
            auto t0 = clk::now();
            hipLaunchKernelGGL(kernel_a, dim3(512), dim3(256), 0, sb, sig, v, cnt, stamps, 2500);
            launch += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            t0 = clk::now();
            CK(hipStreamWaitValue32(sb, sig, v, hipStreamWaitValueGte, 0xFFFFFFFFu));
            busy += std::chrono::duration<double, std::micro>(clk::now() - t0).count();
            CK(hipStreamSynchronize(sb));
        }
        printf("host time of hipStreamWaitValue32: %.1f us into an idle stream, %.1f us behind a 100-us kernel "
               "(the launch of that kernel: %.1f us)\n", idle / 20, busy / 20, launch / 20);
    }
    return 0;
}
