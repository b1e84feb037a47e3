// clock_trace.hip -- the shader clock over time while other kernels run: one wave samples s_memtime (shader cycles) against
// s_memrealtime (100 MHz) every `period` real-time ticks and logs both.  Built as a shared object and launched from a
// Python probe on a stream of its own (tools/probes/cosched_probe.py): the chip clocks to its power budget, and what two
// co-resident kernels draw together comes off the clock of both.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/clock_trace.hip -o tools/ubench/bin/libclock_trace.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(64) void k_clock_trace(unsigned long long* out, int n, unsigned period) {
    if (threadIdx.x != 0) return;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
        const unsigned long long c = __builtin_amdgcn_s_memtime();
        const unsigned long long r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = r;
        out[2 * i + 1] = c;
        while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)period * (i + 1)) __builtin_amdgcn_s_sleep(32);
    }
}

extern "C" int clock_trace_launch(void* stream, unsigned long long* out, int n, unsigned period) {
    hipLaunchKernelGGL(k_clock_trace, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), out, n, period);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
