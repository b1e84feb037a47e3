// Microbenchmark: sustained v_mfma_f32_16x16x4_f32 rate on this box with non-trivial operands.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(float* out, const float* in, int iters) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(t * 8 + i) & 0xffff];
        b[i] = in[(t * 8 + 4 + i) & 0xffff];
    }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 16; ++i) s += acc[i];
    out[t] = s[0] + s[1] + s[2] + s[3];
}

template <int WAVES>
double run(int blocks, int iters, float* out, const float* in) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<WAVES><<<blocks, WAVES * 64>>>(out, in, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<WAVES><<<blocks, WAVES * 64>>>(out, in, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 16 * 2.0 * 16 * 16 * 4;
    return flops / (ms * 1e-3) / 1e12;
}

int main() {
    std::vector<float> h(65536);
    unsigned s = 12345;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v = ((s >> 8) / 8388608.0f) - 1.0f;   // uniform [-1, 1)
    }
    float *in, *out, *zin;
    hipMalloc(&in, h.size() * 4);
    hipMalloc(&zin, h.size() * 4);
    hipMalloc(&out, 256 * 8 * 512 * 4);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(zin, 0, h.size() * 4);
    for (int rep = 0; rep < 2; ++rep) {
        printf("random operands : 4 waves/CU %.1f TF, 8 waves/CU %.1f TF, 16 waves/CU %.1f TF\n", run<4>(256, 200000, out, in),
               run<8>(256, 100000, out, in), run<8>(512, 100000, out, in));
        printf("zero operands   : 4 waves/CU %.1f TF, 8 waves/CU %.1f TF\n", run<4>(256, 200000, out, zin), run<8>(256, 100000, out, zin));
    }
    return 0;
}
