// mfma_shape.hip -- sustained rate and clock of v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 on random data,
// one and two waves per SIMD, 64 accumulator registers per wave either way (16 tiles of 16x16 / 4 tiles of 32x32).
// The split engine is matrix-pipe bound at a power-limited clock; the 32x32 shape reads half the A/B operand registers per flop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_shape.hip -o tools/ubench/bin/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kIters = 20000;

template <int SHAPE>
__global__ __launch_bounds__(512) void k_rate(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* clk) {
    const int lane = threadIdx.x & 63;
    f16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = in[(i * 64 + lane) + 512 * (blockIdx.x & 7)];
        b[i] = in[((4 + i) * 64 + lane) + 512 * (blockIdx.x & 7)];
    }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int it = 0; it < kIters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][3];
    } else {
        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < kIters; ++it) {
            // the same flops per iteration: 2 x 2 tiles of 32x32, K = 32 as two K = 16 steps
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * k + m], b[2 * k + n], acc[m][n], 0, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) sum += acc[m][n][0] + acc[m][n][15];
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = c1 - c0;
        clk[1] = r1 - r0;
    }
}

int main() {
    const int n_in = 8 * 512;
    std::vector<_Float16> h(n_in * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    f16x8* in;
    float* out;
    unsigned long long* clk;
    hipMalloc(&in, h.size() * 2);
    hipMalloc(&out, 1024 * 512 * 4);
    hipMalloc(&clk, 16);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int shape = 16; shape <= 32; shape += 16)
            for (int rep = 0; rep < 3; ++rep) {
                const int grid = 256;   // one workgroup per CU (100 KB of LDS each keeps a second one out)
                hipEventRecord(e0);
                if (shape == 16) hipLaunchKernelGGL(k_rate<16>, dim3(grid), dim3(waves * 64), 100 * 1024, 0, in, out, clk);
                else hipLaunchKernelGGL(k_rate<32>, dim3(grid), dim3(waves * 64), 100 * 1024, 0, in, out, clk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned long long hc[2];
                hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
                const double flops = (double)grid * waves * kIters * 16 * (16.0 * 16 * 32 * 2);
                if (rep == 2)
                    printf("%s  %d waves/SIMD: %7.3f ms  %7.1f TFLOP/s  shader clock %4.0f MHz  cycles per 16x16x32-equivalent per SIMD %.2f\n",
                           shape == 16 ? "16x16x32" : "32x32x16", waves / 4, ms, flops / ms / 1e9, hc[0] * 100.0 / hc[1],
                           (double)hc[0] / ((double)kIters * 16 * (waves / 4)));
            }
    return 0;
}
