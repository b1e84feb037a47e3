// mfma_valu_overlap.hip -- does VALU work hide behind f16 MFMAs on gfx950, and under which issue pattern?
//
// The split-engine kernels of the product measure as (matrix-pipe time) + (vector-ALU time).  This program isolates
// the question on synthetic streams with the product's instruction mix:
//   M : per "phase" 24 x v_mfma_f32_16x16x32_f16 on 8 independent accumulators (one GEMM chunk of the SDF trunk)
//   V : per phase NV Softplus-like chains on independent registers (mul, exp2, add, log2, fma, max, min, cvt)
// variants   0: M only      1: V only      2: M then V in separate blocks (what the product does, per wave)
//            3: M and V in one basic block, compiler's own schedule
//            4: as 3 with sched_group_barrier 1 MFMA : K VALU
// run with 1, 2 and 4 waves per SIMD (256 / 512 / 1024-thread workgroups, one per CU).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_valu_overlap.hip -o tools/ubench/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kPhases = 2000;

__device__ __forceinline__ float chain(float x, float c1, float c2) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * c1);
    float h = fmaf(__builtin_amdgcn_logf(1.0f + e), c2, fmaxf(x, 0.f));
    h = fminf(h, 65504.0f);
    const _Float16 hi = (_Float16)h;
    return h - (float)hi;
}

template <int VARIANT, int NV, int K>
__global__ __launch_bounds__(1024) void k_mix(const f16x8* __restrict__ a_in, float* __restrict__ out, float c1, float c2) {
    const int lane = threadIdx.x & 63;
    f16x8 a[2], b[4];
    a[0] = a_in[lane];
    a[1] = a_in[64 + lane];
    for (int n = 0; n < 4; ++n) b[n] = a_in[128 + n * 64 + lane];
    f32x4 acc[2][4];
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[NV];
    for (int i = 0; i < NV; ++i) v[i] = 0.01f * (float)(lane + i);
#pragma unroll 1
    for (int p = 0; p < kPhases; ++p) {
        if (VARIANT == 0 || VARIANT == 2) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        if (VARIANT == 2) __builtin_amdgcn_sched_barrier(0);
        if (VARIANT == 1 || VARIANT == 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = chain(v[i], c1, c2) + 0.25f;
        }
        if (VARIANT == 3 || VARIANT == 4) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i] = chain(v[i], c1, c2) + 0.25f;
            if (VARIANT == 4) {
#pragma unroll
                for (int q = 0; q < 24; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x2, K, 0);   // K VALU
                }
            }
        }
        // keep the loop body one phase (no cross-iteration motion)
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m)
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    for (int i = 0; i < NV; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int VARIANT, int NV, int K>
static float run(int threads, const f16x8* a, float* out) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_mix<VARIANT, NV, K>), dim3(256), dim3(threads), 0, 0, a, out, 1.3f, 0.7f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_mix<VARIANT, NV, K>), dim3(256), dim3(threads), 0, 0, a, out, 1.3f, 0.7f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}

template <int NV>
static void sweep(const f16x8* a, float* out) {
    printf("NV = %d VALU chains per phase (%d VALU instructions), 24 MFMAs per phase, %d phases\n", NV, NV * 11, kPhases);
    printf("%-10s %10s %10s %10s %12s %12s %12s %12s\n", "waves/SIMD", "M only", "V only", "M then V", "one block", "1:2", "1:3",
           "1:4");
    for (int threads = 256; threads <= 1024; threads *= 2) {
        const float t0 = run<0, NV, 0>(threads, a, out), t1 = run<1, NV, 0>(threads, a, out), t2 = run<2, NV, 0>(threads, a, out),
                    t3 = run<3, NV, 0>(threads, a, out), t42 = run<4, NV, 2>(threads, a, out), t43 = run<4, NV, 3>(threads, a, out),
                    t44 = run<4, NV, 4>(threads, a, out);
        printf("%-10d %10.3f %10.3f %10.3f %12.3f %12.3f %12.3f %12.3f   ms (sum %.3f, max %.3f)\n", threads / 256, t0, t1, t2, t3,
               t42, t43, t44, t0 + t1, t0 > t1 ? t0 : t1);
    }
}

int main() {
    f16x8* a;
    float* out;
    CHECK(hipMalloc(&a, 64 * 6 * sizeof(f16x8)));
    CHECK(hipMalloc(&out, 256 * 1024 * sizeof(float)));
    f16x8 h[64 * 6];
    for (int i = 0; i < 64 * 6; ++i)
        for (int e = 0; e < 8; ++e) h[i][e] = (_Float16)(0.001f * (float)((i * 8 + e) % 97 - 48));
    CHECK(hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice));
    sweep<4>(a, out);
    sweep<8>(a, out);
    return 0;
}
