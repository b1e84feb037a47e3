// mfma_read_hazard.hip -- how many wait states does gfx950 need between v_mfma_f32_16x16x32_f16 and the first VALU read of
// its result, and does a TRANSCENDENTAL reader (v_exp_f32) need more than a plain one (v_add_f32 / v_med3_f32)?
// Round 4 of k_canon_wave reads accumulators with v_exp_f32 ... clamp / v_med3_f32 straight off the matrix pipe; some
// schedules of that kernel gave wrong roots for a fixed set of points (profiles/r04_schedule_dependent_roots.txt).
// The sequence is one asm block on fixed registers, so the distance is exactly K: mfma ; s_nop K-1 ; reader.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_read_hazard.hip -o tools/ubench/bin/mfma_read_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SEQ(K, READER)                                                                                       \
    asm volatile("v_mov_b32 v100, %2\n v_mov_b32 v101, %2\n v_mov_b32 v102, %2\n v_mov_b32 v103, %2\n"      \
                 "v_mov_b32 v104, 0\n s_nop 7\n s_nop 7\n"                                                   \
                 "v_mfma_f32_16x16x32_f16 v[100:103], %3, %4, v[100:103]\n" K READER "\n s_nop 7\n s_nop 7\n" \
                 "v_mov_b32 %0, v104\n v_mov_b32 %1, v100\n"                                                 \
                 : "=v"(r), "=v"(full)                                                                       \
                 : "v"(c0), "v"(a), "v"(b)                                                                   \
                 : "v100", "v101", "v102", "v103", "v104")

template <int K, int R>
__global__ void k(const f16x8* A, const f16x8* B, float* out, float c0, int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 a = A[lane], b = B[lane];
    int bad = 0;
    float first_bad = 0.f;
    for (int it = 0; it < iters; ++it) {
        float r, full;
        if constexpr (R == 0) {   // plain reader: v104 = v100 + 0
            if constexpr (K == 0) SEQ("", "v_add_f32 v104, 0, v100");
            if constexpr (K == 1) SEQ("s_nop 0\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 2) SEQ("s_nop 1\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 3) SEQ("s_nop 2\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 4) SEQ("s_nop 3\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 5) SEQ("s_nop 4\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 6) SEQ("s_nop 5\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 7) SEQ("s_nop 6\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 8) SEQ("s_nop 7\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 9) SEQ("s_nop 7\n s_nop 0\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 10) SEQ("s_nop 7\n s_nop 1\n", "v_add_f32 v104, 0, v100");
            if constexpr (K == 12) SEQ("s_nop 7\n s_nop 3\n", "v_add_f32 v104, 0, v100");
        } else if constexpr (R == 1) {   // transcendental reader; the host compares log2 of it
            if constexpr (K == 0) SEQ("", "v_exp_f32 v104, v100");
            if constexpr (K == 1) SEQ("s_nop 0\n", "v_exp_f32 v104, v100");
            if constexpr (K == 2) SEQ("s_nop 1\n", "v_exp_f32 v104, v100");
            if constexpr (K == 3) SEQ("s_nop 2\n", "v_exp_f32 v104, v100");
            if constexpr (K == 4) SEQ("s_nop 3\n", "v_exp_f32 v104, v100");
            if constexpr (K == 5) SEQ("s_nop 4\n", "v_exp_f32 v104, v100");
            if constexpr (K == 6) SEQ("s_nop 5\n", "v_exp_f32 v104, v100");
            if constexpr (K == 7) SEQ("s_nop 6\n", "v_exp_f32 v104, v100");
            if constexpr (K == 8) SEQ("s_nop 7\n", "v_exp_f32 v104, v100");
            if constexpr (K == 9) SEQ("s_nop 7\n s_nop 0\n", "v_exp_f32 v104, v100");
            if constexpr (K == 10) SEQ("s_nop 7\n s_nop 1\n", "v_exp_f32 v104, v100");
            if constexpr (K == 12) SEQ("s_nop 7\n s_nop 3\n", "v_exp_f32 v104, v100");
        }
        const float expect = R == 0 ? full : __builtin_amdgcn_exp2f(full);
        if (r != expect) {
            if (!bad) first_bad = r;
            ++bad;
        }
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2] = (float)bad;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 2 + 1] = first_bad;
}

template <int K, int R>
void run(const f16x8* dA, const f16x8* dB, float* dOut, int blocks, int threads) {
    const int n = blocks * threads;
    hipMemset(dOut, 0, n * 2 * sizeof(float));
    hipLaunchKernelGGL((k<K, R>), dim3(blocks), dim3(threads), 0, 0, dA, dB, dOut, 0.5f, 2000);
    hipDeviceSynchronize();
    std::vector<float> h(n * 2);
    hipMemcpy(h.data(), dOut, n * 2 * sizeof(float), hipMemcpyDeviceToHost);
    double bad = 0;
    float fb = 0;
    for (int i = 0; i < n; ++i) {
        bad += h[2 * i];
        if (h[2 * i] > 0 && fb == 0) fb = h[2 * i + 1];
    }
    printf("reader %-9s distance %2d : %10.0f wrong of %.0f  (a wrong value: %g)\n", R == 0 ? "v_add_f32" : "v_exp_f32", K, bad, (double)n * 2000, fb);
}

int main() {
    std::vector<_Float16> hA(64 * 8), hB(64 * 8);
    for (int i = 0; i < 64 * 8; ++i) {
        hA[i] = (_Float16)(0.01f * (float)((i * 7) % 13 - 6));
        hB[i] = (_Float16)(0.02f * (float)((i * 5) % 11 - 5));
    }
    f16x8 *dA, *dB;
    float* dOut;
    const int blocks = 512, threads = 512;
    hipMalloc(&dA, 64 * 16);
    hipMalloc(&dB, 64 * 16);
    hipMalloc(&dOut, blocks * threads * 2 * sizeof(float));
    hipMemcpy(dA, hA.data(), 64 * 16, hipMemcpyHostToDevice);
    hipMemcpy(dB, hB.data(), 64 * 16, hipMemcpyHostToDevice);
#define BOTH(K) run<K, 0>(dA, dB, dOut, blocks, threads); run<K, 1>(dA, dB, dOut, blocks, threads);
    BOTH(0) BOTH(1) BOTH(2) BOTH(3) BOTH(4) BOTH(5) BOTH(6) BOTH(7) BOTH(8) BOTH(9) BOTH(10) BOTH(12)
    return 0;
}
