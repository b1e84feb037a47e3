// valu_mfma32_hazard.hip -- the same question as valu_mfma_hazard.hip for v_mfma_f32_16x16x4_f32 (the fp32 MFMA loop C's tail
// gathers and blends with): VALU write of its B operand (v_mov_b32, or v_cndmask_b32_e64 on an SGPR-pair mask as in the T blend),
// then K wait states (s_nop) or only an s_waitcnt, then the MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/valu_mfma32_hazard.hip -o tools/ubench/bin/valu_mfma32_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SEQ(WRITER, K)                                                                                        \
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n"             \
                 "v_mov_b32 v113, %2\n v_mov_b32 v114, %1\n s_nop 7\n s_nop 7\n" WRITER "\n" K                 \
                 "v_mfma_f32_16x16x4_f32 v[100:103], %3, v113, v[100:103]\n s_nop 7\n s_nop 7\n s_nop 7\n"      \
                 "v_mov_b32 %0, v100\n"                                                                        \
                 : "=v"(r) : "v"(b), "v"(decoy), "v"(a), "s"(mask)                                             \
                 : "v100", "v101", "v102", "v103", "v113", "v114")

template <int W, int K>
__global__ void k(const float* A, const float* B, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const float a = A[lane], b = B[lane], decoy = 1000.0f;
    const unsigned long long mask = ~0ull;
    f32x4 e = {0.f, 0.f, 0.f, 0.f};
    e = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, e, 0, 0, 0);
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        float r;
        if constexpr (W == 0) {
            if constexpr (K == 0) SEQ("v_mov_b32 v113, v114", "");
            if constexpr (K == 1) SEQ("v_mov_b32 v113, v114", "s_nop 0\n");
            if constexpr (K == 2) SEQ("v_mov_b32 v113, v114", "s_nop 1\n");
            if constexpr (K == 3) SEQ("v_mov_b32 v113, v114", "s_nop 2\n");
            if constexpr (K == 4) SEQ("v_mov_b32 v113, v114", "s_nop 3\n");
            if constexpr (K == 100) SEQ("v_mov_b32 v113, v114", "s_waitcnt lgkmcnt(2)\n");
        } else {   // v113 = mask ? v114 : v113
            if constexpr (K == 0) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "");
            if constexpr (K == 1) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "s_nop 0\n");
            if constexpr (K == 2) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "s_nop 1\n");
            if constexpr (K == 3) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "s_nop 2\n");
            if constexpr (K == 4) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "s_nop 3\n");
            if constexpr (K == 100) SEQ("v_cndmask_b32_e64 v113, v113, v114, %4", "s_waitcnt lgkmcnt(2)\n");
        }
        if (r != e[0]) ++bad;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)bad;
}

template <int W, int K>
void run(const float* dA, const float* dB, float* dOut, int blocks, int threads) {
    const int n = blocks * threads;
    (void)hipMemset(dOut, 0, n * sizeof(float));
    hipLaunchKernelGGL((k<W, K>), dim3(blocks), dim3(threads), 0, 0, dA, dB, dOut, 2000);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), dOut, n * sizeof(float), hipMemcpyDeviceToHost);
    double bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i];
    if (K >= 100) printf("writer %-18s then s_waitcnt lgkmcnt(2) only : %10.0f wrong of %.0f\n", W ? "v_cndmask_b32_e64" : "v_mov_b32", bad, (double)n * 2000);
    else printf("writer %-18s distance %d : %10.0f wrong of %.0f\n", W ? "v_cndmask_b32_e64" : "v_mov_b32", K, bad, (double)n * 2000);
}

int main() {
    std::vector<float> hA(64), hB(64);
    for (int i = 0; i < 64; ++i) {
        hA[i] = 0.25f + 0.01f * (float)((i * 7) % 13);
        hB[i] = 0.5f + 0.02f * (float)((i * 5) % 11);
    }
    float *dA, *dB, *dOut;
    const int blocks = 512, threads = 512;
    (void)hipMalloc(&dA, 256);
    (void)hipMalloc(&dB, 256);
    (void)hipMalloc(&dOut, blocks * threads * sizeof(float));
    (void)hipMemcpy(dA, hA.data(), 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(dB, hB.data(), 256, hipMemcpyHostToDevice);
#define ALLK(W) run<W, 0>(dA, dB, dOut, blocks, threads); run<W, 1>(dA, dB, dOut, blocks, threads); run<W, 2>(dA, dB, dOut, blocks, threads); run<W, 3>(dA, dB, dOut, blocks, threads); run<W, 4>(dA, dB, dOut, blocks, threads); run<W, 100>(dA, dB, dOut, blocks, threads);
    ALLK(0) ALLK(1)
    return 0;
}
