// scratch_store_war.hip -- does gfx950 read the data register of a scratch_store_dword (and of a global_store_dword) at
// issue?  store v100 ; s_nop K-1 ; v_mov_b32 v100, other ; ... ; load back.  Round 4: builds of k_canon_wave whose register
// allocation spilled loop invariants in the prologue gave wrong roots for the upper body; their spill code is
// "scratch_store_dword off, v9, off ; v_add_u32 v9, ..." back to back, and the reloaded values address the bone table.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/scratch_store_war.hip -o tools/ubench/bin/scratch_store_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SEQ_S(K)                                                                                                  \
    asm volatile("v_mov_b32 v100, %1\n s_nop 7\n scratch_store_dword off, v100, off offset:128\n" K              \
                 "v_mov_b32 v100, %2\n s_waitcnt vmcnt(0)\n s_nop 7\n scratch_load_dword %0, off, off offset:128\n s_waitcnt vmcnt(0)\n" \
                 : "=v"(r) : "v"(val), "v"(other) : "v100", "memory")
#define SEQ_G(K)                                                                                                  \
    asm volatile("v_mov_b32 v100, %1\n s_nop 7\n global_store_dword %3, v100, off\n" K                          \
                 "v_mov_b32 v100, %2\n s_waitcnt vmcnt(0)\n s_nop 7\n global_load_dword %0, %3, off sc0 sc1\n s_waitcnt vmcnt(0)\n" \
                 : "=v"(r) : "v"(val), "v"(other), "v"(gp) : "v100", "memory")

template <int G, int K>
__global__ void k(float* out, int* gbuf, int iters) {
    volatile int priv[64];   // forces a private segment of >= 256 bytes
    for (int i = 0; i < 64; ++i) priv[i] = i + threadIdx.x;
    int* gp = gbuf + blockIdx.x * blockDim.x + threadIdx.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int val = it * 7 + threadIdx.x, other = -1 - it;
        int r;
        if constexpr (G == 0) {
            if constexpr (K == 0) SEQ_S("");
            if constexpr (K == 1) SEQ_S("s_nop 0\n");
            if constexpr (K == 2) SEQ_S("s_nop 1\n");
            if constexpr (K == 4) SEQ_S("s_nop 3\n");
            if constexpr (K == 8) SEQ_S("s_nop 7\n");
        } else {
            if constexpr (K == 0) SEQ_G("");
            if constexpr (K == 1) SEQ_G("s_nop 0\n");
            if constexpr (K == 2) SEQ_G("s_nop 1\n");
            if constexpr (K == 4) SEQ_G("s_nop 3\n");
            if constexpr (K == 8) SEQ_G("s_nop 7\n");
        }
        if (r != val) ++bad;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)bad + (float)(priv[threadIdx.x & 63] & 0);
}

template <int G, int K>
void run(float* dOut, int* gbuf, int blocks, int threads) {
    const int n = blocks * threads;
    (void)hipMemset(dOut, 0, n * sizeof(float));
    hipLaunchKernelGGL((k<G, K>), dim3(blocks), dim3(threads), 0, 0, dOut, gbuf, 2000);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), dOut, n * sizeof(float), hipMemcpyDeviceToHost);
    double bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i];
    printf("%-20s distance %d : %10.0f wrong of %.0f\n", G ? "global_store_dword" : "scratch_store_dword", K, bad, (double)n * 2000);
}

int main() {
    float* dOut;
    int* gbuf;
    const int blocks = 1024, threads = 512;
    (void)hipMalloc(&dOut, blocks * threads * sizeof(float));
    (void)hipMalloc(&gbuf, blocks * threads * sizeof(int));
#define ALLK(G) run<G, 0>(dOut, gbuf, blocks, threads); run<G, 1>(dOut, gbuf, blocks, threads); run<G, 2>(dOut, gbuf, blocks, threads); run<G, 4>(dOut, gbuf, blocks, threads); run<G, 8>(dOut, gbuf, blocks, threads);
    ALLK(0) ALLK(1)
    return 0;
}
