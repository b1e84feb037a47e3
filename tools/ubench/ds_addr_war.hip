// ds_addr_war.hip -- is the ADDRESS register of a DS read safe to overwrite right behind the instruction when the LDS queue
// is deep?  burst of ds_read_b128 (queue filler) ; ds_read_b32 r, v57 ; [s_nop K-1] ; v_mov_b32 v57, other
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/ds_addr_war.hip -o tools/ubench/bin/ds_addr_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FILL "ds_read_b128 v[104:107], %3\n ds_read_b128 v[108:111], %3 offset:1024\n ds_read_b128 v[112:115], %3 offset:2048\n" \
             "ds_read_b128 v[116:119], %3 offset:3072\n ds_read_b128 v[104:107], %3 offset:4096\n ds_read_b128 v[108:111], %3 offset:5120\n" \
             "ds_read_b128 v[112:115], %3 offset:6144\n ds_read_b128 v[116:119], %3 offset:7168\n"
#define SEQ(K)                                                                                                  \
    asm volatile("v_mov_b32 v57, %1\n s_nop 3\n" FILL FILL "ds_read_b32 %0, v57\n" K "v_mov_b32 v57, %2\n s_waitcnt lgkmcnt(0)\n" \
                 : "=&v"(r) : "v"(a0), "v"(a1), "v"(fill)                                                       \
                 : "v57", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "memory")

template <int K>
__global__ void k(float* out, int iters) {
    extern __shared__ int tab[];
    for (int i = threadIdx.x; i < 36864; i += blockDim.x) tab[i] = i * 3 + 1;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned fill = lane * 16u + 65536u;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        const int i0 = (lane * 7 + it) & 1023, i1 = 1024 + ((lane * 5 + it * 3) & 1023);
        const unsigned a0 = i0 * 4, a1 = i1 * 4;
        int r;
        if constexpr (K == 0) SEQ("");
        if constexpr (K == 1) SEQ("s_nop 0\n");
        if constexpr (K == 2) SEQ("s_nop 1\n");
        if constexpr (K == 4) SEQ("s_nop 3\n");
        bad += r != i0 * 3 + 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)bad;
}

template <int K>
void run(float* dOut, int blocks, int threads) {
    const int n = blocks * threads;
    (void)hipMemset(dOut, 0, n * sizeof(float));
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<K>), hipFuncAttributeMaxDynamicSharedMemorySize, 147456);
    hipLaunchKernelGGL((k<K>), dim3(blocks), dim3(threads), 147456, 0, dOut, 4000);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), dOut, n * sizeof(float), hipMemcpyDeviceToHost);
    double bad = 0;
    for (int i = 0; i < n; ++i) bad += h[i];
    printf("overwrite of the address register %d wait states behind the ds_read_b32: %.0f wrong of %.0f\n", K, bad, (double)n * 4000);
}

int main() {
    float* dOut;
    const int blocks = 512, threads = 512;
    (void)hipMalloc(&dOut, blocks * threads * sizeof(float));
    run<0>(dOut, blocks, threads);
    run<1>(dOut, blocks, threads);
    run<2>(dOut, blocks, threads);
    run<4>(dOut, blocks, threads);
    return 0;
}
