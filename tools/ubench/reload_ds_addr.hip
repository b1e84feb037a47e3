// reload_ds_addr.hip -- the spill-reload pattern of the failing k_canon_wave builds, in isolation:
//     scratch_load_dword v57 <- A0 ; s_waitcnt vmcnt(0) ; ds_read_b32 r0, v57 ; scratch_load_dword v57 <- A1 ; s_waitcnt vmcnt(0) ;
//     ds_read_b32 r1, v57 ; scratch_load_dword v57 <- A2 ; s_waitcnt vmcnt(0) ; ds_read_b32 r2, v57
// Does every DS read use the address its own reload delivered?  (The LDS unit is kept busy by the other waves of the block.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/reload_ds_addr.hip -o tools/ubench/bin/reload_ds_addr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int GAP>
__global__ void k(float* out, int iters) {
    __shared__ int tab[4096];
    volatile int priv[64];
    for (int i = 0; i < 64; ++i) priv[i] = i;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) tab[i] = i * 3 + 1;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int bad[3] = {0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int i0 = (lane * 7 + it) & 1023, i1 = 1024 + ((lane * 5 + it * 3) & 1023), i2 = 2048 + ((lane * 3 + it * 5) & 1023);
        const unsigned a0 = i0 * 4, a1 = i1 * 4, a2 = i2 * 4;
        int r0, r1, r2;
        if constexpr (GAP == 0)
            asm volatile("scratch_store_dword off, %3, off offset:192\n scratch_store_dword off, %4, off offset:196\n"
                         "scratch_store_dword off, %5, off offset:200\n s_waitcnt vmcnt(0)\n"
                         "ds_read_b32 v101, %6\n ds_read_b32 v102, %6 offset:64\n ds_read_b32 v103, %6 offset:128\n ds_read_b32 v104, %6 offset:256\n"
                         "scratch_load_dword v57, off, off offset:192\n s_waitcnt vmcnt(0)\n ds_read_b32 %0, v57\n"
                         "scratch_load_dword v57, off, off offset:196\n s_waitcnt vmcnt(0)\n ds_read_b32 %1, v57\n"
                         "scratch_load_dword v57, off, off offset:200\n s_waitcnt vmcnt(0)\n ds_read_b32 %2, v57\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a0)
                         : "v57", "v101", "v102", "v103", "v104", "memory");
        else
            asm volatile("scratch_store_dword off, %3, off offset:192\n scratch_store_dword off, %4, off offset:196\n"
                         "scratch_store_dword off, %5, off offset:200\n s_waitcnt vmcnt(0)\n"
                         "ds_read_b32 v101, %6\n ds_read_b32 v102, %6 offset:64\n ds_read_b32 v103, %6 offset:128\n ds_read_b32 v104, %6 offset:256\n"
                         "scratch_load_dword v57, off, off offset:192\n s_waitcnt vmcnt(0)\n s_nop 1\n ds_read_b32 %0, v57\n"
                         "s_nop 1\n scratch_load_dword v57, off, off offset:196\n s_waitcnt vmcnt(0)\n s_nop 1\n ds_read_b32 %1, v57\n"
                         "s_nop 1\n scratch_load_dword v57, off, off offset:200\n s_waitcnt vmcnt(0)\n s_nop 1\n ds_read_b32 %2, v57\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                         : "v"(a0), "v"(a1), "v"(a2), "v"(a0)
                         : "v57", "v101", "v102", "v103", "v104", "memory");
        bad[0] += r0 != i0 * 3 + 1;
        bad[1] += r1 != i1 * 3 + 1;
        bad[2] += r2 != i2 * 3 + 1;
    }
    float* o = out + (blockIdx.x * blockDim.x + threadIdx.x) * 3;
    o[0] = (float)bad[0] + (float)(priv[lane] & 0);
    o[1] = (float)bad[1];
    o[2] = (float)bad[2];
}

template <int GAP>
void run(float* dOut, int blocks, int threads) {
    const int n = blocks * threads * 3;
    (void)hipMemset(dOut, 0, n * sizeof(float));
    hipLaunchKernelGGL((k<GAP>), dim3(blocks), dim3(threads), 0, 0, dOut, 4000);
    (void)hipDeviceSynchronize();
    std::vector<float> h(n);
    (void)hipMemcpy(h.data(), dOut, n * sizeof(float), hipMemcpyDeviceToHost);
    double bad[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) bad[i % 3] += h[i];
    printf("gap %d: wrong reads  first %.0f  second %.0f  third %.0f  of %.0f each\n", GAP, bad[0], bad[1], bad[2], (double)blocks * threads * 4000);
}

int main() {
    float* dOut;
    const int blocks = 1024, threads = 512;
    (void)hipMalloc(&dOut, blocks * threads * 3 * sizeof(float));
    run<0>(dOut, blocks, threads);
    run<1>(dOut, blocks, threads);
    run<0>(dOut, blocks, threads);
    return 0;
}
