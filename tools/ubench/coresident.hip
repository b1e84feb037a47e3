// coresident.hip -- do workgroups of TWO kernels, launched on two HIP streams, share a CU when their registers and LDS fit
// side by side?  (Round 5: the half-CU instances of loop C and of the density pass fit on paper -- 4 waves x 248 VGPRs +
// 84 KB next to 8 waves x 120 VGPRs + 70 KB -- and ran strictly one after the other: tools/probes/cosched_probe.py.)
//
// Kernel A and kernel B spin for a fixed time (s_memrealtime, 100 MHz); every workgroup records where it ran (XCC id, CU
// id from HW_REG_HW_ID) and when.  Per case the program prints the wall time of {A on stream 1 || B on stream 2} next to
// A alone and B alone, and how many of B's workgroups started while an A workgroup was resident on the same CU.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/coresident.hip -o tools/ubench/bin/coresident
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

struct Rec {
    unsigned long long t0, t1;
    unsigned hw, xcc;
};

template <int VGPRS>
__device__ __forceinline__ void claim_vgprs() {
    // make the kernel's allocation at least VGPRS registers per lane
    if constexpr (VGPRS > 200) asm volatile("v_mov_b32 v247, 0" ::: "v247");
    else if constexpr (VGPRS > 100) asm volatile("v_mov_b32 v119, 0" ::: "v119");
    else if constexpr (VGPRS > 60) asm volatile("v_mov_b32 v63, 0" ::: "v63");
}

template <int THREADS, int VGPRS>
__global__ __launch_bounds__(THREADS) void k_spin(Rec* rec, unsigned long long ticks, int touch) {
    extern __shared__ float lds[];
    claim_vgprs<VGPRS>();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (touch >= 0) lds[threadIdx.x + touch] = 1.0f;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        Rec r;
        r.t0 = t0;
        r.t1 = __builtin_amdgcn_s_memrealtime();
        r.hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
        r.xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID
        rec[blockIdx.x] = r;
    }
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                              \
        }                                                                          \
    } while (0)

template <typename KA, typename KB>
int run_case(const char* name, KA ka, int ga, int ta, size_t la, KB kb, int gb, int tb, size_t lb, hipStream_t s1, hipStream_t s2,
             double ms) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ka), hipFuncAttributeMaxDynamicSharedMemorySize, (int)la));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kb), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb));
    Rec *ra, *rb;
    CK(hipMalloc(&ra, sizeof(Rec) * ga));
    CK(hipMalloc(&rb, sizeof(Rec) * gb));
    const unsigned long long ticks = (unsigned long long)(ms * 1e5);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto wall = [&](bool a, bool b) -> double {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (a) hipLaunchKernelGGL(ka, dim3(ga), dim3(ta), la, s1, ra, ticks, 0);
        if (b) hipLaunchKernelGGL(kb, dim3(gb), dim3(tb), lb, s2, rb, ticks, 0);
        hipDeviceSynchronize();
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    wall(true, true);
    const double wa = wall(true, false), wb = wall(false, true), wab = wall(true, true);
    std::vector<Rec> ha(ga), hb(gb);
    CK(hipMemcpy(ha.data(), ra, sizeof(Rec) * ga, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), rb, sizeof(Rec) * gb, hipMemcpyDeviceToHost));
    // B workgroups that STARTED while an A workgroup was resident on the same (xcc, se, cu)
    auto cu_of = [](const Rec& r) { return (r.xcc << 16) | (r.hw & 0x0000ff00u) | ((r.hw >> 13) & 0x7u) << 4; };   // CU_ID bits 8..11, SH 12, SE 13..15
    int together = 0;
    for (const Rec& b : hb)
        for (const Rec& a : ha)
            if (cu_of(a) == cu_of(b) && b.t0 >= a.t0 && b.t0 < a.t1) {
                ++together;
                break;
            }
    std::vector<unsigned> cus;
    for (const Rec& a : ha) cus.push_back(cu_of(a));
    std::sort(cus.begin(), cus.end());
    const int distinct = (int)(std::unique(cus.begin(), cus.end()) - cus.begin());
    printf("%-66s A alone %6.2f  B alone %6.2f  A||B %6.2f ms   B WGs started beside an A WG on their CU: %d / %d   (A's %d WGs on %d CUs)\n",
           name, wa, wb, wab, together, gb, ga, distinct);
    hipFree(ra);
    hipFree(rb);
    return 0;
}

int main() {
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const double ms = 5.0;
    const size_t K = 1024;
    // small everything: the runtime / dispatcher baseline
    run_case("small: A 256x256thr 32 VGPR 1 KB || B 256x512thr 32 VGPR 1 KB", k_spin<256, 32>, 256, 256, 1 * K, k_spin<512, 32>, 256, 512, 1 * K, s1, s2, ms);
    // the half-CU pair of round 5's probe
    run_case("probe pair: A 256x256thr 248 VGPR 84 KB || B 256x512thr 120 VGPR 70 KB", k_spin<256, 248>, 256, 256, 84 * K, k_spin<512, 120>, 256, 512, 70 * K, s1, s2, ms);
    run_case("registers only: A 248 VGPR 1 KB || B 120 VGPR 1 KB", k_spin<256, 248>, 256, 256, 1 * K, k_spin<512, 120>, 256, 512, 1 * K, s1, s2, ms);
    run_case("LDS only: A 32 VGPR 84 KB || B 32 VGPR 70 KB", k_spin<256, 32>, 256, 256, 84 * K, k_spin<512, 32>, 256, 512, 70 * K, s1, s2, ms);
    run_case("LDS only, smaller: A 32 VGPR 64 KB || B 32 VGPR 64 KB", k_spin<256, 32>, 256, 256, 64 * K, k_spin<512, 32>, 256, 512, 64 * K, s1, s2, ms);
    run_case("LDS only: A 32 VGPR 84 KB || B 32 VGPR 40 KB", k_spin<256, 32>, 256, 256, 84 * K, k_spin<512, 32>, 256, 512, 40 * K, s1, s2, ms);
    run_case("probe pair, B grid 512", k_spin<256, 248>, 256, 256, 84 * K, k_spin<512, 120>, 512, 512, 70 * K, s1, s2, ms);
    run_case("full-CU pair (today's kernels): A 256x512thr 248 VGPR 144 KB || B 256x512thr 248 VGPR 137 KB", k_spin<512, 248>, 256, 512, 144 * K, k_spin<512, 248>, 256, 512, 137 * K, s1, s2, ms);
    run_case("one kernel, 512 WGs of the A shape (2 per CU?)  ||  nothing", k_spin<256, 248>, 512, 256, 64 * K, k_spin<512, 32>, 1, 512, 1 * K, s1, s2, ms);
    return 0;
}
