// Isolates the GEMM inner loop of mlp.hpp: L layers of 256x256 on 64-point tiles, no real epilogue.
//   variants: 0 = as in the product (global A, LDS B, barriers)   1 = no barriers
//             2 = A fragments from registers (no global loads)     3 = B fragments from registers (no LDS)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../arah_release_amd/csrc/mlp.hpp"
using namespace arah;

template <int VAR, int MT, int NWAVES>
__global__ __launch_bounds__(NWAVES * 64) void k(const float* __restrict__ wp, float* out, int layers, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = smem;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < 64 * kSdfLd; i += NWAVES * 64) act[i] = 0.001f * (i % 977);
    __syncthreads();
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < tiles_per_wg; ++t) {
        for (int l = 0; l < layers; ++l) {
            f32x4 acc[MT][kNT];
            for (int m = 0; m < MT; ++m)
                for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
            if (VAR == 4) {   // A prefetched two chunks ahead
                const int j = lane & 15, g = lane >> 4;
                const float* bptr = act + j * kSdfLd + 4 * g;
                const f32x4* aptr = reinterpret_cast<const f32x4*>(wp + (size_t)(l % 5) * 65536) + (size_t)wave * MT * 16 * 64 + lane;
                f32x4 a0[MT], a1[MT], a2[MT];
                for (int m = 0; m < MT; ++m) { a0[m] = aptr[(m * 16 + 0) * 64]; a1[m] = aptr[(m * 16 + 1) * 64]; }
#pragma unroll 1
                for (int kc = 0; kc < 16; ++kc) {
                    const int kn = kc + 2 < 16 ? kc + 2 : 15;
                    for (int m = 0; m < MT; ++m) a2[m] = aptr[(m * 16 + kn) * 64];
                    f32x4 b[kNT];
                    for (int n = 0; n < kNT; ++n) b[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * kSdfLd + kc * 16);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < kNT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][tt], b[n][tt], acc[m][n], 0, 0, 0);
                    for (int m = 0; m < MT; ++m) { a0[m] = a1[m]; a1[m] = a2[m]; }
                }
            } else if (VAR == 5) {   // each A fragment feeds two point tiles (NT = 8): half the weight traffic per flop
                const int j = lane & 15, g = lane >> 4;
                const float* bptr = act + j * kSdfLd + 4 * g;
                const f32x4* aptr = reinterpret_cast<const f32x4*>(wp + (size_t)(l % 5) * 65536) + (size_t)wave * MT * 16 * 64 + lane;
                f32x4 acc2[MT][kNT];
                for (int m = 0; m < MT; ++m) for (int n = 0; n < kNT; ++n) zero_acc(acc2[m][n]);
                f32x4 a0[MT], a1[MT];
                for (int m = 0; m < MT; ++m) a0[m] = aptr[(m * 16 + 0) * 64];
#pragma unroll 1
                for (int kc = 0; kc < 16; ++kc) {
                    const int kn = kc + 1 < 16 ? kc + 1 : 15;
                    for (int m = 0; m < MT; ++m) a1[m] = aptr[(m * 16 + kn) * 64];
                    f32x4 b[kNT], b2[kNT];
                    for (int n = 0; n < kNT; ++n) { b[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * kSdfLd + kc * 16); b2[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * kSdfLd + ((kc * 16 + 128) & 255)); }
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < kNT; ++n) {
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][tt], b[n][tt], acc[m][n], 0, 0, 0);
                                acc2[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[m][tt], b2[n][tt], acc2[m][n], 0, 0, 0);
                            }
                    for (int m = 0; m < MT; ++m) a0[m] = a1[m];
                }
                for (int m = 0; m < MT; ++m) for (int n = 0; n < kNT; ++n) keep += acc2[m][n];
            } else if (VAR == 2 || VAR == 3) {
                const int j = lane & 15, g = lane >> 4;
                const float* bptr = act + j * kSdfLd + 4 * g;
                const f32x4* aptr = reinterpret_cast<const f32x4*>(wp) + (size_t)wave * MT * 16 * 64 + lane;
                f32x4 a_reg[MT], b_reg[kNT];
                for (int m = 0; m < MT; ++m) a_reg[m] = aptr[m * 16 * 64];
                for (int n = 0; n < kNT; ++n) b_reg[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * kSdfLd);
#pragma unroll 1
                for (int kc = 0; kc < 16; ++kc) {
                    f32x4 a[MT], b[kNT];
                    for (int m = 0; m < MT; ++m) a[m] = (VAR == 2) ? a_reg[m] : aptr[(m * 16 + kc) * 64];
                    for (int n = 0; n < kNT; ++n) b[n] = (VAR == 3) ? b_reg[n] : *reinterpret_cast<const f32x4*>(bptr + n * 16 * kSdfLd + kc * 16);
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int n = 0; n < kNT; ++n)
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][tt], b[n][tt], acc[m][n], 0, 0, 0);
                }
            } else {
                gemm_acc<16, MT>(wp + (size_t)(l % 5) * 65536, wave * MT, act, kSdfLd, acc, lane);
            }
            if (VAR != 1) __syncthreads();
            for (int m = 0; m < MT; ++m)
                for (int n = 0; n < kNT; ++n) keep += acc[m][n];
            if (VAR != 1) __syncthreads();
        }
    }
    out[blockIdx.x * NWAVES * 64 + tid] = keep[0] + keep[1] + keep[2] + keep[3];
}

template <int VAR, int MT, int NWAVES>
void run(const char* name, const float* wp, float* out, int wgs) {
    const int layers = 5, tiles = 40;
    const size_t lds = (size_t)64 * kSdfLd * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<VAR, MT, NWAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<VAR, MT, NWAVES><<<wgs, NWAVES * 64, lds>>>(wp, out, layers, tiles);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    k<VAR, MT, NWAVES><<<wgs, NWAVES * 64, lds>>>(wp, out, layers, tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)wgs * tiles * layers * 64.0 * 256 * 256 * 2 * (VAR == 5 ? 2 : 1);
    printf("%-44s WGs %4d  %7.2f ms  %6.1f TF\n", name, wgs, ms, flops / (ms * 1e-3) / 1e12);
}

int main() {
    std::vector<float> h(5 * 65536);
    unsigned s = 1;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 8388608.0f - 1.0f) * 0.01f; }
    float *wp, *out;
    hipMalloc(&wp, h.size() * 4);
    hipMalloc(&out, 1024 * 512 * 4);
    hipMemcpy(wp, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wgs : {256, 512}) {
        run<0, 2, 8>("product loop (8 waves, MT=2)", wp, out, wgs);
        run<1, 2, 8>("  no barriers", wp, out, wgs);
        run<2, 2, 8>("  A from registers (no global loads)", wp, out, wgs);
        run<3, 2, 8>("  B from registers (no LDS reads)", wp, out, wgs);
        run<4, 2, 8>("  A prefetched 2 chunks ahead", wp, out, wgs);
        run<5, 2, 8>("  NT=8 (A fragment feeds 128 points)", wp, out, wgs);
        run<0, 4, 4>("4 waves, MT=4", wp, out, wgs);
        run<2, 4, 4>("  4 waves, MT=4, A from registers", wp, out, wgs);
    }
    return 0;
}
