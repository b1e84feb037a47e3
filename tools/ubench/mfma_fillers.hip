// mfma_fillers.hip -- how many independent vector instructions hide behind an MFMA?  The issue model behind DESIGN.md
// section 10: a stream of [one MFMA, F independent v_fma_f32] per step, for F = 0 .. 8, the two MFMA shapes of the split
// engine (v_mfma_f32_16x16x32_f16: 16 cycles of matrix pipe; v_mfma_f32_32x32x16_f16: 32) and one or two waves per SIMD
// (256 / 512 threads, one workgroup per CU).  Printed: s_memtime ticks per MFMA (per SIMD: wall ticks of a wave / MFMAs
// issued by ALL waves of its SIMD in that time), i.e. what a filler costs once the shadow is full.
// The order inside a step is pinned (sched_barrier(0) after every instruction group); operands are random.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_fillers.hip -o tools/ubench/bin/mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kIters = 4000;   // x 8 steps

// SPLIT: the fillers of a step are issued by the OTHER wave of the SIMD only (waves 4-7 run vector instructions, waves 0-3
// MFMAs): what a partner wave's vector work costs / hides
template <int SHAPE, int F, bool SPLIT>
__global__ __launch_bounds__(512) void k_fill(const f16x8* __restrict__ in, float* __restrict__ out, unsigned long long* clk) {
    extern __shared__ char pad[];   // 100 KB requested: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = in[(i * 64 + lane) + 512 * (blockIdx.x & 7)];
        b[i] = in[((4 + i) * 64 + lane) + 512 * (blockIdx.x & 7)];
    }
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[0][i];
    const float ka = 1.0000001f, kb = 1e-9f;
    f32x4 acc4[4];
    f32x16 acc16[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc16[i][r] = 0.f;
    const bool do_mfma = !SPLIT || wave < 4, do_fill = !SPLIT || wave >= 4;
    volatile int* done = reinterpret_cast<volatile int*>(pad);   // SPLIT: the filler waves run until the four MFMA waves are through
    if (threadIdx.x == 0) *done = 0;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; SPLIT && !do_mfma ? true : it < kIters; ++it) {
        if (SPLIT && !do_mfma && (it & 7) == 7 && *done >= 4) break;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (do_mfma) {
                if constexpr (SHAPE == 16) acc4[s & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s & 1], b[(s >> 1) & 1], acc4[s & 3], 0, 0, 0);
                else acc16[s & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[s & 1], b[(s >> 1) & 1], acc16[s & 1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (do_fill) {
#pragma unroll
                for (int f = 0; f < F; ++f) v[f] = __builtin_fmaf(v[f], ka, kb);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (SPLIT && do_mfma && lane == 0) atomicAdd(const_cast<int*>(done), 1);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += acc4[i][0] + acc4[i][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) sum += acc16[i][0] + acc16[i][15];
#pragma unroll
    for (int i = 0; i < 8; ++i) sum += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (lane == 0 && do_mfma) atomicAdd(clk, c1 - c0);
}

static double g_ns = 0.0;   // wall ns per MFMA of a SIMD of the last run (hipEvents)
template <int SHAPE, int F, bool SPLIT>
double run(int threads, const f16x8* d_in, float* d_out, unsigned long long* d_clk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)k_fill<SHAPE, F, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    unsigned long long zero = 0, ticks = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemcpy(d_clk, &zero, 8, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_fill<SHAPE, F, SPLIT>), dim3(256), dim3(threads), 100 * 1024, 0, d_in, d_out, d_clk);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&ticks, d_clk, 8, hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    const double per_wave = (double)ticks / (256.0 * (SPLIT ? 4 : waves));  // wall ticks of a wave that issues MFMAs
    const int mfma_waves_per_simd = SPLIT ? 1 : waves / 4;                  // waves of a SIMD that issue MFMAs
    g_ns = ms * 1e6 / ((double)kIters * 8 * mfma_waves_per_simd);
    return per_wave / ((double)kIters * 8 * mfma_waves_per_simd);          // ticks per MFMA of the SIMD
}

template <int SHAPE, bool SPLIT, int... Fs>
void sweep(const char* name, int threads, const f16x8* d_in, float* d_out, unsigned long long* d_clk) {
    printf("%-44s", name);
    double ns[sizeof...(Fs)];
    int i = 0;
    ((printf(" %6.1f", run<SHAPE, Fs, SPLIT>(threads, d_in, d_out, d_clk)), ns[i++] = g_ns), ...);
    printf("   ticks\n%-44s", "");
    for (int k = 0; k < i; ++k) printf(" %6.2f", ns[k]);
    printf("   ns\n");
}

int main() {
    std::vector<_Float16> h(8 * 512 * 8);
    srand(1);
    for (auto& x : h) x = (_Float16)((rand() / (float)RAND_MAX) * 2.f - 1.f);
    f16x8* d_in;
    float* d_out;
    unsigned long long* d_clk;
    hipMalloc(&d_in, h.size() * 2);
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_clk, 8);
    hipMemcpy(d_in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    printf("s_memtime ticks per MFMA of a SIMD; columns: F = 0 1 2 3 4 5 6 8 independent v_fma_f32 per MFMA step; second line: wall ns\n");
    sweep<16, false, 0, 1, 2, 3, 4, 5, 6, 8>("16x16x32, one wave per SIMD", 256, d_in, d_out, d_clk);
    sweep<16, false, 0, 1, 2, 3, 4, 5, 6, 8>("16x16x32, two waves per SIMD", 512, d_in, d_out, d_clk);
    sweep<32, false, 0, 1, 2, 3, 4, 5, 6, 8>("32x32x16, one wave per SIMD", 256, d_in, d_out, d_clk);
    sweep<32, false, 0, 1, 2, 3, 4, 5, 6, 8>("32x32x16, two waves per SIMD", 512, d_in, d_out, d_clk);
    printf("(a 32x32x16 step is the flops of two 16x16x32 steps: halve its figures to compare)\n");
    return 0;
}
