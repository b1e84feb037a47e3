// Reproducer and bisection of an MI355X (gfx950, ROCm 7.2 hipcc) correctness hazard found while bringing up the
// split-precision SDF trunk: with TWO workgroups of the kernel resident on a CU, whole 16-point groups of a tile
// came out wrong, irreproducibly (different tiles every run, errors of order 1), while the same binary was
// bit-reproducible with one workgroup per CU.
//
// What the switches below established (every line is a measured run; `dbg flags N` kernels are local copies of the
// trunk with one thing changed, all launched 2 workgroups / CU and re-run 4x with a bitwise compare):
//   * not the GEMM loop: fencing all operand loads ahead of the MFMAs and draining the matrix pipe per chunk
//     changes nothing; the isolated layer chain (gemm_f16x3.hip) is reproducible at 2 workgroups / CU;
//   * not barriers (doubled / fenced / s_sleep variants still fail), not the fused ds_write2st64_b64 store, not
//     s_waitcnt after stores, not the LDS base offset;
//   * not "four f16-MFMA waves per SIMD": 16 waves of ONE workgroup are bit-exact, even with the two halves skewed by
//     a phase (k_trunk_16w, k_trunk_16w_skew);
//   * it is the FIRST layer (K = 3, vector ALU): with its FiLM sine replaced by a clamp the failures vanish (flag 1),
//     with the sine kept there and removed everywhere else they stay (flag 2);
//   * and within that layer it is code generation, not data: the identical math with each value pinned in its own
//     VGPR (flag 64 / 512: `asm volatile("" : "+v"(x))`) or followed by a no-op clamp (flag 128) is reproducible.
//     The failing builds compile the unrolled 3-term dot products into v_pk_fma_f32 with op_sel / op_sel_hi
//     half-broadcasts of the coordinate operand (112 v_pk_fma_f32 in the layer); the passing builds into scalar
//     v_fmac_f32 / v_fmaak_f32.  Fencing ONLY the dot products (flag 512; the sine stays packed) is enough.
// Product fix: `no_pack()` in csrc/mlp.hpp after the K = 3 chains.  The split engine then runs two workgroups per
// CU and is bit-reproducible (tests/test_hip_parity.py::test_render_is_reproducible_under_load).
//
// Also kept here, all bit-exact against the reference and none of them faster than two plain workgroups per CU
// (0.76 ms for 400 k points): s_memtime phase clocks of the trunk (k_trunk_clk); a two-tile variant whose SIMD-mate
// waves run GEMM and epilogue in opposite order (k_pair_clk, 0.90-1.0 ms); 16 waves in one workgroup, plain and skewed
// by a phase (0.78 / 0.84 ms); and a two-tile variant with the epilogue of one tile interleaved INSIDE the GEMM loop
// of the other in the same wave, with and without a 1 MFMA : 3 VALU sched_group_barrier pattern (k_pair_fused,
// 0.89 ms, 256 VGPRs + spills).  In every arrangement kernel time ~ matrix-pipe time + vector-ALU time.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../arah_release_amd/csrc/mlp.hpp"
using namespace arah;

template <bool SPLIT>
__global__ __launch_bounds__(kThreads, 4) void k_trunk(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;
    float* outv = xin + 64 * 4;
    float* act = outv + 64 * 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][kNT];
        sdf_trunk<false, kNT, SPLIT>(net, xin, act, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<SPLIT>(net, act, kSdfLd, outv, 4, tid);
        __syncthreads();
        if (tid < kTile && tile * kTile + tid < n) out[tile * kTile + tid] = outv[tid * 4];
        __syncthreads();
    }
}

__global__ void k_pack32(float* __restrict__ dst, const float* __restrict__ src) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16 * 16 * 64) return;
    const int lane = idx & 63, tile = idx >> 6, kc = tile % 16, mt = tile / 16;
    f32x4 v;
    for (int t = 0; t < 4; ++t) v[t] = src[(mt * 16 + (lane & 15)) * 256 + kc * 16 + 4 * (lane >> 4) + t];
    reinterpret_cast<f32x4*>(dst)[idx] = v;
}
__global__ void k_packs(f16x8* __restrict__ dst, const float* __restrict__ src, float wscale) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 16 * 8 * 64) return;
    const int lane = idx & 63, kc = (idx >> 6) & 7, mt = idx >> 9;
    f16x8 hi, lo;
    for (int e = 0; e < 8; ++e) {
        const float w = src[(mt * 16 + (lane & 15)) * 256 + kc * 32 + (lane >> 4) * 8 + e] * wscale;
        hi[e] = (_Float16)w;
        lo[e] = (_Float16)(w - (float)hi[e]);
    }
    dst[((mt * 8 + kc) * 2 + 0) * 64 + lane] = hi;
    dst[((mt * 8 + kc) * 2 + 1) * 64 + lane] = lo;
}

// local copy of the split trunk with bisection switches
//   1: layer 1 activations = cheap function of x (no FiLM sine)      2: skip FiLM sine in the MFMA layers
//   4: head reads one channel only                                   8: no MFMA loop (acc = small constant)
template <int FL>
__device__ __forceinline__ void trunk_dbg(const SdfNet& net, const float* xin, float* act, int ld, int wave, int lane) {
    constexpr int NT = kNT;
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    {
        f32x4 x[NT];
        for (int n = 0; n < NT; ++n) x[n] = *reinterpret_cast<const f32x4*>(xin + (n * 16 + j) * 4);
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 w[4];
            for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
            for (int n = 0; n < NT; ++n) {
                f32x4 v, h, d;
                if (FL & (16384 | 32768 | 65536)) {
                    // one operand-selection form at a time, written out; the other two products on aligned pairs
                    //   16384: v_pk_mul_f32  src0 = coordinate pair {x0, x1}, op_sel_hi:[0,1]    (both lanes read x0)
                    //   32768: v_pk_fma_f32  src1 = coordinate pair {x0, x1}, op_sel:[0,1,0]     (both lanes read x1)
                    //   65536: v_pk_fma_f32  src1 = pair {x2, unrelated},     op_sel_hi:[1,0,1]  (both lanes read x2)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    for (int p = 0; p < 2; ++p) {
                        f32x2 wa = {w[2 * p][0], w[2 * p + 1][0]}, wb = {w[2 * p][1], w[2 * p + 1][1]},
                              wc = {w[2 * p][2], w[2 * p + 1][2]};
                        f32x2 xa = {x[n][0], x[n][0]}, xb = {x[n][1], x[n][1]}, xc = {x[n][2], x[n][2]}, t;
                        f32x2 x01 = {x[n][0], x[n][1]}, x2j = {x[n][2], x[n][3]};
                        if (FL & 16384) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(x01), "v"(wa));
                        else asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(xa));
                        if (FL & 32768) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(t) : "v"(wb), "v"(x01));
                        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(wb), "v"(xb));
                        if (FL & 65536) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(t) : "v"(wc), "v"(x2j));
                        else asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(wc), "v"(xc));
                        v[2 * p] = t.x;
                        v[2 * p + 1] = t.y;
                    }
                } else if (FL & 8192) {   // the K = 3 chain as explicit packed FMAs on ALIGNED register pairs: no op_sel anywhere
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    for (int p = 0; p < 2; ++p) {
                        f32x2 wa = {w[2 * p][0], w[2 * p + 1][0]}, wb = {w[2 * p][1], w[2 * p + 1][1]},
                              wc = {w[2 * p][2], w[2 * p + 1][2]};
                        f32x2 xa = {x[n][0], x[n][0]}, xb = {x[n][1], x[n][1]}, xc = {x[n][2], x[n][2]}, t;
                        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(wa), "v"(xa));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(wb), "v"(xb));
                        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(wc), "v"(xc));
                        v[2 * p] = t.x;
                        v[2 * p + 1] = t.y;
                    }
                } else {
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], x[n][2], fmaf(w[r][1], x[n][1], w[r][0] * x[n][0]));
                }
                if (FL & 1024) asm volatile("s_nop 7\n\ts_nop 7");        // 16 idle cycles between the packed chain and the sine
                if (FL & 2048) asm volatile("s_nop 0");
                if (FL & 4096) __builtin_amdgcn_sched_barrier(0);
                if (FL & 512) {
                    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(v[r]));   // K=3 chain unpacked, sine free to pack
                }
                if (FL & 1) {
                    for (int r = 0; r < 4; ++r) h[r] = fminf(fmaxf(v[r] * 3.0f, -1.0f), 1.0f) * kActScale;
                } else if (FL & (32 | 64 | 128 | 256)) {
                    for (int r = 0; r < 4; ++r) {
                        float w = fmaf(v[r], fw[r], pw[r]);
                        if (FL & 64) asm volatile("" : "+v"(w));          // one element at a time: no v_pk_* packing
                        const float q = rintf(w);
                        const float rr = w - q;
                        const float r2 = rr * rr;
                        float p = fmaf(r2, 0.0772201280771219f * kActScale, -0.5980451736306471f * kActScale);
                        p = fmaf(p, r2, 2.550031377188653f * kActScale);
                        p = fmaf(p, r2, -5.167706878920042f * kActScale);
                        p = fmaf(p, r2, 3.1415925800446054f * kActScale);
                        float val = p * rr;
                        if (!(FL & 32)) val = __uint_as_float(__float_as_uint(val) ^ ((unsigned)(int)q << 31));
                        if (FL & 128) val = fminf(fmaxf(val, -1000.0f), 1000.0f);
                        if (FL & 64) asm volatile("" : "+v"(val));
                        h[r] = val;
                    }
                }
                else film_sine<false>(v, fw, pw, zero4, kActScale, h, d);
                store_split4(act, ld, 512, n * 16 + j, ch0, h);
            }
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int k = 1; k < 6; ++k) {
        f32x4 acc[kSdfMT][NT];
        for (int m = 0; m < kSdfMT; ++m)
            for (int n = 0; n < NT; ++n) zero_acc(acc[m][n]);
        if (FL & 8) {
            for (int m = 0; m < kSdfMT; ++m)
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(k + m + n + j);
        } else {
            gemm_acc_split<8, kSdfMT, NT>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);
        }
        __syncthreads();
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fws + k * 256 + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
            for (int n = 0; n < NT; ++n) {
                f32x4 h, d;
                if (FL & 2) {
                    for (int r = 0; r < 4; ++r) h[r] = fminf(fmaxf(acc[m][n][r] * fw[r] * 3.0f, -1.0f), 1.0f) * kActScale;
                }
                else film_sine<false>(acc[m][n], fw, pw, zero4, kActScale, h, d);
                store_split4(act, ld, 512, n * 16 + j, ch0, h);
            }
        }
        __syncthreads();
    }
}

template <int FL>
__global__ __launch_bounds__(kThreads, 4) void k_trunk_dbg(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* act = (FL & 16) ? smem : smem + 64 * 8;
    float* xin = (FL & 16) ? smem + 64 * kSdfLd : smem;
    float* outv = xin + 64 * 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        trunk_dbg<FL>(net, xin, act, kSdfLd, wave, lane);
        if (FL & 4) {
            if (tid < kTile) outv[tid * 4] = load_split(act, kSdfLd, 512, tid, 7);
        } else {
            sdf_head<true>(net, act, kSdfLd, outv, 4, tid);
        }
        __syncthreads();
        if (tid < kTile && tile * kTile + tid < n) out[tile * kTile + tid] = outv[tid * 4];
        __syncthreads();
    }
}

template <int FL>
void run_dbg(const char* name, SdfNet net, const float* dX, int n, float* dO, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_dbg<FL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> r0(n), r1(n);
    auto launch = [&] { k_trunk_dbg<FL><<<512, kThreads, lds>>>(net, dX, n, dO); };
    launch(); hipDeviceSynchronize();
    hipMemcpy(r0.data(), dO, n * 4, hipMemcpyDeviceToHost);
    size_t worst = 0;
    for (int rep = 0; rep < 4; ++rep) {
        launch(); hipDeviceSynchronize();
        hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int i = 0; i < n; ++i) bad += memcmp(&r0[i], &r1[i], 4) != 0;
        worst = bad > worst ? bad : worst;
        if (FL == 4 && bad) {
            // are the wrong values some OTHER point's right value (stale or foreign LDS rows)?
            std::vector<float> sorted(r0);
            std::sort(sorted.begin(), sorted.end());
            size_t found = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&r0[i], &r1[i], 4) && std::binary_search(sorted.begin(), sorted.end(), r1[i])) ++found;
            printf("  rep %d: %zu wrong values, %zu of them equal some other point's reference value\n", rep, bad, found);
        }
        if (FL == 0 && rep == 0) {
            int shown = 0, last_tile = -1, tiles = 0;
            for (int i = 0; i < n; ++i)
                if (memcmp(&r0[i], &r1[i], 4)) {
                    if (i / 64 != last_tile) { ++tiles; last_tile = i / 64; }
                    if (shown++ < 48) printf("  idx %d (tile %d, pt %d): %.6f vs %.6f\n", i, i / 64, i % 64, r0[i], r1[i]);
                }
            printf("  %zu mismatches in %d tiles\n", bad, tiles);
        }
    }
    printf("dbg flags %2d (%s): worst rerun differs in %zu of %d\n", FL, name, worst, n);
}

// product split trunk on 16*NT-point tiles; lds_pad forces one workgroup per CU
template <int NT>
__global__ __launch_bounds__(kThreads) void k_trunk_nt(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TW = 16 * NT;
    float* xin = smem;
    float* outv = xin + TW * 4;
    float* act = outv + TW * 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        if (tid < TW) {
            const int i = tile * TW + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][NT];
        sdf_trunk<false, NT, true>(net, xin, act, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<true>(net, act, kSdfLd, outv, 4, tid, TW);
        __syncthreads();
        if (tid < TW && tile * TW + tid < n) out[tile * TW + tid] = outv[tid * 4];
        __syncthreads();
    }
}

template <int NT>
void run_nt(const char* name, SdfNet net, const float* dX, int n, float* dO, size_t lds, int grid, const std::vector<float>& ref) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_nt<NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> r0(n), r1(n);
    auto launch = [&] { k_trunk_nt<NT><<<grid, kThreads, lds>>>(net, dX, n, dO); };
    launch(); hipDeviceSynchronize();
    hipMemcpy(r0.data(), dO, n * 4, hipMemcpyDeviceToHost);
    size_t worst = 0, vs_ref = 0;
    for (int i = 0; i < n; ++i) vs_ref += memcmp(&r0[i], &ref[i], 4) != 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int i = 0; i < n; ++i) bad += memcmp(&r0[i], &r1[i], 4) != 0;
        worst = bad > worst ? bad : worst;
    }
    printf("%-44s grid %3d: %.3f ms (%.0f TF algorithmic), worst rerun differs in %zu, differs from 1-WG/CU reference in %zu\n", name, grid, ms,
           (double)n * 657408 / ms / 1e9, worst, vs_ref);
}

// phase clocks of the split trunk on a CU-owning workgroup (s_memtime ticks, wave 0)
__global__ __launch_bounds__(kThreads, 4) void k_trunk_clk(SdfNet net, const float* __restrict__ x, int n, float* out, float* clk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;
    float* outv = xin + 64 * 4;
    float* act = outv + 64 * 4;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4, mt0 = wave * 2;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    unsigned long long tg = 0, tb1 = 0, te = 0, tb2 = 0, t0 = 0, tl0 = 0, th = 0;
    int tiles = 0;
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        ++tiles;
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        t0 = __builtin_amdgcn_s_memtime();
        {
            f32x4 xx[4];
            for (int nn = 0; nn < 4; ++nn) xx[nn] = *reinterpret_cast<const f32x4*>(xin + (nn * 16 + j) * 4);
            for (int m = 0; m < 2; ++m) {
                const int ch0 = (mt0 + m) * 16 + 4 * g;
                f32x4 w[4];
                for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
                const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
                const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
                for (int nn = 0; nn < 4; ++nn) {
                    f32x4 v, h, d;
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], xx[nn][2], fmaf(w[r][1], xx[nn][1], w[r][0] * xx[nn][0]));
                    film_sine<false>(v, fw, pw, zero4, kActScale, h, d);
                    store_split4(act, kSdfLd, 512, nn * 16 + j, ch0, h);
                }
            }
        }
        __syncthreads();
        tl0 += __builtin_amdgcn_s_memtime() - t0;
#pragma unroll 1
        for (int k = 1; k < 6; ++k) {
            f32x4 acc[2][4];
            for (int m = 0; m < 2; ++m) for (int nn = 0; nn < 4; ++nn) zero_acc(acc[m][nn]);
            unsigned long long a = __builtin_amdgcn_s_memtime();
            gemm_acc_split<8, 2, 4>(net.wps[k - 1], mt0, act, kSdfLd, 512, acc, lane);
            unsigned long long b = __builtin_amdgcn_s_memtime();
            tg += b - a;
            __syncthreads();
            a = __builtin_amdgcn_s_memtime();
            tb1 += a - b;
            for (int m = 0; m < 2; ++m) {
                const int ch0 = (mt0 + m) * 16 + 4 * g;
                const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fws + k * 256 + ch0);
                const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                for (int nn = 0; nn < 4; ++nn) {
                    f32x4 h, d;
                    film_sine<false>(acc[m][nn], fw, pw, zero4, kActScale, h, d);
                    store_split4(act, kSdfLd, 512, nn * 16 + j, ch0, h);
                }
            }
            b = __builtin_amdgcn_s_memtime();
            te += b - a;
            __syncthreads();
            tb2 += __builtin_amdgcn_s_memtime() - b;
        }
        t0 = __builtin_amdgcn_s_memtime();
        sdf_head<true>(net, act, kSdfLd, outv, 4, tid);
        __syncthreads();
        if (tid < kTile && tile * kTile + tid < n) out[tile * kTile + tid] = outv[tid * 4];
        __syncthreads();
        th += __builtin_amdgcn_s_memtime() - t0;
    }
    if (lane == 0 && (wave == 0 || wave == 5) && tiles) {
        float* c = clk + (blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
        c[0] = (float)tg / tiles / 5; c[1] = (float)tb1 / tiles / 5; c[2] = (float)te / tiles / 5; c[3] = (float)tb2 / tiles / 5;
        c[4] = (float)tl0 / tiles; c[5] = (float)th / tiles; c[6] = (float)tiles;
    }
}

// the de-phased two-tile trunk (copy of sdf_trunk_pair_split with clocks): per phase, time of the epilogue part,
// of the GEMM part, and of the barrier wait
__global__ __launch_bounds__(kThreads) void k_pair_clk(SdfNet net, const float* __restrict__ x, int n, float* out, float* clk, int mode) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;
    float* outv = xin + 128 * 4;
    float* actA = outv + 128 * 4;
    float* actB = actA + 64 * kSdfLd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4, mt0 = wave * 2;
    const int ld = kSdfLd;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    unsigned long long tE = 0, tG = 0, tB = 0;
    int steps = 0;
    auto layer0 = [&](const float* xi, float* act) {
        f32x4 xx[4];
        for (int nn = 0; nn < 4; ++nn) xx[nn] = *reinterpret_cast<const f32x4*>(xi + (nn * 16 + j) * 4);
        for (int m = 0; m < 2; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 w[4];
            for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
            for (int nn = 0; nn < 4; ++nn) {
                f32x4 v, h, d;
                for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], xx[nn][2], fmaf(w[r][1], xx[nn][1], w[r][0] * xx[nn][0]));
                film_sine<false>(v, fw, pw, zero4, kActScale, h, d);
                store_split4(act, ld, 512, nn * 16 + j, ch0, h);
            }
        }
    };
    auto gemm = [&](int k, const float* act, f32x4 (&acc)[2][4]) {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        for (int m = 0; m < 2; ++m) for (int nn = 0; nn < 4; ++nn) zero_acc(acc[m][nn]);
        gemm_acc_split<8, 2, 4>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);
        tG += __builtin_amdgcn_s_memtime() - a;
    };
    auto epi = [&](int k, const f32x4 (&acc)[2][4], float* act) {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        for (int m = 0; m < 2; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fws + k * 256 + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
            for (int nn = 0; nn < 4; ++nn) {
                f32x4 h, d;
                film_sine<false>(acc[m][nn], fw, pw, zero4, kActScale, h, d);
                store_split4(act, ld, 512, nn * 16 + j, ch0, h);
            }
        }
        tE += __builtin_amdgcn_s_memtime() - a;
    };
    auto bar = [&]() {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        __syncthreads();
        tB += __builtin_amdgcn_s_memtime() - a;
    };
    const bool epi_first = mode == 0 ? (wave < 4) : (mode == 1 ? (wave & 1) : (mode == 2 ? ((__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1) != 0) : true));
    for (int tile = blockIdx.x; tile * 128 < n; tile += gridDim.x) {
        ++steps;
        if (tid < 128) {
            const int i = tile * 128 + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        layer0(xin, actA);
        layer0(xin + 256, actB);
        __syncthreads();
        f32x4 accA[2][4], accB[2][4];
        gemm(1, actA, accA);
        bar();
#pragma unroll 1
        for (int k = 1; k < 6; ++k) {
#pragma unroll 1
            for (int step = 0; step < 2; ++step) {
                if ((step == 0) == epi_first) epi(k, accA, actA);
                else gemm(k, actB, accB);
            }
            bar();
#pragma unroll 1
            for (int step = 0; step < 2; ++step) {
                if ((step == 0) == epi_first) epi(k, accB, actB);
                else if (k < 5) gemm(k + 1, actA, accA);
            }
            bar();
        }
        sdf_head<true>(net, actA, ld, outv, 4, tid);
        sdf_head<true>(net, actB, ld, outv + 256, 4, tid);
        __syncthreads();
        if (tid < 128 && tile * 128 + tid < n) out[tile * 128 + tid] = outv[tid * 4];
        __syncthreads();
    }
    if (lane == 0 && (wave == 0 || wave == 5) && steps) {
        float* c = clk + (blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
        c[0] = (float)tE / steps / 10; c[1] = (float)tG / steps / 10; c[2] = (float)tB / steps / 11; c[6] = (float)steps;
    }
}

// 16 waves in ONE workgroup (4 per SIMD): waves 0-7 run tile A, waves 8-15 tile B, same code, shared barriers.
// Is the irreproducibility about four f16-MFMA waves per SIMD, or about two workgroups on a CU?
__global__ __launch_bounds__(1024) void k_trunk_16w(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, half = tid >> 9, t9 = tid & 511, wave = t9 >> 6, lane = tid & 63;
    float* xin = smem + half * 64 * 4;
    float* outv = smem + 128 * 4 + half * 64 * 4;
    float* act = smem + 256 * 4 + half * 64 * kSdfLd;
    for (int tile = blockIdx.x * 2 + half; (tile - half) * kTile < n; tile += gridDim.x * 2) {
        if (t9 < kTile) {
            const int i = tile * kTile + t9;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[t9] = v;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][kNT];
        sdf_trunk<false, kNT, true>(net, xin, act, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<true>(net, act, kSdfLd, outv, 4, t9);
        __syncthreads();
        if (t9 < kTile && tile * kTile + t9 < n) out[tile * kTile + t9] = outv[t9 * 4];
        __syncthreads();
    }
}

// 16 waves, two tiles, the second half-workgroup one step behind the first: while waves 0-7 run a GEMM, waves 8-15
// run the previous epilogue of their own tile (and vice versa) -- matrix pipe and vector ALU busy together.
__global__ __launch_bounds__(1024) void k_trunk_16w_skew(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, half = tid >> 9, t9 = tid & 511, wave = t9 >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4, mt0 = wave * 2, ld = kSdfLd;
    float* xin = smem + half * 64 * 4;
    float* outv = smem + 128 * 4 + half * 64 * 4;
    float* act = smem + 256 * 4 + half * 64 * kSdfLd;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x * 2 + half; (tile - half) * kTile < n; tile += gridDim.x * 2) {
        if (t9 < kTile) {
            const int i = tile * kTile + t9;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[t9] = v;
        }
        __syncthreads();
        f32x4 acc[2][4];
#pragma unroll 1
        for (int p = 0; p < 12; ++p) {
            const int st = p - half;   // this half's step: 0 = layer 1 (VALU), 2k-1 = GEMM k, 2k = epilogue k
            if (st == 0) {
                f32x4 xx[4];
                for (int nn = 0; nn < 4; ++nn) xx[nn] = *reinterpret_cast<const f32x4*>(xin + (nn * 16 + j) * 4);
                for (int m = 0; m < 2; ++m) {
                    const int ch0 = (mt0 + m) * 16 + 4 * g;
                    f32x4 w[4];
                    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
                    const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
                    const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
                    for (int nn = 0; nn < 4; ++nn) {
                        f32x4 v, h, d;
                        for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], xx[nn][2], fmaf(w[r][1], xx[nn][1], w[r][0] * xx[nn][0]));
                        film_sine<false>(v, fw, pw, zero4, kActScale, h, d);
                        store_split4(act, ld, 512, nn * 16 + j, ch0, h);
                    }
                }
            } else if (st >= 1 && st <= 10) {
                const int k = (st + 1) >> 1;
                if (st & 1) {
                    for (int m = 0; m < 2; ++m) for (int nn = 0; nn < 4; ++nn) zero_acc(acc[m][nn]);
                    gemm_acc_split<8, 2, 4>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);
                } else {
                    for (int m = 0; m < 2; ++m) {
                        const int ch0 = (mt0 + m) * 16 + 4 * g;
                        const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fws + k * 256 + ch0);
                        const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                        for (int nn = 0; nn < 4; ++nn) {
                            f32x4 h, d;
                            film_sine<false>(acc[m][nn], fw, pw, zero4, kActScale, h, d);
                            store_split4(act, ld, 512, nn * 16 + j, ch0, h);
                        }
                    }
                }
            }
            __syncthreads();
        }
        sdf_head<true>(net, act, kSdfLd, outv, 4, t9);
        __syncthreads();
        if (t9 < kTile && tile * kTile + t9 < n) out[tile * kTile + t9] = outv[t9 * 4];
        __syncthreads();
    }
}

// ---- two tiles per workgroup, epilogue of one tile INTERLEAVED INSIDE the GEMM loop of the other (same wave) ----
// Cross-wave overlap of VALU epilogues and MFMA GEMMs does not materialise on this part (kernel time ~ MFMA time +
// VALU time in every arrangement measured above); within one wave, VALU instructions issued between MFMAs do run in
// the MFMA's shadow.  Each kc chunk of tile X's GEMM (24 MFMAs) carries the FiLM-sine epilogue of one (m, n) group
// of tile Y; sched_group_barrier asks for a 1 MFMA : 3 VALU issue pattern.
template <bool DO_G, bool DO_E, bool HINT>
__device__ __forceinline__ void fused_phase(const SdfNet& net, int kG, const float* actX, f32x4 (&accX)[2][4], int kE,
                                            const f32x4 (&accY)[2][4], float* actY, int ld, int mt0, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 fwE[2], pwE[2];
    if (DO_E) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            fwE[m] = *reinterpret_cast<const f32x4*>(net.fws + kE * 256 + ch0);
            pwE[m] = *reinterpret_cast<const f32x4*>(net.pw + kE * 256 + ch0);
        }
    }
    const char* bptr = reinterpret_cast<const char*>(actX) + j * ld * 4 + g * 16;
    const f16x8* aptr = DO_G ? net.wps[kG - 1] + (size_t)mt0 * 8 * 2 * 64 + lane : nullptr;
    f16x8 ah[2][2], al[2][2];
    if (DO_G) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) zero_acc(accX[m][n]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            ah[0][m] = aptr[((m * 8) * 2 + 0) * 64];
            al[0][m] = aptr[((m * 8) * 2 + 1) * 64];
        }
    }
#pragma unroll
    for (int kc = 0; kc < 8; ++kc) {
        const int c = kc & 1, x = c ^ 1;
        f16x8 bh[4], bl[4];
        if (DO_G) {
            if (kc + 1 < 8) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    ah[x][m] = aptr[((m * 8 + kc + 1) * 2 + 0) * 64];
                    al[x][m] = aptr[((m * 8 + kc + 1) * 2 + 1) * 64];
                }
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bh[n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + kc * 64);
                bl[n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + 512 + kc * 64);
            }
        }
        f32x4 h, d;
        if (DO_E) film_sine<false>(accY[kc >> 2][kc & 3], fwE[kc >> 2], pwE[kc >> 2], zero4, kActScale, h, d);
        if (DO_G) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) accX[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c][m], bh[n], accX[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) accX[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][m], bl[n], accX[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) accX[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][m], bh[n], accX[m][n], 0, 0, 0);
        }
        if (DO_E) store_split4(actY, ld, 512, (kc & 3) * 16 + j, (mt0 + (kc >> 2)) * 16 + 4 * g, h);
        if (DO_G && DO_E && HINT) {
#pragma unroll
            for (int q = 0; q < 24; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);   // 3 VALU
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <bool HINT>
__global__ __launch_bounds__(kThreads) void k_pair_fused(SdfNet net, const float* __restrict__ x, int n, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;
    float* outv = xin + 128 * 4;
    float* actA = outv + 128 * 4;
    float* actB = actA + 64 * kSdfLd;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, g = lane >> 4, mt0 = wave * 2;
    const int ld = kSdfLd;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto layer0 = [&](const float* xi, float* act) {
        f32x4 xx[4];
        for (int nn = 0; nn < 4; ++nn) xx[nn] = *reinterpret_cast<const f32x4*>(xi + (nn * 16 + j) * 4);
        for (int m = 0; m < 2; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 w[4];
            for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
            for (int nn = 0; nn < 4; ++nn) {
                f32x4 v, h, d;
                for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], xx[nn][2], fmaf(w[r][1], xx[nn][1], w[r][0] * xx[nn][0]));
                no_pack(v);
                film_sine<false>(v, fw, pw, zero4, kActScale, h, d);
                store_split4(act, ld, 512, nn * 16 + j, ch0, h);
            }
        }
    };
    for (int tile = blockIdx.x; tile * 128 < n; tile += gridDim.x) {
        if (tid < 128) {
            const int i = tile * 128 + tid;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = f32x4{x[i * 3], x[i * 3 + 1], x[i * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = v;
        }
        __syncthreads();
        layer0(xin, actA);
        layer0(xin + 256, actB);
        __syncthreads();
        f32x4 accA[2][4], accB[2][4];
        fused_phase<true, false, HINT>(net, 1, actA, accA, 0, accB, actB, ld, mt0, lane);       // G(A,1)
        __syncthreads();
#pragma unroll 1
        for (int k = 1; k < 6; ++k) {
            fused_phase<true, true, HINT>(net, k, actB, accB, k, accA, actA, ld, mt0, lane);     // G(B,k) + E(A,k)
            __syncthreads();
            if (k < 5) fused_phase<true, true, HINT>(net, k + 1, actA, accA, k, accB, actB, ld, mt0, lane);   // G(A,k+1) + E(B,k)
            else fused_phase<false, true, HINT>(net, 0, actA, accA, k, accB, actB, ld, mt0, lane);           // E(B,5)
            __syncthreads();
        }
        sdf_head<true>(net, actA, ld, outv, 4, tid);
        sdf_head<true>(net, actB, ld, outv + 256, 4, tid);
        __syncthreads();
        if (tid < 128 && tile * 128 + tid < n) out[tile * 128 + tid] = outv[tid * 4];
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int n = 400000;
    unsigned s = 777;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (s >> 8) / 8388608.0f - 1.0f; };
    std::vector<float> W(5 * 65536), w0(256 * 4), w6(256), cst(6 * 256), X(n * 3);
    const float wmax = sqrtf(6.0f / 256.0f) / 30.0f * 2.0f;
    for (auto& v : W) v = rnd() * wmax;
    for (auto& v : w0) v = rnd() * 0.3f;
    for (auto& v : w6) v = rnd() * 0.05f;
    for (auto& v : X) v = rnd();
    auto dev = [&](const std::vector<float>& h) { float* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice); return d; };
    float *dW = dev(W), *dX = dev(X);
    SdfNet net;
    net.w0 = dev(w0);
    net.w6 = dev(w6);
    std::vector<float> fw(6 * 256), pw(6 * 256), fws(6 * 256), fr(6 * 256, 1.0f), b6(4, 0.01f);
    const float c = (float)kFilmScale, wscale = exp2f(14.0f - ceilf(log2f(wmax)));
    for (int i = 0; i < 6 * 256; ++i) { fw[i] = c; pw[i] = rnd() * 0.02f * c; fws[i] = i < 256 ? c : c / (wscale * kActScale); }
    net.fw = dev(fw); net.pw = dev(pw); net.fws = dev(fws); net.freq = dev(fr); net.phase = net.freq; net.bias = net.freq; net.b6 = dev(b6);
    float* p32; f16x8* ps;
    hipMalloc(&p32, W.size() * 4); hipMalloc(&ps, W.size() * 4);
    for (int l = 0; l < 5; ++l) {
        k_pack32<<<64, 256>>>(p32 + (size_t)l * 65536, dW + (size_t)l * 65536);
        k_packs<<<32, 256>>>(ps + (size_t)l * 16 * 8 * 2 * 64, dW + (size_t)l * 65536, wscale);
        net.wp[l] = p32 + (size_t)l * 65536; net.wpT[l] = net.wp[l]; net.wps[l] = ps + (size_t)l * 16 * 8 * 2 * 64;
    }
    float* dO; hipMalloc(&dO, n * 4);
    const size_t lds = (64 * 4 * 2) * 4 + (size_t)64 * kSdfLd * 4;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> r0(n), r1(n);
    for (int grid : {256, 512}) {
        for (int split = 0; split < 2; ++split) {
            auto launch = [&] { if (split) k_trunk<true><<<grid, kThreads, lds>>>(net, dX, n, dO); else k_trunk<false><<<grid, kThreads, lds>>>(net, dX, n, dO); };
            launch(); hipDeviceSynchronize();
            hipMemcpy(r0.data(), dO, n * 4, hipMemcpyDeviceToHost);
            size_t worst = 0; float md = 0;
            for (int rep = 0; rep < 4; ++rep) {
                launch(); hipDeviceSynchronize();
                hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                for (int i = 0; i < n; ++i) if (memcmp(&r0[i], &r1[i], 4)) { ++bad; md = fmaxf(md, fabsf(r0[i] - r1[i])); }
                worst = bad > worst ? bad : worst;
            }
            printf("grid %3d  %-6s trunk: worst rerun differs in %zu of %d (max |diff| %.3g)  sample out %.6f\n", grid, split ? "split" : "exact", worst, n, md, r0[12345]);
        }
    }
    run_dbg<0>("K=3 layer as hipcc packs it (v_pk_fma + op_sel)", net, dX, n, dO, lds);
    {
        std::vector<float> ref(n);
        k_trunk<true><<<256, kThreads, lds>>>(net, dX, n, dO);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), dO, n * 4, hipMemcpyDeviceToHost);
        const size_t l4 = (64 * 4 * 2) * 4 + (size_t)64 * kSdfLd * 4, l8 = (128 * 4 * 2) * 4 + (size_t)128 * kSdfLd * 4;
        run_nt<4>("64-pt tiles, 2 WG/CU", net, dX, n, dO, l4, 512, ref);
        run_nt<4>("64-pt tiles, LDS padded to 1 WG/CU", net, dX, n, dO, 90 * 1024, 512, ref);
        run_nt<4>("64-pt tiles, LDS padded to 1 WG/CU", net, dX, n, dO, 90 * 1024, 256, ref);
        run_nt<8>("128-pt tiles (1 WG/CU)", net, dX, n, dO, l8, 256, ref);
        run_nt<8>("128-pt tiles (1 WG/CU)", net, dX, n, dO, l8, 512, ref);
    }
    {
        float* dC; hipMalloc(&dC, 512 * 2 * 8 * 4);
        for (int grid : {256, 512, 1024}) {   // 1024: un-padded LDS, two workgroups per CU
            hipMemset(dC, 0, 512 * 2 * 8 * 4);
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_clk), hipFuncAttributeMaxDynamicSharedMemorySize, 90 * 1024);
            const size_t lclk = grid == 1024 ? lds : 90 * 1024;
            if (grid == 1024) grid = 512;
            k_trunk_clk<<<grid, kThreads, lclk>>>(net, dX, n, dO, dC);
            hipDeviceSynchronize();
            std::vector<float> c(grid * 16);
            hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
            for (int wv = 0; wv < 2; ++wv) {
                double a[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int b = 0; b < grid; ++b) for (int k = 0; k < 7; ++k) a[k] += c[(b * 2 + wv) * 8 + k] / grid;
                printf("%s, grid %d, wave %d: per layer gemm %.0f  barrier %.0f  epilogue %.0f  barrier %.0f | layer-1 %.0f  head+io %.0f ticks, %.1f tiles/WG\n",
                       lclk == lds ? "2 WG/CU" : "1 WG/CU", grid, wv ? 5 : 0, a[0], a[1], a[2], a[3], a[4], a[5], a[6]);
            }
        }
    }
    {
        float* dC; hipMalloc(&dC, 512 * 2 * 8 * 4);
        const size_t lp = (128 * 4 * 2) * 4 + (size_t)128 * kSdfLd * 4;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_clk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp);
        for (int mode = 0; mode < 4; ++mode) {
            hipMemset(dC, 0, 512 * 2 * 8 * 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k_pair_clk<<<512, kThreads, lp>>>(net, dX, n, dO, dC, mode);
            hipEventRecord(e0);
            k_pair_clk<<<512, kThreads, lp>>>(net, dX, n, dO, dC, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<float> c(512 * 16);
            hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost);
            for (int wv = 0; wv < 2; ++wv) {
                double a[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int b = 0; b < 512; ++b) for (int k = 0; k < 7; ++k) a[k] += c[(b * 2 + wv) * 8 + k] / 512;
                printf("pair mode %d (%.3f ms), wave %d: per phase epilogue %.0f  gemm %.0f  barrier %.0f ticks\n", mode, ms, wv ? 5 : 0, a[0], a[1], a[2]);
            }
        }
    }
    {
        const size_t l16 = (256 * 4) * 4 + (size_t)128 * kSdfLd * 4;
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_16w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16);
        std::vector<float> ref(n), r1(n);
        k_trunk<true><<<256, kThreads, lds>>>(net, dX, n, dO);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), dO, n * 4, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(dO, 0, n * 4);
            hipEventRecord(e0);
            k_trunk_16w<<<256, 1024, l16>>>(net, dX, n, dO);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (int i = 0; i < n; ++i) bad += memcmp(&ref[i], &r1[i], 4) != 0;
            printf("16 waves in one workgroup (4 per SIMD): %.3f ms, differs from the 1-WG/CU reference in %zu of %d\n", ms, bad, n);
        }
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_trunk_16w_skew), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l16);
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(dO, 0, n * 4);
            hipEventRecord(e0);
            k_trunk_16w_skew<<<256, 1024, l16>>>(net, dX, n, dO);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (int i = 0; i < n; ++i) bad += memcmp(&ref[i], &r1[i], 4) != 0;
            printf("16 waves, halves skewed by one step: %.3f ms, differs from the 1-WG/CU reference in %zu of %d\n", ms, bad, n);
        }
    }
    {
        const size_t lp = (128 * 4 * 2) * 4 + (size_t)128 * kSdfLd * 4;
        std::vector<float> ref(n), r1(n);
        k_trunk<true><<<256, kThreads, lds>>>(net, dX, n, dO);
        hipDeviceSynchronize();
        hipMemcpy(ref.data(), dO, n * 4, hipMemcpyDeviceToHost);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int hint = 0; hint < 2; ++hint) {
            auto launch = [&](int grid) { if (hint) k_pair_fused<true><<<grid, kThreads, lp>>>(net, dX, n, dO); else k_pair_fused<false><<<grid, kThreads, lp>>>(net, dX, n, dO); };
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_fused<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp);
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_fused<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lp);
            for (int grid : {256, 512}) {
                hipMemset(dO, 0, n * 4);
                launch(grid); hipDeviceSynchronize();
                hipEventRecord(e0); launch(grid); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(r1.data(), dO, n * 4, hipMemcpyDeviceToHost);
                size_t bad = 0;
                for (int i = 0; i < n; ++i) bad += memcmp(&ref[i], &r1[i], 4) != 0;
                printf("two tiles, epilogue inside the GEMM loop (%s), grid %d: %.3f ms (%.0f TF algorithmic), differs from reference in %zu of %d\n",
                       hint ? "1 MFMA : 3 VALU hint" : "compiler order", grid, ms, (double)n * 657408 / ms / 1e9, bad, n);
            }
        }
    }
    run_dbg<8192>("K=3 chain as explicit v_pk_mul/fma_f32 on aligned pairs, no op_sel", net, dX, n, dO, lds);
    run_dbg<16384>("explicit packed chain, ONLY v_pk_mul_f32 op_sel_hi:[0,1] on the coordinate pair", net, dX, n, dO, lds);
    run_dbg<32768>("explicit packed chain, ONLY v_pk_fma_f32 op_sel:[0,1,0] on the coordinate pair", net, dX, n, dO, lds);
    run_dbg<65536>("explicit packed chain, ONLY v_pk_fma_f32 op_sel_hi:[1,0,1] on {x2, unrelated}", net, dX, n, dO, lds);
    run_dbg<114688>("explicit packed chain, all three operand-selection forms", net, dX, n, dO, lds);
    run_dbg<1024>("packed chain, then 16 idle cycles (s_nop 7 x2) before the sine", net, dX, n, dO, lds);
    run_dbg<2048>("packed chain, then s_nop 0 before the sine", net, dX, n, dO, lds);
    run_dbg<4096>("packed chain, then a scheduling barrier only", net, dX, n, dO, lds);
    run_dbg<4>("one-channel head", net, dX, n, dO, lds);
    run_dbg<512>("K=3 chain fenced, product sine", net, dX, n, dO, lds);
    run_dbg<256>("layer-1 sine re-written inline (same math)", net, dX, n, dO, lds);
    run_dbg<32>("layer-1 sine without the sign flip", net, dX, n, dO, lds);
    run_dbg<64>("layer-1 sine, packing broken", net, dX, n, dO, lds);
    run_dbg<128>("layer-1 sine clamped to +-1000", net, dX, n, dO, lds);
    run_dbg<1>("clamp instead of sine in layer 1", net, dX, n, dO, lds);
    run_dbg<2>("clamp instead of sine in MFMA layers", net, dX, n, dO, lds);
    run_dbg<3>("clamp everywhere", net, dX, n, dO, lds);
    run_dbg<7>("clamp everywhere, 1-channel head", net, dX, n, dO, lds);
    run_dbg<0>("K=3 layer as hipcc packs it (v_pk_fma + op_sel)", net, dX, n, dO, lds);
    return 0;
}
