// mlp.hpp -- workgroup-cooperative MLP evaluation on the CDNA4 matrix cores (gfx950).
//
// Geometry (one "tile" = 64 points = 64 MFMA columns):
//   * workgroup = 512 threads = 8 wave64; every wave sees all 64 points (4 N-tiles of 16) and owns
//     a slice of the layer's output channels (M-tiles of 16): 2 M-tiles for a 256-wide layer,
//     1 for a 128-wide one.  8 independent f32x4 accumulators per wave hide the 40-cycle
//     dependent latency of v_mfma_f32_16x16x4_f32 (issue 32 cycles).
//   * arithmetic is fp32-class everywhere: root finding needs |residual| < 1e-5 m and sin(30 x) amplifies
//     input error, so plain bf16/fp16 inputs are out.  Two GEMM engines with the same accumulator ownership:
//       - exact: v_mfma_f32_16x16x4_f32 (== an fmaf chain), used by every reverse sweep and, when the frame
//         was prepared with ARAH_PRECISION_FP32, by everything;
//       - split: every fp32 operand is carried as hi + lo, two f16 pre-scaled by a power of two (22 significant
//         bits), and W x = W_lo x_hi + W_hi x_lo + W_hi x_hi runs as three v_mfma_f32_16x16x32_f16 with fp32
//         accumulation (the dropped lo*lo term is 2^-22 relative).  Measured against an fp64 chain of five
//         SIREN layers its error is BELOW the exact-fp32 engine's (tools/ubench/gemm_f16x3.hip) at 3/16 of
//         the matrix-pipe time.  Split activations live in LDS as [point][hi: K halves | lo: K halves].
//   * activations live in LDS as act[point][K] (row stride K+4 floats); B fragments are
//     ds_read_b128: lane (j = lane&15, g = lane>>4) reads act[n*16+j][kc*16 + 4g .. +3].
//   * weights stream from L2 in a layout packed once per frame so that the matching A fragment
//     (W[mt*16 + j][kc*16 + 4g .. +3]) is one coalesced 1 KiB global_load_dwordx4 per wave:
//         packed[((mt*KC + kc)*64 + lane)*4 + t]
//     MFMA step t multiplies A.t by B.t: lane group g supplies k = kc*16 + 4g + t on both
//     sides, so the k-permutation inside a 16-chunk is consistent and harmless.
//   * accumulator ownership: acc[m][n][r] = out[channel (mt0+m)*16 + 4g + r][point n*16 + j].
//     Epilogues (bias, FiLM, sin/softplus/relu, derivative factors) are lane-local, and a layer's
//     output is written back IN PLACE after a barrier (all waves have finished reading it).
#pragma once
#include <hip/hip_runtime.h>

#include "pointwise.hpp"

#ifndef ARAH_SYNC
#ifdef ARAH_ABL_NO_BARRIER   // timing ablation: no workgroup barriers inside the MLPs (results are wrong)
#define ARAH_SYNC() __builtin_amdgcn_wave_barrier()
#else
#define ARAH_SYNC() __syncthreads()
#endif
#endif

namespace arah {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 512;
constexpr int kWaves = 8;
constexpr int kTile = 64;     // points per tile
constexpr int kNT = 4;        // N-tiles (16 points each) per wave

__device__ __forceinline__ void zero_acc(f32x4& a) { a = f32x4{0.f, 0.f, 0.f, 0.f}; }

// Compiler fence for the K = 3 input layers.  hipcc (ROCm 7.2) turns their unrolled dot products into
// v_pk_fma_f32 with op_sel / op_sel_hi half-broadcasts of the coordinate operand; with that code in a kernel, two
// co-resident workgroups of the split engine produced irreproducible results on MI355X (whole 16-point groups
// wrong, run to run) -- tools/ubench/trunk_repro.hip bisects it down to exactly this: the same layer compiled to
// scalar v_fma_f32 is bit-reproducible.  The empty asm pins each value in its own VGPR, which keeps the SLP
// vectoriser away from the chain; the arithmetic (an explicit fmaf chain) is unchanged.
// -DARAH_ALLOW_PACK removes the fence (diagnostic build: the packed listing under profiles/ and trunk_repro come from it).
__device__ __forceinline__ void no_pack(f32x4& v) {
#ifndef ARAH_ALLOW_PACK
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#endif
}

// Weight fragments are addressed as (wave-uniform 64-bit base) + (per-lane 32-bit byte offset): the base stays in
// SGPRs and one offset VGPR serves every layer, instead of one 64-bit VGPR pointer per layer and 4 KB window
// (global_load_dwordx4 v, v_off, s[base:base+1] offset:imm).
template <typename Tp>
__device__ __forceinline__ Tp ld_frag(const void* __restrict__ base, unsigned lane_off, int const_off) {
#ifdef ARAH_FRAG_FLAT
    return *reinterpret_cast<const Tp*>(reinterpret_cast<const char*>(base) + const_off + (size_t)lane_off);
#else
    // a buffer load: the base is four scalar registers, the lane's offset one vector register, the constant part an
    // immediate (+ a scalar for the 4 KB window) -- whatever the unroller and the invariant-code motion make of the loop.
    // (The flat form left 64-bit vector addresses per 4 KB window, hoisted out of the tile loop and spilled: 154 scratch
    // reloads in front of as many fragment loads in round 4's k_shade.)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    static_assert(sizeof(Tp) == 16, "fragments are 16 bytes per lane");
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)lane_off + (const_off & 4095), const_off & ~4095, 0);
    return __builtin_bit_cast(Tp, v);
#endif
}

constexpr float kActScale = 1024.0f;       // activations are stored as f16 pairs of 1024 h
constexpr float kInvActScale = 1.0f / 1024.0f;

// Phase clocks for the instrumented build (tools/phase_clocks.py, -DARAH_CLOCKS): a wave accumulates the s_memtime
// ticks it spends between consecutive marks.  NoClk compiles to nothing.
struct NoClk {
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
};
struct PhaseClk {
    unsigned prev;
    unsigned acc[16];
    __device__ __forceinline__ void start() {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        prev = (unsigned)__builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int i) {
        const unsigned t = (unsigned)__builtin_amdgcn_s_memtime();
        acc[i] += t - prev;
        prev = t;
    }
};

// acc[m][n] += Wp(rows of M-tiles mt0..mt0+MT-1, KC 16-chunks) * act(64 points)
template <int KC, int MT, int NT = kNT>
__device__ __forceinline__ void gemm_acc(const float* __restrict__ wp, int mt0, const float* act, int ld,
                                         f32x4 (&acc)[MT][NT], int lane) {
    const int j = lane & 15, g = lane >> 4;
    const float* bptr = act + j * ld + 4 * g;
    const unsigned aoff = (unsigned)(mt0 * KC * 64 + lane) * 16u;
    auto lda = [&](int idx) { return ld_frag<f32x4>(wp, aoff, idx * 1024); };
    if constexpr (KC * MT <= 8) {
        // narrow layer: the whole A slice of this wave is 8 VGPR-quads -- load it once, no dependent waits
        f32x4 a_all[MT][KC];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) a_all[m][kc] = lda(m * KC + kc);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            __builtin_amdgcn_sched_barrier(0);   // keep the B fragments of later chunks from being hoisted (VGPRs)
            f32x4 b[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * ld + kc * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_all[m][kc][t], b[n][t], acc[m][n], 0, 0, 0);
        }
        return;
    }
    // software pipeline, one 16-chunk deep on both operands: the A fragments (global/L2) and the B fragments
    // (LDS) of chunk kc+1 are in flight while the 4*MT*NT MFMAs of chunk kc issue, so that a single wave per
    // SIMD keeps the matrix pipe busy (the ping-pong kernels rely on that)
    f32x4 a_cur[MT], a_nxt[MT], b_cur[NT], b_nxt[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) a_cur[m] = lda(m * KC);
#pragma unroll
    for (int n = 0; n < NT; ++n) b_cur[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * ld);
#pragma unroll 1
    for (int kc = 0; kc < KC; ++kc) {
        const int kn = (kc + 1 < KC) ? kc + 1 : kc;
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = lda(m * KC + kn);
#pragma unroll
        for (int n = 0; n < NT; ++n) b_nxt[n] = *reinterpret_cast<const f32x4*>(bptr + n * 16 * ld + kn * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[m][t], b_cur[n][t], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
#pragma unroll
        for (int n = 0; n < NT; ++n) b_cur[n] = b_nxt[n];
    }
}

// One 16x16 output tile (M-tile mt, N-tile nt): used for narrow output layers split over waves.
template <int KC>
__device__ __forceinline__ f32x4 gemm_one(const float* __restrict__ wp, int mt, int nt, const float* act, int ld,
                                          int lane) {
    const int j = lane & 15, g = lane >> 4;
    const float* bptr = act + (nt * 16 + j) * ld + 4 * g;
    const unsigned aoff = (unsigned)(mt * KC * 64 + lane) * 16u;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int kc = 0; kc < KC; ++kc) {
        const f32x4 a = ld_frag<f32x4>(wp, aoff, kc * 1024);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bptr + kc * 16);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc1, 0, 0, 0);
    }
    return acc0 + acc1;
}

// ---- split engine ------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// LDS placement of a split activation plane: a point's row is a sequence of 64-byte chunks (32 channels as halves), four
// 16-byte slots each -- slot g of chunk kc is what lane group g of the B fragment reads.  Rows are 8 dwords (mod 64) apart
// (kSdfLd = 264, kSkinLd = 136 floats) and the slots of a chunk are permuted by the point index, slot' = slot ^ ((pt >> 2) & 3):
// the sixteen lanes that one LDS cycle of a ds_read_b128 serves (lane groups of the form {j 0..3, 12..15 | g; j 4..11 | g + 1})
// then fall on sixteen different 16-byte bank groups.  With 260-float rows and no permutation every B-fragment read took
// two passes (tools/lds_layout.py counts them; rocprofv3 SQ_LDS_BANK_CONFLICT was 60 % of the LDS-active cycles of
// k_density).  The epilogue's 8-byte stores stay 2-way, which the store path hides.
__device__ __forceinline__ int split_slot(int pt, int slot) { return slot ^ ((pt >> 2) & 3); }
// byte offset inside a plane of channel ch of point pt
__device__ __forceinline__ int split_byte(int pt, int ch) {
    return (ch >> 5) * 64 + split_slot(pt, (ch >> 3) & 3) * 16 + (ch & 7) * 2;
}

// acc[m][n] += Wsplit(M-tiles mt0..mt0+MT-1, KC32 32-chunks) * act(16 NT points)
//   weights : wp[((mt*KC32 + kc)*2 + s)*64 + lane]  (s = 0 hi, 1 lo), lane (j, g) holds W[mt*16 + j][kc*32 + 8g .. +7]
//   act     : LDS rows of ld floats; hi plane at byte 0, lo plane at byte lo_off; lane reads k = kc*32 + 8g .. +7
// DEEP: fully unrolled, both operands one chunk ahead -- for kernels that run two waves per SIMD (large register
// budgets, little thread-level parallelism to hide latency).  Otherwise rolled, A one chunk ahead, B loaded at the
// top of the chunk: fits the 128-VGPR budget of four waves per SIMD, whose other waves hide the LDS latency.
template <int KC32, int MT, int NT, bool DEEP = false>
__device__ __forceinline__ void gemm_acc_split(const f16x8* __restrict__ wp, int mt0, const float* act, int ld,
                                               int lo_off, f32x4 (&acc)[MT][NT], int lane) {
    const int j = lane & 15, g = lane >> 4;
    const char* bptr = reinterpret_cast<const char*>(act) + j * ld * 4 + split_slot(j, g) * 16;
    const unsigned aoff = (unsigned)(mt0 * KC32 * 2 * 64 + lane) * 16u;
    auto lda = [&](int idx) { return ld_frag<f16x8>(wp, aoff, idx * 1024); };
    if constexpr (DEEP || KC32 <= 4) {
        f16x8 ah[2][MT], al[2][MT], bh[2][NT], bl[2][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[0][m] = lda(((m * KC32) * 2 + 0));
            al[0][m] = lda(((m * KC32) * 2 + 1));
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[0][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4);
            bl[0][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + lo_off);
        }
#pragma unroll
        for (int kc = 0; kc < KC32; ++kc) {
            const int c = kc & 1, x = c ^ 1;
            if (kc + 1 < KC32) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[x][m] = lda(((m * KC32 + kc + 1) * 2 + 0));
                    al[x][m] = lda(((m * KC32 + kc + 1) * 2 + 1));
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    bh[x][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + (kc + 1) * 64);
                    bl[x][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + lo_off + (kc + 1) * 64);
                }
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)   // small terms first, three passes over the MT*NT independent accumulators
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[c][m], bh[c][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][m], bl[c][n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[c][m], bh[c][n], acc[m][n], 0, 0, 0);
            // keep the scheduler from sinking the next chunk's loads below this chunk's MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        f16x8 ah[MT], al[MT], ahn[MT], aln[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[m] = lda(((m * KC32) * 2 + 0));
            al[m] = lda(((m * KC32) * 2 + 1));
        }
#ifdef ARAH_A_AHEAD2   // the A fragments TWO chunks ahead (tuning variant, tools/ablate_trunk.sh)
        f16x8 ah2[MT], al2[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ahn[m] = lda(((m * KC32 + (KC32 > 1 ? 1 : 0)) * 2 + 0));
            aln[m] = lda(((m * KC32 + (KC32 > 1 ? 1 : 0)) * 2 + 1));
        }
#endif
        // Timing ablations of the forward trunks (tools/ablate_density.sh; results are WRONG with any of them defined):
        //   ARAH_ABL_A_FIXED  the A fragments of chunk 0 serve every chunk (no L2 fragment stream)
        //   ARAH_ABL_B_FIXED  the B fragments of chunk 0 serve every chunk (no LDS fragment reads)
        //   ARAH_ABL_ONE_MFMA only the hi x hi product (a third of the matrix-pipe work)
#pragma unroll 1
        for (int kc = 0; kc < KC32; ++kc) {
#ifdef ARAH_ABL_A_FIXED
            const int kn = 0;
#else
            const int kn = kc + 1 < KC32 ? kc + 1 : kc;
#endif
#ifdef ARAH_A_AHEAD2
            const int k2 = kc + 2 < KC32 ? kc + 2 : KC32 - 1;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah2[m] = lda(((m * KC32 + k2) * 2 + 0));
                al2[m] = lda(((m * KC32 + k2) * 2 + 1));
            }
#else
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ahn[m] = lda(((m * KC32 + kn) * 2 + 0));
                aln[m] = lda(((m * KC32 + kn) * 2 + 1));
            }
#endif
            f16x8 bh[NT], bl[NT];
#ifdef ARAH_ABL_B_FIXED
            const int kb = 0;
#else
            const int kb = kc;
#endif
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                bh[n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + kb * 64);
                bl[n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + lo_off + kb * 64);
            }
#ifndef ARAH_ABL_ONE_MFMA
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#endif
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[m] = ahn[m];
                al[m] = aln[m];
#ifdef ARAH_A_AHEAD2
                ahn[m] = ah2[m];
                aln[m] = al2[m];
#endif
            }
        }
    }
}

// compile-time loop: the body sees its index as a constant expression (register arrays are indexed statically whatever
// the unroller's thresholds say)
template <int I>
struct IC {
    static constexpr int value = I;
};
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(IC<I>{});
        static_for<I + 1, N>(f);
    }
}

// The same product as gemm_acc_split<.., DEEP = false>, as an explicit software pipeline for kernels that run TWO waves
// per SIMD (k_density's 128-point tiles): the rolled form leaves the order to the scheduler, which sinks every A load next
// to its first use (ISA of round 3: global_load -> s_waitcnt vmcnt(0) -> v_mfma on the same registers, an L2 round trip
// exposed three times per chunk) and reads all sixteen B fragments of a chunk before its first MFMA.  Here a STAGE is
// 1/H of a chunk's N-tiles (3 MT NT/H MFMAs); while it runs, the B fragments of the next stage (2 NT/H ds_read_b128) and,
// in a chunk's first stage, the A fragments of the next chunk (2 MT global loads) are in flight, and
// sched_group_barrier pins the interleaving: one memory instruction between every two MFMAs.  Registers: two stages of B
// (4 NT/H quads) + two chunks of A (4 MT quads) next to the MT NT accumulators.
// FR: f16x8 (the f16 split) or bf16x8 (bf16 x 3, loop D's reverse sweep and colour MLP)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_frag(const f16x8 a, const f16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma_frag(const bf16x8 a, const bf16x8 b, const f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <int KC32, int MT, int NT, int H, typename FR = f16x8>
__device__ __forceinline__ void gemm_acc_split_pipe(const FR* __restrict__ wp, int mt0, const float* act, int ld,
                                                    int lo_off, f32x4 (&acc)[MT][NT], int lane) {
    constexpr int NH = NT / H, S = KC32 * H;
    const int j = lane & 15, g = lane >> 4;
    const char* bptr = reinterpret_cast<const char*>(act) + j * ld * 4 + split_slot(j, g) * 16;
    const unsigned aoff = (unsigned)(mt0 * KC32 * 2 * 64 + lane) * 16u;
    auto lda = [&](int idx) { return ld_frag<FR>(wp, aoff, idx * 1024); };
    FR ah[2][MT], al[2][MT], bh[2][NH], bl[2][NH];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ah[0][m] = lda((m * KC32) * 2 + 0);
        al[0][m] = lda((m * KC32) * 2 + 1);
    }
#pragma unroll
    for (int n = 0; n < NH; ++n) {
        bh[0][n] = *reinterpret_cast<const FR*>(bptr + n * 16 * ld * 4);
        bl[0][n] = *reinterpret_cast<const FR*>(bptr + n * 16 * ld * 4 + lo_off);
    }
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, S>([&](auto Sc) {
        constexpr int s = decltype(Sc)::value, kc = s / H, h = s % H, cb = s & 1, ca = kc & 1;
        constexpr bool more_b = s + 1 < S, more_a = h == 0 && kc + 1 < KC32;
#ifdef ARAH_ABL_A_FIXED   // timing ablations, as in gemm_acc_split (results are wrong with them)
        constexpr int kA = 0;
#else
        constexpr int kA = kc + 1;
#endif
        if constexpr (more_a) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                ah[ca ^ 1][m] = lda((m * KC32 + kA) * 2 + 0);
                al[ca ^ 1][m] = lda((m * KC32 + kA) * 2 + 1);
            }
        }
        if constexpr (more_b) {
#ifdef ARAH_ABL_B_FIXED
            constexpr int k1 = 0, h1 = (s + 1) % H;
#else
            constexpr int k1 = (s + 1) / H, h1 = (s + 1) % H;
#endif
#pragma unroll
            for (int n = 0; n < NH; ++n) {
                bh[cb ^ 1][n] = *reinterpret_cast<const FR*>(bptr + (h1 * NH + n) * 16 * ld * 4 + k1 * 64);
                bl[cb ^ 1][n] = *reinterpret_cast<const FR*>(bptr + (h1 * NH + n) * 16 * ld * 4 + lo_off + k1 * 64);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)   // small terms first
#pragma unroll
            for (int n = 0; n < NH; ++n)
                acc[m][h * NH + n] = mfma_frag(al[ca][m], bh[cb][n], acc[m][h * NH + n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NH; ++n)
                acc[m][h * NH + n] = mfma_frag(ah[ca][m], bl[cb][n], acc[m][h * NH + n]);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NH; ++n)
                acc[m][h * NH + n] = mfma_frag(ah[ca][m], bh[cb][n], acc[m][h * NH + n]);
        // order inside the stage: MFMA, load, MFMA, MFMA, load, ... (0x008 MFMA, 0x020 VMEM read, 0x100 DS read)
        constexpr int n_vm = more_a ? 2 * MT : 0, n_ds = more_b ? 2 * NH : 0, n_mf = 3 * MT * NH;
        constexpr int per = (n_vm + n_ds) > 0 ? (n_mf - 2) / (n_vm + n_ds) : n_mf;   // MFMAs between two loads
        constexpr int lead = per >= 2 ? 2 : 1;
        static_for<0, n_vm>([&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, lead > per ? per : (per >= 1 ? per : 1), 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        });
        static_for<0, n_ds>([&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, per >= 1 ? per : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, n_mf, 0);   // whatever is left
        __builtin_amdgcn_sched_barrier(0);
    });
}

// ---- bf16 x 3 engine (training backward, round 3) ---------------------------------------------------------------
// The third way to multiply fp32 operands on the matrix pipe: both operands as bf16 hi + bf16 lo, three
// v_mfma_f32_16x16x32_bf16 (lo x hi, hi x lo, hi x hi).  16 mantissa bits per operand (relative error 2^-16 per product,
// against 2^-22 for the f16 split) but the exponent range of fp32: no operand scaling, which is what the adjoints and
// deltas of a backward sweep need (their magnitudes are set by the loss, 1e-7 .. 1e+3 in one step).  The activations stay
// fp32 in LDS exactly where the fp32 engine reads them; a lane converts its eight consecutive k of a point as it loads them
// (three vector instructions per element), so a call site only swaps gemm_acc for gemm_any<true, ..>.  Rate: 3 x 16
// cycles per 16x16x32 product against 8 x 32 on the fp32 MFMA.

struct B3Nets {                 // bf16 hi/lo fragments of the packed matrices (k_b3_from_packed), same M-tile / K order
    const bf16x8* sdf_wp[5];    // W_2..W_6 (tangent pass)
    const bf16x8* sdf_wpT[5];   // their transposes (reverse sweeps)
    const bf16x8* col[6];       // colour MLP forward: w0p, w1p, w2p, w3ap, w3bp, w4p
    const bf16x8* colT[6];      // transposed: w0pT, w1pT, w2pT, w3apT, w3bpT, w4pT
};

__device__ __forceinline__ void bsplit8(const f32x4 a, const f32x4 b, bf16x8& hi, bf16x8& lo) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = i < 2 ? a[2 * i] : b[2 * i - 4], x1 = i < 2 ? a[2 * i + 1] : b[2 * i - 3];
        const bf16x2 h = __builtin_convertvector(f32x2{x0, x1}, bf16x2);
        const bf16x2 l = __builtin_convertvector(f32x2{x0 - (float)h[0], x1 - (float)h[1]}, bf16x2);
        hi[2 * i] = h[0];
        hi[2 * i + 1] = h[1];
        lo[2 * i] = l[0];
        lo[2 * i + 1] = l[1];
    }
}

// acc[m][n] += W(M-tiles mt0.., K = 16 KC16) * act(16 NT points), act fp32 rows of ld floats in LDS.
//   weights : wp[((mt*KC32 + kc)*2 + s)*64 + lane]  (s = 0 hi, 1 lo), lane (j, g) holds W[mt*16 + j][kc*32 + 8g .. +7],
//             KC32 = ceil(KC16 / 2), zero beyond K
template <int KC16, int MT, int NT = kNT>
__device__ __forceinline__ void gemm_acc_b3(const bf16x8* __restrict__ wp, int mt0, const float* act, int ld,
                                            f32x4 (&acc)[MT][NT], int lane) {
    constexpr int KC32 = (KC16 + 1) / 2;
    const int j = lane & 15, g = lane >> 4;
    const unsigned aoff = (unsigned)(mt0 * KC32 * 2 * 64 + lane) * 16u;
    auto lda = [&](int idx) { return ld_frag<bf16x8>(wp, aoff, idx * 1024); };
    const bool tail_ok = (KC16 & 1) == 0 || g < 2;   // odd KC16: the last 32-chunk holds 16 columns
    const float* brow = act + j * ld;
    bf16x8 ah[MT], al[MT], ahn[MT], aln[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ah[m] = lda((m * KC32) * 2 + 0);
        al[m] = lda((m * KC32) * 2 + 1);
    }
#pragma unroll 2
    for (int kc = 0; kc < KC32; ++kc) {
        const int kn = kc + 1 < KC32 ? kc + 1 : kc;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ahn[m] = lda((m * KC32 + kn) * 2 + 0);
            aln[m] = lda((m * KC32 + kn) * 2 + 1);
        }
        const bool ok = kc + 1 < KC32 || tail_ok;
        const float* bp = brow + (ok ? kc * 32 + 8 * g : 0);   // never read past the row
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 x0 = *reinterpret_cast<const f32x4*>(bp + n * 16 * ld);
            f32x4 x1 = *reinterpret_cast<const f32x4*>(bp + n * 16 * ld + 4);
            if (!ok) x0 = x1 = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 bh, bl;
            bsplit8(x0, x1, bh, bl);
#pragma unroll
            for (int m = 0; m < MT; ++m) {   // small terms first
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl, acc[m][n], 0, 0, 0);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh, acc[m][n], 0, 0, 0);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[m] = ahn[m];
            al[m] = aln[m];
        }
    }
}

// one call site, two engines
template <bool B3, int KC16, int MT, int NT = kNT>
__device__ __forceinline__ void gemm_any(const float* __restrict__ wp, const bf16x8* __restrict__ wpb, int mt0, const float* act,
                                         int ld, f32x4 (&acc)[MT][NT], int lane) {
    if constexpr (B3) gemm_acc_b3<KC16, MT, NT>(wpb, mt0, act, ld, acc, lane);
    else gemm_acc<KC16, MT, NT>(wp, mt0, act, ld, acc, lane);
}

// ---- bf16 x 3 with the activations as bf16 hi / lo PLANES in LDS (loop D, round 3) ------------------------------------
// gemm_acc_b3 converts a wave's B fragments as it loads them -- eight waves convert the same tile, and the conversion (three
// vector instructions per element) then costs as much as the MFMAs.  Here the producer's epilogue writes the two bf16
// planes once (the layout of the f16 split: 64-byte chunks of 32 channels, slots permuted by the point, lo plane at
// lo_off bytes) and the product is MFMAs and LDS reads only.
__device__ __forceinline__ void store_bsplit4(float* act, int ld, int lo_off, int pt, int ch0, const f32x4 v) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hi, lo;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const bf16x2 h = __builtin_convertvector(f32x2{v[2 * p], v[2 * p + 1]}, bf16x2);
        const bf16x2 l = __builtin_convertvector(f32x2{v[2 * p] - (float)h[0], v[2 * p + 1] - (float)h[1]}, bf16x2);
        hi[p] = __builtin_bit_cast(unsigned, h);
        lo[p] = __builtin_bit_cast(unsigned, l);
    }
    char* row = reinterpret_cast<char*>(act) + pt * ld * 4 + split_byte(pt, ch0);
    *reinterpret_cast<u32x2*>(row) = hi;
    *reinterpret_cast<u32x2*>(row + lo_off) = lo;
}
__device__ __forceinline__ void store_bsplit1(float* act, int ld, int lo_off, int pt, int ch, float v) {
    const __bf16 h = (__bf16)v;
    char* row = reinterpret_cast<char*>(act) + pt * ld * 4 + split_byte(pt, ch);
    *reinterpret_cast<__bf16*>(row) = h;
    *reinterpret_cast<__bf16*>(row + lo_off) = (__bf16)(v - (float)h);
}
__device__ __forceinline__ float load_bsplit(const float* act, int ld, int lo_off, int pt, int ch) {
    const char* row = reinterpret_cast<const char*>(act) + pt * ld * 4 + split_byte(pt, ch);
    return (float)*reinterpret_cast<const __bf16*>(row) + (float)*reinterpret_cast<const __bf16*>(row + lo_off);
}

// acc[m][n] += W(M-tiles mt0.., KC32 32-chunks) * planes(16 NT points).  Loop D runs two waves per SIMD: the explicit
// pipeline of gemm_acc_split_pipe, half a chunk's N-tiles per stage.  (Round 3's rolled form -- B read at the top of its
// chunk, the A loads sunk next to their use -- cost the reverse sweep 1.8 x the forward trunk's time for the same MFMAs;
// fully unrolled with both operands a chunk ahead: 85.0 -> 67.8 ms per shade-everything frame together with the buffer
// loads of ld_frag; the pinned interleaving another 1.2 ms: profiles/r04_shade_phases.txt.)  Same MFMA order per accumulator.
template <int KC32, int MT, int NT = kNT>
__device__ __forceinline__ void gemm_acc_bsplit(const bf16x8* __restrict__ wp, int mt0, const float* act, int ld, int lo_off,
                                                f32x4 (&acc)[MT][NT], int lane) {
#ifndef ARAH_BSPLIT_ROLLED
    gemm_acc_split_pipe<KC32, MT, NT, NT % 2 == 0 ? 2 : 1, bf16x8>(wp, mt0, act, ld, lo_off, acc, lane);
#else   // round 3's form (A/B reference for the measurement)
    const int j = lane & 15, g = lane >> 4;
    const char* bptr = reinterpret_cast<const char*>(act) + j * ld * 4 + split_slot(j, g) * 16;
    const unsigned aoff = (unsigned)(mt0 * KC32 * 2 * 64 + lane) * 16u;
    auto lda = [&](int idx) { return ld_frag<bf16x8>(wp, aoff, idx * 1024); };
    bf16x8 ah[MT], al[MT], ahn[MT], aln[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ah[m] = lda((m * KC32) * 2 + 0);
        al[m] = lda((m * KC32) * 2 + 1);
    }
#pragma unroll 2
    for (int kc = 0; kc < KC32; ++kc) {
        const int kn = kc + 1 < KC32 ? kc + 1 : kc;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ahn[m] = lda((m * KC32 + kn) * 2 + 0);
            aln[m] = lda((m * KC32 + kn) * 2 + 1);
        }
        bf16x8 bh[NT], bl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            bh[n] = *reinterpret_cast<const bf16x8*>(bptr + n * 16 * ld * 4 + kc * 64);
            bl[n] = *reinterpret_cast<const bf16x8*>(bptr + n * 16 * ld * 4 + lo_off + kc * 64);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)   // small terms first
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[n], acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[m] = ahn[m];
            al[m] = aln[m];
        }
    }
#endif
}

// 4 consecutive channels of one point, already multiplied by kActScale -> hi/lo planes
// hi = f16(h) (round to nearest even), lo = f16(h - hi).  The residual is one mixed-precision FMA per channel that
// reads its f16 operand straight out of the packed hi register and writes the f16 result into its half of the packed
// lo register (v_fma_mixlo_f16 / v_fma_mixhi_f16: h * 1.0 - hi, exact in fp32, one rounding to f16): three VALU
// operations per pair of channels, against seven for convert / convert back / subtract / convert.  Same bits.
// Round 4: written in C, not asm.  gfx940-class parts need wait states around VALU instructions that write half a
// register (a VALU or MFMA reading the word next; an in-flight MFMA still reading the registers the word lands in), and the
// compiler cannot see into an asm statement: with round 4's shorter loop-C epilogue an MFMA came to sit right behind a
// v_fma_mixhi_f16 often enough for 6 % of the F1 points to miss their root, depending on the schedule
// (profiles/r04_hazard_ubench.txt).  Built without packed fp32 (the library always is), instruction selection turns
// fptrunc(fma(a, k, -fpext(hi))) into exactly the two mixed-precision FMAs, provided it cannot fold the multiplier away:
// `one` is 1.0 in a scalar register it cannot see through.
__device__ __forceinline__ float opaque_one() {
    float one;
    asm("s_mov_b32 %0, 1.0" : "=s"(one));
    return one;
}
__device__ __forceinline__ unsigned split_residual2(float a, float b, unsigned hi_pk) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    const float one = opaque_one();
    const f16x2 h = __builtin_bit_cast(f16x2, hi_pk);
    f16x2 l;
    l[0] = (_Float16)__builtin_fmaf(a, one, -(float)h[0]);
    l[1] = (_Float16)__builtin_fmaf(b, one, -(float)h[1]);
    return __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void store_split4(float* act, int ld, int lo_off, int pt, int ch0, const f32x4 hs) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 hi, lo;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const f16x2 h2 = __builtin_convertvector(f32x2{hs[2 * p], hs[2 * p + 1]}, f16x2);
        hi[p] = __builtin_bit_cast(unsigned, h2);
        lo[p] = split_residual2(hs[2 * p], hs[2 * p + 1], hi[p]);
    }
    char* row = reinterpret_cast<char*>(act) + pt * ld * 4 + split_byte(pt, ch0);
    *reinterpret_cast<u32x2*>(row) = hi;
    *reinterpret_cast<u32x2*>(row + lo_off) = lo;
}

// 8 fp32 values -> the hi and lo B fragments of a 16x16x32 step, in registers (same arithmetic as store_split4)
__device__ __forceinline__ void split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 h, l;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f16x2 h2 = __builtin_convertvector(f32x2{v[2 * p], v[2 * p + 1]}, f16x2);
        h[p] = __builtin_bit_cast(unsigned, h2);
        l[p] = split_residual2(v[2 * p], v[2 * p + 1], h[p]);
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}

// 4 fp32 values -> two packed words of hi halves and two of lo halves (half a split8)
__device__ __forceinline__ void split4(const float (&v)[4], unsigned& h0, unsigned& h1, unsigned& l0, unsigned& l1) {
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[0], v[1]}, f16x2));
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v[2], v[3]}, f16x2));
    l0 = split_residual2(v[0], v[1], h0);
    l1 = split_residual2(v[2], v[3], h1);
}

__device__ __forceinline__ float load_split(const float* act, int ld, int lo_off, int pt, int ch) {
    const char* row = reinterpret_cast<const char*>(act) + pt * ld * 4 + split_byte(pt, ch);
    return ((float)*reinterpret_cast<const _Float16*>(row) + (float)*reinterpret_cast<const _Float16*>(row + lo_off)) * kInvActScale;
}

// One 16x16 output tile (M-tile mt, N-tile nt) on the split engine: narrow output layers split over waves.
// The A fragments are a separate step so that the caller can issue their loads (L2 latency) ahead of the barrier and
// the epilogue that precede the layer; three independent accumulators (one per product class) instead of one chain
// of 3 KC32 dependent MFMAs.
template <int KC32>
struct SplitA {
    f16x8 hi[KC32], lo[KC32];
};
template <int KC32>
__device__ __forceinline__ SplitA<KC32> load_split_a(const f16x8* __restrict__ wp, int mt, int lane) {
    const unsigned aoff = (unsigned)(mt * KC32 * 2 * 64 + lane) * 16u;
    SplitA<KC32> a;
#pragma unroll
    for (int kc = 0; kc < KC32; ++kc) {
        a.hi[kc] = ld_frag<f16x8>(wp, aoff, (kc * 2 + 0) * 1024);
        a.lo[kc] = ld_frag<f16x8>(wp, aoff, (kc * 2 + 1) * 1024);
    }
    return a;
}
template <int KC32>
__device__ __forceinline__ f32x4 gemm_one_split(const SplitA<KC32>& a, int nt, const float* act, int ld, int lo_off,
                                                int lane) {
    const int j = lane & 15, g = lane >> 4;
    const char* bptr = reinterpret_cast<const char*>(act) + (nt * 16 + j) * ld * 4 + split_slot(j, g) * 16;
    f16x8 bh[KC32], bl[KC32];
#pragma unroll
    for (int kc = 0; kc < KC32; ++kc) {
        bh[kc] = *reinterpret_cast<const f16x8*>(bptr + kc * 64);
        bl[kc] = *reinterpret_cast<const f16x8*>(bptr + lo_off + kc * 64);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kc = 0; kc < KC32; ++kc) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.lo[kc], bh[kc], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[kc], bl[kc], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a.hi[kc], bh[kc], acc2, 0, 0, 0);
    }
    return (acc0 + acc1) + acc2;
}

// ------------------------------------------------------------------------------------------
// SDF network: FiLM-SIREN 3 -> 256 x6 -> 1
// ------------------------------------------------------------------------------------------
struct SdfNet {
    const float* w0;       // [256][4]
    const float* wp[5];    // packed 256x256
    const float* wpT[5];   // packed transposed
    const float* w6;       // [256]
    const float* bias;     // [6][256]
    const float* freq;     // [6][256]
    const float* phase;    // [6][256]
    const float* b6;       // [1]
    const float* fw;       // [6][256]  f kFilmScale             (z = 2 pi (fw v + pw) by default, v = W h)
    const float* pw;       // [6][256]  (f b + phi) kFilmScale
    const float* fws;      // [6][256]  fw / (weight scale * kActScale) of the split layers (row 0 = fw)
    const f16x8* wps[5];   // split-packed 256x256
};

constexpr int kSdfLd = 264;   // 256 + 8: rows 8 dwords apart mod 64 (see split_slot)
constexpr int kSdfMT = 2;     // 16 M-tiles / 8 waves

// FiLM-SIREN argument z = 30 (f (v + b) + phi) = kFilmTurn (fw v + pw): the folded constants fw, pw (k_fold_film) carry z
// in the unit the sine below wants.  Default: REVOLUTIONS, z = 2 pi w, and the hardware sine / cosine of the reduced
// argument, v_sin_f32(v_fract_f32(w)) = sin(2 pi frac(w)) -- measured on the MI355X against a double-precision sine over
// 600 half-revolutions (tools/ubench/hw_sin_accuracy.hip, profiles/r02_hw_sin_accuracy.txt): max |error| 1.24e-7, rms
// 3.46e-8, i.e. the rounding of the fp32 argument and result and nothing else; three VALU operations instead of eleven.
// -DARAH_POLY_SINE: round 1's branch-free polynomial on HALF-revolutions (q = rint(w), r = w - q in [-1/2, 1/2] exactly,
// sin(pi w) = (-1)^q r S(r^2), cos(pi w) = (-1)^q C(r^2) with minimax S, C; max |error| 1.71e-7, rms 3.67e-8).
#ifndef ARAH_POLY_SINE
constexpr double kFilmScale = 30.0 / 6.28318530717958647692;   // fw = f kFilmScale, pw = (f b + phi) kFilmScale
__device__ __forceinline__ float sinpi_amp(float w, float amp) {
    return __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(w)) * amp;
}
__device__ __forceinline__ void sincospi_amp(float w, float amp, float& s, float& c) {
    const float t = __builtin_amdgcn_fractf(w);
    s = __builtin_amdgcn_sinf(t) * amp;
    c = __builtin_amdgcn_cosf(t);
}
#else
constexpr double kFilmScale = 30.0 / 3.14159265358979323846;
__device__ __forceinline__ float sinpi_amp(float w, float amp) {
    const float q = rintf(w);
    const float r = w - q;
    const float r2 = r * r;
    float p = fmaf(r2, 0.0772201280771219f * amp, -0.5980451736306471f * amp);
    p = fmaf(p, r2, 2.550031377188653f * amp);
    p = fmaf(p, r2, -5.167706878920042f * amp);
    p = fmaf(p, r2, 3.1415925800446054f * amp);
    const unsigned sgn = (unsigned)(int)q << 31;
    return __uint_as_float(__float_as_uint(p * r) ^ sgn);
}

__device__ __forceinline__ void sincospi_amp(float w, float amp, float& s, float& c) {
    const float q = rintf(w);
    const float r = w - q;
    const float r2 = r * r;
    float p = fmaf(r2, 0.0772201280771219f * amp, -0.5980451736306471f * amp);
    p = fmaf(p, r2, 2.550031377188653f * amp);
    p = fmaf(p, r2, -5.167706878920042f * amp);
    p = fmaf(p, r2, 3.1415925800446054f * amp);
    float pc = fmaf(r2, -0.02439671639064982f, 0.23493755882145678f);
    pc = fmaf(pc, r2, -1.335212056639698f);
    pc = fmaf(pc, r2, 4.058709164340391f);
    pc = fmaf(pc, r2, -4.9348021372282975f);
    pc = fmaf(pc, r2, 0.9999999997806512f);
    const unsigned sgn = (unsigned)(int)q << 31;
    s = __uint_as_float(__float_as_uint(p * r) ^ sgn);
    c = __uint_as_float(__float_as_uint(pc) ^ sgn);
}
#endif

// FiLM-SIREN activation of 4 channels: h = amp sin(z), z = 30 (f (v + b) + phi) = pi (fw v + pw);
// GRAD: d = dh/dv / amp = 30 f cos(z)
template <bool GRAD>
__device__ __forceinline__ void film_sine(const f32x4 v, const f32x4 fw, const f32x4 pw, const f32x4 f, float amp,
                                          f32x4& h, f32x4& d) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float w = fmaf(v[r], fw[r], pw[r]);
        if (GRAD) {
            float sn, c;
            sincospi_amp(w, amp, sn, c);
            h[r] = sn;
            d[r] = c * (30.0f * f[r]);
        } else {
            h[r] = sinpi_amp(w, amp);
        }
    }
}

// Training taps (csrc/train.hpp): what the hand-written backward of loop D needs from the forward pass.
//   NoTap     : nothing (inference).
//   TrainTap  : the pre-activations v_k = W_k h_{k-1} of all six layers go to a per-workgroup slab in fragment
//               order (a software register spill, read back lane by lane), and the layer inputs h_0 .. h_5 are
//               streamed to HBM as dense [P][width] matrices (operands of the weight-gradient GEMMs).
struct NoTap {
    static constexpr bool on = false;
};
struct TrainTap {
    static constexpr bool on = true;
    f32x4* aslab;        // [6][kWaves][kSdfMT*kNT][64] pre-activations
    float* h[6];         // h[0]: [P][4] (x), h[1..5]: [P][256]
    long long row0;      // first sample of the tile
    int rows;            // valid samples in the tile (<= 64)
};

// rows [0, rows) of an LDS matrix (row stride ld floats, `width` floats wide, width % 4 == 0) -> dense global rows
__device__ __forceinline__ void stream_rows(const float* lds, int ld, int width, float* dst, long long row0, int rows,
                                            int tid) {
    const int w4 = width >> 2;
    for (int e = tid; e < rows * w4; e += kThreads) {
        const int r = e / w4, c4 = e - r * w4;
        reinterpret_cast<f32x4*>(dst + (row0 + r) * width)[c4] = *reinterpret_cast<const f32x4*>(lds + r * ld + c4 * 4);
    }
}

// the same from split planes (hi at byte 0, lo at byte 512, 256 channels)
__device__ __forceinline__ void stream_rows_split(const float* lds, int ld, float* dst, long long row0, int rows, int tid) {
    for (int e = tid; e < rows * 64; e += kThreads) {
        const int r = e >> 6, c4 = e & 63;
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = load_split(lds, ld, 512, r, c4 * 4 + q);
        reinterpret_cast<f32x4*>(dst + (row0 + r) * 256)[c4] = v;
    }
}

// Forward trunk on a tile of 16 NT points.  xin: LDS [16 NT][4] normalised coords.  act: LDS rows of ld floats,
// receives h6 -- as fp32 [256] (exact engine) or as split planes (hi at byte 0, lo at byte 512; SPLIT).
// GRAD: dact factors of layers 1..5 go to `spill` (global, this workgroup's private slab of
// 5*8*8*64 f32x4), layer 6's stay in `dlast`.
// pc: phase clocks of an instrumented build (tools/phase_clocks.py), nothing otherwise
#define ARAH_PC_MARK(i) do { if (pc) pc->mark(i); } while (0)
// PIPE > 0 (k_shade, two waves per SIMD): the gradient trunk's products on the explicit pipeline too, NT / PIPE N-tiles per stage
// KEEP0 = false (k_shade's bf16 x 3 sweep): the first layer's derivative factors are not written to the slab, the sweep
// recomputes them from x (layer0_dfactor: three FMAs and a sine/cosine per channel, the same instructions, the same bits)
template <bool GRAD, int NT = kNT, bool SPLIT = false, typename TAP = NoTap, typename CLK = NoClk, int PIPE = 0, bool KEEP0 = true>
__device__ __forceinline__ void sdf_trunk(const SdfNet& net, const float* xin, float* act, int ld, f32x4* spill,
                                          f32x4 (&dlast)[kSdfMT][NT], int wave, int lane, const TAP& tap = TAP(),
                                          CLK* pc = nullptr) {
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
    constexpr float amp = SPLIT ? kActScale : 1.0f;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // layer 1: K = 3 on the vector ALU, same accumulator ownership as the MFMA layers
    {
        f32x4 x[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) x[n] = *reinterpret_cast<const f32x4*>(xin + (n * 16 + j) * 4);
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
            const f32x4 f = GRAD ? *reinterpret_cast<const f32x4*>(net.freq + ch0) : zero4;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 v, h, d;
#pragma unroll
                for (int r = 0; r < 4; ++r)   // explicit chain: every instantiation must round identically
                    v[r] = fmaf(w[r][2], x[n][2], fmaf(w[r][1], x[n][1], w[r][0] * x[n][0]));
                no_pack(v);
                if constexpr (TAP::on) tap.aslab[((0 * kWaves + wave) * (kSdfMT * NT) + m * NT + n) * 64 + lane] = v;
                film_sine<GRAD>(v, fw, pw, f, amp, h, d);
                if (SPLIT) store_split4(act, ld, 512, n * 16 + j, ch0, h);
                else *reinterpret_cast<f32x4*>(act + (n * 16 + j) * ld + ch0) = h;
#ifndef ARAH_ABL_NO_SLAB
                if (GRAD && KEEP0) spill[((0 * kWaves + wave) * (kSdfMT * NT) + m * NT + n) * 64 + lane] = d;
#endif
            }
        }
    }
    if constexpr (TAP::on) stream_rows(xin, 4, 4, tap.h[0], tap.row0, tap.rows, wave * 64 + lane);
    ARAH_SYNC();
#pragma unroll 1
    for (int k = 1; k < 6; ++k) {
        f32x4 acc[kSdfMT][NT];
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) zero_acc(acc[m][n]);
        if constexpr (TAP::on) {   // h_k: read only
            if (SPLIT) stream_rows_split(act, ld, tap.h[k], tap.row0, tap.rows, wave * 64 + lane);
            else stream_rows(act, ld, 256, tap.h[k], tap.row0, tap.rows, wave * 64 + lane);
        }
#ifndef ARAH_TRUNK_ROLLED   // 128-point tiles (two waves per SIMD): the explicit pipeline, 12.3 vs 13.1 ms for k_density (r3o)
        if constexpr (SPLIT && !GRAD && NT == 8) gemm_acc_split_pipe<8, kSdfMT, NT, 2>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);
        else
#endif
        if constexpr (SPLIT && PIPE > 0) gemm_acc_split_pipe<8, kSdfMT, NT, PIPE>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);
        else
        if (SPLIT) gemm_acc_split<8, kSdfMT, NT, GRAD>(net.wps[k - 1], mt0, act, ld, 512, acc, lane);   // GRAD kernels: 2 waves/SIMD
        else gemm_acc<16, kSdfMT, NT>(net.wp[k - 1], mt0, act, ld, acc, lane);
        ARAH_PC_MARK(7);
        // the epilogue's per-channel constants travel (L2 latency) while the workgroup gathers at the barrier -- in the
        // forward-only kernels and in k_shade (PIPE; two waves per SIMD); the other gradient kernels sit at their VGPR cap
        // and load them afterwards
        constexpr bool kEarly = !GRAD || (SPLIT && PIPE > 0);
        f32x4 fwm[kSdfMT], pwm[kSdfMT], fm[kSdfMT];
        if constexpr (kEarly) {
#pragma unroll
            for (int m = 0; m < kSdfMT; ++m) {
                const int ch0 = (mt0 + m) * 16 + 4 * g;
                fwm[m] = *reinterpret_cast<const f32x4*>((SPLIT ? net.fws : net.fw) + k * 256 + ch0);
                pwm[m] = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                if constexpr (GRAD) fm[m] = *reinterpret_cast<const f32x4*>(net.freq + k * 256 + ch0);
            }
        }
        ARAH_SYNC();   // everyone is done reading the layer input
        ARAH_PC_MARK(9);
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 fw, pw, f = zero4;
            if constexpr (!kEarly) {
                fw = *reinterpret_cast<const f32x4*>((SPLIT ? net.fws : net.fw) + k * 256 + ch0);
                pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                f = *reinterpret_cast<const f32x4*>(net.freq + k * 256 + ch0);
            } else {
                fw = fwm[m];
                pw = pwm[m];
                if constexpr (GRAD) f = fm[m];
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                f32x4 h, d;
                if constexpr (TAP::on)   // the pre-activation v_k itself: the split engine's accumulator carries the operand scales
                    tap.aslab[((k * kWaves + wave) * (kSdfMT * NT) + m * NT + n) * 64 + lane] =
                        SPLIT ? acc[m][n] * net.fws[6 * 256 + k] : acc[m][n];
#ifdef ARAH_ABL_NO_EPI   //   ARAH_ABL_NO_EPI   no FiLM sine: the accumulators are stored as they are
                h = acc[m][n] * 1e-6f;
                d = zero4;
#else
                film_sine<GRAD>(acc[m][n], fw, pw, f, amp, h, d);
#endif
                if (SPLIT) store_split4(act, ld, 512, n * 16 + j, ch0, h);
                else *reinterpret_cast<f32x4*>(act + (n * 16 + j) * ld + ch0) = h;
                if (GRAD) {
#ifdef ARAH_ABL_NO_SLAB   //   ARAH_ABL_NO_SLAB   timing only: the derivative factors neither leave nor come back
                    dlast[m][n] = d;
#else
                    if (k < 5) spill[((k * kWaves + wave) * (kSdfMT * NT) + m * NT + n) * 64 + lane] = d;
                    else dlast[m][n] = d;
#endif
                }
            }
        }
        ARAH_PC_MARK(8);
        ARAH_SYNC();
        ARAH_PC_MARK(9);
    }
}

// Forward trunk of a tile of 2 x NTH x 16 points on the split engine with the two halves of the tile a phase apart: while
// the MFMAs of one half's layer issue, the FiLM-sine epilogue of the other half's previous GEMM rides between them (one
// 16 x 16 accumulator group per 32-chunk), so that matrix pipe and vector ALU work in the same phase instead of taking
// turns between workgroup barriers.  One barrier per phase, two phases per layer -- the barrier count of sdf_trunk.
//   phase (H0, k): GEMM of half 0, layer k   ||  epilogue of half 1, layer k - 1  (writes half 1's rows)
//   phase (H1, k): GEMM of half 1, layer k   ||  epilogue of half 0, layer k      (writes half 0's rows)
// A barrier separates a half's GEMM (which reads its rows) from the epilogue that rewrites them in place, and the
// epilogue from the next GEMM that reads them.  Same MFMA order per accumulator and the same epilogue arithmetic as
// sdf_trunk: bit-identical results (the density pass and the shading pass must agree on every sample).
template <typename Tp, Tp V>
struct PpConst {
    static constexpr Tp value = V;
};
template <int NTH>
__device__ __forceinline__ void sdf_trunk_pp(const SdfNet& net, const float* xin, float* act, int ld, int wave, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
    constexpr float amp = kActScale;
    constexpr int NT = 2 * NTH;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    {   // layer 1: K = 3 on the vector ALU, both halves (as sdf_trunk)
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            f32x4 w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const f32x4 x = *reinterpret_cast<const f32x4*>(xin + (n * 16 + j) * 4);
                f32x4 v, h, d;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], x[2], fmaf(w[r][1], x[1], w[r][0] * x[0]));
                no_pack(v);
                film_sine<false>(v, fw, pw, zero4, amp, h, d);
                store_split4(act, ld, 512, n * 16 + j, ch0, h);
            }
        }
    }
    ARAH_SYNC();
    f32x4 acc[2][kSdfMT][NTH];   // [half][m][n]
    // one phase: acc[hg] = W_k * rows of half hg; EPI: the epilogue of layer ke on acc[he] -> rows of half he
    auto phase = [&](auto hgc, int k, auto epic, int ke) {
        constexpr int hg = decltype(hgc)::value, he = 1 - hg;
        constexpr bool EPI = decltype(epic)::value;
        const f16x8* wp = net.wps[k - 1];
        const unsigned aoff = (unsigned)(mt0 * 8 * 2 * 64 + lane) * 16u;
        auto lda = [&](int idx) { return ld_frag<f16x8>(wp, aoff, idx * 1024); };
        const char* bptr = reinterpret_cast<const char*>(act) + (hg * NTH * 16 + j) * ld * 4 + split_slot(j, g) * 16;
        f32x4 fwm[kSdfMT], pwm[kSdfMT];
        if constexpr (EPI) {
#pragma unroll
            for (int m = 0; m < kSdfMT; ++m) {
                const int ch0 = (mt0 + m) * 16 + 4 * g;
                fwm[m] = *reinterpret_cast<const f32x4*>(net.fws + ke * 256 + ch0);
                pwm[m] = *reinterpret_cast<const f32x4*>(net.pw + ke * 256 + ch0);
            }
        }
        // explicit pipeline as in gemm_acc_split_pipe, a stage = half a 32-chunk's N-tiles (3 kSdfMT NTH / 2 MFMAs): the
        // next stage's B fragments and (first half) the next chunk's A fragments are requested while the stage's MFMAs
        // run, and the epilogue group of the other half-tile rides between them: FiLM sine in a chunk's first stage,
        // hi/lo split and the two LDS stores in its second
        constexpr int NQ = NTH / 2;
        f16x8 ah[2][kSdfMT], al[2][kSdfMT], bh[2][NQ], bl[2][NQ];
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            ah[0][m] = lda((m * 8) * 2 + 0);
            al[0][m] = lda((m * 8) * 2 + 1);
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            bh[0][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4);
            bl[0][n] = *reinterpret_cast<const f16x8*>(bptr + n * 16 * ld * 4 + 512);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 hcur = zero4;
        static_for<0, 16>([&](auto Sc) {
            constexpr int st = decltype(Sc)::value, kc = st >> 1, hh = st & 1, cb = st & 1, ca = kc & 1;
            constexpr bool more_b = st + 1 < 16, more_a = hh == 0 && kc + 1 < 8;
            if constexpr (more_a) {
#pragma unroll
                for (int m = 0; m < kSdfMT; ++m) {
                    ah[ca ^ 1][m] = lda((m * 8 + kc + 1) * 2 + 0);
                    al[ca ^ 1][m] = lda((m * 8 + kc + 1) * 2 + 1);
                }
            }
            if constexpr (more_b) {
                constexpr int k1 = (st + 1) >> 1, h1 = (st + 1) & 1;
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    bh[cb ^ 1][n] = *reinterpret_cast<const f16x8*>(bptr + (h1 * NQ + n) * 16 * ld * 4 + k1 * 64);
                    bl[cb ^ 1][n] = *reinterpret_cast<const f16x8*>(bptr + (h1 * NQ + n) * 16 * ld * 4 + 512 + k1 * 64);
                }
            }
#pragma unroll
            for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
                for (int n = 0; n < NQ; ++n)
                    acc[hg][m][hh * NQ + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[ca][m], bh[cb][n], kc == 0 ? zero4 : acc[hg][m][hh * NQ + n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
                for (int n = 0; n < NQ; ++n)
                    acc[hg][m][hh * NQ + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ca][m], bl[cb][n], acc[hg][m][hh * NQ + n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
                for (int n = 0; n < NQ; ++n)
                    acc[hg][m][hh * NQ + n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[ca][m], bh[cb][n], acc[hg][m][hh * NQ + n], 0, 0, 0);
            if constexpr (EPI) {   // accumulator group kc of the other half: kSdfMT * NTH == 8 groups, one per chunk
                static_assert(kSdfMT * NTH == 8, "one epilogue group per 32-chunk");
                constexpr int m = kc / NTH, n = kc % NTH;
                if constexpr (hh == 0) {
                    f32x4 d;
                    film_sine<false>(acc[he][m][n], fwm[m], pwm[m], zero4, amp, hcur, d);
                } else {
                    store_split4(act, ld, 512, (he * NTH + n) * 16 + j, (mt0 + m) * 16 + 4 * g, hcur);
                }
            }
            // 0x008 MFMA, 0x002 VALU, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
            constexpr int n_vm = more_a ? 2 * kSdfMT : 0, n_ds = more_b ? 2 * NQ : 0, n_mf = 3 * kSdfMT * NQ;
            static_for<0, n_vm>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                if constexpr (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            });
            static_for<0, n_ds>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if constexpr (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            });
            static_for<0, n_mf - n_vm - n_ds>([&](auto) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if constexpr (EPI) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            });
            if constexpr (EPI) {
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    typedef PpConst<int, 0> H0_;
    typedef PpConst<int, 1> H1_;
    typedef PpConst<bool, true> T_;
    typedef PpConst<bool, false> F_;
    phase(H0_{}, 1, F_{}, 0);
    ARAH_SYNC();
#pragma unroll 1
    for (int k = 1; k < 6; ++k) {
        phase(H1_{}, k, T_{}, k);          // GEMM (H1, k) || epilogue (H0, k)
        ARAH_SYNC();
        if (k < 5) {
            phase(H0_{}, k + 1, T_{}, k);  // GEMM (H0, k + 1) || epilogue (H1, k)
            ARAH_SYNC();
        }
    }
    {   // epilogue (H1, 5)
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
            const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fws + 5 * 256 + ch0);
            const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + 5 * 256 + ch0);
#pragma unroll
            for (int n = 0; n < NTH; ++n) {
                f32x4 h, d;
                film_sine<false>(acc[1][m][n], fw, pw, zero4, amp, h, d);
                store_split4(act, ld, 512, (NTH + n) * 16 + j, ch0, h);
            }
        }
    }
    ARAH_SYNC();
}

// split planes -> fp32 in place for the first n_pts rows (through registers: the fp32 row overlays both planes)
__device__ __forceinline__ void unsplit_rows(float* act, int ld, int tid) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int e = tid + i * kThreads;
        v[i] = load_split(act, ld, 512, e >> 8, e & 255);
    }
    ARAH_SYNC();
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        const int e = tid + i * kThreads;
        act[(e >> 8) * ld + (e & 255)] = v[i];
    }
}

// sdf[pt] = w6 . h6[pt] + b6  -> out[pt*ostride]; 8 threads per point (the first 8*n_pts threads work).
template <bool SPLIT = false>
__device__ __forceinline__ void sdf_head(const SdfNet& net, const float* act, int ld, float* out, int ostride,
                                         int tid, int n_pts = kTile) {
    const int part = tid & 7;
#ifdef ARAH_REG_TRUNK   // experiment builds only (regtrunk.hpp, profiles/r05_reg_trunk.txt): the shipped head is the loop below
    if constexpr (SPLIT) {
        // The split engine's head sums in the order a POINT-OWNING wave can follow without leaving its registers
        // (regtrunk.hpp: lane group g of a point holds the units 32 q + 8 g + e): 32 chains (g, e) per point, each over
        // q = 0 .. 7 with two fmas per unit -- the hi half, then the lo half, against w6 / 1024 (exact) -- then the tree over e
        // and (t0 + t1) + (t2 + t3) over g.  Every kernel of the engine comes through here, so they all agree bit for bit.
        for (int pt = tid >> 3; pt < n_pts; pt += kThreads / 8) {   // whole waves drop out: n_pts is a multiple of 8
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 8; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * q + 8 * g + part;
                    const float w = net.w6[ch] * kInvActScale;
                    const char* row = reinterpret_cast<const char*>(act) + pt * ld * 4 + split_byte(pt, ch);
                    s[g] = fmaf(w, (float)*reinterpret_cast<const _Float16*>(row), s[g]);
                    s[g] = fmaf(w, (float)*reinterpret_cast<const _Float16*>(row + 512), s[g]);
                }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                s[g] += __shfl_xor(s[g], 1);
                s[g] += __shfl_xor(s[g], 2);
                s[g] += __shfl_xor(s[g], 4);
            }
            if (part == 0) out[pt * ostride] = ((s[0] + s[1]) + (s[2] + s[3])) + net.b6[0];
        }
        return;
    }
#endif
    for (int pt = tid >> 3; pt < n_pts; pt += kThreads / 8) {   // whole waves drop out: n_pts is a multiple of 8
        float s = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int ch = part + 8 * i;
            s = fmaf(net.w6[ch], SPLIT ? load_split(act, ld, 512, pt, ch) : act[pt * ld + ch], s);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        if (part == 0) out[pt * ostride] = s + net.b6[0];
    }
}

// Reverse sweep for d sdf / d x on the tile (needs sdf_trunk<true> first).
// bwd: LDS [64][ld] scratch (must not alias the feature buffer).  grad -> out[pt*ostride + 1..3].
template <bool B3 = false>
__device__ __forceinline__ void sdf_backward(const SdfNet& net, float* bwd, int ld, const f32x4* spill,
                                             const f32x4 (&dlast)[kSdfMT][kNT], float* out, int ostride, int wave,
                                             int lane, int tid, const B3Nets* b3 = nullptr) {
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
#pragma unroll
    for (int m = 0; m < kSdfMT; ++m) {
        const int ch0 = (mt0 + m) * 16 + 4 * g;
        const f32x4 w = *reinterpret_cast<const f32x4*>(net.w6 + ch0);
#pragma unroll
        for (int n = 0; n < kNT; ++n) *reinterpret_cast<f32x4*>(bwd + (n * 16 + j) * ld + ch0) = dlast[m][n] * w;
    }
    ARAH_SYNC();
#pragma unroll 1
    for (int k = 4; k >= 0; --k) {   // g_k+1 (layer index k, 0-based) = W_{k+2}^T u_{k+2}
        f32x4 acc[kSdfMT][kNT];
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        gemm_any<B3, 16, kSdfMT>(net.wpT[k], B3 ? b3->sdf_wpT[k] : nullptr, mt0, bwd, ld, acc, lane);
        ARAH_SYNC();
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
#pragma unroll
            for (int n = 0; n < kNT; ++n) {
                const f32x4 d = spill[((k * kWaves + wave) * (kSdfMT * kNT) + m * kNT + n) * 64 + lane];
                *reinterpret_cast<f32x4*>(bwd + (n * 16 + j) * ld + ch0) = acc[m][n] * d;
            }
        }
        ARAH_SYNC();
    }
    // grad_c = sum_ch w0[ch][c] * u1[pt][ch]
    const int pt = tid >> 3, part = tid & 7;
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const int ch = part + 8 * i;
        const float u = bwd[pt * ld + ch];
        const f32x4 w = *reinterpret_cast<const f32x4*>(net.w0 + ch * 4);
        gx += w[0] * u;
        gy += w[1] * u;
        gz += w[2] * u;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        gx += __shfl_xor(gx, o);
        gy += __shfl_xor(gy, o);
        gz += __shfl_xor(gz, o);
    }
    if (part == 0) {
        out[pt * ostride + 1] = gx;
        out[pt * ostride + 2] = gy;
        out[pt * ostride + 3] = gz;
    }
}

// ------------------------------------------------------------------------------------------
// skinning network: 3 -> 128 x4 -> 25, Softplus(beta = 100)
// ------------------------------------------------------------------------------------------
struct SkinNet {
    const float* w0;      // [128][4]
    const float* wp[3];   // packed 128x128
    const float* w4p;     // packed [32][128]
    const float* bias;    // [4][128] + [32]
    const f16x8* wps[4];  // split-packed 128x128 x3, [32][128]
    const float* scales;  // [0..3] activation scale S_k of layer k's output (power of two)
                          // [4..7] 1 / (weight scale of layer k+1 * S_k): turns the split accumulator into W h
};

constexpr int kSkinLd = 136;   // 128 + 8
constexpr int kLogitLd = 33;

// Operands of the POINT-OWNING-wave kernels (csrc/canon_wave.hpp): a wave owns its points AND all 128 channels, a
// layer's accumulators become the next layer's B fragments without leaving the registers.  That fixes the order in
// which a 32-chunk of the contraction walks the channels -- lane (j, g) of a 16x16 accumulator tile holds channels
// mt*16 + 4g + r of point j, so chunk kc, lane group g, element e is channel (2 kc + (e >> 2)) * 16 + 4 g + (e & 3)
// -- and the weights are split-packed in that order (k_pack_split<PERM>).
//
// Units (round 4).  Activations travel in z = 100 log2(e) x, in which Softplus(beta = 100) is S(z) = log2(1 + 2^z), and
// SHIFTED by kCwShift = 24: a layer hands r' = S(z) - 24 to the next one and its accumulator holds z' = z - 24, because
//     S(z) - 24 = log2(2^-24 + 2^(z - 24)) = max(z', log2(2^-24 + min(2^z', 1)))
// is FOUR vector instructions straight off the accumulator -- v_exp_f32 with the clamp modifier (min(., 1): above z' = 0
// the log term is log2(1 + 2^-24) = 0 in fp32 and the max returns z' itself, which is S(z) - 24 to fp32 for z >= 24),
// v_add, v_log_f32, v_med3 -- against seven for max(z, 0) + log2(1 + 2^-|z|) behind an un-scaling fma.  What makes
// the accumulator usable as it is: (1) the weights are split UNSCALED (f16 subnormals carry the lo halves of small
// weights: absolute resolution 2^-25, below the 2^-23 relative resolution of the layer's largest weights, which
// dominate the error), so there is no power of two to undo; (2) the bias, the shift of the incoming activations
// (24 * rowsum(W)) and the layer's own -24 are the accumulator's START value (the C operand of the first MFMA of
// a chain).  Measured against an fp64 chain on the synthetic subject's network: rms logit error 4.0e-7 (scaled
// weights / unshifted: 3.4e-7, the exact fp32 engine: 3.3e-7).
struct SkinWave {
    const f16x8* wpr;      // layer L (0..3) at byte kCwLayerBytes * L: [(mt*4 + kc)*2 + s][64 lanes] of 8 halves (s = 0 hi,
                           // 1 lo) in the permuted channel order -- ONE block, so that a single buffer descriptor plus a
                           // compile-time scalar offset addresses every fragment (no per-fragment address registers)
    const float* consts;   // kCwW0T ..: see the enum
};
constexpr int kCwLayerBytes = 128 * 128 * 4;                 // hi + lo halves of a 128 x 128 layer
constexpr int kCwWeightBytes = 3 * kCwLayerBytes + 32 * 128 * 4;
enum {
    kCwW0T = 0,       // [128][8] words: the K = 3 input layer's A operands by output row, as halves of w0 * 100 log2(e):
                      //           {h0 h1 | h2 0 | h0 h1 | h2 0} (hi fragment of every lane) {l0 l1 | l2 0 | l0 l1 | l2 0} (lo)
    kCwBinit = 1024,  // [4][128] + [32]: accumulator start values of layers 0..4 (bias, shift terms, activation scale)
    kCwInv = 1568,    // [4]       1 / activation scale of layer k - 1, k = 1..3; [3]: accumulator of layer 4 -> 20 log2(e) x logit
    kCwActS = 1572,   // [4]       activation scale of layers 0..3: 1 unless the probed activations (k_skin_probe) call for
                      //           less -- a power of two that keeps 32 x the probed maximum inside the f16 range
    kCwScaled = 1576, // [1]       1 if any activation scale differs from 1 (selects the SCALED instance of the kernel), else 0
    kCwSize = 1580
};
constexpr float kZUnit = 144.269504088896341f;   // 100 log2(e)
constexpr float kCwShift = 24.0f;

// Softplus(beta = 100) in z units: S(z) = log2(1 + 2^z) = max(z, 0) + log2(1 + 2^-|z|).  Two transcendentals and
// three plain operations; the absolute error is that of 1 + e (6e-8 in z, 4e-10 in x).
__device__ __forceinline__ float softplus_z(float z) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(z));
    return fmaxf(z, 0.f) + __builtin_amdgcn_logf(1.0f + e);
}
// The shifted form (see above): zs = z - 24 -> S(z) - 24.  `inf` is +infinity in a register the compiler cannot see
// through: med3(a, b, +inf) = max(a, b) without the canonicalising v_max(a, a) an fmaxf of an MFMA result costs.
#ifndef CW_SP_FORM
#define CW_SP_FORM 0
#endif
__device__ __forceinline__ float softplus_shift(float zs, float inf) {
#if CW_SP_FORM == 3    // bisecting aid: the unshifted form behind two adds
    return softplus_z(zs + kCwShift) - kCwShift;
#elif CW_SP_FORM == 1  // v_min instead of the clamp modifier
    const float e = fminf(__builtin_amdgcn_exp2f(zs), 1.0f);
    return __builtin_amdgcn_fmed3f(zs, __builtin_amdgcn_logf(e + 0x1p-24f), inf);
#elif CW_SP_FORM == 2  // canonicalising max instead of med3
    const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(zs), 0.f, 1.f);
    return fmaxf(zs, __builtin_amdgcn_logf(e + 0x1p-24f));
#else
    const float e = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(zs), 0.f, 1.f);   // v_exp_f32 ... clamp
    return __builtin_amdgcn_fmed3f(zs, __builtin_amdgcn_logf(e + 0x1p-24f), inf);
#endif
}

// Softplus(beta=100): log1p(exp(100 x))/100 == max(x,0) + log1p(exp(-|100 x|))/100.  The correction is
// <= 0.00693 and needs only ABSOLUTE accuracy (~1e-9 here), so the hardware exp2/log2 are fed directly: the
// argument -|x| 100 log2(e) is at most a few tens in magnitude where the correction is not yet negligible, its
// rounding error (|a| 2^-24) moves the correction by < 2e-10; for 100 x > 20 the correction vanishes in fp32 and
// the result is x, like torch's threshold branch.  Six VALU operations, two of them transcendental.
// Scaled form for the split engine: xs = S x  ->  S softplus100(x), with c1 = 100 log2(e) / S, c2 = S ln(2) / 100
// (S a power of two: the scaling commutes with every rounding).
__device__ __forceinline__ float softplus100_scaled(float xs, float c1, float c2) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(xs) * c1);
    return fmaf(__builtin_amdgcn_logf(1.0f + e), c2, fmaxf(xs, 0.f));
}
__device__ __forceinline__ float softplus100(float x) {
    return softplus100_scaled(x, 144.269504088896341f, 6.93147180559945e-3f);
}

// xin LDS [16*NT][4] normalised coords -> logits LDS [16*NT][kLogitLd] (25 valid, un-scaled).
// SPLIT: hidden activations h >= 0 live in LDS as hi/lo f16 planes of S_k h (row = 256 B hi | 256 B lo); S_k comes
// from a per-frame probe of the network (arah_prepare_frame) with 32x headroom, conversions saturate.
// clk marks: 2 = input layer, 3/5/7 = GEMM of hidden layer k, 4/6/8 = its barrier + epilogue, 9 = output layer
template <int NT = kNT, bool SPLIT = false, typename CLK = NoClk>
__device__ __forceinline__ void skin_mlp(const SkinNet& net, const float* xin, float* act, float* logits, int wave,
                                         int lane, CLK&& clk = NoClk()) {
    const int j = lane & 15, g = lane >> 4;
    const int ld = kSkinLd;
    constexpr float kSat = 65504.0f;
    {
        const int ch0 = wave * 16 + 4 * g;
        f32x4 w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(net.bias + ch0);
        const float S = SPLIT ? net.scales[0] : 1.0f;
        const float c1 = 144.269504088896341f / S, c2 = 6.93147180559945e-3f * S;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(xin + (n * 16 + j) * 4);
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r)   // explicit chain: every instantiation must round identically
                h[r] = fmaf(w[r][2], x[2], fmaf(w[r][1], x[1], fmaf(w[r][0], x[0], b[r])));
            no_pack(h);
            if (SPLIT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = fminf(softplus100_scaled(h[r] * S, c1, c2), kSat);
                store_split4(act, ld, 256, n * 16 + j, ch0, h);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = softplus100(h[r]);
                *reinterpret_cast<f32x4*>(act + (n * 16 + j) * ld + ch0) = h;
            }
        }
    }
    ARAH_SYNC();
    clk.mark(2);
    SplitA<4> a_out;
#pragma unroll 1
    for (int k = 1; k < 4; ++k) {
        f32x4 acc[1][NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) zero_acc(acc[0][n]);
        if (SPLIT) gemm_acc_split<4, 1, NT>(net.wps[k - 1], wave, act, ld, 256, acc, lane);
        else gemm_acc<8, 1, NT>(net.wp[k - 1], wave, act, ld, acc, lane);
        clk.mark(1 + 2 * k);
        // the epilogue's constants travel (L2 latency) while the workgroup gathers at the barrier
        const int ch0 = wave * 16 + 4 * g;
        f32x4 b = *reinterpret_cast<const f32x4*>(net.bias + k * 128 + ch0);
        const float S = SPLIT ? net.scales[k] : 1.0f;
        const float inv = SPLIT ? net.scales[4 + k - 1] * S : 1.0f;   // accumulator -> S (W h): S is a power of two
        ARAH_SYNC();
        const float c1 = 144.269504088896341f / S, c2 = 6.93147180559945e-3f * S;
        if (SPLIT) b = b * S;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            f32x4 h;
            if (SPLIT) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    h[r] = fminf(softplus100_scaled(fmaf(acc[0][n][r], inv, b[r]), c1, c2), kSat);
                store_split4(act, ld, 256, n * 16 + j, ch0, h);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) h[r] = softplus100(acc[0][n][r] + b[r]);
                *reinterpret_cast<f32x4*>(act + (n * 16 + j) * ld + ch0) = h;
            }
        }
        if (k < 3) {
            ARAH_SYNC();
            clk.mark(2 + 2 * k);
        }
    }
    // the output layer's A fragments travel (L2 latency) while the workgroup gathers at the barrier
    if (SPLIT && (wave >> 1) < NT) a_out = load_split_a<4>(net.wps[3], wave & 1, lane);
    ARAH_SYNC();
    clk.mark(8);
    {   // output layer 128 -> 25 (padded 32): wave w computes M-tile (w & 1) of N-tile (w >> 1)
        const int mt = wave & 1, nt = wave >> 1;
        if (nt < NT) {
            f32x4 acc;
            if (SPLIT) {
                acc = gemm_one_split<4>(a_out, nt, act, ld, 256, lane) * net.scales[7];
            } else {
                acc = gemm_one<8>(net.w4p, mt, nt, act, ld, lane);
            }
            const int ch0 = mt * 16 + 4 * g;
            const f32x4 b = *reinterpret_cast<const f32x4*>(net.bias + 4 * 128 + ch0);
#pragma unroll
            for (int r = 0; r < 4; ++r) logits[(nt * 16 + j) * kLogitLd + ch0 + r] = acc[r] + b[r];
        }
    }
    ARAH_SYNC();
    clk.mark(9);
}

// ------------------------------------------------------------------------------------------
// colour network: [feat(256), x(3), n(3), (PE4(view) 27)] -> 256 -> 256 -> 128 -> (+in) 256 -> 256 -> 3
// ------------------------------------------------------------------------------------------
struct ColNet {
    const float* w0p;
    const float* w1p;
    const float* w2p;
    const float* w3ap;
    const float* w3bp;
    const float* w4p;
    const float* w5;     // [3][256]
    const float* bias;   // b0'[256] b1[256] b2[128] b3'[256] b4[256] b5[4]
};

template <bool IDR>
struct ColDims {
    static constexpr int kExtra = IDR ? 33 : 6;
    static constexpr int kIn = 256 + kExtra;
    static constexpr int kKC0 = (kIn + 15) / 16;       // 19 / 17
    static constexpr int kInPad = kKC0 * 16;           // 304 / 272
    static constexpr int kLdA = kInPad + 8;            // 8 dwords mod 64 apart (see split_slot)
};

template <int MT>
__device__ __forceinline__ void relu_store(const f32x4 (&acc)[MT][kNT], const float* bias, float* dst, int ld,
                                           int mt0, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ch0 = (mt0 + m) * 16 + 4 * g;
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + ch0);
#pragma unroll
        for (int n = 0; n < kNT; ++n) {
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[m][n][r] + b[r], 0.f);
            *reinterpret_cast<f32x4*>(dst + (n * 16 + j) * ld + ch0) = h;
        }
    }
}

struct ColTap {         // training: the colour MLP's input and hidden activations as dense [P][width] streams
    float* cin;         // [P][kInPad]
    float* c[5];        // c1 [P][256], c2 [P][256], c3 [P][128], c4 [P][256], c5 [P][256]
    long long row0;
    int rows;
};

// f16 split planes of the trunk's h6 (hi at byte 0, lo at 512, scaled by kActScale) -> bf16 planes (lo at lo_off) in place,
// through registers: 64 points x 256 channels
__device__ __forceinline__ void resplit_rows_bf16(float* act, int ld, int lo_off, int tid) {
    f32x4 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * kThreads, pt = e >> 6, ch0 = (e & 63) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] = load_split(act, ld, 512, pt, ch0 + r);
    }
    ARAH_SYNC();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = tid + i * kThreads;
        store_bsplit4(act, ld, lo_off, e >> 6, (e & 63) * 4, v[i]);
    }
}

// d_0 = 30 f cos(z_0) of the wave's channels (M-tile m) at N-tile n, exactly as sdf_trunk's first layer computes it
__device__ __forceinline__ f32x4 layer0_dfactor(const SdfNet& net, const float* xin, int mt0, int m, int n, int lane) {
    const int j = lane & 15, g = lane >> 4;
    const int ch0 = (mt0 + m) * 16 + 4 * g;
    const f32x4 x = *reinterpret_cast<const f32x4*>(xin + (n * 16 + j) * 4);
    const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
    const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
    const f32x4 f = *reinterpret_cast<const f32x4*>(net.freq + ch0);
    f32x4 v, h, d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
        v[r] = fmaf(w[2], x[2], fmaf(w[1], x[1], w[0] * x[0]));
    }
    no_pack(v);
    film_sine<true>(v, fw, pw, f, 1.0f, h, d);
    return d;
}

// sdf_backward with the sweep's activations as bf16 planes in `bwd` (rows of ld floats, lo plane at byte 512)
template <typename CLK = NoClk>
__device__ __forceinline__ void sdf_backward_bp(const SdfNet& net, const B3Nets& b3, float* bwd, int ld, const f32x4* spill,
                                                const f32x4 (&dlast)[kSdfMT][kNT], float* out, int ostride, int wave,
                                                int lane, int tid, CLK* pc = nullptr, const float* xin = nullptr) {
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
#pragma unroll
    for (int m = 0; m < kSdfMT; ++m) {
        const int ch0 = (mt0 + m) * 16 + 4 * g;
        const f32x4 w = *reinterpret_cast<const f32x4*>(net.w6 + ch0);
#pragma unroll
        for (int n = 0; n < kNT; ++n) store_bsplit4(bwd, ld, 512, n * 16 + j, ch0, dlast[m][n] * w);
    }
    ARAH_SYNC();
#pragma unroll 1
    for (int k = 4; k >= 0; --k) {
        f32x4 acc[kSdfMT][kNT];
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        // the layer's derivative factors (the forward trunk's slab, HBM / L2) are requested AHEAD of its product: round 3 read
        // them behind the barrier that follows it and sat out the round trip five times per tile (SQ_WAIT_ANY 70 %)
        f32x4 dk[kSdfMT][kNT];
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n)
#ifdef ARAH_ABL_NO_SLAB
                dk[m][n] = dlast[m][n];
#else
                dk[m][n] = k == 0 && xin ? layer0_dfactor(net, xin, mt0, m, n, lane)
                                         : spill[((k * kWaves + wave) * (kSdfMT * kNT) + m * kNT + n) * 64 + lane];
#endif
        gemm_acc_bsplit<8, kSdfMT>(b3.sdf_wpT[k], mt0, bwd, ld, 512, acc, lane);
        ARAH_PC_MARK(10);
        ARAH_SYNC();
        ARAH_PC_MARK(12);
#pragma unroll
        for (int m = 0; m < kSdfMT; ++m) {
            const int ch0 = (mt0 + m) * 16 + 4 * g;
#pragma unroll
            for (int n = 0; n < kNT; ++n) store_bsplit4(bwd, ld, 512, n * 16 + j, ch0, acc[m][n] * dk[m][n]);
        }
        ARAH_PC_MARK(11);
        ARAH_SYNC();
        ARAH_PC_MARK(12);
    }
    const int pt = tid >> 3, part = tid & 7;
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
        const int ch = part + 8 * i;
        const float u = load_bsplit(bwd, ld, 512, pt, ch);
        const f32x4 w = *reinterpret_cast<const f32x4*>(net.w0 + ch * 4);
        gx += w[0] * u;
        gy += w[1] * u;
        gz += w[2] * u;
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
        gx += __shfl_xor(gx, o);
        gy += __shfl_xor(gy, o);
        gz += __shfl_xor(gz, o);
    }
    if (part == 0) {
        out[pt * ostride + 1] = gx;
        out[pt * ostride + 2] = gy;
        out[pt * ostride + 3] = gz;
    }
}

// A: LDS [64][kLdA] full input (feature in cols 0..255, extras after, zero padded); B: LDS [64][260].
// rgb (after sigmoid) -> out[pt*ostride + 0..2].  Needs a barrier between the writers of A and the call.
// tap != nullptr (training): A and every hidden activation are streamed out; B holds c5 on return.
template <bool IDR, bool B3 = false>
__device__ __forceinline__ void color_mlp(const ColNet& net, const float* A, float* B, float* out, int ostride,
                                          int wave, int lane, int tid, const ColTap* tap = nullptr, const B3Nets* b3 = nullptr) {
    typedef ColDims<IDR> D;
    constexpr int ldB = kSdfLd;
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        gemm_any<B3, D::kKC0, 2>(net.w0p, B3 ? b3->col[0] : nullptr, wave * 2, A, D::kLdA, acc, lane);
        relu_store<2>(acc, net.bias, B, ldB, wave * 2, lane);   // B is not read by this GEMM
    }
    ARAH_SYNC();
    if (tap) {
        stream_rows(A, D::kLdA, D::kInPad, tap->cin, tap->row0, tap->rows, tid);
        stream_rows(B, ldB, 256, tap->c[0], tap->row0, tap->rows, tid);
    }
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        gemm_any<B3, 16, 2>(net.w1p, B3 ? b3->col[1] : nullptr, wave * 2, B, ldB, acc, lane);
        ARAH_SYNC();
        relu_store<2>(acc, net.bias + 256, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    if (tap) stream_rows(B, ldB, 256, tap->c[1], tap->row0, tap->rows, tid);
    {
        f32x4 acc[1][kNT];
#pragma unroll
        for (int n = 0; n < kNT; ++n) zero_acc(acc[0][n]);
        gemm_any<B3, 16, 1>(net.w2p, B3 ? b3->col[2] : nullptr, wave, B, ldB, acc, lane);
        ARAH_SYNC();
        relu_store<1>(acc, net.bias + 512, B, ldB, wave, lane);   // cols 0..127
    }
    ARAH_SYNC();
    if (tap) stream_rows(B, ldB, 128, tap->c[2], tap->row0, tap->rows, tid);
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        gemm_any<B3, D::kKC0, 2>(net.w3ap, B3 ? b3->col[3] : nullptr, wave * 2, A, D::kLdA, acc, lane);
        gemm_any<B3, 8, 2>(net.w3bp, B3 ? b3->col[4] : nullptr, wave * 2, B, ldB, acc, lane);
        ARAH_SYNC();
        relu_store<2>(acc, net.bias + 640, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    if (tap) stream_rows(B, ldB, 256, tap->c[3], tap->row0, tap->rows, tid);
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        gemm_any<B3, 16, 2>(net.w4p, B3 ? b3->col[5] : nullptr, wave * 2, B, ldB, acc, lane);
        ARAH_SYNC();
        relu_store<2>(acc, net.bias + 896, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    if (tap) stream_rows(B, ldB, 256, tap->c[4], tap->row0, tap->rows, tid);
    {
        const int pt = tid >> 3, part = tid & 7;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int ch = part + 8 * i;
            const float h = B[pt * ldB + ch];
            c0 += net.w5[ch] * h;
            c1 += net.w5[256 + ch] * h;
            c2 += net.w5[512 + ch] * h;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            c0 += __shfl_xor(c0, o);
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
        }
        if (part == 0) {
            out[pt * ostride + 0] = 1.0f / (1.0f + expf(-(c0 + net.bias[1152 + 0])));
            out[pt * ostride + 1] = 1.0f / (1.0f + expf(-(c1 + net.bias[1152 + 1])));
            out[pt * ostride + 2] = 1.0f / (1.0f + expf(-(c2 + net.bias[1152 + 2])));
        }
    }
}

// The colour MLP on the bf16 x 3 engine with every activation as bf16 planes.
// A: planes of the full input, rows of ldA floats, lo plane at byte loA, ceil(kInPad / 32) chunks of which the channels
// beyond kIn are zero; B: planes of the hidden activations, rows of kSdfLd floats, lo plane at byte 512.
// the bias of a wave's channels, requested BEFORE the barrier that precedes relu_store_bp (an L2 round trip otherwise)
template <int MT>
__device__ __forceinline__ void load_bias(const float* bias, int mt0, int lane, f32x4 (&b)[MT]) {
    const int g = lane >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) b[m] = *reinterpret_cast<const f32x4*>(bias + (mt0 + m) * 16 + 4 * g);
}
template <int MT>
__device__ __forceinline__ void relu_store_bp(const f32x4 (&acc)[MT][kNT], const f32x4 (&b)[MT], float* dst, int ld,
                                              int mt0, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ch0 = (mt0 + m) * 16 + 4 * g;
#pragma unroll
        for (int n = 0; n < kNT; ++n) {
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = fmaxf(acc[m][n][r] + b[m][r], 0.f);
            store_bsplit4(dst, ld, 512, n * 16 + j, ch0, h);
        }
    }
}

template <bool IDR>
__device__ __forceinline__ void color_mlp_bp(const ColNet& net, const B3Nets& b3, const float* A, int ldA, int loA, float* B,
                                             float* out, int ostride, int wave, int lane, int tid) {
    typedef ColDims<IDR> D;
    constexpr int ldB = kSdfLd;
    constexpr int KCA = (D::kKC0 + 1) / 2;
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        f32x4 bias[2];
        load_bias<2>(net.bias, wave * 2, lane, bias);
        gemm_acc_bsplit<KCA, 2>(b3.col[0], wave * 2, A, ldA, loA, acc, lane);
        relu_store_bp<2>(acc, bias, B, ldB, wave * 2, lane);   // B is not read by this GEMM
    }
    ARAH_SYNC();
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        f32x4 bias[2];
        load_bias<2>(net.bias + 256, wave * 2, lane, bias);
        gemm_acc_bsplit<8, 2>(b3.col[1], wave * 2, B, ldB, 512, acc, lane);
        ARAH_SYNC();
        relu_store_bp<2>(acc, bias, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    {
        f32x4 acc[1][kNT];
#pragma unroll
        for (int n = 0; n < kNT; ++n) zero_acc(acc[0][n]);
        f32x4 bias[1];
        load_bias<1>(net.bias + 512, wave, lane, bias);
        gemm_acc_bsplit<8, 1>(b3.col[2], wave, B, ldB, 512, acc, lane);
        ARAH_SYNC();
        relu_store_bp<1>(acc, bias, B, ldB, wave, lane);   // channels 0..127
    }
    ARAH_SYNC();
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        f32x4 bias[2];
        load_bias<2>(net.bias + 640, wave * 2, lane, bias);
        gemm_acc_bsplit<KCA, 2>(b3.col[3], wave * 2, A, ldA, loA, acc, lane);
        gemm_acc_bsplit<4, 2>(b3.col[4], wave * 2, B, ldB, 512, acc, lane);
        ARAH_SYNC();
        relu_store_bp<2>(acc, bias, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    {
        f32x4 acc[2][kNT];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < kNT; ++n) zero_acc(acc[m][n]);
        f32x4 bias[2];
        load_bias<2>(net.bias + 896, wave * 2, lane, bias);
        gemm_acc_bsplit<8, 2>(b3.col[5], wave * 2, B, ldB, 512, acc, lane);
        ARAH_SYNC();
        relu_store_bp<2>(acc, bias, B, ldB, wave * 2, lane);
    }
    ARAH_SYNC();
    {
        const int pt = tid >> 3, part = tid & 7;
        float c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll 8
        for (int i = 0; i < 32; ++i) {
            const int ch = part + 8 * i;
            const float h = load_bsplit(B, ldB, 512, pt, ch);
            c0 += net.w5[ch] * h;
            c1 += net.w5[256 + ch] * h;
            c2 += net.w5[512 + ch] * h;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            c0 += __shfl_xor(c0, o);
            c1 += __shfl_xor(c1, o);
            c2 += __shfl_xor(c2, o);
        }
        if (part == 0) {
            out[pt * ostride + 0] = 1.0f / (1.0f + expf(-(c0 + net.bias[1152 + 0])));
            out[pt * ostride + 1] = 1.0f / (1.0f + expf(-(c1 + net.bias[1152 + 1])));
            out[pt * ostride + 2] = 1.0f / (1.0f + expf(-(c2 + net.bias[1152 + 2])));
        }
    }
}

}  // namespace arah
