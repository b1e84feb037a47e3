// canon_wave.hpp -- loop C (RFU:267-362, broyden.py:4-78) as a resident kernel of POINT-OWNING waves.
//
// Included by arah_hip.hip inside its anonymous namespace, after k_canon_solve (whose slot-state layout, seed convention
// and per-point tail it shares).  k_canon_solve (8 waves slice the CHANNELS of one 64-point tile) pays ten workgroup
// barriers per pass and leaves four of its eight waves idle through the per-point tail and the refill; here
//
//   * a wave owns kCwNT * 16 points and ALL 128 channels of the skinning MLP: lane (j, g) of a 16x16 accumulator tile
//     holds channels mt*16 + 4g + r of point j -- exactly one lane group's share of the next layer's B fragment, in
//     the channel order the weights were packed for (SkinWave, mlp.hpp).  A layer's output never leaves the registers:
//     no activation traffic through LDS, no workgroup barrier after the launch prologue, and the eight waves of a
//     workgroup drift apart so that one wave's Softplus epilogues run under another's MFMAs;
//   * inside a wave the epilogue of M-tile pair q of layer L (= B chunk q of layer L + 1) is issued between the MFMA
//     steps of chunk q - 1 of layer L + 1, and pair 0 between the steps of the last chunk of layer L itself (the
//     M-tile pair is final as soon as its own step of the last chunk has issued);
//   * A operands: the hi halves of the four weight matrices live in LDS (104 KB, staged once per workgroup), the lo
//     halves stream from L2 two steps ahead -- an all-L2 delivery at 32 points per wave would ask for twice the
//     fragment bytes per point of the tile kernel;
//   * every slot keeps a PREFETCHED start state (claimed from the launch-wide queue when the slot last refilled, its
//     80 bytes loaded while the point before it was being solved), so a retiring point is replaced without waiting for
//     HBM;
//   * the per-point tail runs on the FOUR LANES (j, g) of a point, with no LDS round trip between them (round 4; round 3
//     exchanged logits, gates and T rows through LDS rows behind wave barriers and ran the Broyden step on 32 of the 64
//     lanes: 38-41 % of the kernel's time in its phase clocks).  What one lane has and the other three need crosses
//     lanes without touching LDS memory -- round 4: on the MATRIX pipe (v_mfma_f32_16x16x4_f32 with A[i][k] = (i mod 4 == k)
//     hands every lane of point n the four values B[0..3][n] its lane groups supplied, an exact copy: products with 1 and
//     sums with 0; with A[i][k] = (k < 3) their sum over the three rows of a 3-vector, an fmaf chain in row order); round 5:
//     on the LDS crossbar (ds_bpermute_b32: the same copies, the same sums in the same order, and the matrix pipe stays
//     with the MLP: -3 % per launch).  Lane g keeps row g of everything: the gate
//     sigmoids of its own eight logits (gathered: 8 MFMAs), row g of T (six MFMA steps as before), component g of x, step,
//     residual and best x, row g of J^-1 -- the Broyden update needs one gather of the new residual, one of the old, and
//     three row sums (v^T = dx^T J^-1).  The two N-tiles of a wave are independent instruction streams the compiler
//     interleaves.  State between passes: 2 x 16 bytes per lane plus 2 x 16 per point, read and written as ds_*_b128.
// Activations travel in z = 100 log2(e) x, shifted by 24 (softplus_shift, mlp.hpp: four vector instructions per
// activation straight off the accumulator, whose start value carries bias and shift); the K = 3 input layer is one
// padded MFMA chunk (B = {x_hi | x_lo} of the lane group 0, zeros elsewhere).  Results differ from the tile kernel's
// by rounding only.
#pragma once

#ifndef CW_NT
#define CW_NT 2
#endif
constexpr int kCwNT = CW_NT;                    // N-tiles (16 points each) per wave
static_assert(kCwNT == 2, "the only validated geometry (one N-tile per wave gave wrong roots on the MI355X and was slower)");
#ifndef CW_WAVES
#define CW_WAVES 8
#endif
constexpr int kCwWaves = CW_WAVES;
constexpr int kCwThreads = kCwWaves * 64;
constexpr int kCwSlots = 16 * kCwNT;            // points per wave
constexpr int kCwHiBytes = (3 * 32 + 8) * 1024; // hi fragments: 3 x (8 M-tiles x 4 chunks) + 2 x 4, 1 KB each
constexpr int kCwConstFloats = kCwInv;          // input-layer operands and accumulator start values are staged in LDS
// per wave (floats): the state of its 32 slots between two passes, laid out for conflict-free 16-byte accesses
//   P0 [3][32][4]  row g of the point in slot s: {x_g, step_g, best x_g, g_g}
//   P1 [3][32][4]  {J^-1[g][0..2], -}
//   S0 [32][4]     {|g|_best, evaluations so far (int), id (int; -1: empty), -}
//   S1 [32][4]     {target(3), -}
constexpr int kCwP0 = 0, kCwP1 = 3 * 32 * 4, kCwS0 = 2 * 3 * 32 * 4, kCwS1 = kCwS0 + 32 * 4;
constexpr int kCwWaveFloats = kCwS1 + 32 * 4;
constexpr unsigned kCwWhiOff = (kCwConstFloats + 24 * 16 + kCwWaves * kCwWaveFloats) * 4;   // hi fragments: byte offset in LDS
constexpr size_t kLdsCanonWave = (size_t)kCwHiBytes + (kCwConstFloats + 24 * 16 + kCwWaves * kCwWaveFloats) * 4;
constexpr int kCwSeedChunk = 64;                // seeds a wave takes from the queue per atomic

struct CwStep {   // MFMA step G of a pass: layer L (1..4), chunk kc, M-tile pair mp
    int L, kc, mp;
};
__host__ __device__ constexpr CwStep cw_step(int G) {
    return G < 48 ? CwStep{G / 16 + 1, (G % 16) / 4, G % 4} : CwStep{4, G - 48, 0};
}
constexpr int kCwSteps = 52;

// Which parts (of the 2 NT an M-tile pair's epilogue has: N-tile x M-tile of the pair) ride with MFMA step (L, kc, mp).
// kc < 3: the parts belong to pair kc + 1 of layer L - 1 -- spread over the four steps of a hidden layer's chunk, all of
// them in the single step of an output-layer chunk.  Last chunk of a hidden layer: pair 0 of layer L itself, which is
// final once its own step (mp = 0) has issued.
struct CwParts {
    int first, count;
};
__host__ __device__ constexpr CwParts cw_parts(int NT, int L, int kc, int mp) {
    const int n = 2 * NT;
    if (kc < 3) {
        if (L == 4) return CwParts{0, n};
        if (n >= 4) return CwParts{mp * (n / 4), n / 4};          // n = 4 (two N-tiles): one part per step
        return (mp & 1) ? CwParts{0, 0} : CwParts{mp / 2, 1};      // n = 2 (one N-tile): steps 0 and 2
    }
    if (L == 4 || mp == 0) return CwParts{0, 0};
    if (n >= 4) return mp == 3 ? CwParts{2 * (n / 4), n - 2 * (n / 4)} : CwParts{(mp - 1) * (n / 4), n / 4};
    return mp == 3 ? CwParts{0, 0} : CwParts{mp - 1, 1};
}
#ifndef CW_LO_DIST
#define CW_LO_DIST 2    // MFMA steps the lo fragments (L2) are requested ahead of their use
#endif
#ifndef CW_HI_DIST
#define CW_HI_DIST 0    // the same for the hi fragments (LDS): requested ahead they cost more in registers than they hide
#endif
#ifndef CW_CI_MODE
#define CW_CI_MODE 0    // bisecting aid: 0 start values LDS -> accumulator, 1 zero start + bias added in the epilogue, 2 as 0, one load per N-tile
#endif
#ifndef CW_PIN_MASK
#define CW_PIN_MASK 0x040f
#endif
constexpr int kCwLoDist = CW_LO_DIST, kCwHiDist = CW_HI_DIST;

template <bool HI_LDS, bool SCALED>
__global__ __launch_bounds__(kCwThreads, kCwWaves >= 8 ? kCwWaves / 4 : 2) void k_canon_wave(FrameDev fr, const int* __restrict__ list, const int* count,
                                                              int* queue_head, CanonOut outp, unsigned long long* ctr,
                                                              unsigned long long* ctr_canon,
                                                              unsigned long long* ctr_bad, unsigned long long* clk_out) {
    constexpr int NT = kCwNT;
    // SCALED: the instance for skinning networks whose activations must travel scaled down (kCwActS); the frame's constants
    // say which of the two instances of a launch pair runs
    if ((fr.skw.consts[kCwScaled] != 0.f) != SCALED) return;
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    KernelClk clk;
#ifdef ARAH_CLOCKS
    clk.start();
#endif
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
#ifdef ARAH_PRIO_WAVES   // A/B (profiles/r05_ab_setprio.txt): static issue priority for one half of the workgroup's waves
    if ((wave >= kCwWaves / 2) == (ARAH_PRIO_WAVES == 1)) __builtin_amdgcn_s_setprio(1);
#endif
    // small operands first: their absolute LDS offsets stay below 64 KB, i.e. inside the offset field of the DS
    // instructions (one address register per lane pattern, not one per constant for the compiler to hoist and spill)
    float* w0c = smem;                   // kCwW0T: the input layer's A operands by row (32 bytes each)
    float* binit = w0c + kCwBinit;       // accumulator start values [4][128] + [32]
    float* sbones = w0c + kCwConstFloats;
    float* wst = sbones + 24 * 16 + wave * kCwWaveFloats;     // this wave's slot states (kCwP0 ...)
    char* whi = reinterpret_cast<char*>(sbones + 24 * 16 + kCwWaves * kCwWaveFloats);   // hi fragments (HI_LDS)
    const float* cst = fr.skw.consts;
    // ---- launch prologue: operands into LDS (the only workgroup barrier of the kernel)
    for (int i = tid; i < kCwConstFloats; i += kCwThreads) w0c[i] = cst[i];
    for (int i = tid; i < 24 * 16; i += kCwThreads) sbones[i] = fr.bones[i];
    if (HI_LDS) {
#pragma unroll
        for (int L = 0; L < 4; ++L) {
            const int nf = L < 3 ? 32 : 8;
            for (int f = wave; f < nf; f += kCwWaves)
                reinterpret_cast<f16x8*>(whi + (L * 32 + f) * 1024)[lane] =
                    fr.skw.wpr[(size_t)L * (kCwLayerBytes / 16) + (size_t)(f * 2) * 64 + lane];
        }
    }
    const float inv1 = cst[kCwInv], inv2 = cst[kCwInv + 1], inv3 = cst[kCwInv + 2], c20 = cst[kCwInv + 3];   // c20: accumulator of layer 4 -> 20 log2(e) x logit
    const float as0 = cst[kCwActS], as1 = cst[kCwActS + 1], as2 = cst[kCwActS + 2], as3 = cst[kCwActS + 3];
    float inf;   // +infinity the compiler cannot see through (softplus_shift)
    asm volatile("s_mov_b32 %0, 0x7f800000" : "=s"(inf));
    // RFU:37-44 as one fma per coordinate: x_norm = (x - center - cmin + pad) * 2 / (1.1 rng) - 1
    const float nrm_s = 2.0f / ((bc.cmax - bc.cmin) * 1.1f);
    float nrm_o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) nrm_o[c] = ((bc.cmax - bc.cmin) * 0.05f - bc.center[c] - bc.cmin) * nrm_s - 1.0f;
    const int n = *count;
    const int g2 = min(g, 2);   // lane group 3 mirrors row 2 (it reads it, never writes it)
    float *P0[NT], *P1[NT], *S0[NT], *S1[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        P0[t] = wst + kCwP0 + (g2 * 32 + t * 16 + j) * 4;
        P1[t] = wst + kCwP1 + (g2 * 32 + t * 16 + j) * 4;
        S0[t] = wst + kCwS0 + (t * 16 + j) * 4;
        S1[t] = wst + kCwS1 + (t * 16 + j) * 4;
        if (g == 0) *reinterpret_cast<f32x4*>(S0[t]) = f32x4{0.f, 0.f, __int_as_float(-1), 0.f};
    }
    __syncthreads();

    const unsigned aoff = (unsigned)lane * 16u;
    // weight fragments from L2: one buffer descriptor for the whole block, the lane's 16-byte slot in the vector offset,
    // the fragment as a compile-time scalar offset -- no address registers per fragment for the compiler to hoist
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16x8*>(fr.skw.wpr), 0, kCwWeightBytes, 0x00020000);
    auto ld_w = [&](int L, int f, int s) {
        return __builtin_bit_cast(
            f16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrd, (int)aoff, (L - 1) * kCwLayerBytes + (f * 2 + s) * 1024, 0));
    };
    auto ld_lo = [&](int L, int f) { return ld_w(L, f, 1); };
    // hi fragments in LDS: 104 KB behind 53 KB of small operands -- beyond the 16-bit offset field of a DS instruction from
    // one base.  Three bases 64 KB apart (opaque to the compiler, which otherwise re-derives an address per fragment: 168
    // v_add_u32 per pass in round 3's listing) put every fragment at base + immediate.
    typedef __attribute__((address_space(3))) const f16x8 lds_f16x8;
    unsigned hb0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(smem) + aoff;
    unsigned hb1 = hb0 + 65536u, hb2 = hb0 + 131072u;
    asm volatile("" : "+v"(hb0), "+v"(hb1), "+v"(hb2));
    auto ld_hi = [&](auto Lc, auto fc) -> f16x8 {
        constexpr int L = decltype(Lc)::value, f = decltype(fc)::value;
        if constexpr (HI_LDS) {
            constexpr unsigned o = kCwWhiOff + (unsigned)((L - 1) * 32 + f) * 1024u;
            f16x8 v;
            if constexpr (o < 65536u) v = *reinterpret_cast<lds_f16x8*>(hb0 + o);
            else if constexpr (o < 131072u) v = *reinterpret_cast<lds_f16x8*>(hb1 + (o - 65536u));
            else v = *reinterpret_cast<lds_f16x8*>(hb2 + (o - 131072u));
            return v;
        } else {
            return ld_w(L, f, 0);
        }
    };

    // The NEXT start state of every slot, claimed from the launch-wide queue when the slot last refilled, travels in three
    // stages so that no pass ever waits for HBM: claimed at the top of pass p (cl), requested at the start of pass p's tail
    // -- behind the MLP's fragment loads, with the tail and the next refill to land in -- into (fid, fT, fx), handed over
    // to (nid, nT, nx) at the start of pass p + 1's tail, taken at the top of pass p + 2 at the earliest: a point needs at
    // least two evaluations (broyden.py:44-64), so a slot refilled in pass p is not empty again before the end of pass
    // p + 1.  The hand-over is where the wave waits for the request (an asm "use" pins the s_waitcnt there: by then it is
    // a pass old); left to the compiler the wait sits at the take, right behind the retire stores and the youngest
    // requests, and every pass pays a memory round trip.  All four lanes of a point hold its id; lane g holds row g of
    // T0, with the target riding in row 3; lanes g < 3 hold coordinate g of x0.
    int nid[NT], fid[NT];
    f32x4 nT[NT], fT[NT];
    float nx[NT], fx[NT];
    f32x4 tbr[NT];   // row g of the best transform so far
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        nid[t] = fid[t] = -1;
        nT[t] = fT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        nx[t] = fx[t] = 0.f;
        tbr[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto request = [&](const int (&cl)[NT]) {   // the start state sits where the result will go (pts, T with the target in row 3)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (cl[t] >= 0) {
                fid[t] = cl[t];
                fT[t] = reinterpret_cast<const f32x4*>(outp.T + (size_t)cl[t] * 16)[g];
                fx[t] = outp.pts[(size_t)cl[t] * 3 + min(g, 2)];
            }
        }
    };
    auto hand_over = [&]() {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            asm volatile("" : "+v"(fT[t][0]), "+v"(fT[t][1]), "+v"(fT[t][2]), "+v"(fT[t][3]), "+v"(fx[t]));
            if (fid[t] >= 0) {
                nid[t] = fid[t];
                nT[t] = fT[t];
                nx[t] = fx[t];
                fid[t] = -1;
            }
        }
    };
    int cid = -1;                    // this wave's piece of the queue: lane l holds list[q_base + l]
    int q_base = 0, q_pos = 0, q_end = 0;
    bool exhausted = n <= 0;
    int n_eval = 0, n_bad = 0;
    const unsigned long long below = (1ull << j) - 1ull;
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz

    for (int pass = 0;; ++pass) {
        // watchdog: a launch of this kernel is tens of milliseconds; a wave that is still here after 5 s gives up (its
        // remaining points keep their start states and are reported through n_split_nonfinite)
        if ((pass & 63) == 63 && __builtin_amdgcn_s_memrealtime() - t_start > 500000000ull) {
            n_bad += 1 << 20;
            break;
        }
        // ---- (1) empty slots take their next start state; a slot left without one claims a seed
        int id[NT], cl[NT];
        bool want[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            id[t] = __float_as_int(S0[t][2]);
            cl[t] = -1;
            if (id[t] < 0 && nid[t] >= 0) {
                id[t] = nid[t];
                nid[t] = -1;
                f32x4 t0 = nT[t];
                if (g == 3) {   // row 3 of the seed's T0 carries the target
                    *reinterpret_cast<f32x4*>(S1[t]) = f32x4{t0[0], t0[1], t0[2], 0.f};
                    t0 = f32x4{0.f, 0.f, 0.f, t0[3]};
                } else {
                    *reinterpret_cast<f32x4*>(P0[t]) = f32x4{nx[t], 0.f, nx[t], 0.f};
                }
                tbr[t] = t0;                                   // T0 doubles as the initial best T (broyden.py:41)
                if (g == 0) *reinterpret_cast<f32x4*>(S0[t]) = f32x4{0.f, __int_as_float(0), __int_as_float(id[t]), 0.f};
                want[t] = true;
            } else {
                want[t] = id[t] < 0 && nid[t] < 0 && fid[t] < 0;   // empty, nothing ready, nothing on its way
            }
        }
        {
            unsigned long long m[NT];
            int rank[NT], need = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                m[t] = __ballot(want[t]) & 0xffffull;   // lanes 0..15 speak for their slots
                rank[t] = need + __popcll(m[t] & below);
                need += __popcll(m[t]);
            }
            int given = 0;
            while (need > 0 && !exhausted) {   // wave-uniform
                if (q_pos == q_end) {
                    int p = 0;
                    if (lane == 0) p = atomicAdd(queue_head, kCwSeedChunk);
                    p = __builtin_amdgcn_readfirstlane(p);
                    if (p >= n) {
                        exhausted = true;
                        break;
                    }
                    q_base = q_pos = p;
                    q_end = min(p + kCwSeedChunk, n);
                    cid = p + lane < n ? list[p + lane] : -1;
                    asm volatile("" : "+v"(cid));   // wait for the ids HERE (once per 64 seeds), not at every shuffle below
                }
                const int take = min(q_end - q_pos, need);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int src = q_pos - q_base + (rank[t] - given);   // lane of cid that holds this slot's seed
                    const int got = __shfl(cid, src & 63);
                    if (want[t] && rank[t] >= given && rank[t] < given + take) cl[t] = got;
                }
                q_pos += take;
                given += take;
                need -= take;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long live_any = 0ull;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const unsigned long long live = __ballot(id[t] >= 0) & 0xffffull;
            live_any |= live;
            n_eval += __popcll(live);
        }
        if (live_any == 0ull) {   // start of the launch, or a wave whose slots all ran dry at once: fetch and wait
            request(cl);
            hand_over();
            bool pending = false;
#pragma unroll
            for (int t = 0; t < NT; ++t) pending |= __any(nid[t] >= 0) != 0;
            if (!pending) break;     // nothing in the slots, nothing on its way: the queue is dry
            continue;
        }
        // ---- (2) the input layer's B fragments: normalised coordinates as f16 hi | lo in lane group 0's eight k slots
        //      {x_hi(3), 0, x_lo(3), 0}, zeros in the other groups (the A operand repeats {w, 0, w, 0} in every group)
        f16x8 bx[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float xn[3] = {0.f, 0.f, 0.f};
            if (id[t] >= 0 && g == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) xn[c] = fmaf(wst[kCwP0 + (c * 32 + t * 16 + j) * 4], nrm_s, nrm_o[c]);
            }
            const float v4[4] = {xn[0], xn[1], xn[2], 0.f};
            unsigned u0, u1, u2, u3;
            split4(v4, u0, u1, u2, u3);
            bx[t] = __builtin_bit_cast(f16x8, u32x4{u0, u1, u2, u3});
        }
        clk.mark(0);
        // ---- (3) skinning MLP, accumulators -> B fragments in registers
        f32x4 acc[2][8][NT];
        f16x8 bch[NT], bcl[NT], bnh[NT], bnl[NT];
        u32x4 pkh[NT], pkl[NT];   // the fragments being produced, as packed words
        f16x8 lo_ring[kCwLoDist + 1][2], hi_ring[kCwHiDist + 1][2];
        // part p of the epilogue of layer L's M-tile pair q -> B chunk q of layer L + 1: four vector instructions per
        // activation straight off the accumulator (softplus_shift), three per pair for the hi / lo split
        auto epart = [&](auto Lc, auto qc, auto pc) {
            constexpr int L = decltype(Lc)::value, q = decltype(qc)::value, p = decltype(pc)::value;
            constexpr int t = p >> 1, h = p & 1;
            constexpr int mt = 2 * q + h;
            float v[4];
            f32x4 bq = {0.f, 0.f, 0.f, 0.f};
            if constexpr (CW_CI_MODE == 1) bq = *reinterpret_cast<const f32x4*>(binit + L * 128 + mt * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = acc[L & 1][mt][t][r];
                if constexpr (CW_CI_MODE == 1) a += bq[r];
                if constexpr (SCALED && L > 0) a *= L == 1 ? inv1 : (L == 2 ? inv2 : inv3);   // powers of two: exact
                v[r] = softplus_shift(a, inf);
                if constexpr (SCALED) v[r] *= L == 0 ? as0 : (L == 1 ? as1 : (L == 2 ? as2 : as3));
            }
            unsigned h0, h1, l0, l1;
            split4(v, h0, h1, l0, l1);
            pkh[t][2 * h] = h0;
            pkh[t][2 * h + 1] = h1;
            pkl[t][2 * h] = l0;
            pkl[t][2 * h + 1] = l1;
            if constexpr (h == 1) {
                bnh[t] = __builtin_bit_cast(f16x8, pkh[t]);
                bnl[t] = __builtin_bit_cast(f16x8, pkl[t]);
            }
        };
        auto load_lo = [&](auto Gc) {
            constexpr int G = decltype(Gc)::value;
            constexpr CwStep s = cw_step(G);
            lo_ring[G % (kCwLoDist + 1)][0] = ld_lo(s.L, (2 * s.mp + 0) * 4 + s.kc);
            lo_ring[G % (kCwLoDist + 1)][1] = ld_lo(s.L, (2 * s.mp + 1) * 4 + s.kc);
        };
        auto load_hi = [&](auto Gc) {
            constexpr int G = decltype(Gc)::value;
            constexpr CwStep s = cw_step(G);
            hi_ring[G % (kCwHiDist + 1)][0] = ld_hi(IC<s.L>{}, IC<(2 * s.mp + 0) * 4 + s.kc>{});
            hi_ring[G % (kCwHiDist + 1)][1] = ld_hi(IC<s.L>{}, IC<(2 * s.mp + 1) * 4 + s.kc>{});
        };
        // The start values (bias, shift terms: mlp.hpp) of a chunk-0 step's accumulators are read from LDS INTO the
        // accumulators, a step ahead: they are free by then -- a layer's accumulators were the layer before last's, whose
        // last epilogue part rides with chunk 2 of the layer in between.
        auto load_ci = [&](auto Gc) {   // only chunk-0 steps start an accumulator chain
            constexpr int G = decltype(Gc)::value;
            constexpr CwStep s = cw_step(G);
            if constexpr (s.kc == 0) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if constexpr (CW_CI_MODE == 1 && s.L < 4) {
                            acc[s.L & 1][2 * s.mp + h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                        } else {
                            acc[s.L & 1][2 * s.mp + h][t] = *reinterpret_cast<const f32x4*>(binit + s.L * 128 + (2 * s.mp + h) * 16 + 4 * g);
                            if constexpr (CW_CI_MODE == 2) {
                                f32x4& q = acc[s.L & 1][2 * s.mp + h][t];
                                asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
                            }
                        }
                    }
            }
        };
        static_for<0, kCwLoDist>([&](auto Gc) { load_lo(Gc); });
        static_for<0, kCwHiDist>([&](auto Gc) { load_hi(Gc); });
        load_ci(IC<0>{});
        // input layer: per M-tile two MFMAs (A = w_hi | w_lo of the row, repeated over both halves of the lane group's
        // k slots; B = {x_hi | x_lo}), the start value is b0 100 log2(e) - 24
        static_for<0, 4>([&](auto mpc) {
            constexpr int mp = decltype(mpc)::value;
            f16x8 a_hi[2], a_lo[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* row = w0c + ((2 * mp + h) * 16 + j) * 8;
                a_hi[h] = *reinterpret_cast<const f16x8*>(row);
                a_lo[h] = *reinterpret_cast<const f16x8*>(row + 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    if constexpr (CW_CI_MODE == 1) {
                        acc[0][2 * mp + h][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                    } else {
                        acc[0][2 * mp + h][t] = *reinterpret_cast<const f32x4*>(binit + (2 * mp + h) * 16 + 4 * g);
                        if constexpr (CW_CI_MODE == 2) {
                            f32x4& q = acc[0][2 * mp + h][t];
                            asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
                        }
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[0][2 * mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi[h], bx[t], acc[0][2 * mp + h][t], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[0][2 * mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo[h], bx[t], acc[0][2 * mp + h][t], 0, 0, 0);
        });
        static_for<0, 2 * NT>([&](auto pc) { epart(IC<0>{}, IC<0>{}, pc); });
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bch[t] = bnh[t];
            bcl[t] = bnl[t];
        }
        clk.mark(1);
        static_for<0, kCwSteps>([&](auto Gc) {
            constexpr int G = decltype(Gc)::value;
            constexpr CwStep s = cw_step(G);
            constexpr int MP = s.L < 4 ? 4 : 1;
            if constexpr (G + kCwLoDist < kCwSteps) load_lo(IC<G + kCwLoDist>{});
            if constexpr (G + kCwHiDist < kCwSteps) load_hi(IC<G + kCwHiDist>{});
            if constexpr (G + 1 < kCwSteps) load_ci(IC<G + 1>{});
            // which epilogue parts ride with this step (up to four): kc < 3 -> pair kc + 1 of the previous layer (one part per
            // step of a hidden layer, all four in a step of the output layer); last chunk -> pair 0 of THIS layer, which is
            // final after its own step (mp = 0)
            constexpr bool prev = s.kc < 3;
            constexpr int EL = prev ? s.L - 1 : s.L, EQ = prev ? s.kc + 1 : 0;
            constexpr CwParts parts = cw_parts(NT, s.L, s.kc, s.mp);
            constexpr int P0 = parts.first, NP = parts.count;
#ifndef CW_NO_PIN
            // the requests above stay above: ALU work may cross (mask: ALU | VALU | SALU | MFMA | transcendental), memory
            // operations may not -- under register pressure the scheduler otherwise sinks every request to just ahead of
            // its first use and the wave sits out the L2 / LDS latency fifty times per pass
            __builtin_amdgcn_sched_barrier(CW_PIN_MASK);
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[s.L & 1][2 * s.mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        lo_ring[G % (kCwLoDist + 1)][h], bch[t], acc[s.L & 1][2 * s.mp + h][t], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[s.L & 1][2 * s.mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        hi_ring[G % (kCwHiDist + 1)][h], bcl[t], acc[s.L & 1][2 * s.mp + h][t], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[s.L & 1][2 * s.mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        hi_ring[G % (kCwHiDist + 1)][h], bch[t], acc[s.L & 1][2 * s.mp + h][t], 0, 0, 0);
            static_for<0, NP>([&](auto ic) { epart(IC<EL>{}, IC<EQ>{}, IC<P0 + decltype(ic)::value>{}); });
            if constexpr (s.mp == MP - 1 && !(s.L == 4 && s.kc == 3)) {   // chunk done: the fragments produced meanwhile are next
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    bch[t] = bnh[t];
                    bcl[t] = bnl[t];
                }
            }
            if constexpr (G == 15) clk.mark(2);
            if constexpr (G == 31) clk.mark(3);
            if constexpr (G == 47) clk.mark(4);
        });
        clk.mark(5);
        // ---- (4) per point, on its four lanes: gates, hierarchical softmax, T = sum_j w_j A_j, residual, Broyden step
        // (RFU:54-113, 147-167, utils/utils.py:138-181, broyden.py:44-78).  See the header for how values cross lanes.
        hand_over();     // last pass's requests have landed long ago
        request(cl);     // this pass's claims: a whole pass to land in
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const float selA = ((j & 3) == g) ? 1.0f : 0.0f, sumA = g < 3 ? 1.0f : 0.0f;
        // What one lane of a point has and the other three need.  Round 4 moved these copies from LDS rows (write, wave barrier,
        // read) to the matrix pipe: v_mfma_f32_16x16x4_f32 with A[i][k] = (i mod 4 == k) hands every lane of point n the four
        // values B[0..3][n] its lane groups supplied, with A[i][k] = (k < 3) their sum over three rows.  Exact, no LDS traffic --
        // but 40 fp32 MFMAs per pass, 32 cycles each, on the pipe the OTHER wave of the SIMD needs for its MLP, and matrix time
        // is what a pass is made of (vector work hides to 45 % behind MFMAs, matrix work does not hide at
        // all).  Round 5: ds_bpermute_b32 -- the LDS crossbar, no LDS memory, no barrier: lane (j, g) reads the value of lane
        // (j, r) for r = 0..3.  Same bits (a copy is a copy; the three-row sum adds in the MFMA's row order).  Alternating
        // passes on one box: 14.66 -> 14.24 ms per launch, 38.55 -> 38.2 ms per frame (gpurun_out/r5l).  -DCW_GATHER_MFMA
        // restores the matrix-pipe form.
#ifndef CW_GATHER_MFMA
        const int sh0 = (j) << 2, sh1 = (j + 16) << 2, sh2 = (j + 32) << 2, sh3 = (j + 48) << 2;
        auto bperm = [&](int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); };
        auto gather = [&](float v) { return f32x4{bperm(sh0, v), bperm(sh1, v), bperm(sh2, v), bperm(sh3, v)}; };
        auto sum3 = [&](float v) {
            const float t = (bperm(sh0, v) + bperm(sh1, v)) + bperm(sh2, v);
            return f32x4{t, t, t, t};
        };
        (void)selA;
        (void)sumA;
#else
        auto gather = [&](float v) { return __builtin_amdgcn_mfma_f32_16x16x4f32(selA, v, zero4, 0, 0, 0); };
        auto sum3 = [&](float v) { return __builtin_amdgcn_mfma_f32_16x16x4f32(sumA, v, zero4, 0, 0, 0); };
#endif
        f32x4 p0[NT], p1[NT], s0[NT];
        float tgt[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {   // the point's state: in flight under the gates
            p0[t] = *reinterpret_cast<const f32x4*>(P0[t]);
            p1[t] = *reinterpret_cast<const f32x4*>(P1[t]);
            s0[t] = *reinterpret_cast<const f32x4*>(S0[t]);
            tgt[t] = S1[t][g2];
        }
        // (a) gates of the lane's own eight logits (channels 4 g + r and 16 + 4 g + r), in half-revolution-free base 2:
        //     u = 20 log2(e) logit, sigmoid = 1 / (1 + 2^-u) (2^-u = inf gives 0, 0 gives 1: no clamps), and the two 3-way
        //     softmaxes where their logits live -- channels 1..3 in lane group 0, 12..14 in lane group 3
        f32x4 gq[NT][2];
        float pq[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float u[2][4];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    u[h][r] = acc[0][h][t][r] * c20;
                    gq[t][h][r] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u[h][r]));
                }
            // lane group 0: (u1, u2, u3); lane group 3: (u0, u1, u2) of its first quad
            const float a0 = g == 0 ? u[0][1] : u[0][0], a1 = g == 0 ? u[0][2] : u[0][1], a2 = g == 0 ? u[0][3] : u[0][2];
            const float m = fmaxf(a0, fmaxf(a1, a2));
            const float e0 = __builtin_amdgcn_exp2f(a0 - m), e1 = __builtin_amdgcn_exp2f(a1 - m), e2 = __builtin_amdgcn_exp2f(a2 - m);
            const float sum = e0 + e1 + e2;   // in [1, 3]
            float rs = __builtin_amdgcn_rcpf(sum);
            rs = fmaf(fmaf(-sum, rs, 1.0f), rs, rs);
            pq[t][0] = e0 * rs;
            pq[t][1] = e1 * rs;
            pq[t][2] = e2 * rs;
        }
        // (b) every lane of a point gets all 25 gates and both softmaxes: 11 gathers per N-tile
        float w[NT][24];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 G0[4], G1[4], GP[3];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                G0[r] = gather(gq[t][0][r]);   // [q]: channel 4 q + r
                G1[r] = gather(gq[t][1][r]);   // [q]: channel 16 + 4 q + r
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) GP[r] = gather(pq[t][r]);   // [0]: softmax of 1..3, [3]: softmax of 12..14
            float sgm[25];
#pragma unroll
            for (int c = 0; c < 25; ++c) sgm[c] = c < 16 ? G0[c & 3][c >> 2] : G1[c & 3][(c & 15) >> 2];
            hsoftmax_tree<true>(sgm, GP[0][0], GP[1][0], GP[2][0], GP[0][3], GP[1][3], GP[2][3], w[t]);
        }
        // (c) T on the matrix pipe: D[entry][point] += bones[joint][entry] w[joint][point] in six fp32 steps of four joints
        //     (an fmaf chain in joint order, like the loop it replaces) -- lane (j, g) supplies bones[4 s + g][entry j] and
        //     w[4 s + g] of its point and receives entries 4 g .. 4 g + 3 = row g of T
        f32x4 Trow[NT];
        // w[4 s + g] by selects on the two bits of g, written as selects: indexed by g the compiler would read w from a
        // scratch copy.  Round 3 wrote the three v_cndmask as inline asm; the compiler cannot see that an asm statement
        // writes an MFMA operand, and v_mfma_f32_16x16x4_f32 needs TWO wait states behind a VALU write of its B register
        // (tools/ubench/valu_mfma32_hazard.hip: distance 0 and 1 read the old register, an s_waitcnt that has nothing to wait
        // for is one wait state, not two).  Whenever the scheduler put the fourth step -- joints 12..15 -- right behind its
        // select, points skinned to the head, neck and collars missed their roots (profiles/r04_hazard_ubench.txt).
        const bool g_odd = (g & 1) != 0, g_high = (g & 2) != 0;
        float bone[6];
#pragma unroll
        for (int s6 = 0; s6 < 6; ++s6) bone[s6] = sbones[(4 * s6 + g) * 16 + j];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 tr = zero4;
#pragma unroll
            for (int s6 = 0; s6 < 6; ++s6) {
                const float wl = g_odd ? w[t][4 * s6 + 1] : w[t][4 * s6], wh = g_odd ? w[t][4 * s6 + 3] : w[t][4 * s6 + 2];
                const float wb = g_high ? wh : wl;
                tr = __builtin_amdgcn_mfma_f32_16x16x4f32(bone[s6], wb, tr, 0, 0, 0);
            }
            Trow[t] = tr;
        }
        // (d) residual component g and what the Broyden step needs from the other rows
        f32x4 gv[NT], gpv[NT], Mc[NT][3], vT[NT][3];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const f32x4 xv = gather(p0[t][0]);
            const float gn = fmaf(Trow[t][0], xv[0], fmaf(Trow[t][1], xv[1], fmaf(Trow[t][2], xv[2], Trow[t][3]))) - tgt[t];
            gv[t] = gather(gn);
            gpv[t] = gather(p0[t][3]);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Mc[t][c] = gather(Trow[t][c]);            // [q]: T[q][c] (first evaluation: J^-1_0 = (T[:3,:3])^-1, RFU:327-328)
                vT[t][c] = sum3(p0[t][1] * p1[t][c]);     // v^T[c] = sum_r dx_r J^-1[r][c] (broyden.py:69)
            }
        }
        // (e) the step, lane g holding row g (lane group 3 computes along on row 2's state and writes nothing)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const bool live = id[t] >= 0;
            const int nev = __float_as_int(s0[t][1]);
            const bool first = nev == 0;
            const float x = p0[t][0], dx = p0[t][1];
            float xb = p0[t][2], eb = s0[t][0];
            float Jr[3] = {p1[t][0], p1[t][1], p1[t][2]};
            float gx[3], stp;
            bool keep, improved = false;
            if (first) {
                const float T9[16] = {Mc[t][0][0], Mc[t][1][0], Mc[t][2][0], 0.f, Mc[t][0][1], Mc[t][1][1], Mc[t][2][1], 0.f,
                                      Mc[t][0][2], Mc[t][1][2], Mc[t][2][2], 0.f, 0.f, 0.f, 0.f, 1.f};
                float R[9];
                inv3_of44(T9, R);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Jr[c] = g2 == 0 ? R[c] : (g2 == 1 ? R[3 + c] : R[6 + c]);
                    gx[c] = gv[t][c];
                }
                eb = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
                keep = true;                                        // every point takes at least one step
                // a FIRST residual (at the nearest-vertex start) that is not finite: an activation left the f16 range
                // (later ones also come from iterates that genuinely diverge; either way the point retires below)
                if (live && g == 0 && !(fabsf(gx[0] + gx[1] + gx[2]) < 3.0e38f)) ++n_bad;
            } else {
                float dg[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dg[c] = gv[t][c] - gpv[t][c];
                    gx[c] = gpv[t][c] + dg[c];                      // broyden.py:50-51
                }
                const float err = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
                improved = err < eb;                                // broyden.py:54-61
                if (improved) {
                    eb = err;
                    xb = x;
                }
                keep = (eb > kRootThresh) && (err < kDvg);          // broyden.py:64
                // broyden.py:69-75 on row g: a_g = dx_g - J[g] dg, b = v^T dg (+-eps), J[g] += (a_g / b) v^T
                const float v0 = vT[t][0][0], v1 = vT[t][1][0], v2 = vT[t][2][0];
                const float a_g = dx - (Jr[0] * dg[0] + Jr[1] * dg[1] + Jr[2] * dg[2]);
                float bq = v0 * dg[0] + v1 * dg[1] + v2 * dg[2];
                bq = (bq >= 0.f) ? bq + 1e-6f : bq - 1e-6f;
                const float uq = a_g / bq;
                Jr[0] += uq * v0;
                Jr[1] += uq * v1;
                Jr[2] += uq * v2;
            }
            stp = -(Jr[0] * gx[0] + Jr[1] * gx[1] + Jr[2] * gx[2]);
            if (nev + 1 > kBroydenSteps) keep = false;              // 1 + 50 evaluations (broyden.py:44)
            if (improved) tbr[t] = Trow[t];
            if (live) {
                if (keep) {
                    if (g < 3) {
                        *reinterpret_cast<f32x4*>(P0[t]) = f32x4{x + stp, stp, xb, gx[g2]};
                        *reinterpret_cast<f32x4*>(P1[t]) = f32x4{Jr[0], Jr[1], Jr[2], 0.f};
                    }
                    if (g == 0) *reinterpret_cast<f32x4*>(S0[t]) = f32x4{eb, __int_as_float(nev + 1), __int_as_float(id[t]), 0.f};
                } else {   // retire: the best iterate is the result (broyden.py:78)
                    if (g < 3) outp.pts[(size_t)id[t] * 3 + g] = xb;
                    reinterpret_cast<f32x4*>(outp.T + (size_t)id[t] * 16)[g] = tbr[t];
                    if (g == 0) {
                        outp.err[id[t]] = eb;
                        *reinterpret_cast<f32x4*>(S0[t]) = f32x4{eb, __int_as_float(nev + 1), __int_as_float(-1), 0.f};
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        clk.mark(6);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_bad += __shfl_xor(n_bad, o);
    if (lane == 0) {
        count_add(ctr, n_eval);
        count_add(ctr_canon, n_eval);
        count_add(ctr_bad, n_bad);
    }
#ifdef ARAH_CLOCKS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&clk_out[wave * 16 + i], (unsigned long long)clk.acc[i]);
#endif
}
