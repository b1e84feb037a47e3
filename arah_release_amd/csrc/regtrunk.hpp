// regtrunk.hpp -- the forward FiLM-SIREN trunk of the split engine with POINT-OWNING waves (round 5).
//
// Included by arah_hip.hip inside its anonymous namespace.  The tile kernels (mlp.hpp: sdf_trunk) keep a 128-point tile's
// activations in LDS as hi / lo planes, let eight waves slice the CHANNELS of a layer and pay, per layer, a matrix phase
// (every wave reads the whole tile's B fragments from LDS), a vector phase (the FiLM sine, the hi / lo split and the
// stores back into LDS) and two workgroup barriers -- the matrix pipe is 46 % busy (profiles/r05b_pmc_sq.json).  Here, as in
// loop C's kernel (canon_wave.hpp), a wave owns 32 points and ALL 256 channels:
//
//   * lane (j, g) of a 16x16 accumulator tile holds rows 4g + r of point j.  M-tile mt computes the units
//         unit(mt, row) = 32 (mt >> 1) + 8 (row >> 2) + 4 (mt & 1) + (row & 3),
//     so the eight values a lane gets from the M-tile pair (2q, 2q + 1) are exactly its share (k = 8g .. 8g + 7) of B chunk q
//     of the next layer, in the standard channel order: a layer's output never leaves the registers, and every unit's dot
//     product runs over the same k order, in the same MFMA positions, as in sdf_trunk -- the results are BIT-IDENTICAL to the
//     tile kernels' (the density pass and the shading pass must agree on every sample);
//   * the loop order is chunk-outer: all sixteen M-tiles of a layer accumulate (128 registers) while the B chunks of its input
//     are produced just in time -- the epilogue (FiLM sine, hi / lo split) of the previous layer's M-tile pair kc + 1 rides
//     between the MFMAs of chunk kc (pair 0 of the layer itself in its last chunk, where it is final first);
//   * the A operands are the whole layer (256 KB as hi + lo halves): they stream through a two-slot ring in LDS, one chunk
//     (the sixteen M-tiles' hi and lo fragments of 32 input channels, 32 KB) per slot, filled by global_load_lds_dwordx4 from
//     the frame's ordinary split packing (SdfNet::wps) with the row permutation above applied by the loading lane's source
//     address -- nothing new is packed per frame, no staging registers.  Four waves per workgroup (one per SIMD: 2 x 128
//     accumulator registers), ONE barrier per chunk.  L2 sees what the 128-point tile kernel asks of it: a layer per 128 points;
//   * the input layer (K = 3, vector ALU) writes its pre-activations into the accumulator registers and goes through the same
//     epilogue; the head (w6 . h6 + b6) runs in registers in the chain order sdf_head<true> takes in these builds.
//
// MEASURED, NOT SHIPPED (profiles/r05_reg_trunk.txt): bit-identical images, 16.7 - 17.5 ms per density pass against 14.4 for
// k_density<true, 8>.  With one wave per SIMD nothing hides behind a 16-cycle MFMA: every ds_read, vector instruction and
// v_accvgpr move adds its ~5 cycles (a layer takes 22 k cycles where its MFMAs are 12.3 k).  The file is compiled only with
// -DARAH_REG_TRUNK (tools/build_variant.sh reg -DARAH_REG_TRUNK; ARAH_DENSITY_REG=1 then selects k_density_reg).
#pragma once

#ifndef RT_PIN_MASK
#define RT_PIN_MASK 0x040f
#endif
constexpr int kRtWaves = 4;
constexpr int kRtThreads = kRtWaves * 64;
constexpr int kRtNT = 2;                            // N-tiles (16 points) per wave
constexpr int kRtTile = kRtWaves * 16 * kRtNT;      // 128 points per workgroup pass
constexpr unsigned kRtChunk = 32768;                // sixteen M-tiles x (hi, lo) x 1 KB
constexpr unsigned kRtRing = 0;                     // [2][kRtChunk]: inside the offset field of a DS instruction
constexpr unsigned kRtFilm = 2 * kRtChunk;          // [6][fws 256 | pw 256] floats
constexpr unsigned kRtW6 = kRtFilm + 6 * 2048;      // [256] floats
constexpr unsigned kRtW0 = kRtW6 + 1024;            // [256][4] floats
constexpr unsigned kRtXin = kRtW0 + 4096;           // [128][4] floats
constexpr unsigned kRtIds = kRtXin + kRtTile * 16;  // [128] ints
constexpr unsigned kRtOut = kRtIds + kRtTile * 4;   // [128] floats
constexpr size_t kLdsRegTrunk = kRtOut + kRtTile * 4;
static_assert(kLdsRegTrunk <= 160 * 1024, "one workgroup per CU");

// Consumer of a tile's SDF values (normalised units, b6 added), called by every thread after the trunk: out[128] in LDS.
// io.load(tile, tid, ids, xin): threads tid < 128 fill ids[tid] (-1: no point) and xin[tid] (normalised point).
#ifdef RT_CLOCKS
typedef PhaseClk RtClk;
#else
typedef NoClk RtClk;
#endif

template <typename IO>
__device__ __forceinline__ void rt_run(const FrameDev& fr, char* smem, int n_tiles, IO& io, unsigned long long* clk_out = nullptr) {
    RtClk clk;
    clk.start();
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr int NT = kRtNT;
    const SdfNet& net = fr.sdf;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    constexpr float amp = kActScale;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* film = reinterpret_cast<float*>(smem + kRtFilm);
    float* w6s = reinterpret_cast<float*>(smem + kRtW6);
    float* w0s = reinterpret_cast<float*>(smem + kRtW0);
    float* xin = reinterpret_cast<float*>(smem + kRtXin);
    int* ids = reinterpret_cast<int*>(smem + kRtIds);
    float* outv = reinterpret_cast<float*>(smem + kRtOut);

    // ---- launch prologue: per-channel constants into LDS
    for (int i = tid; i < 6 * 256; i += kRtThreads) {
        const int li = i >> 8, u = i & 255;
        film[li * 512 + u] = li == 0 ? net.fw[u] : net.fws[i];
        film[li * 512 + 256 + u] = net.pw[i];
    }
    for (int i = tid; i < 256; i += kRtThreads) {
        w6s[i] = net.w6[i] * kInvActScale;   // exact: the head sums w6 (hi + lo) / 1024 as two fmas per unit (sdf_head<true>)
        reinterpret_cast<f32x4*>(w0s)[i] = reinterpret_cast<const f32x4*>(net.w0)[i];
    }

    // ---- the weight ring.  A chunk (layer k, input channels 32 kc ..): fragment f = 2 mt + s (s = 0 hi, 1 lo) of the slot holds
    // lane (row, kg) <- W[unit(mt, row)][32 kc + 8 kg ..], i.e. lane 16 kg + (unit & 15) of fragment ((unit >> 4) 8 + kc) 2 + s of
    // the ordinary packing.  Wave w moves fragments 8w .. 8w + 7 (q = 0 .. 7: mt = 4w + (q >> 1), s = q & 1), each as ONE
    // global_load_lds_dwordx4: 16 bytes per lane from the lane's (permuted) source address straight into LDS at the wave's
    // base + 16 lane -- the fragment order; no staging registers, no store pass (issue -> landed 250-400 cycles from L2,
    // MI355X_MICROARCH.md; a chunk's multiplications take ~1700).  The barrier that ends a chunk waits for them (vmcnt(0)).
    const unsigned row = lane & 15, kg = lane >> 4;
    const unsigned src_thr = (4u * wave + (row >> 3)) * 16384u + (16u * kg + 8u * ((row >> 2) & 1u) + (row & 3u)) * 16u;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    char* dst_wave = smem + kRtRing + (8u * wave) * 1024u;
    auto issue = [&](const f16x8* wl, auto kcc, auto slotc) {   // chunk kc of the layer whose split packing is wl -> slot
        constexpr int kc = decltype(kcc)::value, slot = decltype(slotc)::value;
        const char* src = reinterpret_cast<const char*>(wl) + src_thr;
#ifdef RT_ABL_NO_FILL   // timing ablations (results are wrong with them): no weight stream
        if (src_thr != 0xffffffffu) return;
#endif
        static_for<0, 8>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            constexpr unsigned off = (q >> 2) * 32768 + kc * 2048 + (q & 1) * 1024 + ((q >> 1) & 1) * 64;
            __builtin_amdgcn_global_load_lds((glb_void*)(src + off), (lds_void*)(dst_wave + slot * kRtChunk + q * 1024), 16, 0, 0);
        });
    };
    issue(net.wps[0], IC<0>{}, IC<0>{});
    __syncthreads();

    const char* a_thr = smem + kRtRing + lane * 16u;
    auto ld_a = [&](auto slotc, auto mtc, auto sc) -> f16x8 {
        constexpr int slot = decltype(slotc)::value, mt = decltype(mtc)::value, s = decltype(sc)::value;
        return *reinterpret_cast<const f16x8*>(a_thr + slot * kRtChunk + (mt * 2 + s) * 1024);
    };

    f32x4 accA[16][NT], accB[16][NT];   // two sets: a layer accumulates into one while the other holds its input's finals
    f16x8 bch[NT], bcl[NT], bnh[NT], bnl[NT];
    u32x4 pkh[NT], pkl[NT];
    // part p (N-tile t = p >> 1, M-tile 2q + h, h = p & 1) of the epilogue of M-tile pair q: FiLM sine of layer index li
    // (0: the input layer) and the hi / lo split, the arithmetic of sdf_trunk's epilogue (film_sine + store_split4)
    auto epart = [&](const f32x4 (&src)[16][NT], const char* fbl, auto qc, auto pc) {
        constexpr int q = decltype(qc)::value, p = decltype(pc)::value;
        constexpr int t = p >> 1, h = p & 1, mt = 2 * q + h;
        const f32x4 fw = *reinterpret_cast<const f32x4*>(fbl + (32 * q + 4 * h) * 4);
        const f32x4 pw = *reinterpret_cast<const f32x4*>(fbl + 1024 + (32 * q + 4 * h) * 4);
        f32x4 hv, d;
        film_sine<false>(src[mt][t], fw, pw, zero4, amp, hv, d);
        const float v[4] = {hv[0], hv[1], hv[2], hv[3]};
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
    const char* fb = smem + kRtFilm + 32 * g;   // + li * 2048: the lane's row of the FiLM table of layer index li

    // chunk kc of layer k: 96 MFMAs (pair by pair), the ring's upkeep, and the epilogue that produces the next chunk's B
    auto chunk = [&](auto kcc, const f32x4 (&accP)[16][NT], f32x4 (&accC)[16][NT], const f16x8* w_cur, const f16x8* w_next, const char* fbl) {
        constexpr int kc = decltype(kcc)::value, slot = kc & 1;
        if constexpr (kc + 1 < 8) issue(w_cur, IC<kc + 1>{}, IC<(kc + 1) & 1>{});
        else issue(w_next, IC<0>{}, IC<0>{});
        f16x8 ah[2][2], al[2][2];
        ah[0][0] = ld_a(IC<slot>{}, IC<0>{}, IC<0>{});
        al[0][0] = ld_a(IC<slot>{}, IC<0>{}, IC<1>{});
        ah[0][1] = ld_a(IC<slot>{}, IC<1>{}, IC<0>{});
        al[0][1] = ld_a(IC<slot>{}, IC<1>{}, IC<1>{});
        // the requests above stay above (ALU, MFMA and transcendental work may cross, memory operations may not): left alone the
        // scheduler sinks the L2 requests to the end of the chunk and the next chunk's stores wait out an L2 round trip
        __builtin_amdgcn_sched_barrier(RT_PIN_MASK);
        static_for<0, 8>([&](auto mpc) {
            constexpr int mp = decltype(mpc)::value, cur = mp & 1;
            if constexpr (mp + 1 < 8) {
                ah[cur ^ 1][0] = ld_a(IC<slot>{}, IC<2 * mp + 2>{}, IC<0>{});
                al[cur ^ 1][0] = ld_a(IC<slot>{}, IC<2 * mp + 2>{}, IC<1>{});
                ah[cur ^ 1][1] = ld_a(IC<slot>{}, IC<2 * mp + 3>{}, IC<0>{});
                al[cur ^ 1][1] = ld_a(IC<slot>{}, IC<2 * mp + 3>{}, IC<1>{});
            }
            __builtin_amdgcn_sched_barrier(RT_PIN_MASK);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    accC[2 * mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[cur][h], bch[t], kc == 0 ? zero4 : accC[2 * mp + h][t], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    accC[2 * mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][h], bcl[t], accC[2 * mp + h][t], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    accC[2 * mp + h][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[cur][h], bch[t], accC[2 * mp + h][t], 0, 0, 0);
            // the four parts of the next chunk's B ride with steps 1, 3, 5, 7
#ifndef RT_ABL_NO_EPI   // no epilogue parts riding with the chunks
            if constexpr (mp & 1) {
                if constexpr (kc < 7) epart(accP, fbl, IC<kc + 1>{}, IC<(mp >> 1)>{});
                else epart(accC, fbl + 2048, IC<0>{}, IC<(mp >> 1)>{});   // pair 0 of this layer: final since step 0
            }
#endif
        });
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            bch[t] = bnh[t];
            bcl[t] = bnl[t];
        }
#ifdef RT_ABL_NO_BARRIER   // no barrier at the end of a chunk
        __builtin_amdgcn_wave_barrier();
#else
        __syncthreads();
#endif
    };

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        io.load(tile, tid, ids, xin);
        __syncthreads();
        // ---- input layer: pre-activations of all 256 units into accP (the fmaf chain of sdf_trunk), pair 0's epilogue -> B chunk 0
        {
            f32x4 x[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) x[t] = *reinterpret_cast<const f32x4*>(xin + (wave * 32 + t * 16 + j) * 4);
            static_for<0, 16>([&](auto mtc) {
                constexpr int mt = decltype(mtc)::value;
                const int u0 = 32 * (mt >> 1) + 8 * g + 4 * (mt & 1);
                f32x4 w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(w0s + (u0 + r) * 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaf(w[r][2], x[t][2], fmaf(w[r][1], x[t][1], w[r][0] * x[t][0]));
                    no_pack(v);
                    accA[mt][t] = v;
                }
            });
            static_for<0, 2 * NT>([&](auto pc) { epart(accA, fb, IC<0>{}, pc); });
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                bch[t] = bnh[t];
                bcl[t] = bnl[t];
            }
        }
        clk.mark(0);
        // ---- layers 2 .. 6
#ifdef RT_TWO_SETS   // no copies between layers, two layer bodies: measured SLOWER (499 registers: 17.8 against 16.7 ms, profiles/r05_reg_trunk.txt)
#pragma unroll 1
        for (int k = 1;; k += 2) {   // A -> B (k = 1, 3, 5), B -> A (k = 2, 4): no copies between layers
            {
                const f16x8* w_cur = net.wps[k - 1];
                const f16x8* w_next = net.wps[k < 5 ? k : 0];
                const char* fbl = fb + (k - 1) * 2048;
                static_for<0, 8>([&](auto kcc) { chunk(kcc, accA, accB, w_cur, w_next, fbl); });
                clk.mark(k);
            }
            if (k == 5) break;
            {
                const f16x8* w_cur = net.wps[k];
                const f16x8* w_next = net.wps[k + 1];
                const char* fbl = fb + k * 2048;
                static_for<0, 8>([&](auto kcc) { chunk(kcc, accB, accA, w_cur, w_next, fbl); });
                clk.mark(k + 1);
            }
        }
#else
#pragma unroll 1
        for (int k = 1; k < 6; ++k) {   // every layer accumulates into B; its finals move to A behind its last chunk
            const f16x8* w_cur = net.wps[k - 1];
            const f16x8* w_next = net.wps[k < 5 ? k : 0];
            const char* fbl = fb + (k - 1) * 2048;
            static_for<0, 8>([&](auto kcc) { chunk(kcc, accA, accB, w_cur, w_next, fbl); });
            if (k < 5)
                static_for<2, 16>([&](auto mtc) {   // pair 0 went into B chunk 0 already
                    constexpr int mt = decltype(mtc)::value;
#pragma unroll
                    for (int t = 0; t < NT; ++t) accA[mt][t] = accB[mt][t];
                });
            clk.mark(k);
        }
#endif
        // ---- the last layer's epilogue and the head, per N-tile, in registers: lane (j, g) holds units 32 q + 8 g + e of point j
        // and runs its eight chains (g, e) over q exactly as sdf_head<true> does -- two fmas per unit (hi, then lo half)
        static_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            float sc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sc[e] = 0.f;
            static_for<0, 8>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                f16x8 hh, ll;
                if constexpr (q == 0) {
                    hh = bch[t];
                    ll = bcl[t];
                } else {
                    epart(accB, fb + 5 * 2048, qc, IC<2 * t>{});
                    epart(accB, fb + 5 * 2048, qc, IC<2 * t + 1>{});
                    hh = bnh[t];
                    ll = bnl[t];
                }
                const f32x4 wa = *reinterpret_cast<const f32x4*>(w6s + 32 * q + 8 * g);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(w6s + 32 * q + 8 * g + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float w = e < 4 ? wa[e & 3] : wb[e & 3];
                    sc[e] = fmaf(w, (float)hh[e], sc[e]);
                    sc[e] = fmaf(w, (float)ll[e], sc[e]);
                }
            });
            float tg = ((sc[0] + sc[1]) + (sc[2] + sc[3])) + ((sc[4] + sc[5]) + (sc[6] + sc[7]));
            tg += __shfl_xor(tg, 16);   // lane group 0: (t0 + t1) + (t2 + t3)
            tg += __shfl_xor(tg, 32);
            if (g == 0) outv[wave * 32 + t * 16 + j] = tg + net.b6[0];
        });
        __syncthreads();
        clk.mark(6);
        io.store(tile, tid, ids, outv);
        __syncthreads();
        clk.mark(7);
    }
#ifdef RT_CLOCKS
    if (clk_out && lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&clk_out[wave * 16 + i], (unsigned long long)clk.acc[i]);
#endif
}

// loop D, pass 1 on the point-owning trunk: what k_density<true, 8> computes, sample for sample and bit for bit
struct RtDensityIO {
    const float* pts;
    const int* list;
    int n;
    f32x4* shaded;
    int* next_list;
    int* next_count;
    unsigned long long *ctr_fwd, *ctr_dens;
    float scale, inv_beta;
    __device__ __forceinline__ void load(int tile, int tid, int* ids, float* xin) const {
        if (tid < kRtTile) {
            const int i = tile * kRtTile + tid;
            const int id = i < n ? list[i] : -1;
            ids[tid] = id;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) x = f32x4{pts[(size_t)id * 3], pts[(size_t)id * 3 + 1], pts[(size_t)id * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
    }
    __device__ __forceinline__ void store(int tile, int tid, const int* ids, const float* outv) const {
        if (tid == 0) {
            count_add(ctr_fwd, min(kRtTile, n - tile * kRtTile));
            count_add(ctr_dens, min(kRtTile, n - tile * kRtTile));
        }
        if (tid < kRtTile) {   // whole waves
            const int id = ids[tid];
            bool keep = false;
            if (id >= 0) {
                const float dens = volsdf_density(outv[tid] * scale, inv_beta);
                shaded[id] = f32x4{0.f, 0.f, 0.f, dens};
                keep = dens > 0.f;
            }
            append_ids(keep, id, next_list, next_count);
        }
    }
};

__global__ __launch_bounds__(kRtThreads) __attribute__((amdgpu_waves_per_eu(1, 1)))
void k_density_reg(FrameDev fr, const float* pts, const int* list, const int* count, f32x4* shaded, int* next_list,
                   int* next_count, unsigned long long* ctr_fwd, unsigned long long* ctr_dens, unsigned long long* clk_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_rt[];
    const BodyConst bc = load_bc(fr);
    const int n = *count;
    const int n_tiles = (n + kRtTile - 1) / kRtTile;
    if ((int)blockIdx.x >= n_tiles) return;
    RtDensityIO io{pts, list, n, shaded, next_list, next_count, ctr_fwd, ctr_dens, sdf_scale(bc),
                   1.0f / fminf(fmaxf(fabsf(load_beta(fr)), 1e-6f), 1e6f)};
    rt_run(fr, smem_rt, n_tiles, io, clk_out);
}
