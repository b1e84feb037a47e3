// train.hpp -- loop D WITH GRADIENTS (training, reference implicit_differentiable_renderer.py:261-396 with
// self.training set): per-sample forward (SDF value / feature, normal by reverse sweep, colour MLP) and the
// hand-written backward of all of it, including the second-order path through the normal (IDR:336-338 builds it
// with create_graph=True; diff_operators.py:39-50).  Included by arah_hip.hip (uses its FrameDev / helpers).
//
// What is differentiated, per valid sample with canonical (normalised) point x:
//     v_k = W_k h_{k-1},  z_k = 30 (f_k (v_k + b_k) + phi_k),  h_k = sin z_k   (k = 1..6, h_0 = x)
//     s   = w_7 . h_6 + b_7                       (SDF, normalised units)
//     n   = ds/dx = W_1^T D_1 W_2^T D_2 ... D_6 w_7,   D_k = diag(30 f_k cos z_k)
//     rgb = sigmoid(colour MLP([h_6 | x | R n | PE(view) ] ; pose vector folded into two biases))
// Upstream gradients g_s = dL/ds and g_rgb = dL/drgb arrive from the compositing + loss, which stay in PyTorch
// (a handful of element-wise kernels on [rays, S] tensors).
//
// Backward of one 64-sample tile (everything on the exact fp32 MFMA engine):
//   1. recompute the forward (pre-activations v_k go to a per-workgroup slab, h_k / colour activations to streams);
//   2. colour MLP backward: delta_l = (W_{l+1}^T delta_{l+1}) * [c_l > 0], input gradient dCin = W_0^T delta_0 +
//      W_3a^T delta_3  ->  dL/dh_6, dL/dx (direct), nt = dL/dn (rotated back by R^T);
//   3. the normal enters L only through nt . n, and nt . n = w_7 . hd_6 with the TANGENT pass hd_0 = nt,
//      vd_k = W_k hd_{k-1}, hd_k = D_k vd_k -- so dL/dtheta through n is the gradient of that scalar: one reverse
//      sweep carries the adjoints of BOTH streams,
//          zb_k      = cos z_k * adj(h_k) - 30 f_k sin z_k * vd_k * adj(hd_k)          (= adj z_k)
//          adj(v_k)  = 30 f_k zb_k,          adj(vd_k) = 30 f_k cos z_k * adj(hd_k)
//          adj(h_{k-1}) = W_k^T adj(v_k),    adj(hd_{k-1}) = W_k^T adj(vd_k)
//          d f_k  += 30 ((v_k + b_k) zb_k + vd_k cos z_k adj(hd_k)),   d phi_k += 30 zb_k
//      seeded with adj(h_6) = dL/dh_6 (colour) + g_s w_7 and adj(hd_6) = w_7;
//   4. weight gradients are sums over ALL samples of outer products: the kernel streams their operands
//      (h_{k-1}, hd_{k-1}, adj v_k, adj vd_k; colour inputs and deltas) as dense [P][width] matrices and the host
//      finishes them with library GEMMs (dW_k = adj(v_k)^T h_{k-1} + adj(vd_k)^T hd_{k-1}, ...); the FiLM gradients
//      are reduced in the kernel (wave shuffles + one atomic per channel and wave).
#pragma once

namespace {

constexpr int kTrainSlabPerWg = 2 * 6 * kWaves * (kSdfMT * kNT) * 64;   // f32x4 elements: v_k and vd_k of six layers

struct TrainIn {
    int n;                    // P valid samples, dense
    const float* x;           // [P][3] normalised canonical points
    const float* T;           // [P][16] forward transforms (normal rotation when rotate_normal) or null
    const float* view;        // [P][3] view input of the colour net (already canonicalised / augmented by the host)
    const float* view_orig;   // [P][3] un-augmented view input (ray_augm) or null
    int rotate_normal;        // !cano_view_dirs (IDR:339-340)
    int ray_augm;             // IDR:342-350
    const float* g_s;         // [P]    dL/ds     (backward only)
    const float* g_rgb;       // [P][3] dL/drgb   (backward only)
    float* tap_cin;           // forward: colour input stream out (or null); backward with fwd_rgb: the same, read only
    float* tap_c[5];
    const float* fwd_rgb;     // [P][4] the forward call's rgb: the backward then skips the normal sweep and the colour MLP
    int geom_only;            // SDF value and normal only (the regulariser queries): no colour MLP; rgb <- the normal,
                              // g_rgb = dL/dn, the feature h_6 goes out as c[0] in the backward
};

struct TrainOut {
    float* sdf;               // [P]
    float* rgb;               // [P][4]
    float* gx;                // [P][4] dL/dx (backward)
    float* film_f;            // [6][256] dL/d freq  (atomics; caller zeroes)
    float* film_p;            // [6][256] dL/d phase
    // streams (backward), dense row-major [P][width]
    float* h[6];              // h_0 [P][4], h_1..h_5 [P][256]
    float* hd[7];             // hd_0 = nt [P][4], hd_1..hd_6 [P][256]
    float* av[6];             // adj v_1..v_6 [P][256]
    float* avd[6];            // adj vd_1..vd_6 [P][256]
    float* cin;               // [P][kInPad]
    float* c[5];              // c1..c5 (256, 256, 128, 256, 256)
    float* d[6];              // delta_0..delta_4 (256, 256, 128, 256, 256), delta_5 [P][4]
};

struct ColNetT {              // transposed packings of the colour MLP (reverse sweep)
    const float* w0pT;        // [kInPad][256]
    const float* w1pT;        // [256][256]
    const float* w2pT;        // [256][128]
    const float* w3apT;       // [kInPad][256]
    const float* w3bpT;       // [128][256]
    const float* w4pT;        // [256][256]
};

// (sin, cos) of z_k and 30 f_k for the four channels a lane owns
__device__ __forceinline__ void film_sincos(const f32x4 v, const f32x4 fw, const f32x4 pw, f32x4& sn, f32x4& cs) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a, b;
        sincospi_amp(fmaf(v[r], fw[r], pw[r]), 1.0f, a, b);
        sn[r] = a;
        cs[r] = b;
    }
}

// sum over the 64 points of a tile of per-lane partials p[m][r] (already summed over the lane's four N-tiles), then
// one atomic per channel: lanes j = 0 of each group g add to dst[ch]
__device__ __forceinline__ void reduce_channels(const float (&p)[kSdfMT][4], float* dst, int mt0, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s = p[m][r];
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            s += __shfl_xor(s, 4);
            s += __shfl_xor(s, 8);
            if (j == 0) atomicAdd(dst + (mt0 + m) * 16 + 4 * g + r, s);
        }
}

// B3: the backward-direction products (colour reverse sweep, tangent pass, both adjoint sweeps: 21 of the 34 layer products
// of the backward kernel) run on the bf16 x 3 engine (mlp.hpp: gemm_acc_b3); false: everything on the fp32 MFMA
// The call's two pointer tables (TrainIn: 20 words, TrainOut: 37 pointers) are NOT kernel-argument registers: every use
// reads its word from the kernarg segment where it stands (a scalar load behind an address the compiler cannot see through, so it
// neither hoists nor keeps them).  As by-value arguments they sat in ~110 scalar registers for the whole kernel, spilled into
// vector lanes and pushed 200 vector registers to scratch (331 scratch instructions in the idr backward instance).
struct TrainArgs {
    FrameDev fr;
    ColNetT ct;
    B3Nets b3;
    TrainIn in;
    TrainOut out;
    f32x4* spill_all;
    f32x4* slab_all;
};
template <typename Tp>
__device__ __forceinline__ const __attribute__((address_space(4))) Tp& karg_at(unsigned off) {
    typedef const __attribute__((address_space(4))) char c4;
    c4* p = (c4*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *reinterpret_cast<const __attribute__((address_space(4))) Tp*>(p + off);
}
#define TI karg_at<TrainIn>(offsetof(TrainArgs, in))
#define TO karg_at<TrainOut>(offsetof(TrainArgs, out))

template <bool IDR, bool BWD, bool B3>
__global__ __launch_bounds__(kThreads) void k_shade_train(TrainArgs ka) {
    const FrameDev& fr = ka.fr;
    const ColNetT& ct = ka.ct;
    const B3Nets& b3 = ka.b3;
    f32x4* const spill_all = ka.spill_all;
    f32x4* const slab_all = ka.slab_all;
    typedef ColDims<IDR> D;
#ifdef ARAH_TRAIN_FWD_FP32
    constexpr bool FWD_B3 = false;            // A/B: the forward's normal sweep and colour MLP on the fp32 MFMA (rounds 3-5)
#else
    constexpr bool FWD_B3 = B3;
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                        // [64][4]
    float* outv = xin + 64 * 4;               // [64][4] sdf, normal (canonical)
    float* rgbv = outv + 64 * 4;              // [64][4]
    float* gin = rgbv + 64 * 4;               // [64][4] g_rgb(3), g_s
    float* ntl = gin + 64 * 4;                // [64][4] nt = dL/dn (canonical)
    float* d5 = ntl + 64 * 4;                 // [64][4] delta_5
    float* A = d5 + 64 * 4;                   // [64][kLdA]
    float* B = A + 64 * D::kLdA;              // [64][260]
    constexpr int ldA = D::kLdA, ldB = kSdfLd;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int mt0 = wave * kSdfMT;
    const int n = TI.n;
    f32x4* spill = spill_all + (size_t)blockIdx.x * kSpillPerWg;
    f32x4* vslab = slab_all ? slab_all + (size_t)blockIdx.x * kTrainSlabPerWg : nullptr;        // v_k
    f32x4* vdslab = vslab ? vslab + 6 * kWaves * (kSdfMT * kNT) * 64 : nullptr;                 // vd_k
    const SdfNet& net = fr.sdf;
    auto slab_at = [&](int k, int m, int nn) { return ((k * kWaves + wave) * (kSdfMT * kNT) + m * kNT + nn) * 64 + lane; };
    for (int tile = blockIdx.x; (long long)tile * kTile < n; tile += gridDim.x) {
        const long long row0 = (long long)tile * kTile;
        const int rows = min(kTile, (int)(n - row0));
        if (tid < kTile) {
            f32x4 x = {0.f, 0.f, 0.f, 0.f}, gi = {0.f, 0.f, 0.f, 0.f};
            if (tid < rows) {
                const long long p = row0 + tid;
                x = f32x4{TI.x[p * 3], TI.x[p * 3 + 1], TI.x[p * 3 + 2], 0.f};
                if (BWD) gi = f32x4{TI.g_rgb[p * 3], TI.g_rgb[p * 3 + 1], TI.g_rgb[p * 3 + 2], TI.g_s[p]};
            }
            reinterpret_cast<f32x4*>(xin)[tid] = x;
            reinterpret_cast<f32x4*>(gin)[tid] = gi;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][kNT];
        const bool geom = TI.geom_only != 0;
        const bool handed = BWD && (geom || TI.fwd_rgb != nullptr);   // the forward call left cin, c1..c5 and rgb (or there is
        if (handed) {                                                 // no colour MLP): only the trunk is recomputed
            if constexpr (BWD) {
                TrainTap tap;
                tap.aslab = vslab;
                for (int k = 0; k < 6; ++k) tap.h[k] = TO.h[k];
                tap.row0 = row0;
                tap.rows = rows;
                sdf_trunk<false, kNT, B3, TrainTap>(net, xin, A, ldA, nullptr, dlast, wave, lane, tap);   // v_k, h_k only
                if (geom) {   // the feature stream of dw_7; the colour MLP's gradient into h_6 is zero
                    if constexpr (B3) unsplit_rows(A, ldA, tid);
                    __syncthreads();
                    stream_rows(A, ldA, 256, TO.c[0], row0, rows, tid);
                    __syncthreads();
                    for (int e = tid; e < kTile * 64; e += kThreads)
                        *reinterpret_cast<f32x4*>(A + (e >> 6) * ldA + (e & 63) * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
                } else
                for (int e = tid; e < kTile * 64; e += kThreads) {   // B <- c5
                    const int r = e >> 6, c4 = e & 63;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (r < rows) v = reinterpret_cast<const f32x4*>(TI.tap_c[4] + (row0 + r) * 256)[c4];
                    *reinterpret_cast<f32x4*>(B + r * ldB + c4 * 4) = v;
                }
                if (!geom && tid < kTile) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (tid < rows) v = reinterpret_cast<const f32x4*>(TI.fwd_rgb)[row0 + tid];
                    reinterpret_cast<f32x4*>(rgbv)[tid] = v;
                }
                __syncthreads();
            }
        } else {
        // ---- 1. forward: SDF trunk (+ taps), value, normal
        if constexpr (BWD) {
            TrainTap tap;
            tap.aslab = vslab;
            for (int k = 0; k < 6; ++k) tap.h[k] = TO.h[k];
            tap.row0 = row0;
            tap.rows = rows;
            sdf_trunk<true, kNT, B3, TrainTap>(net, xin, A, ldA, spill, dlast, wave, lane, tap);
        } else {
            sdf_trunk<true, kNT, B3>(net, xin, A, ldA, spill, dlast, wave, lane);
        }
        // B3 builds: the forward trunk on the f16 split engine (2^-22 per product, what inference runs); its feature planes
        // become fp32 rows for the stages behind it
        sdf_head<B3>(net, A, ldA, outv, 4, tid);
        if constexpr (B3) unsplit_rows(A, ldA, tid);
        // Round 6: the normal's reverse sweep of the FORWARD on the bf16 x 3 engine (B3 builds: what the eval forward's k_shade
        // runs, 2^-16 per product): a linear sweep, no gates in it; forward kernel 2.28 -> 1.76 ms.  The colour MLP stays on
        // the fp32 MFMA: its results pass ReLU gates and 2^-16 products flip ~1e-5 of them against fp32 arithmetic.  Since the
        // forward hands its activations to the backward (tap_c) the gradient would still be the exact gradient of the function
        // that was evaluated -- but a flipped gate moves one sample's whole contribution: with -DARAH_TRAIN_FWD_COLOR_B3
        // (forward kernel 1.24 ms) one element of col.lin2.bias' gradient in test_shade_samples_op_against_autograd is 0.47 %
        // of the tensor's scale off the fp32 restatement (bound 0.2 %); the reference-pinned F8 test passes either way.
        sdf_backward<FWD_B3>(net, B, ldB, spill, dlast, outv, 4, wave, lane, tid, &b3);
        __syncthreads();
        if (geom) {   // regulariser queries: value and normal are the outputs
            if (tid < rows) {
                const long long p = row0 + tid;
                TO.sdf[p] = outv[tid * 4];
                reinterpret_cast<f32x4*>(TO.rgb)[p] = f32x4{outv[tid * 4 + 1], outv[tid * 4 + 2], outv[tid * 4 + 3], 0.f};
            }
            __syncthreads();
            continue;
        }
        // ---- colour input extras behind the feature: x(3), n(3) (rotated), [PE4(view) 27], zero pad
        if (tid < kTile) {
            float* e = A + tid * ldA + 256;
            float nx = outv[tid * 4 + 1], ny = outv[tid * 4 + 2], nz = outv[tid * 4 + 3];
            float vx = 0.f, vy = 0.f, vz = 0.f;
            if (tid < rows) {
                const long long p = row0 + tid;
                if (TI.rotate_normal) {                               // IDR:340
                    const float* Tq = TI.T + p * 16;
                    const float ax = Tq[0] * nx + Tq[1] * ny + Tq[2] * nz;
                    const float ay = Tq[4] * nx + Tq[5] * ny + Tq[6] * nz;
                    const float az = Tq[8] * nx + Tq[9] * ny + Tq[10] * nz;
                    nx = ax;
                    ny = ay;
                    nz = az;
                }
                vx = TI.view[p * 3];
                vy = TI.view[p * 3 + 1];
                vz = TI.view[p * 3 + 2];
                if (TI.ray_augm) {                                    // IDR:342-350: arccos(n^ . v) >= pi/2  <=>  n . v <= 0
                    if (nx * vx + ny * vy + nz * vz <= 0.f) {
                        vx = TI.view_orig[p * 3];
                        vy = TI.view_orig[p * 3 + 1];
                        vz = TI.view_orig[p * 3 + 2];
                    }
                }
            }
            e[0] = xin[tid * 4];
            e[1] = xin[tid * 4 + 1];
            e[2] = xin[tid * 4 + 2];
            e[3] = nx;
            e[4] = ny;
            e[5] = nz;
            int k = 6;
            if (IDR) {                                                // embedder.py:6-51, multires 4
                e[6] = vx;
                e[7] = vy;
                e[8] = vz;
                k = 9;
                float f = 1.0f;
                for (int o = 0; o < 4; ++o) {
                    e[k + 0] = sinf(vx * f);
                    e[k + 1] = sinf(vy * f);
                    e[k + 2] = sinf(vz * f);
                    e[k + 3] = cosf(vx * f);
                    e[k + 4] = cosf(vy * f);
                    e[k + 5] = cosf(vz * f);
                    k += 6;
                    f *= 2.0f;
                }
            }
            for (; k < D::kInPad - 256; ++k) e[k] = 0.f;
        }
        __syncthreads();
        ColTap ctap;
        const bool taps = BWD || TI.tap_cin != nullptr;
        if (taps) {
            ctap.cin = BWD ? TO.cin : TI.tap_cin;
            for (int l = 0; l < 5; ++l) ctap.c[l] = BWD ? TO.c[l] : TI.tap_c[l];
            ctap.row0 = row0;
            ctap.rows = rows;
        }
#ifdef ARAH_TRAIN_FWD_COLOR_B3
        color_mlp<IDR, FWD_B3>(fr.col, A, B, rgbv, 4, wave, lane, tid, taps ? &ctap : nullptr, &b3);
#else
        color_mlp<IDR, false>(fr.col, A, B, rgbv, 4, wave, lane, tid, taps ? &ctap : nullptr);
#endif
        __syncthreads();
        if (tid < rows) {
            const long long p = row0 + tid;
            TO.sdf[p] = outv[tid * 4];
            reinterpret_cast<f32x4*>(TO.rgb)[p] = f32x4{rgbv[tid * 4], rgbv[tid * 4 + 1], rgbv[tid * 4 + 2], 0.f};
        }
        }   // !handed
        if constexpr (BWD) {
            if (geom) {   // nt = dL/dn as given, no direct dL/dx, adj h_6 starts at zero (set above)
                if (tid < kTile) {
                    const f32x4 nt = {gin[tid * 4], gin[tid * 4 + 1], gin[tid * 4 + 2], 0.f};
                    reinterpret_cast<f32x4*>(ntl)[tid] = nt;
                    reinterpret_cast<f32x4*>(rgbv)[tid] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (tid < rows) reinterpret_cast<f32x4*>(TO.hd[0])[row0 + tid] = nt;
                }
                __syncthreads();
            } else {
            // =========================================================== 2. colour MLP backward
            // B holds c5.  delta_5 = g_rgb * rgb (1 - rgb)
            if (tid < kTile) {
                f32x4 dl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float y = rgbv[tid * 4 + c];
                    dl[c] = gin[tid * 4 + c] * y * (1.0f - y);
                }
                reinterpret_cast<f32x4*>(d5)[tid] = dl;
                if (tid < rows) reinterpret_cast<f32x4*>(TO.d[5])[row0 + tid] = dl;
            }
            __syncthreads();
            // delta_4 = (W_5^T delta_5) * [c5 > 0], in place over c5 (every lane rewrites exactly what it read)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int ch0 = (mt0 + m) * 16 + 4 * g;
                f32x4 w5[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) w5[c] = *reinterpret_cast<const f32x4*>(fr.col.w5 + c * 256 + ch0);
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) {
                    const int pt = nn * 16 + j;
                    const f32x4 dl = *reinterpret_cast<const f32x4*>(d5 + pt * 4);
                    f32x4 c5 = *reinterpret_cast<const f32x4*>(B + pt * ldB + ch0), o;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        o[r] = c5[r] > 0.f ? fmaf(w5[2][r], dl[2], fmaf(w5[1][r], dl[1], w5[0][r] * dl[0])) : 0.f;
                    *reinterpret_cast<f32x4*>(B + pt * ldB + ch0) = o;
                }
            }
            __syncthreads();
            stream_rows(B, ldB, 256, TO.d[4], row0, rows, tid);
            // one step of the reverse sweep: dst(B) <- (W^T B) * [c > 0] with c read from its stream
            auto mask_store = [&](const f32x4 (&acc)[2][kNT], const float* cstream, int width) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int ch0 = (mt0 + m) * 16 + 4 * g;
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) {
                        const int pt = nn * 16 + j;
                        f32x4 c = {0.f, 0.f, 0.f, 0.f};
                        if (pt < rows) c = *reinterpret_cast<const f32x4*>(cstream + (row0 + pt) * width + ch0);
                        f32x4 o;
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[r] = c[r] > 0.f ? acc[m][nn][r] : 0.f;
                        *reinterpret_cast<f32x4*>(B + pt * ldB + ch0) = o;
                    }
                }
            };
            {   // delta_3 = (W_4^T delta_4) * [c4 > 0]
                f32x4 acc[2][kNT];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[m][nn]);
                gemm_any<B3, 16, 2>(ct.w4pT, b3.colT[5], mt0, B, ldB, acc, lane);
                __syncthreads();
                mask_store(acc, TO.c[3], 256);
            }
            __syncthreads();
            stream_rows(B, ldB, 256, TO.d[3], row0, rows, tid);
            {   // [dCin | dc3] = W_3^T delta_3: dCin part into A (kInPad wide), dc3 (128 wide) masked -> delta_2
                for (int mt = wave; mt < D::kKC0; mt += kWaves) {
                    f32x4 acc[1][kNT];
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[0][nn]);
                    gemm_any<B3, 16, 1>(ct.w3apT, b3.colT[3], mt, B, ldB, acc, lane);
                    const int ch0 = mt * 16 + 4 * g;
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) *reinterpret_cast<f32x4*>(A + (nn * 16 + j) * ldA + ch0) = acc[0][nn];
                }
                f32x4 acc3[1][kNT];
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) zero_acc(acc3[0][nn]);
                gemm_any<B3, 16, 1>(ct.w3bpT, b3.colT[4], wave, B, ldB, acc3, lane);
                __syncthreads();   // every wave is done reading delta_3
                const int ch0 = wave * 16 + 4 * g;
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) {
                    const int pt = nn * 16 + j;
                    f32x4 c = {0.f, 0.f, 0.f, 0.f};
                    if (pt < rows) c = *reinterpret_cast<const f32x4*>(TO.c[2] + (row0 + pt) * 128 + ch0);
                    f32x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = c[r] > 0.f ? acc3[0][nn][r] : 0.f;
                    *reinterpret_cast<f32x4*>(B + pt * ldB + ch0) = o;
                }
            }
            __syncthreads();
            stream_rows(B, ldB, 128, TO.d[2], row0, rows, tid);
            {   // delta_1 = (W_2^T delta_2) * [c2 > 0]   (K = 128)
                f32x4 acc[2][kNT];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[m][nn]);
                gemm_any<B3, 8, 2>(ct.w2pT, b3.colT[2], mt0, B, ldB, acc, lane);
                __syncthreads();
                mask_store(acc, TO.c[1], 256);
            }
            __syncthreads();
            stream_rows(B, ldB, 256, TO.d[1], row0, rows, tid);
            {   // delta_0 = (W_1^T delta_1) * [c1 > 0]
                f32x4 acc[2][kNT];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[m][nn]);
                gemm_any<B3, 16, 2>(ct.w1pT, b3.colT[1], mt0, B, ldB, acc, lane);
                __syncthreads();
                mask_store(acc, TO.c[0], 256);
            }
            __syncthreads();
            stream_rows(B, ldB, 256, TO.d[0], row0, rows, tid);
            // dCin += W_0^T delta_0
            for (int mt = wave; mt < D::kKC0; mt += kWaves) {
                f32x4 acc[1][kNT];
                const int ch0 = mt * 16 + 4 * g;
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) acc[0][nn] = *reinterpret_cast<const f32x4*>(A + (nn * 16 + j) * ldA + ch0);
                gemm_any<B3, 16, 1>(ct.w0pT, b3.colT[0], mt, B, ldB, acc, lane);
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) *reinterpret_cast<f32x4*>(A + (nn * 16 + j) * ldA + ch0) = acc[0][nn];
            }
            __syncthreads();
            // A[:, :256] = dL/dh_6 (colour path), A[:, 256..258] = dL/dx (direct), A[:, 259..261] = dL/d(R n)
            if (tid < kTile) {
                const float* e = A + tid * ldA + 256;
                float tx = e[3], ty = e[4], tz = e[5];
                if (TI.rotate_normal && tid < rows) {                 // n_rot = R n  ->  nt = R^T nt_rot
                    const float* Tq = TI.T + (row0 + tid) * 16;
                    const float ax = Tq[0] * tx + Tq[4] * ty + Tq[8] * tz;
                    const float ay = Tq[1] * tx + Tq[5] * ty + Tq[9] * tz;
                    const float az = Tq[2] * tx + Tq[6] * ty + Tq[10] * tz;
                    tx = ax;
                    ty = ay;
                    tz = az;
                }
                const f32x4 nt = {tx, ty, tz, 0.f};
                reinterpret_cast<f32x4*>(ntl)[tid] = nt;
                reinterpret_cast<f32x4*>(rgbv)[tid] = f32x4{e[0], e[1], e[2], 0.f};   // direct dL/dx, kept for the end
                if (tid < rows) reinterpret_cast<f32x4*>(TO.hd[0])[row0 + tid] = nt;
            }
            __syncthreads();
            }   // !geom
            // =========================================================== 3. tangent pass hd_0 = nt  (B <- hd_k)
            {   // layer 1 (K = 3)
                f32x4 xt[kNT];
#pragma unroll
                for (int nn = 0; nn < kNT; ++nn) xt[nn] = *reinterpret_cast<const f32x4*>(ntl + (nn * 16 + j) * 4);
#pragma unroll
                for (int m = 0; m < kSdfMT; ++m) {
                    const int ch0 = (mt0 + m) * 16 + 4 * g;
                    f32x4 w[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = *reinterpret_cast<const f32x4*>(net.w0 + (ch0 + r) * 4);
                    const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + ch0);
                    const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + ch0);
                    const f32x4 f = *reinterpret_cast<const f32x4*>(net.freq + ch0);
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) {
                        f32x4 vd, sn, cs, hd;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            vd[r] = fmaf(w[r][2], xt[nn][2], fmaf(w[r][1], xt[nn][1], w[r][0] * xt[nn][0]));
                        vdslab[slab_at(0, m, nn)] = vd;
                        film_sincos(vslab[slab_at(0, m, nn)], fw, pw, sn, cs);
#pragma unroll
                        for (int r = 0; r < 4; ++r) hd[r] = 30.0f * f[r] * cs[r] * vd[r];
                        *reinterpret_cast<f32x4*>(B + (nn * 16 + j) * ldB + ch0) = hd;
                    }
                }
            }
            __syncthreads();
#pragma unroll 1
            for (int k = 1; k < 6; ++k) {
                stream_rows(B, ldB, 256, TO.hd[k], row0, rows, tid);
                f32x4 acc[kSdfMT][kNT];
#pragma unroll
                for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[m][nn]);
                gemm_any<B3, 16, kSdfMT, kNT>(net.wp[k - 1], b3.sdf_wp[k - 1], mt0, B, ldB, acc, lane);
                __syncthreads();
#pragma unroll
                for (int m = 0; m < kSdfMT; ++m) {
                    const int ch0 = (mt0 + m) * 16 + 4 * g;
                    const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + k * 256 + ch0);
                    const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                    const f32x4 f = *reinterpret_cast<const f32x4*>(net.freq + k * 256 + ch0);
#pragma unroll
                    for (int nn = 0; nn < kNT; ++nn) {
                        f32x4 sn, cs, hd;
                        vdslab[slab_at(k, m, nn)] = acc[m][nn];
                        film_sincos(vslab[slab_at(k, m, nn)], fw, pw, sn, cs);
#pragma unroll
                        for (int r = 0; r < 4; ++r) hd[r] = 30.0f * f[r] * cs[r] * acc[m][nn][r];
                        *reinterpret_cast<f32x4*>(B + (nn * 16 + j) * ldB + ch0) = hd;
                    }
                }
                __syncthreads();
            }
            stream_rows(B, ldB, 256, TO.hd[6], row0, rows, tid);
            __syncthreads();
            // =========================================================== 4. reverse sweep of both streams
            // A[:, :256] <- adj h_6 = dL/dh_6 + g_s w_7 ;  B <- adj hd_6 = w_7
            for (int e = tid; e < kTile * 64; e += kThreads) {
                const int pt = e >> 6, c4 = e & 63;
                const f32x4 w7 = *reinterpret_cast<const f32x4*>(net.w6 + c4 * 4);
                f32x4 a = *reinterpret_cast<const f32x4*>(A + pt * ldA + c4 * 4);
                const float gs = gin[pt * 4 + 3];
                a += w7 * gs;
                *reinterpret_cast<f32x4*>(A + pt * ldA + c4 * 4) = a;
                *reinterpret_cast<f32x4*>(B + pt * ldB + c4 * 4) = w7;
            }
            __syncthreads();
#pragma unroll 1
            for (int k = 5; k >= 0; --k) {   // layer index k (0-based): v_{k+1}
                float pf[kSdfMT][4], pp[kSdfMT][4];
#pragma unroll
                for (int m = 0; m < kSdfMT; ++m) {
                    const int ch0 = (mt0 + m) * 16 + 4 * g;
                    const f32x4 fw = *reinterpret_cast<const f32x4*>(net.fw + k * 256 + ch0);
                    const f32x4 pw = *reinterpret_cast<const f32x4*>(net.pw + k * 256 + ch0);
                    const f32x4 f = *reinterpret_cast<const f32x4*>(net.freq + k * 256 + ch0);
                    const f32x4 b = *reinterpret_cast<const f32x4*>(net.bias + k * 256 + ch0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) pf[m][r] = pp[m][r] = 0.f;
#pragma unroll 1
                    for (int nn = 0; nn < kNT; ++nn) {
                        const int pt = nn * 16 + j;
                        const f32x4 v = vslab[slab_at(k, m, nn)], vd = vdslab[slab_at(k, m, nn)];
                        f32x4 sn, cs;
                        film_sincos(v, fw, pw, sn, cs);
                        const f32x4 ah = *reinterpret_cast<const f32x4*>(A + pt * ldA + ch0);
                        const f32x4 ahd = *reinterpret_cast<const f32x4*>(B + pt * ldB + ch0);
                        f32x4 av, avd;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f30 = 30.0f * f[r];
                            const float E = cs[r] * ahd[r];
                            const float zb = cs[r] * ah[r] - f30 * sn[r] * vd[r] * ahd[r];
                            av[r] = f30 * zb;
                            avd[r] = f30 * E;
                            if (pt < rows) {
                                pf[m][r] += 30.0f * ((v[r] + b[r]) * zb + vd[r] * E);
                                pp[m][r] += 30.0f * zb;
                            }
                        }
                        *reinterpret_cast<f32x4*>(A + pt * ldA + ch0) = av;      // in place: own elements only
                        *reinterpret_cast<f32x4*>(B + pt * ldB + ch0) = avd;
                    }
                }
                reduce_channels(pf, TO.film_f + k * 256, mt0, lane);
                reduce_channels(pp, TO.film_p + k * 256, mt0, lane);
                __syncthreads();
                stream_rows(A, ldA, 256, TO.av[k], row0, rows, tid);
                stream_rows(B, ldB, 256, TO.avd[k], row0, rows, tid);
                if (k > 0) {   // the two streams one after the other: eight accumulators live, not sixteen
#pragma unroll 1
                    for (int which = 0; which < 2; ++which) {
                        float* buf = which ? B : A;
                        const int ld = which ? ldB : ldA;
                        f32x4 acc[kSdfMT][kNT];
#pragma unroll
                        for (int m = 0; m < kSdfMT; ++m)
#pragma unroll
                            for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[m][nn]);
                        gemm_any<B3, 16, kSdfMT>(net.wpT[k - 1], b3.sdf_wpT[k - 1], mt0, buf, ld, acc, lane);
                        __syncthreads();
#pragma unroll
                        for (int m = 0; m < kSdfMT; ++m) {
                            const int ch0 = (mt0 + m) * 16 + 4 * g;
#pragma unroll
                            for (int nn = 0; nn < kNT; ++nn)
                                *reinterpret_cast<f32x4*>(buf + (nn * 16 + j) * ld + ch0) = acc[m][nn];
                        }
                    }
                    __syncthreads();
                }
            }
            // dL/dx = W_1^T adj v_1 + direct colour input
            {
                const int pt = tid >> 3, part = tid & 7;
                float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll 8
                for (int i = 0; i < 32; ++i) {
                    const int ch = part + 8 * i;
                    const float u = A[pt * ldA + ch];
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
                if (part == 0 && pt < rows)
                    reinterpret_cast<f32x4*>(TO.gx)[row0 + pt] =
                        f32x4{gx + rgbv[pt * 4], gy + rgbv[pt * 4 + 1], gz + rgbv[pt * 4 + 2], 0.f};
            }
        }
        __syncthreads();
    }
}

#undef TI
#undef TO

// ---- VolSDF density + alpha compositing of the training forward, with their backward (IDR:363-394) --------------------
// One thread per ray over its L valid samples, which sit next to each other in the compacted per-sample arrays (offset
// off[r]): s = SDF in metres, c = colour, z = depth.  With ib = 1 / beta:
//   psi = 1/2 + 1/2 sign(-s) (1 - exp(-|s| ib)),  d = relu(ib psi),  alpha = 1 - exp(-d delta),
//   q = 1 - alpha + 1e-7,  T_k = prod_{j<k} q_j,  w = alpha T,  C = sum w c,  A = sum w,  acc = clip(A, 0, 1)
// delta_k = z_{k+1} - z_k; the last valid sample takes 1e10 (render_last_pt) or 1 / n_steps.  The arithmetic follows the
// torch expressions of training.shade_composite_train operation by operation (the sums over k run in order here, as a
// tree there).
__device__ __forceinline__ float comp_delta(const float* z, long long p0, int k, int L, int render_last_pt, float inv_steps) {
    return k + 1 < L ? z[p0 + k + 1] - z[p0 + k] : (render_last_pt ? 1e10f : inv_steps);
}
__device__ __forceinline__ void comp_density(float s, float ib, float& dens, float& dpsi_ds, float& dpsi_dib, float& psi) {
    const float e = expf(-fabsf(s) * ib);
    const float sg = s < 0.f ? 1.0f : (s > 0.f ? -1.0f : 0.f);   // sign(-s)
    psi = 0.5f + 0.5f * sg * (1.0f - e);
    dens = fmaxf(ib * psi, 0.f);
    dpsi_ds = s != 0.f ? -0.5f * ib * e : 0.f;
    dpsi_dib = -0.5f * s * e;   // 1/2 sign(-s) |s| e
}

__global__ void k_composite_train_fwd(int n_rays, const int* __restrict__ len, const long long* __restrict__ off,
                                      const float* __restrict__ sdf, const float* __restrict__ rgb, const float* __restrict__ z,
                                      const float* __restrict__ inv_beta, int render_last_pt, float inv_steps,
                                      float* __restrict__ out_rgb, float* __restrict__ out_acc) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int L = len[r];
    const long long p0 = off[r];
    const float ib = inv_beta[0];
    float cr = 0.f, cg = 0.f, cb = 0.f, A = 0.f, T = 1.0f;
    for (int k = 0; k < L; ++k) {
        float dens, a1, a2, psi;
        comp_density(sdf[p0 + k], ib, dens, a1, a2, psi);
        const float alpha = 1.0f - expf(-dens * comp_delta(z, p0, k, L, render_last_pt, inv_steps));
        const float w = alpha * T;
        cr += rgb[(p0 + k) * 3] * w;
        cg += rgb[(p0 + k) * 3 + 1] * w;
        cb += rgb[(p0 + k) * 3 + 2] * w;
        A += w;
        T *= 1.0f - alpha + 1e-7f;
    }
    out_rgb[(size_t)r * 3] = cr;
    out_rgb[(size_t)r * 3 + 1] = cg;
    out_rgb[(size_t)r * 3 + 2] = cb;
    out_acc[r] = fminf(fmaxf(A, 0.f), 1.f);
}

// g_rgb_map [n_rays][3], g_acc [n_rays]  ->  g_sdf [P], g_rgb [P][3], g_ib += (one atomic per ray)
__global__ void k_composite_train_bwd(int n_rays, const int* __restrict__ len, const long long* __restrict__ off,
                                      const float* __restrict__ sdf, const float* __restrict__ rgb, const float* __restrict__ z,
                                      const float* __restrict__ inv_beta, int render_last_pt, float inv_steps,
                                      const float* __restrict__ g_map, const float* __restrict__ g_acc,
                                      float* __restrict__ g_sdf, float* __restrict__ g_rgb, float* __restrict__ g_ib) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int L = len[r];
    const long long p0 = off[r];
    const float ib = inv_beta[0];
    const float gr = g_map[(size_t)r * 3], gg = g_map[(size_t)r * 3 + 1], gb = g_map[(size_t)r * 3 + 2];
    // forward once more for A (the clip's gate) and the transmittances (kept: dividing them back out loses them once the
    // product has underflowed)
    float Tk[ARAH_MAX_STEPS];
    float A = 0.f, T = 1.0f;
    for (int k = 0; k < L; ++k) {
        float dens, a1, a2, psi;
        comp_density(sdf[p0 + k], ib, dens, a1, a2, psi);
        const float alpha = 1.0f - expf(-dens * comp_delta(z, p0, k, L, render_last_pt, inv_steps));
        Tk[k] = T;
        A += alpha * T;
        T *= 1.0f - alpha + 1e-7f;
    }
    const float gA = (A >= 0.f && A <= 1.f) ? g_acc[r] : 0.f;
    // reverse sweep: suffix = sum_{j>k} u_j w_j
    float suffix = 0.f, gib = 0.f;
    for (int k = L - 1; k >= 0; --k) {
        float dens, dpsi_ds, dpsi_dib, psi;
        const float s = sdf[p0 + k];
        comp_density(s, ib, dens, dpsi_ds, dpsi_dib, psi);
        const float delta = comp_delta(z, p0, k, L, render_last_pt, inv_steps);
        const float ex = expf(-dens * delta);
        const float alpha = 1.0f - ex, q = 1.0f - alpha + 1e-7f;
        T = Tk[k];
        const float w = alpha * T;
        const float c0 = rgb[(p0 + k) * 3], c1 = rgb[(p0 + k) * 3 + 1], c2 = rgb[(p0 + k) * 3 + 2];
        g_rgb[(p0 + k) * 3] = gr * w;
        g_rgb[(p0 + k) * 3 + 1] = gg * w;
        g_rgb[(p0 + k) * 3 + 2] = gb * w;
        const float u = gr * c0 + gg * c1 + gb * c2 + gA;
        const float g_alpha = u * T - suffix / q;
        suffix += u * w;
        const float g_dens = (ib * psi > 0.f) ? g_alpha * delta * ex : 0.f;
        g_sdf[p0 + k] = g_dens * ib * dpsi_ds;
        gib += g_dens * (psi + ib * dpsi_dib);
    }
    if (L > 0) atomicAdd(g_ib, gib);
}

template <bool IDR>
constexpr size_t lds_shade_train() {
    return (64 * 4 * 6) * 4 + (size_t)64 * ColDims<IDR>::kLdA * 4 + (size_t)64 * kSdfLd * 4;
}

}  // namespace
