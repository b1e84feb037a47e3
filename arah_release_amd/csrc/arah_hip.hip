// arah_hip.hip -- kernels and C ABI (include/arah_hip.h) of the MI355X-native ARAH hot path.
//
// Structure: the four hot loops of the reference (A sphere tracing, B joint root find,
// C per-sample canonicalisation, D shading + compositing) are sequences of a few kernels that
// work on *compacted active lists* kept on the device: every MLP evaluation is a dense 64-column
// MFMA tile (mlp.hpp), points retire individually (per-lane Broyden state in HBM, ~200 B/point),
// and the host enqueues a fixed number of iterations without ever synchronising -- a launch whose
// list is empty costs a few microseconds.  Intermediates are laid out densely ([N], [N,S]) in the
// caller's workspace: at ~10^5 flop per byte this path is MFMA-bound, HBM traffic is noise.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC arah_hip.hip -o libarah_hip.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "../../include/arah_hip.h"
#include "pointwise.hpp"
#include "mlp.hpp"

using namespace arah;

namespace {

constexpr float kRootThresh = 1e-5f;   // RT:18, models/__init__.py:75
constexpr int kSphereIters = 50;       // RT:19
constexpr float kSurfaceRange = 0.05f; // RT:23
constexpr float kClampDist = 0.1f;     // RT:174
constexpr int kBroydenSteps = 50;      // broyden.py:4
constexpr float kDvg = 1.0f;
constexpr int kMaxVerts = 6912;
constexpr int kKnnThreads = 1024;      // 16 waves share the 119 KB vertex table: the walk is latency-bound
constexpr int kMaxGrid = 512;          // persistent grid cap for the MFMA kernels: 256 CUs x (at most) 2 resident workgroups

// ------------------------------------------------------------------------------------------
// device-side frame view
// ------------------------------------------------------------------------------------------
struct KnnPtrs {
    const float* sorted4;
    const float* spheres;
    const void* grid;
    const unsigned char* cells;
};

struct FrameDev {
    SdfNet sdf;
    SkinNet skin;
    SkinWave skw;
    ColNet col;
    KnnPtrs knn;
    const float* verts4;
    const float* verts_raw;   // caller's [n_verts][3]
    const float* vert_T;      // [n_verts][16] blended bone transform of each vertex (k_vertex_transforms)
    const float* bones;
    const float* scal;   // [9] device: trans(3), center(3), coord_min, coord_max, |variance| -- the host never sees them
    int n_verts;
    int split;   // 1: forward SDF trunks run on the f16-split engine (ARAH_PRECISION_SPLIT_F16)
};

// The per-frame scalars of a kernel: eight scalar loads from a wave-uniform address.  (Kept OUT of the by-value
// FrameDev argument on purpose: writing into a kernel argument makes the compiler copy the whole struct to scratch,
// and the dynamically indexed weight pointers then come from there instead of the kernarg segment.)
__device__ __forceinline__ BodyConst load_bc(const FrameDev& fr) {
    const float* s = fr.scal;
    BodyConst bc;
    bc.trans[0] = s[0];
    bc.trans[1] = s[1];
    bc.trans[2] = s[2];
    bc.center[0] = s[3];
    bc.center[1] = s[4];
    bc.center[2] = s[5];
    bc.cmin = s[6];
    bc.cmax = s[7];
    return bc;
}
__device__ __forceinline__ float load_beta(const FrameDev& fr) { return fr.scal[8]; }

FrameDev to_dev(const ArahFrame& f) {
    FrameDev d;
    d.sdf.w0 = f.sdf_w0;
    for (int i = 0; i < 5; ++i) {
        d.sdf.wp[i] = f.sdf_wp[i];
        d.sdf.wpT[i] = f.sdf_wpT[i];
    }
    d.sdf.w6 = f.sdf_w6;
    d.sdf.bias = f.sdf_bias;
    d.sdf.freq = f.sdf_freq;
    d.sdf.phase = f.sdf_phase;
    d.sdf.b6 = f.sdf_b6;   // device pointer
    d.sdf.fw = f.sdf_fw;
    d.sdf.pw = f.sdf_pw;
    d.sdf.fws = f.sdf_fws;
    for (int i = 0; i < 5; ++i) d.sdf.wps[i] = reinterpret_cast<const f16x8*>(f.sdf_wps[i]);
    d.split = f.precision == ARAH_PRECISION_SPLIT_F16 ? 1 : 0;
    d.skin.w0 = f.skin_w0;
    for (int i = 0; i < 3; ++i) d.skin.wp[i] = f.skin_wp[i];
    d.skin.w4p = f.skin_w4p;
    d.skin.bias = f.skin_bias;
    for (int i = 0; i < 4; ++i) d.skin.wps[i] = reinterpret_cast<const f16x8*>(f.skin_wps[i]);
    d.skin.scales = f.skin_scales;
    d.skw.wpr = reinterpret_cast<const f16x8*>(f.skin_wpr);
    d.skw.consts = f.skin_wconsts;
    d.col.w0p = f.col_w0p;
    d.col.w1p = f.col_w1p;
    d.col.w2p = f.col_w2p;
    d.col.w3ap = f.col_w3ap;
    d.col.w3bp = f.col_w3bp;
    d.col.w4p = f.col_w4p;
    d.col.w5 = f.col_w5;
    d.col.bias = f.col_bias;
    d.verts4 = f.verts4;
    d.verts_raw = f.verts;
    d.knn.sorted4 = f.verts4;
    d.knn.spheres = f.knn_spheres;
    d.knn.grid = f.knn_grid;
    d.knn.cells = reinterpret_cast<const unsigned char*>(f.knn_cells);
    d.vert_T = f.vert_T;
    d.bones = f.bones;
    d.scal = f.scalars;
    d.n_verts = f.n_verts;
    return d;
}

struct TierStatsRaw {   // ArahCounters.n_tier_*: rays classified, of them surface rays / promoted / skipped; samples sent to phase 1,
                        // to phase 2, never evaluated; rays that sent a witness
    unsigned long long rays, rays_surface, rays_promoted, rays_skipped, samples_p1, samples_p2, samples_skipped, witnesses;
    unsigned long long rays_untraced;   // rays whose segment misses the posed fat body: loops A and B not run
};
struct Counters {
    unsigned long long n_sdf_fwd, n_sdf_grad, n_skin_fwd, n_skin_jac, n_col, n_knn, n_density, n_canon, n_split_nonfinite;
    // tiered eval forward (tier.hpp): ray / sample bookkeeping, and the share of n_canon / n_density that phase 2 ran
    TierStatsRaw tier;
    unsigned long long n_canon_p2, n_density_p2;
    unsigned long long tier_snap[2];
    unsigned long long clk[8 * 16];   // instrumented builds (-DARAH_CLOCKS): s_memtime ticks per wave slot and phase
    unsigned long long clk_shade[8 * 16];   // the same for k_shade
};
static_assert(offsetof(Counters, tier_snap) == sizeof(ArahCounters), "the head of Counters is the ABI's ArahCounters");
#ifdef ARAH_CLOCKS
typedef PhaseClk KernelClk;
#else
typedef NoClk KernelClk;
#endif

__device__ __forceinline__ void count_add(unsigned long long* c, int n) {
    if (c && n > 0) atomicAdd(c, (unsigned long long)n);
}

// ------------------------------------------------------------------------------------------
// weight packing (once per frame for the emitted SDF MLP, once per weight update otherwise)
// ------------------------------------------------------------------------------------------
struct ColSegs {
    int n;
    int dst0[4], src0[4], len[4];
};

// dst: packed [M_pad/16][KC][64][4]; element (row, col) = transpose ? src[col_src*ld + row] : src[row*ld + col_src]
// with col_src = segment map of col.  transpose == 2 (transposed operand whose OUTPUT rows are the permuted ones):
// element (row, col) = src[col*ld + row_src], row_src = segment map of row, col < ncol2.
__global__ void k_pack(float* __restrict__ dst, const float* __restrict__ src, int M, int ld, int m_tiles, int KC,
                       ColSegs segs, int transpose, int ncol2) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = m_tiles * KC * 64;
    if (idx >= total) return;
    const int lane = idx & 63;
    const int tile = idx >> 6;
    const int kc = tile % KC, mt = tile / KC;
    const int row = mt * 16 + (lane & 15);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (transpose == 2) {
        int sr = -1;
        for (int s = 0; s < segs.n; ++s)
            if (row >= segs.dst0[s] && row < segs.dst0[s] + segs.len[s]) sr = segs.src0[s] + (row - segs.dst0[s]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = kc * 16 + 4 * (lane >> 4) + t;
            if (sr >= 0 && col < ncol2) v[t] = src[(size_t)col * ld + sr];
        }
        reinterpret_cast<f32x4*>(dst)[idx] = v;
        return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = kc * 16 + 4 * (lane >> 4) + t;
        int sc = -1;
        for (int s = 0; s < segs.n; ++s)
            if (col >= segs.dst0[s] && col < segs.dst0[s] + segs.len[s]) sc = segs.src0[s] + (col - segs.dst0[s]);
        if (sc >= 0 && row < M) v[t] = transpose ? src[(size_t)sc * ld + row] : src[(size_t)row * ld + sc];
    }
    reinterpret_cast<f32x4*>(dst)[idx] = v;
}

// max |src| as float bits (non-negative floats order like unsigned ints); *amax must start at 0
__global__ void k_absmax(const float* __restrict__ src, int n, unsigned* amax) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float a = fabsf(src[i]);
        if (a == a) m = fmaxf(m, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(amax, __float_as_uint(m));
}

// power-of-two weight scale that puts the largest |w| of the layer in [2^13, 2^14): hi never overflows f16
// (max 65504), and lo parts flushed as f16 subnormals cost < 2^-28 of the largest weight
__device__ __forceinline__ float split_weight_scale(unsigned amax_bits) {
    const float a = __uint_as_float(amax_bits);
    if (!(a > 0.f) || a > 3.0e38f) return 1.0f;
    int e;
    frexpf(a, &e);   // a = m 2^e, m in [0.5, 1)
    return ldexpf(1.0f, 14 - e);
}

// split-engine A operand of one 256x256 layer: dst[((mt*8 + kc)*2 + s)*64 + lane] holds 8 halves (s = 0 hi, 1 lo)
// of scale * src[mt*16 + (lane&15)][kc*32 + 8 (lane>>4) .. +7]
// PERM (the point-owning-wave kernels, SkinWave in mlp.hpp): element e of lane group g in chunk kc is the channel an
// accumulator of the previous layer holds there, (2 kc + (e >> 2)) * 16 + 4 g + (e & 3).
template <bool PERM = false>
__global__ void k_pack_split(f16x8* __restrict__ dst, const float* __restrict__ src, int M, int ld, int m_tiles,
                             int KC32, const unsigned* amax) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m_tiles * KC32 * 64) return;
    const float scale = PERM ? 1.0f : split_weight_scale(*amax);
    const int lane = idx & 63, tile = idx >> 6, kc = tile % KC32, mt = tile / KC32;
    const int row = mt * 16 + (lane & 15);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = PERM ? (2 * kc + (e >> 2)) * 16 + 4 * (lane >> 4) + (e & 3) : kc * 32 + (lane >> 4) * 8 + e;
        const float w = row < M ? src[(size_t)row * ld + col] * scale : 0.f;
        const _Float16 h = (_Float16)w;
        hi[e] = h;
        lo[e] = (_Float16)(w - (float)h);
    }
    dst[(size_t)(tile * 2 + 0) * 64 + lane] = hi;
    dst[(size_t)(tile * 2 + 1) * 64 + lane] = lo;
}

// FiLM constants in half-revolutions: z = 30 (f (v + b) + phi) = pi (fw v + pw); fws = fw / (weight scale * act scale)
// of the split layer that produces v (layer 0 runs on the vector ALU: fws = fw)
__global__ void k_fold_film(const float* __restrict__ freq, const float* __restrict__ phase, const float* __restrict__ bias,
                            const unsigned* amax, float* fw, float* pw, float* fws) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * 256) return;
    const int k = i >> 8;
    const double c = kFilmScale;   // revolutions (hardware sine) or half-revolutions (-DARAH_POLY_SINE), see mlp.hpp
    const double f = freq[i];
    fw[i] = (float)(f * c);
    pw[i] = (float)((f * (double)bias[i] + (double)phase[i]) * c);
    const float inv = k == 0 ? 1.0f : 1.0f / (split_weight_scale(amax[k - 1]) * kActScale);
    fws[i] = (float)(f * c) * inv;
    if ((i & 255) == 0) fws[6 * 256 + k] = inv;   // accumulator -> pre-activation of layer k (training taps)
}

// Largest hidden activation of the skinning MLP (Softplus: h >= 0) per layer over a 9^3 lattice of the normalised
// cube [-1.5, 1.5]^3 -- query points are normalised to [-1, 1] over the padded body box (RFU:37-51), root-finding
// iterates stray a little beyond.  One thread per lattice point, plain fp32 loops over the raw row-major weights.
struct SkinRaw {
    const float* w[4];   // [128,3], 3 x [128,128]
    const float* b[4];
};
__global__ __launch_bounds__(128) void k_skin_probe(SkinRaw net, unsigned* amax) {
    __shared__ float h[2][128];
    const int pt = blockIdx.x, c = threadIdx.x;
    const float x = -1.5f + 3.0f * (float)(pt % 9) / 8.0f, y = -1.5f + 3.0f * (float)((pt / 9) % 9) / 8.0f,
                z = -1.5f + 3.0f * (float)(pt / 81) / 8.0f;
    float v = softplus100(net.w[0][c * 3] * x + net.w[0][c * 3 + 1] * y + net.w[0][c * 3 + 2] * z + net.b[0][c]);
    h[0][c] = v;
    float m = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((c & 63) == 0) atomicMax(amax + 0, __float_as_uint(m));
    __syncthreads();
    for (int k = 1; k < 4; ++k) {
        const float* hin = h[(k - 1) & 1];
        float s = net.b[k][c];
        for (int i = 0; i < 128; ++i) s += net.w[k][c * 128 + i] * hin[i];
        v = softplus100(s);
        h[k & 1][c] = v;
        m = v == v ? v : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((c & 63) == 0) atomicMax(amax + k, __float_as_uint(m));
        __syncthreads();
    }
}

// scales[0..3]: S_k = power of two with S_k * probed max in [2^10, 2^11) (32x headroom below the f16 limit);
// scales[4..7]: 1 / (weight scale of layer k+1 * S_k).  amax[0..3] probed activations, amax[4..7] |W| of layers 1..4.
__global__ void k_skin_scales(const unsigned* amax, float* scales) {
    const int k = threadIdx.x;
    if (k >= 4) return;
    const float a = __uint_as_float(amax[k]);
    float S = 1.0f;
    if (a > 0.f && a < 3.0e38f) {
        int e;
        frexpf(a, &e);
        S = ldexpf(1.0f, 11 - e);
    }
    scales[k] = S;
    scales[4 + k] = 1.0f / (split_weight_scale(amax[4 + k]) * S);
}

// constants of the point-owning-wave kernels (SkinWave::consts, layout kCw* in mlp.hpp, where the units are explained);
// amax[0..3]: probed activation maxima (k_skin_probe, x units).  One thread per output row.
__global__ void k_skin_wave_consts(SkinRaw net, const float* __restrict__ w4, const float* __restrict__ w4b, const unsigned* amax,
                                   float* __restrict__ c) {
    const int i = threadIdx.x;
    // activation scale of layer k's output (z units): 1 for an ordinary network, smaller when the probed maximum times 32
    // would leave the f16 range
    auto act_scale = [&](int k) {
        const float a = __uint_as_float(amax[k]) * kZUnit;
        if (!(a > 0.f) || a > 3.0e38f) return 1.0f;
        int e;
        frexpf(a, &e);   // a = m 2^e, m in [0.5, 1)
        return e > 11 ? ldexpf(1.0f, 11 - e) : 1.0f;
    };
    const double M = (double)kCwShift;
    if (i < 128) {
        unsigned* ct = reinterpret_cast<unsigned*>(c + kCwW0T) + i * 8;
        unsigned short h[3], l[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float w = net.w[0][i * 3 + k] * kZUnit;
            const _Float16 hh = (_Float16)w, ll = (_Float16)(w - (float)hh);
            h[k] = __builtin_bit_cast(unsigned short, hh);
            l[k] = __builtin_bit_cast(unsigned short, ll);
        }
        ct[0] = ct[2] = (unsigned)h[0] | ((unsigned)h[1] << 16);
        ct[1] = ct[3] = (unsigned)h[2];
        ct[4] = ct[6] = (unsigned)l[0] | ((unsigned)l[1] << 16);
        ct[5] = ct[7] = (unsigned)l[2];
        c[kCwBinit + i] = (float)((double)net.b[0][i] * (double)kZUnit - M);
#pragma unroll 1
        for (int k = 1; k < 4; ++k) {
            double rs = 0.0;
            for (int col = 0; col < 128; ++col) rs += (double)net.w[k][i * 128 + col];
            c[kCwBinit + k * 128 + i] = (float)(((double)net.b[k][i] * (double)kZUnit + M * rs - M) * (double)act_scale(k - 1));
        }
    }
    if (i < 32) {
        double v = 0.0;
        if (i < 25) {
            double rs = 0.0;
            for (int col = 0; col < 128; ++col) rs += (double)w4[i * 128 + col];
            v = ((double)w4b[i] * (100.0 / 0.6931471805599453) + M * rs) * (double)act_scale(3);
        }
        c[kCwBinit + 512 + i] = (float)v;
    }
    if (i < 4) {
        c[kCwActS + i] = act_scale(i);
        if (i == 0) c[kCwScaled] = (act_scale(0) != 1.0f || act_scale(1) != 1.0f || act_scale(2) != 1.0f || act_scale(3) != 1.0f) ? 1.0f : 0.0f;
        c[kCwInv + i] = i < 3 ? 1.0f / act_scale(i) : 0.2f / act_scale(3);   // [3]: 20 log2(e) ln(2) / 100 = 0.2
    }
}

// the per-frame scalars stay on the device: trans(3), center(3), coord_min, coord_max, |variance| -> scal[9]
__global__ void k_gather_scalars(float* __restrict__ dst, const float* trans, const float* center, const float* cmin,
                                 const float* cmax, const float* beta) {
    const int i = threadIdx.x;
    if (i < 3) dst[i] = trans[i];
    else if (i < 6) dst[i] = center[i - 3];
    else if (i == 6) dst[i] = cmin[0];
    else if (i == 7) dst[i] = cmax[0];
    else if (i == 8) dst[i] = beta ? beta[0] : 1e-3f;
}

// vert_T[v] = sum_j w_vj A_j, once per frame: every nearest-vertex query (loops A, B and the seeds of loop C) used to blend
// the 24 bone transforms of its vertex again -- 6890 blends instead of ~1e7
__global__ __launch_bounds__(64) void k_vertex_transforms(const float* __restrict__ weights, const float* __restrict__ bones,
                                                          int n_verts, float* __restrict__ vert_T) {
    __shared__ __attribute__((aligned(16))) float sb[24 * 16];
    for (int i = threadIdx.x; i < 24 * 16; i += 64) sb[i] = bones[i];
    __syncthreads();
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v >= n_verts) return;
    float T[16];
    blend(weights + (size_t)v * 24, sb, T);
#pragma unroll
    for (int c = 0; c < 4; ++c)
        reinterpret_cast<f32x4*>(vert_T + (size_t)v * 16)[c] = f32x4{T[c * 4], T[c * 4 + 1], T[c * 4 + 2], T[c * 4 + 3]};
}

// dst[r][0..3] = {src[r][0..ncol-1], 0...}
__global__ void k_pad_rows4(float* __restrict__ dst, const float* __restrict__ src, int rows, int rows_pad, int ncol,
                            float fill) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows_pad) return;
    f32x4 v = {fill, fill, fill, 0.f};
    if (r < rows) {
        for (int c = 0; c < 4; ++c) v[c] = c < ncol ? src[(size_t)r * ncol + c] : 0.f;
    }
    reinterpret_cast<f32x4*>(dst)[r] = v;
}

__global__ void k_copy(float* __restrict__ dst, const float* __restrict__ src, int n, int n_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pad) dst[i] = i < n ? src[i] : 0.f;
}

// dst[r] = b[r] + sum_c W[r][col0 + c] * vec[c]    (folds the per-frame constant colour inputs)
__global__ void k_fold_bias(float* __restrict__ dst, const float* __restrict__ b, const float* __restrict__ W, int ld,
                            int col0, const float* __restrict__ vec, int n_vec, int rows) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float s = b[r];
    for (int c = 0; c < n_vec; ++c) s += W[(size_t)r * ld + col0 + c] * vec[c];
    dst[r] = s;
}

// ------------------------------------------------------------------------------------------
// nearest SMPL vertex + inverse LBS with its weights (pytorch3d knn_points K=1, RT:382-400,403-422)
//
// Exact 1-NN, accelerated per frame (arah_prepare_frame):
//   1. k_sort_verts   : vertices sorted along a 30-bit Morton curve (bitonic sort in LDS, one workgroup);
//                       consecutive runs of 32 sorted vertices form a CLUSTER; writes the grid geometry.
//   2. k_cluster_spheres: bounding sphere of every cluster.
//   3. k_cell_clusters: uniform grid over the body's box (+ margin); per cell the (distance-ordered) list of
//                       clusters that can contain the nearest vertex of SOME point of the cell:
//                       mindist(cell, sphere_c) <= min_c' (maxdist(cell, centre_c') + r_c').
// A query looks up its cell, walks the <= 63 candidate clusters with a running-best sphere test and
// scans 32 vertices per surviving cluster: ~10^2 distance evaluations instead of 6890.  Points outside
// the grid (or cells whose list overflowed) walk all clusters with the same pruning.  Ties are resolved
// towards the lower original vertex index, like a first-match linear scan.
// ------------------------------------------------------------------------------------------
enum { SRC_POINTS = 0, SRC_RAYS = 1, SRC_SAMPLES = 2 };

constexpr int kClusterSize = 28;   // slots per cluster (k-d leaves hold 26-27 vertices for n <= 6912)
constexpr int kClusterLds = 29;    // LDS stride in vertices: 464 B = 116 dwords, so clusters start on different banks
constexpr int kMaxClusters = 256;
constexpr int kMaxCells = 65536;
constexpr int kCellBytes = 64;                           // [count | 63 cluster ids]
constexpr float kGridMargin = 0.08f;

struct GridInfo {
    float origin[3];
    float h, inv_h;
    int dims[3];
    int n_cells;
    int n_clusters;
    int pad[2];
};

struct RaySet {
    const float* cam_loc;   // [n_cams][3]
    const float* dirs;      // [N][3]
    int rays_per_cam;
};

__device__ __forceinline__ V3 ray_point(const RaySet& rs, int ray, float t) {
    const int cam = ray / rs.rays_per_cam;
    V3 p;
    p.x = rs.dirs[ray * 3 + 0] * t + rs.cam_loc[cam * 3 + 0];
    p.y = rs.dirs[ray * 3 + 1] * t + rs.cam_loc[cam * 3 + 1];
    p.z = rs.dirs[ray * 3 + 2] * t + rs.cam_loc[cam * 3 + 2];
    return p;
}

// bitonic sort of 8192 (key, index) pairs in LDS by one workgroup of 1024 threads; total order on
// (key, index) keeps it deterministic.  Compare-exchanges with stride j <= 256 stay inside an aligned block of 512
// elements: each of the 16 waves owns one block and runs those sub-stages back to back with wave-level ordering only
// (the LDS operations of one wave complete in order), so a sort costs 15 workgroup barriers instead of 91 -- the
// same network, the same result.
__device__ __forceinline__ void bitonic_cas(unsigned* keys, unsigned short* idxs, int i, int j, int k) {
    const int l = i | j;
    const bool up = (i & k) == 0;
    const unsigned ka = keys[i], kb = keys[l];
    const unsigned short ia = idxs[i], ib = idxs[l];
    const bool gt = ka > kb || (ka == kb && ia > ib);
    if (gt == up) {
        keys[i] = kb;
        keys[l] = ka;
        idxs[i] = ib;
        idxs[l] = ia;
    }
}
__device__ __forceinline__ void bitonic_sort_8192(unsigned* keys, unsigned short* idxs, int tid) {
    const int wave = tid >> 6, lane = tid & 63;
    for (int k = 2; k <= 8192; k <<= 1) {
        int j = k >> 1;
        for (; j >= 512; j >>= 1) {
            for (int t = tid; t < 4096; t += 1024) bitonic_cas(keys, idxs, ((t & ~(j - 1)) << 1) | (t & (j - 1)), j, k);
            __syncthreads();
        }
        for (; j > 0; j >>= 1) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = lane + 64 * u;   // pair number inside the wave's block of 512
                bitonic_cas(keys, idxs, wave * 512 + (((t & ~(j - 1)) << 1) | (t & (j - 1))), j, k);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int float_ordered(float f) {
    const int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float ordered_float(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// One workgroup builds a balanced k-d partition of the vertices: 8 levels of "sort every segment along the
// longest axis of its own bounding box, cut it at the median" (a segmented bitonic sort per level) give
// 256 leaves of 26-27 spatially compact vertices = the clusters.  Output: sorted4[cluster][28] float4
// (x, y, z, original index), unused slots far away.  Also writes the grid geometry used by
// k_cell_clusters and the queries.
__global__ __launch_bounds__(1024) void k_sort_verts(const float* __restrict__ verts, int n, float* __restrict__ sorted4,
                                                      GridInfo* __restrict__ grid) {
    __shared__ unsigned keys[8192];
    __shared__ unsigned short idxs[8192];
    __shared__ unsigned char segid[8192];
    int (*sbox)[6] = reinterpret_cast<int(*)[6]>(keys);   // per-segment boxes live in `keys` between two sorts
    __shared__ int sstart[2][257];
    __shared__ unsigned char saxis[256];
    __shared__ float bb[6];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 1024) {
        idxs[i] = (unsigned short)i;
        segid[i] = 0;
    }
    if (tid == 0) {
        sstart[0][0] = 0;
        sstart[0][1] = n;
    }
    __syncthreads();
    int cur = 0;
    for (int level = 0; level < 8; ++level) {
        const int nseg = 1 << level;
        // bounding box of every segment
        for (int i = tid; i < nseg * 6; i += 1024) sbox[i / 6][i % 6] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const int o = idxs[i], sg = segid[i];
            for (int c = 0; c < 3; ++c) {
                const int v = float_ordered(verts[o * 3 + c]);
                atomicMin(&sbox[sg][c], v);
                atomicMax(&sbox[sg][3 + c], v);
            }
        }
        __syncthreads();
        if (level == 0 && tid < 6) bb[tid] = ordered_float(sbox[0][tid]);
        if (tid < nseg) {
            float e[3];
            for (int c = 0; c < 3; ++c) e[c] = ordered_float(sbox[tid][3 + c]) - ordered_float(sbox[tid][c]);
            saxis[tid] = (unsigned char)(e[0] >= e[1] ? (e[0] >= e[2] ? 0 : 2) : (e[1] >= e[2] ? 1 : 2));
        }
        __syncthreads();
        for (int i = tid; i < 8192; i += 1024) {
            unsigned key = 0xffffffffu;
            if (i < n) {
                const int o = idxs[i], sg = segid[i], a = saxis[sg];
                const float lo = bb[a], ext = fmaxf(bb[3 + a] - bb[a], 1e-9f);
                const unsigned q = (unsigned)fminf(fmaxf((verts[o * 3 + a] - lo) / ext * 65535.0f, 0.f), 65535.f);
                key = ((unsigned)sg << 16) | q;
            }
            keys[i] = key;
        }
        __syncthreads();
        bitonic_sort_8192(keys, idxs, tid);
        // cut every segment at its median
        if (tid < nseg) {
            const int a = sstart[cur][tid], b = sstart[cur][tid + 1];
            sstart[cur ^ 1][2 * tid] = a;
            sstart[cur ^ 1][2 * tid + 1] = a + (b - a + 1) / 2;
            if (tid == nseg - 1) sstart[cur ^ 1][2 * nseg] = b;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 1024) {
            const int sg = segid[i];
            segid[i] = (unsigned char)(2 * sg + (i >= sstart[cur ^ 1][2 * sg + 1] ? 1 : 0));
        }
        cur ^= 1;
        __syncthreads();
    }
    if (tid == 0) {
        GridInfo g;
        float ext[3];
        for (int c = 0; c < 3; ++c) {
            g.origin[c] = bb[c] - kGridMargin;
            ext[c] = (bb[3 + c] + kGridMargin) - g.origin[c];
        }
        float h = fmaxf(cbrtf(ext[0] * ext[1] * ext[2] / 48000.0f), 0.02f);
        for (;;) {
            long long cells = 1;
            for (int c = 0; c < 3; ++c) {
                g.dims[c] = (int)ceilf(ext[c] / h);
                if (g.dims[c] < 1) g.dims[c] = 1;
                cells *= g.dims[c];
            }
            if (cells <= kMaxCells) {
                g.n_cells = (int)cells;
                break;
            }
            h *= 1.1f;
        }
        g.h = h;
        g.inv_h = 1.0f / h;
        g.n_clusters = kMaxClusters;
        g.pad[0] = g.pad[1] = 0;
        *grid = g;
    }
    for (int i = tid; i < kMaxClusters * kClusterSize; i += 1024)
        reinterpret_cast<f32x4*>(sorted4)[i] = f32x4{1e18f, 1e18f, 1e18f, __int_as_float(0x7fffffff)};
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const int sg = segid[i], o = idxs[i];
        reinterpret_cast<f32x4*>(sorted4)[sg * kClusterSize + (i - sstart[cur][sg])] =
            f32x4{verts[o * 3], verts[o * 3 + 1], verts[o * 3 + 2], __int_as_float(o)};
    }
}

__global__ void k_cluster_spheres(const float* __restrict__ sorted4, float* __restrict__ spheres) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= kMaxClusters) return;
    const f32x4* v = reinterpret_cast<const f32x4*>(sorted4) + c * kClusterSize;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    int cnt = 0;
    for (int u = 0; u < kClusterSize; ++u) {
        const f32x4 q = v[u];
        if (__float_as_int(q[3]) == 0x7fffffff) continue;   // unused slot
        ++cnt;
        for (int k = 0; k < 3; ++k) {
            mn[k] = fminf(mn[k], q[k]);
            mx[k] = fmaxf(mx[k], q[k]);
        }
    }
    f32x4 s = {1e18f, 1e18f, 1e18f, 0.f};   // empty cluster: never a candidate
    if (cnt > 0) {
        const float cx = 0.5f * (mn[0] + mx[0]), cy = 0.5f * (mn[1] + mx[1]), cz = 0.5f * (mn[2] + mx[2]);
        float r2 = 0.f;
        for (int u = 0; u < kClusterSize; ++u) {
            const f32x4 q = v[u];
            if (__float_as_int(q[3]) == 0x7fffffff) continue;
            const float dx = q[0] - cx, dy = q[1] - cy, dz = q[2] - cz;
            r2 = fmaxf(r2, dx * dx + dy * dy + dz * dz);
        }
        s = f32x4{cx, cy, cz, sqrtf(r2) * 1.00001f + 1e-7f};   // outward rounding: the sphere must contain its vertices
    }
    reinterpret_cast<f32x4*>(spheres)[c] = s;
}

// thread per cell: upper bound U of the NN distance over the cell, then the clusters with lower bound <= U,
// ordered by lower bound (insertion sort in LDS, <= 63 entries per thread)
constexpr int kCellThreads = 128;
__global__ __launch_bounds__(kCellThreads) void k_cell_clusters(const GridInfo* __restrict__ grid,
                                                                 const float* __restrict__ spheres,
                                                                 unsigned char* __restrict__ cells) {
    __shared__ f32x4 sph[kMaxClusters];
    __shared__ float lbs[63][kCellThreads];            // [slot][thread]: conflict-free
    __shared__ unsigned char ids[64][kCellThreads];
    const GridInfo g = *grid;
    const int t = threadIdx.x;
    for (int i = t; i < kMaxClusters; i += kCellThreads) sph[i] = reinterpret_cast<const f32x4*>(spheres)[i];
    __syncthreads();
    for (int cell = blockIdx.x * kCellThreads + t; cell < g.n_cells; cell += gridDim.x * kCellThreads) {
        const int cx = cell % g.dims[0], cy = (cell / g.dims[0]) % g.dims[1], cz = cell / (g.dims[0] * g.dims[1]);
        const float lo[3] = {g.origin[0] + cx * g.h, g.origin[1] + cy * g.h, g.origin[2] + cz * g.h};
        const float hi[3] = {lo[0] + g.h, lo[1] + g.h, lo[2] + g.h};
        float U = 3.4e38f;
#pragma unroll 4
        for (int c = 0; c < kMaxClusters; ++c) {
            const f32x4 s = sph[c];
            float far2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = fmaxf(fabsf(s[k] - lo[k]), fabsf(s[k] - hi[k]));
                far2 += d * d;
            }
            U = fminf(U, __builtin_amdgcn_sqrtf(far2) + s[3]);   // 1 ulp: the slack below covers it
        }
        U = U * 1.0001f + 1e-6f;
        int cnt = 0;
        bool overflow = false;
#pragma unroll 1
        for (int c = 0; c < kMaxClusters; ++c) {
            const f32x4 s = sph[c];
            float near2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float d = fmaxf(fmaxf(lo[k] - s[k], s[k] - hi[k]), 0.f);
                near2 += d * d;
            }
            const float lb = fmaxf(__builtin_amdgcn_sqrtf(near2) - s[3], 0.f);
            if (lb <= U) {
                if (cnt == 63) {
                    overflow = true;
                    break;
                }
                int pos = cnt;
                while (pos > 0 && lbs[pos - 1][t] > lb) {
                    lbs[pos][t] = lbs[pos - 1][t];
                    ids[pos][t] = ids[pos - 1][t];
                    --pos;
                }
                lbs[pos][t] = lb;
                ids[pos][t] = (unsigned char)c;
                ++cnt;
            }
        }
        unsigned char* dst = cells + (size_t)cell * kCellBytes;
        unsigned w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned v = 0;
#pragma unroll
            for (int bsel = 0; bsel < 4; ++bsel) {
                const int k = q * 4 + bsel;   // byte k of the record: 0 = count, 1.. = ids
                unsigned byte = 0;
                if (k == 0) byte = overflow ? 255u : (unsigned)cnt;
                else if (k - 1 < cnt) byte = ids[k - 1][t];
                v |= byte << (8 * bsel);
            }
            w[q] = v;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            reinterpret_cast<uint4*>(dst)[q] = uint4{w[q * 4], w[q * 4 + 1], w[q * 4 + 2], w[q * 4 + 3]};
    }
}

struct KnnData {
    const float* sorted4;          // [kMaxVerts][4]  (x, y, z, original index)
    const float* spheres;          // [kMaxClusters][4]
    const GridInfo* grid;
    const unsigned char* cells;    // [n_cells][64]
};

// scan one cluster's vertices; lowest original index wins exact ties.  (distance, index) travel as ONE 64-bit key -- the
// bits of a non-negative float order like the float, so "d2 < best || (d2 == best && o < bi)" is a single unsigned 64-bit
// compare and two selects instead of three compares, mask arithmetic and two selects: the kernels that walk clusters lane
// by lane are vector-ALU bound on exactly this loop.
template <int STRIDE>
__device__ __forceinline__ void scan_cluster(const float* sv, int c, V3 p, float& best, int& bi) {
    const f32x4* v = reinterpret_cast<const f32x4*>(sv) + c * STRIDE;
    unsigned long long key = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi;
#pragma unroll 7
    for (int u = 0; u < kClusterSize; ++u) {
        const f32x4 q = v[u];
        const float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(q[3]);
        key = k < key ? k : key;
    }
    best = __uint_as_float((unsigned)(key >> 32));
    bi = (int)(unsigned)key;
}

// Distances of the search kernels' sphere tests.  sqrtf() is the correctly rounded sequence (~15 vector instructions) and
// these kernels are vector-ALU bound (SQ_ACTIVE_INST_VALU of k_nearest_invlbs, profiles/r04e_pmc_sq.json); the tests only
// need BOUNDS.  v_sqrt_f32 is good to 1 ulp, the sum of squares under it to another 1.5: a factor of 1 -+ 5e-7 brackets
// the true distance, so a lower bound made from norm_lo never exceeds the true one (no cluster that can hold the nearest
// vertex is skipped: the search stays exact) and an upper bound made from norm_hi never falls below it.
__device__ __forceinline__ float norm_lo(float dx, float dy, float dz) {
    return __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * 0.9999995f;
}
__device__ __forceinline__ float norm_hi(float dx, float dy, float dz) {
    return __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * 1.0000005f;
}

__device__ __forceinline__ bool sphere_may_hold_nn(const f32x4 s, V3 p, float best) {
    const float dx = s[0] - p.x, dy = s[1] - p.y, dz = s[2] - p.z;
    const float lb = fmaxf(norm_lo(dx, dy, dz) - s[3], 0.f);
    return lb * lb <= best * 1.00001f + 1e-12f;   // slack covers the rounding of the bound itself
}

// SRC_POINTS : id = i, p = pts[i]                 -> x raw canonical
// SRC_RAYS   : id = list[i] (ray), p = o + t[ray] d -> x NORMALISED (sphere tracing evaluates the SDF there)
// SRC_SAMPLES: id = list[i] (q = ray*S + s), p = o + z[q] d -> x raw canonical
// T of the nearest vertex, blended once per frame by k_vertex_transforms (same blend(), same bits)
__device__ __forceinline__ void vertex_transform(const FrameDev& fr, int bi, float (&T)[16]) {
    const f32x4* q = reinterpret_cast<const f32x4*>(fr.vert_T + (size_t)bi * 16);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const f32x4 v = q[c];
        T[c * 4 + 0] = v[0];
        T[c * 4 + 1] = v[1];
        T[c * 4 + 2] = v[2];
        T[c * 4 + 3] = v[3];
    }
}

// the nearest vertex's transform, inverted; write the outputs of one query
template <int SRC>
__device__ __forceinline__ void nearest_finish(const FrameDev& fr, const BodyConst& bc, int i, int id, V3 p, int bi,
                                               int* idx_out, float* x_out, float* T_out, int as_seed) {
    float T[16];
    vertex_transform(fr, bi, T);
    V3 y = V3{p.x - bc.trans[0], p.y - bc.trans[1], p.z - bc.trans[2]};
    V3 xh = inverse_affine_apply(T, y);
    if (SRC == SRC_RAYS) xh = normalize_pt(bc, xh);
    if (idx_out) idx_out[id] = bi;
    x_out[(size_t)id * 3 + 0] = xh.x;
    x_out[(size_t)id * 3 + 1] = xh.y;
    x_out[(size_t)id * 3 + 2] = xh.z;
    if (SRC == SRC_SAMPLES && as_seed) {   // loop C starts from here: the target p - trans rides in the (exactly zero)
        T[12] = y.x;                       // T[3][0..2] of the start transform, k_canon_solve takes it from there
        T[13] = y.y;
        T[14] = y.z;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        reinterpret_cast<f32x4*>(T_out + (size_t)id * 16)[c] = f32x4{T[c * 4], T[c * 4 + 1], T[c * 4 + 2], T[c * 4 + 3]};
}

// one query; sv / ssph point either into LDS (STRIDE = kClusterLds) or straight at the frame buffer in
// global memory (STRIDE = kClusterSize), see k_nearest_invlbs
template <int SRC, int STRIDE>
__device__ __forceinline__ void nearest_invlbs_point(const FrameDev& fr, const BodyConst& bc, const KnnData& kd, const GridInfo& g,
                                                     const float* sv, const float* ssph, int i, int id,
                                                     V3 p, int* idx_out, float* x_out, float* T_out, int as_seed) {
    float best = 3.4e38f;
    int bi = 0x7fffffff;
    if (SRC == SRC_RAYS && idx_out) {   // sphere tracing: the previous step's nearest vertex bounds the search
        const int seed = idx_out[id];
        if (seed >= 0) {
            const float dx = fr.verts_raw[seed * 3] - p.x, dy = fr.verts_raw[seed * 3 + 1] - p.y,
                        dz = fr.verts_raw[seed * 3 + 2] - p.z;
            best = dx * dx + dy * dy + dz * dz;
            bi = seed;
        }
    }
    const float fx = (p.x - g.origin[0]) * g.inv_h, fy = (p.y - g.origin[1]) * g.inv_h, fz = (p.z - g.origin[2]) * g.inv_h;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    int cnt = 255;
    const unsigned char* cl = nullptr;
    if (fx >= 0.f && fy >= 0.f && fz >= 0.f && cx < g.dims[0] && cy < g.dims[1] && cz < g.dims[2]) {
        cl = kd.cells + ((size_t)(cz * g.dims[1] + cy) * g.dims[0] + cx) * kCellBytes;
        cnt = cl[0];
    }
    if (cnt != 255) {
        // four candidates at a time: their ids, spheres and lower bounds are independent loads / arithmetic, in flight
        // together (the walk is latency-bound: ~2 waves per SIMD next to the 119 KB table); each bound is still
        // compared against the running best at the moment its cluster comes up, so no extra cluster is scanned
#pragma unroll 1
        for (int k = 0; k < cnt; k += 4) {
            int c[4];
            float lb2[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = cl[min(1 + k + u, 63)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const f32x4 sp = reinterpret_cast<const f32x4*>(ssph)[c[u]];
                const float dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
                const float lb = fmaxf(norm_lo(dx, dy, dz) - sp[3], 0.f);
                lb2[u] = lb * lb;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k + u < cnt && lb2[u] <= best * 1.00001f + 1e-12f) scan_cluster<STRIDE>(sv, c[u], p, best, bi);
        }
    } else {
        // Outside the grid / overflowed cell: every cluster is a candidate, in k-d order -- not by distance, so the
        // running best would tighten slowly and most clusters would be scanned (measured: ~150 of 256 per wave for
        // the first sphere-tracing steps, whose points sit on the body's bounding box).  A first pass over the
        // spheres gives the bound "some vertex is within d(centre) + r"; the second pass scans only the clusters whose
        // sphere can beat it.  Still exact: the bound is attained by a vertex of a cluster that passes its own test.
        float cap = best;
#pragma unroll 4
        for (int c = 0; c < g.n_clusters; ++c) {
            const f32x4 sp = reinterpret_cast<const f32x4*>(ssph)[c];
            const float dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
            const float ub = norm_hi(dx, dy, dz) + sp[3];
            cap = fminf(cap, ub * ub * 1.00001f);
        }
#pragma unroll 1
        for (int c = 0; c < g.n_clusters; ++c)
            if (sphere_may_hold_nn(reinterpret_cast<const f32x4*>(ssph)[c], p, fminf(best, cap)))
                scan_cluster<STRIDE>(sv, c, p, best, bi);
    }
    nearest_finish<SRC>(fr, bc, i, id, p, bi, idx_out, x_out, T_out, as_seed);
}

// (d2, index) minimum over the wave, lowest index on ties
__device__ __forceinline__ void wave_argmin(float& d2, int& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float od = __shfl_xor(d2, o);
        const int oi = __shfl_xor(idx, o);
        if (od < d2 || (od == d2 && oi < idx)) {
            d2 = od;
            idx = oi;
        }
    }
}

// One WAVE per query (short lists): the lanes test the candidate clusters' spheres in parallel, then the
// surviving clusters are scanned one after the other with one vertex slot per lane.  Same result as the
// serial walk (exact nearest vertex, lowest index on ties); latency ~1-2 us instead of ~100 us.
// Returns the nearest vertex (valid in every lane).
__device__ __forceinline__ int nearest_vertex_wave(const KnnData& kd, const GridInfo& g, V3 p, float best, int bi,
                                                   int lane) {
    const float fx = (p.x - g.origin[0]) * g.inv_h, fy = (p.y - g.origin[1]) * g.inv_h, fz = (p.z - g.origin[2]) * g.inv_h;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    int cnt = 255;
    const unsigned char* cl = nullptr;
    if (fx >= 0.f && fy >= 0.f && fz >= 0.f && cx < g.dims[0] && cy < g.dims[1] && cz < g.dims[2]) {
        cl = kd.cells + ((size_t)(cz * g.dims[1] + cy) * g.dims[0] + cx) * kCellBytes;
        cnt = cl[0];
    }
    const int rounds = cnt == 255 ? (g.n_clusters + 63) / 64 : 1;
    for (int r = 0; r < rounds; ++r) {
        int c = -1;
        if (cnt == 255) {
            if (r * 64 + lane < g.n_clusters) c = r * 64 + lane;
        } else if (lane < cnt) {
            c = cl[1 + lane];
        }
        float lb2 = 3.4e38f, ub2 = 3.4e38f;
        if (c >= 0) {
            const f32x4 sp = reinterpret_cast<const f32x4*>(kd.spheres)[c];
            const float dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
            const float lb = fmaxf(norm_lo(dx, dy, dz) - sp[3], 0.f), ub = norm_hi(dx, dy, dz) + sp[3];
            lb2 = lb * lb;
            ub2 = ub * ub * 1.00001f;   // some vertex of the cluster is at most this far
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ub2 = fminf(ub2, __shfl_xor(ub2, o));
        const float bound = fminf(best, ub2);
        unsigned long long live = __ballot(c >= 0 && lb2 <= bound * 1.00001f + 1e-12f);
        while (live) {
            const int src = __ffsll((long long)live) - 1;
            live &= live - 1;
            const int cc = __shfl(c, src);
            const float clb2 = __shfl(lb2, src);
            if (clb2 > best * 1.00001f + 1e-12f) continue;   // wave-uniform
            float d2 = 3.4e38f;
            int vi = 0x7fffffff;
            if (lane < kClusterSize) {
                const f32x4 q = reinterpret_cast<const f32x4*>(kd.sorted4)[cc * kClusterSize + lane];
                const float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
                d2 = dx * dx + dy * dy + dz * dz;
                vi = __float_as_int(q[3]);
            }
            wave_argmin(d2, vi);
            if (d2 < best || (d2 == best && vi < bi)) {
                best = d2;
                bi = vi;
            }
        }
    }
    return bi;
}

// SIXTEEN LANES per query, four queries per wave: a cluster's 28 slots are two rounds of 16 lanes instead of one
// round that leaves 36 of 64 lanes idle, a cell's candidate list is tested 16 spheres at a time, and the serial
// tail of a query (blend + inverse, one lane) is paid once per four queries.  Same result as the serial walk and the
// one-wave walk (exact nearest vertex, lowest index on ties).  Every lane of the group holds the query and returns
// the answer; control flow is uniform per group, so the shuffles below only read lanes of the own group.
__device__ __forceinline__ void group16_argmin(float& d2, int& idx) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        const float od = __shfl_xor(d2, o);
        const int oi = __shfl_xor(idx, o);
        if (od < d2 || (od == d2 && oi < idx)) {
            d2 = od;
            idx = oi;
        }
    }
}

template <int STRIDE>
__device__ __forceinline__ int nearest_vertex_group16(const KnnData& kd, const float* sv, const float* ssph, const GridInfo& g,
                                                      V3 p, float best, int bi, int lane) {
    const int sub = lane & 15, gbase = lane & 48;
    const float fx = (p.x - g.origin[0]) * g.inv_h, fy = (p.y - g.origin[1]) * g.inv_h, fz = (p.z - g.origin[2]) * g.inv_h;
    const int cx = (int)floorf(fx), cy = (int)floorf(fy), cz = (int)floorf(fz);
    int cnt = 255;
    const unsigned char* cl = nullptr;
    if (fx >= 0.f && fy >= 0.f && fz >= 0.f && cx < g.dims[0] && cy < g.dims[1] && cz < g.dims[2]) {
        cl = kd.cells + ((size_t)(cz * g.dims[1] + cy) * g.dims[0] + cx) * kCellBytes;
        cnt = cl[0];
    }
    const bool all = cnt == 255;
    const int total = all ? g.n_clusters : cnt;
    // some vertex of cluster c is at most d(centre) + r away: the smallest such bound over the candidates prunes the
    // list before the first scan (outside the grid every cluster is a candidate, in k-d order, not by distance)
    float cap = best;
    for (int b = 0; b < total; b += 16) {
        const int k = b + sub;
        if (k < total) {
            const int c = all ? k : (int)cl[1 + k];
            const f32x4 sp = reinterpret_cast<const f32x4*>(ssph)[c];
            const float dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
            const float ub = norm_hi(dx, dy, dz) + sp[3];
            cap = fminf(cap, ub * ub * 1.00001f);
        }
        if (!all) break;   // inside the grid the lists are ordered by distance: the first 16 spheres are bound enough
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) cap = fminf(cap, __shfl_xor(cap, o));
    for (int b = 0; b < total; b += 16) {
        const int k = b + sub;
        int c = -1;
        float lb2 = 3.4e38f;
        if (k < total) {
            c = all ? k : (int)cl[1 + k];
            const f32x4 sp = reinterpret_cast<const f32x4*>(ssph)[c];
            const float dx = sp[0] - p.x, dy = sp[1] - p.y, dz = sp[2] - p.z;
            const float lb = fmaxf(norm_lo(dx, dy, dz) - sp[3], 0.f);
            lb2 = lb * lb;
        }
        const unsigned long long m = __ballot(c >= 0 && lb2 <= fminf(best, cap) * 1.00001f + 1e-12f);
        unsigned live = (unsigned)(m >> gbase) & 0xffffu;
        while (live) {
            const int src = __ffs((int)live) - 1;
            live &= live - 1;
            const int cc = __shfl(c, gbase + src);
            const float clb2 = __shfl(lb2, gbase + src);
            if (clb2 > best * 1.00001f + 1e-12f) continue;   // uniform over the group
            float d2 = 3.4e38f;
            int vi = 0x7fffffff;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int v = sub + 16 * r;
                if (v < kClusterSize) {
                    const f32x4 q = reinterpret_cast<const f32x4*>(sv)[cc * STRIDE + v];
                    const float dx = q[0] - p.x, dy = q[1] - p.y, dz = q[2] - p.z;
                    const float e2 = dx * dx + dy * dy + dz * dz;
                    const int ei = __float_as_int(q[3]);
                    if (e2 < d2 || (e2 == d2 && ei < vi)) {
                        d2 = e2;
                        vi = ei;
                    }
                }
            }
            group16_argmin(d2, vi);
            if (d2 < best || (d2 == best && vi < bi)) {
                best = d2;
                bi = vi;
            }
        }
    }
    return bi;
}

// SRC_POINTS : id = i, p = pts[i]                 -> x raw canonical
// SRC_RAYS   : id = list[i] (ray), p = o + t[ray] d -> x NORMALISED (sphere tracing evaluates the SDF there)
// SRC_SAMPLES: id = list[i] (q = ray*S + s), p = o + z[q] d -> x raw canonical
// Three kernels share the walk (launch_nearest picks; the list length lives on the device, so a kernel whose range the
// length is not in returns at once):
//   ray and point lists  : k_nearest_group -- SIXTEEN lanes per query, four queries per wave, tables from L2 (round 3;
//                          ARAH_KNN_GROUP=0: k_nearest_wave, one wave per query);
//   sample list, n >= wave_below: k_nearest_invlbs -- one THREAD per query against the vertex table staged in 119 KB of
//                          LDS (neighbouring samples walk the same clusters: the reads are broadcasts);
//   sample list, n <  wave_below: k_nearest_wave.
template <int SRC>
__device__ __forceinline__ V3 knn_point_of(const float* pts, const RaySet& rs, const float* depth, int n_steps,
                                           const int* list, int i, int& id) {
    if (SRC == SRC_POINTS) {
        id = i;
        return V3{pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    }
    id = list[i];
    return SRC == SRC_RAYS ? ray_point(rs, id, depth[id]) : ray_point(rs, id / n_steps, depth[id]);
}

constexpr int kKnnWaveThreads = 256;

template <int SRC>
__global__ __launch_bounds__(kKnnWaveThreads) void k_nearest_wave(FrameDev fr, KnnData kd, const float* pts, RaySet rs,
                                                                   const float* depth, int n_steps, const int* list,
                                                                   const int* count, int n_direct, int wave_below,
                                                                   int* idx_out, float* x_out, float* T_out,
                                                                   int as_seed, unsigned long long* ctr) {
    const BodyConst bc = load_bc(fr);
    const int n = (SRC == SRC_POINTS) ? n_direct : *count;
    if (n >= wave_below) return;
    const GridInfo g = *kd.grid;
    if (blockIdx.x == 0 && threadIdx.x == 0) count_add(ctr, n);
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave_global; i < n; i += n_waves) {
        int id;
        const V3 p = knn_point_of<SRC>(pts, rs, depth, n_steps, list, i, id);
        float best = 3.4e38f;
        int bi = 0x7fffffff;
        if (SRC == SRC_RAYS && idx_out) {   // sphere tracing: the previous step's nearest vertex bounds the search
            const int seed = idx_out[id];
            if (seed >= 0) {
                const float dx = fr.verts_raw[seed * 3] - p.x, dy = fr.verts_raw[seed * 3 + 1] - p.y,
                            dz = fr.verts_raw[seed * 3 + 2] - p.z;
                best = dx * dx + dy * dy + dz * dz;
                bi = seed;
            }
        }
        bi = nearest_vertex_wave(kd, g, p, best, bi, lane);
        if (lane == 0) nearest_finish<SRC>(fr, bc, i, id, p, bi, idx_out, x_out, T_out, as_seed);
    }
}

// nearest_finish over a group of sixteen lanes: lane e loads entry e of the vertex's transform (64 contiguous bytes per
// group); the entries then meet in every lane and lane 0 finishes the query
template <int SRC>
__device__ __forceinline__ void nearest_finish_group16(const FrameDev& fr, const BodyConst& bc, int id, V3 p, int bi,
                                                       int lane, int* idx_out, float* x_out, float* T_out, int as_seed) {
    const int sub = lane & 15;
    float te = fr.vert_T[(size_t)bi * 16 + sub];
    float T[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) T[e] = __shfl(te, (lane & 48) + e);
    if (sub == 0) {
        V3 y = V3{p.x - bc.trans[0], p.y - bc.trans[1], p.z - bc.trans[2]};
        V3 xh = inverse_affine_apply(T, y);
        if (SRC == SRC_RAYS) xh = normalize_pt(bc, xh);
        if (idx_out) idx_out[id] = bi;
        x_out[(size_t)id * 3 + 0] = xh.x;
        x_out[(size_t)id * 3 + 1] = xh.y;
        x_out[(size_t)id * 3 + 2] = xh.z;
    }
    if (SRC == SRC_SAMPLES && as_seed && sub >= 12 && sub < 15)
        te = sub == 12 ? p.x - bc.trans[0] : sub == 13 ? p.y - bc.trans[1] : p.z - bc.trans[2];
    T_out[(size_t)id * 16 + sub] = te;   // 64 contiguous bytes per query
}

// the sphere-tracing lists (a few 1e2 .. 1.5e5 rays): four queries per wave straight from L2
template <int SRC>
__global__ __launch_bounds__(kKnnWaveThreads) void k_nearest_group(FrameDev fr, KnnData kd, const float* pts, RaySet rs,
                                                                    const float* depth, int n_steps, const int* list,
                                                                    const int* count, int n_direct, int wave_below,
                                                                    int* idx_out, float* x_out, float* T_out,
                                                                    int as_seed, unsigned long long* ctr) {
    const BodyConst bc = load_bc(fr);
    const int n = (SRC == SRC_POINTS) ? n_direct : *count;
    if (n >= wave_below) return;
    const GridInfo g = *kd.grid;
    if (blockIdx.x == 0 && threadIdx.x == 0) count_add(ctr, n);
    const int lane = threadIdx.x & 63;
    const int group_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, n_groups = (gridDim.x * blockDim.x) >> 4;
    for (int i = group_global; i < n; i += n_groups) {
        int id;
        const V3 p = knn_point_of<SRC>(pts, rs, depth, n_steps, list, i, id);
        float best = 3.4e38f;
        int bi = 0x7fffffff;
        if (SRC == SRC_RAYS && idx_out) {   // sphere tracing: the previous step's nearest vertex bounds the search
            const int seed = idx_out[id];
            if (seed >= 0) {
                const float dx = fr.verts_raw[seed * 3] - p.x, dy = fr.verts_raw[seed * 3 + 1] - p.y,
                            dz = fr.verts_raw[seed * 3 + 2] - p.z;
                best = dx * dx + dy * dy + dz * dz;
                bi = seed;
            }
        }
        bi = nearest_vertex_group16<kClusterSize>(kd, kd.sorted4, kd.spheres, g, p, best, bi, lane);
        nearest_finish_group16<SRC>(fr, bc, id, p, bi, lane, idx_out, x_out, T_out, as_seed);
    }
}

template <int SRC>
__global__ __launch_bounds__(kKnnThreads) void k_nearest_invlbs(FrameDev fr, KnnData kd, const float* pts, RaySet rs,
                                                                 const float* depth, int n_steps, const int* list,
                                                                 const int* count, int n_direct, int wave_below,
                                                                 int* idx_out, float* x_out, float* T_out,
                                                                 int as_seed, unsigned long long* ctr) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = (SRC == SRC_POINTS) ? n_direct : *count;
    // ray / point lists are spread evenly over the grid (the table fill is per workgroup: one workgroup per CU,
    // all CUs busy); the big sample list is walked block-cyclically by a persistent grid
    const int per = SRC == SRC_SAMPLES ? 0 : ((((n + (int)gridDim.x - 1) / (int)gridDim.x) + 63) & ~63);
    if (n < wave_below || (SRC == SRC_SAMPLES ? (int)(blockIdx.x * blockDim.x) : (int)blockIdx.x * per) >= n) return;
    const GridInfo g = *kd.grid;
    if (blockIdx.x == 0 && threadIdx.x == 0) count_add(ctr, n);
    float* sv = smem;                              // [kMaxClusters][29][4] clustered vertices, one pad slot per cluster
    float* ssph = sv + (size_t)kMaxClusters * kClusterLds * 4;   // [kMaxClusters][4]
    for (int i = threadIdx.x; i < kMaxClusters * kClusterSize; i += blockDim.x)
        reinterpret_cast<f32x4*>(sv)[(i / kClusterSize) * kClusterLds + (i % kClusterSize)] =
            reinterpret_cast<const f32x4*>(kd.sorted4)[i];
    for (int i = threadIdx.x; i < kMaxClusters; i += blockDim.x)
        reinterpret_cast<f32x4*>(ssph)[i] = reinterpret_cast<const f32x4*>(kd.spheres)[i];
    __syncthreads();
    // Sample lists are ray-major (64 consecutive entries ~ one ray from near to far).  Within a block of 512 entries a
    // wave takes 8 runs of 8 consecutive entries, 64 apart: 8 neighbouring rays x 8 neighbouring depths instead of one
    // ray end to end, so that its lanes walk (mostly) the same clusters.
    const int t = threadIdx.x;
    if (SRC != SRC_SAMPLES) {
        const int end = min(n, ((int)blockIdx.x + 1) * per);
        for (int i = blockIdx.x * per + t; i < end; i += blockDim.x) {
            int id;
            const V3 p = knn_point_of<SRC>(pts, rs, depth, n_steps, list, i, id);
            nearest_invlbs_point<SRC, kClusterLds>(fr, bc, kd, g, sv, ssph, i, id, p, idx_out, x_out, T_out, as_seed);
        }
        return;
    }
    // (sixteen lanes per query against this table: 5.2 ms for the 8.6e6-sample list, against 2.6 ms with a thread per query (round 3; 1.15 ms since round 4's key compare) --
    // neighbouring samples walk the same clusters, so the serial walk's LDS reads are broadcasts; gpurun_out r3q)
    const int slot = (t & ~511) + ((t >> 3) & 7) * 64 + ((t >> 6) & 7) * 8 + (t & 7);
    for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + slot;
        if (i >= n) continue;
        int id;
        const V3 p = knn_point_of<SRC>(pts, rs, depth, n_steps, list, i, id);
        nearest_invlbs_point<SRC, kClusterLds>(fr, bc, kd, g, sv, ssph, i, id, p, idx_out, x_out, T_out, as_seed);
    }
}

// ------------------------------------------------------------------------------------------
// shared helpers of the MFMA kernels
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_count(int n) { return (n + kTile - 1) / kTile; }

// wave-0 append of kept ids to a device list
__device__ __forceinline__ void append_ids(bool keep, int id, int* list, int* count) {
    const unsigned long long m = __ballot(keep);
    int base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == 0 && m) base = atomicAdd(count, __popcll(m));
    base = __shfl(base, 0);
    if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = id;
}

// per-point tail of the skinning network: logits -> w -> T, x_bar  (RFU:99, 13-34).
// logit_row is this thread's private LDS row (>= 25 floats); it is overwritten with the 24 weights.
__device__ __forceinline__ void skin_tail(float* logit_row, const float* sbones, V3 xhat, float (&T)[16], V3& xbar) {
    hsoftmax_row(logit_row);
    blend(logit_row, sbones, T);
    xbar = apply34(T, xhat);
}

__device__ __forceinline__ void store_T(float* dst, const float (&T)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
        reinterpret_cast<f32x4*>(dst)[c] = f32x4{T[c * 4], T[c * 4 + 1], T[c * 4 + 2], T[c * 4 + 3]};
}

// ------------------------------------------------------------------------------------------
// unit seam: SDF value / feature / gradient at normalised points
// ------------------------------------------------------------------------------------------
constexpr int kSpillPerWg = 5 * kWaves * (kSdfMT * kNT) * 64;   // f32x4 elements

template <bool GRAD, bool SPLIT>
__global__ __launch_bounds__(kThreads, (GRAD || SPLIT) ? 2 : 4) void k_sdf_eval(FrameDev fr, const float* x_norm, const int* list,
                                                        const int* count, int n_direct, float* sdf_out,
                                                        float* feat_out, float* grad_out, f32x4* spill_all,
                                                        unsigned long long* ctr_fwd, unsigned long long* ctr_grad,
                                                        int grid_n) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                       // [64][4]
    float* outv = xin + 64 * 4;              // [64][4]
    int* ids = reinterpret_cast<int*>(outv + 64 * 4);   // [64]
    float* actA = reinterpret_cast<float*>(ids + 64);   // [64][260]
    float* actB = actA + 64 * kSdfLd;        // [64][260] (GRAD only)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = list ? *count : n_direct;
    f32x4* spill = spill_all ? spill_all + (size_t)blockIdx.x * kSpillPerWg : nullptr;
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            const int id = i < n ? (list ? list[i] : i) : -1;
            ids[tid] = id;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0 && grid_n > 0) {
                // lattice point id = (ix * N + iy) * N + iz of [-1, 1]^3, coordinates as sdf_meshing.py:26-38 forms them
                const float vs = (float)(2.0 / (double)(grid_n - 1));   // voxel_size is a Python float, rounded once (sdf_meshing.py:21)
                const int iz = id % grid_n, iy = (id / grid_n) % grid_n, ix = id / (grid_n * grid_n);
                // product and sum rounded separately, like the two tensor operations of the reference.  (__fmul_rn / __fadd_rn are
                // plain * and + to hipcc, which contracts them into one fma under its default -ffp-contract=fast: the asm pins
                // the rounded product.  Fixture F15 holds the reference's own coordinates: test_sdf_grid_is_the_references_lattice.)
                float px = (float)ix * vs, py = (float)iy * vs, pz = (float)iz * vs;
                asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
                x = f32x4{px + -1.0f, py + -1.0f, pz + -1.0f, 0.f};
            } else if (id >= 0) {
                x = f32x4{x_norm[(size_t)id * 3], x_norm[(size_t)id * 3 + 1], x_norm[(size_t)id * 3 + 2], 0.f};
            }
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][kNT];
        sdf_trunk<GRAD, kNT, SPLIT>(fr.sdf, xin, actA, kSdfLd, spill, dlast, wave, lane);
        sdf_head<SPLIT>(fr.sdf, actA, kSdfLd, outv, 4, tid);
        if (GRAD) sdf_backward(fr.sdf, actB, kSdfLd, spill, dlast, outv, 4, wave, lane, tid);
        __syncthreads();
        if (tid == 0) {
            const int cnt = min(kTile, n - tile * kTile);
            count_add(ctr_fwd, cnt);
            if (GRAD) count_add(ctr_grad, cnt);
        }
        if (tid < kTile && ids[tid] >= 0) {
            const int id = ids[tid];
            sdf_out[id] = outv[tid * 4];
            if (GRAD && grad_out) {
                grad_out[(size_t)id * 3 + 0] = outv[tid * 4 + 1];
                grad_out[(size_t)id * 3 + 1] = outv[tid * 4 + 2];
                grad_out[(size_t)id * 3 + 2] = outv[tid * 4 + 3];
            }
        }
        if (feat_out) {
            if (SPLIT) {
                unsplit_rows(actA, kSdfLd, tid);
                __syncthreads();
            }
            for (int e = tid; e < kTile * 64; e += kThreads) {   // 64 float4 per point
                const int pt = e >> 6, c4 = e & 63;
                if (ids[pt] >= 0)
                    reinterpret_cast<f32x4*>(feat_out + (size_t)ids[pt] * 256)[c4] =
                        *reinterpret_cast<const f32x4*>(actA + pt * kSdfLd + c4 * 4);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// loop A: one sphere-tracing step on the active rays (RT:198-241)
// ------------------------------------------------------------------------------------------
struct TraceState {
    float* t;            // [N] current depth
    const float* far;    // [N]
    float* xcur;         // [N][3] normalised canonical point of the last evaluation
    uint8_t* diverged;   // [N]
};

// Lists shorter than this are processed in 16-point tiles (NT = 1): a 64-point tile keeps one CU busy for
// ~100 us, and the tail iterations of the loops (a handful of stragglers) are pure latency.  Per-point
// results are bit-identical for both tile widths (same per-accumulator k-order, same reductions).
constexpr int kNarrowBelow = 16 * 512;

template <int NT, bool SPLIT>
__device__ __forceinline__ void sdf_march_tiles(const FrameDev& fr, const BodyConst& bc, const TraceState& st, const int* list, int n,
                                                int* next_list, int* next_count, unsigned long long* ctr_fwd,
                                                float* smem) {
    constexpr int TW = 16 * NT;
    float* xin = smem;
    float* outv = xin + 64 * 4;
    int* ids = reinterpret_cast<int*>(outv + 64 * 4);
    float* actA = reinterpret_cast<float*>(ids + 64);
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float scale = sdf_scale(bc);
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        if (tid < TW) {
            const int i = tile * TW + tid;
            const int id = i < n ? list[i] : -1;
            ids[tid] = id;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) x = f32x4{st.xcur[(size_t)id * 3], st.xcur[(size_t)id * 3 + 1], st.xcur[(size_t)id * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][NT];
        sdf_trunk<false, NT, SPLIT>(fr.sdf, xin, actA, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<SPLIT>(fr.sdf, actA, kSdfLd, outv, 4, tid, TW);
        __syncthreads();
        if (tid == 0) count_add(ctr_fwd, min(TW, n - tile * TW));
        if (tid < 64) {   // whole wave 0 takes part in the ballot
            const int id = tid < TW ? ids[tid] : -1;
            bool keep = false;
            if (id >= 0) {
                const float sdf = outv[tid * 4] * scale;                       // RT:528
                const float march = fminf(fmaxf(sdf, -kClampDist), kClampDist);   // RT:228
                bool div = false;
                if (fabsf(march) > kRootThresh) {                               // RT:231-235
                    const float t = st.t[id] + march;
                    st.t[id] = t;
                    div = t >= st.far[id];
                    st.diverged[id] = div ? 1 : 0;
                }
                keep = !((fabsf(sdf) <= kRootThresh) || div);                   // RT:238-241
            }
            append_ids(keep, id, next_list, next_count);
        }
        __syncthreads();
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(kThreads, 4) void k_sdf_march(FrameDev fr, TraceState st, const int* list, const int* count,
                                                         int* next_list, int* next_count,
                                                         unsigned long long* ctr_fwd) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = *count;
    if (n < kNarrowBelow) sdf_march_tiles<1, SPLIT>(fr, bc, st, list, n, next_list, next_count, ctr_fwd, smem);
    else sdf_march_tiles<kNT, SPLIT>(fr, bc, st, list, n, next_list, next_count, ctr_fwd, smem);
}

// ------------------------------------------------------------------------------------------
// unit seam: skinning weights / forward LBS
// ------------------------------------------------------------------------------------------
// n_items / per_item (arah_skin_lbs_counted): the number of points is min(n, *n_items * per_item), known on the device only
__global__ __launch_bounds__(kThreads, 4) void k_skin_eval(FrameDev fr, const float* x_hat, int n, float* w_out,
                                                         float* xbar_out, float* T_out, unsigned long long* ctr,
                                                         const int* n_items = nullptr, int per_item = 1) {
    if (n_items) n = (int)min((long long)n, (long long)max(*n_items, 0) * per_item);
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                        // [64][4] normalised
    float* xraw = xin + 64 * 4;               // [64][4] raw
    float* sbones = xraw + 64 * 4;            // [24][16]
    float* logits = sbones + 24 * 16;         // [64][33]
    float* act = logits + 64 * kLogitLd;      // [64][132]; 64*33 floats keeps the 16-byte alignment
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int i = tid; i < 24 * 16; i += kThreads) sbones[i] = fr.bones[i];
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            V3 p = V3{0.f, 0.f, 0.f};
            if (i < n) p = V3{x_hat[(size_t)i * 3], x_hat[(size_t)i * 3 + 1], x_hat[(size_t)i * 3 + 2]};
            const V3 q = normalize_pt(bc, p);
            reinterpret_cast<f32x4*>(xin)[tid] = f32x4{q.x, q.y, q.z, 0.f};
            reinterpret_cast<f32x4*>(xraw)[tid] = f32x4{p.x, p.y, p.z, 0.f};
        }
        __syncthreads();
        skin_mlp(fr.skin, xin, act, logits, wave, lane);
        if (tid == 0) count_add(ctr, min(kTile, n - tile * kTile));
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            if (i < n) {
                float T[16];
                V3 xb;
                skin_tail(logits + tid * kLogitLd, sbones, V3{xraw[tid * 4], xraw[tid * 4 + 1], xraw[tid * 4 + 2]}, T,
                          xb);
                if (w_out)
                    for (int c = 0; c < 24; ++c) w_out[(size_t)i * 24 + c] = logits[tid * kLogitLd + c];
                if (xbar_out) {
                    xbar_out[(size_t)i * 3] = xb.x;
                    xbar_out[(size_t)i * 3 + 1] = xb.y;
                    xbar_out[(size_t)i * 3 + 2] = xb.z;
                }
                if (T_out) store_T(T_out + (size_t)i * 16, T);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// d x_bar / d x_hat by forward-mode tangents through the skinning MLP (RFU:170-226): a tile is
// 16 points x {value, d/dx, d/dy, d/dz}; tile n=0 carries values, n=1..3 the tangents of the
// same 16 points, so every derivative factor is lane-local.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_skin_jac(FrameDev fr, const float* x_hat, const int* list,
                                                        const int* count, int n_direct, float* jac_out,
                                                        unsigned long long* ctr) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                        // [16][4] normalised
    float* xraw = xin + 16 * 4;               // [16][4]
    float* sbones = xraw + 16 * 4;            // [24][16]
    float* logits = sbones + 24 * 16;         // [64][33]  (col = n*16 + j)
    float* act = logits + 64 * kLogitLd;   // 64*33 floats is a multiple of 4: stays 16-byte aligned, stays an LDS pointer   // [64][132]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int n = list ? *count : n_direct;
    const int ld = kSkinLd;
    const float sN = 2.0f / (1.1f * (bc.cmax - bc.cmin));   // d x_norm / d x_hat
    for (int i = tid; i < 24 * 16; i += kThreads) sbones[i] = fr.bones[i];
    for (int tile = blockIdx.x; tile * 16 < n; tile += gridDim.x) {
        if (tid < 16) {
            const int i = tile * 16 + tid;
            V3 p = V3{0.f, 0.f, 0.f};
            if (i < n) {
                const int id = list ? list[i] : i;
                p = V3{x_hat[(size_t)id * 3], x_hat[(size_t)id * 3 + 1], x_hat[(size_t)id * 3 + 2]};
            }
            const V3 q = normalize_pt(bc, p);
            reinterpret_cast<f32x4*>(xin)[tid] = f32x4{q.x, q.y, q.z, 0.f};
            reinterpret_cast<f32x4*>(xraw)[tid] = f32x4{p.x, p.y, p.z, 0.f};
        }
        __syncthreads();
        {   // layer 0
            const int ch0 = wave * 16 + 4 * g;
            const f32x4 x = *reinterpret_cast<const f32x4*>(xin + j * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(fr.skin.bias + ch0);
            f32x4 h, t0, t1, t2;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(fr.skin.w0 + (ch0 + r) * 4);
                const float v = w[0] * x[0] + w[1] * x[1] + w[2] * x[2] + b[r];
                const float s = sigm(100.0f * v);
                h[r] = softplus100(v);
                t0[r] = s * w[0] * sN;
                t1[r] = s * w[1] * sN;
                t2[r] = s * w[2] * sN;
            }
            *reinterpret_cast<f32x4*>(act + (0 * 16 + j) * ld + ch0) = h;
            *reinterpret_cast<f32x4*>(act + (1 * 16 + j) * ld + ch0) = t0;
            *reinterpret_cast<f32x4*>(act + (2 * 16 + j) * ld + ch0) = t1;
            *reinterpret_cast<f32x4*>(act + (3 * 16 + j) * ld + ch0) = t2;
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 1; k < 4; ++k) {
            f32x4 acc[1][kNT];
#pragma unroll
            for (int nn = 0; nn < kNT; ++nn) zero_acc(acc[0][nn]);
            gemm_acc<8, 1>(fr.skin.wp[k - 1], wave, act, ld, acc, lane);
            __syncthreads();
            const int ch0 = wave * 16 + 4 * g;
            const f32x4 b = *reinterpret_cast<const f32x4*>(fr.skin.bias + k * 128 + ch0);
            f32x4 h, s;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[0][0][r] + b[r];
                h[r] = softplus100(v);
                s[r] = sigm(100.0f * v);
            }
            *reinterpret_cast<f32x4*>(act + (0 * 16 + j) * ld + ch0) = h;
#pragma unroll
            for (int nn = 1; nn < kNT; ++nn) *reinterpret_cast<f32x4*>(act + (nn * 16 + j) * ld + ch0) = acc[0][nn] * s;
            __syncthreads();
        }
        {
            const int mt = wave & 1, nt = wave >> 1;
            const f32x4 acc = gemm_one<8>(fr.skin.w4p, mt, nt, act, ld, lane);
            const int ch0 = mt * 16 + 4 * g;
            const f32x4 b = *reinterpret_cast<const f32x4*>(fr.skin.bias + 4 * 128 + ch0);
#pragma unroll
            for (int r = 0; r < 4; ++r) logits[(nt * 16 + j) * kLogitLd + ch0 + r] = acc[r] + (nt == 0 ? b[r] : 0.f);
        }
        __syncthreads();
        if (tid == 0) count_add(ctr, min(16, n - tile * 16));
        if (tid < 16) {
            const int i = tile * 16 + tid;
            if (i < n) {
                const int id = list ? list[i] : i;
                Dual3 x[25], w[24];
#pragma unroll
                for (int c = 0; c < 25; ++c) {
                    x[c].v = logits[(0 * 16 + tid) * kLogitLd + c] * 20.0f;
                    x[c].d[0] = logits[(1 * 16 + tid) * kLogitLd + c] * 20.0f;
                    x[c].d[1] = logits[(2 * 16 + tid) * kLogitLd + c] * 20.0f;
                    x[c].d[2] = logits[(3 * 16 + tid) * kLogitLd + c] * 20.0f;
                }
                hsoftmax<Dual3>(x, w);
                const float px = xraw[tid * 4], py = xraw[tid * 4 + 1], pz = xraw[tid * 4 + 2];
                float J[9];
#pragma unroll
                for (int e = 0; e < 9; ++e) J[e] = 0.f;
#pragma unroll
                for (int jn = 0; jn < 24; ++jn) {
                    const float* A = sbones + jn * 16;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float ax = A[r * 4] * px + A[r * 4 + 1] * py + A[r * 4 + 2] * pz + A[r * 4 + 3];
#pragma unroll
                        for (int c = 0; c < 3; ++c) J[r * 3 + c] += w[jn].v * A[r * 4 + c] + w[jn].d[c] * ax;
                    }
                }
#pragma unroll
                for (int e = 0; e < 9; ++e) jac_out[(size_t)id * 9 + e] = J[e];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// loop C: 3-D Broyden on g(x) = LBS(x) - target (RFU:267-362, broyden.py:4-78) as ONE resident kernel.
//
// A workgroup keeps 64 SLOTS (= the 64 columns of its MFMA tile).  Every pass of its loop evaluates the skinning
// MLP on the 64 slots (8 waves, split or exact engine), then the four lanes {j, j+16, j+32, j+48} of wave n < 4
// finish slot 16 n + j: hierarchical softmax, row g = lane >> 4 of the blended transform, residual, best iterate and
// the "good Broyden" bookkeeping.  The whole Broyden state of a point (x, step, g, J^-1, best x / T, |g|_best: 44
// floats) lives in the workgroup's LDS from the moment the point enters a slot until it retires; from HBM nothing
// but the 80-byte seed is read and nothing but the final best iterate (x, T, |g|) is written, exactly once.
// A slot whose point retires (converged / diverged / 1 + 50 evaluations) is refilled in the same pass from the
// launch-wide queue of seeds (one atomic per 64 seeds and wave), so tiles stay full while points need different
// numbers of iterations (mean 3.7, a fraction of a percent the full 51), and there is no per-iteration launch.
// Per-point arithmetic does not depend on the slot or on its neighbours: results are deterministic although the
// schedule is not.
//   seed = {x0(3), id | T0 rows 0..2 | target(3), T0[3][3]}: nearest-vertex start (RT:403-422) written by
//          k_nearest_invlbs<SAMPLES> or k_canon_seed.  T0 doubles as the initial best T (broyden.py:41); its last row
//          is (0, 0, 0, sum of the vertex weights) exactly, so the three zeros carry the target of g.
// ------------------------------------------------------------------------------------------
struct CanonOut {
    float* pts;    // [Q][3] raw canonical best x
    float* T;      // [Q][16]
    float* err;    // [Q]
};

constexpr int kSeedChunk = 64;   // seeds a wave takes from the queue per atomic

// 25 gates (sigmoids; entries 1..3 and 12..14 unused) + the two 3-way softmaxes -> 24 weights along the SMPL tree
// (utils/utils.py:138-181).  A gate q splits a parent's weight into child = parent q and parent (1 - q).
// SUB = false (the tile kernel k_canon_solve, i.e. the exact fp32 engine's loop C and the range guard's fallback): the
// remaining parent weight is parent * (1 - q), the reference's own expression and the arithmetic of hsoftmax<T> /
// hsoftmax_row (pointwise.hpp: loop B, arah_skin_lbs) -- every fp32-engine kernel blends with the same weights.
// SUB = true (k_canon_wave only): parent - child, one operation instead of two; parent + child stays the weight that came in
// to the last bit (the weights sum to one as exactly as the reference's do; each differs from its own (1 - q) product by
// one rounding, up to ~6e-8 absolute where q is close to one).
template <bool SUB>
__device__ __forceinline__ void hsoftmax_tree(const float (&sgm)[25], float ra, float rb, float rc, float sa, float sb, float sc,
                                              float (&w)[24]) {
    const float g0 = sgm[0];
    w[1] = g0 * ra;
    w[2] = g0 * rb;
    w[3] = g0 * rc;
    w[0] = 1.0f - g0;
    auto split = [&](int parent, int child, float q) {
        w[child] = w[parent] * q;
        w[parent] = SUB ? w[parent] - w[child] : w[parent] * (1.0f - q);
    };
#pragma unroll
    for (int k = 0; k < 3; ++k) split(1 + k, 4 + k, sgm[4 + k]);     // hips / spine1 -> knees / spine2
#pragma unroll
    for (int k = 0; k < 3; ++k) split(4 + k, 7 + k, sgm[7 + k]);     // -> ankles / spine3
#pragma unroll
    for (int k = 0; k < 2; ++k) split(7 + k, 10 + k, sgm[10 + k]);   // ankles -> feet
    const float up = w[9] * sgm[24];                                   // spine3 -> neck and collars
    w[12] = up * sa;
    w[13] = up * sb;
    w[14] = up * sc;
    w[9] = SUB ? w[9] - up : w[9] * (1.0f - sgm[24]);
    split(12, 15, sgm[15]);                                            // neck -> head
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl)   // 13,14 -> 16,17 -> 18,19 -> 20,21 -> 22,23
#pragma unroll
        for (int k = 0; k < 2; ++k) split((lvl == 0 ? 13 : 14 + 2 * lvl) + k, 16 + 2 * lvl + k, sgm[16 + 2 * lvl + k]);
}

// 25 raw logits of a slot -> 24 weights in registers; the four lanes of the slot share the 25 sigmoids through the
// slot's LDS row (in place).  Same arithmetic as hsoftmax<float>(20 * logits), the remaining parent weights included
// (hsoftmax_tree<false>).
// X20: the row already holds 20 x logit (k_canon_wave folds the factor into the accumulator's conversion).
template <bool X20 = false>
__device__ __forceinline__ void hsoftmax_quad(float* row, int g, float (&w)[24]) {
    float ra, rb, rc, sa, sb, sc;
    const float k20 = X20 ? 1.0f : 20.0f;
    softmax3<float>(row[1] * k20, row[2] * k20, row[3] * k20, ra, rb, rc);
    softmax3<float>(row[12] * k20, row[13] * k20, row[14] * k20, sa, sb, sc);
    float sg_own[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) sg_own[k] = sigm(row[min(g + 4 * k, 24)] * k20);
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (g + 4 * k < 25) row[g + 4 * k] = sg_own[k];
    // the four lanes are in one wave: LDS operations of a wave complete in order, only the compiler must not move
    // the reads below above the writes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float sgm[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) sgm[i] = row[i];
    hsoftmax_tree<false>(sgm, ra, rb, rc, sa, sb, sc, w);
}

// per-slot state block in LDS (floats): the Broyden state of the point in the slot between two passes
enum { ST_X = 0, ST_STEP = 3, ST_G = 6, ST_XB = 9, ST_J = 12, ST_EB = 21, ST_TG = 22, ST_ID = 25, ST_NEV = 26, ST_SIZE = 27 };   // odd stride: slots on distinct banks
constexpr size_t kLdsCanonSolve =
    (64 * 4 + 64 * kLogitLd + 24 * 16 + 16 + 64 * ST_SIZE + 64 * 16) * 4 + (size_t)64 * kSkinLd * 4;

template <bool SPLIT>
__global__ __launch_bounds__(kThreads, 4) void k_canon_solve(FrameDev fr, const int* __restrict__ list,
                                                             const int* count, int* queue_head, CanonOut outp,
                                                             unsigned long long* ctr, unsigned long long* ctr_canon,
                                                             unsigned long long* clk_out) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    KernelClk clk;
#ifdef ARAH_CLOCKS
    clk.start();
#endif
    float* xin = smem;                        // [64][4] normalised
    float* logits = xin + 64 * 4;             // [64][33]
    float* sbones = logits + 64 * kLogitLd;   // [24][16]; 64*33 floats is a multiple of 4: stays 16-byte aligned
    int* alive = reinterpret_cast<int*>(sbones + 24 * 16);   // [4] owner waves with a live slot (+ pad to 16)
    float* state = reinterpret_cast<float*>(alive + 16);     // [64][ST_SIZE]
    float* tbest = state + 64 * ST_SIZE;                     // [64][16] best T so far
    float* act = tbest + 64 * 16;                            // [64][132]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const bool owner = wave < 4;
    const int slot = (wave & 3) * 16 + j;
    float* row = logits + slot * kLogitLd;
    float* st = state + slot * ST_SIZE;
    int* sti = reinterpret_cast<int*>(st);
    f32x4* tb = reinterpret_cast<f32x4*>(tbest + slot * 16) + g;   // this lane's row of the slot's best T
    const int n = *count;
    for (int i = tid; i < 24 * 16; i += kThreads) sbones[i] = fr.bones[i];
    if (owner && g == 0) sti[ST_ID] = -1;
    int q_pos = 0, q_end = 0;        // this wave's private piece of the queue (wave-uniform)
    bool exhausted = n <= 0;
    int n_eval = 0;
    for (;;) {
        if (owner) {
            // ---- refill the empty slots of this wave (LDS operations of one wave complete in order)
            int id = sti[ST_ID];
            const unsigned long long m = __ballot(id < 0) & 0xffffull;   // lanes 0..15 speak for their slots
            int need = __popcll(m);
            const int rank = __popcll(m & ((1ull << j) - 1ull));
            int src = -1, given = 0;
            while (need > 0 && !exhausted) {   // wave-uniform
                if (q_pos == q_end) {
                    int p = 0;
                    if (lane == 0) p = atomicAdd(queue_head, kSeedChunk);
                    p = __builtin_amdgcn_readfirstlane(p);
                    if (p >= n) {
                        exhausted = true;
                        break;
                    }
                    q_pos = p;
                    q_end = min(p + kSeedChunk, n);
                }
                const int take = min(q_end - q_pos, need);
                if (id < 0 && rank >= given && rank < given + take) src = q_pos + (rank - given);
                q_pos += take;
                given += take;
                need -= take;
            }
            if (src >= 0) {   // the start state sits where the result will go: x0 in pts[id], T0 (+ target in row 3) in T[id]
                id = list[src];
                f32x4 t0 = reinterpret_cast<const f32x4*>(outp.T + (size_t)id * 16)[g];
                f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s4 = {0.f, 0.f, 0.f, 0.f};
                if (g == 0) s0 = f32x4{outp.pts[(size_t)id * 3], outp.pts[(size_t)id * 3 + 1], outp.pts[(size_t)id * 3 + 2], 0.f};
                if (g == 3) {
                    s4 = t0;                                   // {target, T0[3][3]}
                    t0 = f32x4{0.f, 0.f, 0.f, t0[3]};
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) s4[r] = __shfl(s4[r], j + 48);   // lane g == 0 files the target
                *tb = t0;                                      // T0 doubles as the initial best T (broyden.py:41)
                if (g == 0) {
                    const V3 tg = V3{s4[0], s4[1], s4[2]};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        st[ST_X + r] = s0[r];
                        st[ST_XB + r] = s0[r];
                    }
                    st[ST_TG] = tg.x;
                    st[ST_TG + 1] = tg.y;
                    st[ST_TG + 2] = tg.z;
                    sti[ST_ID] = id;
                    sti[ST_NEV] = 0;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (g == 0) {
                f32x4 xn = {0.f, 0.f, 0.f, 0.f};
                if (id >= 0) {
                    const V3 q = normalize_pt(bc, V3{st[ST_X], st[ST_X + 1], st[ST_X + 2]});
                    xn = f32x4{q.x, q.y, q.z, 0.f};
                }
                reinterpret_cast<f32x4*>(xin)[slot] = xn;
            }
            const unsigned long long live = __ballot(id >= 0) & 0xffffull;
            if (lane == 0) alive[wave] = live != 0ull;
            n_eval += __popcll(live);
        }
        clk.mark(0);
        __syncthreads();
        clk.mark(1);
        if (!(alive[0] | alive[1] | alive[2] | alive[3])) break;
        skin_mlp<kNT, SPLIT>(fr.skin, xin, act, logits, wave, lane, clk);   // ends with a barrier
        if (owner) {
            // ---- weights, row g of T = sum_j w_j A_j, x_bar, residual
            f32x4 Trow = {0.f, 0.f, 0.f, 0.f};
            {
                float w[24];
                hsoftmax_quad(row, g, w);
#pragma unroll
                for (int jn = 0; jn < 24; ++jn) {
                    if ((jn & 3) == 0) __builtin_amdgcn_sched_barrier(0);   // four bone rows in flight, not all 24 (VGPRs)
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sbones + jn * 16 + g * 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) Trow[c] = fmaf(w[jn], b[c], Trow[c]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int id = sti[ST_ID];
            const int nev = sti[ST_NEV];
            const float x0 = st[ST_X], x1 = st[ST_X + 1], x2 = st[ST_X + 2];
            const float xbar_g = fmaf(Trow[0], x0, fmaf(Trow[1], x1, fmaf(Trow[2], x2, Trow[3])));
            const float gn_g = xbar_g - st[ST_TG + min(g, 2)];
            float gnew[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) gnew[r] = __shfl(gn_g, j + 16 * r);
            const bool first = nev == 0;
            float J[9];
            if (__any(id >= 0 && first)) {   // wave-uniform: J^-1_0 = (T[:3,:3])^-1 from the same weights (RFU:327-328)
                float T9[16];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) T9[r * 4 + c] = __shfl(Trow[c], j + 16 * r);
                inv3_of44(T9, J);
            }
            int flags = 0;   // bit 0: keep, bit 1: improved
            if (g == 0 && id >= 0) {
                float gx[3], stp[3], eb;
                bool keep, improved = false;
                if (first) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) gx[r] = gnew[r];
                    eb = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
                    keep = true;                                        // every point takes at least one step
#pragma unroll
                    for (int r = 0; r < 3; ++r) stp[r] = -(J[r * 3] * gx[0] + J[r * 3 + 1] * gx[1] + J[r * 3 + 2] * gx[2]);
                } else {
                    float dg[3], dxv[3];
                    eb = st[ST_EB];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float gp = st[ST_G + r];
                        dxv[r] = st[ST_STEP + r];
                        dg[r] = gnew[r] - gp;
                        gx[r] = gp + dg[r];                             // broyden.py:50-51
                    }
                    const float err = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2]);
                    improved = err < eb;                                // broyden.py:54-61
                    if (improved) {
                        eb = err;
                        st[ST_XB] = x0;
                        st[ST_XB + 1] = x1;
                        st[ST_XB + 2] = x2;
                    }
                    keep = (eb > kRootThresh) && (err < kDvg);          // broyden.py:64
                    if (keep) {
#pragma unroll
                        for (int e = 0; e < 9; ++e) J[e] = st[ST_J + e];
                        broyden_update<3>(J, dxv, dg, gx, stp);          // broyden.py:69-75
                    }
                }
                if (nev + 1 > kBroydenSteps) keep = false;              // 1 + 50 evaluations (broyden.py:44)
                sti[ST_NEV] = nev + 1;
                st[ST_EB] = eb;
                if (keep) {
                    st[ST_X] = x0 + stp[0];
                    st[ST_X + 1] = x1 + stp[1];
                    st[ST_X + 2] = x2 + stp[2];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        st[ST_STEP + r] = stp[r];
                        st[ST_G + r] = gx[r];
                    }
#pragma unroll
                    for (int e = 0; e < 9; ++e) st[ST_J + e] = J[e];
                }
                flags = (keep ? 1 : 0) | (improved ? 2 : 0);
            }
            flags = __shfl(flags, j);
            if (flags & 2) *tb = Trow;
            if (id >= 0 && !(flags & 1)) {   // retire: the best iterate is the result (broyden.py:78)
                reinterpret_cast<f32x4*>(outp.T + (size_t)id * 16)[g] = *tb;
                if (g == 0) {
                    outp.pts[(size_t)id * 3] = st[ST_XB];
                    outp.pts[(size_t)id * 3 + 1] = st[ST_XB + 1];
                    outp.pts[(size_t)id * 3 + 2] = st[ST_XB + 2];
                    outp.err[id] = st[ST_EB];
                    sti[ST_ID] = -1;
                }
            }
        }
        clk.mark(10);
        __syncthreads();   // logits rows / alive flags are rewritten by the next pass
        clk.mark(11);
    }
    if (owner && lane == 0) {
        count_add(ctr, n_eval);
        count_add(ctr_canon, n_eval);
    }
#ifdef ARAH_CLOCKS
    if (lane == 0)
        for (int i = 0; i < 16; ++i) atomicAdd(&clk_out[wave * 16 + i], (unsigned long long)clk.acc[i]);
#endif
}

#include "mcubes.hpp"
#include "canon_wave.hpp"

// explicit targets (arah_broyden3_lbs): file them where k_canon_solve expects them, in row 3 of the start transform
__global__ void k_canon_seed(const int* list, const int* count, const float* tgt, float* T0) {
    const int n = *count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int id = list[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) T0[(size_t)id * 16 + 12 + c] = tgt[(size_t)id * 3 + c];
    }
}

// ------------------------------------------------------------------------------------------
// loop B: joint 4-D root find on u = (x_hat, depth)  (RFU:365-484)
// ------------------------------------------------------------------------------------------
struct Broyden4State {
    float* ueval;     // [N][4]
    float* step;      // [N][4]
    float* gx;        // [N][4]
    float* Jinv;      // [N][16]
    float* err_best;  // [N]
    float* xbest;     // [N][3] raw canonical
    float* zbest;     // [N]
    float* Tbest;     // [N][16]
};

// J = [[grad_sdf, 0], [J_lbs, -d]] -> J^-1   (RFU:406-418); also seeds ueval/xbest/zbest.
__global__ void k_joint_init(FrameDev fr, Broyden4State st, RaySet rs, const int* list, const int* count,
                             const float* grad_sdf, const float* jac_lbs, const float* xcur_norm, const float* t,
                             float* x0_raw) {
    const BodyConst bc = load_bc(fr);
    const int n = *count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int id = list[i];
        float J[16], Ji[16];
        J[0] = grad_sdf[(size_t)id * 3];
        J[1] = grad_sdf[(size_t)id * 3 + 1];
        J[2] = grad_sdf[(size_t)id * 3 + 2];
        J[3] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) J[(r + 1) * 4 + c] = jac_lbs[(size_t)id * 9 + r * 3 + c];
            J[(r + 1) * 4 + 3] = -rs.dirs[(size_t)id * 3 + r];
        }
        inv4(J, Ji);
#pragma unroll
        for (int e = 0; e < 16; ++e) st.Jinv[(size_t)id * 16 + e] = Ji[e];
        st.ueval[(size_t)id * 4 + 0] = x0_raw[(size_t)id * 3];
        st.ueval[(size_t)id * 4 + 1] = x0_raw[(size_t)id * 3 + 1];
        st.ueval[(size_t)id * 4 + 2] = x0_raw[(size_t)id * 3 + 2];
        st.ueval[(size_t)id * 4 + 3] = t[id];
    }
}

template <bool FIRST, int NT, bool SPLIT>
__device__ __forceinline__ void joint_tiles(const FrameDev& fr, const BodyConst& bc, const Broyden4State& st, const RaySet& rs, const int* list,
                                            int n, int* next_list, int* next_count, unsigned long long* ctr_skin,
                                            unsigned long long* ctr_sdf, float* smem) {
    constexpr int TW = 16 * NT;
    float* xin = smem;                        // [64][4] normalised
    float* xraw = xin + 64 * 4;               // [64][4] raw x_hat + depth
    float* outv = xraw + 64 * 4;              // [64][4]
    float* sbones = outv + 64 * 4;            // [24][16]
    int* ids = reinterpret_cast<int*>(sbones + 24 * 16);
    float* logits = reinterpret_cast<float*>(ids + 64);    // [64][33]
    float* act = logits + 64 * kLogitLd;   // 64*33 floats is a multiple of 4: stays 16-byte aligned, stays an LDS pointer   // [64][260]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float scale = sdf_scale(bc);
    for (int i = tid; i < 24 * 16; i += kThreads) sbones[i] = fr.bones[i];
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        if (tid < TW) {
            const int i = tile * TW + tid;
            const int id = i < n ? list[i] : -1;
            ids[tid] = id;
            f32x4 u = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) u = reinterpret_cast<const f32x4*>(st.ueval)[id];
            const V3 q = normalize_pt(bc, V3{u[0], u[1], u[2]});
            reinterpret_cast<f32x4*>(xin)[tid] = f32x4{q.x, q.y, q.z, 0.f};
            reinterpret_cast<f32x4*>(xraw)[tid] = u;
        }
        __syncthreads();
        skin_mlp<NT>(fr.skin, xin, act, logits, wave, lane);
        f32x4 dlast[kSdfMT][NT];
        sdf_trunk<false, NT, SPLIT>(fr.sdf, xin, act, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<SPLIT>(fr.sdf, act, kSdfLd, outv, 4, tid, TW);
        __syncthreads();
        if (tid == 0) {
            const int cnt = min(TW, n - tile * TW);
            count_add(ctr_skin, cnt);
            count_add(ctr_sdf, cnt);
        }
        if (tid < 64) {   // whole wave 0 takes part in the ballot
            const int id = tid < TW ? ids[tid] : -1;
            bool keep = false;
            if (id >= 0) {
                float T[16];
                V3 xb;
                const f32x4 u = reinterpret_cast<const f32x4*>(xraw)[tid];
                skin_tail(logits + tid * kLogitLd, sbones, V3{u[0], u[1], u[2]}, T, xb);
                const V3 p = ray_point(rs, id, u[3]);                    // RFU:435-436
                float gnew[4] = {outv[tid * 4] * scale, xb.x - (p.x - bc.trans[0]), xb.y - (p.y - bc.trans[1]),
                                 xb.z - (p.z - bc.trans[2])};
                float J[16], stp[4];
#pragma unroll
                for (int e = 0; e < 16; ++e) J[e] = st.Jinv[(size_t)id * 16 + e];
                if (FIRST) {
                    st.err_best[id] = sqrtf(gnew[0] * gnew[0] + gnew[1] * gnew[1] + gnew[2] * gnew[2] + gnew[3] * gnew[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float s = 0.f;
#pragma unroll
                        for (int c = 0; c < 4; ++c) s += J[r * 4 + c] * gnew[c];
                        stp[r] = -s;
                        st.gx[(size_t)id * 4 + r] = gnew[r];
                    }
                    keep = true;
                } else {
                    float gx[4], dg[4], dx[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dx[r] = st.step[(size_t)id * 4 + r];
                        const float gold = st.gx[(size_t)id * 4 + r];
                        dg[r] = gnew[r] - gold;
                        gx[r] = gold + dg[r];
                    }
                    const float err = sqrtf(gx[0] * gx[0] + gx[1] * gx[1] + gx[2] * gx[2] + gx[3] * gx[3]);
                    float eb = st.err_best[id];
                    if (err < eb) {
                        eb = err;
                        st.err_best[id] = err;
                        st.xbest[(size_t)id * 3] = u[0];
                        st.xbest[(size_t)id * 3 + 1] = u[1];
                        st.xbest[(size_t)id * 3 + 2] = u[2];
                        st.zbest[id] = u[3];
                        store_T(st.Tbest + (size_t)id * 16, T);
                    }
                    keep = (eb > kRootThresh) && (err < kDvg);
                    if (keep) {
                        broyden_update<4>(J, dx, dg, gx, stp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st.gx[(size_t)id * 4 + r] = gx[r];
#pragma unroll
                        for (int e = 0; e < 16; ++e) st.Jinv[(size_t)id * 16 + e] = J[e];
                    }
                }
                if (keep) {
                    reinterpret_cast<f32x4*>(st.step)[id] = f32x4{stp[0], stp[1], stp[2], stp[3]};
                    reinterpret_cast<f32x4*>(st.ueval)[id] = f32x4{u[0] + stp[0], u[1] + stp[1], u[2] + stp[2], u[3] + stp[3]};
                }
            }
            append_ids(keep, id, next_list, next_count);
        }
        __syncthreads();
    }
}

template <bool FIRST, bool SPLIT>
__global__ __launch_bounds__(kThreads) void k_joint_iter(FrameDev fr, Broyden4State st, RaySet rs, const int* list,
                                                          const int* count, int* next_list, int* next_count,
                                                          unsigned long long* ctr_skin,
                                                          unsigned long long* ctr_sdf) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int n = *count;
    if (n < kNarrowBelow) joint_tiles<FIRST, 1, SPLIT>(fr, bc, st, rs, list, n, next_list, next_count, ctr_skin, ctr_sdf, smem);
    else joint_tiles<FIRST, kNT, SPLIT>(fr, bc, st, rs, list, n, next_list, next_count, ctr_skin, ctr_sdf, smem);
}

// ------------------------------------------------------------------------------------------
// small per-ray / per-sample kernels of loops A..C
// ------------------------------------------------------------------------------------------
__global__ void k_trace_begin(const float* near_far, int n, float* t, float* far, uint8_t* diverged, float* xcur,
                              float* Tcur, int* list, int* count, const uint8_t* skip) {
    // RT:179-193: unfinished = near < far, diverged = near >= far; x/T start as zeros (RT:530-531)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const float nr = near_far[i * 2], fr_ = near_far[i * 2 + 1];
        t[i] = nr;
        far[i] = fr_;
        // skip (tiered forward, tier.hpp: k_tier_rays): a ray that cannot meet the surface starts as diverged
        const bool sk = skip && skip[i];
        diverged[i] = (nr >= fr_ || sk) ? 1 : 0;
        keep = nr < fr_ && !sk;
        xcur[(size_t)i * 3] = xcur[(size_t)i * 3 + 1] = xcur[(size_t)i * 3 + 2] = 0.f;
        for (int e = 0; e < 16; ++e) Tcur[(size_t)i * 16 + e] = 0.f;
    }
    append_ids(keep, i, list, count);
}

// after the 50 marching steps: un-normalise the start points, select the rays of the joint root
// find (eval: non-diverged, RT:249) and seed the best-iterate arrays with the sphere-tracing result.
__global__ void k_joint_select(FrameDev fr, int n, const float* xcur_norm, const float* Tcur, const float* t,
                               const uint8_t* diverged, int root_find_all, float* x0_raw, float* xbest, float* zbest,
                               float* Tbest, float* err_best, int* list, int* count) {
    const BodyConst bc = load_bc(fr);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const V3 xr = unnormalize_pt(bc, V3{xcur_norm[(size_t)i * 3], xcur_norm[(size_t)i * 3 + 1], xcur_norm[(size_t)i * 3 + 2]});   // RT:245
        x0_raw[(size_t)i * 3] = xbest[(size_t)i * 3] = xr.x;
        x0_raw[(size_t)i * 3 + 1] = xbest[(size_t)i * 3 + 1] = xr.y;
        x0_raw[(size_t)i * 3 + 2] = xbest[(size_t)i * 3 + 2] = xr.z;
        zbest[i] = t[i];
        for (int e = 0; e < 16; ++e) Tbest[(size_t)i * 16 + e] = Tcur[(size_t)i * 16 + e];
        err_best[i] = 3.4e38f;   // rays outside the root find never count as converged (RFU:473-476)
        keep = root_find_all || diverged[i] == 0;   // RT:249: all rays while training, the non-diverged ones in eval
    }
    append_ids(keep, i, list, count);
}

// unit seam of loop B (search_iso_surface_depth on caller-supplied starts, RFU:365-484): seed the per-ray arrays from
// (x0, z0, T0, valid); rays outside `valid` keep their start and never count as converged (RFU:473-484)
__global__ void k_joint_seed(FrameDev fr, int n, const float* x0, const float* z0, const float* T0, const uint8_t* valid,
                             float* x0_raw, float* xcur_norm, float* t, float* xbest, float* zbest, float* Tbest,
                             float* err_best, int* list, int* count) {
    const BodyConst bc = load_bc(fr);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool keep = false;
    if (i < n) {
        const V3 xr = V3{x0[(size_t)i * 3], x0[(size_t)i * 3 + 1], x0[(size_t)i * 3 + 2]};
        const V3 xn = normalize_pt(bc, xr);
        x0_raw[(size_t)i * 3] = xbest[(size_t)i * 3] = xr.x;
        x0_raw[(size_t)i * 3 + 1] = xbest[(size_t)i * 3 + 1] = xr.y;
        x0_raw[(size_t)i * 3 + 2] = xbest[(size_t)i * 3 + 2] = xr.z;
        xcur_norm[(size_t)i * 3] = xn.x;
        xcur_norm[(size_t)i * 3 + 1] = xn.y;
        xcur_norm[(size_t)i * 3 + 2] = xn.z;
        t[i] = zbest[i] = z0[i];
        for (int e = 0; e < 16; ++e) Tbest[(size_t)i * 16 + e] = T0[(size_t)i * 16 + e];
        err_best[i] = 3.4e38f;
        keep = valid[i] != 0;
    }
    append_ids(keep, i, list, count);
}

__global__ void k_joint_seam_out(int n, const float* xbest, const float* zbest, const float* err_best, const uint8_t* valid,
                                 float* x, float* z, uint8_t* conv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    x[(size_t)i * 3] = xbest[(size_t)i * 3];
    x[(size_t)i * 3 + 1] = xbest[(size_t)i * 3 + 1];
    x[(size_t)i * 3 + 2] = xbest[(size_t)i * 3 + 2];
    z[i] = zbest[i];
    conv[i] = (valid[i] != 0 && err_best[i] < kRootThresh) ? 1 : 0;
}

// RT:266-296
__global__ void k_trace_finalize(FrameDev fr, int n, const float* near_far, const float* xbest, const float* zbest,
                                 const float* err_best, float* points_hat_norm, uint8_t* conv, float* start,
                                 float* end) {
    const BodyConst bc = load_bc(fr);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float nr = near_far[i * 2], fa = near_far[i * 2 + 1];
    const float z = zbest[i];
    const bool c = (err_best[i] < kRootThresh) && (z >= nr) && (z <= fa);
    const V3 xn = normalize_pt(bc, V3{xbest[(size_t)i * 3], xbest[(size_t)i * 3 + 1], xbest[(size_t)i * 3 + 2]});
    points_hat_norm[(size_t)i * 3] = xn.x;
    points_hat_norm[(size_t)i * 3 + 1] = xn.y;
    points_hat_norm[(size_t)i * 3 + 2] = xn.z;
    conv[i] = c ? 1 : 0;
    start[i] = c ? z : nr;
    end[i] = fa;
}

#include "finish.hpp"

// stratified jitter of an ascending run v[0..m) (perturb_z_vals, RT:298-311): sample i moves inside
// [mid(i-1,i), mid(i,i+1)] by t in [0,1); `val(i)` evaluates the un-jittered run
template <typename F>
__device__ __forceinline__ float jittered(F val, int i, int m, float t) {
    const float v = val(i);
    const float lower = i == 0 ? v : 0.5f * (v + val(i - 1));
    const float upper = i == m - 1 ? v : 0.5f * (val(i + 1) + v);
    return lower + (upper - lower) * t;
}

// depth samples of one ray (RT:313-350): thread per ray.  lin_* are torch.linspace tables; rand_* are the
// caller's torch.rand draws of training mode (NULL in eval mode).
__global__ void k_sample_depths(int n, int S, int n_near, int n_far, const float* near_far, const uint8_t* conv,
                                const float* start, const float* end, const float* lin_s, const float* lin_near,
                                const float* lin_far, const float* rand_s, const float* rand_near,
                                const float* rand_far, float* z, uint8_t* mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float st = start[i], en = end[i];
    float* zr = z + (size_t)i * S;
    uint8_t* mr = mask + (size_t)i * S;
    auto uni = [&](int s) { return st + (en - st) * lin_s[s]; };
    auto uni_j = [&](int s) { return rand_s ? jittered(uni, s, S, rand_s[(size_t)i * S + s]) : uni(s); };
    if (!conv[i] || (n_near <= 0 && n_far <= 0)) {
        for (int s = 0; s < S; ++s) {
            zr[s] = uni_j(s);
            mr[s] = 1;
        }
        return;
    }
    const int nc = n_near + 1 + n_far;
    const float base = st - kSurfaceRange;
    const float nb = near_far[i * 2];
    const float span = fmaxf(st - kSurfaceRange - nb, 1e-5f);
    auto surf = [&](int a) { return base + (kSurfaceRange * 2.0f) * lin_near[a]; };
    auto farv = [&](int b) { return nb + span * lin_far[b]; };
    auto surf_j = [&](int a) {   // the surface sample itself (index near/2) is not moved (RT:334, 305-307)
        if (!rand_near) return surf(a);
        return jittered(surf, a, n_near + 1, a == n_near / 2 ? 0.5f : rand_near[(size_t)i * (n_near + 1) + a]);
    };
    auto far_j = [&](int b) { return rand_far ? jittered(farv, b, n_far, rand_far[(size_t)i * n_far + b]) : farv(b); };
    // merge the two ascending runs (== torch.sort of their concatenation, RT:348-350)
    int a = 0, b = 0;
    float va = surf_j(0), vb = n_far > 0 ? far_j(0) : 3.4e38f;
    for (int s = 0; s < nc; ++s) {
        if (vb <= va) {
            zr[s] = vb;
            ++b;
            vb = b < n_far ? far_j(b) : 3.4e38f;
        } else {
            zr[s] = va;
            ++a;
            va = a <= n_near ? surf_j(a) : 3.4e38f;
        }
        mr[s] = 1;
    }
    for (int s = nc; s < S; ++s) {
        zr[s] = uni_j(s);
        mr[s] = 0;
    }
}

// list of all q with flag[q] != 0.  One atomic per 4096 elements (a single device-scope counter saturates
// near 90 atomics/us): each thread takes 4 consecutive flags, waves are combined through LDS.
__global__ __launch_bounds__(1024) void k_build_list(const uint8_t* flag, int n, int* list, int* count) {
    __shared__ int wave_cnt[16];
    __shared__ int block_base;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = (blockIdx.x * 1024 + tid) * 4;
    unsigned bits = 0;
    if (i0 + 3 < n) {
        const unsigned w = *reinterpret_cast<const unsigned*>(flag + i0);   // n_steps-aligned rows keep this 4-byte aligned
        bits = (w & 0xffu ? 1u : 0u) | (w & 0xff00u ? 2u : 0u) | (w & 0xff0000u ? 4u : 0u) | (w & 0xff000000u ? 8u : 0u);
    } else {
        for (int k = 0; k < 4; ++k)
            if (i0 + k < n && flag[i0 + k]) bits |= 1u << k;
    }
    const int mine = __popc(bits);
    int incl = mine;   // inclusive prefix over the wave
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wave_cnt[wave] = incl;
    __syncthreads();
    if (tid == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) {
            const int c = wave_cnt[k];
            wave_cnt[k] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    int pos = block_base + wave_cnt[wave] + incl - mine;
    for (int k = 0; k < 4; ++k)
        if (bits & (1u << k)) list[pos++] = i0 + k;
}

__global__ void k_iota(int n, int* list, int* count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) list[i] = i;
    if (i == 0) *count = n;
}

// RT:447-461, 549-555: normalise the solution, converged = |g|_best < thr; masked-off samples are zeros
__global__ void k_canon_finalize(FrameDev fr, int nq, const uint8_t* sample_mask, const float* err_best, float* pts,
                                 float* T, uint8_t* conv) {
    const BodyConst bc = load_bc(fr);
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    if (!sample_mask[q]) {
        pts[(size_t)q * 3] = pts[(size_t)q * 3 + 1] = pts[(size_t)q * 3 + 2] = 0.f;
        for (int c = 0; c < 4; ++c) reinterpret_cast<f32x4*>(T + (size_t)q * 16)[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        conv[q] = 0;
        return;
    }
    const V3 xn = normalize_pt(bc, V3{pts[(size_t)q * 3], pts[(size_t)q * 3 + 1], pts[(size_t)q * 3 + 2]});
    pts[(size_t)q * 3] = xn.x;
    pts[(size_t)q * 3 + 1] = xn.y;
    pts[(size_t)q * 3 + 2] = xn.z;
    conv[q] = err_best[q] < kRootThresh ? 1 : 0;
}

__global__ void k_broyden3_finalize(int n, const float* err_best, float* err_out, uint8_t* conv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (err_out) err_out[i] = err_best[i];
    conv[i] = err_best[i] < kRootThresh ? 1 : 0;
}

// VolSDF density of a metric SDF value (IDR:366-368)
__device__ __forceinline__ float volsdf_density(float sdf, float inv_beta) {
    const float sgn = (-sdf > 0.f) ? 1.f : ((-sdf < 0.f) ? -1.f : 0.f);
    return fmaxf(inv_beta * (0.5f + 0.5f * sgn * (1.0f - expf(-fabsf(sdf) * inv_beta))), 0.f);
}

// ------------------------------------------------------------------------------------------
// loop D, pass 1 (exact lazy shading): SDF value -> density for every valid sample.  A sample whose
// density is exactly +0 (outside the surface by more than 16.6 beta: exp(-s / beta) < 2^-24, so 1 - exp(.) rounds to 1 and
// 0.5 - 0.5 * 1 is 0) has
// alpha = 1 - exp(-0 * delta) = 0 and weight 0 whatever its colour, and it scales the transmittance by
// (1 - 0 + 1e-7) independently of its colour (IDR:387-394).  Its normal and colour are therefore dead
// values: only samples with density > 0 go on to pass 2 (k_shade).  The composited image is
// bit-identical to shading everything (tests/test_hip_parity.py::test_lazy_shading_is_exact).
// ------------------------------------------------------------------------------------------
// NT = 8 (the split engine on long lists): tiles of 128 points, one workgroup per CU.  The weight fragments of a layer
// (256 KB as hi + lo halves) are then fetched from L2 once per 128 points instead of once per 64: at 64 points the
// fragment stream of this kernel is 146 GB per 512x512x64 frame, two thirds of what the L2 delivers, and the matrix pipe
// waits for it half of the time (rocprofv3 SQ counters, profiles/).
template <bool SPLIT, int NT = kNT>
__global__ __launch_bounds__(kThreads, NT > 4 ? 2 : 4) void k_density(FrameDev fr, const float* pts, const int* list, const int* count,
                                                       f32x4* shaded, int* next_list, int* next_count,
                                                       unsigned long long* ctr_fwd, unsigned long long* ctr_dens) {
    const BodyConst bc = load_bc(fr);
    constexpr int TW = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                   // [TW][4]
    float* outv = xin + TW * 4;                          // [TW][4]
    int* ids = reinterpret_cast<int*>(outv + TW * 4);    // [TW]
    float* actA = reinterpret_cast<float*>(ids + TW);    // [TW][kSdfLd]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#ifdef ARAH_PRIO_WAVES   // A/B (profiles/r05_ab_setprio.txt): static issue priority for one half of the workgroup's waves
    if ((wave >= kWaves / 2) == (ARAH_PRIO_WAVES == 1)) __builtin_amdgcn_s_setprio(1);
#endif
    const int n = *count;
    const float scale = sdf_scale(bc);
    const float inv_beta = 1.0f / fminf(fmaxf(fabsf(load_beta(fr)), 1e-6f), 1e6f);
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        if (tid < TW) {
            const int i = tile * TW + tid;
            const int id = i < n ? list[i] : -1;
            ids[tid] = id;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) x = f32x4{pts[(size_t)id * 3], pts[(size_t)id * 3 + 1], pts[(size_t)id * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
        __syncthreads();
        if constexpr (SPLIT && NT == 8) {
#ifdef ARAH_DENSITY_PP
            // the two 64-point halves a phase apart (mlp.hpp: sdf_trunk_pp).  Measured on the MI355X, same box, alternating
            // runs: 12.94-12.98 ms per launch against 12.91-13.08 without -- the pass is not bound by the serialisation of
            // GEMM and epilogue phases; the plain trunk stays the default, this is the reproducer
            sdf_trunk_pp<4>(fr.sdf, xin, actA, kSdfLd, wave, lane);
#else
            f32x4 dlast[kSdfMT][NT];
            sdf_trunk<false, NT, SPLIT>(fr.sdf, xin, actA, kSdfLd, nullptr, dlast, wave, lane);
#endif
        } else {
            f32x4 dlast[kSdfMT][NT];
            sdf_trunk<false, NT, SPLIT>(fr.sdf, xin, actA, kSdfLd, nullptr, dlast, wave, lane);
        }
        sdf_head<SPLIT>(fr.sdf, actA, kSdfLd, outv, 4, tid, TW);
        __syncthreads();
        if (tid == 0) {
            count_add(ctr_fwd, min(TW, n - tile * TW));
            count_add(ctr_dens, min(TW, n - tile * TW));
        }
        if (tid < TW) {   // whole waves: TW is a multiple of 64
            const int id = ids[tid];
            bool keep = false;
            if (id >= 0) {
                const float dens = volsdf_density(outv[tid * 4] * scale, inv_beta);
                shaded[id] = f32x4{0.f, 0.f, 0.f, dens};
                keep = dens > 0.f;
            }
            append_ids(keep, id, next_list, next_count);
        }
        __syncthreads();
    }
}

// The SDF on the N^3 lattice of [-1,1]^3 (arah_sdf_grid) as 128-point tiles of the split engine: the trunk of k_density<true, 8>
// (one workgroup per CU, the weight fragments fetched once per 128 points, the pipelined GEMM) with the lattice coordinates
// formed in the kernel as k_sdf_eval forms them.  16.8 M points per test.py frame: 30.7 ms through k_sdf_eval's 64-point tiles.
__global__ __launch_bounds__(kThreads, 2) void k_sdf_lattice(FrameDev fr, int grid_n, int n, float* __restrict__ sdf_out,
                                                             unsigned long long* ctr_fwd) {
    constexpr int NT = 8, TW = 16 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                   // [TW][4]
    float* outv = xin + TW * 4;                          // [TW][4]
    float* actA = outv + TW * 4 + TW;                    // [TW][kSdfLd] (k_density's layout: an id array sits in between there)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const float vs = (float)(2.0 / (double)(grid_n - 1));   // voxel_size is a Python float, rounded once (sdf_meshing.py:21)
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        if (tid < TW) {
            const int id = tile * TW + tid;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id < n) {
                const int iz = id % grid_n, iy = (id / grid_n) % grid_n, ix = id / (grid_n * grid_n);
                float px = (float)ix * vs, py = (float)iy * vs, pz = (float)iz * vs;   // product and sum rounded separately: see k_sdf_eval
                asm volatile("" : "+v"(px), "+v"(py), "+v"(pz));
                x = f32x4{px + -1.0f, py + -1.0f, pz + -1.0f, 0.f};
            }
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
        __syncthreads();
        f32x4 dlast[kSdfMT][NT];
        sdf_trunk<false, NT, true>(fr.sdf, xin, actA, kSdfLd, nullptr, dlast, wave, lane);
        sdf_head<true>(fr.sdf, actA, kSdfLd, outv, 4, tid, TW);
        __syncthreads();
        if (tid == 0) count_add(ctr_fwd, min(TW, n - tile * TW));
        if (tid < TW && tile * TW + tid < n) sdf_out[tile * TW + tid] = outv[tid * 4];
        __syncthreads();
    }
}

#ifdef ARAH_REG_TRUNK   // round 5 experiment, not in the shipped library: the density pass on point-owning waves
#include "regtrunk.hpp"
#endif

// ------------------------------------------------------------------------------------------
// loop D: SDF value + normal (reverse sweep) + colour MLP + VolSDF density per valid sample
// (IDR:291-368), then per-ray compositing (IDR:370-394)
// ------------------------------------------------------------------------------------------
// B3 (with SPLIT): the normal sweep and the colour MLP on the bf16 x 3 engine instead of the fp32 MFMA (mlp.hpp)
template <bool IDR, bool SPLIT, bool B3 = false>
__global__ __launch_bounds__(kThreads) void k_shade(FrameDev fr, int S, int cano_view_dirs, const float* dirs,
                                                     const float* pts, const float* T, const int* list,
                                                     const int* count, int n_direct, f32x4* shaded,
                                                     f32x4* spill_all, unsigned long long* ctr_fwd,
                                                     unsigned long long* ctr_grad, unsigned long long* ctr_col, B3Nets b3,
                                                     f32x4* sdfn_out = nullptr) {
    // sdfn_out (the per-sample seam arah_shade_points): {sdf (normalised units), d sdf / d x} as the sweep left them
    const BodyConst bc = load_bc(fr);
    typedef ColDims<IDR> D;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                        // [64][4]
    float* outv = xin + 64 * 4;               // [64][4] sdf, grad
    float* rgbv = outv + 64 * 4;              // [64][4]
    int* ids = reinterpret_cast<int*>(rgbv + 64 * 4);
    // B3: every activation of the normal sweep and of the colour MLP as bf16 hi / lo planes (mlp.hpp); the colour input
    // then needs ceil(kInPad / 32) 64-byte chunks per plane, i.e. rows of kInPad + 24 floats (8 x odd dwords apart mod 64)
    constexpr int ldA = B3 ? D::kInPad + 24 : D::kLdA;
    constexpr int loA = ((D::kInPad + 31) / 32) * 64;   // byte offset of the lo plane of A
    float* actA = reinterpret_cast<float*>(ids + 64);   // [64][ldA]
    float* actB = actA + 64 * ldA;                      // [64][kSdfLd]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = list ? *count : n_direct;
    f32x4* spill = spill_all + (size_t)blockIdx.x * kSpillPerWg;
    const float scale = sdf_scale(bc);
    const float beta = fminf(fmaxf(fabsf(load_beta(fr)), 1e-6f), 1e6f);   // IDR:366
    const float inv_beta = 1.0f / beta;
    // phases: 0 tile load, 1 forward trunk (7 its products, 8 its epilogues, 9 its barriers; 1 the rest), 2 head + resplit,
    // 3 reverse sweep (10 products, 11 epilogues, 12 barriers), 4 colour input, 5 colour MLP, 6 store
    KernelClk clk;
    clk.start();
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            const int id = i < n ? (list ? list[i] : i) : -1;
            ids[tid] = id;
            f32x4 x = {0.f, 0.f, 0.f, 0.f};
            if (id >= 0) x = f32x4{pts[(size_t)id * 3], pts[(size_t)id * 3 + 1], pts[(size_t)id * 3 + 2], 0.f};
            reinterpret_cast<f32x4*>(xin)[tid] = x;
        }
        __syncthreads();
        clk.mark(0);
        f32x4 dlast[kSdfMT][kNT];
        sdf_trunk<true, kNT, SPLIT, NoTap, KernelClk, 2, !B3>(fr.sdf, xin, actA, ldA, spill, dlast, wave, lane, NoTap(), &clk);
        clk.mark(1);
        sdf_head<SPLIT>(fr.sdf, actA, ldA, outv, 4, tid);
        if constexpr (B3) {
            resplit_rows_bf16(actA, ldA, loA, tid);   // the feature: f16 split planes -> bf16 planes
            clk.mark(2);
            sdf_backward_bp(fr.sdf, b3, actB, kSdfLd, spill, dlast, outv, 4, wave, lane, tid, &clk, xin);
        } else {
            if (SPLIT) unsplit_rows(actA, ldA, tid);   // the colour MLP (exact engine) reads the feature as fp32
            sdf_backward(fr.sdf, actB, kSdfLd, spill, dlast, outv, 4, wave, lane, tid);
        }
        __syncthreads();
        clk.mark(3);
        if (tid < kTile) {   // colour-input extras behind the feature: x(3), n(3), [PE4(view) 27], zero pad
            const int id = ids[tid];
            float ebuf[64];   // B3: staged here, written as planes below
            float* e = B3 ? ebuf : actA + tid * ldA + 256;
            float nx = outv[tid * 4 + 1], ny = outv[tid * 4 + 2], nz = outv[tid * 4 + 3];
            float vx = 0.f, vy = 0.f, vz = 0.f;
            if (id >= 0) {
                const float* Tq = T + (size_t)id * 16;
                const int ray = id / S;
                const float dx = -dirs[(size_t)ray * 3], dy = -dirs[(size_t)ray * 3 + 1], dz = -dirs[(size_t)ray * 3 + 2];
                if (cano_view_dirs) {                                   // IDR:295-298
                    float Tm[16], R[9];
#pragma unroll
                    for (int c = 0; c < 16; ++c) Tm[c] = Tq[c];
                    inv3_of44(Tm, R);
                    vx = R[0] * dx + R[1] * dy + R[2] * dz;
                    vy = R[3] * dx + R[4] * dy + R[5] * dz;
                    vz = R[6] * dx + R[7] * dy + R[8] * dz;
                } else {                                                // IDR:300, 340
                    vx = dx;
                    vy = dy;
                    vz = dz;
                    const float ax = Tq[0] * nx + Tq[1] * ny + Tq[2] * nz;
                    const float ay = Tq[4] * nx + Tq[5] * ny + Tq[6] * nz;
                    const float az = Tq[8] * nx + Tq[9] * ny + Tq[10] * nz;
                    nx = ax;
                    ny = ay;
                    nz = az;
                }
            }
            e[0] = xin[tid * 4];
            e[1] = xin[tid * 4 + 1];
            e[2] = xin[tid * 4 + 2];
            e[3] = nx;
            e[4] = ny;
            e[5] = nz;
            int k = 6;
            if (IDR) {                                                  // embedder.py:6-51, multires 4
                e[6] = vx;
                e[7] = vy;
                e[8] = vz;
                k = 9;
                float f = 1.0f;
#pragma unroll
                for (int o = 0; o < 4; ++o) {   // unrolled: e is a register array in the B3 instances, indexed by constants only
                    e[9 + 6 * o + 0] = sinf(vx * f);
                    e[9 + 6 * o + 1] = sinf(vy * f);
                    e[9 + 6 * o + 2] = sinf(vz * f);
                    e[9 + 6 * o + 3] = cosf(vx * f);
                    e[9 + 6 * o + 4] = cosf(vy * f);
                    e[9 + 6 * o + 5] = cosf(vz * f);
                    f *= 2.0f;
                }
                k = 33;
            }
#pragma unroll
            for (int kk = IDR ? 33 : 6; kk < D::kInPad - 256; ++kk) e[kk] = 0.f;
            if constexpr (B3) {
                constexpr int n_extra = ((D::kInPad + 31) / 32) * 32 - 256;   // up to the end of the last 32-chunk
#pragma unroll
                for (int q = 0; q < n_extra; ++q) store_bsplit1(actA, ldA, loA, tid, 256 + q, q < D::kInPad - 256 ? ebuf[q] : 0.f);
            }
        }
        __syncthreads();
        clk.mark(4);
        if constexpr (B3) color_mlp_bp<IDR>(fr.col, b3, actA, ldA, loA, actB, rgbv, 4, wave, lane, tid);
        else color_mlp<IDR>(fr.col, actA, actB, rgbv, 4, wave, lane, tid);
        __syncthreads();
        clk.mark(5);
        if (tid == 0) {
            const int cnt = min(kTile, n - tile * kTile);
            count_add(ctr_fwd, cnt);
            count_add(ctr_grad, cnt);
            count_add(ctr_col, cnt);
        }
        if (tid < kTile && ids[tid] >= 0) {
            const float dens = volsdf_density(outv[tid * 4] * scale, inv_beta);   // IDR:359, 368
            shaded[ids[tid]] = f32x4{rgbv[tid * 4], rgbv[tid * 4 + 1], rgbv[tid * 4 + 2], dens};
            if (sdfn_out) sdfn_out[ids[tid]] = reinterpret_cast<const f32x4*>(outv)[tid];
        }
        __syncthreads();
        clk.mark(6);
    }
#ifdef ARAH_CLOCKS
    if (lane == 0 && ctr_fwd)   // ctr_fwd is the first counter of the workspace's Counters
        for (int i = 0; i < 16; ++i)
            atomicAdd(&reinterpret_cast<Counters*>(ctr_fwd)->clk_shade[wave * 16 + i], (unsigned long long)clk.acc[i]);
#endif
}

// unit seam of the colour MLP alone
template <bool IDR>
__global__ __launch_bounds__(kThreads) void k_color_eval(FrameDev fr, const float* x_norm, const float* normal,
                                                          const float* view, const float* feat, int n, float* rgb,
                                                          unsigned long long* ctr) {
    const BodyConst bc = load_bc(fr);
    typedef ColDims<IDR> D;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* rgbv = smem;                        // [64][4]
    float* actA = rgbv + 64 * 4;
    float* actB = actA + 64 * D::kLdA;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    for (int tile = blockIdx.x; tile * kTile < n; tile += gridDim.x) {
        for (int e = tid; e < kTile * 64; e += kThreads) {
            const int pt = e >> 6, c4 = e & 63;
            const int i = tile * kTile + pt;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (i < n) v = reinterpret_cast<const f32x4*>(feat + (size_t)i * 256)[c4];
            *reinterpret_cast<f32x4*>(actA + pt * D::kLdA + c4 * 4) = v;
        }
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            float* e = actA + tid * D::kLdA + 256;
            int k = 0;
            for (; k < D::kInPad - 256; ++k) e[k] = 0.f;
            if (i < n) {
                for (int c = 0; c < 3; ++c) {
                    e[c] = x_norm[(size_t)i * 3 + c];
                    e[3 + c] = normal[(size_t)i * 3 + c];
                }
                if (IDR) {
                    const float vx = view[(size_t)i * 3], vy = view[(size_t)i * 3 + 1], vz = view[(size_t)i * 3 + 2];
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
            }
        }
        __syncthreads();
        color_mlp<IDR>(fr.col, actA, actB, rgbv, 4, wave, lane, tid);
        __syncthreads();
        if (tid == 0) count_add(ctr, min(kTile, n - tile * kTile));
        if (tid < kTile) {
            const int i = tile * kTile + tid;
            if (i < n)
                for (int c = 0; c < 3; ++c) rgb[(size_t)i * 3 + c] = rgbv[tid * 4 + c];
        }
        __syncthreads();
    }
}

// IDR:284-289, 370-394: left-pack the valid samples of a ray and integrate.
// A thread owns a ray and walks its S samples in order (the products of the transmittance are sequential).  CHUNK (S a
// multiple of 8): the 8 samples of a 128-byte line of `shaded` (and their depths and mask bytes) are requested together
// and then consumed -- sample by sample every lane touched its own line 8 times with the other 63 lanes' lines in
// between, and the lines did not survive in L1: 1.1 GB of fetches per frame for 0.18 GB of operands
// (profiles/r04a_pmc_traffic.json against r04e_pmc_traffic.json).  Same arithmetic in the same order.
template <bool CHUNK>
__global__ void k_composite(int n, int S, int render_last_pt, const float* z, const uint8_t* mask,
                            const f32x4* shaded, float* rgb, float* acc, uint8_t* vol_mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float inv_steps = 1.0f / (float)S;
    float r = 0.f, g = 0.f, b = 0.f, wsum = 0.f, trans = 1.0f;
    int prev = -1;
    f32x4 ps = {0.f, 0.f, 0.f, 0.f};
    float pz = 0.f;
    bool any = false;
    // one valid sample (s < S: its depth zs and shaded value sh) or the end of the ray (s == S)
    auto step = [&](int s, float zs, const f32x4 sh) {
        if (prev >= 0) {
            float delta;
            if (s < S) delta = zs - pz;
            else delta = render_last_pt ? 1e10f : inv_steps;           // last valid sample (IDR:379-385)
            const float alpha = 1.0f - expf(-ps[3] * delta);
            const float w = alpha * trans;
            r += ps[0] * w;
            g += ps[1] * w;
            b += ps[2] * w;
            wsum += w;
            trans *= (1.0f - alpha + 1e-7f);
        }
        if (s < S) {
            prev = s;
            ps = sh;
            pz = zs;
            any = true;
        }
    };
    const size_t base = (size_t)i * S;
    if (CHUNK) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        for (int s0 = 0; s0 < S; s0 += 8) {
            const u32x2 m8 = *reinterpret_cast<const u32x2*>(mask + base + s0);
            const f32x4 z0 = *reinterpret_cast<const f32x4*>(z + base + s0), z1 = *reinterpret_cast<const f32x4*>(z + base + s0 + 4);
            f32x4 sh[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) sh[u] = shaded[base + s0 + u];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (((u < 4 ? m8[0] : m8[1]) >> (8 * (u & 3))) & 0xffu) step(s0 + u, u < 4 ? z0[u & 3] : z1[u & 3], sh[u]);
        }
    } else {
        for (int s = 0; s < S; ++s)
            if (mask[base + s] != 0) step(s, z[base + s], shaded[base + s]);
    }
    step(S, 0.f, ps);
    rgb[(size_t)i * 3] = any ? r : 0.f;
    rgb[(size_t)i * 3 + 1] = any ? g : 0.f;
    rgb[(size_t)i * 3 + 2] = any ? b : 0.f;
    if (acc) acc[i] = any ? fminf(fmaxf(wsum, 0.f), 1.f) : 0.f;
    vol_mask[i] = any ? 1 : 0;
}

// IDR:114-115, 142-143, 251: camera-space surface points, zeroed off-surface
__global__ void k_points_cam(FrameDev fr, int n, RaySet rs, const float* dists, const uint8_t* conv,
                             const float* points_hat_norm, const float* __restrict__ pose, float* points_cam) {
    const BodyConst bc = load_bc(fr);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 pw = ray_point(rs, i, dists[i]);
    const float wx = (pw.x - bc.trans[0]) + bc.trans[0], wy = (pw.y - bc.trans[1]) + bc.trans[1],
                wz = (pw.z - bc.trans[2]) + bc.trans[2];
    const float ax = fabsf(points_hat_norm[(size_t)i * 3]), ay = fabsf(points_hat_norm[(size_t)i * 3 + 1]),
                az = fabsf(points_hat_norm[(size_t)i * 3 + 2]);
    const bool surf = conv[i] && ax <= 1.0f && ay <= 1.0f && az <= 1.0f;
    for (int r = 0; r < 3; ++r) {
        const float v = pose[r * 4] * wx + pose[r * 4 + 1] * wy + pose[r * 4 + 2] * wz + pose[r * 4 + 3];
        points_cam[(size_t)i * 3 + r] = surf ? v : 0.f;
    }
}

#include "tier.hpp"
}  // namespace
#include "train.hpp"
#include "meshquery.hpp"
namespace {

// ------------------------------------------------------------------------------------------
// host side: workspace carving, launch helpers
// ------------------------------------------------------------------------------------------
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {
    char* base;
    size_t off;
    template <typename Tp>
    Tp* take(size_t count) {
        off = align_up(off, 256);
        Tp* p = base ? reinterpret_cast<Tp*>(base + off) : nullptr;
        off += count * sizeof(Tp);
        return p;
    }
};

constexpr int kNumCounts = 64;   // per-iteration list sizes

struct Workspace {
    Counters* ctr;
    int* counts;          // [3][kNumCounts]
    int* listA;           // [max(N*S, N)]
    int* listB;
    f32x4* spill;         // [kMaxGrid][kSpillPerWg]
    // per ray
    float *t, *far, *xcur, *Tcur, *x0raw, *grad_sdf, *jac_lbs;
    uint8_t* diverged;
    int* nn_idx;
    float *u_eval, *u_step, *u_gx, *u_Jinv, *err_best_ray, *xbest_ray, *zbest_ray;
    // trace outputs when the caller keeps them in the workspace (arah_render)
    float *o_xnorm, *o_Tray, *o_start, *o_end, *o_acc;
    uint8_t* o_conv;
    // per sample
    float* q_err;
    float *o_z, *o_pts, *o_T;
    uint8_t* o_mask;
    uint8_t* q_smask;
    f32x4* shaded;
    // tiered eval forward: two more sample lists, its device-side counts, the tier of every ray
    int *listC, *listD;
    int* tcounts;         // [TC_COUNT]
    uint8_t* ray_tier;    // [N] 0 skipped (certified zero), 1 surface ray, 2 promoted
    size_t bytes;
};

Workspace carve(void* base, int n_rays, int n_steps) {
    Workspace w;
    Carver c{reinterpret_cast<char*>(base), 0};
    const size_t N = (size_t)(n_rays > 0 ? n_rays : 1), Q = N * (size_t)(n_steps > 0 ? n_steps : 1);
    // order matters: everything whose size does not depend on n_steps comes first, so that a
    // (n_rays, 1) carve and a (n_rays, S) carve agree on the per-ray offsets
    w.ctr = c.take<Counters>(1);
    w.counts = c.take<int>(3 * kNumCounts);
    w.spill = c.take<f32x4>((size_t)kMaxGrid * kSpillPerWg);
    w.t = c.take<float>(N);
    w.far = c.take<float>(N);
    w.xcur = c.take<float>(N * 3);
    w.Tcur = c.take<float>(N * 16);
    w.x0raw = c.take<float>(N * 3);
    w.grad_sdf = c.take<float>(N * 3);
    w.jac_lbs = c.take<float>(N * 9);
    w.diverged = c.take<uint8_t>(N);
    w.nn_idx = c.take<int>(N);
    w.u_eval = c.take<float>(N * 4);
    w.u_step = c.take<float>(N * 4);
    w.u_gx = c.take<float>(N * 4);
    w.u_Jinv = c.take<float>(N * 16);
    w.err_best_ray = c.take<float>(N);
    w.xbest_ray = c.take<float>(N * 3);
    w.zbest_ray = c.take<float>(N);
    w.o_xnorm = c.take<float>(N * 3);
    w.o_Tray = c.take<float>(N * 16);
    w.o_start = c.take<float>(N);
    w.o_end = c.take<float>(N);
    w.o_acc = c.take<float>(N);
    w.o_conv = c.take<uint8_t>(N);
    w.listA = c.take<int>(Q);
    w.listB = c.take<int>(Q);
    w.q_err = c.take<float>(Q);
    w.o_z = c.take<float>(Q);
    w.o_pts = c.take<float>(Q * 3);
    w.o_T = c.take<float>(Q * 16);
    w.o_mask = c.take<uint8_t>(Q);
    w.q_smask = c.take<uint8_t>(Q);
    w.shaded = c.take<f32x4>(Q);
    w.listC = c.take<int>(Q);
    w.listD = c.take<int>(Q);
    w.tcounts = c.take<int>(TC_COUNT);
    w.ray_tier = c.take<uint8_t>(N);
    w.bytes = align_up(c.off, 256);
    return w;
}

// The library's only process-wide data: tuning / diagnostic knobs, read from the environment ONCE (first use, thread-safe
// static initialisation) and immutable afterwards.  Nothing an entry point does depends on mutable global state; what a
// caller may want to change per call (engines, solver, events, shading mode) travels in ArahSampling / ArahFrame.
struct Knobs {
    int max_grid;        // ARAH_MAX_GRID          cap of the persistent grids (<= kMaxGrid, the spill slab is sized for it)
    bool split_solo;     // ARAH_SPLIT_SOLO=1      every split-engine workgroup owns its CU (diagnostic, see split_lds)
    int knn_group;       // ARAH_KNN_GROUP         sixteen-lane search for the ray lists
    int knn_below[3];    // ARAH_KNN_WAVE_{POINTS,RAYS,SAMPLES}  list lengths below which a wave per query is used
    int joint_bulk;      // ARAH_JOINT_BULK_ITERS  loop B iterations launched wide before the finisher
    int trace_bulk;      // ARAH_TRACE_BULK_STEPS  loop A steps launched wide before the finisher
    int trace_small;     // ARAH_TRACE_SMALL       ray lists up to this length go to the finisher at once
    bool density_wide;   // ARAH_DENSITY_TILE=128  128-point tiles in the density pass
    bool density_reg;    // ARAH_DENSITY_REG=1     the density pass on the point-owning trunk (regtrunk.hpp)
    int canon_lds_min;   // ARAH_CANON_LDS_MIN     loop C's point-owning-wave kernel asks for at least this much LDS (bytes): with
                         //                        more than half of the CU's 160 KB a half-size build (-DCW_WAVES=4) owns one slot per CU
    int canon_wg_per_cu; // ARAH_CANON_WG_PER_CU   resident workgroups of that kernel per CU (grid = this x CUs; 1)
    bool train_b3;       // ARAH_TRAIN_ENGINE!=fp32  bf16 x 3 / f16 split training kernels on split frames
    int canon_tier_wgs;  // ARAH_CANON_TIER_WGS    workgroups of loop C's solver on the tiered forward's lists (0 = one per CU)
};
inline const Knobs& knobs() {
    static const Knobs k = [] {
        auto env_int = [](const char* name, int dflt) {
            const char* e = getenv(name);
            return e ? atoi(e) : dflt;
        };
        Knobs v;
        const int g = env_int("ARAH_MAX_GRID", kMaxGrid);
        v.max_grid = g >= 1 && g <= kMaxGrid ? g : kMaxGrid;
        v.split_solo = env_int("ARAH_SPLIT_SOLO", 0) == 1;
        v.knn_group = env_int("ARAH_KNN_GROUP", 1);
        v.knn_below[0] = env_int("ARAH_KNN_WAVE_POINTS", 4096);
        v.knn_below[1] = env_int("ARAH_KNN_WAVE_RAYS", v.knn_group ? (1 << 30) : 32768);
        v.knn_below[2] = env_int("ARAH_KNN_WAVE_SAMPLES", 32768);
        v.joint_bulk = max(1, min(kBroydenSteps + 1, env_int("ARAH_JOINT_BULK_ITERS", 3)));
        v.trace_bulk = max(0, min(kSphereIters, env_int("ARAH_TRACE_BULK_STEPS", 24)));
        v.trace_small = env_int("ARAH_TRACE_SMALL", 4096);
        v.density_wide = env_int("ARAH_DENSITY_TILE", 128) == 128;
        v.density_reg = env_int("ARAH_DENSITY_REG", 0) == 1;
        v.canon_lds_min = max(0, min((int)kLdsCanonWave, env_int("ARAH_CANON_LDS_MIN", 0)));
        v.canon_wg_per_cu = max(1, min(4, env_int("ARAH_CANON_WG_PER_CU", 1)));
        v.canon_tier_wgs = max(0, env_int("ARAH_CANON_TIER_WGS", 0));
        const char* e = getenv("ARAH_TRAIN_ENGINE");
        v.train_b3 = !(e && strcmp(e, "fp32") == 0);
        return v;
    }();
    return k;
}

inline int grid_for(long long n_items, int per_block) {
    long long g = (n_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    const int cap = knobs().max_grid;
    if (g > cap) g = cap;
    return (int)g;
}

inline int check_launch() { return hipGetLastError() == hipSuccess ? ARAH_OK : ARAH_E_LAUNCH; }

// compute units of the current device (grid of the resident one-workgroup-per-CU kernels)
inline int num_cus() {
    constexpr int kMaxDevices = 64;
    static std::atomic<int> cus[kMaxDevices];   // 0 = not asked yet; every thread that asks stores the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 256;
    int v = cus[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}

// dynamic LDS sizes (bytes)
constexpr size_t kLdsSdfFwd = (64 * 4 * 2 + 64) * 4 + (size_t)64 * kSdfLd * 4;
constexpr size_t kLdsSdfGrad = kLdsSdfFwd + (size_t)64 * kSdfLd * 4;
constexpr size_t kLdsDensityWide = (128 * 4 * 2 + 128) * 4 + (size_t)128 * kSdfLd * 4;
constexpr size_t kLdsSkin = (64 * 4 * 3 + 24 * 16 + 64 + 64 * kLogitLd + 32) * 4 + (size_t)64 * kSkinLd * 4;
constexpr size_t kLdsJoint = (64 * 4 * 3 + 24 * 16 + 64 + 64 * kLogitLd + 32) * 4 + (size_t)64 * kSdfLd * 4;
constexpr size_t kLdsKnn = ((size_t)kMaxClusters * kClusterLds * 4 + kMaxClusters * 4) * 4;
template <bool IDR>
constexpr size_t lds_shade() {
    return (64 * 4 * 3 + 64) * 4 + (size_t)64 * ColDims<IDR>::kLdA * 4 + (size_t)64 * kSdfLd * 4;
}
template <bool IDR>
constexpr size_t lds_shade_b3() {
    return (64 * 4 * 3 + 64) * 4 + (size_t)64 * (ColDims<IDR>::kInPad + 24) * 4 + (size_t)64 * kSdfLd * 4;
}
template <bool IDR>
constexpr size_t lds_color() {
    return (64 * 4) * 4 + (size_t)64 * ColDims<IDR>::kLdA * 4 + (size_t)64 * kSdfLd * 4;
}

// Launch KS when the frame was prepared for the split engine, KE (exact fp32) otherwise.
// ARAH_SPLIT_SOLO=1 makes every split-engine workgroup own its CU (asks for more than half of the 160 KB LDS):
// the workaround that was in place until the irreproducibility of co-resident workgroups was traced to the packed
// K = 3 input layer (no_pack in mlp.hpp); kept as a diagnostic switch.
inline size_t split_lds(size_t lds) {
    constexpr size_t kSolo = 84 * 1024;
    return knobs().split_solo && lds < kSolo ? kSolo : lds;
}
constexpr size_t kLdsSplitSolo = 84 * 1024;
#define LAUNCH_ENGINE(split, KS, KE, GRID, BLOCK, LDS, ...)                              \
    do {                                                                                 \
        if (split) hipLaunchKernelGGL(KS, GRID, BLOCK, split_lds(LDS), __VA_ARGS__);     \
        else hipLaunchKernelGGL(KE, GRID, BLOCK, LDS, __VA_ARGS__);                      \
    } while (0)

// hipFuncSetAttribute is per DEVICE: the raised dynamic-LDS limits are set once for every device ordinal a call
// arrives on (the current device at the time of the call -- the host binding makes the buffers' device current), under
// a mutex: first calls may arrive on several host threads at the same time.
// A failure is NOT latched: a first call that arrives behind an earlier sticky HIP error fails, the next one tries again;
// only success is remembered (one atomic flag per device, the setup itself under a mutex).
template <typename K>
inline void allow_lds(K kernel, size_t bytes, bool& failed) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)bytes) != hipSuccess)
        failed = true;
}

int setup_attributes_once();
int setup_attributes() {
    constexpr int kMaxDevices = 64;
    static std::atomic<bool> done[kMaxDevices];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return ARAH_E_LAUNCH;
    if (done[dev].load(std::memory_order_acquire)) return ARAH_OK;
    std::lock_guard<std::mutex> lock(mu);
    if (done[dev].load(std::memory_order_relaxed)) return ARAH_OK;
    const int rc = setup_attributes_once();
    if (rc == ARAH_OK) done[dev].store(true, std::memory_order_release);
    return rc;
}

int setup_attributes_once() {
    bool failed = false;
    allow_lds(k_nearest_invlbs<SRC_POINTS>, kLdsKnn, failed);
    allow_lds(k_nearest_invlbs<SRC_RAYS>, kLdsKnn, failed);
    allow_lds(k_nearest_invlbs<SRC_SAMPLES>, kLdsKnn, failed);
    allow_lds(k_sdf_eval<false, false>, kLdsSdfFwd, failed);
    allow_lds(k_sdf_eval<false, true>, kLdsSplitSolo, failed);
    allow_lds(k_sdf_eval<true, false>, kLdsSdfGrad, failed);
    allow_lds(k_sdf_eval<true, true>, kLdsSdfGrad, failed);
    allow_lds(k_sdf_march<false>, kLdsSdfFwd, failed);
    allow_lds(k_sdf_march<true>, kLdsSplitSolo, failed);
    allow_lds(k_density<false>, kLdsSdfFwd, failed);
    allow_lds(k_density<true>, kLdsSplitSolo, failed);
    allow_lds(k_density<true, 8>, kLdsDensityWide, failed);
#ifdef ARAH_REG_TRUNK
    allow_lds(k_density_reg, kLdsRegTrunk, failed);
#endif
    allow_lds(k_sdf_lattice, kLdsDensityWide, failed);
    allow_lds(k_skin_eval, kLdsSkin, failed);
    allow_lds(k_skin_jac, kLdsSkin, failed);
    allow_lds(k_canon_solve<false>, kLdsCanonSolve, failed);
    allow_lds(k_canon_solve<true>, kLdsSplitSolo, failed);
    allow_lds((k_canon_wave<true, false>), kLdsCanonWave, failed);
    allow_lds((k_canon_wave<true, true>), kLdsCanonWave, failed);
    allow_lds((k_canon_wave<false, false>), kLdsCanonWave, failed);
    allow_lds((k_canon_wave<false, true>), kLdsCanonWave, failed);
    allow_lds(k_joint_iter<true, false>, kLdsJoint, failed);
    allow_lds(k_joint_iter<true, true>, kLdsSplitSolo, failed);
    allow_lds(k_joint_iter<false, false>, kLdsJoint, failed);
    allow_lds(k_joint_iter<false, true>, kLdsSplitSolo, failed);
    allow_lds(k_trace_finish<true>, kLdsTraceFinish, failed);
    allow_lds(k_trace_finish<false>, kLdsTraceFinish, failed);
    allow_lds(k_joint_finish<true>, kLdsJointFinish, failed);
    allow_lds(k_joint_finish<false>, kLdsJointFinish, failed);
    allow_lds(k_shade<false, false>, lds_shade<false>(), failed);
    allow_lds(k_shade<false, true>, lds_shade<false>(), failed);
    allow_lds(k_shade<true, false>, lds_shade<true>(), failed);
    allow_lds(k_shade<true, true>, lds_shade<true>(), failed);
    allow_lds((k_shade<false, true, true>), lds_shade_b3<false>(), failed);
    allow_lds((k_shade<true, true, true>), lds_shade_b3<true>(), failed);
    allow_lds(k_color_eval<false>, lds_color<false>(), failed);
    allow_lds(k_color_eval<true>, lds_color<true>(), failed);
    allow_lds(k_shade_train<false, false, false>, lds_shade_train<false>(), failed);
    allow_lds(k_shade_train<false, true, false>, lds_shade_train<false>(), failed);
    allow_lds(k_shade_train<true, false, false>, lds_shade_train<true>(), failed);
    allow_lds(k_shade_train<true, true, false>, lds_shade_train<true>(), failed);
    allow_lds(k_shade_train<false, false, true>, lds_shade_train<false>(), failed);
    allow_lds(k_shade_train<false, true, true>, lds_shade_train<false>(), failed);
    allow_lds(k_shade_train<true, false, true>, lds_shade_train<true>(), failed);
    allow_lds(k_shade_train<true, true, true>, lds_shade_train<true>(), failed);
    return failed ? ARAH_E_LAUNCH : ARAH_OK;
}

// optional profiling hook: events recorded around the k_shade launch (bench.py's roofline leg)

KnnData knn_of(const FrameDev& fd) {
    return KnnData{fd.knn.sorted4, fd.knn.spheres, reinterpret_cast<const GridInfo*>(fd.knn.grid), fd.knn.cells};
}

// nearest vertex + inverse LBS of up to n_max queries: the wave kernel and the bulk kernel, one of which returns at once
template <int SRC>
void launch_nearest(hipStream_t s, const FrameDev& fd, long long n_max, const float* pts, const RaySet& rs,
                    const float* depth, int n_steps, const int* list, const int* count, int n_direct, int* idx_out,
                    float* x_out, float* T_out, int as_seed, unsigned long long* ctr) {
    // list lengths below which one wave per query beats one thread per query (measured)
    // 512x512 frame: sphere-tracing lists (<= 1.5e5 rays, shrinking) 68.0 -> 65.4 ms per frame with the wave kernel below
    // 16k..64k entries; the 8.6e6-sample list of loop C wants the LDS table (86 ms when forced onto the wave kernel)
    // round 3: the sixteen-lane kernel takes the ray lists at every length (47.2 vs 48.0 ms per frame against handing
    // lists above 32k rays to the LDS-table kernel, gpurun_out r3n), so sphere tracing launches one search kernel per step
    const int group = knobs().knn_group;
    const int wave_below = knobs().knn_below[SRC];
    long long gw = (n_max * 64 + kKnnWaveThreads - 1) / kKnnWaveThreads;
    if (gw > 4096) gw = 4096;
    if (gw < 1) gw = 1;
    if (group && SRC != SRC_SAMPLES && n_max >= 0 && (SRC != SRC_POINTS || n_direct < wave_below)) {
        long long gg = (n_max * 16 + kKnnWaveThreads - 1) / kKnnWaveThreads;
        gg = gg > 4096 ? 4096 : gg < 1 ? 1 : gg;
        hipLaunchKernelGGL(k_nearest_group<SRC>, dim3((int)gg), dim3(kKnnWaveThreads), 0, s, fd, knn_of(fd), pts, rs, depth,
                           n_steps, list, count, n_direct, wave_below, idx_out, x_out, T_out, as_seed, ctr);
    } else if (n_max >= 0 && (SRC != SRC_POINTS || n_direct < wave_below))
        hipLaunchKernelGGL(k_nearest_wave<SRC>, dim3((int)gw), dim3(kKnnWaveThreads), 0, s, fd, knn_of(fd), pts, rs, depth,
                           n_steps, list, count, n_direct, wave_below, idx_out, x_out, T_out, as_seed, ctr);
    if (SRC == SRC_POINTS ? n_direct >= wave_below : n_max >= wave_below)
        hipLaunchKernelGGL(k_nearest_invlbs<SRC>, dim3(SRC == SRC_SAMPLES ? grid_for(n_max, kKnnThreads) : min(256, grid_for(n_max, 64))),
                           dim3(kKnnThreads), kLdsKnn, s, fd,
                           knn_of(fd), pts, rs, depth, n_steps, list, count, n_direct, wave_below, idx_out, x_out, T_out,
                           as_seed, ctr);
}

RaySet make_rays(const float* cam_loc, const float* dirs, int rays_per_cam) {
    RaySet rs;
    rs.cam_loc = cam_loc;
    rs.dirs = dirs;
    rs.rays_per_cam = rays_per_cam > 0 ? rays_per_cam : 1;
    return rs;
}

// ---- frame buffer layout -----------------------------------------------------------------
// ---- one launch per KIND of packing job (arah_prepare_frame: ~95 launches of a few microseconds each became 13) ----------
// A job list travels in the kernel arguments; a workgroup finds its job from the first-block table (uniform scan).
constexpr int kMaxPackJobs = 30, kMaxCopyJobs = 40, kMaxAbsmaxJobs = 12, kMaxSplitJobs = 16, kMaxB3Jobs = 24;
struct PackJob {
    float* dst;
    const float* src;
    int M, ld, m_tiles, KC, transpose, ncol2;
    ColSegs segs;
    int block0;
};
struct PackJobs {
    int n, blocks;
    PackJob j[kMaxPackJobs];
};
struct CopyJob {   // rows4 == 0: dst[i] = i < n ? src[i] : 0 for i < n_pad;  rows4 != 0: k_pad_rows4(rows = n, rows_pad = n_pad, ncol = rows4)
    float* dst;
    const float* src;
    int n, n_pad, rows4, block0;
};
struct CopyJobs {
    int n, blocks;
    CopyJob j[kMaxCopyJobs];
};
struct AbsmaxJob {
    const float* src;
    unsigned* amax;
    int n, blocks, block0, pad;
};
struct AbsmaxJobs {
    int n, blocks;
    AbsmaxJob j[kMaxAbsmaxJobs];
};
struct SplitJob {
    f16x8* dst;
    const float* src;
    const unsigned* amax;
    int M, ld, m_tiles, KC32, perm, block0;
};
struct SplitJobs {
    int n, blocks;
    SplitJob j[kMaxSplitJobs];
};
struct B3Job {
    bf16x8* dst;
    const float* packed;
    int m_tiles, KC16, block0, pad;
};
struct B3Jobs {
    int n, blocks;
    B3Job j[kMaxB3Jobs];
};

template <typename JOBS>
__device__ __forceinline__ int job_of_block(const JOBS& jobs, int b) {
    int k = 0;
    while (k + 1 < jobs.n && b >= jobs.j[k + 1].block0) ++k;
    return k;
}

__global__ void k_pack_multi(PackJobs jobs) {
    const int k = job_of_block(jobs, blockIdx.x);
    const PackJob& J = jobs.j[k];
    const int idx = (blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    const int total = J.m_tiles * J.KC * 64;
    if (idx >= total) return;
    const int lane = idx & 63, tile = idx >> 6;
    const int kc = tile % J.KC, mt = tile / J.KC;
    const int row = mt * 16 + (lane & 15);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (J.transpose == 2) {
        int sr = -1;
        for (int s = 0; s < J.segs.n; ++s)
            if (row >= J.segs.dst0[s] && row < J.segs.dst0[s] + J.segs.len[s]) sr = J.segs.src0[s] + (row - J.segs.dst0[s]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = kc * 16 + 4 * (lane >> 4) + t;
            if (sr >= 0 && col < J.ncol2) v[t] = J.src[(size_t)col * J.ld + sr];
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = kc * 16 + 4 * (lane >> 4) + t;
            int sc = -1;
            for (int s = 0; s < J.segs.n; ++s)
                if (col >= J.segs.dst0[s] && col < J.segs.dst0[s] + J.segs.len[s]) sc = J.segs.src0[s] + (col - J.segs.dst0[s]);
            if (sc >= 0 && row < J.M) v[t] = J.transpose ? J.src[(size_t)sc * J.ld + row] : J.src[(size_t)row * J.ld + sc];
        }
    }
    reinterpret_cast<f32x4*>(J.dst)[idx] = v;
}

__global__ void k_copy_multi(CopyJobs jobs) {
    const int k = job_of_block(jobs, blockIdx.x);
    const CopyJob& J = jobs.j[k];
    const int i = (blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    if (i >= J.n_pad) return;
    if (J.rows4) {   // rows of `rows4` floats -> rows of 4, zero padded
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < J.n)
            for (int c = 0; c < 4; ++c) v[c] = c < J.rows4 ? J.src[(size_t)i * J.rows4 + c] : 0.f;
        reinterpret_cast<f32x4*>(J.dst)[i] = v;
    } else {
        J.dst[i] = i < J.n ? J.src[i] : 0.f;
    }
}

__global__ void k_absmax_multi(AbsmaxJobs jobs) {
    const int k = job_of_block(jobs, blockIdx.x);
    const AbsmaxJob& J = jobs.j[k];
    float m = 0.f;
    for (int i = (blockIdx.x - J.block0) * blockDim.x + threadIdx.x; i < J.n; i += J.blocks * blockDim.x) {
        const float a = fabsf(J.src[i]);
        if (a == a) m = fmaxf(m, a);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(J.amax, __float_as_uint(m));
}

__global__ void k_zero_words(unsigned* a, int na, unsigned* b, int nb) {
    for (int i = threadIdx.x; i < na; i += blockDim.x) a[i] = 0u;
    for (int i = threadIdx.x; i < nb; i += blockDim.x) b[i] = 0u;
}

__global__ void k_pack_split_multi(SplitJobs jobs) {
    const int k = job_of_block(jobs, blockIdx.x);
    const SplitJob& J = jobs.j[k];
    const int idx = (blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    if (idx >= J.m_tiles * J.KC32 * 64) return;
    // the point-owning-wave operands (perm) stay UNSCALED: f16 subnormals carry the lo halves of small weights (mlp.hpp)
    const float scale = J.perm ? 1.0f : split_weight_scale(*J.amax);
    const int lane = idx & 63, tile = idx >> 6, kc = tile % J.KC32, mt = tile / J.KC32;
    const int row = mt * 16 + (lane & 15);
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = J.perm ? (2 * kc + (e >> 2)) * 16 + 4 * (lane >> 4) + (e & 3) : kc * 32 + (lane >> 4) * 8 + e;
        const float w = row < J.M ? J.src[(size_t)row * J.ld + col] * scale : 0.f;
        const _Float16 h = (_Float16)w;
        hi[e] = h;
        lo[e] = (_Float16)(w - (float)h);
    }
    J.dst[(size_t)(tile * 2 + 0) * 64 + lane] = hi;
    J.dst[(size_t)(tile * 2 + 1) * 64 + lane] = lo;
}

// ---- bf16 x 3 operands of the training kernels: made from the fp32 packings (any orientation / column permutation) ----
constexpr int kB3Count = 22;
struct B3Src {
    int m_tiles, kc16;
};
// 0..4 sdf_wp, 5..9 sdf_wpT, 10..15 colour w0p w1p w2p w3ap w3bp w4p, 16..21 colour w0pT w1pT w2pT w3apT w3bpT w4pT
inline B3Src b3_source(int i, int kc0) {
    if (i < 10) return B3Src{16, 16};
    switch (i) {
    case 10: return B3Src{16, kc0};
    case 11: return B3Src{16, 16};
    case 12: return B3Src{8, 16};
    case 13: return B3Src{16, kc0};
    case 14: return B3Src{16, 8};
    case 15: return B3Src{16, 16};
    case 16: return B3Src{kc0, 16};
    case 17: return B3Src{16, 16};
    case 18: return B3Src{16, 8};
    case 19: return B3Src{kc0, 16};
    case 20: return B3Src{8, 16};
    default: return B3Src{16, 16};
    }
}

// packed fp32 (k_pack: dst[((mt*KC16 + kc)*64 + lane)] = W[mt*16 + (lane&15)][kc*16 + 4 (lane>>4) + 0..3])
//   -> bf16 fragments dst[((mt*KC32 + kc)*2 + s)*64 + lane], lane (j, g) holds W[mt*16 + j][kc*32 + 8g .. +7], zero beyond K
__global__ void k_b3_from_packed(bf16x8* __restrict__ dst, const float* __restrict__ packed, int m_tiles, int KC16) {
    const int KC32 = (KC16 + 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m_tiles * KC32 * 64) return;
    const int lane = idx & 63, tile = idx >> 6, kc = tile % KC32, mt = tile / KC32;
    const int j = lane & 15, g = lane >> 4;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = kc * 32 + 8 * g + e, kc16 = col >> 4, gg = (col & 15) >> 2, r = col & 3;
        const float w = kc16 < KC16 ? packed[((size_t)(mt * KC16 + kc16) * 64 + gg * 16 + j) * 4 + r] : 0.f;
        const __bf16 h = (__bf16)w;
        hi[e] = h;
        lo[e] = (__bf16)(w - (float)h);
    }
    dst[(size_t)(tile * 2 + 0) * 64 + lane] = hi;
    dst[(size_t)(tile * 2 + 1) * 64 + lane] = lo;
}

__global__ void k_b3_from_packed_multi(B3Jobs jobs) {
    const int k = job_of_block(jobs, blockIdx.x);
    const B3Job& J = jobs.j[k];
    const int KC32 = (J.KC16 + 1) / 2;
    const int idx = (blockIdx.x - J.block0) * blockDim.x + threadIdx.x;
    if (idx >= J.m_tiles * KC32 * 64) return;
    const int lane = idx & 63, tile = idx >> 6, kc = tile % KC32, mt = tile / KC32;
    const int jj = lane & 15, g = lane >> 4;
    bf16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int col = kc * 32 + 8 * g + e, kc16 = col >> 4, gg = (col & 15) >> 2, r = col & 3;
        const float w = kc16 < J.KC16 ? J.packed[((size_t)(mt * J.KC16 + kc16) * 64 + gg * 16 + jj) * 4 + r] : 0.f;
        const __bf16 h = (__bf16)w;
        hi[e] = h;
        lo[e] = (__bf16)(w - (float)h);
    }
    J.dst[(size_t)(tile * 2 + 0) * 64 + lane] = hi;
    J.dst[(size_t)(tile * 2 + 1) * 64 + lane] = lo;
}

struct FrameLayout {
    size_t sdf_w0, sdf_wp[5], sdf_wpT[5], sdf_w6, sdf_b6, sdf_bias, sdf_freq, sdf_phase;
    size_t sdf_wps[5], sdf_fw, sdf_pw, sdf_fws, sdf_amax;
    size_t skin_w0, skin_wp[3], skin_w4p, skin_bias, skin_wps[4], skin_scales, skin_amax, skin_wpr, skin_wconsts;
    size_t col_w0p, col_w1p, col_w2p, col_w3ap, col_w3bp, col_w4p, col_w5, col_bias;
    size_t col_w0pT, col_w1pT, col_w2pT, col_w3apT, col_w3bpT, col_w4pT;   // transposed packings (training backward)
    size_t b3[kB3Count];   // bf16 hi/lo fragments of 22 of the packed matrices (training: gemm_acc_b3), order of B3Src
    size_t verts4, knn_spheres, knn_grid, knn_cells, scalars, vert_T;
    size_t bytes;
};

FrameLayout frame_layout(int col_mode) {
    FrameLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) {
        off = align_up(off, 256);
        size_t o = off;
        off += floats * 4;
        return o;
    };
    const int kin_pad = col_mode == ARAH_COLOR_IDR ? ColDims<true>::kInPad : ColDims<false>::kInPad;
    L.sdf_w0 = take(256 * 4);
    for (int i = 0; i < 5; ++i) L.sdf_wp[i] = take(256 * 256);
    for (int i = 0; i < 5; ++i) L.sdf_wpT[i] = take(256 * 256);
    L.sdf_w6 = take(256);
    L.sdf_b6 = take(64);
    L.sdf_bias = take(6 * 256);
    L.sdf_freq = take(6 * 256);
    L.sdf_phase = take(6 * 256);
    for (int i = 0; i < 5; ++i) L.sdf_wps[i] = take(256 * 256);   // hi + lo halves = 4 bytes per weight
    L.sdf_fw = take(6 * 256);
    L.sdf_pw = take(6 * 256);
    L.sdf_fws = take(6 * 256 + 64);   // + per-layer inverse operand scales
    L.sdf_amax = take(64);
    L.skin_w0 = take(128 * 4);
    for (int i = 0; i < 3; ++i) L.skin_wp[i] = take(128 * 128);
    L.skin_w4p = take(32 * 128);
    L.skin_bias = take(4 * 128 + 32);
    for (int i = 0; i < 3; ++i) L.skin_wps[i] = take(128 * 128);
    L.skin_wps[3] = take(32 * 128);
    L.skin_scales = take(64);
    L.skin_amax = take(64);
    L.skin_wpr = take(kCwWeightBytes / 4);
    L.skin_wconsts = take((kCwSize + 255) / 256 * 256);
    L.col_w0p = take((size_t)256 * kin_pad);
    L.col_w1p = take(256 * 256);
    L.col_w2p = take(128 * 256);
    L.col_w3ap = take((size_t)256 * kin_pad);
    L.col_w3bp = take(256 * 128);
    L.col_w4p = take(256 * 256);
    L.col_w5 = take(3 * 256);
    L.col_bias = take(256 + 256 + 128 + 256 + 256 + 4);
    L.col_w0pT = take((size_t)kin_pad * 256);
    L.col_w1pT = take(256 * 256);
    L.col_w2pT = take(256 * 128);
    L.col_w3apT = take((size_t)kin_pad * 256);
    L.col_w3bpT = take(128 * 256);
    L.col_w4pT = take(256 * 256);
    for (int i = 0; i < kB3Count; ++i) {
        const B3Src e = b3_source(i, kin_pad / 16);
        L.b3[i] = take((size_t)e.m_tiles * ((e.kc16 + 1) / 2) * 512);
    }
    L.verts4 = take((size_t)kMaxClusters * kClusterSize * 4);
    L.vert_T = take((size_t)kMaxVerts * 16);
    L.knn_spheres = take((size_t)kMaxClusters * 4);
    L.knn_grid = take(sizeof(GridInfo) / 4);
    L.knn_cells = take((size_t)kMaxCells * kCellBytes / 4);
    L.scalars = take(64);
    L.bytes = align_up(off, 256);
    return L;
}

void launch_pack(float* dst, const float* src, int M, int ld, int m_tiles, int KC, const ColSegs& segs, int transpose,
                 hipStream_t s, int ncol2 = 0) {
    const int total = m_tiles * KC * 64;
    hipLaunchKernelGGL(k_pack, dim3((total + 255) / 256), dim3(256), 0, s, dst, src, M, ld, m_tiles, KC, segs, transpose,
                       ncol2);
}

ColSegs one_seg(int len) {
    ColSegs s;
    memset(&s, 0, sizeof(s));
    s.n = 1;
    s.dst0[0] = 0;
    s.src0[0] = 0;
    s.len[0] = len;
    return s;
}

// host side of the multi-job packing kernels: collect, then one launch per kind
struct PrepJobs {
    PackJobs pack;
    CopyJobs copy;
    AbsmaxJobs amax;
    SplitJobs split;
    B3Jobs b3;
    bool ok = true;
    PrepJobs() {
        pack.n = pack.blocks = copy.n = copy.blocks = amax.n = amax.blocks = split.n = split.blocks = b3.n = b3.blocks = 0;
    }
    static int blocks_for(int items) { return (items + 255) / 256; }
    void add_pack(float* dst, const float* src, int M, int ld, int m_tiles, int KC, const ColSegs& segs, int transpose,
                  int ncol2 = 0) {
        if (pack.n >= kMaxPackJobs) { ok = false; return; }
        PackJob& j = pack.j[pack.n++];
        j.dst = dst; j.src = src; j.M = M; j.ld = ld; j.m_tiles = m_tiles; j.KC = KC; j.transpose = transpose; j.ncol2 = ncol2;
        j.segs = segs;
        j.block0 = pack.blocks;
        pack.blocks += blocks_for(m_tiles * KC * 64);
    }
    void add_copy(float* dst, const float* src, int n, int n_pad, int rows4 = 0) {
        if (copy.n >= kMaxCopyJobs) { ok = false; return; }
        CopyJob& j = copy.j[copy.n++];
        j.dst = dst; j.src = src; j.n = n; j.n_pad = n_pad; j.rows4 = rows4;
        j.block0 = copy.blocks;
        copy.blocks += blocks_for(n_pad);
    }
    void add_rows4(float* dst, const float* src, int rows, int rows_pad, int ncol) { add_copy(dst, src, rows, rows_pad, ncol); }
    void add_absmax(const float* src, int n, unsigned* dst, int blocks) {
        if (amax.n >= kMaxAbsmaxJobs) { ok = false; return; }
        AbsmaxJob& j = amax.j[amax.n++];
        j.src = src; j.amax = dst; j.n = n; j.blocks = blocks; j.pad = 0;
        j.block0 = amax.blocks;
        amax.blocks += blocks;
    }
    void add_split(f16x8* dst, const float* src, int M, int ld, int m_tiles, int KC32, const unsigned* mx, int perm) {
        if (split.n >= kMaxSplitJobs) { ok = false; return; }
        SplitJob& j = split.j[split.n++];
        j.dst = dst; j.src = src; j.amax = mx; j.M = M; j.ld = ld; j.m_tiles = m_tiles; j.KC32 = KC32; j.perm = perm;
        j.block0 = split.blocks;
        split.blocks += blocks_for(m_tiles * KC32 * 64);
    }
    void add_b3(bf16x8* dst, const float* packed, int m_tiles, int KC16) {
        if (b3.n >= kMaxB3Jobs) { ok = false; return; }
        B3Job& j = b3.j[b3.n++];
        j.dst = dst; j.packed = packed; j.m_tiles = m_tiles; j.KC16 = KC16; j.pad = 0;
        j.block0 = b3.blocks;
        b3.blocks += blocks_for(m_tiles * ((KC16 + 1) / 2) * 64);
    }
    void flush_a(hipStream_t s) {   // no dependencies
        if (pack.n) hipLaunchKernelGGL(k_pack_multi, dim3(pack.blocks), dim3(256), 0, s, pack);
        if (copy.n) hipLaunchKernelGGL(k_copy_multi, dim3(copy.blocks), dim3(256), 0, s, copy);
        if (amax.n) hipLaunchKernelGGL(k_absmax_multi, dim3(amax.blocks), dim3(256), 0, s, amax);
    }
    void flush_b(hipStream_t s) {   // after flush_a on the same stream
        if (split.n) hipLaunchKernelGGL(k_pack_split_multi, dim3(split.blocks), dim3(256), 0, s, split);
        if (b3.n) hipLaunchKernelGGL(k_b3_from_packed_multi, dim3(b3.blocks), dim3(256), 0, s, b3);
    }
};

// the nearest-vertex tables of one posed body, in a buffer of their own (arah_body_bytes) or inside the frame buffer
struct BodyTables {
    float* verts4 = nullptr;
    float* spheres = nullptr;
    GridInfo* grid = nullptr;
    unsigned char* cells = nullptr;
};
constexpr size_t kBodyOffSpheres = (size_t)kMaxClusters * kClusterSize * 16;
constexpr size_t kBodyOffGrid = kBodyOffSpheres + (size_t)kMaxClusters * 16;
constexpr size_t kBodyOffCells = kBodyOffGrid + ((sizeof(GridInfo) + 255) / 256) * 256;
constexpr size_t kBodyBytes = kBodyOffCells + (size_t)kMaxCells * kCellBytes;

BodyTables body_tables(void* buf) {
    char* b = reinterpret_cast<char*>(buf);
    BodyTables t;
    t.verts4 = reinterpret_cast<float*>(b);
    t.spheres = reinterpret_cast<float*>(b + kBodyOffSpheres);
    t.grid = reinterpret_cast<GridInfo*>(b + kBodyOffGrid);
    t.cells = reinterpret_cast<unsigned char*>(b + kBodyOffCells);
    return t;
}

void launch_body_tables(const float* verts, int n_verts, const BodyTables& t, hipStream_t s) {
    hipLaunchKernelGGL(k_sort_verts, dim3(1), dim3(1024), 0, s, verts, n_verts, t.verts4, t.grid);
    hipLaunchKernelGGL(k_cluster_spheres, dim3(1), dim3(256), 0, s, (const float*)t.verts4, t.spheres);
    hipLaunchKernelGGL(k_cell_clusters, dim3(kMaxCells / kCellThreads), dim3(kCellThreads), 0, s, (const GridInfo*)t.grid,
                       (const float*)t.spheres, t.cells);
}

}  // namespace

static B3Nets b3_of(const ArahFrame& f) {
    B3Nets b;
    for (int i = 0; i < 5; ++i) {
        b.sdf_wp[i] = reinterpret_cast<const bf16x8*>(f.b3[i]);
        b.sdf_wpT[i] = reinterpret_cast<const bf16x8*>(f.b3[5 + i]);
    }
    for (int i = 0; i < 6; ++i) {
        b.col[i] = reinterpret_cast<const bf16x8*>(f.b3[10 + i]);
        b.colT[i] = reinterpret_cast<const bf16x8*>(f.b3[16 + i]);
    }
    return b;
}

// Engine of loop D's normal sweep and colour MLP on a split-engine frame: bf16 x 3 (default), or the fp32 MFMA when the CALL
// says so (ArahSampling::shade_engine / the argument of arah_shade_points; ARAH_PRECISION_FP32 frames are fp32 throughout).
// Round 4: a field of the call, not an environment variable read into a process-wide static.
static bool shade_b3(int32_t shade_engine) { return shade_engine != ARAH_SHADE_ENGINE_FP32; }

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

size_t arah_body_bytes(void) { return kBodyBytes; }

int arah_prepare_body(const float* verts, int32_t n_verts, void* body_buf, size_t body_bytes, void* stream) {
    if (!verts || !body_buf || (reinterpret_cast<uintptr_t>(body_buf) & 255) != 0) return ARAH_E_BADARG;
    if (n_verts <= 0 || n_verts > kMaxVerts) return ARAH_E_SHAPE;
    if (body_bytes < kBodyBytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    launch_body_tables(verts, n_verts, body_tables(body_buf), reinterpret_cast<hipStream_t>(stream));
    return check_launch();
}

const char* arah_dominant_kernel(void) { return "k_canon_wave"; }   // largest single launch of the default path (loop C)

size_t arah_frame_bytes(const ArahNets* h_nets, const ArahBody* h_body) {
    if (!h_nets || !h_body) return 0;
    return frame_layout(h_nets->col_mode).bytes;
}

int arah_prepare_frame(const ArahNets* nets, const ArahBody* body, void* frame_buf, size_t frame_bytes,
                       ArahFrame* out, void* stream) {
    if (!nets || !body || !frame_buf || !out) return ARAH_E_BADARG;
    if (nets->col_mode != ARAH_COLOR_IDR && nets->col_mode != ARAH_COLOR_NO_VIEW_DIR) return ARAH_E_SHAPE;
    if (body->n_verts <= 0 || body->n_verts > kMaxVerts || nets->n_pose < 0) return ARAH_E_SHAPE;
    if (nets->precision != ARAH_PRECISION_SPLIT_F16 && nets->precision != ARAH_PRECISION_FP32) return ARAH_E_BADARG;
    const FrameLayout L = frame_layout(nets->col_mode);
    if (frame_bytes < L.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    char* base = reinterpret_cast<char*>(frame_buf);
    auto P = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    const bool idr = nets->col_mode == ARAH_COLOR_IDR;
    const int kin_pad = idr ? ColDims<true>::kInPad : ColDims<false>::kInPad;
    const int kc0 = kin_pad / 16;
    const int n_view = idr ? 27 : 0;
    const int in_dim = 3 + n_view + 3 + 256 + nets->n_pose;   // decoder.py:29-41, config.py:101-127
    // Every packing job is collected by kind and goes out as ONE launch per kind (PrepJobs): phase A has no dependencies
    // (fp32 packings, copies, |w| maxima); phase B needs A's maxima (f16 split packings, FiLM folding, skinning scales)
    // or A's packings (bf16 fragments).
    PrepJobs jobs;
    unsigned* amax_sdf = reinterpret_cast<unsigned*>(base + L.sdf_amax);
    unsigned* amax_skin = reinterpret_cast<unsigned*>(base + L.skin_amax);
    hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, amax_sdf, 64, amax_skin, 64);
    // ---- SDF MLP
    jobs.add_rows4(P(L.sdf_w0), nets->sdf_w[0], 256, 256, 3);
    for (int i = 0; i < 5; ++i) {
        jobs.add_pack(P(L.sdf_wp[i]), nets->sdf_w[i + 1], 256, 256, 16, 16, one_seg(256), 0);
        jobs.add_pack(P(L.sdf_wpT[i]), nets->sdf_w[i + 1], 256, 256, 16, 16, one_seg(256), 1);
    }
    jobs.add_copy(P(L.sdf_w6), nets->sdf_w[6], 256, 256);
    for (int i = 0; i < 6; ++i) jobs.add_copy(P(L.sdf_bias) + i * 256, nets->sdf_b[i], 256, 256);
    jobs.add_copy(P(L.sdf_freq), nets->film_freq, 6 * 256, 6 * 256);
    jobs.add_copy(P(L.sdf_phase), nets->film_phase, 6 * 256, 6 * 256);
    jobs.add_copy(P(L.sdf_b6), nets->sdf_b[6], 1, 64);
    for (int i = 0; i < 5; ++i) {
        jobs.add_absmax(nets->sdf_w[i + 1], 256 * 256, amax_sdf + i, 16);
        jobs.add_split(reinterpret_cast<f16x8*>(base + L.sdf_wps[i]), nets->sdf_w[i + 1], 256, 256, 16, 8, amax_sdf + i, 0);
    }
    // ---- skinning MLP
    jobs.add_rows4(P(L.skin_w0), nets->skin_w[0], 128, 128, 3);
    for (int i = 0; i < 3; ++i) jobs.add_pack(P(L.skin_wp[i]), nets->skin_w[i + 1], 128, 128, 8, 8, one_seg(128), 0);
    jobs.add_pack(P(L.skin_w4p), nets->skin_w[4], 25, 128, 2, 8, one_seg(128), 0);
    for (int i = 0; i < 4; ++i) jobs.add_copy(P(L.skin_bias) + i * 128, nets->skin_b[i], 128, 128);
    jobs.add_copy(P(L.skin_bias) + 512, nets->skin_b[4], 25, 32);
    SkinRaw raw;
    for (int i = 0; i < 4; ++i) {
        raw.w[i] = nets->skin_w[i];
        raw.b[i] = nets->skin_b[i];
    }
    hipLaunchKernelGGL(k_skin_probe, dim3(9 * 9 * 9), dim3(128), 0, s, raw, amax_skin);
    for (int i = 0; i < 4; ++i) {
        const int M = i < 3 ? 128 : 25, mt = i < 3 ? 8 : 2;
        jobs.add_absmax(nets->skin_w[i + 1], M * 128, amax_skin + 4 + i, 8);
        jobs.add_split(reinterpret_cast<f16x8*>(base + L.skin_wps[i]), nets->skin_w[i + 1], M, 128, mt, 4, amax_skin + 4 + i, 0);
        jobs.add_split(reinterpret_cast<f16x8*>(base + L.skin_wpr + (size_t)i * kCwLayerBytes), nets->skin_w[i + 1], M, 128, mt, 4,
                       amax_skin + 4 + i, 1);
    }
    // ---- colour MLP: permute input columns to [feat(256), x(3), n(3), view(27)], fold the pose tail
    {
        ColSegs sg;
        memset(&sg, 0, sizeof(sg));
        const int feat0 = 3 + n_view + 3;
        sg.n = idr ? 4 : 3;
        sg.dst0[0] = 0; sg.src0[0] = feat0; sg.len[0] = 256;             // feature
        sg.dst0[1] = 256; sg.src0[1] = 0; sg.len[1] = 3;                 // x
        sg.dst0[2] = 259; sg.src0[2] = 3 + n_view; sg.len[2] = 3;        // normal
        if (idr) { sg.dst0[3] = 262; sg.src0[3] = 3; sg.len[3] = 27; }   // PE(view)
        jobs.add_pack(P(L.col_w0p), nets->col_w[0], 256, in_dim, 16, kc0, sg, 0);
        jobs.add_pack(P(L.col_w3ap), nets->col_w[3], 256, in_dim + 128, 16, kc0, sg, 0);
        ColSegs sb = one_seg(128);
        sb.src0[0] = in_dim;                                             // cat([input, x], -1): x comes last
        jobs.add_pack(P(L.col_w3bp), nets->col_w[3], 256, in_dim + 128, 16, 8, sb, 0);
        jobs.add_pack(P(L.col_w1p), nets->col_w[1], 256, 256, 16, 16, one_seg(256), 0);
        jobs.add_pack(P(L.col_w2p), nets->col_w[2], 128, 256, 8, 16, one_seg(256), 0);
        jobs.add_pack(P(L.col_w4p), nets->col_w[4], 256, 256, 16, 16, one_seg(256), 0);
        // transposed operands of the training backward: rows = (permuted) inputs of the layer, K = its outputs
        jobs.add_pack(P(L.col_w0pT), nets->col_w[0], kin_pad, in_dim, kc0, 16, sg, 2, 256);
        jobs.add_pack(P(L.col_w3apT), nets->col_w[3], kin_pad, in_dim + 128, kc0, 16, sg, 2, 256);
        jobs.add_pack(P(L.col_w3bpT), nets->col_w[3], 128, in_dim + 128, 8, 16, sb, 2, 256);
        jobs.add_pack(P(L.col_w1pT), nets->col_w[1], 256, 256, 16, 16, one_seg(256), 1);
        jobs.add_pack(P(L.col_w2pT), nets->col_w[2], 256, 256, 16, 8, one_seg(128), 1);
        jobs.add_pack(P(L.col_w4pT), nets->col_w[4], 256, 256, 16, 16, one_seg(256), 1);
        jobs.add_copy(P(L.col_w5), nets->col_w[5], 3 * 256, 3 * 256);
        float* cb = P(L.col_bias);
        hipLaunchKernelGGL(k_fold_bias, dim3(1), dim3(256), 0, s, cb, nets->col_b[0], nets->col_w[0], in_dim,
                           feat0 + 256, nets->pose_vec, nets->pose_vec ? nets->n_pose : 0, 256);
        jobs.add_copy(cb + 256, nets->col_b[1], 256, 256);
        jobs.add_copy(cb + 512, nets->col_b[2], 128, 128);
        hipLaunchKernelGGL(k_fold_bias, dim3(1), dim3(256), 0, s, cb + 640, nets->col_b[3], nets->col_w[3],
                           in_dim + 128, feat0 + 256, nets->pose_vec, nets->pose_vec ? nets->n_pose : 0, 256);
        jobs.add_copy(cb + 896, nets->col_b[4], 256, 256);
        jobs.add_copy(cb + 1152, nets->col_b[5], 3, 4);
    }
    // ---- bf16 hi/lo fragments for loop D's normal sweep (W^T of the trunk) and colour MLP
    if (nets->precision == ARAH_PRECISION_SPLIT_F16) {   // always: the call decides whether it uses them
        const size_t src[kB3Count] = {L.sdf_wp[0], L.sdf_wp[1], L.sdf_wp[2], L.sdf_wp[3], L.sdf_wp[4],
                                      L.sdf_wpT[0], L.sdf_wpT[1], L.sdf_wpT[2], L.sdf_wpT[3], L.sdf_wpT[4],
                                      L.col_w0p, L.col_w1p, L.col_w2p, L.col_w3ap, L.col_w3bp, L.col_w4p,
                                      L.col_w0pT, L.col_w1pT, L.col_w2pT, L.col_w3apT, L.col_w3bpT, L.col_w4pT};
        for (int i = 5; i < 16; ++i) {
            const B3Src e = b3_source(i, kc0);
            jobs.add_b3(reinterpret_cast<bf16x8*>(base + L.b3[i]), (const float*)P(src[i]), e.m_tiles, e.kc16);
        }
    }
    if (!jobs.ok) return ARAH_E_LAUNCH;
    jobs.flush_a(s);
    jobs.flush_b(s);
    hipLaunchKernelGGL(k_fold_film, dim3(6), dim3(256), 0, s, (const float*)P(L.sdf_freq), (const float*)P(L.sdf_phase),
                       (const float*)P(L.sdf_bias), (const unsigned*)amax_sdf, P(L.sdf_fw), P(L.sdf_pw), P(L.sdf_fws));
    hipLaunchKernelGGL(k_skin_scales, dim3(1), dim3(64), 0, s, (const unsigned*)amax_skin, P(L.skin_scales));
    hipLaunchKernelGGL(k_skin_wave_consts, dim3(1), dim3(128), 0, s, raw, nets->skin_w[4], nets->skin_b[4],
                       (const unsigned*)amax_skin, P(L.skin_wconsts));
    // ---- body
    if (!body->trans || !body->center || !body->coord_min || !body->coord_max || !body->verts || !body->vert_weights ||
        !body->bones)
        return ARAH_E_BADARG;
    hipLaunchKernelGGL(k_gather_scalars, dim3(1), dim3(64), 0, s, P(L.scalars), body->trans, body->center, body->coord_min,
                       body->coord_max, nets->beta);
    // ---- body: k-d clustered vertices, cluster spheres, per-cell candidate clusters (exact 1-NN acceleration).  A caller
    // that has the posed vertices before the networks (SMPL runs first) can have built the tables already with
    // arah_prepare_body, on another stream next to the hypernetwork; it orders that stream before this one itself.
    BodyTables bt;
    if (body->prepared) {
        bt = body_tables(const_cast<void*>(body->prepared));
    } else {
        bt.verts4 = P(L.verts4);
        bt.spheres = P(L.knn_spheres);
        bt.grid = reinterpret_cast<GridInfo*>(base + L.knn_grid);
        bt.cells = reinterpret_cast<unsigned char*>(base + L.knn_cells);
        launch_body_tables(body->verts, body->n_verts, bt, s);
    }
    hipLaunchKernelGGL(k_vertex_transforms, dim3((body->n_verts + 63) / 64), dim3(64), 0, s, body->vert_weights, body->bones,
                       body->n_verts, P(L.vert_T));
    memset(out, 0, sizeof(*out));
    out->sdf_w0 = P(L.sdf_w0);
    for (int i = 0; i < 5; ++i) {
        out->sdf_wp[i] = P(L.sdf_wp[i]);
        out->sdf_wpT[i] = P(L.sdf_wpT[i]);
    }
    out->sdf_w6 = P(L.sdf_w6);
    out->sdf_bias = P(L.sdf_bias);
    out->sdf_freq = P(L.sdf_freq);
    out->sdf_phase = P(L.sdf_phase);
    for (int i = 0; i < 5; ++i) out->sdf_wps[i] = base + L.sdf_wps[i];
    out->sdf_fw = P(L.sdf_fw);
    out->sdf_pw = P(L.sdf_pw);
    out->sdf_fws = P(L.sdf_fws);
    out->precision = nets->precision;
    out->skin_w0 = P(L.skin_w0);
    for (int i = 0; i < 3; ++i) out->skin_wp[i] = P(L.skin_wp[i]);
    out->skin_w4p = P(L.skin_w4p);
    out->skin_bias = P(L.skin_bias);
    for (int i = 0; i < 4; ++i) out->skin_wps[i] = base + L.skin_wps[i];
    out->skin_scales = P(L.skin_scales);
    out->skin_wpr = base + L.skin_wpr;
    out->skin_wconsts = P(L.skin_wconsts);
    out->col_w0p = P(L.col_w0p);
    out->col_w1p = P(L.col_w1p);
    out->col_w2p = P(L.col_w2p);
    out->col_w3ap = P(L.col_w3ap);
    out->col_w3bp = P(L.col_w3bp);
    out->col_w4p = P(L.col_w4p);
    out->col_w5 = P(L.col_w5);
    out->col_bias = P(L.col_bias);
    out->col_w0pT = P(L.col_w0pT);
    out->col_w1pT = P(L.col_w1pT);
    out->col_w2pT = P(L.col_w2pT);
    out->col_w3apT = P(L.col_w3apT);
    out->col_w3bpT = P(L.col_w3bpT);
    out->col_w4pT = P(L.col_w4pT);
    for (int i = 0; i < kB3Count; ++i) out->b3[i] = base + L.b3[i];
    out->verts4 = bt.verts4;
    out->knn_spheres = bt.spheres;
    out->knn_grid = bt.grid;
    out->knn_cells = bt.cells;
    out->verts = body->verts;
    out->vert_T = P(L.vert_T);
    out->bones = body->bones;
    out->sdf_b6 = P(L.sdf_b6);
    out->scalars = P(L.scalars);
    out->n_verts = body->n_verts;
    out->col_mode = nets->col_mode;
    return check_launch();
}

size_t arah_workspace_bytes(int32_t n_rays, int32_t n_steps) {
    if (n_rays < 0 || n_steps < 0) return 0;
    return carve(nullptr, n_rays, n_steps).bytes;
}

int arah_counters_reset(void* workspace, void* stream) {
    if (!workspace) return ARAH_E_BADARG;
    Workspace w = carve(workspace, 1, 1);
    return hipMemsetAsync(w.ctr, 0, sizeof(Counters), reinterpret_cast<hipStream_t>(stream)) == hipSuccess ? ARAH_OK
                                                                                                           : ARAH_E_LAUNCH;
}

#if defined(ARAH_CLOCKS) || defined(RT_CLOCKS)
// instrumented builds only (tools/phase_clocks.py): the 2 x 8 x 16 phase clocks (loop C, k_shade) behind the counters;
// syncs the stream
int arah_debug_clocks(const void* workspace, unsigned long long* h_out, void* stream) {
    if (!workspace || !h_out) return ARAH_E_BADARG;
    Workspace w = carve(const_cast<void*>(workspace), 1, 1);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(h_out, w.ctr->clk, sizeof(w.ctr->clk) + sizeof(w.ctr->clk_shade), hipMemcpyDeviceToHost, s) != hipSuccess) return ARAH_E_LAUNCH;
    return hipStreamSynchronize(s) == hipSuccess ? ARAH_OK : ARAH_E_LAUNCH;
}
#endif

int arah_counters_read(const void* workspace, ArahCounters* h_out, void* stream) {
    if (!workspace || !h_out) return ARAH_E_BADARG;
    Workspace w = carve(const_cast<void*>(workspace), 1, 1);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(h_out, w.ctr, sizeof(ArahCounters), hipMemcpyDeviceToHost, s) != hipSuccess) return ARAH_E_LAUNCH;
    return hipStreamSynchronize(s) == hipSuccess ? ARAH_OK : ARAH_E_LAUNCH;
}

// ---- unit seams -----------------------------------------------------------------------------
int arah_sdf_eval(const ArahFrame* f, const float* x_norm, int32_t n, float* sdf, float* feat, float* grad,
                  void* workspace, size_t wbytes, void* stream) {
    if (!f || !x_norm || !sdf || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const int g = grid_for(n, kTile);
    if (grad)
        LAUNCH_ENGINE(fd.split, (k_sdf_eval<true, true>), (k_sdf_eval<true, false>), dim3(g), dim3(kThreads), kLdsSdfGrad,
                      s, fd, x_norm, (const int*)nullptr, (const int*)nullptr, n, sdf, feat, grad, w.spill,
                      &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, 0);
    else
        LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(g), dim3(kThreads),
                      kLdsSdfFwd, s, fd, x_norm, (const int*)nullptr, (const int*)nullptr, n, sdf, feat, (float*)nullptr,
                      (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr, 0);
    return check_launch();
}

// SDF on the N^3 lattice of [-1, 1]^3 (create_mesh_vertices_and_faces, utils/sdf_meshing.py:13-70: N = 256, the
// reference evaluates it in 64 chunks with a host copy each): one launch, coordinates formed in the kernel, the
// values stay on the device.  sdf[(ix * N + iy) * N + iz], normalised units.
int arah_sdf_grid(const ArahFrame* f, int32_t n_side, float* sdf, void* workspace, size_t wbytes, void* stream) {
    if (!f || !sdf || !workspace || n_side < 2 || n_side > 1024) return ARAH_E_BADARG;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const long long n = (long long)n_side * n_side * n_side;
    if (n > 0x7fffffffLL) return ARAH_E_BADARG;
    if (fd.split && knobs().density_wide && n >= 128ll * 1024)   // long lattices on the split engine: 128-point tiles, like the density pass
        hipLaunchKernelGGL(k_sdf_lattice, dim3(min(num_cus(), grid_for(n, 128))), dim3(kThreads), kLdsDensityWide, s, fd, (int)n_side,
                           (int)n, sdf, &w.ctr->n_sdf_fwd);
    else
        LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(grid_for(n, kTile)), dim3(kThreads),
                      kLdsSdfFwd, s, fd, (const float*)nullptr, (const int*)nullptr, (const int*)nullptr, (int)n, sdf,
                      (float*)nullptr, (float*)nullptr, (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr,
                      (int)n_side);
    return check_launch();
}

// ---- triangle rasteriser (pix_to_face of pytorch3d's MeshRasterizer as models/__init__.py:232-237,268-276 use it:
// one face per pixel, no blur, no culling).  tri [F][3][3] = (u, v, z) per corner in PIXEL coordinates (pixel (i, j)
// has its centre at u = j + 0.5, v = i + 0.5) and view-space depth z > 0.  zbuf [H*W] 64-bit keys
// (depth bits << 32 | face), pre-set to ~0 by the caller; the nearest covering face wins.
__global__ void k_raster(const float* __restrict__ tri, int n_faces, int H, int W, float z_near,
                         unsigned long long* __restrict__ zbuf) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_faces) return;
    const float* t = tri + (size_t)f * 9;
    const float x0 = t[0], y0 = t[1], z0 = t[2], x1 = t[3], y1 = t[4], z1 = t[5], x2 = t[6], y2 = t[7], z2 = t[8];
    if (!(z0 > z_near && z1 > z_near && z2 > z_near)) return;   // faces that reach the near plane are dropped (also NaNs)
    const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
    if (area == 0.f || !(area == area)) return;
    const float inv_area = 1.0f / area;
    int j0 = max(0, (int)floorf(fminf(x0, fminf(x1, x2)) - 0.5f)), j1 = min(W - 1, (int)ceilf(fmaxf(x0, fmaxf(x1, x2)) - 0.5f));
    int i0 = max(0, (int)floorf(fminf(y0, fminf(y1, y2)) - 0.5f)), i1 = min(H - 1, (int)ceilf(fmaxf(y0, fmaxf(y1, y2)) - 0.5f));
    if (j1 - j0 > 4096 || i1 - i0 > 4096) return;
    for (int i = i0; i <= i1; ++i)
        for (int j = j0; j <= j1; ++j) {
            const float px = (float)j + 0.5f, py = (float)i + 0.5f;
            const float w0 = ((x1 - px) * (y2 - py) - (x2 - px) * (y1 - py)) * inv_area;
            const float w1 = ((x2 - px) * (y0 - py) - (x0 - px) * (y2 - py)) * inv_area;
            const float w2 = 1.0f - w0 - w1;
            if (w0 < 0.f || w1 < 0.f || w2 < 0.f) continue;
            const float z = w0 * z0 + w1 * z1 + w2 * z2;
            const unsigned long long key = ((unsigned long long)__float_as_uint(z) << 32) | (unsigned)f;
            atomicMin(&zbuf[(size_t)i * W + j], key);
        }
}

int arah_rasterize(const float* tri_uvz, int32_t n_faces, int32_t H, int32_t W, float z_near, uint64_t* zbuf, void* stream) {
    if (n_faces < 0 || H <= 0 || W <= 0 || !zbuf) return ARAH_E_BADARG;
    if (n_faces == 0) return ARAH_OK;
    if (!tri_uvz) return ARAH_E_BADARG;
    hipLaunchKernelGGL(k_raster, dim3((n_faces + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), tri_uvz,
                       (int)n_faces, (int)H, (int)W, z_near, reinterpret_cast<unsigned long long*>(zbuf));
    return check_launch();
}

int arah_skin_lbs(const ArahFrame* f, const float* x_hat, int32_t n, float* wout, float* x_bar, float* T,
                  void* workspace, size_t wbytes, void* stream) {
    if (!f || !x_hat || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipLaunchKernelGGL(k_skin_eval, dim3(grid_for(n, kTile)), dim3(kThreads), kLdsSkin,
                       reinterpret_cast<hipStream_t>(stream), to_dev(*f), x_hat, n, wout, x_bar, T, &w.ctr->n_skin_fwd);
    return check_launch();
}

int arah_skin_lbs_counted(const ArahFrame* f, const float* x_hat, int32_t n_max, const int32_t* n_items, int32_t per_item,
                          float* x_bar, void* workspace, size_t wbytes, void* stream) {
    if (!f || !x_hat || !x_bar || !n_items || n_max < 0 || per_item <= 0 || !workspace) return ARAH_E_BADARG;
    if (n_max == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipLaunchKernelGGL(k_skin_eval, dim3(grid_for(n_max, kTile)), dim3(kThreads), kLdsSkin,
                       reinterpret_cast<hipStream_t>(stream), to_dev(*f), x_hat, n_max, (float*)nullptr, x_bar, (float*)nullptr,
                       &w.ctr->n_skin_fwd, n_items, per_item);
    return check_launch();
}

// ---- marching cubes (csrc/mcubes.hpp) ------------------------------------------------------------------
size_t arah_marching_cubes_scratch_bytes(int32_t n_side) {
    if (n_side < 2) return 0;
    return (size_t)(n_side - 1) * (n_side - 1) * 2 * sizeof(int);
}

int arah_marching_cubes(const float* sdf, int32_t n_side, float level, const int8_t* tri_table, const int32_t* n_tri,
                        float* tris, int32_t cap, int32_t* n_tris, void* scratch, size_t scratch_bytes, void* stream) {
    if (!sdf || !tri_table || !n_tri || !tris || !n_tris || !scratch || n_side < 2 || n_side > 1024 || cap <= 0) return ARAH_E_BADARG;
    if (scratch_bytes < arah_marching_cubes_scratch_bytes(n_side)) return ARAH_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = (n_side - 1) * (n_side - 1);
    int* row_count = reinterpret_cast<int*>(scratch);
    int* row_base = row_count + rows;
    const float vs = (float)(2.0 / (n_side - 1));   // sdf_meshing.py:27: voxel_size = 2.0 / (N - 1), rounded to fp32 once
    const signed char* table = reinterpret_cast<const signed char*>(tri_table);
    hipLaunchKernelGGL(k_mcubes<false>, dim3(rows), dim3(kMcThreads), 0, s, sdf, (int)n_side, level, vs, table, (const int*)n_tri,
                       row_count, (const int*)nullptr, (float*)nullptr, 0);
    hipLaunchKernelGGL(k_mc_scan, dim3(1), dim3(1024), 0, s, (const int*)row_count, rows, row_base, (int*)n_tris);
    hipLaunchKernelGGL(k_mcubes<true>, dim3(rows), dim3(kMcThreads), 0, s, sdf, (int)n_side, level, vs, table, (const int*)n_tri,
                       (int*)nullptr, (const int*)row_base, tris, (int)cap);
    hipLaunchKernelGGL(k_mc_pad, dim3(1024), dim3(256), 0, s, tris, (const int*)n_tris, (int)cap);
    return check_launch();
}

int arah_skin_jacobian(const ArahFrame* f, const float* x_hat, int32_t n, float* jac, void* workspace, size_t wbytes,
                       void* stream) {
    if (!f || !x_hat || !jac || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipLaunchKernelGGL(k_skin_jac, dim3(grid_for(n, 16)), dim3(kThreads), kLdsSkin,
                       reinterpret_cast<hipStream_t>(stream), to_dev(*f), x_hat, (const int*)nullptr,
                       (const int*)nullptr, n, jac, &w.ctr->n_skin_jac);
    return check_launch();
}

int arah_color_eval(const ArahFrame* f, const float* x_norm, const float* normal, const float* view,
                    const float* feat, int32_t n, float* rgb, void* workspace, size_t wbytes, void* stream) {
    if (!f || !x_norm || !normal || !feat || !rgb || n < 0 || !workspace) return ARAH_E_BADARG;
    if (f->col_mode == ARAH_COLOR_IDR && !view) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    if (f->col_mode == ARAH_COLOR_IDR)
        hipLaunchKernelGGL(k_color_eval<true>, dim3(grid_for(n, kTile)), dim3(kThreads), lds_color<true>(), s, fd,
                           x_norm, normal, view, feat, n, rgb, &w.ctr->n_col);
    else
        hipLaunchKernelGGL(k_color_eval<false>, dim3(grid_for(n, kTile)), dim3(kThreads), lds_color<false>(), s, fd,
                           x_norm, normal, view, feat, n, rgb, &w.ctr->n_col);
    return check_launch();
}

int arah_nearest_inverse_lbs(const ArahFrame* f, const float* pts, int32_t n, int32_t* idx, float* x_hat0, float* T0,
                             void* workspace, size_t wbytes, void* stream) {
    if (!f || !pts || !x_hat0 || !T0 || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    RaySet rs = make_rays(nullptr, nullptr, 1);
    launch_nearest<SRC_POINTS>(reinterpret_cast<hipStream_t>(stream), to_dev(*f), n, pts, rs, (const float*)nullptr, 1,
                               (const int*)nullptr, (const int*)nullptr, n, idx, x_hat0, T0, 0,
                               &w.ctr->n_knn);
    return check_launch();
}

// shared driver of loop C.  The start states of the ids in w.listA[0 .. cnt[0]) sit in outp.pts (x0) and outp.T (T0, with
// the target in row 3 -- written there by k_nearest_invlbs<SAMPLES>, or here from tgt); the results replace them.
// w.counts: [0] = number of entries, [1] = head of the queue (zeroed by the caller's memset of w.counts).
static int run_broyden3(const FrameDev& fd, Workspace& w, const float* tgt, CanonOut outp, long long max_pts,
                        hipStream_t s, int mode_arg = ARAH_CANON_KERNEL_WAVE, void* const* ev = nullptr,
                        const int* list_arg = nullptr, int* cnt_arg = nullptr) {
    // the list and its two device-side counts {entries, head of the queue}: w.listA / w.counts unless the caller (the tiered
    // forward) brings its own
    int* cnt = cnt_arg ? cnt_arg : w.counts;
    const int* const list = list_arg ? list_arg : w.listA;
    if (tgt)
        hipLaunchKernelGGL(k_canon_seed, dim3(grid_for(max_pts, 256)), dim3(256), 0, s, list,
                           (const int*)&cnt[0], tgt, outp.T);
    // mode: ARAH_CANON_KERNEL_WAVE (default: point-owning waves, hi fragments in LDS), _WAVE_L2 (all fragments from L2), _TILE
    // (round 2's channel-sliced tiles).  The exact engine always runs the tile kernel.  A field of the call
    // (ArahSampling::canon_kernel), like the profiling events recorded around the launch: nothing here is process-global.
    const int mode = mode_arg == ARAH_CANON_KERNEL_TILE ? 0 : (mode_arg == ARAH_CANON_KERNEL_WAVE_L2 ? 2 : 1);
    hipEvent_t ev0 = ev ? reinterpret_cast<hipEvent_t>(ev[0]) : nullptr, ev1 = ev ? reinterpret_cast<hipEvent_t>(ev[1]) : nullptr;
    if (ev0 && ev1) hipEventRecord(ev0, s);
    unsigned long long* const clk_arg = w.ctr->clk;
    if (fd.split && mode != 0) {
        long long gw = (max_pts + kCwWaves * kCwSlots - 1) / (kCwWaves * kCwSlots);
        int cus = num_cus() * knobs().canon_wg_per_cu;
        if (list_arg && knobs().canon_tier_wgs > 0) cus = min(cus, knobs().canon_tier_wgs);   // the tiers' short lists (see Knobs)
        if (gw > cus) gw = cus;
        if (gw < 1) gw = 1;
        // two instances are launched, one returns at once: whether the activations of this frame's skinning MLP need scaling
        // down to stay inside the f16 range is known on the device only (k_skin_probe), and the plain instance is the faster
        if (mode == 1) {
            hipLaunchKernelGGL((k_canon_wave<true, false>), dim3((int)gw), dim3(kCwThreads), kLdsCanonWave, s, fd, list,
                               (const int*)&cnt[0], &cnt[1], outp, &w.ctr->n_skin_fwd, &w.ctr->n_canon,
                               &w.ctr->n_split_nonfinite, clk_arg);
            hipLaunchKernelGGL((k_canon_wave<true, true>), dim3((int)gw), dim3(kCwThreads), kLdsCanonWave, s, fd, list,
                               (const int*)&cnt[0], &cnt[1], outp, &w.ctr->n_skin_fwd, &w.ctr->n_canon,
                               &w.ctr->n_split_nonfinite, clk_arg);
        } else {
            const size_t lds_l2 = max(kLdsCanonWave - kCwHiBytes, (size_t)knobs().canon_lds_min);
            hipLaunchKernelGGL((k_canon_wave<false, false>), dim3((int)gw), dim3(kCwThreads), lds_l2, s, fd,
                               list, (const int*)&cnt[0], &cnt[1], outp, &w.ctr->n_skin_fwd,
                               &w.ctr->n_canon, &w.ctr->n_split_nonfinite, clk_arg);
            hipLaunchKernelGGL((k_canon_wave<false, true>), dim3((int)gw), dim3(kCwThreads), lds_l2, s, fd,
                               list, (const int*)&cnt[0], &cnt[1], outp, &w.ctr->n_skin_fwd,
                               &w.ctr->n_canon, &w.ctr->n_split_nonfinite, clk_arg);
        }
    } else {
        LAUNCH_ENGINE(fd.split, k_canon_solve<true>, k_canon_solve<false>, dim3(grid_for(max_pts, kTile)), dim3(kThreads),
                      kLdsCanonSolve, s, fd, list, (const int*)&cnt[0], &cnt[1], outp,
                      &w.ctr->n_skin_fwd, &w.ctr->n_canon, w.ctr->clk);
    }
    if (ev0 && ev1) hipEventRecord(ev1, s);
    return check_launch();
}

int arah_broyden3_lbs(const ArahFrame* f, const float* tgt, const float* x0, const float* T0, int32_t n, float* x,
                      float* T, float* err, uint8_t* conv, int32_t canon_kernel, void* workspace, size_t wbytes, void* stream) {
    if (!f || !tgt || !x0 || !T0 || !x || !T || !conv || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, n, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    hipMemcpyAsync(x, x0, (size_t)n * 3 * 4, hipMemcpyDeviceToDevice, s);
    hipMemcpyAsync(T, T0, (size_t)n * 16 * 4, hipMemcpyDeviceToDevice, s);
    hipMemsetAsync(w.counts, 0, sizeof(int) * 3 * kNumCounts, s);
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, n, w.listA, &w.counts[0]);
    int rc = run_broyden3(fd, w, tgt, CanonOut{x, T, w.q_err}, n, s, canon_kernel);
    if (rc) return rc;
    hipLaunchKernelGGL(k_broyden3_finalize, dim3((n + 255) / 256), dim3(256), 0, s, n, (const float*)w.q_err, err, conv);
    return check_launch();
}

// ---- loop B on prepared starts ------------------------------------------------------------------
// Expects, for every ray id in w.listA[0 .. cntB[0]): x0raw (raw canonical start), xcur (the same point normalised),
// t (start depth); and for EVERY ray the best-iterate arrays seeded with the start (xbest_ray, zbest_ray, T, err_best
// = huge).  Leaves the best iterates there (RFU:365-484, broyden.py:4-78).
static void joint_impl(const FrameDev& fd, Workspace& w, const RaySet& rs, int n, float* T, hipStream_t s) {
    int* cntB = w.counts + kNumCounts;
    const int gm = grid_for(n, kTile);
    hipLaunchKernelGGL(k_skin_jac, dim3(grid_for(n, 16)), dim3(kThreads), kLdsSkin, s, fd, (const float*)w.x0raw,
                       (const int*)w.listA, (const int*)&cntB[0], 0, w.jac_lbs, &w.ctr->n_skin_jac);
    // d sdf / d x at the normalised start point == d(metric sdf)/d(metric x)  (RFU:408-413)
    LAUNCH_ENGINE(fd.split, (k_sdf_eval<true, true>), (k_sdf_eval<true, false>), dim3(gm), dim3(kThreads), kLdsSdfGrad, s,
                  fd, (const float*)w.xcur, (const int*)w.listA, (const int*)&cntB[0], 0, w.u_gx /*scratch: sdf*/,
                  (float*)nullptr, w.grad_sdf, w.spill, &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, 0);
    Broyden4State st{w.u_eval, w.u_step, w.u_gx, w.u_Jinv, w.err_best_ray, w.xbest_ray, w.zbest_ray, T};
    hipLaunchKernelGGL(k_joint_init, dim3(grid_for(n, 256)), dim3(256), 0, s, fd, st, rs, (const int*)w.listA,
                       (const int*)&cntB[0], (const float*)w.grad_sdf, (const float*)w.jac_lbs, (const float*)w.xcur,
                       (const float*)w.t, w.x0raw);
    // the first iterations as one launch each over the compacted list; then the finisher adopts what is left
    // (ARAH_JOINT_BULK_ITERS >= 51: every iteration as its own launch, round 2's schedule)
    const int bulk = knobs().joint_bulk;
    for (int it = 0; it < bulk; ++it) {
        int* lin = (it & 1) ? w.listB : w.listA;
        int* lout = (it & 1) ? w.listA : w.listB;
        if (it == 0)
            LAUNCH_ENGINE(fd.split, (k_joint_iter<true, true>), (k_joint_iter<true, false>), dim3(gm), dim3(kThreads),
                          kLdsJoint, s, fd, st, rs, (const int*)lin, (const int*)&cntB[it], lout, &cntB[it + 1],
                          &w.ctr->n_skin_fwd, &w.ctr->n_sdf_fwd);
        else
            LAUNCH_ENGINE(fd.split, (k_joint_iter<false, true>), (k_joint_iter<false, false>), dim3(gm), dim3(kThreads),
                          kLdsJoint, s, fd, st, rs, (const int*)lin, (const int*)&cntB[it], lout, &cntB[it + 1],
                          &w.ctr->n_skin_fwd, &w.ctr->n_sdf_fwd);
    }
    if (bulk <= kBroydenSteps) {
        const int* lin = (bulk & 1) ? w.listB : w.listA;
        const int gf = min(4 * num_cus(), max(1, (n + kFinTile - 1) / kFinTile));
        LAUNCH_ENGINE(fd.split, k_joint_finish<true>, k_joint_finish<false>, dim3(gf), dim3(kThreads), kLdsJointFinish, s, fd,
                      st, rs, lin, (const int*)&cntB[bulk], kBroydenSteps + 1 - bulk, &w.ctr->n_skin_fwd, &w.ctr->n_sdf_fwd);
    }
}

// ---- loops A + B -----------------------------------------------------------------------------
static int trace_impl(const ArahFrame* f, Workspace& w, const float* cam_loc, int32_t rays_per_cam,
                      const float* dirs, const float* near_far, int32_t n, int root_find_all, float* points_hat_norm,
                      float* T, uint8_t* conv, float* start, float* end, hipStream_t s, const uint8_t* skip = nullptr) {
    const FrameDev fd = to_dev(*f);
    const RaySet rs = make_rays(cam_loc, dirs, rays_per_cam);
    const int gb = (n + 255) / 256;
    int* cntA = w.counts;                 // sphere tracing: cntA[it]
    int* cntB = w.counts + kNumCounts;    // joint root find
    hipMemsetAsync(w.counts, 0, sizeof(int) * 3 * kNumCounts, s);
    hipMemsetAsync(w.nn_idx, 0xff, sizeof(int) * (size_t)n, s);   // -1: no previous nearest vertex yet
    hipLaunchKernelGGL(k_trace_begin, dim3(gb), dim3(256), 0, s, near_far, n, w.t, w.far, w.diverged, w.xcur, w.Tcur,
                       w.listA, &cntA[0], skip);
    TraceState ts{w.t, w.far, w.xcur, w.diverged};
    const int gm = grid_for(n, kTile);
    // the first steps as launches over the compacted list; after ARAH_TRACE_BULK_STEPS of them the resident finisher takes the
    // list (finish.hpp; 50 = never)
    // Short ray lists (a training view: 2048 rays) never fill a launch: there every step is two kernel latencies, and the
    // finisher (one launch, sixteen lanes per nearest-vertex search) takes the whole loop (ARAH_TRACE_SMALL: below how
    // many rays; 0 = never)
    // Full frames: 24 wide steps, then the finisher adopts the stragglers (a few thousand rays: 52 launches of tens of
    // microseconds each become one; 43.15 -> 42.75 ms per frame one at a time, 41.7 -> 41.0 with three in flight; handing
    // over after 8 or 12 steps loses: too many rays left for sixteen-ray tiles -- profiles/r03_ab_trace_finish.txt)
    const int bulk = n <= knobs().trace_small ? 0 : knobs().trace_bulk;
    for (int it = 0; it < bulk; ++it) {
        int* lin = (it & 1) ? w.listB : w.listA;
        int* lout = (it & 1) ? w.listA : w.listB;
        launch_nearest<SRC_RAYS>(s, fd, n, (const float*)nullptr, rs, (const float*)w.t, 1, (const int*)lin,
                                 (const int*)&cntA[it], 0, w.nn_idx, w.xcur, w.Tcur, 0, &w.ctr->n_knn);
        LAUNCH_ENGINE(fd.split, k_sdf_march<true>, k_sdf_march<false>, dim3(gm), dim3(kThreads), kLdsSdfFwd, s, fd, ts,
                      (const int*)lin, (const int*)&cntA[it], lout, &cntA[it + 1], &w.ctr->n_sdf_fwd);
    }
    if (bulk < kSphereIters) {
        const int* lin = (bulk & 1) ? w.listB : w.listA;
        const int gf = min(4 * num_cus(), max(1, (n + kFinTile - 1) / kFinTile));
        LAUNCH_ENGINE(fd.split, k_trace_finish<true>, k_trace_finish<false>, dim3(gf), dim3(kThreads), kLdsTraceFinish, s, fd,
                      knn_of(fd), rs, ts, w.Tcur, w.nn_idx, lin, (const int*)&cntA[bulk], kSphereIters - bulk, &w.ctr->n_knn,
                      &w.ctr->n_sdf_fwd);
    }
    // joint root find on the non-diverged rays; best-iterate arrays: x -> xbest_ray, depth -> zbest_ray, T -> T (output)
    hipLaunchKernelGGL(k_joint_select, dim3(gb), dim3(256), 0, s, fd, n, (const float*)w.xcur, (const float*)w.Tcur,
                       (const float*)w.t, (const uint8_t*)w.diverged, root_find_all, w.x0raw, w.xbest_ray, w.zbest_ray, T,
                       w.err_best_ray, w.listA, &cntB[0]);
    joint_impl(fd, w, rs, n, T, s);
    hipLaunchKernelGGL(k_trace_finalize, dim3(gb), dim3(256), 0, s, fd, n, near_far, (const float*)w.xbest_ray,
                       (const float*)w.zbest_ray, (const float*)w.err_best_ray, points_hat_norm, conv, start, end);
    return check_launch();
}

int arah_trace(const ArahFrame* f, const float* cam_loc, int32_t rays_per_cam, const float* dirs,
               const float* near_far, int32_t n, int32_t root_find_all, float* points_hat_norm, float* T,
               uint8_t* conv, float* start, float* end, void* workspace, size_t wbytes, void* stream) {
    if (!f || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    if (!cam_loc || !dirs || !near_far || !points_hat_norm || !T || !conv || !start || !end) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    return trace_impl(f, w, cam_loc, rays_per_cam, dirs, near_far, n, root_find_all, points_hat_norm, T, conv, start,
                      end, reinterpret_cast<hipStream_t>(stream));
}

int arah_joint_root_find(const ArahFrame* f, const float* cam_loc, int32_t rays_per_cam, const float* dirs,
                         const uint8_t* valid, const float* x0, const float* z0, const float* T0, int32_t n, float* x,
                         float* z, float* T, uint8_t* conv, void* workspace, size_t wbytes, void* stream) {
    if (!f || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    if (!cam_loc || !dirs || !valid || !x0 || !z0 || !T0 || !x || !z || !T || !conv) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const RaySet rs = make_rays(cam_loc, dirs, rays_per_cam);
    const int gb = (n + 255) / 256;
    hipMemsetAsync(w.counts, 0, sizeof(int) * 3 * kNumCounts, s);
    hipLaunchKernelGGL(k_joint_seed, dim3(gb), dim3(256), 0, s, fd, n, x0, z0, T0, valid, w.x0raw, w.xcur, w.t, w.xbest_ray,
                       w.zbest_ray, T, w.err_best_ray, w.listA, &w.counts[kNumCounts]);
    joint_impl(fd, w, rs, n, T, s);
    hipLaunchKernelGGL(k_joint_seam_out, dim3(gb), dim3(256), 0, s, n, (const float*)w.xbest_ray, (const float*)w.zbest_ray,
                       (const float*)w.err_best_ray, valid, x, z, conv);
    return check_launch();
}

// ---- sampler + loop C -------------------------------------------------------------------------
static int sample_impl(const ArahFrame* f, const ArahSampling* cfg, Workspace& w, const float* cam_loc,
                       int32_t rays_per_cam, const float* dirs, const float* near_far, const uint8_t* conv,
                       const float* start, const float* end, int32_t n, const float* rand_s, const float* rand_near,
                       const float* rand_far, float* z, float* pts, float* T, uint8_t* mask, hipStream_t s) {
    const int S = cfg->n_steps;
    const FrameDev fd = to_dev(*f);
    const RaySet rs = make_rays(cam_loc, dirs, rays_per_cam);
    const long long Q = (long long)n * S;
    hipMemsetAsync(w.counts, 0, sizeof(int) * 3 * kNumCounts, s);
    hipLaunchKernelGGL(k_sample_depths, dim3((n + 127) / 128), dim3(128), 0, s, n, S, cfg->n_near, cfg->n_far, near_far,
                       conv, start, end, cfg->lin_steps, cfg->lin_near, cfg->lin_far, rand_s, rand_near, rand_far, z, w.q_smask);
    const int gq = (int)((Q + 255) / 256);
    hipLaunchKernelGGL(k_build_list, dim3((int)((Q + 4095) / 4096)), dim3(1024), 0, s, (const uint8_t*)w.q_smask, (int)Q, w.listA, &w.counts[0]);
    // x0 -> pts (raw canonical, doubles as x_best), T0 -> T (doubles as T_best)
    launch_nearest<SRC_SAMPLES>(s, fd, Q, (const float*)nullptr, rs, (const float*)z, S, (const int*)w.listA,
                                (const int*)&w.counts[0], 0, (int*)nullptr, pts, T, 1, &w.ctr->n_knn);
    int rc = run_broyden3(fd, w, nullptr, CanonOut{pts, T, w.q_err}, Q, s, cfg->canon_kernel, cfg->ev_canon);
    if (rc) return rc;
    hipLaunchKernelGGL(k_canon_finalize, dim3(gq), dim3(256), 0, s, fd, (int)Q, (const uint8_t*)w.q_smask,
                       (const float*)w.q_err, pts, T, mask);
    return check_launch();
}

static int check_sampling(const ArahSampling* cfg) {
    const int S = cfg->n_steps;
    if (S <= 0 || S > ARAH_MAX_STEPS || cfg->n_near < 0 || cfg->n_far < 0 || S < cfg->n_near + cfg->n_far + 1)
        return ARAH_E_SAMPLING;
    if (!cfg->lin_steps || !cfg->lin_near || (cfg->n_far > 0 && !cfg->lin_far)) return ARAH_E_BADARG;
    return ARAH_OK;
}

int arah_sample_canonicalize(const ArahFrame* f, const ArahSampling* cfg, const float* cam_loc, int32_t rays_per_cam,
                             const float* dirs, const float* near_far, const uint8_t* conv, const float* start,
                             const float* end, int32_t n, const float* rand_steps, const float* rand_near,
                             const float* rand_far, float* z, float* pts, float* T, uint8_t* mask, void* workspace,
                             size_t wbytes, void* stream) {
    if (!f || !cfg || n < 0 || !workspace) return ARAH_E_BADARG;
    int rc = check_sampling(cfg);
    if (rc) return rc;
    if (n == 0) return ARAH_OK;
    if (!cam_loc || !dirs || !near_far || !conv || !start || !end || !z || !pts || !T || !mask) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n, cfg->n_steps);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    if ((rand_steps || rand_near || rand_far) && (!rand_steps || !rand_near || (cfg->n_far > 0 && !rand_far)))
        return ARAH_E_BADARG;   // jitter is all or nothing
    return sample_impl(f, cfg, w, cam_loc, rays_per_cam, dirs, near_far, conv, start, end, n, rand_steps, rand_near,
                       rand_far, z, pts, T, mask, reinterpret_cast<hipStream_t>(stream));
}

#ifdef ARAH_REG_TRUNK
// phase clocks of the point-owning trunk (instrumented builds, -DRT_CLOCKS: tools/probes/density_probe.py clocks)
static unsigned long long* rt_clk_of(const Workspace& w) {
#ifdef RT_CLOCKS
    return w.ctr->clk;
#else
    (void)w;
    return nullptr;
#endif
}
#endif

// the density pre-pass of lazy shading over list[0 .. *count): sigma of every listed sample -> w.shaded, the samples with
// sigma > 0 appended to next_list
static void launch_density(const FrameDev& fd, Workspace& w, const float* pts, long long Q, const int* list, const int* count,
                           int* next_list, int* next_count, hipStream_t s) {
    const int g = grid_for(Q, kTile);
    // long lists on the split engine: 128-point tiles, one workgroup per CU (ARAH_DENSITY_TILE=64 keeps the 64-point kernel)
#ifdef ARAH_REG_TRUNK
    if (fd.split && knobs().density_reg && Q >= 128ll * 1024)
        hipLaunchKernelGGL(k_density_reg, dim3(min(num_cus(), grid_for(Q, kRtTile))), dim3(kRtThreads), kLdsRegTrunk, s, fd,
                           pts, list, count, w.shaded, next_list, next_count, &w.ctr->n_sdf_fwd, &w.ctr->n_density, rt_clk_of(w));
    else
#endif
    if (fd.split && knobs().density_wide && Q >= 128ll * 1024)
        hipLaunchKernelGGL((k_density<true, 8>), dim3(min(num_cus(), grid_for(Q, 128))), dim3(kThreads), kLdsDensityWide, s, fd,
                           pts, list, count, w.shaded, next_list, next_count, &w.ctr->n_sdf_fwd, &w.ctr->n_density);
    else
        LAUNCH_ENGINE(fd.split, k_density<true>, k_density<false>, dim3(g), dim3(kThreads), kLdsSdfFwd, s, fd, pts, list, count,
                      w.shaded, next_list, next_count, &w.ctr->n_sdf_fwd, &w.ctr->n_density);
}

static int shade_tail(const ArahFrame* f, const ArahSampling* cfg, Workspace& w, const FrameDev& fd, const float* dirs,
                      const float* z, const float* pts, const float* T, const uint8_t* mask, int32_t n, const int* slist,
                      const int* scount, float* rgb, float* acc, uint8_t* vol_mask, hipStream_t s);

// ---- loop D -----------------------------------------------------------------------------------
static int shade_impl(const ArahFrame* f, const ArahSampling* cfg, Workspace& w, const float* dirs, const float* z,
                      const float* pts, const float* T, const uint8_t* mask, int32_t n, float* rgb, float* acc,
                      uint8_t* vol_mask, hipStream_t s) {
    const int S = cfg->n_steps;
    const FrameDev fd = to_dev(*f);
    const long long Q = (long long)n * S;
    hipMemsetAsync(w.counts, 0, sizeof(int) * 3 * kNumCounts, s);
    hipLaunchKernelGGL(k_build_list, dim3((int)((Q + 4095) / 4096)), dim3(1024), 0, s, mask, (int)Q, w.listA, &w.counts[0]);
    const int g = grid_for(Q, kTile);
    const int* slist = w.listA;
    const int* scount = &w.counts[0];
    if (!cfg->full_shading) {   // pass 1: densities; only samples that can receive weight reach k_shade
        if (cfg->ev_density[0] && cfg->ev_density[1]) hipEventRecord(reinterpret_cast<hipEvent_t>(cfg->ev_density[0]), s);
        launch_density(fd, w, pts, Q, (const int*)w.listA, (const int*)&w.counts[0], w.listB, &w.counts[1], s);
        if (cfg->ev_density[0] && cfg->ev_density[1]) hipEventRecord(reinterpret_cast<hipEvent_t>(cfg->ev_density[1]), s);
        slist = w.listB;
        scount = &w.counts[1];
    }
    return shade_tail(f, cfg, w, fd, dirs, z, pts, T, mask, n, slist, scount, rgb, acc, vol_mask, s);
}

// normal + colour of the samples in slist[0 .. *scount), then the per-ray compositing
static int shade_tail(const ArahFrame* f, const ArahSampling* cfg, Workspace& w, const FrameDev& fd, const float* dirs,
                      const float* z, const float* pts, const float* T, const uint8_t* mask, int32_t n, const int* slist,
                      const int* scount, float* rgb, float* acc, uint8_t* vol_mask, hipStream_t s) {
    const int S = cfg->n_steps;
    const long long Q = (long long)n * S;
    const int g = grid_for(Q, kTile);
    if (cfg->ev_shade[0] && cfg->ev_shade[1]) hipEventRecord(reinterpret_cast<hipEvent_t>(cfg->ev_shade[0]), s);
    const B3Nets b3 = b3_of(*f);
    if (fd.split && shade_b3(cfg->shade_engine)) {
        if (f->col_mode == ARAH_COLOR_IDR)
            hipLaunchKernelGGL((k_shade<true, true, true>), dim3(g), dim3(kThreads), split_lds(lds_shade_b3<true>()), s, fd, S,
                               cfg->cano_view_dirs, dirs, pts, T, slist, scount, 0, w.shaded, w.spill, &w.ctr->n_sdf_fwd,
                               &w.ctr->n_sdf_grad, &w.ctr->n_col, b3);
        else
            hipLaunchKernelGGL((k_shade<false, true, true>), dim3(g), dim3(kThreads), split_lds(lds_shade_b3<false>()), s, fd, S,
                               cfg->cano_view_dirs, dirs, pts, T, slist, scount, 0, w.shaded, w.spill, &w.ctr->n_sdf_fwd,
                               &w.ctr->n_sdf_grad, &w.ctr->n_col, b3);
    } else if (f->col_mode == ARAH_COLOR_IDR)
        LAUNCH_ENGINE(fd.split, (k_shade<true, true>), (k_shade<true, false>), dim3(g), dim3(kThreads), lds_shade<true>(),
                      s, fd, S, cfg->cano_view_dirs, dirs, pts, T, slist, scount, 0, w.shaded, w.spill, &w.ctr->n_sdf_fwd,
                      &w.ctr->n_sdf_grad, &w.ctr->n_col, b3);
    else
        LAUNCH_ENGINE(fd.split, (k_shade<false, true>), (k_shade<false, false>), dim3(g), dim3(kThreads),
                      lds_shade<false>(), s, fd, S, cfg->cano_view_dirs, dirs, pts, T, slist, scount, 0, w.shaded, w.spill,
                      &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, &w.ctr->n_col, b3);
    if (cfg->ev_shade[0] && cfg->ev_shade[1]) hipEventRecord(reinterpret_cast<hipEvent_t>(cfg->ev_shade[1]), s);
    const bool chunk = S % 8 == 0 && (reinterpret_cast<uintptr_t>(z) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 7) == 0;
    if (chunk) hipLaunchKernelGGL(k_composite<true>, dim3((n + 127) / 128), dim3(128), 0, s, n, S, cfg->render_last_pt, z, mask,
                       (const f32x4*)w.shaded, rgb, acc, vol_mask);
    else hipLaunchKernelGGL(k_composite<false>, dim3((n + 127) / 128), dim3(128), 0, s, n, S, cfg->render_last_pt, z, mask,
                       (const f32x4*)w.shaded, rgb, acc, vol_mask);
    return check_launch();
}

int arah_shade_points(const ArahFrame* f, const float* x_norm, const float* T, const float* dirs, int32_t n,
                      int32_t cano_view_dirs, int32_t shade_engine, float* rgbs, float* sdfn, void* workspace, size_t wbytes,
                      void* stream) {
    if (!f || !x_norm || !T || !dirs || !rgbs || !sdfn || n < 0 || !workspace) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    Workspace w = carve(workspace, n, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const B3Nets b3 = b3_of(*f);
    const int g = grid_for(n, kTile);
    f32x4* out = reinterpret_cast<f32x4*>(rgbs);
    f32x4* dbg = reinterpret_cast<f32x4*>(sdfn);
    // the kernels of shade_impl, one sample per "ray" (S = 1: dirs are per point), no list
    if (fd.split && shade_b3(shade_engine)) {
        if (f->col_mode == ARAH_COLOR_IDR)
            hipLaunchKernelGGL((k_shade<true, true, true>), dim3(g), dim3(kThreads), split_lds(lds_shade_b3<true>()), s, fd, 1,
                               cano_view_dirs, dirs, x_norm, T, (const int*)nullptr, (const int*)nullptr, n, out, w.spill,
                               &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, &w.ctr->n_col, b3, dbg);
        else
            hipLaunchKernelGGL((k_shade<false, true, true>), dim3(g), dim3(kThreads), split_lds(lds_shade_b3<false>()), s, fd, 1,
                               cano_view_dirs, dirs, x_norm, T, (const int*)nullptr, (const int*)nullptr, n, out, w.spill,
                               &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, &w.ctr->n_col, b3, dbg);
    } else if (f->col_mode == ARAH_COLOR_IDR)
        LAUNCH_ENGINE(fd.split, (k_shade<true, true>), (k_shade<true, false>), dim3(g), dim3(kThreads), lds_shade<true>(),
                      s, fd, 1, cano_view_dirs, dirs, x_norm, T, (const int*)nullptr, (const int*)nullptr, n, out, w.spill,
                      &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, &w.ctr->n_col, b3, dbg);
    else
        LAUNCH_ENGINE(fd.split, (k_shade<false, true>), (k_shade<false, false>), dim3(g), dim3(kThreads),
                      lds_shade<false>(), s, fd, 1, cano_view_dirs, dirs, x_norm, T, (const int*)nullptr, (const int*)nullptr, n,
                      out, w.spill, &w.ctr->n_sdf_fwd, &w.ctr->n_sdf_grad, &w.ctr->n_col, b3, dbg);
    return check_launch();
}

int arah_shade_composite(const ArahFrame* f, const ArahSampling* cfg, const float* dirs, const float* z,
                         const float* pts, const float* T, const uint8_t* mask, int32_t n, float* rgb, float* acc,
                         uint8_t* vol_mask, void* workspace, size_t wbytes, void* stream) {
    if (!f || !cfg || n < 0 || !workspace) return ARAH_E_BADARG;
    if (cfg->n_steps <= 0 || cfg->n_steps > ARAH_MAX_STEPS) return ARAH_E_SAMPLING;
    if (n == 0) return ARAH_OK;
    if (!dirs || !z || !pts || !T || !mask || !rgb || !vol_mask) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n, cfg->n_steps);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    return shade_impl(f, cfg, w, dirs, z, pts, T, mask, n, rgb, acc, vol_mask, reinterpret_cast<hipStream_t>(stream));
}

// ---- loop D with gradients (training) -------------------------------------------------------------
static ColNetT colT_of(const ArahFrame& f) {
    return ColNetT{f.col_w0pT, f.col_w1pT, f.col_w2pT, f.col_w3apT, f.col_w3bpT, f.col_w4pT};
}

// Engine of the training kernels: bf16 x 3 backward-direction products and the f16 split forward trunk on a split-engine
// frame; a frame prepared for ARAH_PRECISION_FP32 -- the caller's choice, or the renderer's after its range guard fired --
// gets the all-fp32 kernels in training as in inference (round 4: it used to follow the environment only).
// ARAH_TRAIN_ENGINE=fp32 (read once, immutable) forces them for split frames too.
static bool train_b3(const ArahFrame& f) {
    return knobs().train_b3 && f.precision == ARAH_PRECISION_SPLIT_F16;
}

size_t arah_shade_train_slab_bytes(void) { return (size_t)kMaxGrid / 2 * kTrainSlabPerWg * sizeof(f32x4); }

int arah_shade_train_forward(const ArahFrame* f, const ArahTrainIn* in, float* sdf, float* rgb4, void* workspace,
                             size_t wbytes, void* stream) {
    if (!f || !in || !sdf || !rgb4 || !workspace || in->n < 0) return ARAH_E_BADARG;
    if (in->n == 0) return ARAH_OK;
    if (!in->x) return ARAH_E_BADARG;
    if (!in->geom_only && (!in->view || (in->rotate_normal && !in->T) || (in->ray_augm && !in->view_orig))) return ARAH_E_BADARG;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    TrainIn ti{in->n, in->x, in->T, in->view, in->view_orig, in->rotate_normal, in->ray_augm, nullptr, nullptr};
    ti.geom_only = in->geom_only ? 1 : 0;
    ti.tap_cin = in->tap_cin;
    for (int l = 0; l < 5; ++l) {
        if (in->tap_cin && !in->tap_c[l]) return ARAH_E_BADARG;
        ti.tap_c[l] = in->tap_cin ? in->tap_c[l] : nullptr;
    }
    ti.fwd_rgb = nullptr;
    TrainOut to;
    memset(&to, 0, sizeof(to));
    to.sdf = sdf;
    to.rgb = rgb4;
    const int g = min(grid_for(in->n, kTile), kMaxGrid / 2);   // one workgroup per CU (140 KB of LDS)
    const bool idr = f->col_mode == ARAH_COLOR_IDR;
#define ARAH_LAUNCH_TRAIN(IDR_, BWD_, B3_, SLAB_)                                                                        \
    hipLaunchKernelGGL((k_shade_train<IDR_, BWD_, B3_>), dim3(g), dim3(kThreads), lds_shade_train<IDR_>(), s,            \
                       TrainArgs{fd, colT_of(*f), b3_of(*f), ti, to, w.spill, SLAB_})
    if (train_b3(*f)) {
        if (idr) ARAH_LAUNCH_TRAIN(true, false, true, (f32x4*)nullptr);
        else ARAH_LAUNCH_TRAIN(false, false, true, (f32x4*)nullptr);
    } else {
        if (idr) ARAH_LAUNCH_TRAIN(true, false, false, (f32x4*)nullptr);
        else ARAH_LAUNCH_TRAIN(false, false, false, (f32x4*)nullptr);
    }
    return check_launch();
}

int arah_shade_train_backward(const ArahFrame* f, const ArahTrainIn* in, const ArahTrainGrads* gr, void* slab,
                              size_t slab_bytes, void* workspace, size_t wbytes, void* stream) {
    if (!f || !in || !gr || !slab || !workspace || in->n < 0) return ARAH_E_BADARG;
    if (in->n == 0) return ARAH_OK;
    if (!in->x || !in->g_s || !in->g_rgb) return ARAH_E_BADARG;
    if (!in->geom_only && (!in->view || (in->rotate_normal && !in->T) || (in->ray_augm && !in->view_orig))) return ARAH_E_BADARG;
    if (slab_bytes < arah_shade_train_slab_bytes()) return ARAH_E_WORKSPACE;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    TrainIn ti{in->n, in->x, in->T, in->view, in->view_orig, in->rotate_normal, in->ray_augm, in->g_s, in->g_rgb};
    // hand-over from the forward call: its streams must be the ones the weight-gradient GEMMs will read
    ti.geom_only = in->geom_only ? 1 : 0;
    const bool handed = !in->geom_only && in->tap_cin && in->fwd_rgb4;
    if (handed) {
        if (in->tap_cin != gr->cin) return ARAH_E_BADARG;
        for (int l = 0; l < 5; ++l)
            if (!in->tap_c[l] || in->tap_c[l] != gr->c[l]) return ARAH_E_BADARG;
    }
    ti.tap_cin = handed ? in->tap_cin : nullptr;
    for (int l = 0; l < 5; ++l) ti.tap_c[l] = handed ? in->tap_c[l] : nullptr;
    ti.fwd_rgb = handed ? in->fwd_rgb4 : nullptr;
    TrainOut to;
    to.sdf = gr->sdf;
    to.rgb = gr->rgb4;
    to.gx = gr->gx4;
    to.film_f = gr->film_freq;
    to.film_p = gr->film_phase;
    for (int k = 0; k < 6; ++k) {
        to.h[k] = gr->h[k];
        to.av[k] = gr->av[k];
        to.avd[k] = gr->avd[k];
        to.d[k] = gr->d[k];
    }
    for (int k = 0; k < 7; ++k) to.hd[k] = gr->hd[k];
    for (int k = 0; k < 5; ++k) to.c[k] = gr->c[k];
    to.cin = gr->cin;
    if (hipMemsetAsync(gr->film_freq, 0, 6 * 256 * 4, s) != hipSuccess) return ARAH_E_LAUNCH;
    if (hipMemsetAsync(gr->film_phase, 0, 6 * 256 * 4, s) != hipSuccess) return ARAH_E_LAUNCH;
    const int g = min(grid_for(in->n, kTile), kMaxGrid / 2);
    const bool idr = f->col_mode == ARAH_COLOR_IDR;
    f32x4* slab4 = reinterpret_cast<f32x4*>(slab);
    if (train_b3(*f)) {
        // bf16 hi/lo fragments of the matrices of the backward-direction products (W and W^T of the SDF trunk, W^T of the
        // colour MLP), made here from the frame's fp32 packings: sixteen small launches per training step, none in inference
        const int kc0 = (idr ? ColDims<true>::kInPad : ColDims<false>::kInPad) / 16;
        const float* src[kB3Count] = {f->sdf_wp[0], f->sdf_wp[1], f->sdf_wp[2], f->sdf_wp[3], f->sdf_wp[4],
                                      f->sdf_wpT[0], f->sdf_wpT[1], f->sdf_wpT[2], f->sdf_wpT[3], f->sdf_wpT[4],
                                      f->col_w0p, f->col_w1p, f->col_w2p, f->col_w3ap, f->col_w3bp, f->col_w4p,
                                      f->col_w0pT, f->col_w1pT, f->col_w2pT, f->col_w3apT, f->col_w3bpT, f->col_w4pT};
        PrepJobs jobs;
        for (int i = 0; i < kB3Count; ++i) {
            if (i >= 10 && i < 16) continue;   // forward colour products stay on the fp32 MFMA (train.hpp)
            const B3Src e = b3_source(i, kc0);
            jobs.add_b3(reinterpret_cast<bf16x8*>(const_cast<void*>(f->b3[i])), src[i], e.m_tiles, e.kc16);
        }
        if (!jobs.ok) return ARAH_E_LAUNCH;
        jobs.flush_b(s);
        if (idr) ARAH_LAUNCH_TRAIN(true, true, true, slab4);
        else ARAH_LAUNCH_TRAIN(false, true, true, slab4);
    } else {
        if (idr) ARAH_LAUNCH_TRAIN(true, true, false, slab4);
        else ARAH_LAUNCH_TRAIN(false, true, false, slab4);
    }
#undef ARAH_LAUNCH_TRAIN
    return check_launch();
}

// ---- density + compositing of the training forward and their backward (csrc/train.hpp) --------------------------------
int arah_composite_train_forward(int32_t n_rays, int32_t n_steps, int32_t render_last_pt, const int32_t* len,
                                 const int64_t* off, const float* sdf, const float* rgb, const float* z, const float* inv_beta,
                                 float* out_rgb, float* out_acc, void* stream) {
    if (n_rays < 0 || n_steps <= 0 || n_steps > ARAH_MAX_STEPS) return ARAH_E_BADARG;
    if (n_rays == 0) return ARAH_OK;
    if (!len || !off || !sdf || !rgb || !z || !inv_beta || !out_rgb || !out_acc) return ARAH_E_BADARG;
    hipLaunchKernelGGL(k_composite_train_fwd, dim3((n_rays + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), n_rays,
                       len, reinterpret_cast<const long long*>(off), sdf, rgb, z, inv_beta, render_last_pt, 1.0f / (float)n_steps,
                       out_rgb, out_acc);
    return check_launch();
}

int arah_composite_train_backward(int32_t n_rays, int32_t n_steps, int32_t render_last_pt, const int32_t* len,
                                  const int64_t* off, const float* sdf, const float* rgb, const float* z, const float* inv_beta,
                                  const float* g_rgb_map, const float* g_acc, float* g_sdf, float* g_rgb, float* g_inv_beta,
                                  void* stream) {
    if (n_rays < 0 || n_steps <= 0 || n_steps > ARAH_MAX_STEPS) return ARAH_E_BADARG;
    if (!g_inv_beta) return ARAH_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(g_inv_beta, 0, 4, s) != hipSuccess) return ARAH_E_LAUNCH;
    if (n_rays == 0) return ARAH_OK;
    if (!len || !off || !sdf || !rgb || !z || !inv_beta || !g_rgb_map || !g_acc || !g_sdf || !g_rgb) return ARAH_E_BADARG;
    hipLaunchKernelGGL(k_composite_train_bwd, dim3((n_rays + 63) / 64), dim3(64), 0, s, n_rays, len,
                       reinterpret_cast<const long long*>(off), sdf, rgb, z, inv_beta, render_last_pt, 1.0f / (float)n_steps,
                       g_rgb_map, g_acc, g_sdf, g_rgb, g_inv_beta);
    return check_launch();
}

// ---- skinny weight-gradient products of the training step -------------------------------------------------------
// out[i][j] = sum_p a[p][i] b[p][j] with m <= 4 rows (the 1 x 256 SDF head, the 3 x 256 colour head, the 256 x 3 first
// SIREN layer transposed): one pass over b at HBM speed.  A workgroup reduces kGramRows rows into partial[block][m][n];
// the caller adds the partials up (fixed order: the result is deterministic).
constexpr int kGramRows = 256;
__global__ __launch_bounds__(256) void k_gram_skinny(const float* __restrict__ a, int lda, int m,
                                                     const float* __restrict__ b, int ldb, int n, int P,
                                                     float* __restrict__ partial) {
    const int r0 = blockIdx.x * kGramRows, r1 = min(r0 + kGramRows, P);
    // round 6: four columns per lane and several rows side by side when b allows 16-byte loads (the first version walked its
    // 256 rows one dependent 4-byte load at a time: 0.9 TB/s; this one follows k_colsum)
    if ((n & 3) == 0 && (ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(b) & 15) == 0 && n <= 1024) {
        __shared__ f32x4 red[256];
        const int n4 = n >> 2, t = threadIdx.x;
        for (int c0 = 0; c0 < n4; c0 += 256) {
            const int C = min(n4 - c0, 256), RL = 256 / C;
            const int c = t % C, rl = t / C;
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rl < RL)
                for (int p = r0 + rl; p < r1; p += RL) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(b + (size_t)p * ldb) + c0 + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < m) acc[i] += v * a[(size_t)p * lda + i];
                }
            for (int i = 0; i < m; ++i) {
                red[t] = acc[i];
                __syncthreads();
                if (t < C) {
                    f32x4 sum = red[t];
                    for (int k = 1; k < RL; ++k) sum += red[k * C + t];
                    reinterpret_cast<f32x4*>(partial + ((size_t)blockIdx.x * m + i) * n)[c0 + t] = sum;
                }
                __syncthreads();
            }
        }
        return;
    }
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int p = r0; p < r1; ++p) {
            const float bj = b[(size_t)p * ldb + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < m) acc[i] = fmaf(a[(size_t)p * lda + i], bj, acc[i]);
        }
        for (int i = 0; i < m; ++i) partial[((size_t)blockIdx.x * m + i) * n + j] = acc[i];
    }
}

int32_t arah_gram_skinny_blocks(int32_t n_rows) { return n_rows > 0 ? (n_rows + kGramRows - 1) / kGramRows : 0; }

int arah_gram_skinny(const float* a, int32_t lda, int32_t m, const float* b, int32_t ldb, int32_t n, int32_t n_rows,
                     float* partial, void* stream) {
    if (!a || !b || !partial || m < 1 || m > 4 || n < 1 || lda < m || ldb < n || n_rows < 0) return ARAH_E_BADARG;
    if (n_rows == 0) return ARAH_OK;
    hipLaunchKernelGGL(k_gram_skinny, dim3(arah_gram_skinny_blocks(n_rows)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), a, lda, m, b, ldb, n, n_rows, partial);
    return check_launch();
}

// ---- column sums over the sample axis (training step): y[c] = sum_r s[r] a[r][c], s = 1 when null ---------------------
// The bias gradients of the tall layers (sum over 1.2e5 samples of a 25 .. 256-wide delta) and, with s = the upstream gradient
// and a = the 65 792 x 256 weight matrix of a hypernetwork head, the head's input gradient g W (the GEMM library runs that
// batch-1 product at 0.5 TB/s, torch's reduction a 25-wide sum at 0.04).  One pass at HBM speed: a workgroup reduces kColRows
// rows into partial[block][n_cols] (fixed order), k_colsum_finish adds the partials up in block order: deterministic.
constexpr int kColRows = 256;
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ a, long long lda, int n, long long n_rows,
                                                const float* __restrict__ scale, float* __restrict__ partial) {
    __shared__ float red[256 * 4];
    const long long r0 = (long long)blockIdx.x * kColRows, r1 = min(r0 + (long long)kColRows, n_rows);
    const int t = threadIdx.x;
    const bool vec = (n & 3) == 0 && (lda & 3) == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
    if (vec) {
        const int n4 = n >> 2;
        for (int c0 = 0; c0 < n4; c0 += 256) {
            const int C = min(n4 - c0, 256), RL = 256 / C;   // C column quads side by side, RL rows at a time
            const int c = t % C, rl = t / C;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (rl < RL)
                for (long long r = r0 + rl; r < r1; r += RL) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a + r * lda) + c0 + c);
                    const float sc = scale ? scale[r] : 1.0f;
                    acc += v * sc;
                }
            reinterpret_cast<f32x4*>(red)[t] = acc;
            __syncthreads();
            if (t < C) {
                f32x4 sum = reinterpret_cast<f32x4*>(red)[t];
                for (int k = 1; k < RL; ++k) sum += reinterpret_cast<f32x4*>(red)[k * C + t];
                reinterpret_cast<f32x4*>(partial + (size_t)blockIdx.x * n)[c0 + t] = sum;
            }
            __syncthreads();
        }
        return;
    }
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int C = min(n - c0, 256), RL = 256 / C;
        const int c = t % C, rl = t / C;
        float acc = 0.f;
        if (rl < RL)
            for (long long r = r0 + rl; r < r1; r += RL) acc = fmaf(a[r * lda + c0 + c], scale ? scale[r] : 1.0f, acc);
        red[t] = acc;
        __syncthreads();
        if (t < C) {
            float sum = red[t];
            for (int k = 1; k < RL; ++k) sum += red[k * C + t];
            partial[(size_t)blockIdx.x * n + c0 + t] = sum;
        }
        __syncthreads();
    }
}
// eight columns x thirty-two row lanes per workgroup: lane (rl, c) adds partial[rl], partial[rl + 32], ... of its column, the
// lane sums are added in lane order -- a fixed order, and ~15 dependent additions per thread instead of ~500
__global__ __launch_bounds__(256) void k_colsum_finish(const float* __restrict__ partial, int blocks, int n, float* __restrict__ y) {
    __shared__ float red[256];
    const int c = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
    float s = 0.f;
    if (c < n)
        for (int b = rl; b < blocks; b += 32) s += partial[(size_t)b * n + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 8 && c < n) {
        float t = red[threadIdx.x];
        for (int k = 1; k < 32; ++k) t += red[k * 8 + threadIdx.x];
        y[c] = t;
    }
}

int32_t arah_colsum_blocks(int64_t n_rows) { return n_rows > 0 ? (int32_t)((n_rows + kColRows - 1) / kColRows) : 0; }

int arah_colsum(const float* a, int64_t lda, int32_t n_cols, int64_t n_rows, const float* scale, float* partial, float* y,
                void* stream) {
    if (!a || !partial || !y || n_cols < 1 || lda < n_cols || n_rows < 0) return ARAH_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        hipMemsetAsync(y, 0, sizeof(float) * (size_t)n_cols, s);
        return check_launch();
    }
    const int blocks = arah_colsum_blocks(n_rows);
    hipLaunchKernelGGL(k_colsum, dim3(blocks), dim3(256), 0, s, a, (long long)lda, n_cols, (long long)n_rows, scale, partial);
    hipLaunchKernelGGL(k_colsum_finish, dim3((n_cols + 7) / 8), dim3(256), 0, s, (const float*)partial, blocks, n_cols, y);
    return check_launch();
}

// ---- inverses of P 3 x 3 matrices (the Jacobians of the re-attachment, IDR:315-334: torch.inverse there) -----------------
// inv[p] = (scale * m[p])^-1 by cofactors; the library's batched LU (factor + two substitutions + a row swap kernel) takes
// 0.5 ms for the 1.2e5 well-conditioned matrices of a training step.
__global__ void k_inverse3x3(const float* __restrict__ m, int n, float scale, float* __restrict__ inv) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float T[16], R[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) T[r * 4 + c] = m[(size_t)p * 9 + r * 3 + c] * scale;
    T[3] = T[7] = T[11] = T[12] = T[13] = T[14] = 0.f;
    T[15] = 1.f;
    inv3_of44(T, R);
#pragma unroll
    for (int k = 0; k < 9; ++k) inv[(size_t)p * 9 + k] = R[k];
}

int arah_inverse3x3(const float* m, int32_t n, float scale, float* inv, void* stream) {
    if (!m || !inv || n < 0) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    hipLaunchKernelGGL(k_inverse3x3, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), m, n, scale, inv);
    return check_launch();
}

// ---- hierarchical softmax of the training step's skinning queries, with its backward (utils/utils.py:138-181) -------------
// w = hsoftmax(scale * logits): a thread per point, the recursion of pointwise.hpp's hsoftmax written as a table of SPLITS
// (parent keeps w (1 - gate), child gets w gate) so that the reverse sweep walks the same table backwards.  On autograd the
// recursion is a gather of 240 factors per point and nine products (26 element-wise launches, a one-hot GEMM each way, ~1 ms
// of GPU time per step for the 1.2e5 re-attached samples); here one launch each way, 12 MB in and 12 MB out.
struct HsSplit {
    unsigned char parent, child, gate;
};
__device__ constexpr HsSplit kHsA[8] = {{1, 4, 4}, {2, 5, 5}, {3, 6, 6}, {4, 7, 7}, {5, 8, 8}, {6, 9, 9}, {7, 10, 10}, {8, 11, 11}};
__device__ constexpr HsSplit kHsB[9] = {{12, 15, 15}, {13, 16, 16}, {14, 17, 17}, {16, 18, 18}, {17, 19, 19},
                                        {18, 20, 20}, {19, 21, 21}, {20, 22, 22}, {21, 23, 23}};

extern "C++" {
template <bool BWD>
__global__ __launch_bounds__(128) void k_hsoftmax_train(const float* __restrict__ logits, int n, float scale,
                                                        const float* __restrict__ g_w, float* __restrict__ out) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float sg[25], w[24], keep[18];   // keep: the parent's weight before each split (reverse sweep)
    float xs[6];
#pragma unroll
    for (int i = 0; i < 25; ++i) {
        const float x = logits[(size_t)p * 25 + i] * scale;
        sg[i] = sigm(x);
        if (i >= 1 && i <= 3) xs[i - 1] = x;
        if (i >= 12 && i <= 14) xs[i - 9] = x;
    }
    float h[3], c[3];
    softmax3<float>(xs[0], xs[1], xs[2], h[0], h[1], h[2]);
    softmax3<float>(xs[3], xs[4], xs[5], c[0], c[1], c[2]);
    w[0] = 1.0f - sg[0];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[1 + k] = sg[0] * h[k];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const HsSplit s = kHsA[q];
        keep[q] = w[s.parent];
        w[s.child] = w[s.parent] * sg[s.gate];
        w[s.parent] = w[s.parent] * (1.0f - sg[s.gate]);
    }
    const float w9 = w[9], t = w9 * sg[24];
#pragma unroll
    for (int k = 0; k < 3; ++k) w[12 + k] = t * c[k];
    w[9] = w9 * (1.0f - sg[24]);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const HsSplit s = kHsB[q];
        keep[8 + q] = w[s.parent];
        w[s.child] = w[s.parent] * sg[s.gate];
        w[s.parent] = w[s.parent] * (1.0f - sg[s.gate]);
    }
    if (!BWD) {
#pragma unroll
        for (int k = 0; k < 24; ++k) out[(size_t)p * 24 + k] = w[k];
        return;
    }
    float a[24], gs[25];   // adjoints of the weights (of their CURRENT versions while walking back) and of the gates
#pragma unroll
    for (int k = 0; k < 24; ++k) a[k] = g_w[(size_t)p * 24 + k];
#pragma unroll
    for (int i = 0; i < 25; ++i) gs[i] = 0.f;
#pragma unroll
    for (int q = 8; q >= 0; --q) {
        const HsSplit s = kHsB[q];
        gs[s.gate] += keep[8 + q] * (a[s.child] - a[s.parent]);
        a[s.parent] = a[s.child] * sg[s.gate] + a[s.parent] * (1.0f - sg[s.gate]);
    }
    float gc[3], at = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gc[k] = a[12 + k] * t;
        at += a[12 + k] * c[k];
    }
    gs[24] += w9 * (at - a[9]);
    a[9] = at * sg[24] + a[9] * (1.0f - sg[24]);
#pragma unroll
    for (int q = 7; q >= 0; --q) {
        const HsSplit s = kHsA[q];
        gs[s.gate] += keep[q] * (a[s.child] - a[s.parent]);
        a[s.parent] = a[s.child] * sg[s.gate] + a[s.parent] * (1.0f - sg[s.gate]);
    }
    float gh[3], a0 = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gh[k] = a[1 + k] * sg[0];
        a0 += a[1 + k] * h[k];
    }
    gs[0] += a0 - a[0];
    float gx[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) gx[i] = gs[i] * sg[i] * (1.0f - sg[i]);
    const float dh = gh[0] * h[0] + gh[1] * h[1] + gh[2] * h[2], dc = gc[0] * c[0] + gc[1] * c[1] + gc[2] * c[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gx[1 + k] += h[k] * (gh[k] - dh);
        gx[12 + k] += c[k] * (gc[k] - dc);
    }
#pragma unroll
    for (int i = 0; i < 25; ++i) out[(size_t)p * 25 + i] = gx[i] * scale;
}
}   // extern "C++"

int arah_hsoftmax_train_forward(const float* logits, int32_t n, float scale, float* weights, void* stream) {
    if (!logits || !weights || n < 0) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    hipLaunchKernelGGL(k_hsoftmax_train<false>, dim3((n + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), logits, n,
                       scale, (const float*)nullptr, weights);
    return check_launch();
}

int arah_hsoftmax_train_backward(const float* logits, int32_t n, float scale, const float* g_weights, float* g_logits, void* stream) {
    if (!logits || !g_weights || !g_logits || n < 0) return ARAH_E_BADARG;
    if (n == 0) return ARAH_OK;
    hipLaunchKernelGGL(k_hsoftmax_train<true>, dim3((n + 127) / 128), dim3(128), 0, reinterpret_cast<hipStream_t>(stream), logits, n,
                       scale, g_weights, g_logits);
    return check_launch();
}

// ---- the hierarchical pose encoder of a training step as one launch each way (siren_modules.py:196-244) --------------------
// 24 joints, each a 19 -> 19 -> ReLU -> 6 MLP on [own(13) | feature of the parent (6)] (the root: the global feature), walked down
// the kinematic tree.  11 kFLOP -- and, level by level on autograd, ~70 launches forward and ~130 backward of a step whose host
// side is its bottleneck.  One workgroup: the joints in index order (SMPL numbers parents before children), a thread per output
// unit; the reverse sweep walks the joints backwards and hands each joint's input adjoint to its parent.
constexpr int kPtIn = 19, kPtHid = 19, kPtOut = 6, kPtOwn = 13, kPtMaxJoints = 64;
struct PoseTreeParents {
    signed char p[kPtMaxJoints];
};
__global__ __launch_bounds__(64) void k_pose_tree_fwd(const float* __restrict__ own, const float* __restrict__ glob,
                                                      const float* __restrict__ W1, const float* __restrict__ b1,
                                                      const float* __restrict__ W2, const float* __restrict__ b2,
                                                      PoseTreeParents par, int J, float* __restrict__ feats, float* __restrict__ hid) {
    __shared__ float f[kPtMaxJoints][kPtOut], x[kPtIn], h[kPtHid];
    const int t = threadIdx.x;
    for (int j = 0; j < J; ++j) {
        if (t < kPtOwn) x[t] = own[j * kPtOwn + t];
        else if (t < kPtIn) x[t] = par.p[j] < 0 ? glob[t - kPtOwn] : f[par.p[j]][t - kPtOwn];
        __syncthreads();
        if (t < kPtHid) {
            float a = b1[j * kPtHid + t];
            for (int k = 0; k < kPtIn; ++k) a = fmaf(W1[(j * kPtHid + t) * kPtIn + k], x[k], a);
            a = fmaxf(a, 0.f);
            h[t] = a;
            hid[j * kPtHid + t] = a;
        }
        __syncthreads();
        if (t < kPtOut) {
            float a = b2[j * kPtOut + t];
            for (int k = 0; k < kPtHid; ++k) a = fmaf(W2[(j * kPtOut + t) * kPtHid + k], h[k], a);
            f[j][t] = a;
            feats[j * kPtOut + t] = a;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(64) void k_pose_tree_bwd(const float* __restrict__ own, const float* __restrict__ glob,
                                                      const float* __restrict__ W1, const float* __restrict__ W2,
                                                      PoseTreeParents par, int J, const float* __restrict__ feats,
                                                      const float* __restrict__ hid, const float* __restrict__ g_feats,
                                                      float* __restrict__ gW1, float* __restrict__ gb1, float* __restrict__ gW2,
                                                      float* __restrict__ gb2, float* __restrict__ g_glob) {
    __shared__ float adj[kPtMaxJoints][kPtOut], gg[kPtOut], x[kPtIn], h[kPtHid], dh[kPtHid], df[kPtOut];
    const int t = threadIdx.x;
    for (int i = t; i < J * kPtOut; i += blockDim.x) adj[i / kPtOut][i % kPtOut] = g_feats[i];
    if (t < kPtOut) gg[t] = 0.f;
    __syncthreads();
    for (int j = J - 1; j >= 0; --j) {
        if (t < kPtOwn) x[t] = own[j * kPtOwn + t];
        else if (t < kPtIn) x[t] = par.p[j] < 0 ? glob[t - kPtOwn] : feats[par.p[j] * kPtOut + t - kPtOwn];
        if (t < kPtHid) h[t] = hid[j * kPtHid + t];
        if (t < kPtOut) df[t] = adj[j][t];
        __syncthreads();
        if (t < kPtHid) {
            float a = 0.f;
            for (int o = 0; o < kPtOut; ++o) a = fmaf(W2[(j * kPtOut + o) * kPtHid + t], df[o], a);
            a = h[t] > 0.f ? a : 0.f;
            dh[t] = a;
            gb1[j * kPtHid + t] = a;
            for (int o = 0; o < kPtOut; ++o) gW2[(j * kPtOut + o) * kPtHid + t] = df[o] * h[t];
        }
        if (t < kPtOut) gb2[j * kPtOut + t] = df[t];
        __syncthreads();
        if (t < kPtHid)
            for (int k = 0; k < kPtIn; ++k) gW1[(j * kPtHid + t) * kPtIn + k] = dh[t] * x[k];
        if (t >= kPtOwn && t < kPtIn) {      // the parent's feature was this joint's input k = t
            float a = 0.f;
            for (int i = 0; i < kPtHid; ++i) a = fmaf(W1[(j * kPtHid + i) * kPtIn + t], dh[i], a);
            if (par.p[j] < 0) gg[t - kPtOwn] += a;
            else adj[par.p[j]][t - kPtOwn] += a;
        }
        __syncthreads();
    }
    if (t < kPtOut) g_glob[t] = gg[t];
}

static int pose_tree_parents(const int32_t* parents, int32_t n_joints, PoseTreeParents& par) {
    if (!parents || n_joints < 1 || n_joints > kPtMaxJoints) return ARAH_E_BADARG;
    for (int j = 0; j < n_joints; ++j) {
        if (parents[j] >= j || parents[j] < -1) return ARAH_E_BADARG;    // parents precede their children
        par.p[j] = (signed char)parents[j];
    }
    return ARAH_OK;
}

int arah_pose_tree_forward(const float* own, const float* glob, const float* W1, const float* b1, const float* W2,
                           const float* b2, const int32_t* parents_host, int32_t n_joints, float* feats, float* hidden,
                           void* stream) {
    if (!own || !glob || !W1 || !b1 || !W2 || !b2 || !feats || !hidden) return ARAH_E_BADARG;
    PoseTreeParents par;
    if (int rc = pose_tree_parents(parents_host, n_joints, par)) return rc;
    hipLaunchKernelGGL(k_pose_tree_fwd, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), own, glob, W1, b1, W2, b2, par,
                       n_joints, feats, hidden);
    return check_launch();
}

int arah_pose_tree_backward(const float* own, const float* glob, const float* W1, const float* W2, const int32_t* parents_host,
                            int32_t n_joints, const float* feats, const float* hidden, const float* g_feats, float* gW1,
                            float* gb1, float* gW2, float* gb2, float* g_glob, void* stream) {
    if (!own || !glob || !W1 || !W2 || !feats || !hidden || !g_feats || !gW1 || !gb1 || !gW2 || !gb2 || !g_glob) return ARAH_E_BADARG;
    PoseTreeParents par;
    if (int rc = pose_tree_parents(parents_host, n_joints, par)) return rc;
    hipLaunchKernelGGL(k_pose_tree_bwd, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), own, glob, W1, W2, par, n_joints,
                       feats, hidden, g_feats, gW1, gb1, gW2, gb2, g_glob);
    return check_launch();
}

// ---- the wide output layers of the SDF hypernetwork (hyperlayers.py:418-465) ----------------------------
// y[r] = W[r, :] . x + b0[r] (+ b1[r]): a 256 -> 65 792 linear layer per emitted 256 x 256 SDF layer, 67 MB of weights that
// are read once per frame -- 337 MB for the five hidden layers, a pure HBM stream.  The GEMM library runs these batch-1
// products at ~0.9 TB/s (76 us each, rocprofv3); here a wave owns rows: a row of 256 floats is ONE coalesced 1 KiB load
// (float4 per lane), eight rows are in flight per wave, the row sums go through six xor-shuffles.
constexpr int kGemvRows = 8;
__global__ __launch_bounds__(256) void k_gemv_rows(const float* __restrict__ W, int n_rows, int n_cols,
                                                    const float* __restrict__ x, const float* __restrict__ b0,
                                                    const float* __restrict__ b1, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const int n4 = n_cols >> 2;                       // float4 per row
    for (int r0 = wave * kGemvRows; r0 < n_rows; r0 += n_waves * kGemvRows) {
        float acc[kGemvRows];
#pragma unroll
        for (int u = 0; u < kGemvRows; ++u) acc[u] = 0.f;
        for (int c = lane; c < n4; c += 64) {
            const f32x4 xv = reinterpret_cast<const f32x4*>(x)[c];
            f32x4 wv[kGemvRows];
#pragma unroll
            for (int u = 0; u < kGemvRows; ++u) {
                const int r = min(r0 + u, n_rows - 1);
                wv[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(W + (size_t)r * n_cols) + c);
            }
#pragma unroll
            for (int u = 0; u < kGemvRows; ++u)
                acc[u] = fmaf(wv[u][3], xv[3], fmaf(wv[u][2], xv[2], fmaf(wv[u][1], xv[1], fmaf(wv[u][0], xv[0], acc[u]))));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < kGemvRows; ++u) acc[u] += __shfl_xor(acc[u], o);
        if (lane < kGemvRows && r0 + lane < n_rows) {
            float v = acc[0];
#pragma unroll
            for (int u = 1; u < kGemvRows; ++u) v = lane == u ? acc[u] : v;
            const int r = r0 + lane;
            y[r] = v + (b0 ? b0[r] : 0.f) + (b1 ? b1[r] : 0.f);
        }
    }
}

int arah_gemv_rows(const float* W, int32_t n_rows, int32_t n_cols, const float* x, const float* b0, const float* b1,
                   float* y, void* stream) {
    if (!W || !x || !y || n_rows < 1 || n_cols < 4 || (n_cols & 3)) return ARAH_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(x)) & 15) return ARAH_E_BADARG;
    long long g = ((long long)n_rows + 4 * kGemvRows - 1) / (4 * kGemvRows);   // four waves per workgroup
    if (g > 8 * num_cus()) g = 8 * num_cus();
    hipLaunchKernelGGL(k_gemv_rows, dim3((int)g), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), W, n_rows, n_cols, x,
                       b0, b1, y);
    return check_launch();
}

// ---- mesh queries of the training data path ------------------------------------------------------------------
size_t arah_mesh_query_scratch_bytes(void) { return 256; }

int arah_mesh_query(const float* verts, int32_t n_verts, const int32_t* faces, int32_t n_faces, const void* pts,
                    int32_t pts_are_f64, int32_t n_pts, double* d2, int32_t* face, double* closest, double* bary,
                    uint8_t* inside, void* scratch, void* stream) {
    if (!verts || !faces || n_verts < 1 || n_faces < 1 || n_pts < 0 || !scratch) return ARAH_E_BADARG;
    if (n_pts == 0) return ARAH_OK;
    if (!pts || !d2 || !face || !closest || !bary || !inside) return ARAH_E_BADARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MeshBox* box = reinterpret_cast<MeshBox*>(scratch);
    hipLaunchKernelGGL(k_mesh_box, dim3(1), dim3(256), 0, s, verts, faces, n_faces, 512.0, box);
    const int g = (n_pts + kMeshThreads - 1) / kMeshThreads;
    if (pts_are_f64)
        hipLaunchKernelGGL(k_mesh_query<double>, dim3(g), dim3(kMeshThreads), 0, s, verts, faces, n_faces, (const MeshBox*)box,
                           reinterpret_cast<const double*>(pts), n_pts, d2, face, closest, bary, inside);
    else
        hipLaunchKernelGGL(k_mesh_query<float>, dim3(g), dim3(kMeshThreads), 0, s, verts, faces, n_faces, (const MeshBox*)box,
                           reinterpret_cast<const float*>(pts), n_pts, d2, face, closest, bary, inside);
    return check_launch();
}

}  // extern "C"

// ---- tiered eval forward (tier.hpp) --------------------------------------------------------------
static OccBuf carve_occ(void* base) {
    OccBuf o;
    Carver c{reinterpret_cast<char*>(base), 0};
    o.info = c.take<OccInfo>(1);
    o.bits = c.take<unsigned>(kOccMaxVox / 32);
    o.dist = c.take<uint8_t>(kOccMaxVox);
    o.csdf = c.take<float>((size_t)kOccNc * kOccNc * kOccNc);
    o.cpts = c.take<float>((size_t)kOccNc * kOccNc * kOccNc * 3);
    o.cell_lip = c.take<float>(kOccMaxCells);
    o.cell_stretch = c.take<float>(kOccMaxCells);
    o.fnorm = c.take<float>((size_t)kOccMaxFine * 3);
    o.fsdf = c.take<float>(kOccMaxFine);
    o.iota = c.take<int>(kOccMaxFine);
    o.sel_raw = c.take<float>((size_t)kOccMaxFine * 3);
    o.sel_bar = c.take<float>((size_t)kOccMaxFine * 3);
    o.sel_idx = c.take<int>(kOccMaxFine);
    o.sel_of = c.take<int>(kOccMaxFine);
    o.bytes = align_up(c.off, 256);
    return o;
}

extern "C" {
size_t arah_occupancy_bytes(void) { return carve_occ(nullptr).bytes; }

// The posed fat body of a prepared frame as a bitmap (tier.hpp): eleven short launches on `stream`, ~2.5e5 SDF and a few 1e4
// skinning evaluations.  The buffer is the caller's (arah_occupancy_bytes(), 256-byte aligned) and belongs to THIS frame: hand it
// to arah_render through ArahSampling.occupancy.
int arah_prepare_occupancy(const ArahFrame* f, void* occ_buf, size_t occ_bytes, void* workspace, size_t wbytes, void* stream) {
    if (!f || !occ_buf || !workspace) return ARAH_E_BADARG;
    OccBuf o = carve_occ(occ_buf);
    if (occ_bytes < o.bytes) return ARAH_E_WORKSPACE;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const int nc3 = kOccNc * kOccNc * kOccNc, m3 = (kOccNc - 1) * (kOccNc - 1) * (kOccNc - 1);
    hipLaunchKernelGGL(k_occ_begin, dim3(1), dim3(64), 0, s, fd, knn_of(fd).grid, o.info);
    hipLaunchKernelGGL(k_occ_lattice_pts, dim3((nc3 + 255) / 256), dim3(256), 0, s, kOccNc, kOccL, o.cpts);
    LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(grid_for(nc3, kTile)), dim3(kThreads),
                  kLdsSdfFwd, s, fd, (const float*)o.cpts, (const int*)nullptr, (const int*)nullptr, nc3, o.csdf, (float*)nullptr,
                  (float*)nullptr, (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr, 0);
    hipLaunchKernelGGL(k_occ_cells, dim3((m3 + 255) / 256), dim3(256), 0, s, fd, (const float*)o.csdf, kOccNc, kOccL, o.info,
                       o.cell_lip, o.fnorm, o.iota);
    hipLaunchKernelGGL(k_occ_fix, dim3(1), dim3(64), 0, s, o.info);
    LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(grid_for(kOccMaxFine, kTile)),
                  dim3(kThreads), kLdsSdfFwd, s, fd, (const float*)o.fnorm, (const int*)o.iota, (const int*)&o.info->n_fine, 0,
                  o.fsdf, (float*)nullptr, (float*)nullptr, (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr, 0);
    hipLaunchKernelGGL(k_occ_cell_slope, dim3((kOccMaxCells + 3) / 4), dim3(256), 0, s, kOccNc, kOccL, o.info, (const float*)o.fsdf, o.cell_lip);
    hipMemsetAsync(o.sel_of, 0xff, sizeof(int) * (size_t)kOccMaxFine, s);
    hipLaunchKernelGGL(k_occ_select, dim3((kOccMaxFine + 255) / 256), dim3(256), 0, s, fd, kOccNc, kOccL, o.info,
                       (const float*)o.cell_lip, (const float*)o.fnorm, (const float*)o.fsdf, o.sel_raw, o.sel_idx, o.sel_of);
    hipLaunchKernelGGL(k_skin_eval, dim3(grid_for(kOccMaxFine, kTile)), dim3(kThreads), kLdsSkin, s, fd, (const float*)o.sel_raw,
                       kOccMaxFine, (float*)nullptr, o.sel_bar, (float*)nullptr, &w.ctr->n_skin_fwd, (const int*)&o.info->n_sel, 1);
    hipLaunchKernelGGL(k_occ_cell_stretch, dim3((kOccMaxCells + 3) / 4), dim3(256), 0, s, fd, kOccNc, kOccL, o.info,
                       (const int*)o.sel_of, (const float*)o.sel_bar, o.cell_stretch);
    hipMemsetAsync(o.bits, 0, sizeof(unsigned) * (kOccMaxVox / 32), s);
    hipLaunchKernelGGL(k_occ_mark, dim3((kOccMaxFine + 255) / 256), dim3(256), 0, s, fd, kOccNc, kOccL, o.info,
                       (const float*)o.sel_bar, (const int*)o.sel_idx, (const float*)o.cell_stretch, o.bits);
    // the distance transform walks lines of the bitmap's box: its dimensions live on the device, the launches cover the
    // largest box the buffer can hold per pair of axes (lines beyond the box return at once)
    const int max_lines = kOccMaxVox / 4;   // a box with fewer than 4 voxels along an axis does not occur (>= 0.16 m / voxel)
    hipLaunchKernelGGL(k_occ_dist_x, dim3((max_lines + 255) / 256), dim3(256), 0, s, (const OccInfo*)o.info, (const unsigned*)o.bits, o.dist);
    hipLaunchKernelGGL(k_occ_dist_yz<1>, dim3((max_lines + 255) / 256), dim3(256), 0, s, (const OccInfo*)o.info, o.dist);
    hipLaunchKernelGGL(k_occ_dist_yz<2>, dim3((max_lines + 255) / 256), dim3(256), 0, s, (const OccInfo*)o.info, o.dist);
    return check_launch();
}

struct BandScratch {
    int* count;       // points evaluated exactly (first word of the buffer)
    float *cpts, *csdf;
    uint8_t *flag, *flag2;
    size_t bytes;
};
static BandScratch carve_band(void* base) {
    constexpr int nc3 = kBandNc * kBandNc * kBandNc, m3 = (kBandNc - 1) * (kBandNc - 1) * (kBandNc - 1);
    BandScratch b;
    Carver c{reinterpret_cast<char*>(base), 0};
    b.count = c.take<int>(1);
    b.cpts = c.take<float>((size_t)nc3 * 3);
    b.csdf = c.take<float>(nc3);
    b.flag = c.take<uint8_t>(m3);
    b.flag2 = c.take<uint8_t>(m3);
    b.bytes = align_up(c.off, 256);
    return b;
}
size_t arah_sdf_grid_band_scratch_bytes(void) { return carve_band(nullptr).bytes; }

// The lattice of arah_sdf_grid for marching cubes at level 0: exact values wherever the level set can pass (and one coarse cell
// around), the right sign elsewhere (tier.hpp).  list: [n_side^3] ints of scratch; scratch: arah_sdf_grid_band_scratch_bytes().
// The number of evaluated points stays on the device (first word of `scratch`); n_side >= 33.
int arah_sdf_grid_band(const ArahFrame* f, int32_t n_side, float* sdf, int32_t* list, void* scratch, size_t scratch_bytes,
                       void* workspace, size_t wbytes, void* stream) {
    if (!f || !sdf || !list || !scratch || !workspace || n_side < kBandNc || n_side > 1024) return ARAH_E_BADARG;
    if (scratch_bytes < arah_sdf_grid_band_scratch_bytes()) return ARAH_E_WORKSPACE;
    Workspace w = carve(workspace, 1, 1);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const FrameDev fd = to_dev(*f);
    const long long n = (long long)n_side * n_side * n_side;
    if (n > 0x7fffffffLL) return ARAH_E_BADARG;
    constexpr int nc3 = kBandNc * kBandNc * kBandNc, m3 = (kBandNc - 1) * (kBandNc - 1) * (kBandNc - 1);
    const BandScratch b = carve_band(scratch);
    float *cpts = b.cpts, *csdf = b.csdf;
    uint8_t *flag = b.flag, *flag2 = b.flag2;
    int* count = b.count;
    hipMemsetAsync(count, 0, sizeof(int), s);
    hipLaunchKernelGGL(k_occ_lattice_pts, dim3((nc3 + 255) / 256), dim3(256), 0, s, kBandNc, 1.0f, cpts);
    LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(grid_for(nc3, kTile)), dim3(kThreads),
                  kLdsSdfFwd, s, fd, (const float*)cpts, (const int*)nullptr, (const int*)nullptr, nc3, csdf, (float*)nullptr,
                  (float*)nullptr, (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr, 0);
    hipLaunchKernelGGL(k_band_cells, dim3((m3 + 255) / 256), dim3(256), 0, s, (const float*)csdf, flag);
    hipLaunchKernelGGL(k_band_dilate, dim3((m3 + 255) / 256), dim3(256), 0, s, (const uint8_t*)flag, flag2);
    hipLaunchKernelGGL(k_band_fill, dim3((int)((n + 1023) / 1024)), dim3(1024), 0, s, (int)n_side, (const float*)csdf,
                       (const uint8_t*)flag2, sdf, list, count);
    LAUNCH_ENGINE(fd.split, (k_sdf_eval<false, true>), (k_sdf_eval<false, false>), dim3(grid_for(n, kTile)), dim3(kThreads),
                  kLdsSdfFwd, s, fd, (const float*)nullptr, (const int*)list, (const int*)count, 0, sdf, (float*)nullptr,
                  (float*)nullptr, (f32x4*)nullptr, &w.ctr->n_sdf_fwd, (unsigned long long*)nullptr, (int)n_side);
    return check_launch();
}

// tests / bench: the tier of every ray of the workspace's last arah_render (0 certified zero, 1 surface ray, 2 promoted;
// written by the tiered path only) and whether any valid sample of it carries density > 0 (either path, lazy shading)
int arah_tier_debug(void* workspace, size_t wbytes, int32_t n_rays, int32_t n_steps, uint8_t* ray_tier, uint8_t* ray_sigma_pos,
                    void* stream) {
    if (!workspace || n_rays <= 0 || n_steps <= 0) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n_rays, n_steps);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (ray_tier && hipMemcpyAsync(ray_tier, w.ray_tier, (size_t)n_rays, hipMemcpyDeviceToDevice, s) != hipSuccess) return ARAH_E_LAUNCH;
    if (ray_sigma_pos)
        hipLaunchKernelGGL(k_tier_ray_sigma, dim3((n_rays + 255) / 256), dim3(256), 0, s, n_rays, n_steps, (const uint8_t*)w.o_mask,
                           (const f32x4*)w.shaded, ray_sigma_pos);
    return check_launch();
}

// tests: the per-sample arrays of the workspace's last arah_render (device to device; any pointer may be NULL):
// z [N,S], pts [N,S,3], T [N,S,16], mask [N,S], shaded [N,S,4] = {rgb, density}, state [N,S] (tiered path: TS_*)
int arah_debug_samples(void* workspace, size_t wbytes, int32_t n_rays, int32_t n_steps, float* z, float* pts, float* T,
                       uint8_t* mask, float* shaded, uint8_t* state, void* stream) {
    if (!workspace || n_rays <= 0 || n_steps <= 0) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n_rays, n_steps);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t Q = (size_t)n_rays * n_steps;
    bool ok = true;
    if (z) ok = ok && hipMemcpyAsync(z, w.o_z, Q * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (pts) ok = ok && hipMemcpyAsync(pts, w.o_pts, Q * 12, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (T) ok = ok && hipMemcpyAsync(T, w.o_T, Q * 64, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (mask) ok = ok && hipMemcpyAsync(mask, w.o_mask, Q, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (shaded) ok = ok && hipMemcpyAsync(shaded, w.shaded, Q * 16, hipMemcpyDeviceToDevice, s) == hipSuccess;
    if (state) ok = ok && hipMemcpyAsync(state, w.q_smask, Q, hipMemcpyDeviceToDevice, s) == hipSuccess;
    return ok ? ARAH_OK : ARAH_E_LAUNCH;
}

// tests: the header of an occupancy buffer (synchronises the stream)
int arah_occupancy_info(const void* occ_buf, int32_t* h_out16, void* stream) {
    if (!occ_buf || !h_out16) return ARAH_E_BADARG;
    static_assert(sizeof(OccInfo) == 64, "OccInfo is 16 words");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(h_out16, occ_buf, sizeof(OccInfo), hipMemcpyDeviceToHost, s) != hipSuccess) return ARAH_E_LAUNCH;
    return hipStreamSynchronize(s) == hipSuccess ? ARAH_OK : ARAH_E_LAUNCH;
}
}  // extern "C"

// one phase of the tiers: nearest vertex + inverse LBS, loop C, normalisation + masks, density of the converged samples
static int tier_phase(const ArahFrame* f, const ArahSampling* cfg, const FrameDev& fd, Workspace& w, const RaySet& rs, long long Q,
                      const int* list, int* cnt /* {n, queue head, n converged} */, int* dens_list, int* n_shade, int phase,
                      hipStream_t s) {
    const int S = cfg->n_steps;
    launch_nearest<SRC_SAMPLES>(s, fd, Q, (const float*)nullptr, rs, (const float*)w.o_z, S, list, (const int*)&cnt[0], 0,
                                (int*)nullptr, w.o_pts, w.o_T, 1, &w.ctr->n_knn);
    void* const* evd = phase == 1 ? cfg->ev_density : cfg->ev_density2;
    int rc = run_broyden3(fd, w, nullptr, CanonOut{w.o_pts, w.o_T, w.q_err}, Q, s, cfg->canon_kernel,
                          phase == 1 ? cfg->ev_canon : cfg->ev_canon2, list, cnt);
    if (rc) return rc;
    // phase 1: the witnesses at the head of the list are certified sigma = +0; phase 2: every sample is (tier.hpp)
    hipLaunchKernelGGL(k_tier_finalize, dim3(grid_for(Q, 256)), dim3(256), 0, s, fd, list, (const int*)&cnt[0],
                       phase == 1 ? (const int*)&w.tcounts[TC_NWIT] : (const int*)&cnt[0], (const float*)w.q_err, w.o_pts, w.o_mask,
                       w.shaded, dens_list, &cnt[2]);
    if (phase == 1) {
        const bool ev = evd[0] && evd[1];
        if (ev) hipEventRecord(reinterpret_cast<hipEvent_t>(evd[0]), s);
        launch_density(fd, w, w.o_pts, Q, (const int*)dens_list, (const int*)&cnt[2], w.listB, n_shade, s);
        if (ev) hipEventRecord(reinterpret_cast<hipEvent_t>(evd[1]), s);
    }
    (void)f;
    return check_launch();
}

static int render_tiers(const ArahFrame* f, const ArahSampling* cfg, Workspace& w, const float* cam_loc, int32_t rays_per_cam,
                        const float* dirs, const float* near_far, const uint8_t* conv, const float* start, const float* end,
                        int32_t n, float* rgb, float* acc, uint8_t* vol_mask, hipStream_t s) {
    const int S = cfg->n_steps;
    const FrameDev fd = to_dev(*f);
    const RaySet rs = make_rays(cam_loc, dirs, rays_per_cam);
    const long long Q = (long long)n * S;
    OccBuf o = carve_occ(const_cast<void*>(cfg->occupancy));
    int* tc = w.tcounts;
    TierStats* stats = &w.ctr->tier;
    hipMemsetAsync(tc, 0, sizeof(int) * TC_COUNT, s);
    hipMemsetAsync(w.o_mask, 0, (size_t)Q, s);
    hipLaunchKernelGGL(k_sample_depths, dim3((n + 127) / 128), dim3(128), 0, s, n, S, cfg->n_near, cfg->n_far, near_far, conv,
                       start, end, cfg->lin_steps, cfg->lin_near, cfg->lin_far, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, w.o_z, w.q_smask);
    const dim3 gt(min(2048, (n + kTierWaves - 1) / kTierWaves)), bt(kTierWaves * 64);
    hipLaunchKernelGGL(k_tier_classify<0>, gt, bt, 0, s, n, S, rs, conv, (const float*)w.o_z, w.q_smask,
                       (const OccInfo*)o.info, (const unsigned*)o.bits, (const uint8_t*)o.dist, w.listA, &tc[TC_N1], stats);
    hipMemcpyAsync(&tc[TC_NWIT], &tc[TC_N1], sizeof(int), hipMemcpyDeviceToDevice, s);   // the witnesses head the list
    hipLaunchKernelGGL(k_tier_classify<1>, gt, bt, 0, s, n, S, rs, conv, (const float*)w.o_z, w.q_smask,
                       (const OccInfo*)o.info, (const unsigned*)o.bits, (const uint8_t*)o.dist, w.listA, &tc[TC_N1], stats);
    // phase 1: surface rays, marked samples, witnesses
    int rc = tier_phase(f, cfg, fd, w, rs, Q, w.listA, &tc[TC_N1], w.listC, &tc[TC_NSHADE], 1, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tier_promote<0>, gt, bt, 0, s, n, S, conv, w.q_smask, (const uint8_t*)w.o_mask,
                       (const f32x4*)w.shaded, w.listD, &tc[TC_N2], w.ray_tier, stats);
    hipLaunchKernelGGL(k_tier_promote<1>, gt, bt, 0, s, n, S, conv, w.q_smask, (const uint8_t*)w.o_mask,
                       (const f32x4*)w.shaded, w.listD, &tc[TC_N2], w.ray_tier, stats);
    // phase 2: the remaining samples of the promoted rays
    hipLaunchKernelGGL(k_tier_snap, dim3(1), dim3(64), 0, s, (const unsigned long long*)&w.ctr->n_canon,
                       (const unsigned long long*)&w.ctr->n_density, w.ctr->tier_snap, &w.ctr->n_canon_p2, &w.ctr->n_density_p2, 0);
    rc = tier_phase(f, cfg, fd, w, rs, Q, w.listD, &tc[TC_N2], w.listA, &tc[TC_NSHADE], 2, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tier_snap, dim3(1), dim3(64), 0, s, (const unsigned long long*)&w.ctr->n_canon,
                       (const unsigned long long*)&w.ctr->n_density, w.ctr->tier_snap, &w.ctr->n_canon_p2, &w.ctr->n_density_p2, 1);
    return shade_tail(f, cfg, w, fd, dirs, w.o_z, w.o_pts, w.o_T, w.o_mask, n, w.listB, &tc[TC_NSHADE], rgb, acc, vol_mask, s);
}

extern "C" {
// ---- whole eval forward -------------------------------------------------------------------------
int arah_render(const ArahFrame* f, const ArahSampling* cfg, const float* cam_loc, int32_t rays_per_cam,
                const float* dirs, const float* near_far, const float* d_pose34, int32_t n, float* rgb,
                float* points_cam, uint8_t* vol_mask, float* acc, float* dists, uint8_t* surface_conv,
                void* workspace, size_t wbytes, void* stream) {
    if (!f || !cfg || n < 0 || !workspace) return ARAH_E_BADARG;
    int rc = check_sampling(cfg);
    if (rc) return rc;
    if (n == 0) return ARAH_OK;   // empty ray set: nothing to do, data pointers may be NULL
    if (!cam_loc || !dirs || !near_far || !rgb || !vol_mask) return ARAH_E_BADARG;
    if (points_cam && !d_pose34) return ARAH_E_BADARG;
    Workspace w = carve(workspace, n, cfg->n_steps);
    if (wbytes < w.bytes) return ARAH_E_WORKSPACE;
    if (int arc = setup_attributes()) return arc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    float* o_start = dists ? dists : w.o_start;
    uint8_t* o_conv = surface_conv ? surface_conv : w.o_conv;
    const uint8_t* skip = nullptr;
    if (cfg->occupancy) {   // rays whose segment misses the posed fat body cannot converge in loops A+B (tier.hpp)
        OccBuf o = carve_occ(const_cast<void*>(cfg->occupancy));
        hipLaunchKernelGGL(k_tier_rays, dim3((n + 255) / 256), dim3(256), 0, s, n, make_rays(cam_loc, dirs, rays_per_cam), near_far,
                           (const OccInfo*)o.info, (const uint8_t*)o.dist, w.ray_tier, &w.ctr->tier);
        skip = w.ray_tier;   // (the promote step overwrites it with the rays' tiers once the tracer is done)
    }
    rc = trace_impl(f, w, cam_loc, rays_per_cam, dirs, near_far, n, 0, w.o_xnorm, w.o_Tray, o_conv, o_start, w.o_end, s, skip);
    if (rc) return rc;
    if (cfg->occupancy && !cfg->full_shading) {
        rc = render_tiers(f, cfg, w, cam_loc, rays_per_cam, dirs, near_far, o_conv, o_start, w.o_end, n, rgb,
                          acc ? acc : w.o_acc, vol_mask, s);
        if (rc) return rc;
    } else {
        rc = sample_impl(f, cfg, w, cam_loc, rays_per_cam, dirs, near_far, o_conv, o_start, w.o_end, n, nullptr, nullptr,
                         nullptr, w.o_z, w.o_pts, w.o_T, w.o_mask, s);
        if (rc) return rc;
        rc = shade_impl(f, cfg, w, dirs, w.o_z, w.o_pts, w.o_T, w.o_mask, n, rgb, acc ? acc : w.o_acc, vol_mask, s);
        if (rc) return rc;
    }
    if (points_cam)
        hipLaunchKernelGGL(k_points_cam, dim3((n + 255) / 256), dim3(256), 0, s, to_dev(*f), n,
                           make_rays(cam_loc, dirs, rays_per_cam), (const float*)o_start, (const uint8_t*)o_conv,
                           (const float*)w.o_xnorm, d_pose34, points_cam);
    return check_launch();
}

}  // extern "C"
