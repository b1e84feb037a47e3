// pointwise.hpp -- per-point (one lane = one point) pieces of the hot path:
// coordinate normalisation, hierarchical softmax, LBS blend, small inverses, Broyden updates.
// Each function cites the reference lines whose arithmetic it restates (paths relative to the
// reference root; RFU = im2mesh/utils/root_finding_utils.py).
#pragma once
#include <hip/hip_runtime.h>

namespace arah {

struct BodyConst {
    float trans[3];
    float center[3];
    float cmin, cmax;
};

struct V3 {
    float x, y, z;
};

// RFU:37-44
// The padding is rounded on its own, like the tensor operation of the reference (RFU:39): under hipcc's default
// -ffp-contract=fast the sum below became fma(rng, 0.05, .) in some kernels and stayed mul + add in others (where the
// loop-invariant product was hoisted), and the same point normalised by two kernels differed in its last bit (round 6:
// k_canon_finalize against k_tier_finalize).  The empty asm keeps the product out of the contraction everywhere.
__device__ __forceinline__ V3 normalize_pt(const BodyConst& bc, V3 p) {
    const float rng = bc.cmax - bc.cmin;
    float pad = rng * 0.05f;
    asm volatile("" : "+v"(pad));
    V3 o;
    o.x = (((p.x - bc.center[0]) - bc.cmin + pad) / rng / 1.1f - 0.5f) * 2.0f;
    o.y = (((p.y - bc.center[1]) - bc.cmin + pad) / rng / 1.1f - 0.5f) * 2.0f;
    o.z = (((p.z - bc.center[2]) - bc.cmin + pad) / rng / 1.1f - 0.5f) * 2.0f;
    return o;
}

// RFU:47-51
__device__ __forceinline__ V3 unnormalize_pt(const BodyConst& bc, V3 p) {
    const float rng = bc.cmax - bc.cmin;
    const float pad = rng * 0.05f;
    V3 o;
    o.x = (p.x / 2.0f + 0.5f) * 1.1f * rng + bc.cmin - pad + bc.center[0];
    o.y = (p.y / 2.0f + 0.5f) * 1.1f * rng + bc.cmin - pad + bc.center[1];
    o.z = (p.z / 2.0f + 0.5f) * 1.1f * rng + bc.cmin - pad + bc.center[2];
    return o;
}

__device__ __forceinline__ float sdf_scale(const BodyConst& bc) { return (bc.cmax - bc.cmin) * 1.1f / 2.0f; }

// ---- scalar type with 3 tangents, for forward-mode derivatives of the per-point functions ------
struct Dual3 {
    float v, d[3];
};
__device__ __forceinline__ Dual3 mk(float v) { return Dual3{v, {0.f, 0.f, 0.f}}; }
__device__ __forceinline__ Dual3 operator+(Dual3 a, Dual3 b) {
    return Dual3{a.v + b.v, {a.d[0] + b.d[0], a.d[1] + b.d[1], a.d[2] + b.d[2]}};
}
__device__ __forceinline__ Dual3 operator-(Dual3 a, Dual3 b) {
    return Dual3{a.v - b.v, {a.d[0] - b.d[0], a.d[1] - b.d[1], a.d[2] - b.d[2]}};
}
__device__ __forceinline__ Dual3 operator*(Dual3 a, Dual3 b) {
    return Dual3{a.v * b.v, {a.d[0] * b.v + a.v * b.d[0], a.d[1] * b.v + a.v * b.d[1], a.d[2] * b.v + a.v * b.d[2]}};
}
__device__ __forceinline__ Dual3 operator/(Dual3 a, Dual3 b) {
    const float q = a.v / b.v, ib = 1.0f / b.v;
    return Dual3{q, {(a.d[0] - q * b.d[0]) * ib, (a.d[1] - q * b.d[1]) * ib, (a.d[2] - q * b.d[2]) * ib}};
}
// exp(y), y <= 0 in every use (sigmoid of -|x|, softmax after the max shift): the hardware exp2 fed with y log2(e).
// The product's rounding perturbs the argument by |y| 2^-24, i.e. the result by that RELATIVE amount: < 1.2e-6 at
// y = -20, where exp(y) itself is 2e-9 -- far below fp32 resolution of every quantity these exponentials are added
// to or normalised by (1 + e, sums of softmax terms).  One multiply + one transcendental instead of the ten
// operations of a Cody-Waite reduction: transcendentals cost ~16 cycles per wave on gfx950, plain VALU ~2.
__device__ __forceinline__ float exp_fast(float y) {
    return __builtin_amdgcn_exp2f(fmaxf(y, -126.0f) * 1.44269504088896341f);
}
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float one_minus(float a) { return 1.0f - a; }
__device__ __forceinline__ Dual3 one_minus(Dual3 a) { return Dual3{1.0f - a.v, {-a.d[0], -a.d[1], -a.d[2]}}; }
// sigmoid via exp(-|x|): no overflow, ~2 ulp
__device__ __forceinline__ float sigm(float x) {
    const float e = exp_fast(-fabsf(x));
    const float s = rcp_fast(1.0f + e);
    return x >= 0.f ? s : e * s;
}
__device__ __forceinline__ Dual3 sigm(Dual3 x) {
    const float s = sigm(x.v), ds = s * (1.0f - s);
    return Dual3{s, {ds * x.d[0], ds * x.d[1], ds * x.d[2]}};
}
__device__ __forceinline__ float expo(float x) { return exp_fast(x); }
__device__ __forceinline__ Dual3 expo(Dual3 x) {
    const float e = exp_fast(x.v);
    return Dual3{e, {e * x.d[0], e * x.d[1], e * x.d[2]}};
}
__device__ __forceinline__ float val(float x) { return x; }
__device__ __forceinline__ float val(Dual3 x) { return x.v; }
__device__ __forceinline__ float shift(float x, float m) { return x - m; }
__device__ __forceinline__ Dual3 shift(Dual3 x, float m) {
    x.v -= m;
    return x;
}

template <typename S>
__device__ __forceinline__ void softmax3(S a, S b, S c, S& oa, S& ob, S& oc) {
    const float m = fmaxf(val(a), fmaxf(val(b), val(c)));
    const S ea = expo(shift(a, m)), eb = expo(shift(b, m)), ec = expo(shift(c, m));
    const S sum = ea + eb + ec;
    oa = ea / sum;
    ob = eb / sum;
    oc = ec / sum;
}
// float version: one reciprocal (1 ulp, refined by one Newton step) shared by the three quotients
template <>
__device__ __forceinline__ void softmax3<float>(float a, float b, float c, float& oa, float& ob, float& oc) {
    const float m = fmaxf(a, fmaxf(b, c));
    const float ea = exp_fast(a - m), eb = exp_fast(b - m), ec = exp_fast(c - m);
    const float sum = ea + eb + ec;   // in [1, 3]
    float r = rcp_fast(sum);
    r = fmaf(fmaf(-sum, r, 1.0f), r, r);
    oa = ea * r;
    ob = eb * r;
    oc = ec * r;
}

// 25 logits (already x20) -> 24 weights along the SMPL tree (utils/utils.py:138-181)
template <typename S>
__device__ __forceinline__ void hsoftmax(const S (&x)[25], S (&w)[24]) {
    S sg[25];
#pragma unroll
    for (int i = 0; i < 25; ++i) sg[i] = sigm(x[i]);
    S a, b, c;
    softmax3(x[1], x[2], x[3], a, b, c);
    w[1] = sg[0] * a;
    w[2] = sg[0] * b;
    w[3] = sg[0] * c;
    w[0] = one_minus(sg[0]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // hips -> knees
        w[4 + k] = w[1 + k] * sg[4 + k];
        w[1 + k] = w[1 + k] * one_minus(sg[4 + k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // knees / spine2 -> ankles / spine3
        w[7 + k] = w[4 + k] * sg[7 + k];
        w[4 + k] = w[4 + k] * one_minus(sg[7 + k]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {   // ankles -> feet
        w[10 + k] = w[7 + k] * sg[10 + k];
        w[7 + k] = w[7 + k] * one_minus(sg[10 + k]);
    }
    softmax3(x[12], x[13], x[14], a, b, c);
    w[12] = w[9] * sg[24] * a;
    w[13] = w[9] * sg[24] * b;
    w[14] = w[9] * sg[24] * c;
    w[9] = w[9] * one_minus(sg[24]);
    w[15] = w[12] * sg[15];
    w[12] = w[12] * one_minus(sg[15]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[16 + k] = w[13 + k] * sg[16 + k];
        w[13 + k] = w[13 + k] * one_minus(sg[16 + k]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[18 + k] = w[16 + k] * sg[18 + k];
        w[16 + k] = w[16 + k] * one_minus(sg[18 + k]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[20 + k] = w[18 + k] * sg[20 + k];
        w[18 + k] = w[18 + k] * one_minus(sg[20 + k]);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        w[22 + k] = w[20 + k] * sg[22 + k];
        w[20 + k] = w[20 + k] * one_minus(sg[22 + k]);
    }
}

// In-place float version on a thread-private LDS row: in 25 raw logits, out 24 weights.  Same arithmetic
// as hsoftmax<float>(20 * logits); the 25 sigmoids run in a rolled loop so that they do not all sit in
// VGPRs at once (the fully unrolled form needs > 128 registers next to the MFMA accumulators).
__device__ __forceinline__ void hsoftmax_row(float* row) {
    float ra, rb, rc, sa, sb, sc;
    softmax3<float>(row[1] * 20.0f, row[2] * 20.0f, row[3] * 20.0f, ra, rb, rc);
    softmax3<float>(row[12] * 20.0f, row[13] * 20.0f, row[14] * 20.0f, sa, sb, sc);
#pragma unroll 1
    for (int i = 0; i < 25; ++i) row[i] = sigm(row[i] * 20.0f);
    float w[24];
    const float g0 = row[0];
    w[1] = g0 * ra;
    w[2] = g0 * rb;
    w[3] = g0 * rc;
    w[0] = 1.0f - g0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = row[4 + k];
        w[4 + k] = w[1 + k] * g;
        w[1 + k] = w[1 + k] * (1.0f - g);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = row[7 + k];
        w[7 + k] = w[4 + k] * g;
        w[4 + k] = w[4 + k] * (1.0f - g);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float g = row[10 + k];
        w[10 + k] = w[7 + k] * g;
        w[7 + k] = w[7 + k] * (1.0f - g);
    }
    const float g24 = row[24];
    w[12] = w[9] * g24 * sa;
    w[13] = w[9] * g24 * sb;
    w[14] = w[9] * g24 * sc;
    w[9] = w[9] * (1.0f - g24);
    {
        const float g = row[15];
        w[15] = w[12] * g;
        w[12] = w[12] * (1.0f - g);
    }
#pragma unroll
    for (int lvl = 0; lvl < 4; ++lvl) {   // 13,14 -> 16,17 -> 18,19 -> 20,21 -> 22,23
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = (lvl == 0 ? 13 : 14 + 2 * lvl) + k, c = 16 + 2 * lvl + k;
            const float g = row[c];
            w[c] = w[p] * g;
            w[p] = w[p] * (1.0f - g);
        }
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) row[i] = w[i];
}

// T = sum_j w_j A_j  (RFU:26).  w: 24 floats (LDS or global, stride 1), bones: [24][16] in LDS.
// Rolled on purpose: 384 FMAs with everything unrolled cost > 256 VGPRs in the callers.
__device__ __forceinline__ void blend(const float* w, const float* bones, float (&T)[16]) {
#pragma unroll
    for (int e = 0; e < 16; ++e) T[e] = 0.f;
#pragma unroll 2
    for (int jn = 0; jn < 24; ++jn) {
        const float wj = w[jn];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 b = *reinterpret_cast<const float4*>(bones + jn * 16 + c * 4);
            T[c * 4 + 0] += wj * b.x;
            T[c * 4 + 1] += wj * b.y;
            T[c * 4 + 2] += wj * b.z;
            T[c * 4 + 3] += wj * b.w;
        }
    }
}

__device__ __forceinline__ V3 apply34(const float (&T)[16], V3 p) {
    V3 o;
    o.x = T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3];
    o.y = T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7];
    o.z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11];
    return o;
}

// inverse of the upper-left 3x3 of a row-major 4x4
__device__ __forceinline__ void inv3_of44(const float (&T)[16], float (&R)[9]) {
    const float a = T[0], b = T[1], c = T[2], d = T[4], e = T[5], f = T[6], g = T[8], h = T[9], i = T[10];
    const float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    const float det = a * A + b * B + c * C;
    const float id = 1.0f / det;
    R[0] = A * id;
    R[1] = -(b * i - c * h) * id;
    R[2] = (b * f - c * e) * id;
    R[3] = B * id;
    R[4] = (a * i - c * g) * id;
    R[5] = -(a * f - c * d) * id;
    R[6] = C * id;
    R[7] = -(a * h - b * g) * id;
    R[8] = (a * e - b * d) * id;
}

// x_h = T^-1 [y;1] for T = [[R t],[0 0 0 s]] (torch.inverse of the blended 4x4, RT:393-397, 416-420)
__device__ __forceinline__ V3 inverse_affine_apply(const float (&T)[16], V3 y) {
    float R[9];
    inv3_of44(T, R);
    const float is = 1.0f / T[15];
    const float qx = y.x - T[3] * is, qy = y.y - T[7] * is, qz = y.z - T[11] * is;
    V3 o;
    o.x = R[0] * qx + R[1] * qy + R[2] * qz;
    o.y = R[3] * qx + R[4] * qy + R[5] * qz;
    o.z = R[6] * qx + R[7] * qy + R[8] * qz;
    return o;
}

// general 4x4 inverse by Gauss-Jordan with partial pivoting (J of the joint root find, RFU:416-418)
__device__ __forceinline__ void inv4(const float (&Min)[16], float (&out)[16]) {
    float a[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            a[r][c] = Min[r * 4 + c];
            a[r][4 + c] = (r == c) ? 1.0f : 0.0f;
        }
#pragma unroll
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        float best = fabsf(a[col][col]);
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            const float v = fabsf(a[r][col]);
            if (v > best) {
                best = v;
                piv = r;
            }
        }
#pragma unroll
        for (int r = col + 1; r < 4; ++r) {
            if (r == piv) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float t = a[col][c];
                    a[col][c] = a[r][c];
                    a[r][c] = t;
                }
            }
        }
        const float ip = 1.0f / a[col][col];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[col][c] *= ip;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r != col) {
                const float fct = a[r][col];
#pragma unroll
                for (int c = 0; c < 8; ++c) a[r][c] -= fct * a[col][c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = a[r][4 + c];
}

// "good Broyden" rank-1 update of J^-1 and the next step (broyden.py:69-75):
//   vT = dx^T J,  a = dx - J dg,  b = vT dg (+-eps),  J += (a/b) vT,  step = -J g
template <int D>
__device__ __forceinline__ void broyden_update(float (&J)[D * D], const float (&dx)[D], const float (&dg)[D],
                                               const float (&g)[D], float (&step)[D]) {
    float vT[D], a[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < D; ++r) s += dx[r] * J[r * D + c];
        vT[c] = s;
    }
    float b = 0.f;
#pragma unroll
    for (int r = 0; r < D; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) s += J[r * D + c] * dg[c];
        a[r] = dx[r] - s;
        b += vT[r] * dg[r];
    }
    b = (b >= 0.f) ? b + 1e-6f : b - 1e-6f;
#pragma unroll
    for (int r = 0; r < D; ++r) {
        const float u = a[r] / b;
#pragma unroll
        for (int c = 0; c < D; ++c) J[r * D + c] += u * vT[c];
    }
#pragma unroll
    for (int r = 0; r < D; ++r) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) s += J[r * D + c] * g[c];
        step[r] = -s;
    }
}

}  // namespace arah
