// meshquery.hpp -- geometric queries of the TRAINING data path against the canonical SMPL mesh (13 776 triangles):
// containment, closest point / face / squared distance, barycentric coordinates.  The reference does these on CPU
// workers per item with three third-party pieces (im2mesh/data/zju_mocap.py:461-543):
//   * im2mesh/utils/libmesh/inside_mesh.py check_mesh_contains (in the reference tree; restated 1:1 below, in double
//     precision like the original, the Cython triangle hash being only a candidate filter in front of the same tests);
//   * igl.point_mesh_squared_distance and igl.barycentric_coordinates_tri (libigl, not in the tree): exact closest
//     point on a triangle by Voronoi regions (Ericson, Real-Time Collision Detection 5.1.5), lowest face index on ties.
// One thread per query point walks all triangles, staged through LDS in chunks; double precision throughout (the
// reference's query points are float64).  Included by arah_hip.hip.
#pragma once

namespace {

constexpr int kMeshChunk = 128;     // triangles per LDS chunk (9 doubles each)
constexpr int kMeshThreads = 64;

struct MeshBox {   // libmesh's rescaling to [0.5, resolution - 0.5]^3 (inside_mesh.py:19-23,121-123)
    double scale[3], translate[3];
    double resolution;
};

// min / max of the vertices that some face uses -> scale / translate   (one workgroup)
__global__ void k_mesh_box(const float* __restrict__ verts, const int* __restrict__ faces, int n_faces, double resolution,
                           MeshBox* box) {
    __shared__ double smin[3][256], smax[3][256];
    const int tid = threadIdx.x;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = tid; i < n_faces * 3; i += blockDim.x) {
        const int v = faces[i];
        for (int c = 0; c < 3; ++c) {
            const double x = (double)verts[(size_t)v * 3 + c];
            lo[c] = fmin(lo[c], x);
            hi[c] = fmax(hi[c], x);
        }
    }
    for (int c = 0; c < 3; ++c) {
        smin[c][tid] = lo[c];
        smax[c][tid] = hi[c];
    }
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if (tid < s)
            for (int c = 0; c < 3; ++c) {
                smin[c][tid] = fmin(smin[c][tid], smin[c][tid + s]);
                smax[c][tid] = fmax(smax[c][tid], smax[c][tid + s]);
            }
        __syncthreads();
    }
    if (tid < 3) {
#pragma clang fp contract(off)
        const double sc = (resolution - 1.0) / (smax[tid][0] - smin[tid][0]);
        box->scale[tid] = sc;
        box->translate[tid] = 0.5 - sc * smin[tid][0];
        if (tid == 0) box->resolution = resolution;
    }
}

struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 dsub(D3 a, D3 b) { return D3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ double ddot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// closest point of triangle (a, b, c) to p, as barycentric weights (Ericson 5.1.5)
__device__ __forceinline__ void closest_on_triangle(D3 p, D3 a, D3 b, D3 c, double& wa, double& wb, double& wc) {
    const D3 ab = dsub(b, a), ac = dsub(c, a), ap = dsub(p, a);
    const double d1 = ddot(ab, ap), d2 = ddot(ac, ap);
    if (d1 <= 0.0 && d2 <= 0.0) { wa = 1.0; wb = 0.0; wc = 0.0; return; }
    const D3 bp = dsub(p, b);
    const double d3 = ddot(ab, bp), d4 = ddot(ac, bp);
    if (d3 >= 0.0 && d4 <= d3) { wa = 0.0; wb = 1.0; wc = 0.0; return; }
    const double vc = d1 * d4 - d3 * d2;
    if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) {
        const double v = d1 / (d1 - d3);
        wa = 1.0 - v; wb = v; wc = 0.0; return;
    }
    const D3 cp = dsub(p, c);
    const double d5 = ddot(ab, cp), d6 = ddot(ac, cp);
    if (d6 >= 0.0 && d5 <= d6) { wa = 0.0; wb = 0.0; wc = 1.0; return; }
    const double vb = d5 * d2 - d1 * d6;
    if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) {
        const double w = d2 / (d2 - d6);
        wa = 1.0 - w; wb = 0.0; wc = w; return;
    }
    const double va = d3 * d6 - d5 * d4;
    if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        wa = 0.0; wb = 1.0 - w; wc = w; return;
    }
    const double denom = 1.0 / (va + vb + vc);
    wb = vb * denom;
    wc = vc * denom;
    wa = 1.0 - wb - wc;
}

// pts [P][3] float64 or float32 (TP) -> d2 [P], face [P], closest [P][3], bary [P][3] (weights of the face's three vertices),
// inside [P] (libmesh's answer: odd crossing counts in BOTH z directions)
template <typename TP>
__global__ __launch_bounds__(kMeshThreads) void k_mesh_query(const float* __restrict__ verts, const int* __restrict__ faces,
                                                            int n_faces, const MeshBox* __restrict__ box,
                                                            const TP* __restrict__ pts, int n_pts, double* __restrict__ d2_out,
                                                            int* __restrict__ face_out, double* __restrict__ closest_out,
                                                            double* __restrict__ bary_out, uint8_t* __restrict__ inside_out) {
#pragma clang fp contract(off)   // libmesh is numpy: one rounding per operation
    __shared__ double tri[kMeshChunk][9];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_pts;
    D3 p = {0.0, 0.0, 0.0};
    if (live) p = D3{(double)pts[(size_t)i * 3], (double)pts[(size_t)i * 3 + 1], (double)pts[(size_t)i * 3 + 2]};
    const MeshBox bx = *box;
    // libmesh rescales points and triangles alike (scale * x + translate)
    const D3 q = {bx.scale[0] * p.x + bx.translate[0], bx.scale[1] * p.y + bx.translate[1], bx.scale[2] * p.z + bx.translate[2]};
    const bool in_box = 0.0 <= q.x && q.x <= bx.resolution && 0.0 <= q.y && q.y <= bx.resolution && 0.0 <= q.z &&
                        q.z <= bx.resolution;
    int n0 = 0, n1 = 0;
    double best = 1e300, bwa = 0.0, bwb = 0.0, bwc = 0.0;
    int bface = -1;
    for (int base = 0; base < n_faces; base += kMeshChunk) {
        const int cnt = min(kMeshChunk, n_faces - base);
        __syncthreads();
        for (int e = threadIdx.x; e < cnt * 9; e += blockDim.x) {
            const int f = e / 9, r = e % 9;
            tri[f][r] = (double)verts[(size_t)faces[(size_t)(base + f) * 3 + r / 3] * 3 + r % 3];
        }
        __syncthreads();
        if (!live) continue;
        for (int f = 0; f < cnt; ++f) {
            const D3 a = {tri[f][0], tri[f][1], tri[f][2]}, b = {tri[f][3], tri[f][4], tri[f][5]},
                     c = {tri[f][6], tri[f][7], tri[f][8]};
            // ---- closest point (metric space)
            double wa, wb, wc;
            closest_on_triangle(p, a, b, c, wa, wb, wc);
            const D3 cp = {wa * a.x + wb * b.x + wc * c.x, wa * a.y + wb * b.y + wc * c.y, wa * a.z + wb * b.z + wc * c.z};
            const D3 dv = dsub(p, cp);
            const double d2 = ddot(dv, dv);
            if (d2 < best) {
                best = d2;
                bface = base + f;
                bwa = wa;
                bwb = wb;
                bwc = wc;
            }
            // ---- z-ray crossing (rescaled space), inside_mesh.py:129-160 then :64-100
            if (in_box) {
                const double t1x = bx.scale[0] * a.x + bx.translate[0], t1y = bx.scale[1] * a.y + bx.translate[1],
                             t1z = bx.scale[2] * a.z + bx.translate[2];
                const double t2x = bx.scale[0] * b.x + bx.translate[0], t2y = bx.scale[1] * b.y + bx.translate[1],
                             t2z = bx.scale[2] * b.z + bx.translate[2];
                const double t3x = bx.scale[0] * c.x + bx.translate[0], t3y = bx.scale[1] * c.y + bx.translate[1],
                             t3z = bx.scale[2] * c.z + bx.translate[2];
                // A = (triangles[:, :2] - triangles[:, 2:]).T : columns t1 - t3, t2 - t3 (2-D);  y = point - t3
                const double A00 = t1x - t3x, A01 = t2x - t3x, A10 = t1y - t3y, A11 = t2y - t3y;
                const double yx = q.x - t3x, yy = q.y - t3y;
                const double detA = A00 * A11 - A01 * A10;
                if (fabs(detA) != 0.0) {
                    const double s = detA > 0.0 ? 1.0 : -1.0, ad = fabs(detA);
                    const double u = (A11 * yx - A01 * yy) * s, v = (-A10 * yx + A00 * yy) * s;
                    const double suv = u + v;
                    if (0.0 < u && u < ad && 0.0 < v && v < ad && 0.0 < suv && suv < ad) {
                        // compute_intersection_depth: v1 = t3 - t1, v2 = t2 - t1, normals = cross(v1, v2)
                        const double v1x = t3x - t1x, v1y = t3y - t1y, v1z = t3z - t1z;
                        const double v2x = t2x - t1x, v2y = t2y - t1y, v2z = t2z - t1z;
                        const double nx = v1y * v2z - v1z * v2y, ny = v1z * v2x - v1x * v2z, nz = v1x * v2y - v1y * v2x;
                        const double alpha = nx * (t1x - q.x) + ny * (t1y - q.y);
                        const double an = fabs(nz);
                        if (an != 0.0) {
                            const double sn = nz > 0.0 ? 1.0 : -1.0;
                            const double depth = t1z * an + alpha * sn;
                            if (depth >= q.z * an) ++n0;
                            else ++n1;
                        }   // n_2 == 0: depth is NaN in the reference, both comparisons false
                    }
                }
            }
        }
    }
    if (!live) return;
    d2_out[i] = best;
    face_out[i] = bface;
    if (bface >= 0) {
        const int va = faces[(size_t)bface * 3], vb = faces[(size_t)bface * 3 + 1], vc = faces[(size_t)bface * 3 + 2];
        for (int c = 0; c < 3; ++c)
            closest_out[(size_t)i * 3 + c] = bwa * (double)verts[(size_t)va * 3 + c] + bwb * (double)verts[(size_t)vb * 3 + c] +
                                             bwc * (double)verts[(size_t)vc * 3 + c];
    }
    bary_out[(size_t)i * 3] = bwa;
    bary_out[(size_t)i * 3 + 1] = bwb;
    bary_out[(size_t)i * 3 + 2] = bwc;
    inside_out[i] = (in_box && (n0 & 1) && (n1 & 1)) ? 1 : 0;
}

}  // namespace
