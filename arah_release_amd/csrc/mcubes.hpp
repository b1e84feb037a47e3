// mcubes.hpp -- marching cubes of the canonical-mesh branch on the device, without a host round trip.
//
// Reference: utils/sdf_meshing.py:83-101 hands the 256^3 SDF volume to skimage.measure.marching_cubes_lewiner on the CPU
// (models/__init__.py:203-311 then skins and rasterises the mesh).  Round 2-4 of this build ran the same extraction as ~30
// torch operations over the surface cells (meshing.marching_cubes: torch.nonzero / repeat_interleave, i.e. two device -> host
// round trips per frame, a 256^3 int64 case volume); this is that function as three kernels, triangle for triangle:
//
//   k_mcubes<false>  one workgroup per lattice row (ix, iy), one thread per cell iz: the 8 corner values, the inside / outside
//                    pattern, the triangle count of the case -> triangles per row
//   k_mc_scan        exclusive scan of the row counts (one workgroup) -> first triangle of every row, the total
//   k_mcubes<true>   the same walk again; a cell writes its triangles at row base + prefix inside the row
//   k_mc_pad         zero-fills the unused tail of the caller's buffer: triangles beyond the count are degenerate (all
//                    three corners equal), which every later stage -- skinning, projection, rasterisation -- ignores
//
// Output ORDER is the torch formulation's: cells in (ix, iy, iz) order, a cell's triangles in table order; the crossing
// point of a lattice edge is interpolated from its lower to its higher end whichever cell asks (shared vertices are
// bit-equal: the mesh is watertight), every operation rounded on its own like the tensor expression
// (pa + t (pb - pa)) * vs - 1 (no fma contraction); orientation: right-hand normals down the gradient.
// The case table (which edges of a cell form its triangles) is the CALLER's: a device copy of meshing.case_table().
#pragma once

constexpr int kMcThreads = 256;
constexpr int kMcTableWidth = 16;   // 5 triangles x 3 edge ids, padded

__device__ __constant__ signed char kMcCorner[8][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0}, {0, 1, 0}, {0, 0, 1}, {1, 0, 1}, {1, 1, 1}, {0, 1, 1}};
__device__ __constant__ signed char kMcEdge[12][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {4, 5}, {5, 6}, {6, 7}, {7, 4}, {0, 4}, {1, 5}, {2, 6}, {3, 7}};

template <bool EMIT>
__global__ __launch_bounds__(kMcThreads) void k_mcubes(const float* __restrict__ sdf, int N, float level, float vs,
                                                       const signed char* __restrict__ table, const int* __restrict__ ntri,
                                                       int* __restrict__ row_count, const int* __restrict__ row_base,
                                                       float* __restrict__ tris, int cap) {
    __shared__ int wsum[kMcThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, M = N - 1;
    const int ix = row / M, iy = row - ix * M;
    int done = 0;   // triangles of this row in front of the current chunk of cells
    for (int z0 = 0; z0 < M; z0 += kMcThreads) {
        const int iz = z0 + tid;
        const bool ok = iz < M;
        float cv[8];
        int cs = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            cv[c] = ok ? sdf[((size_t)(ix + kMcCorner[c][0]) * N + (iy + kMcCorner[c][1])) * N + (iz + kMcCorner[c][2])] : 0.f;
            cs |= (cv[c] < level ? 1 : 0) << c;
        }
        const int cnt = (ok && cs != 0 && cs != 255) ? ntri[cs] : 0;
        // exclusive prefix of cnt over the workgroup, in thread order
        int inc = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < kMcThreads / 64; ++w) {
            if (w < wave) before += wsum[w];
            total += wsum[w];
        }
        __syncthreads();
        if constexpr (EMIT) {
            const int first = row_base[row] + done + before + inc - cnt;
            for (int s = 0; s < cnt; ++s) {
                // one rounding per operation, like the tensor operations this restates (hipcc contracts * and + into fma by
                // default, and __fmul_rn / __fadd_rn are plain operators to it)
#pragma clang fp contract(off)
                if (first + s >= cap) break;
                float v[3][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int e = table[cs * kMcTableWidth + 3 * s + k];
                    int a = kMcEdge[e][0], b = kMcEdge[e][1];
                    // from the lower to the higher end of the lattice edge (the two ends differ in one coordinate)
                    if (kMcCorner[a][0] > kMcCorner[b][0] || kMcCorner[a][1] > kMcCorner[b][1] || kMcCorner[a][2] > kMcCorner[b][2]) {
                        const int t = a;
                        a = b;
                        b = t;
                    }
                    // the two end values are read again rather than taken from cv[]: a and b are run-time indices, and a
                    // register array indexed at run time lives in scratch
                    const int pa[3] = {ix + kMcCorner[a][0], iy + kMcCorner[a][1], iz + kMcCorner[a][2]};
                    const int pb[3] = {ix + kMcCorner[b][0], iy + kMcCorner[b][1], iz + kMcCorner[b][2]};
                    const float va = sdf[((size_t)pa[0] * N + pa[1]) * N + pa[2]] - level;
                    const float vb = sdf[((size_t)pb[0] * N + pb[1]) * N + pb[2]] - level;
                    const float t = fminf(fmaxf(va / (va - vb), 0.0f), 1.0f);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float d = (float)(kMcCorner[b][c] - kMcCorner[a][c]);
                        v[k][c] = ((float)pa[c] + t * d) * vs - 1.0f;
                    }
                }
                // orientation: normals point DOWN the gradient of the cell's corner values
                float grad[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 8; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) grad[k] = grad[k] + (kMcCorner[c][k] ? cv[c] : -cv[c]);
                float e1[3], e2[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    e1[c] = v[1][c] - v[0][c];
                    e2[c] = v[2][c] - v[0][c];
                }
                const float nx = e1[1] * e2[2] - e1[2] * e2[1];
                const float ny = e1[2] * e2[0] - e1[0] * e2[2];
                const float nz = e1[0] * e2[1] - e1[1] * e2[0];
                const float dn = (nx * grad[0] + ny * grad[1]) + nz * grad[2];
                const bool flip = dn > 0.f;
                float* o = tris + (size_t)(first + s) * 9;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    o[c] = v[0][c];
                    o[3 + c] = flip ? v[2][c] : v[1][c];
                    o[6 + c] = flip ? v[1][c] : v[2][c];
                }
            }
        }
        done += total;
    }
    if (!EMIT && tid == 0) row_count[row] = done;
}

// exclusive scan of n row counts by one workgroup: base[i] = sum of count[0 .. i), *total = the sum
__global__ __launch_bounds__(1024) void k_mc_scan(const int* __restrict__ count, int n, int* __restrict__ base, int* total) {
    __shared__ int part[1024];
    const int tid = threadIdx.x;
    const int per = (n + 1023) / 1024;
    const int lo = min(tid * per, n), hi = min(lo + per, n);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += count[i];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partial sums
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - s;
    for (int i = lo; i < hi; ++i) {
        base[i] = run;
        run += count[i];
    }
    if (tid == 1023) *total = part[1023];
}

__global__ void k_mc_pad(float* __restrict__ tris, const int* total, int cap) {
    const size_t first = (size_t)min(max(*total, 0), cap) * 9, end = (size_t)cap * 9;
    for (size_t i = first + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (size_t)gridDim.x * blockDim.x) tris[i] = 0.f;
}
