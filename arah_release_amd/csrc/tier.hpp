// Tiered evaluation of the eval forward (round 6): prove sigma = +0 cheaply, run loops C and D only where the image comes from.
//
// The reference canonicalises and evaluates the SDF at EVERY depth sample of EVERY ray (RT:313-380, IDR:261-396), although the
// VolSDF density (IDR:366-368) is exactly +0 in fp32 once the metric SDF exceeds 17.33 beta (1 - exp(-s / beta) rounds to 1), and
// a ray whose valid samples all have sigma = +0 renders to rgb = 0, acc = 0 exactly, whatever its samples' canonical points are:
// 82 % of the rays of the benchmark frame.  What such a ray still owes the output is `network_body_mask = any valid sample`
// (IDR:148, 232).
//
// The certificate.  A converged sample x has a canonical point x* with |LBS(x*) - (x - trans)| < 1e-5 (broyden.py:64,78).  If its
// density is > 0 then sdf(x*) <= 17.33 beta, i.e. x* lies in the canonical FAT BODY F = {y : sdf(y) <= band}, hence x lies within
// 1e-5 of the POSED fat body p(F), p(y) = LBS_w(y)(y) + trans the forward skinning with the skinning MLP's weights -- whatever path
// Broyden's iteration took to x*.  So a sample OUTSIDE p(F) is invalid or has sigma = +0.  p(F) is a fixed region of posed space per
// frame; k_occ_* below voxelise a conservative superset of it once per frame:
//   1. the SDF on a coarse lattice of the canonical box [-L, L]^3 (normalised units, L = 1.5: the body lives in [-1, 1]^3);
//   2. a coarse cell is refined when min(corners) - Lc * half_diagonal <= band, Lc = max(kOccLipMin, kOccLipSlack x the steepest
//      slope along the cell's own twelve edges): every y in F lies in a refined cell if the SDF's Lipschitz constant over the cell
//      is at most Lc;
//   3. the SDF on the f^3 sub-lattice of every refined cell; sub-points with sdf <= band + Lc * half_diagonal_fine are SELECTED:
//      every y in F is within half_diagonal_fine of a selected sub-point;
//   4. selected points are skinned forward (the skinning MLP itself, exact engine) and every voxel of the posed-space bitmap whose
//      centre lies within Lp * half_diagonal_fine + voxel half-diagonal of an image is marked, Lp = max(kOccLipPoseMin,
//      kOccLipSlack x the stretch MEASURED between the cell's own selected points): every x in p(F) falls into a marked voxel if
//      p is Lp-Lipschitz over the fine point's cube (a rigid blend has constant 1; most cells of the synthetic subject measure
//      1.0-1.2, a few where the softmax tree switches bones within a centimetre up to 3.4).
// Samples in unmarked voxels are "far".  A ray that converged in loops A+B, and every sample in a marked voxel, goes through the
// exact kernels (phase 1).  A non-surface ray is PROMOTED -- all its remaining samples evaluated exactly (phase 2) -- when one of its
// phase-1 samples has density > 0 (delta chains and the (1 - alpha + 1e-7) factors of IDR:379-390 need every valid sample), or when
// none of them converged (any-valid is then decided by the remaining ones).  A ray without a marked sample sends one WITNESS to
// phase 1, the sample nearest to the fat body (L1 distance transform of the bitmap).  Per-sample results do not depend on which
// list a sample travels in, so every ray the exact path renders non-zero is rendered bit-identically, and every other ray is 0
// with the exact mask -- PROVIDED the two Lipschitz assumptions hold and no point of the fat body lies outside [-L, L]^3 (a
// selected fine point in a boundary cell invalidates the bitmap: the body may continue where nothing was looked at): they are
// checked where they can be (tests/test_tiered.py renders both ways and compares bits; ArahCounters.n_tier_* count the work).
// An occupancy that overflowed its buffers marks itself invalid: every sample then counts as marked (one phase, the old path's work).
#pragma once

constexpr float kTierBand = 18.0f;          // sigma == +0 beyond 17.33 beta; 18 leaves room for the rounding of the SDF itself
constexpr float kOccL = 1.5f;               // canonical lattice box [-L, L]^3 in normalised coordinates
constexpr int kOccNc = 49;                  // coarse lattice points per axis (48 cells: 6.6 cm at a 2.1 m cube)
constexpr int kOccF = 3;                    // fine sub-lattice per coarse cell and axis
constexpr int kOccF3 = kOccF * kOccF * kOccF;
constexpr int kOccMaxCells = 12288;         // refined coarse cells (the synthetic subject needs ~4 000)
constexpr int kOccMaxFine = kOccMaxCells * kOccF3;
constexpr int kOccMaxVox = 1 << 22;         // bitmap voxels (512 KB of bits, 4 MB of distances)
constexpr float kOccVoxel = 0.015f;         // metres; grows when the body's box would need more than kOccMaxVox voxels
constexpr float kOccLipMin = 1.5f, kOccLipSlack = 1.25f;
constexpr float kOccLipPoseMin = 1.5f;     // floor of the forward skinning's per-cell constant (a rigid blend stretches by 1)

struct OccInfo {                 // head of the occupancy buffer (device)
    float origin[3];
    float v, inv_v;
    int dims[3];
    int n_vox;
    int valid;                   // 0: unusable -> every sample counts as marked
    int n_cells, n_fine, n_sel;  // refined cells, fine points (clamped), selected points
    int overflow;
    float band_m;                // kTierBand * beta (metres) the bitmap was built for
    float lip_pose;              // steepest |p(a) - p(b)| / |a - b| over adjacent selected fine points (assumption ii, measured)
};

struct OccBuf {                  // carved view of the caller's occupancy buffer
    OccInfo* info;
    unsigned* bits;              // [kOccMaxVox / 32]
    uint8_t* dist;               // [kOccMaxVox] L1 distance (voxels, saturated at 255) to the nearest marked voxel
    float* csdf;                 // [kOccNc^3] coarse lattice SDF (normalised units)
    float* cpts;                 // [kOccNc^3][3]
    float* cell_lip;             // [kOccMaxCells] the SDF's constant of a refined cell (negative: the cell sits on the lattice's boundary)
    float* cell_stretch;         // [kOccMaxCells] measured stretch of the forward skinning over the cell's selected points
    float* fnorm;                // [kOccMaxFine][3]
    float* fsdf;                 // [kOccMaxFine]
    int* iota;                   // [kOccMaxFine]
    float* sel_raw;              // [kOccMaxFine][3] raw canonical coordinates of the selected points
    float* sel_bar;              // [kOccMaxFine][3] their forward-skinned images (without the translation)
    int* sel_idx;                // [kOccMaxFine] fine index of a selected point
    int* sel_of;                 // [kOccMaxFine] selected slot of a fine point, -1 if not selected
    size_t bytes;
};

// ---- 1. coarse lattice coordinates
__global__ void k_occ_lattice_pts(int nc, float L, float* __restrict__ pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nc * nc * nc) return;
    const float step = 2.0f * L / (float)(nc - 1);
    const int iz = i % nc, iy = (i / nc) % nc, ix = i / (nc * nc);
    pts[(size_t)i * 3] = -L + step * (float)ix;
    pts[(size_t)i * 3 + 1] = -L + step * (float)iy;
    pts[(size_t)i * 3 + 2] = -L + step * (float)iz;
}

// bitmap geometry from the nearest-vertex grid of the posed body (its box is the vertices' box + kGridMargin)
__global__ void k_occ_begin(FrameDev fr, const GridInfo* __restrict__ grid, OccInfo* info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const GridInfo g = *grid;
    float ext[3];
    for (int a = 0; a < 3; ++a) ext[a] = (float)g.dims[a] * g.h;
    float v = kOccVoxel;
    for (int it = 0; it < 32; ++it) {
        const double n = ceil((double)ext[0] / v) * ceil((double)ext[1] / v) * ceil((double)ext[2] / v);
        if (n <= (double)kOccMaxVox) break;
        v *= 1.1f;
    }
    int nv = 1;
    for (int a = 0; a < 3; ++a) {
        info->origin[a] = g.origin[a];
        info->dims[a] = max(1, (int)ceilf(ext[a] / v));
        nv *= info->dims[a];
    }
    info->v = v;
    info->inv_v = 1.0f / v;
    info->n_vox = nv;
    info->valid = nv <= kOccMaxVox ? 1 : 0;
    info->n_cells = info->n_fine = info->n_sel = info->overflow = 0;
    info->lip_pose = 0.f;
    const float beta = fminf(fmaxf(fabsf(load_beta(fr)), 1e-6f), 1e6f);
    info->band_m = kTierBand * beta;
}

// ---- 2. refine the coarse cells that may hold a point of the fat body; write their fine sub-lattices
__global__ void k_occ_cells(FrameDev fr, const float* __restrict__ csdf, int nc, float L, OccInfo* info, float* __restrict__ cell_lip,
                            float* __restrict__ fnorm, int* __restrict__ iota) {
    const int m = nc - 1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m * m * m) return;
    const BodyConst bc = load_bc(fr);
    const float scale = sdf_scale(bc);                       // metres per normalised unit (of lengths and of the SDF alike)
    const float band_n = info->band_m / scale;
    const int cz = c % m, cy = (c / m) % m, cx = c / (m * m);
    float v[2][2][2];
    float mn = 3.4e38f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                v[i][j][k] = csdf[((size_t)(cx + i) * nc + (cy + j)) * nc + (cz + k)];
                mn = fminf(mn, v[i][j][k]);
            }
    const float step = 2.0f * L / (float)m;
    float sl = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            sl = fmaxf(sl, fabsf(v[1][a][b] - v[0][a][b]));
            sl = fmaxf(sl, fabsf(v[a][1][b] - v[a][0][b]));
            sl = fmaxf(sl, fabsf(v[a][b][1] - v[a][b][0]));
        }
    const float lip = fmaxf(kOccLipMin, kOccLipSlack * sl / step);
    const float half = step * 0.8660254f;
    if (!(mn - lip * half <= band_n)) return;                 // (NaN corners refine)
    const int slot = atomicAdd(&info->n_cells, 1);
    if (slot >= kOccMaxCells) {
        info->overflow = 1;
        return;
    }
    // a cell on the lattice's own boundary travels with a negative sign: a SELECTED fine point in it means the fat body may continue
    // outside [-L, L]^3, where nothing was looked at (k_occ_select then drops the bitmap for this frame).  The coarse test alone
    // says little there: far from the body the emitted SIREN is steep and many boundary cells are refined for their slope only.
    const bool edge = cx == 0 || cy == 0 || cz == 0 || cx == m - 1 || cy == m - 1 || cz == m - 1;
    cell_lip[slot] = edge ? -lip : lip;
    const float fs = step / (float)kOccF;
    const float x0 = -L + step * (float)cx, y0 = -L + step * (float)cy, z0 = -L + step * (float)cz;
    for (int j = 0; j < kOccF3; ++j) {
        const int a = j / (kOccF * kOccF), b = (j / kOccF) % kOccF, d = j % kOccF;
        const size_t o = (size_t)slot * kOccF3 + j;
        fnorm[o * 3] = x0 + fs * ((float)a + 0.5f);
        fnorm[o * 3 + 1] = y0 + fs * ((float)b + 0.5f);
        fnorm[o * 3 + 2] = z0 + fs * ((float)d + 0.5f);
        iota[o] = (int)o;
    }
}

__global__ void k_occ_fix(OccInfo* info) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (info->overflow) info->valid = 0;
    info->n_cells = min(info->n_cells, kOccMaxCells);
    info->n_fine = info->n_cells * kOccF3;
}

// ---- 3. select the fine points that may lie within half a fine diagonal of the fat body
__global__ void k_occ_select(FrameDev fr, int nc, float L, OccInfo* info, const float* __restrict__ cell_lip,
                             const float* __restrict__ fnorm, const float* __restrict__ fsdf, float* __restrict__ sel_raw,
                             int* __restrict__ sel_idx, int* __restrict__ sel_of) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = info->n_fine;
    const BodyConst bc = load_bc(fr);
    bool keep = false;
    V3 raw = V3{0.f, 0.f, 0.f};
    if (i < n) {
        const float scale = sdf_scale(bc);
        const float band_n = info->band_m / scale;
        const float step = 2.0f * L / (float)(nc - 1) / (float)kOccF;
        const float half = step * 0.8660254f;
        const float s = fsdf[i];
        const float cl = cell_lip[i / kOccF3];
        keep = !(s - fabsf(cl) * half > band_n);   // (a NaN value selects)
        if (keep && cl < 0.f) info->valid = 0;     // assumption (iii) cannot be checked: the body reaches the lattice's boundary
        raw = unnormalize_pt(bc, V3{fnorm[(size_t)i * 3], fnorm[(size_t)i * 3 + 1], fnorm[(size_t)i * 3 + 2]});
    }
    const unsigned long long mk = __ballot(keep);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && mk) base = atomicAdd(&info->n_sel, __popcll(mk));
    base = __shfl(base, 0);
    if (keep) {
        const size_t o = (size_t)(base + __popcll(mk & ((1ull << lane) - 1ull)));
        sel_raw[o * 3] = raw.x;
        sel_raw[o * 3 + 1] = raw.y;
        sel_raw[o * 3 + 2] = raw.z;
        sel_idx[o] = i;
        sel_of[i] = (int)o;
    }
}

// The forward skinning's constant, MEASURED per refined cell on its own selected fine points (one wave per cell, lane = fine point,
// every pair of the cell): the steepest |p(a) - p(b)| / |a - b|.  The dilation radius of the cell's images is max(kOccLipPoseMin,
// kOccLipSlack x that) half-diagonals.  On the synthetic subject 99 % of the cells measure <= 1.6 and 4-11 of 4 200, where the
// hierarchical softmax switches bones within a centimetre, 2-10: a fixed constant of 2 was not a bound there (it held by the other
// margins), a fixed 10 would have swollen the band everywhere.  The steepest value of the frame is kept in the header.
// (The SDF's constant stays the coarse cell's own slope x kOccLipSlack, floor kOccLipMin.  Raising it by the slopes between the
// fine points was tried: far outside the body the emitted SIREN is steep (slopes of 10-15 between [-1, 1]^3 and the lattice's
// boundary), boundary cells then select fine points and the bitmap drops itself in a third of the frames -- 12.9 -> 15.8 ms.)
// The SDF's constant of an INTERIOR refined cell, raised by what its own fine points show: kOccLipSlack x the steepest
// |sdf(a) - sdf(b)| / |a - b| over the cell's 27 fine points (a SIREN detail finer than the coarse cell).  Cells on the lattice's
// boundary keep their coarse constant (see the note above).
__global__ __launch_bounds__(256) void k_occ_cell_slope(int nc, float L, OccInfo* info, const float* __restrict__ fsdf,
                                                        float* __restrict__ cell_lip) {
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cell >= info->n_cells) return;
    const float fs = 2.0f * L / (float)(nc - 1) / (float)kOccF;   // normalised units: slopes are the same in metres
    const bool on = lane < kOccF3;
    const float mine = on ? fsdf[(size_t)cell * kOccF3 + lane] : 0.f;
    const int a = lane / (kOccF * kOccF), b = (lane / kOccF) % kOccF, d = lane % kOccF;
    float worst = 0.f;
    for (int o = 1; o < kOccF3; ++o) {
        const int other = (lane + o) % kOccF3;
        const float v = __shfl(mine, other);
        const int a2 = other / (kOccF * kOccF), b2 = (other / kOccF) % kOccF, d2 = other % kOccF;
        const float dist = fs * sqrtf((float)((a - a2) * (a - a2) + (b - b2) * (b - b2) + (d - d2) * (d - d2)));
        if (on) worst = fmaxf(worst, fabsf(v - mine) / dist);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
    if (lane == 0) {
        const float cl = cell_lip[cell];
        if (cl > 0.f) cell_lip[cell] = fmaxf(cl, kOccLipSlack * worst);
    }
}

__global__ __launch_bounds__(256) void k_occ_cell_stretch(FrameDev fr, int nc, float L, OccInfo* info, const int* __restrict__ sel_of,
                                                          const float* __restrict__ sel_bar, float* __restrict__ cell_stretch) {
    const int lane = threadIdx.x & 63;
    const int cell = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (cell >= info->n_cells) return;
    const BodyConst bc = load_bc(fr);
    const float fs = 2.0f * L / (float)(nc - 1) / (float)kOccF * sdf_scale(bc);   // metres between adjacent fine points
    const int o = lane < kOccF3 ? sel_of[(size_t)cell * kOccF3 + lane] : -1;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (o >= 0) {
        px = sel_bar[(size_t)o * 3];
        py = sel_bar[(size_t)o * 3 + 1];
        pz = sel_bar[(size_t)o * 3 + 2];
    }
    const int a = lane / (kOccF * kOccF), b = (lane / kOccF) % kOccF, d = lane % kOccF;
    float worst = 0.f;
    for (int k = 1; k < kOccF3; ++k) {
        const int other = (lane + k) % kOccF3;
        const int o2 = __shfl(o, other);
        const float qx = __shfl(px, other), qy = __shfl(py, other), qz = __shfl(pz, other);
        const int a2 = other / (kOccF * kOccF), b2 = (other / kOccF) % kOccF, d2 = other % kOccF;
        const float dist = fs * sqrtf((float)((a - a2) * (a - a2) + (b - b2) * (b - b2) + (d - d2) * (d - d2)));
        if (o >= 0 && o2 >= 0)
            worst = fmaxf(worst, sqrtf((px - qx) * (px - qx) + (py - qy) * (py - qy) + (pz - qz) * (pz - qz)) / dist);
    }
    for (int off = 32; off > 0; off >>= 1) worst = fmaxf(worst, __shfl_xor(worst, off));
    if (lane == 0) {
        cell_stretch[cell] = worst;
        if (worst > 0.f) atomicMax(reinterpret_cast<int*>(&info->lip_pose), __float_as_int(worst));   // non-negative floats order like their bits
        if (!(worst == worst)) info->valid = 0;
    }
}

// ---- 4. mark the voxels around the posed images of the selected points
__global__ void k_occ_mark(FrameDev fr, int nc, float L, OccInfo* info, const float* __restrict__ sel_bar, const int* __restrict__ sel_idx,
                           const float* __restrict__ cell_stretch, unsigned* __restrict__ bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= info->n_sel || !info->valid) return;
    const BodyConst bc = load_bc(fr);
    const float scale = sdf_scale(bc);
    const float half_m = 2.0f * L / (float)(nc - 1) / (float)kOccF * 0.8660254f * scale;
    const float v = info->v, inv_v = info->inv_v;
    const float lip_pose = fmaxf(kOccLipPoseMin, kOccLipSlack * cell_stretch[sel_idx[i] / kOccF3]);   // this cell's measured constant
    const float rad = (lip_pose * half_m + 1e-4f) * inv_v + 0.8660254f;     // voxel units, centre to point
    // position in voxel units relative to voxel centres: centre of voxel k sits at k + 0.5
    const float px = (sel_bar[(size_t)i * 3] + bc.trans[0] - info->origin[0]) * inv_v - 0.5f;
    const float py = (sel_bar[(size_t)i * 3 + 1] + bc.trans[1] - info->origin[1]) * inv_v - 0.5f;
    const float pz = (sel_bar[(size_t)i * 3 + 2] + bc.trans[2] - info->origin[2]) * inv_v - 0.5f;
    if (!(px == px && py == py && pz == pz)) {   // a non-finite image: nothing can be said about this frame
        info->valid = 0;
        return;
    }
    (void)v;
    const int dx = info->dims[0], dy = info->dims[1], dz = info->dims[2];
    const int z0 = max(0, (int)ceilf(pz - rad)), z1 = min(dz - 1, (int)floorf(pz + rad));
    const int y0 = max(0, (int)ceilf(py - rad)), y1 = min(dy - 1, (int)floorf(py + rad));
    for (int z = z0; z <= z1; ++z)
        for (int y = y0; y <= y1; ++y) {
            const float rem = rad * rad - ((float)z - pz) * ((float)z - pz) - ((float)y - py) * ((float)y - py);
            if (rem < 0.f) continue;
            const float e = sqrtf(rem);
            const int xa = max(0, (int)ceilf(px - e)), xb = min(dx - 1, (int)floorf(px + e));
            if (xa > xb) continue;
            const long long row = ((long long)z * dy + y) * dx;
            long long b0 = row + xa, b1 = row + xb;
            for (long long w = b0 >> 5; w <= (b1 >> 5); ++w) {
                const int lo = (int)max(b0 - (w << 5), 0ll), hi = (int)min(b1 - (w << 5), 31ll);
                const unsigned mask = (hi == 31 ? 0xffffffffu : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u);
                if ((bits[w] & mask) != mask) atomicOr(&bits[w], mask);
            }
        }
}

// ---- L1 distance transform of the bitmap (voxels, saturated): one thread per line, three passes
__global__ void k_occ_dist_x(const OccInfo* info, const unsigned* __restrict__ bits, uint8_t* __restrict__ dist) {
    const int line = blockIdx.x * blockDim.x + threadIdx.x;
    const int dx = info->dims[0], dy = info->dims[1], dz = info->dims[2];
    if (line >= dy * dz) return;
    const long long row = (long long)line * dx;
    int d = 255;
    for (int x = 0; x < dx; ++x) {
        const long long b = row + x;
        d = ((bits[b >> 5] >> (b & 31)) & 1u) ? 0 : min(255, d + 1);
        dist[b] = (uint8_t)d;
    }
    d = 255;
    for (int x = dx - 1; x >= 0; --x) {
        const long long b = row + x;
        d = min((int)dist[b], min(255, d + 1));
        dist[b] = (uint8_t)d;
    }
}
template <int AXIS>   // 1: along y, 2: along z
__global__ void k_occ_dist_yz(const OccInfo* info, uint8_t* __restrict__ dist) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int dx = info->dims[0], dy = info->dims[1], dz = info->dims[2];
    const int n_lines = AXIS == 1 ? dx * dz : dx * dy;
    if (t >= n_lines) return;
    // consecutive threads walk consecutive x: coalesced byte rows
    const int x = t % dx, o = t / dx;
    const int len = AXIS == 1 ? dy : dz;
    const long long stride = AXIS == 1 ? dx : (long long)dx * dy;
    const long long base = AXIS == 1 ? (long long)o * dx * dy + x : (long long)o * dx + x;
    int d = 255;
    for (int k = 0; k < len; ++k) {
        const long long b = base + k * stride;
        d = min((int)dist[b], min(255, d + 1));
        dist[b] = (uint8_t)d;
    }
    d = 255;
    for (int k = len - 1; k >= 0; --k) {
        const long long b = base + k * stride;
        d = min((int)dist[b], min(255, d + 1));
        dist[b] = (uint8_t)d;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the tiers of one frame
// ---------------------------------------------------------------------------------------------------------------------
enum { TS_NONE = 0, TS_PHASE1 = 1, TS_PENDING = 2, TS_PHASE2 = 3, TS_WITNESS = 4 /* between the two classify passes only */ };
enum { TC_N1 = 0, TC_HEAD1, TC_ND1, TC_N2, TC_HEAD2, TC_ND2, TC_NSHADE, TC_NWIT, TC_COUNT = 16 };

typedef TierStatsRaw TierStats;   // tail of Counters (ArahCounters.n_tier_*)

// marked?  (+ the distance byte).  Samples outside the bitmap's box, and every sample of an invalid occupancy, count as marked.
__device__ __forceinline__ bool occ_lookup(const OccInfo& oi, const unsigned* bits, const uint8_t* dist, V3 p, int& d) {
    d = 0;
    if (!oi.valid) return true;
    const float fx = (p.x - oi.origin[0]) * oi.inv_v, fy = (p.y - oi.origin[1]) * oi.inv_v, fz = (p.z - oi.origin[2]) * oi.inv_v;
    const int x = (int)floorf(fx), y = (int)floorf(fy), z = (int)floorf(fz);
    if (!(fx >= 0.f && fy >= 0.f && fz >= 0.f) || x >= oi.dims[0] || y >= oi.dims[1] || z >= oi.dims[2]) return true;
    const long long b = ((long long)z * oi.dims[1] + y) * oi.dims[0] + x;
    d = dist[b];
    return (bits[b >> 5] >> (b & 31)) & 1u;
}

// Rays that cannot meet the surface.  A ray on which loops A+B converge has a surface point x(z), z in [near, far], with
// |sdf| < 1e-5 at its canonical point and |LBS - x| < 1e-5 (RFU:426-457, RT:266): x(z) lies in a marked voxel.  A ray whose
// segment [near, far] walks through no voxel that is marked or shares a face with a marked one (distance byte <= 1: the slack
// covers a voxel walk that clips a corner differently from floor()) is therefore a certain miss: skip[ray] = 1, loops A and B are
// not run for it, the tracer reports converged = 0, start = near like the reference does for a ray that left the box (RT:235,
// 274-277).  Segments that leave the bitmap's box, and every ray of an invalid occupancy, are traced.
__global__ void k_tier_rays(int n, RaySet rs, const float* __restrict__ near_far, const OccInfo* __restrict__ info,
                            const uint8_t* __restrict__ dist, uint8_t* __restrict__ skip, TierStats* stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool sk = false;
    if (i < n) {
        const OccInfo oi = *info;
        const float t0 = near_far[i * 2] - 1e-4f, t1 = near_far[i * 2 + 1] + 1e-4f;
        if (oi.valid && t0 < t1) {
            const int cam = i / rs.rays_per_cam;
            const float o[3] = {(rs.cam_loc[cam * 3] - oi.origin[0]) * oi.inv_v, (rs.cam_loc[cam * 3 + 1] - oi.origin[1]) * oi.inv_v,
                                (rs.cam_loc[cam * 3 + 2] - oi.origin[2]) * oi.inv_v};
            const float d[3] = {rs.dirs[i * 3] * oi.inv_v, rs.dirs[i * 3 + 1] * oi.inv_v, rs.dirs[i * 3 + 2] * oi.inv_v};
            // both ends inside the box (the segment is then inside: the box is convex)
            bool inside = true;
            int c[3];
            float tmax[3], tdel[3];
            int stp[3];
            for (int a = 0; a < 3; ++a) {
                const float p0 = o[a] + d[a] * t0, p1 = o[a] + d[a] * t1;
                inside = inside && p0 >= 0.f && p1 >= 0.f && p0 < (float)oi.dims[a] && p1 < (float)oi.dims[a];
                c[a] = (int)floorf(p0);
                stp[a] = d[a] > 0.f ? 1 : -1;
                if (d[a] != 0.f) {
                    const float nb = d[a] > 0.f ? (float)(c[a] + 1) : (float)c[a];
                    tmax[a] = t0 + (nb - p0) / d[a];
                    tdel[a] = fabsf(1.0f / d[a]);
                } else {
                    tmax[a] = 3.4e38f;
                    tdel[a] = 3.4e38f;
                }
            }
            if (inside) {
                sk = true;
                for (int it = 0; it < 4096; ++it) {
                    if (c[0] < 0 || c[1] < 0 || c[2] < 0 || c[0] >= oi.dims[0] || c[1] >= oi.dims[1] || c[2] >= oi.dims[2]) {
                        sk = false;   // rounding walked it out of the box: trace it
                        break;
                    }
                    if (dist[((long long)c[2] * oi.dims[1] + c[1]) * oi.dims[0] + c[0]] <= 1) {
                        sk = false;
                        break;
                    }
                    const int a = tmax[0] <= tmax[1] ? (tmax[0] <= tmax[2] ? 0 : 2) : (tmax[1] <= tmax[2] ? 1 : 2);
                    if (tmax[a] > t1) break;
                    c[a] += stp[a];
                    tmax[a] += tdel[a];
                    if (it == 4095) sk = false;
                }
            }
        }
        skip[i] = sk ? 1 : 0;
    }
    const unsigned long long m = __ballot(sk);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&stats->rays_untraced, (unsigned long long)__popcll(m));
}

// A workgroup's share of a list append: every wave brings n (wave-uniform); one device atomic per workgroup and batch -- a single
// device-scope counter takes ~90 atomics / us, and one per ray made these two kernels 3-4 ms each (155 k rays).
constexpr int kTierWaves = 16;
__device__ __forceinline__ int tier_block_base(int n, int wave, int lane, int* count, int* sh /* [kTierWaves + 1] */) {
    if (lane == 0) sh[wave] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < kTierWaves; ++k) {
            const int c = sh[k];
            sh[k] = tot;
            tot += c;
        }
        sh[kTierWaves] = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    const int base = sh[kTierWaves] + sh[wave];
    __syncthreads();
    return base;
}
// per-wave tallies -> one atomic per counter and workgroup
template <int NS>
__device__ __forceinline__ void tier_flush_stats(const unsigned (&loc)[NS], unsigned long long* const (&dst)[NS], int wave, int lane,
                                                 unsigned* sh /* [kTierWaves][NS] */) {
    if (lane == 0)
        for (int k = 0; k < NS; ++k) sh[wave * NS + k] = loc[k];
    __syncthreads();
    if (threadIdx.x < NS) {
        unsigned long long tot = 0;
        for (int w = 0; w < kTierWaves; ++w) tot += sh[w * NS + threadIdx.x];
        if (tot) atomicAdd(dst[threadIdx.x], tot);
    }
}

// One wave per ray, one lane per depth sample (two for n_steps > 64).  state[q]: TS_* of every sample; phase-1 samples -> list1.
// Two passes so that the WITNESSES head the list: they are the samples farthest from the body, the slow ones of Broyden's
// iteration (up to 51 evaluations; near the body two) -- loop C's resident kernel takes the list front to back, and long jobs
// first is what keeps its tail short.  PASS 0 classifies and appends the witnesses, PASS 1 appends the rest.
template <int PASS>
__global__ __launch_bounds__(kTierWaves * 64) void k_tier_classify(int n, int S, RaySet rs, const uint8_t* __restrict__ conv,
                                                        const float* __restrict__ z, uint8_t* __restrict__ state,
                                                        const OccInfo* __restrict__ info, const unsigned* __restrict__ bits,
                                                        const uint8_t* __restrict__ dist, int* __restrict__ list1, int* count1,
                                                        TierStats* stats) {
    __shared__ int sh_base[kTierWaves + 1];
    __shared__ unsigned sh_stats[kTierWaves * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const OccInfo oi = *info;
    unsigned loc[4] = {0u, 0u, 0u, 0u};   // rays, surface rays, witnesses, phase-1 samples
    for (int r0 = blockIdx.x * kTierWaves; r0 < n; r0 += gridDim.x * kTierWaves) {
        const int ray = r0 + wave;
        const bool live = ray < n;
        const bool surf = live && conv[ray] != 0;
        int st[2] = {TS_NONE, TS_NONE};
        bool witness = false;
        if (PASS == 0) {
            int dd[2] = {255, 255};
            bool any_marked = false;
            for (int it = 0; it < 2; ++it) {
                const int s = lane + it * 64;
                if (live && s < S) {
                    const size_t q = (size_t)ray * S + s;
                    if (state[q]) {
                        int d;
                        const bool mk = occ_lookup(oi, bits, dist, ray_point(rs, ray, z[q]), d);
                        // a surface ray's samples are all evaluated (its delta chain needs every valid one), but those outside
                        // the fat body -- most of the n_far samples in front of the surface -- are CERTIFIED sigma = +0 like a
                        // witness: convergence only, no density pass (and they head the list with the witnesses)
                        if (surf) st[it] = mk ? TS_PHASE1 : TS_WITNESS;
                        else st[it] = mk ? TS_PHASE1 : TS_PENDING;
                        dd[it] = d;
                    }
                }
                any_marked = any_marked || __ballot(st[it] == TS_PHASE1) != 0ull;
            }
            if (live && !surf && !any_marked) {   // the pending sample nearest to the fat body witnesses "any valid"
                unsigned key = 0xffffffffu;
                for (int it = 0; it < 2; ++it)
                    if (st[it] == TS_PENDING) key = min(key, ((unsigned)dd[it] << 16) | (unsigned)(lane + it * 64));
                unsigned best = key;
                for (int o = 32; o > 0; o >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, o));
                if (best != 0xffffffffu) {
                    const int ws = (int)(best & 0xffffu);
                    if ((ws & 63) == lane) st[ws >> 6] = TS_WITNESS;
                    witness = true;
                    // ... and two more, a quarter of the ray in from either end.  Whether Broyden's iteration converges is
                    // mostly a property of the REGION (far from the body the skinning field is unlike the nearest vertex's
                    // weights), so a second witness next to a failed one fails too, one elsewhere on the ray often does not:
                    // with the nearest sample alone 2.6-23 % of these rays had to be promoted (63 slow samples each, 25-75
                    // skinning evaluations per ray of the frame), with three witnesses 0.8-18 % (oracle study, profiles/
                    // r06_tier_design_study.txt: 18-49 evaluations per ray including the extra witnesses)
                    for (int it = 0; it < 2; ++it) {
                        const int sidx = lane + it * 64;
                        if (st[it] == TS_PENDING && (sidx == S / 4 || sidx == (3 * S) / 4)) st[it] = TS_WITNESS;
                    }
                }
            }
        } else {
            for (int it = 0; it < 2; ++it) {
                const int s = lane + it * 64;
                if (live && s < S) st[it] = state[(size_t)ray * S + s];
            }
        }
        // what this pass appends: the witnesses (PASS 0) / the other phase-1 samples (PASS 1)
        int n1 = 0, n_p1 = 0;
        unsigned long long m1[2];
        for (int it = 0; it < 2; ++it) {
            m1[it] = __ballot(st[it] == (PASS == 0 ? TS_WITNESS : TS_PHASE1));
            n1 += __popcll(m1[it]);
            n_p1 += __popcll(__ballot(st[it] == TS_PHASE1 || st[it] == TS_WITNESS));
        }
        int base = tier_block_base(n1, wave, lane, count1, sh_base);
        if (live && PASS == 0) {
            loc[0] += 1u;
            loc[1] += surf ? 1u : 0u;
            loc[2] += witness ? 1u : 0u;
            loc[3] += (unsigned)n_p1;
        }
        for (int it = 0; it < 2; ++it) {
            const int s = lane + it * 64;
            if (live && s < S) {
                const size_t q = (size_t)ray * S + s;
                if (PASS == 0) state[q] = (uint8_t)st[it];
                else if (st[it] == TS_WITNESS) state[q] = TS_PHASE1;
                if (st[it] == (PASS == 0 ? TS_WITNESS : TS_PHASE1)) list1[base + __popcll(m1[it] & ((1ull << lane) - 1ull))] = (int)q;
            }
            base += __popcll(m1[it]);
        }
    }
    if (PASS == 0) {
        unsigned long long* const dst[4] = {&stats->rays, &stats->rays_surface, &stats->witnesses, &stats->samples_p1};
        tier_flush_stats<4>(loc, dst, wave, lane, sh_stats);
    }
}

// RT:447-461, 549-555 for the samples of one phase: normalise the solution, converged = |g|_best < thr; the converged ones go
// on to the density pass -- except the CERTIFIED ones, list[0 .. *n_certified): witnesses (they head phase 1's list) and every
// phase-2 sample lie outside the posed fat body, their density is +0 by the certificate that let their ray's other samples go
// unevaluated, and only their convergence is asked for (the ray's mask, the delta chain and the (1 - alpha + 1e-7) factors of
// a promoted ray): sigma = +0 is written, the SDF is not evaluated (29 % of the density pass's samples on the benchmark).
__global__ __launch_bounds__(256) void k_tier_finalize(FrameDev fr, const int* __restrict__ list, const int* count,
                                                        const int* n_certified, const float* __restrict__ err_best,
                                                        float* __restrict__ pts, uint8_t* __restrict__ mask,
                                                        f32x4* __restrict__ shaded, int* __restrict__ dens_list, int* dens_count) {
    const int n = *count;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int n_cert = *n_certified;
    const BodyConst bc = load_bc(fr);
    for (int i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + threadIdx.x;
        bool ok = false;
        int q = -1;
        if (i < n) {
            q = list[i];
            const V3 xn = normalize_pt(bc, V3{pts[(size_t)q * 3], pts[(size_t)q * 3 + 1], pts[(size_t)q * 3 + 2]});
            pts[(size_t)q * 3] = xn.x;
            pts[(size_t)q * 3 + 1] = xn.y;
            pts[(size_t)q * 3 + 2] = xn.z;
            ok = err_best[q] < kRootThresh;
            mask[q] = ok ? 1 : 0;
            if (i < n_cert) {
                if (ok) shaded[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                ok = false;
            }
        }
        append_ids(ok, q, dens_list, dens_count);
    }
}

// One wave per ray after phase 1: promote the non-surface rays whose phase-1 samples show density > 0, or of which none
// converged; their pending samples -> list2.
// PASS 0 decides and appends the rays promoted because NONE of their phase-1 samples converged (far from the body: the slow
// samples, see k_tier_classify), PASS 1 appends the rays promoted for a density > 0.
template <int PASS>
__global__ __launch_bounds__(kTierWaves * 64) void k_tier_promote(int n, int S, const uint8_t* __restrict__ conv, uint8_t* __restrict__ state,
                                                       const uint8_t* __restrict__ mask, const f32x4* __restrict__ shaded,
                                                       int* __restrict__ list2, int* count2, uint8_t* __restrict__ ray_tier,
                                                       TierStats* stats) {
    __shared__ int sh_base[kTierWaves + 1];
    __shared__ unsigned sh_stats[kTierWaves * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned loc[4] = {0u, 0u, 0u, 0u};   // promoted rays, skipped rays, phase-2 samples, samples never evaluated
    for (int r0 = blockIdx.x * kTierWaves; r0 < n; r0 += gridDim.x * kTierWaves) {
        const int ray = r0 + wave;
        const bool live = ray < n;
        const bool surf = live && conv[ray] != 0;
        int st[2] = {TS_NONE, TS_NONE};
        bool pos = false, ok = false;
        for (int it = 0; it < 2; ++it) {
            const int s = lane + it * 64;
            if (live && s < S) {
                const size_t q = (size_t)ray * S + s;
                st[it] = state[q];
                if (st[it] == TS_PHASE1 && mask[q]) {
                    ok = true;
                    pos = pos || shaded[q][3] > 0.f;
                }
            }
        }
        const bool any_pos = __ballot(pos) != 0ull, any_ok = __ballot(ok) != 0ull;
        const bool promote = live && !surf && (any_pos || !any_ok);
        const bool mine = promote && (PASS == 0 ? !any_ok : any_ok);   // the rays this pass appends
        unsigned long long m2[2];
        int n2 = 0, n_pend = 0;
        for (int it = 0; it < 2; ++it) {
            const unsigned long long pend = __ballot(st[it] == TS_PENDING);
            n_pend += __popcll(pend);
            m2[it] = mine ? pend : 0ull;
            n2 += __popcll(m2[it]);
        }
        int base = tier_block_base(n2, wave, lane, count2, sh_base);
        if (live && PASS == 0) {
            if (lane == 0 && ray_tier) ray_tier[ray] = surf ? 1 : (promote ? 2 : 0);
            loc[0] += promote ? 1u : 0u;
            loc[1] += (!promote && !surf) ? 1u : 0u;
            loc[2] += promote ? (unsigned)n_pend : 0u;
            loc[3] += promote ? 0u : (unsigned)n_pend;
        }
        for (int it = 0; it < 2; ++it) {
            const int s = lane + it * 64;
            if (live && s < S && mine && st[it] == TS_PENDING) {
                const size_t q = (size_t)ray * S + s;
                state[q] = TS_PHASE2;
                list2[base + __popcll(m2[it] & ((1ull << lane) - 1ull))] = (int)q;
            }
            base += __popcll(m2[it]);
        }
    }
    if (PASS == 0) {
        unsigned long long* const dst[4] = {&stats->rays_promoted, &stats->rays_skipped, &stats->samples_p2, &stats->samples_skipped};
        tier_flush_stats<4>(loc, dst, wave, lane, sh_stats);
    }
}

// phase-2 share of the work counters: snapshot before (mode 0), difference after (mode 1)
__global__ void k_tier_snap(const unsigned long long* n_canon, const unsigned long long* n_density, unsigned long long* snap,
                            unsigned long long* canon_p2, unsigned long long* density_p2, int mode) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (mode == 0) {
        snap[0] = *n_canon;
        snap[1] = *n_density;
    } else {
        *canon_p2 += *n_canon - snap[0];
        *density_p2 += *n_density - snap[1];
    }
}

// tests: per ray, does any valid sample carry density > 0 ?  (what the exact path would have to render)
__global__ void k_tier_ray_sigma(int n, int S, const uint8_t* __restrict__ mask, const f32x4* __restrict__ shaded, uint8_t* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bool pos = false;
    for (int s = 0; s < S; ++s) {
        const size_t q = (size_t)i * S + s;
        pos = pos || (mask[q] && shaded[q][3] > 0.f);
    }
    out[i] = pos ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// The canonical-mesh branch's lattice, evaluated only where the level set can be (arah_sdf_grid_band)
// ---------------------------------------------------------------------------------------------------------------------
// utils/sdf_meshing.py:44-57 evaluates the SDF at all N^3 = 16.8 M lattice points of [-1,1]^3 and hands the volume to marching
// cubes (:95), which only LOOKS at the values of cells whose corners change sign.  The same Lipschitz test that refines the
// occupancy's lattice says where that can happen: a coarse cell (1/32 of the box, 6.6 cm) may hold a zero of the SDF only if
// min(corners) - Lc * half_diagonal <= 0 <= max(corners) + Lc * half_diagonal, Lc = max(kOccLipMin, kOccLipSlack x its steepest
// edge slope).  Lattice points in such cells AND in their 26 neighbours (so that every corner of a lattice cell with a sign
// change is one of them) are evaluated exactly; every other point gets the value of its coarse cell's first corner -- the right
// SIGN, which is all marching cubes asks of a cell without a sign change.  The triangle soup is the full lattice's, triangle
// for triangle and bit for bit (tests/test_meshing.py), at ~6 % of the evaluations.
constexpr int kBandNc = 33;   // coarse lattice points per axis over [-1, 1]

__global__ void k_band_cells(const float* __restrict__ csdf, uint8_t* __restrict__ flag) {
    constexpr int nc = kBandNc, m = nc - 1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m * m * m) return;
    const int cz = c % m, cy = (c / m) % m, cx = c / (m * m);
    float v[2][2][2], mn = 3.4e38f, mx = -3.4e38f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int k = 0; k < 2; ++k) {
                v[i][j][k] = csdf[((size_t)(cx + i) * nc + (cy + j)) * nc + (cz + k)];
                mn = fminf(mn, v[i][j][k]);
                mx = fmaxf(mx, v[i][j][k]);
            }
    const float step = 2.0f / (float)m;
    float sl = 0.f;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            sl = fmaxf(sl, fabsf(v[1][a][b] - v[0][a][b]));
            sl = fmaxf(sl, fabsf(v[a][1][b] - v[a][0][b]));
            sl = fmaxf(sl, fabsf(v[a][b][1] - v[a][b][0]));
        }
    const float reach = fmaxf(kOccLipMin, kOccLipSlack * sl / step) * step * 0.8660254f;
    flag[c] = (!(mn - reach > 0.f) && !(mx + reach < 0.f)) ? 1 : 0;   // (NaN corners flag the cell)
}

__global__ void k_band_dilate(const uint8_t* __restrict__ flag, uint8_t* __restrict__ out) {
    constexpr int m = kBandNc - 1;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= m * m * m) return;
    const int cz = c % m, cy = (c / m) % m, cx = c / (m * m);
    uint8_t f = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int x = cx + dx, y = cy + dy, z = cz + dz;
                if (x >= 0 && y >= 0 && z >= 0 && x < m && y < m && z < m) f |= flag[((size_t)x * m + y) * m + z];
            }
    out[c] = f;
}

// every lattice point: in the band -> on the list (evaluated next); else the value of its coarse cell's first corner
__global__ __launch_bounds__(1024) void k_band_fill(int N, const float* __restrict__ csdf, const uint8_t* __restrict__ flag2,
                                                    float* __restrict__ sdf, int* __restrict__ list, int* count) {
    constexpr int nc = kBandNc, m = nc - 1;
    __shared__ int wave_cnt[16];
    __shared__ int block_base;
    const long long n = (long long)N * N * N;
    const long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool band = false;
    if (id < n) {
        const int iz = (int)(id % N), iy = (int)((id / N) % N), ix = (int)(id / ((long long)N * N));
        // the coarse cells whose CLOSED box holds the point: floor(i m / (N - 1)), and the one below when the point sits on a face
        int c0[3], c1[3];
        const int idx[3] = {ix, iy, iz};
        for (int a = 0; a < 3; ++a) {
            const long long t = (long long)idx[a] * m;
            int c = (int)(t / (N - 1));
            const bool on_face = t % (N - 1) == 0;
            c1[a] = min(c, m - 1);
            c0[a] = on_face ? max(c - 1, 0) : c1[a];
        }
        for (int x = c0[0]; x <= c1[0]; ++x)
            for (int y = c0[1]; y <= c1[1]; ++y)
                for (int z = c0[2]; z <= c1[2]; ++z) band = band || flag2[((size_t)x * m + y) * m + z] != 0;
        if (!band) sdf[id] = csdf[((size_t)c1[0] * nc + c1[1]) * nc + c1[2]];
    }
    const unsigned long long mk = __ballot(band);
    if (lane == 0) wave_cnt[wave] = __popcll(mk);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < 16; ++k) {
            const int c = wave_cnt[k];
            wave_cnt[k] = tot;
            tot += c;
        }
        block_base = tot ? atomicAdd(count, tot) : 0;
    }
    __syncthreads();
    if (band) list[block_base + wave_cnt[wave] + __popcll(mk & ((1ull << lane) - 1ull))] = (int)id;
}
