// finish.hpp -- resident finishers of loops A and B (RT:174-241, RFU:365-484).
//
// Included by arah_hip.hip inside its anonymous namespace.  Loops A and B run as one launch per iteration over
// compacted lists (k_nearest_* + k_sdf_march, k_joint_iter): fine while the lists are wide, but after a few
// iterations a few thousand rays are left and every further iteration is the latency of two or three launches on a
// nearly empty machine -- 37 + 47 iterations of 52 / 32 us in a 512 x 512 frame.  A finisher takes over there: a
// workgroup adopts a 16-ray tile of the list and runs its rays to the end of the loop without leaving the kernel (the
// same device functions in the same order as the per-iteration kernels: results are bit-identical per ray).
// Measured on the MI355X (profiles/r03_*): the loop-B finisher replaces 48 launches (1.5 ms) by one of 1.36 ms and is
// the default after three wide iterations.  The loop-A finisher takes over after ARAH_TRACE_BULK_STEPS = 24 wide steps of a
// full frame (earlier hand-overs lose: too many rays left for sixteen-ray tiles, profiles/r03_ab_trace_finish.txt) and
// runs the WHOLE loop for ray lists of at most ARAH_TRACE_SMALL = 4096 rays (a training view); its nearest-vertex search
// is the sixteen-lane one (four slots per wave in one pass) -- with one wave per query its step was 95 us against 52 for
// the per-step kernels and it was not the default.
#pragma once

constexpr int kFinTile = 16;
constexpr size_t kLdsTraceFinish = (kFinTile * 4 * 2 + kFinTile * 16 + kFinTile * 8) * 4 + (size_t)kFinTile * kSdfLd * 4;
constexpr size_t kLdsJointFinish = (kFinTile * 4 * 3 + 24 * 16 + kFinTile * 2 + kFinTile * kLogitLd + 32) * 4 + (size_t)kFinTile * kSdfLd * 4;

// ---- loop A: the remaining sphere-tracing steps of the rays in list[0 .. *count)
template <bool SPLIT>
__global__ __launch_bounds__(kThreads, 4) void k_trace_finish(FrameDev fr, KnnData kd, RaySet rs, TraceState st, float* Tcur,
                                                            int* nn_idx, const int* list, const int* count, int steps_left,
                                                            unsigned long long* ctr_knn, unsigned long long* ctr_fwd) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xin = smem;                                   // [16][4] normalised canonical point of the step
    float* outv = xin + kFinTile * 4;                    // [16][4]
    float* Tl = outv + kFinTile * 4;                     // [16][16] inverse-LBS transform of the step
    int* ids = reinterpret_cast<int*>(Tl + kFinTile * 16);   // [16] ray id, -1 = slot done
    float* tl = reinterpret_cast<float*>(ids + kFinTile);   // [16] depth
    float* farl = tl + kFinTile;                         // [16]
    int* nnl = reinterpret_cast<int*>(farl + kFinTile);  // [16] nearest vertex of the previous step
    int* divl = nnl + kFinTile;                          // [16]
    int* s_live = divl + kFinTile;                       // [1] (+ pad to 16)
    float* actA = reinterpret_cast<float*>(s_live + 3 * kFinTile);   // [16][kSdfLd]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = *count;
    const GridInfo g = *kd.grid;
    const float scale = sdf_scale(bc);
    int n_done = 0;   // evaluations of this workgroup (one atomic at the end: a counter bumped every step by every tile is the
                      // most contended word of the launch, and the waves' next loads queue behind it)
    for (int tile = blockIdx.x; tile * kFinTile < n; tile += gridDim.x) {
        __syncthreads();
        if (tid < kFinTile) {
            const int i = tile * kFinTile + tid;
            const int id = i < n ? list[i] : -1;
            ids[tid] = id;
            tl[tid] = id >= 0 ? st.t[id] : 0.f;
            farl[tid] = id >= 0 ? st.far[id] : 0.f;
            nnl[tid] = id >= 0 ? nn_idx[id] : -1;
            divl[tid] = 0;
        }
        for (int step = 0; step < steps_left; ++step) {
            __syncthreads();
            if (wave == 0) {
                const unsigned long long live = __ballot(lane < kFinTile && ids[lane] >= 0);
                n_done += __popcll(live);
                if (lane == 0) *s_live = live != 0ull;
            }
            __syncthreads();
            if (!*s_live) break;
            // nearest vertex + inverse LBS: sixteen lanes per slot (nearest_vertex_group16), waves 0..3 take four slots each
            // in one pass; the first lane of a group finishes its slot
            if (wave < 4) {
                const int s = wave * 4 + (lane >> 4);
                const int id = ids[s];
                if (id >= 0) {   // uniform over the group
                    const V3 p = ray_point(rs, id, tl[s]);
                    float best = 3.4e38f;
                    int bi = 0x7fffffff;
                    const int seed = nnl[s];
                    if (seed >= 0) {   // the previous step's nearest vertex bounds the search
                        const float dx = fr.verts_raw[seed * 3] - p.x, dy = fr.verts_raw[seed * 3 + 1] - p.y,
                                    dz = fr.verts_raw[seed * 3 + 2] - p.z;
                        best = dx * dx + dy * dy + dz * dz;
                        bi = seed;
                    }
                    bi = nearest_vertex_group16<kClusterSize>(kd, kd.sorted4, kd.spheres, g, p, best, bi, lane);
                    if ((lane & 15) == 0) {
                        float T[16];
                        vertex_transform(fr, bi, T);
                        const V3 y = V3{p.x - bc.trans[0], p.y - bc.trans[1], p.z - bc.trans[2]};
                        const V3 xh = normalize_pt(bc, inverse_affine_apply(T, y));
                        store_T(Tl + s * 16, T);
                        nnl[s] = bi;
                        reinterpret_cast<f32x4*>(xin)[s] = f32x4{xh.x, xh.y, xh.z, 0.f};
                    }
                } else if ((lane & 15) == 0) {
                    reinterpret_cast<f32x4*>(xin)[s] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            __syncthreads();
            f32x4 dlast[kSdfMT][1];
            sdf_trunk<false, 1, SPLIT>(fr.sdf, xin, actA, kSdfLd, nullptr, dlast, wave, lane);
            sdf_head<SPLIT>(fr.sdf, actA, kSdfLd, outv, 4, tid, kFinTile);
            __syncthreads();
            if (tid < kFinTile) {
                const int id = ids[tid];
                if (id >= 0) {
                    const float sdf = outv[tid * 4] * scale;                          // RT:528
                    const float march = fminf(fmaxf(sdf, -kClampDist), kClampDist);   // RT:228
                    bool div = false;
                    if (fabsf(march) > kRootThresh) {                                 // RT:231-235
                        const float t = tl[tid] + march;
                        tl[tid] = t;
                        div = t >= farl[tid];
                        divl[tid] = div ? 1 : 0;
                    }
                    const bool keep = !((fabsf(sdf) <= kRootThresh) || div);          // RT:238-241
                    if (!keep || step + 1 == steps_left) {   // the ray leaves the loop: what the per-step kernels leave behind
                        st.t[id] = tl[tid];
                        st.diverged[id] = divl[tid] ? 1 : 0;
                        const f32x4 xn = reinterpret_cast<const f32x4*>(xin)[tid];
                        st.xcur[(size_t)id * 3] = xn[0];
                        st.xcur[(size_t)id * 3 + 1] = xn[1];
                        st.xcur[(size_t)id * 3 + 2] = xn[2];
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            reinterpret_cast<f32x4*>(Tcur + (size_t)id * 16)[c] = reinterpret_cast<const f32x4*>(Tl + tid * 16)[c];
                        nn_idx[id] = nnl[tid];
                        ids[tid] = -1;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        count_add(ctr_knn, n_done);
        count_add(ctr_fwd, n_done);
    }
}

// ---- loop B: the remaining Broyden iterations (none of them the first) of the rays in list[0 .. *count)
template <bool SPLIT>
__global__ __launch_bounds__(kThreads, 4) void k_joint_finish(FrameDev fr, Broyden4State st, RaySet rs, const int* list,
                                                            const int* count, int iters_left, unsigned long long* ctr_skin,
                                                            unsigned long long* ctr_sdf) {
    const BodyConst bc = load_bc(fr);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int TW = kFinTile;
    float* xin = smem;                        // [16][4] normalised
    float* xraw = xin + TW * 4;               // [16][4] raw x_hat + depth
    float* outv = xraw + TW * 4;              // [16][4]
    float* sbones = outv + TW * 4;            // [24][16]
    int* ids = reinterpret_cast<int*>(sbones + 24 * 16);   // [16]
    int* s_live = ids + TW;                                 // [1] (+ pad)
    float* logits = reinterpret_cast<float*>(s_live + TW);  // [16][33]
    float* act = logits + TW * kLogitLd;   // 16 * 33 = 528 floats: a multiple of 4, the tile stays 16-byte aligned
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int n = *count;
    const float scale = sdf_scale(bc);
    int n_done = 0;
    for (int i = tid; i < 24 * 16; i += kThreads) sbones[i] = fr.bones[i];
    for (int tile = blockIdx.x; tile * TW < n; tile += gridDim.x) {
        __syncthreads();
        if (tid < TW) {
            const int i = tile * TW + tid;
            ids[tid] = i < n ? list[i] : -1;
        }
        for (int it = 0; it < iters_left; ++it) {
            __syncthreads();
            if (wave == 0) {
                const unsigned long long live = __ballot(lane < TW && ids[lane] >= 0);
                n_done += __popcll(live);
                if (lane == 0) *s_live = live != 0ull;
            }
            if (tid < TW) {
                const int id = ids[tid];
                f32x4 u = {0.f, 0.f, 0.f, 0.f};
                if (id >= 0) u = reinterpret_cast<const f32x4*>(st.ueval)[id];
                const V3 q = normalize_pt(bc, V3{u[0], u[1], u[2]});
                reinterpret_cast<f32x4*>(xin)[tid] = f32x4{q.x, q.y, q.z, 0.f};
                reinterpret_cast<f32x4*>(xraw)[tid] = u;
            }
            __syncthreads();
            if (!*s_live) break;
            skin_mlp<1>(fr.skin, xin, act, logits, wave, lane);
            f32x4 dlast[kSdfMT][1];
            sdf_trunk<false, 1, SPLIT>(fr.sdf, xin, act, kSdfLd, nullptr, dlast, wave, lane);
            sdf_head<SPLIT>(fr.sdf, act, kSdfLd, outv, 4, tid, TW);
            __syncthreads();
            if (tid < TW) {
                const int id = ids[tid];
                if (id >= 0) {   // the body of k_joint_iter<false>
                    float T[16];
                    V3 xb;
                    const f32x4 u = reinterpret_cast<const f32x4*>(xraw)[tid];
                    skin_tail(logits + tid * kLogitLd, sbones, V3{u[0], u[1], u[2]}, T, xb);
                    const V3 p = ray_point(rs, id, u[3]);                    // RFU:435-436
                    float gnew[4] = {outv[tid * 4] * scale, xb.x - (p.x - bc.trans[0]), xb.y - (p.y - bc.trans[1]),
                                     xb.z - (p.z - bc.trans[2])};
                    float J[16], stp[4], gx[4], dg[4], dx[4];
#pragma unroll
                    for (int e = 0; e < 16; ++e) J[e] = st.Jinv[(size_t)id * 16 + e];
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
                    const bool keep = (eb > kRootThresh) && (err < kDvg);
                    if (keep) {
                        broyden_update<4>(J, dx, dg, gx, stp);
#pragma unroll
                        for (int r = 0; r < 4; ++r) st.gx[(size_t)id * 4 + r] = gx[r];
#pragma unroll
                        for (int e = 0; e < 16; ++e) st.Jinv[(size_t)id * 16 + e] = J[e];
                        reinterpret_cast<f32x4*>(st.step)[id] = f32x4{stp[0], stp[1], stp[2], stp[3]};
                        reinterpret_cast<f32x4*>(st.ueval)[id] = f32x4{u[0] + stp[0], u[1] + stp[1], u[2] + stp[2], u[3] + stp[3]};
                    } else {
                        ids[tid] = -1;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        count_add(ctr_skin, n_done);
        count_add(ctr_sdf, n_done);
    }
}
