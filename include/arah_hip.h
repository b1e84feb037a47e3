/*
 * arah_hip.h -- C ABI of the MI355X (gfx950) implementation of ARAH's articulated-SDF
 * volume-rendering hot path.
 *
 * The reference (taconite/arah-release) is pure Python: there is no FFI in it.  The interfaces
 * these entry points replace are the Python seams of the hot path (paths relative to the
 * reference root):
 *
 *   arah_trace              BodyRayTracing.sphere_tracing + search_iso_surface_depth
 *                           im2mesh/metaavatar_render/renderer/ray_tracing.py:174-296,
 *                           im2mesh/utils/root_finding_utils.py:365-484
 *   arah_sample_canonicalize BodyRayTracing.ray_sampler / inv_transform_points_opt /
 *                           search_canonical_corr  ray_tracing.py:313-461, root_finding_utils.py:267-362
 *   arah_shade_composite    IDHRNetwork.get_rbg_value_vol_sdf + the eval tail of forward
 *   arah_shade_points       its per-sample half (SDF, normal, colour, density) on the shipped engine
 *                           renderer/implicit_differentiable_renderer.py:261-396, :142-148,:225-257
 *   arah_render             IDHRNetwork.forward (eval)  implicit_differentiable_renderer.py:42-259
 *   arah_sdf_eval           sdf_network(x) / gradient(sdf, x)   hyperlayers.py:385-415,
 *                           siren_modules.py:35-37, diff_operators.py:39-50
 *   arah_skin_lbs           forward_skinning / query_weights  root_finding_utils.py:54-167,
 *                           utils/utils.py:138-181, metaavatar/models/decoder.py:201-233
 *   arah_skin_jacobian      forward_skinning_jac  root_finding_utils.py:170-226
 *   arah_color_eval         RenderingNetwork.forward  metaavatar_render/models/decoder.py:69-124
 *   arah_nearest_inverse_lbs inv_transform_points_smpl_verts  ray_tracing.py:382-400
 *                           (pytorch3d.ops.knn_points K=1 + nearest-vertex inverse LBS)
 *   arah_broyden3_lbs       search_canonical_corr on caller-supplied initial guesses
 *                           (broyden.py:4-78 with g = LBS(x) - target)
 *   arah_joint_root_find    search_iso_surface_depth on caller-supplied starts  root_finding_utils.py:365-484
 *   arah_sdf_grid           create_mesh_vertices_and_faces' lattice evaluation  utils/sdf_meshing.py:13-70
 *   arah_marching_cubes     skimage.measure.marching_cubes_lewiner as called at utils/sdf_meshing.py:95 (+ :96-101)
 *   arah_rasterize          pytorch3d MeshRasterizer (pix_to_face) as used at metaavatar_render/models/__init__.py:232-276
 *   arah_shade_train_*      get_rbg_value_vol_sdf with self.training: per-sample forward and backward
 *                           renderer/implicit_differentiable_renderer.py:291-361, diff_operators.py:39-50
 *   arah_gram_skinny        the matmul backward of autograd for the 1- / 3-row heads and the K = 3 first layer
 *                           (weight gradients summed over ~1e5 samples)
 *   arah_mesh_query         check_mesh_contains + igl.point_mesh_squared_distance + igl.barycentric_coordinates_tri
 *                           im2mesh/utils/libmesh/inside_mesh.py:4-160, im2mesh/data/zju_mocap.py:466-529
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with "h_"; the caller owns all
 *     buffers, including the workspace; nothing is allocated, freed or synchronised inside;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); calls are re-entrant
 *     across streams as long as workspaces differ;
 *   - return value: 0 on success, negative ARAH_E_* on error; nothing throws;
 *   - floats are IEEE fp32, masks are uint8 (0/1), indices int32; arrays are dense row-major.
 */
#ifndef ARAH_HIP_H
#define ARAH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARAH_OK 0
#define ARAH_E_BADARG (-1)      /* null pointer / negative size */
#define ARAH_E_SHAPE (-2)       /* network shape not supported by the compiled kernels */
#define ARAH_E_WORKSPACE (-3)   /* workspace or frame buffer too small */
#define ARAH_E_LAUNCH (-4)      /* HIP launch error (hipGetLastError) */
#define ARAH_E_SAMPLING (-5)    /* n_steps < n_near + n_far + 1, or n_steps > ARAH_MAX_STEPS */

#define ARAH_MAX_STEPS 128
#define ARAH_N_JOINTS 24
#define ARAH_COLOR_NO_VIEW_DIR 0 /* input [x, n, feat, pose]         (ZJUMOCAP-377-mono) */
#define ARAH_COLOR_IDR 1         /* input [x, PE4(view), n, feat, pose] (ZJUMOCAP-313, H36M) */

/* GEMM engine of the forward SDF trunks (every result stays fp32-class, see csrc/mlp.hpp):
 *   SPLIT_F16: fp32 operands carried as hi + lo f16 pairs, three v_mfma_f32_16x16x32_f16 per product, fp32
 *              accumulation (22 significant bits per operand; error vs fp64 at or below the exact engine's);
 *   FP32     : v_mfma_f32_16x16x4_f32 everywhere (bit-for-bit an fmaf chain), 16/3 x the matrix-pipe time. */
#define ARAH_PRECISION_SPLIT_F16 0
#define ARAH_PRECISION_FP32 1

/* Row-major, un-packed network weights as PyTorch holds them (weight-norm already folded:
 * W = g * v / |v|).  Shapes are the ones every ARAH config uses; arah_prepare_frame rejects
 * anything else with ARAH_E_SHAPE. */
typedef struct ArahNets {
    /* emitted FiLM-SIREN SDF MLP 3 -> 256 x6 -> 1  (hyperlayers.py:497-510) */
    const float* sdf_w[7];      /* [256,3], 5 x [256,256], [1,256] */
    const float* sdf_b[7];      /* [256] x6, [1] */
    const float* film_freq;     /* [6*256] */
    const float* film_phase;    /* [6*256] */
    /* skinning MLP 3 -> 128 x4 -> 25, Softplus(beta=100) */
    const float* skin_w[5];     /* [128,3], 3 x [128,128], [25,128] */
    const float* skin_b[5];
    /* colour MLP in -> 256 -> 256 -> 128 -> (in+128) -> 256 -> 256 -> 3, ReLU, sigmoid, skip at 3 */
    const float* col_w[6];      /* [256,in], [256,256], [128,256], [256,in+128], [256,256], [3,256] */
    const float* col_b[6];
    const float* pose_vec;      /* [n_pose] per-frame constant tail of the colour input (may be NULL if n_pose==0) */
    int32_t col_mode;           /* ARAH_COLOR_* */
    int32_t n_pose;             /* 128 for color_pose_encoder 'latent' */
    const float* beta;          /* [1] DEVICE: |variance|, un-clipped (NULL: 1e-3, the reference's initial value) */
    int32_t precision;          /* ARAH_PRECISION_* */
} ArahNets;

/* Per-frame body (lightning_model.py:581-632 keys smpl_verts, skinning_weights, bone_transforms,
 * trans, coord_min, coord_max, center), batch element 0. */
typedef struct ArahBody {
    const float* verts;         /* [n_verts,3] posed + trans */
    const float* vert_weights;  /* [n_verts,24] */
    const float* bones;         /* [24,4,4] */
    /* the four per-frame scalars are DEVICE pointers too: they are inputs of the forward (dataset tensors) and are
     * consumed by the kernels only, so that building a frame needs no device->host copy and no stream drain */
    const float* trans;         /* [3] */
    const float* center;        /* [3] */
    const float* coord_min;     /* [1] */
    const float* coord_max;     /* [1] */
    int32_t n_verts;            /* <= 6912 */
    const void* prepared;       /* NULL, or a buffer filled by arah_prepare_body for these verts: arah_prepare_frame then
                                   skips the nearest-vertex tables (its kernels read them from here; keep it alive with the
                                   frame, and order the stream that built it before the frame's stream) */
} ArahBody;

enum { ARAH_SHADE_ENGINE_DEFAULT = 0 /* bf16 x 3 on split frames */, ARAH_SHADE_ENGINE_FP32 = 1 };
enum { ARAH_CANON_KERNEL_WAVE = 0 /* point-owning waves, hi fragments in LDS */, ARAH_CANON_KERNEL_TILE = 1,
       ARAH_CANON_KERNEL_WAVE_L2 = 2 };

typedef struct ArahSampling {
    int32_t n_steps, n_near, n_far;   /* configs/default.yaml:49-51 */
    int32_t cano_view_dirs;           /* model.cano_view_dirs */
    int32_t render_last_pt;           /* model.render_last_pt */
    int32_t full_shading;             /* 0 (default): exact lazy shading -- normal + colour only for samples whose
                                         density is > 0, the rest provably get weight 0; 1: shade every valid sample
                                         like the reference does (same image, bit for bit) */
    /* device copies of torch.linspace(0, 1, k) for k = n_steps, n_near + 1, n_far (RT:317,330,340);
     * the caller builds them once per config so that depth samples are bit-identical to torch's */
    const float* lin_steps;
    const float* lin_near;
    const float* lin_far;             /* may be NULL when n_far == 0 */
    /* Per-call switches and profiling hooks (round 4: they were environment variables read into process-wide statics and
     * process-wide event setters; the library now keeps no state between calls, so calls on different streams with different
     * ArahSampling objects do not see each other). */
    int32_t shade_engine;             /* ARAH_SHADE_ENGINE_*: loop D's normal sweep + colour MLP on a split-engine frame */
    int32_t canon_kernel;             /* ARAH_CANON_KERNEL_*: loop C's solver on a split-engine frame */
    /* pairs of hipEvent_t (start, stop), both non-NULL to take effect: recorded on the call's stream immediately before /
     * after the launch of loop C's solver, of the density pre-pass and of the shading kernel */
    void* ev_canon[2];
    void* ev_density[2];
    void* ev_shade[2];
    /* Tiered evaluation (round 6, csrc/tier.hpp; arah_render with full_shading == 0 only): the occupancy buffer that
     * arah_prepare_occupancy filled for THIS frame, or NULL = every sample of every ray through loops C and D like the reference
     * (ray_tracing.py:313-380, implicit_differentiable_renderer.py:261-396).  With it, samples outside the posed fat body are
     * certified sigma = +0 without being evaluated; images and masks are the untiered path's bit for bit. */
    const void* occupancy;
    /* the tiered path launches loop C's solver and the density pass twice (phase 1 / phase 2): event pairs of the second launches */
    void* ev_canon2[2];
    void* ev_density2[2];
} ArahSampling;

/* Opaque-ish handle filled by arah_prepare_frame: device pointers into the caller's frame
 * buffer (MFMA-packed weights, padded vertices) plus scalars.  POD, copy freely. */
typedef struct ArahFrame {
    const float* sdf_w0;        /* [256][4] */
    const float* sdf_wp[5];     /* packed fwd */
    const float* sdf_wpT[5];    /* packed transposed (reverse sweep) */
    const float* sdf_w6;        /* [256] */
    const float* sdf_b6;        /* [1] */
    const float* sdf_bias;      /* [6][256] */
    const float* sdf_freq;      /* [6][256] */
    const float* sdf_phase;     /* [6][256] */
    const void* sdf_wps[5];     /* split-packed fwd: hi/lo f16 planes, pre-scaled by a power of two per layer */
    const float* sdf_fw;        /* [6][256] 30 f / pi */
    const float* sdf_pw;        /* [6][256] 30 (f b + phi) / pi */
    const float* sdf_fws;       /* [6][256] fw divided by the split scales of the producing layer */
    const float* skin_w0;       /* [128][4] */
    const float* skin_wp[3];    /* packed 128x128 */
    const float* skin_w4p;      /* packed [32][128] */
    const float* skin_bias;     /* [4][128] + [32] */
    const void* skin_wps[4];    /* split-packed 128x128 x3 and [32][128] */
    const float* skin_scales;   /* [8] activation scales S_k (probed per frame) and accumulator un-scales */
    const void* skin_wpr;       /* 4 layers, split-packed in the channel order of the point-owning-wave kernels (csrc/canon_wave.hpp) */
    const float* skin_wconsts;  /* their constants: first layer, biases in z = 100 log2(e) x units, un-scales */
    const float* col_w0p;       /* packed [256][KIN_PAD] (columns permuted to [feat,x,n,view]) */
    const float* col_w1p;       /* packed [256][256] */
    const float* col_w2p;       /* packed [128][256] */
    const float* col_w3ap;      /* packed [256][KIN_PAD] */
    const float* col_w3bp;      /* packed [256][128] */
    const float* col_w4p;       /* packed [256][256] */
    const float* col_w5;        /* [3][256] */
    const float* col_bias;      /* b0'[256] b1[256] b2[128] b3'[256] b4[256] b5[4] */
    const float* col_w0pT;      /* transposed packings of the colour MLP (reverse sweep of the training backward) */
    const float* col_w1pT;
    const float* col_w2pT;
    const float* col_w3apT;
    const float* col_w3bpT;
    const float* col_w4pT;
    const void* b3[22];         /* bf16 hi/lo fragments of the SDF (W, W^T) and colour (W, W^T) matrices: operands of the
                                   training kernels' bf16 x 3 products (arah_shade_train_forward / _backward) */
    const float* verts4;        /* [256][28][4] k-d clustered vertices (x, y, z, original index) */
    const float* knn_spheres;   /* [256][4] bounding spheres of the clusters */
    const void* knn_grid;       /* grid geometry (device) */
    const void* knn_cells;      /* [n_cells][64] candidate clusters per cell */
    const float* verts;         /* caller's [n_verts][3] */
    const float* vert_T;        /* [n_verts][16] blended bone transform of each vertex, sum_j w_vj A_j (frame buffer) */
    const float* bones;         /* caller's [24][16] */
    const float* scalars;       /* [9] device: trans(3), center(3), coord_min, coord_max, |variance| */
    int32_t n_verts;
    int32_t col_mode;
    int32_t precision;          /* ARAH_PRECISION_* the frame was prepared for */
} ArahFrame;

/* Work counters (points evaluated), SURVEY 8(d). */
typedef struct ArahCounters {
    uint64_t n_sdf_fwd, n_sdf_grad, n_skin_fwd, n_skin_jac, n_col, n_knn;
    uint64_t n_density;   /* samples seen by the density pre-pass of lazy shading (a subset of n_sdf_fwd) */
    uint64_t n_canon;     /* skinning-MLP evaluations of loop C (k_canon_wave / k_canon_solve; a subset of n_skin_fwd) */
    uint64_t n_split_nonfinite; /* loop-C evaluations of the split engine whose residual was not finite (an activation
                                   left the f16 range): non-zero means the frame should be re-prepared with
                                   ARAH_PRECISION_FP32 */
    /* tiered eval forward: rays classified; of them surface rays (loops A+B converged), promoted to the exact tier, skipped
     * (certified rgb = 0); samples evaluated in phase 1 (surface rays, marked samples, witnesses), in phase 2 (the rest of the
     * promoted rays), never evaluated; rays that sent a witness */
    uint64_t n_tier_rays, n_tier_rays_surface, n_tier_rays_promoted, n_tier_rays_skipped;
    uint64_t n_tier_samples_p1, n_tier_samples_p2, n_tier_samples_skipped, n_tier_witnesses;
    uint64_t n_tier_rays_untraced;       /* rays whose [near, far] segment misses the posed fat body: loops A+B not run */
    uint64_t n_canon_p2, n_density_p2;   /* the share of n_canon / n_density that phase 2 ran */
} ArahCounters;

/* ---- frame preparation ------------------------------------------------------------------ */
/* Optional early half: the nearest-vertex tables of a posed body (what ray_tracing.py:382-400 asks pytorch3d's
 * knn_points for, per call) depend on the vertices only.  arah_prepare_body builds them into a caller buffer of
 * arah_body_bytes() bytes (256-byte aligned) on `stream` -- typically a side stream, while the caller's stream runs the
 * pose encoder and the hypernetwork; pass the buffer as ArahBody.prepared to arah_prepare_frame. */
size_t arah_body_bytes(void);
int arah_prepare_body(const float* verts, int32_t n_verts, void* body_buf, size_t body_bytes, void* stream);
size_t arah_frame_bytes(const ArahNets* h_nets, const ArahBody* h_body);
int arah_prepare_frame(const ArahNets* h_nets, const ArahBody* h_body, void* frame_buf, size_t frame_bytes,
                       ArahFrame* h_frame_out, void* stream);

/* ---- workspace -------------------------------------------------------------------------- */
size_t arah_workspace_bytes(int32_t n_rays, int32_t n_steps);
/* zero / read the device-side work counters that live at the head of a workspace */
int arah_counters_reset(void* workspace, void* stream);
int arah_counters_read(const void* workspace, ArahCounters* h_out, void* stream); /* syncs the stream */

/* ---- unit seams (parity tests; also usable on their own) --------------------------------- */
/* x_norm [P,3] -> sdf [P] (normalised units), optional feat [P,256], optional grad [P,3] */
int arah_sdf_eval(const ArahFrame* h_frame, const float* x_norm, int32_t n_pts, float* sdf, float* feat,
                  float* grad, void* workspace, size_t workspace_bytes, void* stream);
/* SDF on the N^3 lattice of [-1,1]^3 (utils/sdf_meshing.py:13-70): sdf[(ix*N + iy)*N + iz], normalised units */
int arah_sdf_grid(const ArahFrame* h_frame, int32_t n_side, float* sdf, void* workspace, size_t workspace_bytes,
                  void* stream);
/* The same lattice for marching cubes at level 0 (utils/sdf_meshing.py:95 only reads the corners of cells that change sign):
 * exact values wherever the level set can pass -- coarse cells (1/32 of the box) whose corner values and own slope admit a zero,
 * plus their 26 neighbours -- and a value of the right sign elsewhere.  list: [N^3] int32 scratch; scratch:
 * arah_sdf_grid_band_scratch_bytes() bytes.  Same triangle soup as arah_sdf_grid + arah_marching_cubes; N >= 33. */
size_t arah_sdf_grid_band_scratch_bytes(void);
int arah_sdf_grid_band(const ArahFrame* h_frame, int32_t n_side, float* sdf, int32_t* list, void* scratch,
                       size_t scratch_bytes, void* workspace, size_t workspace_bytes, void* stream);
/* nearest covering face per pixel (pix_to_face of the rasteriser models/__init__.py:232-237 uses): tri [F,3,3] =
 * (u, v, z) per corner in pixel coordinates / view depth; zbuf [H*W] keys (depth bits << 32 | face), pre-set to ~0 */
int arah_rasterize(const float* tri_uvz, int32_t n_faces, int32_t height, int32_t width, float z_near,
                   uint64_t* zbuf, void* stream);
/* raw canonical x_hat [P,3] -> optional w [P,24], x_bar [P,3], T [P,16] */
int arah_skin_lbs(const ArahFrame* h_frame, const float* x_hat, int32_t n_pts, float* w, float* x_bar,
                  float* T, void* workspace, size_t workspace_bytes, void* stream);
/* arah_skin_lbs for a point list whose length is known on the DEVICE only: the first min(n_max, *n_items * per_item) rows
 * of x_hat are skinned into x_bar, the other rows of x_bar are left as the caller set them.  (The vertices of the mesh
 * arah_marching_cubes just extracted: per_item = 3 corners per triangle; no host round trip for the count.) */
int arah_skin_lbs_counted(const ArahFrame* h_frame, const float* x_hat, int32_t n_max, const int32_t* n_items,
                          int32_t per_item, float* x_bar, void* workspace, size_t workspace_bytes, void* stream);
/* Level set of a lattice volume as a triangle soup (the call utils/sdf_meshing.py:95 makes to
 * skimage.measure.marching_cubes_lewiner, followed by :96-101's vertex = origin + index * voxel_size on the lattice of
 * [-1,1]^3).  sdf [n][n][n] indexed [ix][iy][iz]; tri_table [256][16] int8 / n_tri [256] int32 (DEVICE): for every
 * inside-outside pattern of a cell's 8 corners (bit c set = corner c below `level`; corner / edge numbering of
 * arah_release_amd/meshing.py) the edge ids of its triangles, -1 padded, and their number (<= 5).  -> tris [cap][3][3]
 * coordinates in [-1,1]^3, right-hand normals towards decreasing values, cells in (ix, iy, iz) order; rows beyond the count
 * are ZERO (degenerate triangles); *n_tris (device) = the number of triangles of the level set, which may exceed cap (then
 * only the first cap were written).  Shared vertices of neighbouring cells are bit-equal.  scratch:
 * arah_marching_cubes_scratch_bytes(n_side) device bytes.  No host synchronisation. */
size_t arah_marching_cubes_scratch_bytes(int32_t n_side);
int arah_marching_cubes(const float* sdf, int32_t n_side, float level, const int8_t* tri_table, const int32_t* n_tri,
                        float* tris, int32_t cap, int32_t* n_tris, void* scratch, size_t scratch_bytes, void* stream);
/* raw canonical x_hat [P,3] -> d x_bar / d x_hat [P,3,3] */
int arah_skin_jacobian(const ArahFrame* h_frame, const float* x_hat, int32_t n_pts, float* jac,
                       void* workspace, size_t workspace_bytes, void* stream);
/* x_norm [P,3], normal [P,3], view [P,3] (ignored for NO_VIEW_DIR), feat [P,256] -> rgb [P,3] */
int arah_color_eval(const ArahFrame* h_frame, const float* x_norm, const float* normal, const float* view,
                    const float* feat, int32_t n_pts, float* rgb, void* workspace, size_t workspace_bytes,
                    void* stream);
/* posed points [P,3] -> nearest vertex idx [P] (optional), raw canonical x_hat0 [P,3], T0 [P,16] */
int arah_nearest_inverse_lbs(const ArahFrame* h_frame, const float* pts, int32_t n_pts, int32_t* idx,
                             float* x_hat0, float* T0, void* workspace, size_t workspace_bytes, void* stream);
/* Broyden on g(x) = LBS(x) - tgt from caller-supplied x0 [P,3], T0 [P,16];
 * J^-1_0 = (sum_j w_j(x0) A_j)[:3,:3]^-1.  -> x [P,3] raw canonical, T [P,16], err [P], conv [P] */
int arah_broyden3_lbs(const ArahFrame* h_frame, const float* tgt, const float* x0, const float* T0,
                      int32_t n_pts, float* x, float* T, float* err, uint8_t* conv, int32_t canon_kernel /* ARAH_CANON_KERNEL_* */,
                      void* workspace, size_t workspace_bytes, void* stream);

/* joint root find on u = (x_hat, depth) from caller-supplied starts (search_iso_surface_depth, root_finding_utils.py:
 * 365-484): valid [N], x0 [N,3] raw canonical, z0 [N], T0 [N,16]  ->  x [N,3], z [N], T [N,16], conv [N].
 * Rays outside `valid` keep (x0, z0, T0) and are reported as not converged. */
int arah_joint_root_find(const ArahFrame* h_frame, const float* cam_loc, int32_t rays_per_cam, const float* dirs,
                         const uint8_t* valid, const float* x0, const float* z0, const float* T0, int32_t n_rays,
                         float* x, float* z, float* T, uint8_t* conv, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- loop D with gradients (training; get_rbg_value_vol_sdf with self.training, IDR:261-396) ---------
 * Per-sample forward (SDF value, normal, colour) and its hand-written backward incl. the second-order path through
 * the normal.  Compositing and the loss stay with the caller (autograd); this pair is the custom op in between. */
typedef struct ArahTrainIn {
    int32_t n;                  /* P valid samples, dense */
    int32_t rotate_normal;      /* !cano_view_dirs: the colour net sees R n with R = T[:3,:3] (IDR:339-340) */
    int32_t ray_augm;           /* IDR:342-350: where n . view <= 0 the un-augmented view is used */
    int32_t geom_only;          /* 1: SDF value and normal only (the regulariser queries of IDR:104-128): the forward returns the
                                   normal in rgb4[:, :3]; the backward takes dL/dn in g_rgb and writes the feature stream h_6
                                   to ArahTrainGrads.c[0]; T / view are ignored, no hand-over */
    const float* x;             /* [P,3] normalised canonical points */
    const float* T;             /* [P,16] forward transforms (rotate_normal) or NULL */
    const float* view;          /* [P,3] view input of the colour net */
    const float* view_orig;     /* [P,3] (ray_augm) or NULL */
    const float* g_s;           /* [P]   dL/d sdf (normalised units)   -- backward only */
    const float* g_rgb;         /* [P,3] dL/d rgb                      -- backward only */
    /* Optional hand-over from the forward to the backward call (all NULL: the backward recomputes the whole forward).
     * Forward: tap_cin / tap_c non-NULL -> the colour MLP's input and hidden activations are written there.  Backward:
     * the same buffers (they are also ArahTrainGrads.cin / .c) plus the forward's rgb -> the normal sweep and the colour
     * MLP are not recomputed (9 of the backward kernel's 34 layer products, all on the fp32 MFMA). */
    float* tap_cin;             /* [P,kInPad] */
    float* tap_c[5];            /* [P,256] [P,256] [P,128] [P,256] [P,256] */
    const float* fwd_rgb4;      /* [P,4] rgb of the forward call       -- backward only */
} ArahTrainIn;

/* Outputs of the backward.  Weight gradients are sums of outer products over all samples; the kernel streams their
 * operands as dense row-major [P,width] matrices and the caller finishes them with library GEMMs:
 *   dW_k = av[k]^T h[k] + avd[k]^T hd[k]  (k = 0..5, SDF layer k+1),  db_k = colsum(av[k]),
 *   dw_7 = g_s^T cin[:, :256] + colsum(hd[6]),  db_7 = sum(g_s),
 *   colour layer l: dW_l = d[l]^T X_l with X_0 = cin, X_1..X_2 = c[0..1], X_3 = [cin | c[2]], X_4..X_5 = c[3..4]. */
typedef struct ArahTrainGrads {
    float* sdf;                 /* [P]   forward values again (recomputed) */
    float* rgb4;                /* [P,4] */
    float* gx4;                 /* [P,4] dL/dx */
    float* film_freq;           /* [6,256] dL/d freq  */
    float* film_phase;          /* [6,256] dL/d phase */
    float* h[6];                /* h_0 [P,4] (x), h_1..h_5 [P,256] */
    float* hd[7];               /* tangent stream: hd_0 = dL/dn [P,4], hd_1..hd_6 [P,256] */
    float* av[6];               /* adj v_1..v_6 [P,256] */
    float* avd[6];              /* adj vd_1..vd_6 [P,256] */
    float* cin;                 /* [P,KIN_PAD] colour input, columns [feat(256) | x | n | PE(view) | 0] */
    float* c[5];                /* colour hidden activations: 256, 256, 128, 256, 256 wide */
    float* d[6];                /* colour deltas: 256, 256, 128, 256, 256 wide, delta_5 [P,4] */
} ArahTrainGrads;

size_t arah_shade_train_slab_bytes(void);   /* scratch of the backward (pre-activation spill, per workgroup) */
/* -> sdf [P] (normalised units), rgb4 [P,4] */
int arah_shade_train_forward(const ArahFrame* h_frame, const ArahTrainIn* h_in, float* sdf, float* rgb4,
                             void* workspace, size_t workspace_bytes, void* stream);
int arah_shade_train_backward(const ArahFrame* h_frame, const ArahTrainIn* h_in, const ArahTrainGrads* h_out,
                              void* slab, size_t slab_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* VolSDF density + alpha compositing of the TRAINING forward and their backward (IDR:363-394 with self.training): one
 * thread per ray over its `len[r]` valid samples, which are contiguous from `off[r]` in the compacted per-sample arrays
 * sdf [P] (metres), rgb [P,3], z [P].  inv_beta: device scalar 1 / beta.  acc is clip(sum of weights, 0, 1).  The
 * backward takes dL/d rgb_map [n_rays,3] and dL/d acc [n_rays] and returns dL/d sdf [P], dL/d rgb [P,3] and the scalar
 * dL/d inv_beta.  Replaces ~40 element-wise launches forward and ~100 backward of the autograd formulation. */
int arah_composite_train_forward(int32_t n_rays, int32_t n_steps, int32_t render_last_pt, const int32_t* len,
                                 const int64_t* off, const float* sdf, const float* rgb, const float* z, const float* inv_beta,
                                 float* out_rgb, float* out_acc, void* stream);
int arah_composite_train_backward(int32_t n_rays, int32_t n_steps, int32_t render_last_pt, const int32_t* len,
                                  const int64_t* off, const float* sdf, const float* rgb, const float* z, const float* inv_beta,
                                  const float* g_rgb_map, const float* g_acc, float* g_sdf, float* g_rgb, float* g_inv_beta,
                                  void* stream);

/* Skinny weight-gradient product of the training step: partial[blk][i][j] = sum over the block's rows p of
 * a[p*lda + i] * b[p*ldb + j], i < m <= 4, j < n; blk < arah_gram_skinny_blocks(n_rows).  The caller sums over blk.
 * (The reference leaves these to autograd's matmul backward: IDR:336-361 through torch.autograd.) */
int32_t arah_gram_skinny_blocks(int32_t n_rows);

/* y[r] = W[r, :] . x + b0[r] + b1[r]: the batch-1 output layers of the SDF hypernetwork (hyperlayers.py:418-465, 256 ->
 * in*out + out per emitted layer; im2mesh/metaavatar/models/siren_modules.py:244-300 calls them once per frame), an HBM
 * stream of the weight matrix.  W [n_rows][n_cols] row-major, 16-byte aligned, n_cols a multiple of 4; b0 / b1 [n_rows]
 * or NULL (bias of the layer, hypo_params_init).  Inference only (no gradient). */
int arah_gemv_rows(const float* W, int32_t n_rows, int32_t n_cols, const float* x, const float* b0, const float* b1,
                   float* y, void* stream);
int arah_gram_skinny(const float* a, int32_t lda, int32_t m, const float* b, int32_t ldb, int32_t n, int32_t n_rows,
                     float* partial, void* stream);

/* y[c] = sum over rows r of scale[r] * a[r*lda + c] (scale NULL: 1), c < n_cols: the bias gradients of the training step's tall
 * layers (sums of per-sample deltas over ~1.2e5 samples) and, with scale = the upstream gradient g and a = a hypernetwork head's
 * weight matrix [in*out + out][256], the head's input gradient g W -- the reference leaves both to autograd (sum / mm backward
 * of hyperlayers.py:418-465 and of the skinning decoder).  partial: arah_colsum_blocks(n_rows) * n_cols floats of scratch; the
 * partial sums are added in block order (deterministic).  One pass over `a` at HBM speed, no host synchronisation. */
int32_t arah_colsum_blocks(int64_t n_rows);
int arah_colsum(const float* a, int64_t lda, int32_t n_cols, int64_t n_rows, const float* scale, float* partial, float* y,
                void* stream);

/* Hierarchical softmax of the skinning queries of a training step with its backward (utils/utils.py:138-181, called from
 * root_finding_utils.py:54-113 with the 25 logits x 20): weights [n][24] = hsoftmax(scale * logits [n][25]);
 * g_logits [n][25] = d L / d logits for upstream gradients g_weights [n][24] (the forward is recomputed).  The reference runs the
 * recursion on autograd. */
int arah_hsoftmax_train_forward(const float* logits, int32_t n, float scale, float* weights, void* stream);
int arah_hsoftmax_train_backward(const float* logits, int32_t n, float scale, const float* g_weights, float* g_logits, void* stream);

/* HierarchicalPoseEncoder of one frame (im2mesh/metaavatar/models/siren_modules.py:196-244) with its backward, one launch each
 * way: n_joints MLPs 19 -> 19 -> ReLU -> 6 on [own (13: rotation 9, joint 3, bone length 1) | the parent's feature (6); the
 * root: glob (6) = layer_0's output], walked down the kinematic tree.  own [J][13], W1 [J][19][19], b1 [J][19], W2 [J][6][19],
 * b2 [J][6] (the per-joint nn.Linear parameters stacked, (out, in) row-major), parents_host [J] on the HOST (-1: root; parents
 * precede children), n_joints <= 64 -> feats [J][6], hidden [J][19] (post-ReLU, kept for the backward).
 * Backward: g_feats [J][6] -> gradients of the stacked parameters and of glob.  The reference runs the joints on autograd. */
int arah_pose_tree_forward(const float* own, const float* glob, const float* W1, const float* b1, const float* W2,
                           const float* b2, const int32_t* parents_host, int32_t n_joints, float* feats, float* hidden,
                           void* stream);
int arah_pose_tree_backward(const float* own, const float* glob, const float* W1, const float* W2, const int32_t* parents_host,
                            int32_t n_joints, const float* feats, const float* hidden, const float* g_feats, float* gW1,
                            float* gb1, float* gW2, float* gb2, float* g_glob, void* stream);

/* inv[p] = (scale * m[p])^-1 for n row-major 3 x 3 matrices (cofactors): the Jacobians d x_bar / d x_hat of the implicit
 * re-attachment of the canonical points to the skinning network (implicit_differentiable_renderer.py:315-334, torch.inverse
 * there).  A singular matrix gives non-finite entries, as torch.inverse does on the device. */
int arah_inverse3x3(const float* m, int32_t n, float scale, float* inv, void* stream);

/* Mesh queries of the training data path (zju_mocap.py:461-543): for every query point the closest point of the triangle
 * mesh -- squared distance, face (lowest index on ties), the point, barycentric weights of the face's three vertices
 * (igl.point_mesh_squared_distance + igl.barycentric_coordinates_tri) -- and containment exactly as
 * im2mesh/utils/libmesh/inside_mesh.py:4-100 decides it (z-ray crossing parity in both directions, hash resolution 512,
 * double precision).  verts [V,3] f32, faces [F,3] i32, pts [P,3] f32 or f64 -> d2 [P] f64, face [P] i32,
 * closest [P,3] f64, bary [P,3] f64, inside [P] u8.  scratch: arah_mesh_query_scratch_bytes() device bytes. */
size_t arah_mesh_query_scratch_bytes(void);
int arah_mesh_query(const float* verts, int32_t n_verts, const int32_t* faces, int32_t n_faces, const void* pts,
                    int32_t pts_are_f64, int32_t n_pts, double* d2, int32_t* face, double* closest, double* bary,
                    uint8_t* inside, void* scratch, void* stream);

/* ---- the hot path ----------------------------------------------------------------------- */
/* rays: cam_loc [n_cams,3], ray r belongs to camera r / rays_per_cam; dirs [N,3]; near_far [N,2].
 * root_find_all: 0 = joint root find on the non-diverged rays (eval), 1 = on every ray (training, RT:249).
 * -> points_hat_norm [N,3], T [N,16], conv [N], start [N], end [N]   (RT:283-296) */
int arah_trace(const ArahFrame* h_frame, const float* cam_loc, int32_t rays_per_cam, const float* dirs,
               const float* near_far, int32_t n_rays, int32_t root_find_all, float* points_hat_norm, float* T,
               uint8_t* conv, float* start, float* end, void* workspace, size_t workspace_bytes, void* stream);
/* rand_steps [N,S], rand_near [N,near+1], rand_far [N,far]: uniform [0,1) draws for the stratified jitter of
 * training mode (perturb_z_vals, RT:298-311, in the order the reference draws them); all NULL = eval mode.
 * -> z [N,S], pts [N,S,3] normalised canonical, T [N,S,16], mask [N,S]   (RT:380, 549-555) */
int arah_sample_canonicalize(const ArahFrame* h_frame, const ArahSampling* h_cfg, const float* cam_loc,
                             int32_t rays_per_cam, const float* dirs, const float* near_far,
                             const uint8_t* conv, const float* start, const float* end, int32_t n_rays,
                             const float* rand_steps, const float* rand_near, const float* rand_far,
                             float* z, float* pts, float* T, uint8_t* mask, void* workspace,
                             size_t workspace_bytes, void* stream);
/* -> rgb [N,3], acc [N], vol_mask [N]   (IDR:148, 225-230, 261-396) */
int arah_shade_composite(const ArahFrame* h_frame, const ArahSampling* h_cfg, const float* dirs,
                         const float* z, const float* pts, const float* T, const uint8_t* mask,
                         int32_t n_rays, float* rgb, float* acc, uint8_t* vol_mask, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Per-sample half of loop D as a seam of its own (IDR:291-368 before the compositing): the SHIPPED shading kernel -- on a
 * split-engine frame the bf16 x 3 normal sweep and colour MLP (shade_engine = ARAH_SHADE_ENGINE_FP32: the fp32 MFMA), on an fp32 frame the
 * exact engine -- on n normalised canonical points with their own blended transforms T [n,16] and ray directions dirs [n,3].
 * -> rgbs [n,4] = {rgb, VolSDF density}, sdfn [n,4] = {sdf (normalised units), d sdf / d x_norm}.  Needs a workspace of
 * arah_workspace_bytes(n, 1). */
int arah_shade_points(const ArahFrame* h_frame, const float* x_norm, const float* T, const float* dirs, int32_t n_pts,
                      int32_t cano_view_dirs, int32_t shade_engine, float* rgbs, float* sdfn, void* workspace,
                      size_t workspace_bytes, void* stream);
/* whole eval forward.  pose34 = DEVICE [3][4] world->camera (R|t), read by the last kernel only (no host copy of
 * the pose, no stream drain).  Any of the optional outputs may be NULL, then they live in the workspace.
 * -> rgb [N,3], points_cam [N,3], vol_mask [N] */
int arah_render(const ArahFrame* h_frame, const ArahSampling* h_cfg, const float* cam_loc,
                int32_t rays_per_cam, const float* dirs, const float* near_far, const float* pose34,
                int32_t n_rays, float* rgb, float* points_cam, uint8_t* vol_mask, float* acc,
                float* dists, uint8_t* surface_conv, void* workspace, size_t workspace_bytes,
                void* stream);

/* ---- tiered evaluation (csrc/tier.hpp) --------------------------------------------------------- */
/* The reference evaluates every depth sample of every ray (ray_tracing.py:313-380 -> search_canonical_corr,
 * implicit_differentiable_renderer.py:336-368); a sample whose canonical point lies outside {sdf <= 18 beta} has density
 * exactly +0 there.  arah_prepare_occupancy voxelises the POSED image of that set for a prepared frame (SDF lattice of
 * [-1.5,1.5]^3, refined where the band can be, skinned forward with the skinning MLP, dilated by the lattice's reach) into a
 * caller buffer of arah_occupancy_bytes() bytes; ArahSampling.occupancy hands it to arah_render. */
size_t arah_occupancy_bytes(void);
int arah_prepare_occupancy(const ArahFrame* h_frame, void* occ_buf, size_t occ_bytes, void* workspace,
                           size_t workspace_bytes, void* stream);
/* the 16 words at the head of an occupancy buffer: origin[3], voxel, 1/voxel, dims[3], n_vox, valid, n_cells, n_fine,
 * n_selected, overflow, band (m), steepest measured stretch of the forward skinning between adjacent selected lattice points;
 * synchronises the stream */
int arah_occupancy_info(const void* occ_buf, int32_t* h_out16, void* stream);
/* after an arah_render on this workspace: ray_tier [N] (0 certified zero, 1 surface ray, 2 promoted; tiered path only) and
 * ray_sigma_pos [N] (1: some valid sample of the ray has density > 0); device pointers, either may be NULL */
int arah_tier_debug(void* workspace, size_t workspace_bytes, int32_t n_rays, int32_t n_steps, uint8_t* ray_tier,
                    uint8_t* ray_sigma_pos, void* stream);

/* tests: the per-sample arrays of the workspace's last arah_render, copied device to device (any pointer may be NULL):
 * z [N,S], pts [N,S,3] normalised canonical, T [N,S,16], mask [N,S], shaded [N,S,4] = {rgb, density}, state [N,S] */
int arah_debug_samples(void* workspace, size_t workspace_bytes, int32_t n_rays, int32_t n_steps, float* z, float* pts,
                       float* T, uint8_t* mask, float* shaded, uint8_t* state, void* stream);

/* name of the dominant kernel, for profilers */
const char* arah_dominant_kernel(void);
#ifdef __cplusplus
}
#endif
#endif /* ARAH_HIP_H */
