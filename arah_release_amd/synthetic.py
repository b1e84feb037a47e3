"""Seeded synthetic stand-in for the licence-gated inputs of the hot path.

The reference consumes (i) an SMPL body per frame and (ii) a flat input dict built by
``LightningModel.compose_inputs`` (reference lightning_model.py:463-634) from the ZJU-MoCap
dataset (zju_mocap_odp.py:270-330).  Neither SMPL nor the datasets are redistributable, so
benchmarks and parity tests use this generator, which emits *the same dict keys, shapes and
conventions* from a 24-joint capsule figure:

  * canonical ("Vitruvian") joints on the SMPL kinematic tree, 6890 surface points on capsules,
    soft-min-of-distance skinning weights (24 per vertex),
  * random axis-angle pose -> bone transforms A_j (canonical -> posed, no global translation),
  * pin-hole camera at the origin, body translated to z ~ 3 m; rays = pixels inside the projected
    AABB (+5 cm margin) whose ray/AABB interval is non-empty (utils.py:56-73 semantics).

Everything is numpy + a seed: the same call produces the same bytes on any box.
"""
import numpy as np
import torch

from .nets import SMPL_PARENTS

N_VERTS = 6890
N_JOINTS = 24


def _rest_joints():
    s30, c30 = np.sin(np.pi / 6), np.cos(np.pi / 6)
    J = np.zeros((24, 3), dtype=np.float64)
    J[0] = (0.0, 0.0, 0.0)
    J[1] = (0.07, -0.09, 0.0)
    J[2] = (-0.07, -0.09, 0.0)
    J[3] = (0.0, 0.11, 0.0)
    leg_l = np.array([s30, -c30, 0.0])
    leg_r = np.array([-s30, -c30, 0.0])
    J[4] = J[1] + 0.40 * leg_l
    J[5] = J[2] + 0.40 * leg_r
    J[6] = (0.0, 0.25, 0.0)
    J[7] = J[4] + 0.40 * leg_l
    J[8] = J[5] + 0.40 * leg_r
    J[9] = (0.0, 0.33, 0.0)
    J[10] = J[7] + np.array([0.02, -0.05, 0.10])
    J[11] = J[8] + np.array([-0.02, -0.05, 0.10])
    J[12] = (0.0, 0.50, 0.0)
    J[13] = (0.08, 0.42, 0.0)
    J[14] = (-0.08, 0.42, 0.0)
    J[15] = (0.0, 0.60, 0.0)
    J[16] = (0.18, 0.44, 0.0)
    J[17] = (-0.18, 0.44, 0.0)
    arm_l = np.array([c30, -s30, 0.0])
    arm_r = np.array([-c30, -s30, 0.0])
    J[18] = J[16] + 0.27 * arm_l
    J[19] = J[17] + 0.27 * arm_r
    J[20] = J[18] + 0.25 * arm_l
    J[21] = J[19] + 0.25 * arm_r
    J[22] = J[20] + 0.08 * arm_l
    J[23] = J[21] + 0.08 * arm_r
    return J


# leaf joints get a short stub so that every joint owns at least one segment
_LEAF_TIPS = {10: (0.0, 0.0, 0.08), 11: (0.0, 0.0, 0.08), 15: (0.0, 0.12, 0.0),
              22: (0.06, -0.035, 0.0), 23: (-0.06, -0.035, 0.0)}
_RADIUS = {0: 0.11, 1: 0.075, 2: 0.075, 3: 0.12, 4: 0.055, 5: 0.055, 6: 0.125, 7: 0.04, 8: 0.04,
           9: 0.12, 10: 0.035, 11: 0.035, 12: 0.055, 13: 0.06, 14: 0.06, 15: 0.09, 16: 0.05,
           17: 0.05, 18: 0.04, 19: 0.04, 20: 0.033, 21: 0.033, 22: 0.03, 23: 0.03}


def body_segments():
    """Capsule list: (owner joint, p0 (3,), p1 (3,), radius). Segment j->child is owned by j."""
    J = _rest_joints()
    segs = []
    for c in range(1, 24):
        p = SMPL_PARENTS[c]
        segs.append((p, J[p], J[c], 0.5 * (_RADIUS[p] + _RADIUS[c]) if p in (0, 3, 6, 9) else _RADIUS[p]))
    for j, tip in _LEAF_TIPS.items():
        segs.append((j, J[j], J[j] + np.asarray(tip), _RADIUS[j]))
    return segs


def _seg_dist(x, p0, p1):
    d = p1 - p0
    t = np.clip(((x - p0) @ d) / (d @ d), 0.0, 1.0)
    return np.linalg.norm(x - (p0 + t[:, None] * d), axis=-1)


def capsule_union_sdf(x):
    """Signed distance (metres, canonical space) to the union of the body capsules. x: (P,3)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.full(x.shape[0], np.inf)
    for _, p0, p1, r in body_segments():
        out = np.minimum(out, _seg_dist(x, p0, p1) - r)
    return out


def soft_skinning_weights(x, tau=0.035):
    """(P,24) weights: softmax over joints of -(distance to the joint's own capsules / tau)^2."""
    x = np.asarray(x, dtype=np.float64)
    d = np.full((x.shape[0], N_JOINTS), np.inf)
    for owner, p0, p1, r in body_segments():
        d[:, owner] = np.minimum(d[:, owner], np.maximum(_seg_dist(x, p0, p1) - r, 0.0))
    logit = -(d / tau) ** 2
    logit -= logit.max(axis=1, keepdims=True)
    w = np.exp(logit)
    return w / w.sum(axis=1, keepdims=True)


def _orthobasis(d):
    d = d / np.linalg.norm(d)
    a = np.array([1.0, 0.0, 0.0]) if abs(d[0]) < 0.9 else np.array([0.0, 1.0, 0.0])
    u = np.cross(d, a)
    u /= np.linalg.norm(u)
    return d, u, np.cross(d, u)


def canonical_body(seed=0):
    """Canonical vertices (6890,3), skinning weights (6890,24), joints (24,3); float32."""
    rng = np.random.RandomState(seed)
    segs = body_segments()
    area = np.array([2 * np.pi * r * np.linalg.norm(p1 - p0) + 4 * np.pi * r * r for _, p0, p1, r in segs])
    counts = np.floor(area / area.sum() * N_VERTS).astype(int)
    counts[np.argmax(counts)] += N_VERTS - counts.sum()
    pts = []
    for (_, p0, p1, r), n in zip(segs, counts):
        L = np.linalg.norm(p1 - p0)
        d, u, v = _orthobasis(p1 - p0)
        # uniform on the capsule surface: cylinder with prob ~ its area, else one of the two caps
        on_cyl = rng.rand(n) < (2 * np.pi * r * L) / (2 * np.pi * r * L + 4 * np.pi * r * r)
        phi = rng.rand(n) * 2 * np.pi
        t = rng.rand(n)
        cz = rng.rand(n) * 2 - 1  # cap: uniform direction on the sphere
        sr = np.sqrt(np.maximum(1 - cz * cz, 0))
        ring = np.cos(phi)[:, None] * u + np.sin(phi)[:, None] * v
        cyl = p0 + (t * L)[:, None] * d + r * ring
        cap_dir = sr[:, None] * ring + cz[:, None] * d
        cap = np.where((cz > 0)[:, None], p1, p0) + r * cap_dir
        pts.append(np.where(on_cyl[:, None], cyl, cap))
    verts = np.concatenate(pts, axis=0)
    # push interior points (capsule overlaps) onto the outer hull of the union
    eps = 1e-4
    for _ in range(4):
        sd = capsule_union_sdf(verts)
        g = np.stack([(capsule_union_sdf(verts + eps * e) - capsule_union_sdf(verts - eps * e)) / (2 * eps)
                      for e in np.eye(3)], axis=-1)
        verts = verts - sd[:, None] * g
    weights = soft_skinning_weights(verts)
    return verts.astype(np.float32), weights.astype(np.float32), _rest_joints().astype(np.float32)


def _rodrigues(aa):
    th = np.linalg.norm(aa, axis=-1, keepdims=True)
    k = aa / np.maximum(th, 1e-12)
    K = np.zeros(aa.shape[:-1] + (3, 3))
    K[..., 0, 1], K[..., 0, 2] = -k[..., 2], k[..., 1]
    K[..., 1, 0], K[..., 1, 2] = k[..., 2], -k[..., 0]
    K[..., 2, 0], K[..., 2, 1] = -k[..., 1], k[..., 0]
    th = th[..., None]
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pose_body(joints, pose_seed=0, sigma=0.25):
    """Random pose -> (A (24,4,4) canonical->posed bone transforms, local rots (24,3,3),
    posed joints (24,3)).  Root is turned by pi about z so that 'up' is -y (image convention)."""
    rng = np.random.RandomState(1000 + pose_seed)
    aa = rng.randn(24, 3) * sigma
    R = _rodrigues(aa)
    R[0] = _rodrigues(np.array([0.0, 0.0, np.pi])) @ _rodrigues(aa[0] * 0.5)
    J = joints.astype(np.float64)
    G = np.zeros((24, 4, 4))
    for j in range(24):
        L = np.eye(4)
        L[:3, :3] = R[j]
        p = SMPL_PARENTS[j]
        L[:3, 3] = J[j] - (J[p] if p >= 0 else 0.0)
        G[j] = L if p < 0 else G[p] @ L
    A = G.copy()
    A[:, :3, 3] = G[:, :3, 3] - np.einsum("jab,jb->ja", G[:, :3, :3], J)
    return A.astype(np.float32), R.astype(np.float32), G[:, :3, 3].astype(np.float32)


def normalize_points_np(pts, coord_min, coord_max, center):
    pad = (coord_max - coord_min) * 0.05
    return ((pts - center - coord_min + pad) / (coord_max - coord_min) / 1.1 - 0.5) * 2.0


class SyntheticScene:
    """One subject: canonical body (fixed by ``seed``) that can be posed per frame."""

    def __init__(self, seed=0):
        self.verts_cano, self.weights, self.joints = canonical_body(seed)
        self.center = self.verts_cano.mean(axis=0).astype(np.float32)
        centered = self.verts_cano - self.center
        self.coord_max = np.float32(centered.max())
        self.coord_min = np.float32(centered.min())

    def frame(self, frame_idx=0, trans=(0.1, 0.0, 3.0)):
        A, R_local, J_posed = pose_body(self.joints, pose_seed=frame_idx)
        T = np.einsum("vj,jab->vab", self.weights.astype(np.float64), A.astype(np.float64))
        vh = np.concatenate([self.verts_cano, np.ones((N_VERTS, 1), np.float32)], axis=1).astype(np.float64)
        trans = np.asarray(trans, dtype=np.float32)
        verts_posed = (np.einsum("vab,vb->va", T, vh)[:, :3] + trans).astype(np.float32)
        return dict(bone_transforms=A, rots_local=R_local, joints_posed=J_posed + trans,
                    smpl_verts=verts_posed, trans=trans)

    def make_inputs(self, H, W, frame_idx=0, device="cpu", latent_idx=0, box_margin=0.05,
                    focal_scale=1.2, max_rays=None, eval_mode=True):
        """Input dict with the keys/shapes of compose_inputs (lightning_model.py:581-632), B=1."""
        fr = self.frame(frame_idx)
        K = np.array([[focal_scale * H, 0, W / 2.0], [0, focal_scale * H, H / 2.0], [0, 0, 1]], np.float64)
        bmin = fr["smpl_verts"].min(axis=0) - box_margin
        bmax = fr["smpl_verts"].max(axis=0) + box_margin
        corners = np.array([[x, y, z] for x in (bmin[0], bmax[0]) for y in (bmin[1], bmax[1])
                            for z in (bmin[2], bmax[2])], np.float64)
        uv = (K @ corners.T).T
        uv = uv[:, :2] / uv[:, 2:3]
        # projected-AABB pixel mask: convex hull of the 8 projected corners == its own bounding
        # polygon union (the reference fills the 6 faces); for an axis-aligned camera this is the
        # 2-D bounding box of the 4 near-face and 4 far-face corners.
        ys, xs = np.mgrid[0:H, 0:W]
        from scipy.spatial import Delaunay
        inside = Delaunay(uv).find_simplex(np.stack([xs.ravel(), ys.ravel()], -1).astype(np.float64)) >= 0
        y_inds, x_inds = ys.ravel()[inside], xs.ravel()[inside]
        pix = np.stack([x_inds, y_inds, np.ones_like(x_inds)], -1).astype(np.float64)
        d_cam = pix @ np.linalg.inv(K).T
        d = d_cam / np.linalg.norm(d_cam, axis=-1, keepdims=True)  # R = I: world == camera frame
        vd = d.copy()
        vd[(vd < 1e-5) & (vd > -1e-10)] = 1e-5
        vd[(vd > -1e-5) & (vd < 1e-10)] = -1e-5
        t0, t1 = bmin[None] / vd, bmax[None] / vd  # camera at the origin
        near = np.minimum(t0, t1).max(axis=-1)
        far = np.maximum(t0, t1).min(axis=-1)
        ok = near < far
        d, near, far, y_inds, x_inds = d[ok], near[ok], far[ok], y_inds[ok], x_inds[ok]
        if max_rays is not None and d.shape[0] > max_rays:
            sel = np.linspace(0, d.shape[0] - 1, max_rays).round().astype(int)
            d, near, far, y_inds, x_inds = d[sel], near[sel], far[sel], y_inds[sel], x_inds[sel]
        image_mask = np.zeros((H, W), bool)
        image_mask[y_inds, x_inds] = True
        N = d.shape[0]
        f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
        rots_full = fr["rots_local"].reshape(1, 24, 9)
        rots = rots_full.copy()
        rots[0, 0] = np.eye(3, dtype=np.float32).reshape(9)
        Jn = normalize_points_np(self.joints, self.coord_min, self.coord_max, self.center)
        pose = np.eye(4, dtype=np.float32)[None]
        inputs = {
            "intrinsics": f32(K[None]),
            "ray_dirs": f32(d[None]),
            "body_bounds_intersections": f32(np.stack([near, far], -1)[None]),
            "cam_loc": f32(np.zeros((1, 3))),
            "cam_rot": f32(np.eye(3)[None]),
            "cam_trans": f32(np.zeros((1, 3))),
            "pose": f32(pose),
            "body_mask": torch.ones(1, N, dtype=torch.bool, device=device),
            "smpl_verts": f32(fr["smpl_verts"][None]),
            "skinning_weights": f32(self.weights[None]),
            "bone_transforms": f32(fr["bone_transforms"][None]),
            "trans": f32(fr["trans"].reshape(1, 1, 3)),
            "coord_min": f32(np.array(self.coord_min).reshape(1, 1, 1)),
            "coord_max": f32(np.array(self.coord_max).reshape(1, 1, 1)),
            "center": f32(self.center.reshape(1, 1, 3)),
            "minimal_shape": f32(self.verts_cano[None]),
            "pose_cond": {"rots_full": f32(rots_full), "Jtrs_posed": f32(fr["joints_posed"][None]),
                          "latent_code_idx": torch.tensor([latent_idx], dtype=torch.int64, device=device)},
            "Jtrs": f32(Jn[None]),
            "rots": f32(rots),
            "cam_idx": torch.zeros(1, dtype=torch.int64, device=device),
            "geo_latent_code_idx": torch.tensor([latent_idx], dtype=torch.int64, device=device),
        }
        if eval_mode:
            inputs["image_mask"] = torch.from_numpy(image_mask[None]).to(device)
            inputs["ray_dirs_cam"] = f32(d[None])
        else:
            # training-only keys of compose_inputs (lightning_model.py:612-630); the dataset draws them with
            # igl / trimesh / libmesh (zju_mocap.py:461-543), here: seeded numpy on the capsule body
            rng = np.random.RandomState(77 + frame_idx)
            n_reg = 1024
            sel = rng.randint(0, N_VERTS, n_reg)
            surf = self.verts_cano[sel].astype(np.float64)
            eps = 1e-4
            grad = np.stack([(capsule_union_sdf(surf + eps * e) - capsule_union_sdf(surf - eps * e)) / (2 * eps)
                             for e in np.eye(3)], axis=-1)
            inside = surf - 0.01 * grad                                    # 1 cm under the surface
            inputs["rgb_values"] = f32((0.5 + 0.5 * np.sin(np.stack([7 * d[:, 0], 9 * d[:, 1], 5 * d[:, 0] + 3 * d[:, 1]], -1)))[None])
            inputs["points_skinning"] = f32(surf[None])
            inputs["sampled_weights"] = f32(self.weights[sel][None])
            inputs["points_inside"] = f32(normalize_points_np(inside, self.coord_min, self.coord_max, self.center)[None])
            inputs["points_uniform"] = f32((rng.rand(1, n_reg, 3) * 2 - 1))
        return inputs
