"""Data side of the hot path: the reference builds the model's inputs on CPU workers in numpy
(data/zju_mocap_odp.py:200-415: SMPL posing, projected-box pixel mask, ray directions, near/far) -- at ~20 ms per frame
of rendering that becomes the bottleneck of a sharded sequence, so here the same arithmetic runs as tensor
operations on the device the frame will be rendered on.  Also the readers of the on-disk formats the
reference's preprocessing writes (preprocess_datasets/preprocess_ZJU-MoCap.py:150-158, preprocess_aist.py:116-124):

    <subject>/cam_params.json       {"all_cam_names": [...], "<cam>": {"K": 3x3, "D": [...], "R": 3x3, "T": 3x1}}
    <subject>/models/*.npz          minimal_shape (6890,3), betas (1,10), Jtr_posed (24,3), bone_transforms (24,4,4),
                                    trans (3,), root_orient (3,), pose_body (63,), pose_hand (6,)
"""
import glob
import json
import os

import numpy as np
import torch

from . import smpl

MODEL_KEYS = ("minimal_shape", "betas", "Jtr_posed", "bone_transforms", "trans", "root_orient", "pose_body", "pose_hand")
MODEL_SHAPES = {"minimal_shape": (6890, 3), "Jtr_posed": (24, 3), "bone_transforms": (24, 4, 4), "trans": (3,),
                "root_orient": (3,), "pose_body": (63,), "pose_hand": (6,)}


def load_cam_params(path):
    """cam_params.json -> {"all_cam_names": [...], name: {"K","D","R","T"} as float32 arrays}."""
    with open(path, "r") as f:
        raw = json.load(f)
    if "all_cam_names" not in raw:
        raise ValueError("%s: no 'all_cam_names' entry" % path)
    out = {"all_cam_names": list(raw["all_cam_names"])}
    for name in out["all_cam_names"]:
        if name not in raw:
            raise ValueError("%s: camera %r listed but not described" % (path, name))
        c = raw[name]
        out[name] = {"K": np.asarray(c["K"], np.float32).reshape(3, 3), "D": np.asarray(c.get("D", []), np.float32).ravel(),
                     "R": np.asarray(c["R"], np.float32).reshape(3, 3), "T": np.asarray(c["T"], np.float32).reshape(3)}
    return out


def load_model_npz(path):
    """One frame's SMPL registration; raises on a missing key or a wrong shape (float16 minimal shapes are accepted)."""
    with np.load(path) as a:
        missing = [k for k in MODEL_KEYS if k not in a.files]
        if missing:
            raise ValueError("%s: missing %s" % (path, ", ".join(missing)))
        out = {k: np.asarray(a[k]) for k in MODEL_KEYS}
    for k, shp in MODEL_SHAPES.items():
        if tuple(out[k].reshape(shp).shape) != shp:
            raise ValueError("%s: %s has shape %s, expected %s" % (path, k, out[k].shape, shp))
        out[k] = out[k].reshape(shp).astype(np.float32)
    out["betas"] = out["betas"].astype(np.float32).reshape(1, -1)
    return out


def list_sequence(subject_dir, pose_dir="models", start_frame=0, end_frame=-1, sampling_rate=1):
    """(frame indices, model files) like the dataset constructor slices them (zju_mocap_odp.py:103-112)."""
    files = sorted(glob.glob(os.path.join(subject_dir, pose_dir, "*.npz")))
    frames = list(range(len(files)))
    sl = slice(start_frame, end_frame if end_frame > 0 else None, sampling_rate)
    return frames[sl], files[sl]


# ------------------------------------------------------------------------------------------------
# rays (utils/utils.py:16-73, zju_mocap_odp.py:160-180,285-315) on the device
# ------------------------------------------------------------------------------------------------
_QUADS = ((0, 1, 3, 2), (4, 5, 7, 6, 5), (0, 1, 5, 4), (2, 3, 7, 6), (0, 2, 6, 4), (1, 3, 7, 5))   # utils.py:48-53 verbatim


def bound_corners(bounds):
    (x0, y0, z0), (x1, y1, z1) = bounds[0], bounds[1]
    return torch.stack([torch.stack(c) for c in ((x0, y0, z0), (x0, y0, z1), (x0, y1, z0), (x0, y1, z1),
                                                 (x1, y0, z0), (x1, y0, z1), (x1, y1, z0), (x1, y1, z1))])


def bound_2d_mask(bounds, K, pose34, H, W):
    """Pixels of the projected bounding box (get_bound_2d_mask, utils.py:43-54): the eight corners are projected and
    ROUNDED to integers, then six polygons are filled -- one of them listed as (4,5,7,6,5), i.e. a triangle plus a
    segment, kept as written.  cv2.fillPoly is not in this image: a pixel is taken when its integer coordinate is
    inside or on the boundary of a polygon (exact integer arithmetic), which is cv2's rule for simple polygons."""
    dev = bounds.device
    c3 = bound_corners(bounds)
    xyz = c3 @ pose34[:, :3].t() + pose34[:, 3]
    xyz = xyz @ K.t()
    c2 = torch.round(xyz[:, :2] / xyz[:, 2:]).to(torch.int64)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    mask = torch.zeros(H, W, dtype=torch.bool, device=dev)
    for quad in _QUADS:
        poly = c2[list(quad)]
        n = poly.shape[0]
        winding = torch.zeros(H, W, dtype=torch.int64, device=dev)
        on_edge = torch.zeros(H, W, dtype=torch.bool, device=dev)
        for i in range(n):
            (ax, ay), (bx, by) = poly[i], poly[(i + 1) % n]
            cross = (bx - ax) * (ys - ay) - (by - ay) * (xs - ax)
            up = (ay <= ys) & (by > ys) & (cross > 0)
            down = (ay > ys) & (by <= ys) & (cross < 0)
            winding += up.to(torch.int64) - down.to(torch.int64)
            within = (xs >= torch.minimum(ax, bx)) & (xs <= torch.maximum(ax, bx)) & (ys >= torch.minimum(ay, by)) & \
                (ys <= torch.maximum(ay, by))
            on_edge |= (cross == 0) & within
        mask |= (winding != 0) | on_edge
    return mask


def near_far(bounds, ray_o, ray_d):
    """get_near_far (utils.py:56-73): slab test against the axis-aligned box, with the reference's epsilon rules."""
    norm = ray_d.norm(dim=-1, keepdim=True)
    v = ray_d / norm
    v = torch.where((v < 1e-5) & (v > -1e-10), torch.full_like(v, 1e-5), v)
    v = torch.where((v > -1e-5) & (v < 1e-10), torch.full_like(v, -1e-5), v)
    tmin = (bounds[:1] - ray_o[:1]) / v
    tmax = (bounds[1:2] - ray_o[:1]) / v
    near = torch.minimum(tmin, tmax).max(dim=-1)[0]
    far = torch.maximum(tmin, tmax).min(dim=-1)[0]
    ok = near < far
    return near / norm[..., 0], far / norm[..., 0], ok


def frame_item(model, camera, body, img_size, orig_img_size, box_margin=0.05, device="cpu", cam_idx=0, frame_idx=0,
               data_idx=0, gender="neutral"):
    """Device-side counterpart of ZJUMOCAPODPDataset.__getitem__ in test mode (zju_mocap_odp.py:200-415) + the default
    collate (leading batch dimension of 1): the flat 'image.*' / 'inputs.*' dict LightningModel.compose_inputs takes.
    model: load_model_npz dict (numpy); camera: {"K","R","T"}; body: smpl.BodyModel (numpy arrays)."""
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=device)
    H, W = (img_size, img_size) if np.isscalar(img_size) else img_size
    oh, ow = (orig_img_size, orig_img_size) if np.isscalar(orig_img_size) else orig_img_size
    K, R, T = f32(camera["K"]).clone(), f32(camera["R"]), f32(camera["T"]).reshape(3)
    cam_loc = -R.t() @ T
    side = float(max(oh, ow))
    K[0, 0], K[1, 1] = K[0, 0] / side * max(H, W), K[1, 1] / side * max(H, W)
    K[:2, 2] = K[:2, 2] / side * max(H, W)
    K_inv = torch.linalg.inv(K)
    trans, minimal_shape = f32(model["trans"]), f32(model["minimal_shape"])
    bone_transforms = f32(model["bone_transforms"])
    pose = torch.cat([f32(model["root_orient"]), f32(model["pose_body"]), f32(model["pose_hand"])]).reshape(-1, 3)
    rot_full = _rotvec_to_matrix(pose)                                  # scipy Rotation.from_rotvec(...).as_matrix()
    rots = torch.cat([torch.eye(3, device=device).unsqueeze(0), rot_full[1:]], dim=0).reshape(24, 9)
    J_regressor, posedirs, weights = f32(body.J_regressor), f32(body.posedirs), f32(body.lbs_weights)
    Jtr = J_regressor @ minimal_shape
    pose_feature = (rot_full[1:] - torch.eye(3, device=device)).reshape(207)
    minimal_shape = minimal_shape + (pose_feature @ posedirs).reshape(-1, 3)
    Tv = (weights @ bone_transforms.reshape(24, 16)).reshape(-1, 4, 4)
    verts = torch.einsum("vij,vj->vi", Tv[:, :3, :3], minimal_shape) + Tv[:, :3, 3] + trans
    bounds = torch.stack([verts.min(dim=0)[0] - box_margin, verts.max(dim=0)[0] + box_margin])
    mask = bound_2d_mask(bounds, K, torch.cat([R, T.reshape(3, 1)], dim=1), H, W)
    y_inds, x_inds = torch.nonzero(mask, as_tuple=True)
    homo = torch.stack([x_inds.float(), y_inds.float(), torch.ones_like(x_inds, dtype=torch.float32)], dim=-1)
    uv = homo @ K_inv.t()
    rays_cam = uv / (uv.norm(dim=-1, keepdim=True) + 1e-12)
    rays = uv @ R
    rays = rays / (rays.norm(dim=-1, keepdim=True) + 1e-12)
    near, far, ok = near_far(bounds, cam_loc.expand_as(rays), rays)
    image_mask = torch.zeros(H, W, dtype=torch.bool, device=device)
    image_mask[y_inds[ok], x_inds[ok]] = True
    b02v = smpl.get_transforms_02v(Jtr)
    T02 = (weights @ b02v.reshape(24, 16)).reshape(-1, 4, 4)
    shape_v = torch.einsum("vij,vj->vi", T02[:, :3, :3], minimal_shape) + T02[:, :3, 3]
    center = shape_v.mean(dim=0)
    cmax, cmin = (shape_v - center).max(), (shape_v - center).min()
    pad = (cmax - cmin) * 0.05
    Jn = (((Jtr - center) - cmin + pad) / (cmax - cmin) / 1.1 - 0.5) * 2.0
    n_rays = int(ok.sum())
    b = lambda t: t.unsqueeze(0)
    return {
        "image.trans": b(trans), "image.bone_transforms": b(bone_transforms), "image.bone_transforms_02v": b(b02v),
        "image.coord_max": b(cmax), "image.coord_min": b(cmin), "image.center": b(center),
        "image.minimal_shape": b(shape_v), "image.smpl_vertices": b(verts), "image.skinning_weights": b(weights),
        "image.root_orient": b(f32(model["root_orient"])), "image.pose_hand": b(f32(model["pose_hand"])),
        "image.pose_body": b(f32(model["pose_body"])), "image.rots": b(rots), "image.Jtrs": b(Jn),
        "image.rots_full": b(rot_full.reshape(24, 9)), "image.Jtrs_posed": b(f32(model["Jtr_posed"])),
        "image.center_cam": b(K[:2, 2].clone()), "image.focal_length": b(torch.stack([K[0, 0], K[1, 1]])),
        "image.K": b(K), "image.R": b(R), "image.T": b(T), "image.cam_loc": b(cam_loc),
        "inputs": torch.zeros(1, n_rays, 3, device=device), "inputs.mask": torch.ones(1, n_rays, dtype=torch.bool, device=device),
        "inputs.mask_erode": torch.ones(1, n_rays, dtype=torch.bool, device=device), "inputs.uv": b(uv[ok]),
        "inputs.ray_dirs": b(rays[ok]), "inputs.ray_dirs_cam": b(rays_cam[ok]),
        "inputs.body_bounds_intersections": b(torch.stack([near[ok], far[ok]], dim=-1)), "inputs.gender": [gender],
        "inputs.img_height": torch.tensor([H]), "inputs.img_width": torch.tensor([W]),
        "inputs.cam_idx": torch.tensor([cam_idx], device=device), "inputs.frame_idx": torch.tensor([frame_idx]),
        "inputs.data_idx": torch.tensor([data_idx], device=device), "inputs.novel_seq": torch.tensor([True]),
        "inputs.image_mask": b(image_mask),
    }


def _rotvec_to_matrix(rv):
    """scipy.spatial.transform.Rotation.from_rotvec(rv).as_matrix() through the quaternion, in float32 on the device."""
    angle = rv.norm(dim=1)
    small = angle <= 1e-3
    a2 = angle * angle
    scale = torch.where(small, 0.5 - a2 / 48 + a2 * a2 / 3840, torch.sin(angle / 2) / angle.clamp_min(1e-30))
    q = torch.cat([rv * scale.unsqueeze(1), torch.cos(angle / 2).unsqueeze(1)], dim=1)
    return smpl.quaternion_to_rotation_matrix_xyzw(q)


# ---------------------------------------------------------------------------------------------------------------------
# training-only samplers (data/zju_mocap.py:455-543) on the device: regularisation points off the canonical
# SMPL surface, skinning supervision points on it, points inside it
# ---------------------------------------------------------------------------------------------------------------------
def sample_surface(verts, faces, count, generator=None):
    """``trimesh.Trimesh.sample(count, return_index=True)`` (trimesh 3.9, not in the reference tree; restated from
    trimesh/sample.py sample_surface): faces drawn proportionally to area by a search in the cumulative areas, the point is
    origin + r1 e1 + r2 e2 with (r1, r2) uniform in the unit square, reflected (|r - 1|) when r1 + r2 > 1.
    verts (V,3), faces (F,3) integer -> points (count,3), face index (count,)."""
    dev = verts.device
    tri = verts[faces.long()]
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    area = torch.linalg.cross(e1, e2).norm(dim=-1) * 0.5
    cum = torch.cumsum(area, 0)
    pick = torch.rand(count, device=dev, generator=generator) * cum[-1]
    fi = torch.searchsorted(cum, pick).clamp(max=faces.shape[0] - 1)
    r = torch.rand(count, 2, device=dev, generator=generator)
    flip = r.sum(1) > 1.0
    r = torch.where(flip[:, None], (r - 1.0).abs(), r)
    return tri[fi, 0] + r[:, :1] * e1[fi] + r[:, 1:] * e2[fi], fi


def _choose(n_have, n_want, dev, generator):
    """np.random.choice(n_have, size=n_want, replace=n_have < n_want) (zju_mocap.py:472-475)."""
    if n_have >= n_want:
        return torch.randperm(n_have, device=dev, generator=generator)[:n_want]
    return torch.randint(0, n_have, (n_want,), device=dev, generator=generator)


def training_samples(minimal_shape_v, faces, skinning_weights, coord_min, coord_max, center, sample_reg_surface=False,
                     sample_inside=False, off_surface_thr=0.2, inside_thr=0.001, generator=None):
    """The per-item point sets of the training losses (zju_mocap.py:455-543), from the canonical (Vitruvian-pose) SMPL mesh:

      points_uniform   (1024,3) normalised: uniform in [-1,1]^3, outside the mesh and farther than the threshold from it
                       (the reference compares the SQUARED distance igl returns with ``off_surface_thr``; kept);
      points_skinning / sampled_weights: 1024 surface samples with barycentrically interpolated skinning weights
                       (sample_reg_surface) or the 24 part centroids with one-hot weights;
      points_inside    (1024,3) normalised (sample_inside): surface samples displaced by N(0, 0.5 m) that fall inside the
                       mesh, not nearer than ``inside_thr`` (squared) to it and not on a hand part, plus the 22 part centroids.

    Containment, closest face and barycentric weights come from one device call per point set (hip.mesh_query: the
    reference's libmesh test restated exactly; igl's closest-point query).  Random draws: a torch generator stands in for
    numpy's global state.  minimal_shape_v (6890,3) float32 on the GPU, faces (13776,3) int32, skinning_weights (6890,24)."""
    from . import hip
    dev = minimal_shape_v.device
    v = minimal_shape_v.float().contiguous()
    f = faces.to(torch.int32).contiguous()
    cmin, cmax, cen = coord_min.reshape(()), coord_max.reshape(()), center.reshape(1, 3)
    span = cmax - cmin
    normalize = lambda p: (((p - cen) - cmin + span * 0.05) / span / 1.1 - 0.5) * 2.0
    unnormalize = lambda p: (p / 2.0 + 0.5) * 1.1 * span + cmin - span * 0.05 + cen
    part_idx = skinning_weights.argmax(-1)
    onehot = torch.nn.functional.one_hot(part_idx, 24).float()
    centroids = (onehot.t() @ v) / onehot.sum(0).clamp(min=1.0)[:, None]           # per-part mean vertex (:457-459,503-506)

    def interpolated_weights(face, bary):
        vid = f[face.long()].long()
        return (skinning_weights[vid] * bary.float()[..., None]).sum(1)            # :485-486

    out = {}
    uniform = torch.rand(4096, 3, device=dev, generator=generator) * 2.0 - 1.0     # :464 / :491
    query = unnormalize(uniform)
    if sample_reg_surface:
        pts_skin, _ = sample_surface(v, f, 1024, generator)                         # :468
        d2, face, _, bary, inside = hip.mesh_query(v, f, torch.cat([query, pts_skin], 0))
        keep = (~inside[:4096]) & (d2[:4096] > off_surface_thr)                    # :471
        out["points_skinning"] = pts_skin
        out["sampled_weights"] = interpolated_weights(face[4096:], bary[4096:])
    else:
        d2, _, _, _, inside = hip.mesh_query(v, f, query)
        keep = (~inside) & (d2 > off_surface_thr)                                  # :495
        out["points_skinning"] = centroids
        out["sampled_weights"] = torch.eye(24, device=dev)
    cand = uniform[keep]
    if cand.shape[0] == 0:
        raise ValueError("no regularisation point survives the off-surface threshold")   # numpy's choice raises here too
    out["points_uniform"] = cand[_choose(cand.shape[0], 1024, dev, generator)]
    if sample_inside:
        pts, _ = sample_surface(v, f, 4096, generator)                              # :517
        pts = pts + torch.randn(pts.shape, device=dev, generator=generator) * 0.5
        d2, face, _, bary, inside = hip.mesh_query(v, f, pts)
        part = interpolated_weights(face, bary).argmax(-1)                          # :528-529
        ok = inside & (part != 22) & (part != 23) & (d2 >= inside_thr)             # :521,530
        pts = torch.cat([pts[ok], centroids[:22]], 0)                               # :532-535 (22 body parts, no hands)
        out["points_inside"] = normalize(pts[_choose(pts.shape[0], 1024, dev, generator)])
    return out


def training_rays(image, mask, mask_erode, bounds, K, R, T, num_fg_samples=1024, num_bg_samples=1024, generator=None):
    """Pixel / ray sampling of a TRAINING item (zju_mocap.py:330-400, sampling == 'default') on the device of its inputs.

    image (H,W,3) float in [0,1] (undistorted, resized), mask / mask_erode (H,W) integer (the eroded mask is 1 on the body,
    0 on the background, 100 on the rim the reference leaves out); bounds (2,3) posed-body box, K (3,3) of the resized image,
    R (3,3), T (3,).  As the reference: num_fg + 1024 pixels with mask_erode == 1 and num_bg + 1024 pixels of the projected
    box with mask_erode == 0 are drawn without replacement, rays through them are intersected with the box, and num_fg /
    num_bg of those that hit it are kept (the 1024 spare ones absorb rays that graze the box, near > far).  Background
    pixels are black.  -> dict with the 'inputs*' entries of the item (leading batch dimension 1)."""
    dev = image.device
    H, W = mask.shape
    K_inv = torch.linalg.inv(K)
    cam_loc = -R.t() @ T
    fg_sample, bg_sample = mask_erode == 1, mask_erode == 0

    def pick(flat_idx, want):
        if flat_idx.shape[0] < want:   # np.random.choice(replace=False) raises here as well
            raise ValueError("Cannot take a larger sample than population when 'replace=False'")
        return flat_idx[torch.randperm(flat_idx.shape[0], device=dev, generator=generator)[:want]]

    fg = pick(torch.nonzero(fg_sample.reshape(-1)).reshape(-1), num_fg_samples + 1024)
    box = bound_2d_mask(bounds, K, torch.cat([R, T.reshape(3, 1)], dim=1), H, W)
    bg = pick(torch.nonzero((box & bg_sample).reshape(-1)).reshape(-1), num_bg_samples + 1024)
    idx = torch.cat([fg, bg])
    ys, xs = torch.div(idx, W, rounding_mode="floor"), idx % W
    pixels = image[ys, xs].clone()
    pixels[fg.shape[0]:] = 0.0
    m = mask[ys, xs] != 0
    me = mask_erode[ys, xs] != 0
    homo = torch.stack([xs.float(), ys.float(), torch.ones_like(xs, dtype=torch.float32)], dim=-1)
    uv = homo @ K_inv.t()
    rays_cam = uv / (uv.norm(dim=-1, keepdim=True) + 1e-12)
    rays = uv @ R
    rays = rays / (rays.norm(dim=-1, keepdim=True) + 1e-12)
    near, far, ok = near_far(bounds, cam_loc.expand_as(rays), rays)
    n_fg = fg.shape[0]
    keep_fg = pick(torch.nonzero(ok[:n_fg]).reshape(-1), num_fg_samples)
    keep_bg = pick(torch.nonzero(ok[n_fg:]).reshape(-1), num_bg_samples) + n_fg
    keep = torch.cat([keep_fg, keep_bg])
    b = lambda t: t[keep].unsqueeze(0)
    return {"inputs": b(pixels), "inputs.mask": b(m), "inputs.mask_erode": b(me), "inputs.uv": b(uv),
            "inputs.ray_dirs": b(rays), "inputs.ray_dirs_cam": b(rays_cam),
            "inputs.body_bounds_intersections": torch.stack([near[keep], far[keep]], dim=-1).unsqueeze(0)}


def training_item(model, camera, body, faces, image, mask, mask_erode, img_size, orig_img_size, box_margin=0.05,
                  num_fg_samples=1024, num_bg_samples=1024, sample_reg_surface=False, sample_inside=False,
                  off_surface_thr=0.2, inside_thr=0.001, device="cuda", generator=None, **ids):
    """A whole training item (ZJUMOCAPDataset.__getitem__, mode 'train', zju_mocap.py:226-600) on the device: the frame
    composition of `frame_item`, the pixel / ray sample of `training_rays` in place of the full box, and the point sets of
    `training_samples`.  image / mask / mask_erode: already undistorted and resized (cv2 is not part of this build)."""
    item = frame_item(model, camera, body, img_size, orig_img_size, box_margin=box_margin, device=device, **ids)
    f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=device)
    verts = item["image.smpl_vertices"][0]
    bounds = torch.stack([verts.min(dim=0)[0] - box_margin, verts.max(dim=0)[0] + box_margin])
    rays = training_rays(torch.as_tensor(image, dtype=torch.float32, device=device), torch.as_tensor(mask, device=device),
                         torch.as_tensor(mask_erode, device=device), bounds, item["image.K"][0], item["image.R"][0],
                         item["image.T"][0], num_fg_samples, num_bg_samples, generator)
    item.update(rays)
    item.pop("inputs.image_mask", None)
    item.pop("inputs.novel_seq", None)   # the training dataset never emits the key (zju_mocap.py); its PRESENCE is what
                                         # compose_inputs tests (lightning_model.py:497-498) before dropping the frame index
    pts = training_samples(item["image.minimal_shape"][0], torch.as_tensor(np.asarray(faces), device=device),
                           f32(body.lbs_weights), item["image.coord_min"][0], item["image.coord_max"][0],
                           item["image.center"][0], sample_reg_surface, sample_inside, off_surface_thr, inside_thr, generator)
    item["image.points_uniform"] = pts["points_uniform"].unsqueeze(0)
    item["image.points_skinning"] = pts["points_skinning"].unsqueeze(0)
    item["image.sampled_weights"] = pts["sampled_weights"].unsqueeze(0)
    if "points_inside" in pts:
        item["image.points_inside"] = pts["points_inside"].unsqueeze(0)
    return item


# ---------------------------------------------------------------------------------------------------------------------
# the out-of-distribution-pose test dataset (reference ZJUMOCAPODPDataset, data/zju_mocap_odp.py:20-120) and its factory
# ---------------------------------------------------------------------------------------------------------------------
class SequenceDataset:
    """<dataset_folder>/<subject>/cam_params.json + <dataset_folder>/<subject>/<pose_dir>/*.npz, enumerated camera-major
    like the reference (``.cameras``, ``.cam_names``, ``.data`` -- what get_model(cfg, dataset=...) reads); ``item(idx,
    device)`` composes the frame on the device instead of in a numpy worker."""

    def __init__(self, dataset_folder, subjects, pose_dir, body=None, mode="test", orig_img_size=(1024, 1024), img_size=(512, 512),
                 num_fg_samples=1024, num_bg_samples=1024, sampling_rate=1, start_frame=0, end_frame=-1, views=(),
                 box_margin=0.05, body_models="body_models/misc"):
        """The reference's keyword arguments (zju_mocap_odp.py:24-38; mode / num_*_samples are accepted and unused, as
        there); body: smpl.BodyModel, default the neutral SMPL model under `body_models` like the reference (:40-58)."""
        if len(subjects) != 1:
            raise AssertionError("one subject per dataset, like the reference (zju_mocap_odp.py:81)")
        if body is None:
            body = smpl.BodyModel.from_files("neutral", body_models)
        as2 = lambda v: (int(v), int(v)) if np.isscalar(v) else tuple(v)
        self.mode = mode
        self.body, self.img_size, self.orig_img_size, self.box_margin = body, as2(img_size), as2(orig_img_size), box_margin
        subject_dir = os.path.join(dataset_folder, subjects[0])
        self.cameras = load_cam_params(os.path.join(subject_dir, "cam_params.json"))
        self.cam_names = list(views) if len(views) else list(self.cameras["all_cam_names"])
        for c in self.cam_names:
            if c not in self.cameras:
                raise KeyError("camera %r is not in %s" % (c, os.path.join(subject_dir, "cam_params.json")))
        frames, files = list_sequence(subject_dir, pose_dir, start_frame, end_frame, sampling_rate)
        if not files:
            raise FileNotFoundError("no *.npz under %s" % os.path.join(subject_dir, pose_dir))
        self.data = [{"subject": subjects[0], "gender": "neutral", "cam_idx": ci, "cam_name": c, "frame_idx": f, "data_idx": d,
                      "model_file": mf} for ci, c in enumerate(self.cam_names) for d, (f, mf) in enumerate(zip(frames, files))]

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        """Host-resident item WITH the leading batch dimension of 1 (use DataLoader(batch_size=None)); `item(idx, device)`
        composes the same dict directly on the GPU."""
        return self.item(idx, "cpu")

    def item(self, idx, device):
        d = self.data[idx]
        return frame_item(load_model_npz(d["model_file"]), self.cameras[d["cam_name"]], self.body, self.img_size,
                          self.orig_img_size, box_margin=self.box_margin, device=device, cam_idx=d["cam_idx"],
                          frame_idx=d["frame_idx"], data_idx=d["data_idx"], gender=d["gender"])


def get_dataset(mode, cfg, body):
    """im2mesh.config.get_dataset for dataset type 'zju_mocap_odp' (reference im2mesh/config.py:78-262): split, views and
    frame range of `mode` from cfg['data'], 1024 -> 512 images."""
    d = cfg["data"]
    if d["dataset"] != "zju_mocap_odp":
        raise ValueError('Invalid dataset "%s" (this build reads the pose-sequence format zju_mocap_odp)' % d["dataset"])
    if mode not in ("train", "val", "test"):
        raise ValueError("Invalid mode %r" % mode)
    return SequenceDataset(d["path"], d[mode + "_split"], d["pose_dir"], body=body, mode=mode, img_size=(512, 512), orig_img_size=(1024, 1024),
                           sampling_rate=d[mode + "_subsampling_rate"], start_frame=d[mode + "_start_frame"],
                           end_frame=d[mode + "_end_frame"], views=d[mode + "_views"], box_margin=d["box_margin"])


class TrainingDataset:
    """The training dataset (reference ZJUMOCAPDataset, data/zju_mocap.py:20-150): <subject>/models/*.npz,
    <subject>/<camera>/*.jpg (images) and *.png (masks), <subject>/cam_params.json, enumerated camera-major.  ``item(idx,
    device)`` reads the image pair (PIL), prepares it on the device (imageops: undistort, resize, mask rim), samples pixels /
    rays and the regularisation point sets (training_item)."""

    def __init__(self, dataset_folder, subjects=("CoreView_313",), mode="train", img_size=(512, 512), num_fg_samples=1024,
                 num_bg_samples=1024, sampling_rate=1, start_frame=0, end_frame=-1, views=(), off_surface_thr=0.2,
                 inside_thr=0.001, box_margin=0.05, sampling="default", sample_reg_surface=False, sample_inside=False,
                 erode_mask=True, body=None, faces=None, body_models="body_models/misc"):
        if len(subjects) != 1:
            raise AssertionError("one subject per dataset, like the reference (zju_mocap.py:95)")
        if sampling != "default":
            raise ValueError("Sampling strategy {} is not supported!".format(sampling))          # zju_mocap.py:401
        self.mode, self.sampling = mode, sampling
        self.img_size = (int(img_size), int(img_size)) if np.isscalar(img_size) else tuple(img_size)
        self.num_fg_samples, self.num_bg_samples = num_fg_samples, num_bg_samples
        self.off_surface_thr, self.inside_thr, self.box_margin = off_surface_thr, inside_thr, box_margin
        self.sample_reg_surface, self.sample_inside, self.erode_mask = sample_reg_surface, sample_inside, erode_mask
        self.body = body if body is not None else smpl.BodyModel.from_files(self._gender(subjects[0]), body_models)
        self.faces = faces if faces is not None else np.load(os.path.join(body_models, "faces.npz"))["faces"]
        subject_dir = self._subject_dir(dataset_folder, subjects[0])
        self.cameras = self._load_cameras(subject_dir)
        self.cam_names = list(views) if len(views) else list(self.cameras["all_cam_names"])
        sl = slice(start_frame, end_frame if end_frame > 0 else None, sampling_rate)
        model_files = sorted(glob.glob(os.path.join(subject_dir, "models/*.npz")))[sl]
        self.data = []
        for ci, cam in enumerate(self.cam_names):
            img_dir, mask_dir = self._image_dirs(subject_dir, cam)
            all_imgs = sorted(glob.glob(os.path.join(img_dir, "*.jpg")))
            frames = list(range(len(all_imgs)))[sl]
            imgs, masks = all_imgs[sl], sorted(glob.glob(os.path.join(mask_dir, "*.png")))[sl]
            if not (len(model_files) == len(imgs) == len(masks)):
                raise AssertionError("camera %s: %d images, %d masks for %d model files" % (cam, len(imgs), len(masks),
                                                                                             len(model_files)))
            for d_idx, (f_idx, img, msk, mf) in enumerate(zip(frames, imgs, masks, model_files)):
                self.data.append({"subject": subjects[0], "gender": self._gender(subjects[0]), "cam_idx": ci, "cam_name": cam,
                                  "frame_idx": f_idx, "data_idx": d_idx, "img_file": img, "mask_file": msk, "model_file": mf})

    # ---- what differs between the capture formats (zju_mocap.py / h36m.py / people_snapshot.py)
    def _subject_dir(self, dataset_folder, subject):
        return os.path.join(dataset_folder, subject)

    def _load_cameras(self, subject_dir):
        return load_cam_params(os.path.join(subject_dir, "cam_params.json"))

    def _image_dirs(self, subject_dir, cam):
        return os.path.join(subject_dir, cam), os.path.join(subject_dir, cam)

    def _gender(self, subject):
        return "neutral"

    def _rim(self, mask):
        from . import imageops
        return imageops.rim_mask(mask, self.erode_mask or self.mode in ("val", "test"))                  # zju_mocap.py:212

    def _prepare(self, image, mask, rim, K, D):
        """-> image (H,W,3) in [0,1], mask, rim at img_size, and the size the intrinsics refer to (zju_mocap.py:246-262)."""
        from . import imageops
        orig = (image.shape[0], image.shape[1])
        image, mask, rim = imageops.undistort(image, K, D), imageops.undistort(mask, K, D), imageops.undistort(rim, K, D)
        image = imageops.resize_linear(image, self.img_size) / 255.0
        return image, imageops.resize_nearest(mask, self.img_size), imageops.resize_nearest(rim, self.img_size), orig

    def __len__(self):
        return len(self.data)

    def item(self, idx, device, generator=None):
        from PIL import Image
        d = self.data[idx]
        cam = self.cameras[d["cam_name"]]
        image = torch.as_tensor(np.array(Image.open(d["img_file"]).convert("RGB")), device=device).float()
        mask = torch.as_tensor(np.array(Image.open(d["mask_file"]).convert("L")), device=device)
        K, D = torch.as_tensor(np.asarray(cam["K"], np.float32)), np.asarray(cam["D"], np.float64).ravel()
        image, mask, rim, orig = self._prepare(image, mask, self._rim(mask), K, D)
        return training_item(load_model_npz(d["model_file"]), cam, self.body, self.faces, image, mask, rim, self.img_size, orig,
                             box_margin=self.box_margin, num_fg_samples=self.num_fg_samples,
                             num_bg_samples=self.num_bg_samples, sample_reg_surface=self.sample_reg_surface,
                             sample_inside=self.sample_inside, off_surface_thr=self.off_surface_thr,
                             inside_thr=self.inside_thr, device=device, generator=generator, cam_idx=d["cam_idx"],
                             frame_idx=d["frame_idx"], data_idx=d["data_idx"], gender=d["gender"])


class H36MDataset(TrainingDataset):
    """Human3.6M captures (reference data/h36m.py): the subject's files sit under <subject>/Posing, the image is reduced to
    img_size FIRST (INTER_AREA) and undistorted at that size with intrinsics that already refer to it, the mask rim is left
    out only while training (h36m.py:96,113,211,246-272)."""

    def __init__(self, dataset_folder, subjects=("S1",), mode="train", img_size=(1002, 1000), **kw):
        super().__init__(dataset_folder, subjects=subjects, mode=mode, img_size=img_size, **kw)

    def _subject_dir(self, dataset_folder, subject):
        return os.path.join(dataset_folder, subject, "Posing")

    def _rim(self, mask):
        from . import imageops
        return imageops.rim_mask(mask, self.erode_mask and self.mode not in ("val", "test"))

    def _prepare(self, image, mask, rim, K, D):
        from . import imageops
        image = imageops.resize_area(image, self.img_size)
        mask, rim = imageops.resize_nearest(mask, self.img_size), imageops.resize_nearest(rim, self.img_size)
        image, mask, rim = imageops.undistort(image, K, D), imageops.undistort(mask, K, D), imageops.undistort(rim, K, D)
        return image / 255.0, mask, rim, self.img_size            # intrinsics are for img_size: no rescaling


class PeopleSnapshotDataset(TrainingDataset):
    """People-Snapshot captures (reference data/people_snapshot.py): one camera from <subject>/camera.pkl (camera_f,
    camera_c, camera_k, identity pose), images under image/, masks under mask/, gender from the subject's name
    (people_snapshot.py:94-150,222-232)."""

    def __init__(self, dataset_folder, subjects=("female-3-casual",), mode="train", img_size=(1080, 1080), **kw):
        kw.pop("views", None)
        super().__init__(dataset_folder, subjects=subjects, mode=mode, img_size=img_size, **kw)

    def _load_cameras(self, subject_dir):
        import pickle
        with open(os.path.join(subject_dir, "camera.pkl"), "rb") as f:
            cam = pickle.load(f, encoding="latin1")
        K = np.zeros((3, 3), np.float32)
        K[0, 0], K[1, 1] = cam["camera_f"][0], cam["camera_f"][1]
        K[:2, 2] = cam["camera_c"]
        K[2, 2] = 1
        self.orig_img_size = (cam["height"], cam["width"])
        return {"all_cam_names": ["1"], "1": {"K": K, "D": np.asarray(cam["camera_k"], np.float32), "R": np.eye(3, dtype=np.float32),
                                              "T": np.zeros(3, np.float32)}}

    def _image_dirs(self, subject_dir, cam):
        return os.path.join(subject_dir, "image"), os.path.join(subject_dir, "mask")

    def _gender(self, subject):
        return "female" if "female" in subject else "male"
