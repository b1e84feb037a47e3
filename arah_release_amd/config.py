"""Factories with the reference's entry-point names (im2mesh/config.py:12-75,
im2mesh/metaavatar_render/config.py:147-302): ``load_config``, ``get_model``, ``method_dict``.

The reference's yaml files are data and are not shipped; the model-relevant keys of the three
config families it uses are restated in ``builtin_config``.  A user's own yaml (with recursive
``inherit_from``) loads through ``load_config`` exactly like in the reference.
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn

from . import nets
from .renderer import MetaAvatarRender

_ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

_SKIN_KW = {"d_in": 3, "d_out": 25, "d_hidden": 128, "n_layers": 4, "skip_in": [], "cond_in": [], "multires": 0,
            "bias": 1.0, "geometric_init": False, "weight_norm": True}
_SDF_KW = {"in_features": 3, "num_hidden_layers": 5, "hierarchical_pose": True, "hyper_in_ch": 144, "use_FiLM": True}
_DEFAULT = {
    "method": "metaavatar_render",
    "model": {"decoder": "hyper_bvp", "skinning_decoder": "deformer_mlp", "decoder_kwargs": _SDF_KW,
              "skinning_decoder_kwargs": _SKIN_KW, "renderer": "mlp", "latent_dim": 128, "train_cameras": False,
              "train_smpl": False, "geo_pose_encoder": "latent", "color_pose_encoder": "latent",
              "cano_view_dirs": True, "n_steps": 64, "near_surface_samples": 16, "far_surface_samples": 16,
              "render_last_pt": False},
    # configs/default.yaml:53-75 overridden by configs/arah-zju/ZJUMOCAP-*_4gpus.yaml:46-60
    "training": {"train_skinning_net": True, "pose_input_noise": True, "view_input_noise": True,
                 "nv_noise_type": "rotation", "lr": 1.0e-6, "skinning_lr": 1.0e-4, "pose_net_factor": 100,
                 "rgb_weight": 3.0e+1, "perceptual_weight": 0.0, "eikonal_weight": 5.0e+1, "mask_weight": 0.0,
                 "off_surface_weight": 1.0e+2, "inside_weight": 10.0, "params_weight": 1.0e+2,
                 "skinning_weight": 10.0, "rgb_loss_type": "l1", "batch_size": 1},
}
_IDR = {"mode": "idr", "d_in": 9, "d_out": 3, "d_hidden": 256, "n_layers": 5, "weight_norm": True, "multires": 0,
        "multires_view": 4, "skips": [3], "squeeze_out": True}
_NOVIEW = {"mode": "no_view_dir", "d_in": 6, "d_out": 3, "d_hidden": 256, "n_layers": 5, "weight_norm": True,
           "multires": 0, "multires_view": 0, "skips": [3], "squeeze_out": True}


def builtin_config(name, n_steps=64, near=16, far=16):
    """Model keys of configs/arah-zju/ZJUMOCAP-377-mono_4gpus.yaml:30-45 ('zju377_mono'),
    configs/arah-zju/ZJUMOCAP-313_4gpus.yaml:30-45 ('zju313') and configs/arah-h36m/H36M_S9_4gpus.yaml:29-44
    ('h36m')."""
    cfg = copy.deepcopy(_DEFAULT)
    if name == "zju377_mono":
        cfg["model"].update(renderer_kwargs=dict(_NOVIEW), cano_view_dirs=False)
    elif name == "zju313":
        cfg["model"].update(renderer_kwargs=dict(_IDR), cano_view_dirs=False)
    elif name == "h36m":
        cfg["model"].update(renderer_kwargs=dict(_IDR), cano_view_dirs=True)
    else:
        raise KeyError(name)
    cfg["model"].update(n_steps=n_steps, near_surface_samples=near, far_surface_samples=far)
    return cfg


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            _merge(dst.setdefault(k, {}), v)
        else:
            dst[k] = v


def load_config(path, default_path=None):
    """yaml + recursive ``inherit_from`` + defaults (reference im2mesh/config.py:12-56)."""
    import yaml
    with open(path, "r") as f:
        special = yaml.safe_load(f)
    parent = special.get("inherit_from")
    if parent is not None:
        cfg = load_config(parent, default_path)
    elif default_path is not None:
        with open(default_path, "r") as f:
            cfg = yaml.safe_load(f)
    else:
        cfg = {}
    _merge(cfg, special)
    return cfg


def _color_feature_dim(cfg):
    enc = cfg["model"]["color_pose_encoder"]
    extra = {None: 0, "leap": 144, "root": 12, "latent": cfg["model"]["latent_dim"],
             "hybrid": 12 + cfg["model"]["latent_dim"]}
    if enc not in extra:
        raise ValueError("Unsupported rendering network pose encoder %r" % enc)
    return 256 + extra[enc]


def _load_pretrained(module, path, prefix, what):
    """MetaAvatar initialisation (metaavatar_render/config.py:32-45,72-84): entries of ckpt['model'] under `prefix`
    (after an optional 'module.'), loaded non-strictly like the reference -- but a checkpoint that names the path and
    does not exist, or that contributes NOTHING to the module, is an error here instead of a silently untrained net."""
    if not os.path.exists(path):
        raise FileNotFoundError("%s initialisation %r not found (cfg['model'] names it; mode 'val'/'test' skips it)"
                                % (what, path))
    ckpt = torch.load(path, map_location="cpu")
    sd = {}
    for k, v in ckpt["model"].items():
        if k.startswith("module"):
            k = k[7:]
        if k.startswith(prefix):
            sd[k[len(prefix) + 1:]] = v
    res = module.load_state_dict(sd, strict=False)
    used = [k for k in sd if k not in res.unexpected_keys]
    if not used:
        raise ValueError("%s: no entry of %r matches the %s network" % (path, prefix, what))
    return res


def get_render_model(cfg, mode="test", low_vram=False, checkpoint_path=None, n_data_points=None, dataset=None,
                     **kwargs):
    """metaavatar_render.config.get_model (config.py:147-302): returns the bare MetaAvatarRender module.
    `dataset` (mode 'train'/'val'): object with ``cam_names``, ``cameras`` and ``data`` (list of dicts with 'cam_idx',
    'frame_idx', 'model_file', 'gender') as the reference's datasets expose them."""
    m = cfg["model"]
    init_weights = mode not in ("val", "test")                               # config.py:154-160
    sdf_decoder = nets.decoder_dict[m["decoder"]](**m["decoder_kwargs"])
    if init_weights and m.get("geometry_net"):
        _load_pretrained(sdf_decoder, m["geometry_net"], "decoder", "SDF")
    skin_dec = nets.decoder_dict[m["skinning_decoder"]](**m["skinning_decoder_kwargs"])
    if init_weights and m.get("skinning_net2"):
        _load_pretrained(skin_dec, m["skinning_net2"], "skinning_decoder_fwd", "skinning")
    skinning = nets.SkinningModel(skin_dec)
    color = nets.RenderingNetwork(d_feature=_color_feature_dim(cfg), pose_encoder=m["color_pose_encoder"],
                                  **m["renderer_kwargs"])
    deviation = nets.SingleVarianceNetwork(1e-3)
    model_kwargs = {}
    train_cameras = bool(m.get("train_cameras")) and mode in ("train", "val")
    if train_cameras:                                                        # config.py:166-177
        from scipy.spatial.transform import Rotation
        cams = dataset.cameras
        model_kwargs["cam_rots"] = np.stack([Rotation.from_matrix(np.asarray(cams[c]["R"])).as_quat().astype(np.float32)
                                             for c in dataset.cam_names], axis=0)
        model_kwargs["cam_trans"] = np.stack([np.asarray(cams[c]["T"], np.float32).ravel() for c in dataset.cam_names],
                                             axis=0)
    train_smpl = bool(m.get("train_smpl")) and mode in ("train", "val")
    if train_smpl:                                                           # config.py:179-224
        from . import data as data_mod
        acc = {k: [] for k in ("root_orient", "pose_body", "pose_hand", "trans", "frames")}
        cam_idx = dataset.data[0]["cam_idx"]
        betas, gender = None, None
        for d_idx, item in enumerate(dataset.data):
            if item["cam_idx"] != cam_idx:
                break                      # one batch = one frame: the parameters of one camera view are enough
            md = data_mod.load_model_npz(item["model_file"])
            for key, width in (("root_orient", 3), ("pose_body", 3), ("pose_hand", 3)):
                v = md[key].astype(np.float32).reshape(-1, width)
                v[(v == 0.0).all(axis=-1)] += 1e-8                          # exact zeros have no axis
                acc[key].append(v.reshape(-1))
            if d_idx == 0:
                betas, gender = md["betas"].astype(np.float32), item["gender"]
            acc["trans"].append(md["trans"].astype(np.float32))
            acc["frames"].append(item["frame_idx"])
        model_kwargs.update(acc, betas=betas, gender=gender, body_model=kwargs.get("body_model"))
    train_latent = m["color_pose_encoder"] in ("hybrid", "latent")
    train_geo_latent = m["geo_pose_encoder"] in ("latent",)
    ckpt = None
    if checkpoint_path is not None:
        ckpt = torch.load(checkpoint_path, map_location="cpu")
    if (train_latent or train_geo_latent) and mode in ("train", "val") and dataset is not None:   # config.py:233-250
        cam_idx = dataset.data[0]["cam_idx"]
        frames = []
        for item in dataset.data:
            if item["cam_idx"] != cam_idx:
                break
            frames.append(item["frame_idx"])
        model_kwargs.update(n_data_points=len(frames), frames=frames)
    elif train_latent or train_geo_latent:
        if ckpt is not None:
            n_data_points = ckpt["state_dict"]["model.latent.weight"].size(0)   # config.py:253-257
        if n_data_points is None:
            raise ValueError("need n_data_points or a checkpoint to size the latent embedding")
        model_kwargs.update(n_data_points=n_data_points)
        model_kwargs.setdefault("frames", [])
    t = cfg.get("training", {})
    model = MetaAvatarRender(sdf_decoder=sdf_decoder, skinning_model=skinning, color_decoder=color,
                             deviation_decoder=deviation, train_cameras=train_cameras, train_smpl=train_smpl,
                             train_latent_code=train_latent, train_geo_latent_code=train_geo_latent,
                             cano_view_dirs=m["cano_view_dirs"], near_surface_samples=m["near_surface_samples"],
                             far_surface_samples=m["far_surface_samples"], n_steps=m["n_steps"],
                             train_skinning_net=t.get("train_skinning_net", False),
                             render_last_pt=m["render_last_pt"], pose_input_noise=t.get("pose_input_noise", False),
                             view_input_noise=t.get("view_input_noise", False),
                             nv_noise_type=t.get("nv_noise_type", "rotation"), low_vram=low_vram, **model_kwargs)
    if ckpt is not None:   # Lightning prefixes the module with 'model.' (config.py:291-300)
        sd = {k[6:]: v for k, v in ckpt["state_dict"].items() if k.startswith("model.")}
        model.load_state_dict(sd, strict=False)
    return model


class LightningModel(nn.Module):
    """The model-facing half of the reference's Lightning harness (metaavatar_render/lightning_model.py:101-653):
    ``compose_inputs`` (dataset dict -> model inputs, on the device, incl. the train_cameras / train_smpl branches),
    ``compute_loss``, ``training_step`` / ``validation_step`` / ``test_step`` (image scatter included) and
    ``configure_optimizers`` with the reference's parameter groups.  ``model`` sits under the same attribute so that
    checkpoints keep the 'model.' prefix.  Metrics (LPIPS / SSIM), image writing and the Lightning runtime itself are
    the caller's."""

    def __init__(self, model, cfg, val_size=None):
        super().__init__()
        self.model = model
        self.cfg = cfg
        self.val_size = val_size
        from . import training
        self.criteria = training.build_loss(cfg) if "training" in cfg else None

    @property
    def device(self):
        return next(self.model.parameters()).device

    def forward(self, inputs, gen_cano_mesh=False, eval=True):
        return self.model(inputs, gen_cano_mesh=gen_cano_mesh, eval=eval)

    def configure_optimizers(self):
        from . import training
        return training.configure_optimizers(self.model, self.cfg)

    def compose_inputs(self, data, eval):
        """lightning_model.py:463-634, tensor for tensor; everything stays on the device `data` lives on."""
        from . import smpl, training
        model = self.model
        smpl_verts = data.get("image.smpl_vertices")
        cam_idx = data.get("inputs.cam_idx")
        if model.train_cameras and not eval:          # optimised extrinsics: rays from the stored pixel coordinates
            uv = data.get("inputs.uv")
            cam_rot = smpl.quaternion_to_rotation_matrix_xyzw(model.cam_rots[cam_idx])
            cam_trans = model.cam_trans[cam_idx]
            rays = torch.matmul(uv, cam_rot)
            ray_dirs = rays / (torch.norm(rays, p=2, dim=-1, keepdim=True) + 1e-12)
            cam_loc = torch.matmul(-cam_rot.transpose(1, 2), cam_trans.unsqueeze(-1)).squeeze(-1)
        else:
            ray_dirs, cam_loc = data.get("inputs.ray_dirs"), data.get("image.cam_loc")
            cam_rot, cam_trans = data.get("image.R"), data.get("image.T")
        B = smpl_verts.size(0)
        f_idx = int(data.get("inputs.frame_idx")[0])
        if data.get("inputs.novel_seq") is not None:
            f_idx = -1
        if model.train_smpl and f_idx in model.frames:
            bp = model.body_poses
            root_orient, pose_body = bp["root_orient_%d" % f_idx].unsqueeze(0), bp["pose_body_%d" % f_idx].unsqueeze(0)
            pose_hand, trans = bp["pose_hand_%d" % f_idx].unsqueeze(0), bp["trans_%d" % f_idx].unsqueeze(0)
            verts_posed, Jtrs, Jtrs_posed, bone_transforms, minimal_shape = model.forward_smpl(
                model.betas, root_orient, pose_body, pose_hand, trans)
            smpl_verts = (verts_posed + trans.unsqueeze(1)).repeat(B, 1, 1)
            b02v = smpl.get_transforms_02v(Jtrs.squeeze(0))
            T = torch.matmul(model.lbs_weights, b02v.reshape(-1, 16)).reshape(-1, 4, 4)
            shape_v = torch.matmul(T[:, :3, :3], minimal_shape.reshape(-1, 3, 1)).squeeze(-1) + T[:, :3, -1]
            center = torch.mean(shape_v, dim=0, keepdim=True)
            centred = shape_v - center
            center = center.view(1, 1, -1).repeat(B, 1, 1)
            coord_max = centred.max().view(1, 1, -1).repeat(B, 1, 1)
            coord_min = centred.min().view(1, 1, -1).repeat(B, 1, 1)
            minimal_shape = shape_v.reshape(1, -1, 3).repeat(B, 1, 1)
            b02v = b02v.unsqueeze(0).repeat(B, 1, 1, 1)
            bone_transforms = bone_transforms.repeat(B, 1, 1, 1)
            Jtrs = training.normalize_canonical_points(Jtrs, coord_min[:1], coord_max[:1], center[:1]).repeat(B, 1, 1)
            Jtrs_posed = Jtrs_posed + trans.unsqueeze(1)
            full_pose = torch.cat([root_orient, pose_body, pose_hand], dim=-1).reshape(-1, 3)
            full_mat = smpl.angle_axis_to_rotation_matrix(full_pose).reshape(-1, 9)
            local = torch.cat([torch.eye(3, device=full_mat.device).reshape(1, 9), full_mat[1:]], dim=0)
            rots, rots_full = local.unsqueeze(0).repeat(B, 1, 1), full_mat.unsqueeze(0)
        else:
            minimal_shape, rots, Jtrs = data.get("image.minimal_shape"), data.get("image.rots"), data.get("image.Jtrs")
            rots_full, Jtrs_posed = data.get("image.rots_full"), data.get("image.Jtrs_posed")
            coord_min = data.get("image.coord_min").view(B, 1, -1)
            coord_max = data.get("image.coord_max").view(B, 1, -1)
            center = data.get("image.center").view(B, 1, -1)
            bone_transforms, b02v = data.get("image.bone_transforms"), data.get("image.bone_transforms_02v")
            trans = data.get("image.trans").unsqueeze(1)
        bone_transforms = torch.matmul(bone_transforms, torch.inverse(b02v))   # Vitruvian canonical pose -> posed, no translation
        pose = torch.cat([nn.functional.pad(cam_rot, pad=(0, 0, 0, 1)),
                          nn.functional.pad(cam_trans, pad=(0, 1), value=1).unsqueeze(-1)], dim=-1)
        pose_cond = {"rots_full": rots_full, "Jtrs_posed": Jtrs_posed}

        def code_index():
            if f_idx in model.frames:
                return data.get("inputs.data_idx")[:1]
            return torch.tensor([model.latent.num_embeddings - 1], dtype=torch.int64, device=smpl_verts.device)

        if model.train_latent_code:
            pose_cond["latent_code_idx"] = code_index()
        inputs = {"intrinsics": data.get("image.K"), "ray_dirs": ray_dirs,
                  "body_bounds_intersections": data.get("inputs.body_bounds_intersections"), "cam_loc": cam_loc,
                  "cam_rot": cam_rot, "cam_trans": cam_trans, "pose": pose, "body_mask": data.get("inputs.mask_erode"),
                  "smpl_verts": smpl_verts, "skinning_weights": data.get("image.skinning_weights"),
                  "bone_transforms": bone_transforms, "trans": trans, "coord_min": coord_min, "coord_max": coord_max,
                  "center": center, "minimal_shape": minimal_shape, "pose_cond": pose_cond, "Jtrs": Jtrs, "rots": rots,
                  "cam_idx": cam_idx}
        if model.train_geo_latent_code:
            inputs["geo_latent_code_idx"] = code_index()
        if not eval:
            inputs["rgb_values"] = data.get("inputs")
            for key in ("sampled_weights", "points_skinning", "points_inside", "points_uniform"):
                if data.get("image." + key) is not None:
                    inputs[key] = data.get("image." + key)
        else:
            inputs["image_mask"] = data.get("inputs.image_mask")
            inputs["ray_dirs_cam"] = data.get("inputs.ray_dirs_cam")
        return inputs

    def compute_loss(self, data):
        """lightning_model.py:636-653."""
        inputs = self.compose_inputs(data, eval=False)
        out = self.model(inputs)
        gt = {"rgb": inputs["rgb_values"]}
        if "sampled_weights" in inputs:
            gt["sampled_weights"] = inputs["sampled_weights"]
        return self.criteria(out, gt)

    def training_step(self, data, data_idx=None):
        return self.compute_loss(data)["loss"]

    def render_image(self, data, gen_cano_mesh=False):
        """The model half of validation_step / test_step (lightning_model.py:158-181,300-330): forward in eval mode,
        then pred_pixels.masked_scatter_(image_mask, rgb) into a (B,H,W,3) image."""
        inputs = self.compose_inputs(data, eval=True)
        with torch.no_grad():
            out = self.model(inputs, gen_cano_mesh=gen_cano_mesh, eval=True)
        mask = inputs["image_mask"]
        img = torch.zeros(*mask.shape, 3, device=mask.device)
        img.masked_scatter_(mask.unsqueeze(-1), out["rgb_values"].reshape(-1, 3))
        out["image"] = img
        return out

    @staticmethod
    def normals_from_points(points_img):
        """Normal map from the camera-space surface points of a frame (H,W,3; zeros where the ray missed), by the
        reference's finite differences dz/dy, dz/dx (lightning_model.py:184-203): NaNs -- from 0 / 0 on the background and
        at the silhouette -- become -1 before the map to [0, 1]."""
        zs, xs, ys = points_img[:, :, 2], points_img[:, :, 0], points_img[:, :, 1]
        zy = (zs[1:, :] - zs[:-1, :]) / (ys[1:, :] - ys[:-1, :])
        zx = (zs[:, 1:] - zs[:, :-1]) / (xs[:, 1:] - xs[:, :-1])
        nrm = torch.zeros_like(points_img)
        nrm[:-1, :, 1] = -zy
        nrm[:, :-1, 0] = -zx
        nrm[:, :, 2] = 1
        nrm = nrm / torch.linalg.norm(nrm, dim=-1, keepdim=True)
        nrm[nrm.isnan()] = -1
        return ((nrm + 1) / 2.0).clip(0.0, 1.0)

    def validation_step(self, data, data_idx=None, ssim_fn=None, lpips_fn=None):
        """lightning_model.py:160-230: predicted image, normal map (the canonical-mesh one when the model produced it, else
        finite differences of the surface points), ground-truth image, PSNR (im2mesh/utils/eval.py:6-9).  SSIM and LPIPS
        come from skimage / lpips in the reference: pass callables (pred HxWx3, gt HxWx3, box mask) to have them filled."""
        import numpy as np
        out = self.render_image(data, gen_cano_mesh=False)
        mask = data.get("inputs.image_mask")
        n = int(mask.sum())
        pred_pixels = out["image"].squeeze(0)
        if "output_normal" in out:
            pred_normals = out["output_normal"].squeeze(0)
        else:
            pts = torch.zeros(*mask.shape, 3, device=mask.device)
            pts.masked_scatter_(mask.unsqueeze(-1), out["points_cam"].reshape(-1, 3)[:n])
            pred_normals = self.normals_from_points(pts.squeeze(0))
        image = data.get("inputs")
        gt_pixels = torch.zeros(*mask.shape, 3, device=mask.device)
        gt_pixels.masked_scatter_(mask.unsqueeze(-1), image.reshape(-1, 3)[:n])
        gt_pixels = gt_pixels.squeeze(0)
        pred_img = out["rgb_values"].reshape(-1, 3).detach().cpu().numpy()
        gt_img = image.reshape(-1, 3).detach().cpu().numpy()
        with np.errstate(divide="ignore"):   # identical images: the reference's formula gives inf (with numpy's warning)
            res = {"psnr": -10 * np.log(np.mean((pred_img - gt_img) ** 2)) / np.log(10)}
        box = mask.squeeze(0).detach().cpu().numpy()
        if ssim_fn is not None:
            res["ssim"] = ssim_fn(pred_pixels.detach().cpu().numpy(), gt_pixels.detach().cpu().numpy(), box)
        if lpips_fn is not None:
            res["lpips"] = lpips_fn(pred_pixels.detach().cpu().numpy(), gt_pixels.detach().cpu().numpy(), box)
        res.update({"rgb_pred": pred_pixels.permute(2, 0, 1), "normal_pred": pred_normals.permute(2, 0, 1),
                    "rgb_gt": gt_pixels.permute(2, 0, 1)})
        return res

    def test_step(self, data, data_idx=None):
        """lightning_model.py:306-338: the frame's image and the three normal maps of the canonical mesh, channels first
        (keys rgb_pred / normal_pred / normal_front / normal_back, as test_epoch_end takes them)."""
        if not isinstance(data, dict) or "inputs.ray_dirs" not in data:   # already-composed model inputs
            with torch.no_grad():
                return self.model(data, gen_cano_mesh=False, eval=True)
        out = self.render_image(data, gen_cano_mesh=True)                 # lightning_model.py:320
        return {"rgb_pred": out["image"].squeeze(0).permute(2, 0, 1),
                "normal_pred": out["output_normal"].squeeze(0).permute(2, 0, 1),
                "normal_front": out["normal_cano_front"].squeeze(0).permute(2, 0, 1),
                "normal_back": out["normal_cano_back"].squeeze(0).permute(2, 0, 1)}

    def test_epoch_end(self, test_step_outputs, first_index=0, index_stride=1, clear=True):
        """lightning_model.py:351-401 without the Lightning all_gather: writes rgb_ / normal_ / front_ / back_%06d.png into
        <out_dir>/vis (PNG through PIL; the reference's imageio and its vis.mp4 are not part of this build).  A rank of a
        frame-sharded run passes first_index = rank, index_stride = world size and clear = (rank == 0)."""
        import shutil
        import numpy as np
        from PIL import Image
        vis_dir = os.path.join(self.cfg["training"]["out_dir"], "vis")
        if clear and os.path.exists(vis_dir):
            shutil.rmtree(vis_dir)
        os.makedirs(vis_dir, exist_ok=True)
        written = []
        for k, out in enumerate(test_step_outputs):
            idx = first_index + k * index_stride
            for name, key in (("rgb", "rgb_pred"), ("normal", "normal_pred"), ("front", "normal_front"), ("back", "normal_back")):
                img = (out[key].permute(1, 2, 0).detach().cpu().numpy() * 255.0).astype(np.uint8)   # the reference's cast
                path = os.path.join(vis_dir, "%s_%06d.png" % (name, idx))
                Image.fromarray(img).save(path)
                written.append(path)
        return written


class _Method:
    class config:   # noqa: N801  (mirrors ``method_dict[method].config.get_model``)
        get_model = staticmethod(get_render_model)

    class lightning_model:   # noqa: N801
        LightningModel = LightningModel


method_dict = {"metaavatar_render": _Method}


def get_model(cfg, dataset=None, val_size=None, mode="train", low_vram=False, checkpoint_path=None, **kwargs):
    """im2mesh.config.get_model (reference im2mesh/config.py:60-75)."""
    method = method_dict[cfg["method"]]
    model = method.config.get_model(cfg, dataset=dataset, mode=mode, low_vram=low_vram,
                                    checkpoint_path=checkpoint_path, **kwargs)
    return method.lightning_model.LightningModel(model=model, cfg=cfg, val_size=val_size)


# ---------------------------------------------------------------------------------------------
# synthetic weights (no reference checkpoint is redistributable)
# ---------------------------------------------------------------------------------------------
_REF_POSE_FRAME = 10007   # pose at which the synthetic hypernetwork reproduces the fitted SIREN exactly
_HYPER_CACHE = {}


def _synthetic_hyper_state(a, seed=0):
    """COMPLETE state of the SDF hypernetwork (every parameter and buffer of ``sdf_decoder``, reference names), built
    once per process.

    Pose / latent dependence (SURVEY 8a2): a fresh model's residual heads end in an all-zero layer
    (hyperlayers.py:418-423), so that the pose encoder -> FCBlock / LayerNorm -> weight-reshape path would contribute
    exactly nothing and every gradient upstream of the heads would vanish.  Here the heads' last layers and the last
    layer of the FiLM mapping network are seeded and NON-ZERO; ``hypo_params_init`` and the mapping network's last
    bias are shifted so that the emitted network equals the fitted SIREN at ONE reference pose / latent code that no
    test uses (frame %d, a latent code of all 0.5): every other pose emits the fitted weights plus a pose-dependent
    residual (a few 1e-3 of the SDF's range -- millimetres of geometry, three orders above the parity tolerances).
    All remaining hypernetwork parameters (pose encoder, hidden layers of the heads, LayerNorms, mapping network)
    are part of the returned state too, so that the reference's own module, whose initialisers draw random numbers in
    another order, becomes the SAME function.""" % _REF_POSE_FRAME
    key = (id(a), seed)
    if key in _HYPER_CACHE:
        return _HYPER_CACHE[key]
    from . import synthetic
    t = lambda k: torch.from_numpy(np.asarray(a[k], dtype=np.float32))
    rng = torch.random.get_rng_state()
    torch.manual_seed(seed)
    dec = nets.decoder_dict["hyper_bvp"](**_SDF_KW)
    torch.random.set_rng_state(rng)

    def seeded(shape, std, s_):
        g = torch.Generator().manual_seed(s_)
        return torch.randn(shape, generator=g) * std

    with torch.no_grad():
        heads = [l.hyper_linear.hypo_params for l in dec.net.layers[:-1]] + [dec.net.layers[-1].hypo_params]
        for i, h in enumerate(heads):
            std = 6.0e-4 if i == 0 else 9.0e-6   # residual ~ a few 1e-3 of the layer's weight scale between two poses
            h.net[2].weight.copy_(seeded(tuple(h.net[2].weight.shape), std, 7000 + i))
            h.net[2].bias.copy_(seeded(tuple(h.net[2].bias.shape), std, 7100 + i))
        last = dec.net.mapping_network.network[6]
        last.weight.copy_(seeded(tuple(last.weight.shape), 5.0e-5, 7200))
        # reference condition: pose of a frame no test renders, latent code 0.5 * ones
        scene = synthetic.SyntheticScene(0)
        fr = scene.frame(_REF_POSE_FRAME)
        rots = torch.from_numpy(fr["rots_local"].reshape(1, 24, 9).copy())
        rots[0, 0] = torch.eye(3).reshape(9)
        Jn = torch.from_numpy(synthetic.normalize_points_np(scene.joints, scene.coord_min, scene.coord_max,
                                                            scene.center).astype(np.float32))[None]
        cond = dec.pose_encoder(rots, Jn)
        for i, h in enumerate(heads):
            fitted = torch.cat([t("sdf_w%d" % i).reshape(-1), t("sdf_b%d" % i).reshape(-1)]).reshape(1, -1)
            holder = dec.net.layers[i].hyper_linear if i < 6 else dec.net.layers[6]
            holder.hypo_params_init.copy_(fitted - h(cond))
        film = torch.cat([t("film_freq").reshape(-1), t("film_phase").reshape(-1)])
        last.bias.zero_()
        last.bias.copy_(film - dec.net.mapping_network.network(torch.full((1, 128), 0.5))[0])
    sd = {"sdf_decoder." + k: v.detach().clone() for k, v in dec.state_dict().items()}
    _HYPER_CACHE[key] = sd
    return sd


def synthetic_state_dict(cfg, asset_path=None):
    """State-dict entries (reference names) that turn a freshly built model into the synthetic subject: the complete
    SDF hypernetwork (fitted SIREN + seeded pose / latent dependence, see _synthetic_hyper_state), fitted skinning
    MLP, seeded colour MLP / latent codes / beta."""
    path = asset_path or os.path.join(_ASSETS, "synthetic_weights.npz")
    if path not in _HYPER_CACHE:
        _HYPER_CACHE[path] = np.load(path)
    a = _HYPER_CACHE[path]
    t = lambda k: torch.from_numpy(np.asarray(a[k], dtype=np.float32))
    sd = dict(_synthetic_hyper_state(a))
    tag = cfg["model"]["renderer_kwargs"]["mode"]
    for k in a.files:
        if k.startswith("skin."):
            sd["skinning_model.skinning_decoder_fwd." + k[5:]] = t(k)
        if k.startswith("color_%s." % tag):
            sd["color_decoder." + k[len("color_%s." % tag):]] = t(k)
    sd["latent.weight"] = t("latent")
    sd["deviation_decoder.variance"] = t("variance").reshape(())
    return sd


def widen_skinning_(module_or_state, scale=8.0):
    """Test subject with a WIDE-RANGE skinning MLP: the weight-norm gains of its three hidden 128 x 128 layers times
    `scale` (pre-activations and Softplus outputs grow by about that factor per layer).  Works in place on a model (this
    build's or the reference's: same parameter names) or on a state dict; fixture F17 is the reference on such a subject."""
    pre = "skinning_model.skinning_decoder_fwd."
    sd = module_or_state if isinstance(module_or_state, dict) else module_or_state.state_dict()
    with torch.no_grad():
        for k in (1, 2, 3):
            sd[pre + "lin%d.weight_g" % k].mul_(scale)
    return module_or_state


def build_synthetic_model(name="zju377_mono", n_steps=64, near=16, far=16, device="cpu", seed=0, training=None):
    """Builtin config + synthetic weights; deterministic.  The emitted SDF MLP is the fitted SIREN plus a seeded,
    pose- and latent-dependent residual of about a percent per weight (synthetic_state_dict)."""
    cfg = builtin_config(name, n_steps, near, far)
    if training:
        cfg["training"].update(training)
    gen = torch.random.get_rng_state()
    torch.manual_seed(seed)
    model = get_render_model(cfg, mode="test", n_data_points=4)
    torch.random.set_rng_state(gen)
    missing = model.load_state_dict(synthetic_state_dict(cfg), strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    return model.to(device).eval(), cfg
