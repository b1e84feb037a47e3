"""SMPL-side pieces of the callers of the hot path, as device-agnostic torch (SURVEY 8 f2 / "missing" #3):

  * ``lbs`` / ``batch_rodrigues`` / ``batch_rigid_transform``: linear blend skinning of the SMPL template as the
    reference's ``MetaAvatarRender.forward_smpl`` uses it (human_body_prior/body_model/lbs.py:34-260, in the
    reference tree);
  * ``get_transforms_02v``: T-pose -> "Vitruvian" A-pose bone transforms (metaavatar_render/lightning_model.py:37-99,
    numpy twin data/zju_mocap_odp.py get_02v_bone_transforms);
  * ``angle_axis_to_rotation_matrix`` / ``quaternion_to_rotation_matrix``: kornia 0.5.10 conversions used by
    ``compose_inputs`` (lightning_model.py:477,539; third-party, restated from their documented formulas);
  * ``BodyModel``: the six arrays ``body_models/misc/*.npz`` hold per gender (not redistributable), loadable from those
    files when present, or synthesised from the capsule figure of ``synthetic.py`` for tests and benchmarks.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from .nets import SMPL_PARENTS


def batch_rodrigues(aa):
    """(N,3) axis-angle -> (N,3,3); the 1e-8 inside the norm is the reference's (lbs.py:164-190)."""
    n = aa.shape[0]
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    d = aa / angle
    cos, sin = torch.cos(angle).unsqueeze(1), torch.sin(angle).unsqueeze(1)
    rx, ry, rz = torch.split(d, 1, dim=1)
    z = torch.zeros((n, 1), dtype=aa.dtype, device=aa.device)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(n, 3, 3)
    return torch.eye(3, dtype=aa.dtype, device=aa.device).unsqueeze(0) + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:205-260: (B,J,3,3), (B,J,3) -> posed joints (B,J,3), relative transforms A (B,J,4,4), absolute (B,J,4,4)."""
    B, J = rot_mats.shape[0], joints.shape[1]
    joints = joints.unsqueeze(-1)
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    tm = torch.cat([F.pad(rot_mats.reshape(-1, 3, 3), [0, 0, 0, 1]),
                    F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1)], dim=2).view(B, J, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    transforms = torch.stack(chain, dim=1)
    posed = transforms[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros(B, J, 1, 1, dtype=joints.dtype, device=joints.device)], dim=2)
    init_bone = F.pad(torch.matmul(transforms, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, transforms - init_bone, transforms


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights):
    """human_body_prior lbs (clothed_v_template=None): -> verts (B,V,3), posed joints, rest joints J, A, abs_A, v_posed."""
    B = betas.shape[0]
    v_shaped = v_template + torch.einsum("bl,mkl->bmk", betas, shapedirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, J_regressor)
    rot = batch_rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
    if posedirs is not None:
        feat = (rot[:, 1:] - torch.eye(3, dtype=rot.dtype, device=rot.device)).reshape(B, -1)
        v_posed = torch.matmul(feat, posedirs).view(B, -1, 3) + v_shaped
    else:
        v_posed = v_shaped
    J_t, A, abs_A = batch_rigid_transform(rot, J, parents)
    W = lbs_weights.unsqueeze(0).expand(B, -1, -1)
    T = torch.matmul(W, A.reshape(B, -1, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=v_posed.dtype, device=v_posed.device)], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_t, J, A, abs_A, v_posed


def _rotz(deg, device):
    a = np.deg2rad(deg)
    return torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32,
                        device=device)


def get_transforms_02v(Jtr):
    """(24,3) rest joints -> (24,4,4): legs spread by +-45 degrees about z, chain by chain (lightning_model.py:37-99)."""
    dev = Jtr.device
    out = torch.eye(4, dtype=torch.float32, device=dev).reshape(1, 4, 4).repeat(24, 1, 1)
    for chain, rot in (([1, 4, 7, 10], _rotz(45.0, dev)), ([2, 5, 8, 11], _rotz(-45.0, dev))):
        ts = []
        for i, j in enumerate(chain):
            t = Jtr[j]
            if i > 0:
                t = torch.matmul(rot, t - Jtr[chain[i - 1]]) + ts[i - 1]
            ts.append(t)
        t = torch.stack(ts, dim=0) - torch.matmul(Jtr[chain], rot.transpose(0, 1))
        R = F.pad(rot.unsqueeze(0).repeat(4, 1, 1), (0, 0, 0, 1))
        out[chain] = torch.cat([R, F.pad(t, (0, 1), value=1.0).unsqueeze(-1)], dim=-1)
    return out


def angle_axis_to_rotation_matrix(aa):
    """kornia.geometry.conversions.angle_axis_to_rotation_matrix (0.5.10): Rodrigues with a first-order Taylor branch
    below theta^2 = 1e-6.  (N,3) -> (N,3,3)."""
    theta2 = (aa * aa).sum(dim=1)
    theta = torch.sqrt(theta2)
    w = aa / (theta.unsqueeze(1) + 1e-6)
    wx, wy, wz = w[:, 0], w[:, 1], w[:, 2]
    c, s = torch.cos(theta), torch.sin(theta)
    k1 = 1.0 - c
    normal = torch.stack([c + wx * wx * k1, wx * wy * k1 - wz * s, wy * s + wx * wz * k1,
                          wz * s + wx * wy * k1, c + wy * wy * k1, -wx * s + wy * wz * k1,
                          -wy * s + wx * wz * k1, wx * s + wy * wz * k1, c + wz * wz * k1], dim=1).view(-1, 3, 3)
    rx, ry, rz = aa[:, 0], aa[:, 1], aa[:, 2]
    one = torch.ones_like(rx)
    taylor = torch.stack([one, -rz, ry, rz, one, -rx, -ry, rx, one], dim=1).view(-1, 3, 3)
    return torch.where((theta2 > 1e-6).view(-1, 1, 1), normal, taylor)


def quaternion_to_rotation_matrix_xyzw(q):
    """kornia quaternion_to_rotation_matrix with QuaternionCoeffOrder.XYZW (normalised first). (N,4) -> (N,3,3)."""
    q = F.normalize(q, p=2, dim=-1, eps=1e-12)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    return torch.stack([1 - (ty * y + tz * z), tx * y - tz * w, tx * z + ty * w,
                        tx * y + tz * w, 1 - (tx * x + tz * z), ty * z - tx * w,
                        tx * z - ty * w, ty * z + tx * w, 1 - (tx * x + ty * y)], dim=-1).view(-1, 3, 3)


class BodyModel:
    """v_template (V,3), lbs_weights (V,24), posedirs (207, V*3), shapedirs (V,3,10), J_regressor (24,V),
    kintree_table (2,24): the arrays models/__init__.py:93-110 reads from body_models/misc/ for one gender."""

    FILES = {"v_template": "v_templates.npz", "lbs_weights": "skinning_weights_all.npz", "posedirs": "posedirs_all.npz",
             "shapedirs": "shapedirs_all.npz", "J_regressor": "J_regressors.npz"}

    def __init__(self, v_template, lbs_weights, posedirs, shapedirs, J_regressor, kintree_table):
        self.v_template, self.lbs_weights, self.posedirs = v_template, lbs_weights, posedirs
        self.shapedirs, self.J_regressor, self.kintree_table = shapedirs, J_regressor, kintree_table

    @classmethod
    def from_files(cls, gender, root="body_models/misc"):
        missing = [f for f in list(cls.FILES.values()) + ["kintree_table.npy"] if not os.path.exists(os.path.join(root, f))]
        if missing:
            raise FileNotFoundError("SMPL body-model files missing under %s: %s (they are licence-gated; pass a BodyModel "
                                    "through model kwargs 'body_model' to use another one)" % (root, ", ".join(missing)))
        a = {k: np.load(os.path.join(root, f))[gender] for k, f in cls.FILES.items()}
        pd = a["posedirs"]
        a["posedirs"] = pd.reshape([pd.shape[0] * 3, -1]).T
        return cls(kintree_table=np.load(os.path.join(root, "kintree_table.npy")), **a)

    @classmethod
    def synthetic(cls, scene, seed=0):
        """A body model with SMPL's shapes on the capsule figure: template = the figure's canonical vertices, joints
        regressed from the 32 nearest vertices of each joint (exactly: the weights are corrected so that
        J_regressor @ v_template reproduces the joints), small seeded pose / shape blend shapes."""
        rng = np.random.RandomState(seed)
        V = scene.verts_cano.astype(np.float64)
        J = scene.joints.astype(np.float64)
        reg = np.zeros((24, V.shape[0]))
        for j in range(24):
            d = np.linalg.norm(V - J[j], axis=1)
            nn = np.argsort(d)[:32]
            w = np.exp(-(d[nn] / (d[nn].mean() + 1e-9)) ** 2)
            w /= w.sum()
            # affine correction: minimal-norm change of w that keeps sum(w) = 1 and hits the joint exactly
            Acon = np.concatenate([V[nn].T, np.ones((1, nn.size))], axis=0)              # (4, 32)
            w += np.linalg.lstsq(Acon, np.concatenate([J[j], [1.0]]) - Acon @ w, rcond=None)[0]
            reg[j, nn] = w
        posedirs = rng.randn(207, V.shape[0] * 3) * 1e-3
        shapedirs = rng.randn(V.shape[0], 3, 10) * 5e-3
        kin = np.stack([np.array(SMPL_PARENTS, np.int64), np.arange(24)], axis=0)
        kin[0, 0] = 4294967295 if False else -1
        return cls(scene.verts_cano.astype(np.float32), scene.weights.astype(np.float32), posedirs.astype(np.float32),
                   shapedirs.astype(np.float32), reg.astype(np.float32), kin.astype(np.int32))
