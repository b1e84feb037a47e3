"""Import recipe for the *reference* (taconite/arah-release) in THIS container only.

The reference is pure Python/PyTorch but imports a dozen packages that are not
installed here (pytorch3d, pytorch_lightning, cv2, ...).  None of them is needed
by the hot path except ``pytorch3d.ops.knn_points`` (exact 1-NN, only ``.idx``
is consumed: ray_tracing.py:386,407), which is backed here by scipy's cKDTree.

This module is used only by ``make_golden.py`` to generate the committed
fixtures.  It never travels to the GPU box in a meaningful way: it needs
``/root/reference`` and refuses to work without it.
"""
import os
import sys
import types
import collections

REF_ROOT = os.environ.get("ARAH_REFERENCE_ROOT", "/root/reference")

_STUBS = [
    "cv2", "imageio", "lpips", "wandb", "plyfile", "trimesh", "igl", "PIL", "PIL.Image",
    "h5py", "skimage", "skimage.measure", "skimage.metrics",
    "kornia", "kornia.geometry", "kornia.geometry.conversions",
    "torchvision", "torchvision.utils", "torchvision.transforms",
    "torchvision.transforms.functional", "torchvision.datasets", "torchvision.datasets.utils",
    "pytorch3d", "pytorch3d.ops", "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.utils",
    "pytorch_lightning", "pytorch_lightning.callbacks", "pytorch_lightning.loggers",
    "im2mesh.utils.libmesh.triangle_hash",
]


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        return _Dummy()


def _make_stub(name):
    mod = types.ModuleType(name)
    mod.__path__ = []  # behave like a package so sub-imports resolve

    def _getattr(attr, _name=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return _Dummy

    mod.__getattr__ = _getattr
    return mod


_KNN = collections.namedtuple("_KNN", ["dists", "idx", "knn"])


def _knn_points(p1, p2, K=1, **kwargs):
    """Exact K-NN (squared L2) of p1 (B,P,3) in p2 (B,V,3) via cKDTree."""
    import numpy as np
    import torch
    from scipy.spatial import cKDTree
    assert p1.shape[0] == 1 and p2.shape[0] == 1
    tree = cKDTree(p2[0].detach().cpu().double().numpy())
    d, i = tree.query(p1[0].detach().cpu().double().numpy(), k=K, workers=-1)
    if K == 1:
        d = d[:, None]
        i = i[:, None]
    idx = torch.from_numpy(np.ascontiguousarray(i)).long()[None].to(p1.device)
    dists = torch.from_numpy(np.ascontiguousarray(d ** 2)).float()[None].to(p1.device)
    return _KNN(dists=dists, idx=idx, knn=None)


class _BruteTriangleHash:
    """Stand-in for the reference's Cython triangle hash (im2mesh/utils/libmesh/triangle_hash.pyx, not built here).  The
    hash only PRE-SELECTS candidate (point, triangle) pairs -- every triangle registered in the grid cell of the point;
    each candidate then goes through inside_mesh.py's own strict 2-D containment test -- so any superset of the pairs
    that pass that test gives the same answer.  Here: every triangle whose 2-D bounding box contains the point."""

    def __init__(self, triangles, resolution):
        import numpy as np
        t = np.asarray(triangles, np.float64)
        self.lo, self.hi = t.min(1), t.max(1)

    def query(self, points):
        import numpy as np
        pts = np.asarray(points, np.float64)
        pi, ti = [], []
        for c0 in range(0, len(pts), 512):
            p = pts[c0:c0 + 512]
            m = ((p[:, None, :] >= self.lo[None]) & (p[:, None, :] <= self.hi[None])).all(-1)
            a, b = np.nonzero(m)
            pi.append(a + c0)
            ti.append(b)
        return np.concatenate(pi), np.concatenate(ti)


def install():
    """Make ``import im2mesh`` resolve to the reference tree. Returns the im2mesh package."""
    if not os.path.isdir(os.path.join(REF_ROOT, "im2mesh")):
        raise RuntimeError("reference tree not found at %s (golden vectors can only be "
                           "generated in the build container)" % REF_ROOT)
    sys.dont_write_bytecode = True  # /root/reference is read-only
    import torch
    import torch.nn as nn
    for name in _STUBS:
        if name not in sys.modules:
            sys.modules[name] = _make_stub(name)
    for name in _STUBS:  # make ``import a.b.c as x`` resolve through attributes too
        if "." in name and not name.startswith("im2mesh"):
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    sys.modules["pytorch3d.ops"].knn_points = _knn_points
    sys.modules["im2mesh.utils.libmesh.triangle_hash"].TriangleHash = _BruteTriangleHash
    # kornia 0.5.10 conversions used by LightningModel.compose_inputs (lightning_model.py:477,539): third-party, restated
    # from kornia's documented formulas (the same restatement the build uses: these two functions pin nothing)
    import enum
    from arah_release_amd import smpl as _smpl

    class QuaternionCoeffOrder(enum.Enum):
        XYZW = "xyzw"
        WXYZ = "wxyz"

    conv = sys.modules["kornia.geometry.conversions"]
    conv.QuaternionCoeffOrder = QuaternionCoeffOrder
    conv.angle_axis_to_rotation_matrix = _smpl.angle_axis_to_rotation_matrix

    def _q2r(q, order=QuaternionCoeffOrder.XYZW):
        assert order == QuaternionCoeffOrder.XYZW
        return _smpl.quaternion_to_rotation_matrix_xyzw(q)

    conv.quaternion_to_rotation_matrix = _q2r
    pl = sys.modules["pytorch_lightning"]
    pl.LightningModule = nn.Module
    # the reference tree must win over this repo's own ``im2mesh`` drop-in package
    for k in [k for k in sys.modules if k == "im2mesh" or k.startswith("im2mesh.")]:
        if k != "im2mesh.utils.libmesh.triangle_hash":
            del sys.modules[k]
    sys.path.insert(0, REF_ROOT)
    # import order matters (circular import hyperlayers <-> siren_modules), see SURVEY 8c
    import im2mesh.metaavatar.models  # noqa: F401
    import im2mesh
    assert os.path.realpath(im2mesh.__file__).startswith(os.path.realpath(REF_ROOT))
    return im2mesh
