"""The certificate of the tiered forward (csrc/tier.hpp), restated on the CPU with the oracle's own functions: every depth
sample whose density is > 0 in the oracle's render lies within the dilation radius of the POSED image of the canonical fat
body's lattice points.  This is the METHOD's property (the kernels are held to the exact path bit for bit in
tests/test_tiered.py on the GPU); it needs no GPU and no library."""
import numpy as np
import torch
from scipy.spatial import cKDTree

from conftest import get_model
from oracle import arah_oracle as O

BAND = 18.0          # kTierBand
LIP_SDF = 1.5        # kOccLipMin (the kernels take max(1.5, 1.25 x the cell's own slope): at least this)
LIP_POSE = 2.0       # kOccLipPose


def test_density_lives_inside_the_posed_fat_body(scene):
    model, cfg = get_model("zju377_mono")
    inputs = scene.make_inputs(128, 128, frame_idx=5, max_rays=400)
    out = O.render_inputs(model, inputs, cfg["model"]["cano_view_dirs"], 64, 16, 16, return_intermediates=True)
    fr = out["frame"]
    beta = min(max(abs(fr.beta), 1e-6), 1e6)
    scale = float(fr.sdf_scale)
    # canonical lattice at the kernels' fine spacing (a third of 3 / 48 normalised units) over [-1.5, 1.5]^3
    h = 3.0 / 48 / 3
    ax = torch.arange(-1.5 + h / 2, 1.5, h)
    half_m = h * np.sqrt(3) / 2 * scale
    fat = []
    with torch.no_grad():
        for x in ax:
            pts = torch.stack(torch.meshgrid(x[None], ax, ax, indexing="ij"), -1).reshape(-1, 3)
            sdf = O.sdf_forward(fr, pts, count=False)[0] * scale
            fat.append(pts[sdf <= BAND * beta + LIP_SDF * half_m])
        fat = torch.cat(fat)
        posed = (O.lbs_forward(fr, O.unnormalize_points(fr, fat), count=False)[0] + fr.trans).numpy()
        # the oracle's samples and their densities
        o = inputs["cam_loc"].reshape(1, 3).float()
        d = inputs["ray_dirs"][0].float()
        z, pts_c, mask = out["sampler_dists"], out["sampler_pts"], out["sampler_converge_mask"]
        x = (o[:, None, :] + z[..., None] * d[:, None, :])[mask].numpy()
        sdf = O.sdf_forward(fr, pts_c[mask], count=False)[0] * scale
        inv_beta = 1.0 / beta
        sigma = torch.relu(inv_beta * (0.5 + 0.5 * torch.sign(-sdf) * (1 - torch.exp(-sdf.abs() * inv_beta))))
    assert len(posed) > 5000 and int((sigma > 0).sum()) > 100
    dist, _ = cKDTree(posed).query(x[(sigma > 0).numpy()], k=1)
    r = LIP_POSE * half_m + 1e-4
    assert dist.max() <= r, "a sample with density > 0 lies %.4f m from the posed fat body's lattice (radius %.4f)" % (dist.max(), r)
    # ... with room to spare: the measured reach is about half the radius
    assert dist.max() <= 0.75 * r
    # and the certificate is not vacuous: most valid samples are outside
    far, _ = cKDTree(posed).query(x, k=1)
    assert (far > r).mean() > 0.5
