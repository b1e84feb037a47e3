"""The occupancy build's per-cell data of a few frames, read straight out of the buffer (layout of carve_occ in csrc/arah_hip.hip):
how steep is the forward skinning between the fine points of a refined cell, for several ways of choosing the pairs."""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import __graft_entry__
__graft_entry__.build()
from arah_release_amd import config, synthetic

NC, F, MAXCELLS, MAXVOX = 49, 3, 12288, 1 << 22
F3 = F ** 3
MAXFINE = MAXCELLS * F3


def carve(buf):
    off = [0]

    def take(nbytes):
        off[0] = (off[0] + 255) // 256 * 256
        o = off[0]
        off[0] += nbytes
        return o
    lay = {}
    lay["info"] = take(64)
    lay["bits"] = take(MAXVOX // 32 * 4)
    lay["dist"] = take(MAXVOX)
    lay["csdf"] = take(NC ** 3 * 4)
    lay["cpts"] = take(NC ** 3 * 12)
    lay["cell_lip"] = take(MAXCELLS * 4)
    lay["cell_stretch"] = take(MAXCELLS * 4)
    lay["fnorm"] = take(MAXFINE * 12)
    lay["fsdf"] = take(MAXFINE * 4)
    lay["iota"] = take(MAXFINE * 4)
    lay["sel_raw"] = take(MAXFINE * 12)
    lay["sel_bar"] = take(MAXFINE * 12)
    lay["sel_idx"] = take(MAXFINE * 4)
    lay["sel_of"] = take(MAXFINE * 4)
    return lay


dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
scene = synthetic.SyntheticScene(0)
tracer = model.idhr_network.ray_tracer
for fi in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,5").split(",")]:
    inputs = scene.make_inputs(256, 256, frame_idx=fi, device=dev, max_rays=2048)
    with torch.no_grad():
        model(dict(inputs), eval=True)
    torch.cuda.synchronize()
    ws = tracer.workspace(dev)
    info = ws.occupancy_info()
    buf = ws.occ
    lay = carve(buf)
    n_cells, n_sel = info["n_cells"], info["n_selected"]

    def arr(name, dtype, count):
        nbytes = count * torch.tensor([], dtype=dtype).element_size()
        return buf[lay[name]:lay[name] + nbytes].view(dtype).cpu().numpy()
    fsdf = arr("fsdf", torch.float32, n_cells * F3).reshape(n_cells, F3)
    sel_of = arr("sel_of", torch.int32, n_cells * F3).reshape(n_cells, F3)
    bar = arr("sel_bar", torch.float32, n_sel * 3).reshape(n_sel, 3)
    lip = arr("cell_lip", torch.float32, n_cells)
    stretch = arr("cell_stretch", torch.float32, n_cells)
    scale = info["band_m"] / 18.0 / 1e-3 * 0 + 1.0   # (metres per normalised unit is not in the header: spacing from the points)
    fnorm = arr("fnorm", torch.float32, n_cells * F3 * 3).reshape(n_cells, F3, 3)
    idx = np.arange(F3)
    a, b, d = idx // (F * F), (idx // F) % F, idx % F
    grid = np.stack([a, b, d], -1)
    dd = np.linalg.norm(grid[:, None, :] - grid[None, :, :], axis=-1)           # in fine steps
    # metres per fine step: from the kernel's own numbers (stretch of a rigid pair is 1): use the median adjacent posed distance
    P = np.zeros((n_cells, F3, 3), np.float32)
    ok = sel_of >= 0
    P[ok] = bar[sel_of[ok]]
    dist = np.linalg.norm(P[:, :, None, :] - P[:, None, :, :], axis=-1)
    adj = (dd == 1)[None] & ok[:, :, None] & ok[:, None, :]
    step_m = np.median(dist[adj])
    band_n = None
    print("frame", fi, info, "fine step ~%.4f m" % step_m)
    both = ok[:, :, None] & ok[:, None, :]
    for name, pair in (("all pairs", dd > 0), ("adjacent + face/space diagonals (dd < 2)", (dd > 0) & (dd < 2)), ("axis neighbours only", dd == 1)):
        m = both & pair[None]
        r = np.where(m, dist / (np.maximum(dd, 1e-9)[None] * step_m), 0.0).reshape(n_cells, -1).max(1)
        print("  %-42s cells: max %.2f  q50 %.2f q90 %.2f q99 %.2f  | cells > 2: %d  > 3: %d  > 5: %d of %d" % (
            name, r.max(), np.quantile(r, .5), np.quantile(r, .9), np.quantile(r, .99), (r > 2).sum(), (r > 3).sum(), (r > 5).sum(), n_cells))
    print("  kernel's cell_stretch: max %.2f q99 %.2f; cell_lip: max %.2f q50 %.2f q99 %.2f" % (stretch.max(), np.quantile(stretch, .99), np.abs(lip).max(), np.median(np.abs(lip)), np.quantile(np.abs(lip), .99)))
