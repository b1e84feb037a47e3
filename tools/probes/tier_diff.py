"""Where do the tiered and the untiered forward differ?  Per-sample arrays of both, run twice each."""
import os, sys
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
import __graft_entry__
__graft_entry__.build()
from arah_release_amd import config, synthetic
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
idhr = model.idhr_network
idhr.adaptive_shading = False
scene = synthetic.SyntheticScene(0)
tracer = idhr.ray_tracer
fi = int(sys.argv[1]) if len(sys.argv) > 1 else 0
inputs = scene.make_inputs(512, 512, frame_idx=fi, device=dev)
n, S = inputs["ray_dirs"].shape[1], 64
runs = {}
for name, tier in [("exact1", False), ("exact2", False), ("tier1", True), ("tier2", True)]:
    idhr.tiering = tier
    with torch.no_grad():
        out = model(dict(inputs), eval=True)
        torch.cuda.synchronize()
        ws = tracer.workspace(dev)
        d = ws.debug_samples(n, S)
        t, p = ws.tier_debug(n, S)
        torch.cuda.synchronize()
    d = {k: v.clone() for k, v in d.items()}
    d["rgb"] = out["rgb_values"][0].clone()
    d["tier"] = t.clone()
    runs[name] = d
def cmp(a, b, label):
    A, B = runs[a], runs[b]
    both = (A["mask"] == 1) & (B["mask"] == 1)
    print(label, "rgb rays differ", int((A["rgb"] != B["rgb"]).any(-1).sum()),
          "| samples valid in both", int(both.sum()), "mask differ (where B evaluated)", int(((A["mask"] != B["mask"]) & ((B["state"] == 1) | (B["state"] == 3) | (b.startswith("exact")))).sum()),
          "| pts differ", int(((A["pts"] != B["pts"]).any(-1) & both).sum()), "T differ", int(((A["T"] != B["T"]).any(-1) & both).sum()),
          "sigma differ", int(((A["shaded"][:, 3] != B["shaded"][:, 3]) & both).sum()),
          "rgb-sample differ (sigma>0)", int(((A["shaded"][:, :3] != B["shaded"][:, :3]).any(-1) & both & (A["shaded"][:, 3] > 0)).sum()),
          "z differ", int((A["z"] != B["z"]).sum()))
cmp("exact1", "exact2", "exact vs exact:")
cmp("tier1", "tier2", "tier vs tier:")
cmp("exact1", "tier1", "exact vs tier:")
A, B = runs["exact1"], runs["tier1"]
both = (A["mask"] == 1) & (B["mask"] == 1)
bad = ((A["pts"] != B["pts"]).any(-1) & both).nonzero()[:, 0]
print("first differing samples:", bad[:10].tolist())
for q in bad[:6].tolist():
    print(q, "ray", q // S, "s", q % S, "state", int(B["state"][q]), "tier", int(B["tier"][q // S]), A["pts"][q].tolist(), B["pts"][q].tolist(), "sig", float(A["shaded"][q, 3]), float(B["shaded"][q, 3]))
rd = (A["rgb"] != B["rgb"]).any(-1).nonzero()[:, 0]
print("rays differing:", rd[:10].tolist(), "tiers", B["tier"][rd[:10]].tolist(), "max abs diff", float((A["rgb"] - B["rgb"]).abs().max()))
for r in rd[:3].tolist():
    sl = slice(r * S, r * S + S)
    print("ray", r, "mask exact", A["mask"][sl].tolist())
    print("      mask tier ", B["mask"][sl].tolist())
    print("      state     ", B["state"][sl].tolist())
    print("      sig>0 exact", (A["shaded"][sl, 3] > 0).int().tolist())
    print("      sig>0 tier ", ((B["shaded"][sl, 3] > 0) & (B["mask"][sl] == 1)).int().tolist())
