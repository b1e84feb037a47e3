"""exit-code probe: render one frame through the model entry, then leave the interpreter"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__
__graft_entry__.build()
from arah_release_amd import config, synthetic
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju377_mono", 64, 16, 16, device=dev)
model.eval()
scene = synthetic.SyntheticScene(0)
with torch.no_grad():
    out = model(scene.make_inputs(128, 128, frame_idx=0, device=dev), eval=True)
torch.cuda.synchronize()
print("rendered", float(out["rgb_values"].sum()))
