import os, sys, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__; __graft_entry__.build()
from arah_release_amd import config, synthetic, training
dev = torch.device("cuda", 0)
model, cfg = config.build_synthetic_model("zju313", device=dev)
model.train()
crit = training.build_loss(cfg)
scene = synthetic.SyntheticScene(0)
orig = training.CompositeSamples.apply
stats = []
class Hook:
    @staticmethod
    def apply(len32, offsets, z, n_steps, last, sdf, rgb, inv_beta):
        x = (sdf.detach() * inv_beta.detach()).float()
        P = x.numel()
        stats.append((P, float((x > 110).sum()) / P, float((x > 17.4).sum()) / P, float((x < 0).sum()) / P, float((x < -110).sum()) / P, float(inv_beta)))
        return orig(len32, offsets, z, n_steps, last, sdf, rgb, inv_beta)
training.CompositeSamples = Hook
for k in range(6):
    inp = scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev)
    out = training.training_step(model, crit, inp)
    out["loss"].backward()
for s in stats: print("P=%d  s*ib>110: %.3f  >17.4: %.3f  s<0: %.3f  s*ib<-110: %.3f  ib=%.1f" % s)
