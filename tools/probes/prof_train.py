import os, sys
sys.path.insert(0, os.getcwd())
import torch
from torch.profiler import profile, ProfilerActivity
from arah_release_amd import config, synthetic, training
dev = torch.device("cuda:0")
model, cfg = config.build_synthetic_model("zju313", device=dev)
model.train()
opt = training.configure_optimizers(model, cfg)
crit = training.build_loss(cfg)
scene = synthetic.SyntheticScene(0)
batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev) for k in range(3)]
def step(inp):
    opt.zero_grad(set_to_none=True)
    losses = training.training_step(model, crit, inp)
    losses["loss"].backward()
    opt.step()
step(batches[0]); step(batches[1]); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(batches[2]); torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=40, max_shapes_column_width=70))
