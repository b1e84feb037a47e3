"""GPU time of one training step by aten op and input shape (torch.profiler, key_averages(group_by_input_shape=True)): which
framework ops the step's element-wise / reduction / GEMM launches come from.   python tools/probes/train_ops.py"""
import os, sys
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def main():
    import __graft_entry__
    __graft_entry__.build()
    from arah_release_amd import config, synthetic, training
    from torch.profiler import profile, ProfilerActivity
    dev = torch.device("cuda", 0)
    model, cfg = config.build_synthetic_model("zju313", device=dev)
    model.train()
    opt = training.configure_optimizers(model, cfg)
    crit = training.build_loss(cfg)
    scene = synthetic.SyntheticScene(0)
    batches = [scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=dev) for k in range(6)]

    def step(inp):
        opt.zero_grad(set_to_none=True)
        training.training_step(model, crit, inp)["loss"].backward()
        opt.step()

    for k in range(4):
        step(batches[k])
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        step(batches[4])
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_device_time_total", row_limit=70, max_name_column_width=48,
                                                            max_shapes_column_width=90))


if __name__ == "__main__":
    main()
