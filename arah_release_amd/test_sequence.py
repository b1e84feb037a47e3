"""Render a pose sequence -- the reference's ``test.py`` (test.py:1-90) on this build.

    python -m arah_release_amd.test_sequence CONFIG.yaml --pose-dir gBR_sBM_cAll_d04_mBR1_ch06_view1 --test-views 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m arah_release_amd.test_sequence CONFIG.yaml ...

Same arguments and the same overrides of the configuration as test.py:39-52; the model comes from
``<out_dir>/checkpoints/last.ckpt``; frames are composed on the GPU (data.SequenceDataset), rendered with
``LightningModel.test_step`` -- image plus the three normal maps of the canonical mesh -- and written by
``test_epoch_end`` as ``<out_dir>/vis/{rgb,normal,front,back}_%06d.png``.  With N processes frame i goes to rank i mod N
(no collective on the data path; the reference's Lightning DDP gathers the images on rank 0 instead)."""
import argparse
import os

import torch


def build_parser():
    p = argparse.ArgumentParser(description="Test function that renders images without quantitative evaluation.")
    p.add_argument("config", type=str, help="Path to config file.")
    p.add_argument("--pose-dir", type=str, default="gBR_sBM_cAll_d04_mBR1_ch06_view1",
                   help="Which out-of-distribution pose sequence to render.")
    p.add_argument("--test-views", type=str, default="1", help="Which views to render.")
    p.add_argument("--subsampling-rate", type=int, default=1, help="Sampling rate for poses.")
    p.add_argument("--start-frame", type=int, default=0, help="Frame index to start rendering.")
    p.add_argument("--end-frame", type=int, default=0, help="Frame index to stop rendering.")
    p.add_argument("--low-vram", action="store_true", help="Accepted for compatibility; the workspace is 1 GB per frame.")
    p.add_argument("--multi-gpu", action="store_true", help="Accepted for compatibility (test.py:30): frames are sharded over "
                                                            "the ranks of torch.distributed.run whenever WORLD_SIZE > 1.")
    p.add_argument("--num-workers", type=int, default=4, help="Accepted for compatibility: items are composed on the GPU.")
    p.add_argument("--default-config", type=str, default="configs/default.yaml")
    p.add_argument("--body-models", type=str, default="body_models/misc", help="Directory of the SMPL model files.")
    return p


def apply_overrides(cfg, args):
    """test.py:46-52."""
    cfg["data"]["test_views"] = args.test_views.split(",")
    cfg["data"]["dataset"] = "zju_mocap_odp"
    cfg["data"]["path"] = "data/odp"
    cfg["data"]["test_subsampling_rate"] = args.subsampling_rate
    cfg["data"]["test_start_frame"] = args.start_frame
    cfg["data"]["test_end_frame"] = args.end_frame
    cfg["data"]["pose_dir"] = args.pose_dir
    return cfg


def render(lm, dataset, device, rank=0, world=1):
    """Frames rank, rank + world, ... of the dataset through test_step; returns the written files.  A single process
    replaces an existing vis directory like the reference; with several, rank 0 has done that before anyone renders."""
    from . import renderer
    lm = lm.to(device).eval()
    mine = list(range(rank, len(dataset), world))
    outs = []
    for c in range(0, len(mine), 20):   # twenty frames resident at a time, renderer.frames_in_flight of them in flight
        items = [dataset.item(i, device) for i in mine[c:c + 20]]
        outs += renderer.map_in_flight(lm.test_step, items, owner=lm.model)
    # the triangle counts of the last frames' meshes are still on their way to the host: drain them, so that a level set that
    # outgrew the device buffer in the sequence's LAST frames is reported too (meshing._mc_poll only runs from a later call)
    from . import meshing
    _, truncated = meshing.mesh_counts(device, wait=True)
    if truncated:
        print("rank %d: %d canonical meshes of this sequence were truncated (see the warnings above)" % (rank, truncated))
    return lm.test_epoch_end(outs, first_index=rank, index_stride=world, clear=(world == 1))


def main(argv=None, body=None):
    from . import config, data, smpl
    args = build_parser().parse_args(argv)
    cfg = apply_overrides(config.load_config(args.config, args.default_config), args)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("test_sequence needs a GPU (the renderer has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    checkpoint_path = os.path.join(cfg["training"]["out_dir"], "checkpoints/last.ckpt")
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError("No checkpoint is found!")          # test.py:58-59
    body = body if body is not None else smpl.BodyModel.from_files("neutral", args.body_models)
    dataset = data.get_dataset("test", cfg, body)
    cfg["model"]["train_smpl"] = False                               # test-time construction takes none of it (config.py:166)
    lm = config.get_model(cfg, val_size=len(dataset), mode="test", low_vram=args.low_vram, checkpoint_path=checkpoint_path)
    if world > 1:   # only to keep rank 0's clearing of the output directory ahead of the other ranks' writes
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        if rank == 0:
            import shutil
            shutil.rmtree(os.path.join(cfg["training"]["out_dir"], "vis"), ignore_errors=True)
            os.makedirs(os.path.join(cfg["training"]["out_dir"], "vis"))
        dist.barrier()
    files = render(lm, dataset, device, rank, world)
    print("rank %d: %d frames -> %s" % (rank, len(files) // 4, os.path.join(cfg["training"]["out_dir"], "vis")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
