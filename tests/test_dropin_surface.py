"""The reference's entry-point names resolve to this build and keep their signatures."""
import inspect

import torch


def test_import_paths_and_signatures():
    from im2mesh import config
    from im2mesh.metaavatar_render.models import MetaAvatarRender
    from im2mesh.metaavatar_render.renderer.ray_tracing import BodyRayTracing
    from im2mesh.metaavatar_render.renderer.implicit_differentiable_renderer import IDHRNetwork
    from im2mesh.metaavatar.models import decoder_dict
    assert "metaavatar_render" in config.method_dict
    assert list(inspect.signature(config.get_model).parameters)[:6] == \
        ["cfg", "dataset", "val_size", "mode", "low_vram", "checkpoint_path"]
    assert list(inspect.signature(MetaAvatarRender.forward).parameters) == ["self", "inputs", "gen_cano_mesh", "eval"]
    want = ["self", "sdf_network", "skinning_model", "cam_loc", "ray_directions", "body_bounds_intersections", "loc",
            "sc_factor", "smpl_verts", "smpl_verts_cano", "skinning_weights", "vol_feat", "bone_transforms", "trans",
            "coord_min", "coord_max", "center", "eval_mode"]
    assert list(inspect.signature(BodyRayTracing.forward).parameters)[:len(want)] == want
    assert list(inspect.signature(IDHRNetwork.__init__).parameters) == \
        ["self", "deviation_network", "rendering_network", "skinning_model", "ray_tracer", "cano_view_dirs",
         "train_skinning_net", "render_last_pt", "low_vram"]
    assert set(decoder_dict) >= {"hyper_bvp", "deformer_mlp"}
    # the harness and the test dataset under the reference's names (lightning_model.py:101, data/zju_mocap_odp.py:20-38)
    from im2mesh import data
    from im2mesh.metaavatar_render.lightning_model import LightningModel
    for name in ("compose_inputs", "training_step", "validation_step", "test_step", "test_epoch_end", "configure_optimizers"):
        assert callable(getattr(LightningModel, name)), name
    assert list(inspect.signature(LightningModel.compose_inputs).parameters) == ["self", "data", "eval"]
    ref_kwargs = ["dataset_folder", "subjects", "pose_dir", "mode", "orig_img_size", "img_size", "num_fg_samples",
                  "num_bg_samples", "sampling_rate", "start_frame", "end_frame", "views", "box_margin"]
    assert set(ref_kwargs) <= set(inspect.signature(data.ZJUMOCAPODPDataset.__init__).parameters)


def test_state_dict_names_and_checkpoint_roundtrip(tmp_path):
    """Reference checkpoints are Lightning state dicts with a 'model.' prefix (config.py:291-300)."""
    from arah_release_amd import config
    cfg = config.builtin_config("zju313")
    model = config.get_render_model(cfg, mode="test", n_data_points=3)
    names = set(model.state_dict().keys())
    for k in ["sdf_decoder.net.layers.0.hyper_linear.hypo_params.net.0.net.0.weight",
              "sdf_decoder.net.layers.0.hyper_linear.hypo_params.net.0.net.1.bias",
              "sdf_decoder.net.layers.5.hyper_linear.hypo_params.net.2.weight",
              "sdf_decoder.net.layers.3.hyper_linear.hypo_params_init",
              "sdf_decoder.net.layers.6.hypo_params.net.2.bias", "sdf_decoder.net.layers.6.hypo_params_init",
              "sdf_decoder.net.mapping_network.network.6.weight", "sdf_decoder.pose_encoder.layer_0.weight",
              "sdf_decoder.pose_encoder.layers.23.2.bias", "skinning_model.skinning_decoder_fwd.lin4.weight_g",
              "skinning_model.skinning_decoder_fwd.lin0.weight_v", "color_decoder.lin3.weight_v",
              "color_decoder.lin5.bias", "deviation_decoder.variance", "latent.weight",
              "idhr_network.rendering_network.lin0.weight_g"]:
        assert k in names, k
    assert model.color_decoder.lin0.weight_v.shape == (256, 417)      # ZJUMOCAP-313: idr, 417 inputs
    assert model.color_decoder.lin3.weight_v.shape == (256, 545)
    assert 87.0e6 < sum(p.numel() for p in model.parameters()) < 87.1e6   # ~87.0 M parameters
    ckpt = {"state_dict": {"model." + k: v + 1.0 if v.dtype.is_floating_point else v
                           for k, v in model.state_dict().items()}}
    path = tmp_path / "last.ckpt"
    torch.save(ckpt, path)
    lm = config.get_model(cfg, mode="test", checkpoint_path=str(path))
    a = lm.model.state_dict()["color_decoder.lin2.bias"]
    b = model.state_dict()["color_decoder.lin2.bias"] + 1.0
    assert torch.equal(a, b) and lm.model.latent.weight.shape == (3, 128)
