"""Host-side (PyTorch) parameter containers of the ARAH hot path.

These modules own the learnable parameters under the *same state-dict names* as
the reference so that its Lightning checkpoints load unchanged
(reference: im2mesh/metaavatar_render/config.py:291-300 strips the ``model.``
prefix and calls ``load_state_dict(strict=False)``):

    sdf_decoder.net.layers.{0..5}.hyper_linear.hypo_params.net.{0,1}.net.{0,1}.{weight,bias}
    sdf_decoder.net.layers.{0..5}.hyper_linear.hypo_params.net.2.{weight,bias}
    sdf_decoder.net.layers.{0..5}.hyper_linear.hypo_params_init
    sdf_decoder.net.layers.6.hypo_params...            (last, sine-free layer)
    sdf_decoder.net.mapping_network.network.{0,2,4,6}.{weight,bias}
    sdf_decoder.pose_encoder.layer_0 / .layers.{0..23}.{0,2}
    skinning_model.skinning_decoder_fwd.lin{0..4}.{weight_g,weight_v,bias}
    color_decoder.lin{0..5}.{weight_g,weight_v,bias}
    deviation_decoder.variance,  latent.weight

What is evaluated per *sample* (SDF MLP, skinning MLP, colour MLP) runs in the
HIP kernels (csrc/); the modules here are evaluated once per frame (the
hypernetwork that emits the SDF MLP, reference siren_modules.py:280-316 and
hyperlayers.py:270-285,497-510) or only export folded weight tensors.
The torch ``forward`` of the per-sample networks is kept because the training
path differentiates through them with autograd.
"""
import os

import torch
import torch.nn as nn

# SMPL kinematic tree (parent of each of the 24 joints); data, same table as
# reference siren_modules.py:204-205.
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


# --------------------------------------------------------------------------------------
# emitted (per-frame) SDF network
# --------------------------------------------------------------------------------------
class Sine(nn.Module):
    """sin(30 x) (reference siren_modules.py:31-37)."""

    def forward(self, x):
        return torch.sin(30.0 * x)


class EmittedLinear(nn.Module):
    """x @ W^T + b with W (1,out,in), b (1,1,out) emitted by the hypernetwork
    (reference hyperlayers.py:368-388)."""

    def __init__(self, weights, biases):
        super().__init__()
        self.weights = weights
        self.biases = biases

    def forward(self, x):
        return torch.matmul(x, self.weights.transpose(-1, -2)) + self.biases


class EmittedFiLMLinear(EmittedLinear):
    """freq * (x @ W^T + b) + phase (reference hyperlayers.py:391-415)."""

    def __init__(self, weights, biases, freq, phase_shift):
        super().__init__(weights, biases)
        self.freq = freq
        self.phase_shift = phase_shift

    def forward(self, x):
        return self.freq * super().forward(x) + self.phase_shift


# --------------------------------------------------------------------------------------
# hypernetwork
# --------------------------------------------------------------------------------------
class _NormedLinear(nn.Module):
    """Linear -> LayerNorm -> ReLU, stored as ``net.{0,1,2}`` (pytorch_prototyping FCLayer)."""

    def __init__(self, n_in, n_out):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(n_in, n_out), nn.LayerNorm([n_out]), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.net(x)


class _HyperHead(nn.Module):
    """cond(144) -> 256 -> 256 -> n_out, stored as ``net.{0,1,2}``
    (pytorch_prototyping FCBlock with one hidden layer and a linear output)."""

    def __init__(self, n_in, n_hidden, n_out):
        super().__init__()
        self.net = nn.Sequential(_NormedLinear(n_in, n_hidden), _NormedLinear(n_hidden, n_hidden),
                                 nn.Linear(n_hidden, n_out))
        for m in self.net.modules():
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=0.0, nonlinearity="relu", mode="fan_in")
        # residual head starts at zero: emitted params == hypo_params_init (hyperlayers.py:418-423)
        nn.init.zeros_(self.net[2].weight)
        nn.init.zeros_(self.net[2].bias)

    def forward(self, x):
        return self.net(x)


class _WideHead(torch.autograd.Function):
    """weight @ hidden + bias + init for ONE hidden vector and a wide output layer (256 -> in*out + out, 67 MB of weights), with
    gradients: both directions are HBM streams of the weight matrix -- arah_gemv_rows forward, arah_colsum (rows scaled by the
    upstream gradient) for the input gradient, where the GEMM library's batch-1 products run at 0.9 and 0.5 TB/s."""

    @staticmethod
    def forward(ctx, hidden, weight, bias, init):
        from . import hip
        ctx.save_for_backward(hidden, weight)
        return hip.gemv_rows(weight, hidden, bias, init).reshape(*hidden.shape[:-1], -1)

    @staticmethod
    def backward(ctx, g):
        from . import hip
        hidden, weight = ctx.saved_tensors
        g1 = g.reshape(-1).contiguous()
        gh = hip.colsum(weight.detach(), g1).reshape(hidden.shape) if ctx.needs_input_grad[0] else None
        gw = torch.outer(g1, hidden.detach().reshape(-1)) if ctx.needs_input_grad[1] else None
        gb = g1.reshape(-1) if ctx.needs_input_grad[2] else None
        return gh, gw, gb, None


class HyperLinear(nn.Module):
    """Emits one linear layer (W, b) of the SDF MLP from the pose condition."""

    def __init__(self, in_ch, out_ch, hyper_in_ch=144, hyper_hidden_ch=256):
        super().__init__()
        self.in_ch, self.out_ch = in_ch, out_ch
        self.register_buffer("hypo_params_init", torch.zeros(1, in_ch * out_ch + out_ch))
        self.hypo_params = _HyperHead(hyper_in_ch, hyper_hidden_ch, in_ch * out_ch + out_ch)

    def emit(self, cond, hidden=None):
        """hidden: the output of the head's two normed layers, when the caller has run them for all heads at once."""
        head = self.hypo_params.net
        if (hidden is not None and not torch.is_grad_enabled() and hidden.is_cuda and hidden.numel() == hidden.shape[-1]
                and head[2].out_features >= 4096 and os.environ.get("ARAH_HYPER_GEMV", "1") != "0"):
            from . import hip   # inference, one condition vector: the wide output layer as an HBM stream (see below)
            p = hip.gemv_rows(head[2].weight, hidden, head[2].bias, self.hypo_params_init).reshape(*hidden.shape[:-1], -1)
            nw = self.in_ch * self.out_ch
            w, b = p.split([nw, self.out_ch], dim=-1)
            return w.reshape(*p.shape[:-1], self.out_ch, self.in_ch), b.reshape(*p.shape[:-1], 1, self.out_ch)
        if hidden is not None:
            if (hidden.is_cuda and hidden.numel() == hidden.shape[-1] and head[2].out_features >= 4096
                    and hidden.dtype == torch.float32 and os.environ.get("ARAH_HYPER_GEMV", "1") != "0"):
                p = _WideHead.apply(hidden, head[2].weight, head[2].bias, self.hypo_params_init)   # training, one condition vector
            else:
                p = head[2](hidden) + self.hypo_params_init
            nw = self.in_ch * self.out_ch
            w, b = p.split([nw, self.out_ch], dim=-1)
            return w.reshape(*p.shape[:-1], self.out_ch, self.in_ch), b.reshape(*p.shape[:-1], 1, self.out_ch)
        if (not torch.is_grad_enabled() and cond.is_cuda and cond.numel() == cond.shape[-1] and head[2].out_features >= 4096
                and os.environ.get("ARAH_HYPER_GEMV", "1") != "0"):
            # inference, one condition vector: the 256 -> in*out + out layer is a stream of its 67 MB weight matrix --
            # the HBM-bound row kernel of the C ABI instead of a batch-1 GEMM (same sums up to their order)
            from . import hip
            h = head[1](head[0](cond))
            p = hip.gemv_rows(head[2].weight, h, head[2].bias, self.hypo_params_init).reshape(*cond.shape[:-1], -1)
        else:
            p = self.hypo_params(cond) + self.hypo_params_init
        nw = self.in_ch * self.out_ch
        w, b = p.split([nw, self.out_ch], dim=-1)       # one backward node (a cat) instead of two zero-filled slices
        return w.reshape(*p.shape[:-1], self.out_ch, self.in_ch), b.reshape(*p.shape[:-1], 1, self.out_ch)

    def forward(self, cond, hidden=None):
        return EmittedLinear(*self.emit(cond, hidden))


class HyperLinearFiLM(HyperLinear):
    def forward(self, cond, freq, phase_shift, hidden=None):
        w, b = self.emit(cond, hidden)
        return EmittedFiLMLinear(w, b, freq, phase_shift)


class HyperLayerFiLM(nn.Module):
    def __init__(self, in_ch, out_ch, **kw):
        super().__init__()
        self.hyper_linear = HyperLinearFiLM(in_ch, out_ch, **kw)

    def forward(self, cond, freq, phase_shift, hidden=None):
        return nn.Sequential(self.hyper_linear(cond, freq, phase_shift, hidden), Sine())


class MappingNetwork(nn.Module):
    """latent(128) -> FiLM frequencies / phase shifts (reference hyperlayers.py:107-139)."""

    def __init__(self, z_dim, hidden, n_out):
        super().__init__()
        self.network = nn.Sequential(nn.Linear(z_dim, hidden), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(hidden, hidden), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(hidden, hidden), nn.LeakyReLU(0.2, inplace=True),
                                     nn.Linear(hidden, n_out))
        for m in self.network:
            if isinstance(m, nn.Linear):
                nn.init.kaiming_normal_(m.weight, a=0.2, mode="fan_in", nonlinearity="leaky_relu")
        with torch.no_grad():  # identity FiLM at init: freq = 1, phase = 0
            self.network[-1].weight.zero_()
            self.network[-1].bias[:n_out // 2] = 1.0
            self.network[-1].bias[n_out // 2:] = 0.0

    def forward(self, z):
        out = self.network(z)
        return out.split(out.shape[-1] // 2, dim=-1)


class HyperFCFiLM(nn.Module):
    """Hypernetwork emitting in_ch -> hidden x (num_hidden_layers+1) -> out_ch FiLM-SIREN."""

    def __init__(self, hyper_in_ch, hidden_ch, num_hidden_layers, in_ch, out_ch):
        super().__init__()
        kw = dict(hyper_in_ch=hyper_in_ch, hyper_hidden_ch=256)
        self.hidden_ch = hidden_ch
        layers = [HyperLayerFiLM(in_ch, hidden_ch, **kw)]
        layers += [HyperLayerFiLM(hidden_ch, hidden_ch, **kw) for _ in range(num_hidden_layers)]
        layers += [HyperLinear(hidden_ch, out_ch, **kw)]
        self.layers = nn.ModuleList(layers)
        self.batched_heads = True   # False: head by head also while training (tests compare the two)
        self.mapping_network = MappingNetwork(128, 256, (len(layers) - 1) * hidden_ch * 2)

    def forward(self, cond, latent_code):
        freqs, phases = self.mapping_network(latent_code)
        mods = []
        fs, ps = freqs.split(self.hidden_ch, dim=-1), phases.split(self.hidden_ch, dim=-1)
        hidden = (self._heads_hidden(cond) if (self.batched_heads and (torch.is_grad_enabled() or cond.is_cuda))
                  else [None] * len(self.layers))
        for i, layer in enumerate(self.layers[:-1]):
            mods.append(layer(cond, fs[i], ps[i], hidden[i]))
        mods.append(self.layers[-1](cond, hidden[-1]))
        return nn.Sequential(*mods)

    def _heads_hidden(self, cond):
        """The two normed 256-wide layers of all seven hypernetwork heads as two batched products (training: 42 launches and
        ~90 in backward become 18 and ~40; the per-head parameters are stacked on the fly, their gradients come back as views
        of the stacked gradient).  Inference keeps the head-by-head modules (the reference's operation order)."""
        import torch.nn.functional as F
        heads = [l.hyper_linear for l in self.layers[:-1]] + [self.layers[-1]]
        seq = [h.hypo_params.net for h in heads]
        x = cond.reshape(1, -1, cond.shape[-1]).expand(len(seq), -1, -1)                       # (7, B, 144)
        for j in (0, 1):
            W = torch.stack([n[j].net[0].weight for n in seq])                                  # (7, 256, in)
            b = torch.stack([n[j].net[0].bias for n in seq]).unsqueeze(1)
            gw = torch.stack([n[j].net[1].weight for n in seq]).unsqueeze(1)
            gb = torch.stack([n[j].net[1].bias for n in seq]).unsqueeze(1)
            eps = seq[0][j].net[1].eps
            x = torch.relu(F.layer_norm(torch.baddbmm(b, x, W.transpose(1, 2)), (W.shape[1],), eps=eps) * gw + gb)
        return [h.reshape(*cond.shape[:-1], -1) for h in x.unbind(0)]


class _PoseTree(torch.autograd.Function):
    """The tree of per-joint MLPs of HierarchicalPoseEncoder for ONE frame as one launch each way (hip.pose_tree_forward /
    _backward).  apply(own (J,13) [no gradient], glob (6,), W1 (J,19,19), b1 (J,19), W2 (J,6,19), b2 (J,6), parents) -> (J,6)."""

    @staticmethod
    def forward(ctx, own, glob, W1, b1, W2, b2, parents):
        from . import hip
        feats, hidden = hip.pose_tree_forward(own, glob, W1, b1, W2, b2, parents)
        ctx.save_for_backward(own, glob, W1, W2, feats, hidden)
        ctx.parents = parents
        return feats

    @staticmethod
    def backward(ctx, g):
        from . import hip
        own, glob, W1, W2, feats, hidden = ctx.saved_tensors
        gW1, gb1, gW2, gb2, gg = hip.pose_tree_backward(own, glob, W1, W2, ctx.parents, feats, hidden, g.contiguous())
        return None, gg.reshape(glob.shape), gW1, gb1, gW2, gb2, None


class HierarchicalPoseEncoder(nn.Module):
    """LEAP-style encoder: (rots (B,24,9), Jtrs (B,24,3)) -> (B,144)
    (reference siren_modules.py:196-244)."""

    def __init__(self, num_joints=24, rel_joints=False, **kwargs):
        super().__init__()
        self.num_joints = num_joints
        self.rel_joints = rel_joints
        self.parents = SMPL_PARENTS
        self._index_cache = {}
        self.batched_levels = True   # False: the joint-by-joint form of the reference (tests compare the two)
        self.layer_0 = nn.Linear(12 * num_joints, 6)
        self.layers = nn.ModuleList([nn.Sequential(nn.Linear(19, 19), nn.ReLU(), nn.Linear(19, 6))
                                     for _ in range(num_joints)])

    def forward(self, rots, Jtrs):
        B = rots.shape[0]
        if self.rel_joints:
            par = torch.as_tensor(self.parents[1:], device=Jtrs.device, dtype=torch.long)
            Jtrs = torch.cat([Jtrs[:, :1], Jtrs[:, 1:] - Jtrs[:, par]], dim=1).detach()
        glob = self.layer_0(torch.cat([rots.reshape(B, -1), Jtrs.reshape(B, -1)], dim=-1))
        # training, and inference on the GPU (whose sums already run in another order than the CPU's); on the CPU without
        # gradients -- the oracle's side of the fixtures -- the reference's operation order is kept
        if self.batched_levels and (torch.is_grad_enabled() or rots.is_cuda):
            return self._forward_levels(rots, Jtrs, glob)
        feats = []
        for j in range(self.num_joints):
            p = self.parents[j]
            if p < 0:
                ref, up = Jtrs[:, j], glob
            else:
                ref = Jtrs[:, j] if self.rel_joints else Jtrs[:, j] - Jtrs[:, p]
                up = feats[p]
            bone_len = ref.norm(dim=-1, keepdim=True)
            feats.append(self.layers[j](torch.cat([rots[:, j], Jtrs[:, j], bone_len, up], dim=-1)))
        return torch.cat(feats, dim=-1)

    def _forward_levels(self, rots, Jtrs, glob):
        """The same encoder, one batched matrix product per LEVEL of the kinematic tree (9 levels) instead of two small
        products per joint (24 joints, ~190 launches and ~500 in backward for a 19-wide MLP): the per-joint parameters are
        stacked on the fly (their gradients come back as views of the stacked gradient), joints of equal depth share
        a baddbmm.  Same sums per joint up to the order inside a 19-term dot product."""
        B, Jn = rots.shape[0], self.num_joints
        key = str(Jtrs.device)
        if key not in self._index_cache:
            self._index_cache[key] = torch.as_tensor([max(p, 0) for p in self.parents], device=Jtrs.device, dtype=torch.long)
        ref = Jtrs if self.rel_joints else Jtrs - Jtrs[:, self._index_cache[key]]
        ref = torch.cat([Jtrs[:, :1], ref[:, 1:]], dim=1)                      # the root measures its own position
        bone_len = ref.norm(dim=-1, keepdim=True)                              # (B, J, 1)
        own = torch.cat([rots, Jtrs, bone_len], dim=-1).transpose(0, 1)        # (J, B, 13)
        W1 = torch.stack([self.layers[j][0].weight for j in range(Jn)])        # (J, 19, 19)
        b1 = torch.stack([self.layers[j][0].bias for j in range(Jn)]).unsqueeze(1)
        W2 = torch.stack([self.layers[j][2].weight for j in range(Jn)])        # (J, 6, 19)
        b2 = torch.stack([self.layers[j][2].bias for j in range(Jn)]).unsqueeze(1)
        if (B == 1 and rots.is_cuda and torch.is_grad_enabled() and not own.requires_grad and own.dtype == torch.float32
                and Jn <= 64 and os.environ.get("ARAH_POSE_TREE_OP", "1") != "0"):
            # a training step on the device: the whole tree as one launch each way (round 6; the level-by-level form below is
            # ~70 launches and ~130 in backward of a step that is bound by its host side)
            feats = _PoseTree.apply(own[:, 0].contiguous(), glob.reshape(-1), W1, b1.squeeze(1), W2, b2.squeeze(1),
                                    tuple(int(p) for p in self.parents))
            return feats.reshape(1, -1)
        feats = [None] * Jn
        levels = self._levels()
        sizes = [len(l) for l in levels]
        assert [j for l in levels for j in l] == list(range(Jn))               # SMPL numbers its joints level by level
        parts = [t.split(sizes) for t in (own, W1, b1, W2, b2)]                # one backward node per stacked tensor
        for li, level in enumerate(levels):
            own_l, W1_l, b1_l, W2_l, b2_l = (p[li] for p in parts)
            up = torch.stack([glob if self.parents[j] < 0 else feats[self.parents[j]] for j in level])   # (k, B, 6)
            x = torch.cat([own_l, up], dim=-1)                                 # (k, B, 19)
            h = torch.relu(torch.baddbmm(b1_l, x, W1_l.transpose(1, 2)))
            f = torch.baddbmm(b2_l, h, W2_l.transpose(1, 2))                   # (k, B, 6)
            for j, fj in zip(level, f.unbind(0)):
                feats[j] = fj
        return torch.cat(feats, dim=-1)

    def _levels(self):
        depth = {}
        for j in range(self.num_joints):   # parents precede children in SMPL's order
            depth[j] = 0 if self.parents[j] < 0 else depth[self.parents[j]] + 1
        out = [[] for _ in range(max(depth.values()) + 1)]
        for j in range(self.num_joints):
            out[depth[j]].append(j)
        return out


class HyperBVPNet(nn.Module):
    """Pose-conditioned hypernetwork for the canonical SDF (reference siren_modules.py:247-316).

    forward(model_input) -> {'model_in','model_out','params','decoder'}; ``decoder`` is an
    ``nn.Sequential`` of 6 x [EmittedFiLMLinear, Sine] + EmittedLinear, sliceable like the
    reference's (IDR:336-337 uses ``sdf_network[:-1]`` / ``[-1]``).
    """

    def __init__(self, out_features=1, type="sine", in_features=2, hyper_in_ch=92, mode="mlp",
                 hidden_features=256, num_hidden_layers=3, hierarchical_pose=False,
                 rel_joints=False, use_FiLM=False, **kwargs):
        super().__init__()
        if type != "sine":
            raise NotImplementedError("HyperBVPNet only supports sine activations")
        if not (use_FiLM and hierarchical_pose):
            raise NotImplementedError("only the FiLM + hierarchical-pose variant used by the ARAH "
                                      "configs is implemented")
        self.mode = mode
        self.use_FiLM = use_FiLM
        self.hierarchical_pose = hierarchical_pose
        self.net = HyperFCFiLM(hyper_in_ch, hidden_features, num_hidden_layers, in_features, out_features)
        self.pose_encoder = HierarchicalPoseEncoder(rel_joints=rel_joints)

    def forward(self, model_input):
        coords = model_input["coords"].clone().detach().requires_grad_(True)
        if "rots_noise" in model_input:
            model_input["rots"] = model_input["rots"] + model_input["rots_noise"]
        cond = self.pose_encoder(model_input["rots"], model_input["Jtrs"])
        latent = model_input.get("latent")
        if latent is None:
            raise NotImplementedError("FiLM SDF decoder needs a geometry latent code")
        decoder = self.net(cond, latent)
        # the emitted network AT `coords` (siren_modules.py:303-316 returns it; the renderer passes one dummy point and never
        # looks at the value): skipped when the caller says so -- 27 launches of a host-bound training step
        out = None if model_input.get("skip_model_out") else decoder(coords)
        B = coords.shape[0]
        params = [decoder[i][0].weights.reshape(B, -1) for i in range(len(decoder) - 1)]
        params.append(decoder[-1].weights.reshape(B, -1))
        return {"model_in": coords, "model_out": out, "params": params, "decoder": decoder}


# --------------------------------------------------------------------------------------
# per-sample networks with learnable (not emitted) weights
# --------------------------------------------------------------------------------------
def _wn_linear(n_in, n_out, weight_norm):
    lin = nn.Linear(n_in, n_out)
    if weight_norm:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            lin = nn.utils.weight_norm(lin)  # keeps reference names weight_g / weight_v
    return lin


_FOLD_CACHE = [None]   # inside `fold_cache()`: id(layer) -> its folded weight (one autograd node for all uses in a step)


class fold_cache:
    """Within this context a weight-normed layer is folded ONCE however often it is evaluated: a training step runs the skinning
    MLP three times (the loss' query, the re-attachment, ...) and folded its five layers each time -- ten launches, seventeen in
    backward and the additions that merge the three gradients.  Same values; the gradient is the sum autograd forms anyway."""

    def __enter__(self):
        self.prev = _FOLD_CACHE[0]
        _FOLD_CACHE[0] = {}
        return self

    def __exit__(self, *exc):
        _FOLD_CACHE[0] = self.prev
        return False


def folded_weight(lin):
    """Effective (out,in) weight of a (possibly weight-normed) linear layer: g * v / |v|_row."""
    if hasattr(lin, "weight_g"):
        cache = _FOLD_CACHE[0]
        if cache is not None and torch.is_grad_enabled():
            hit = cache.get(id(lin))
            if hit is None:
                hit = cache[id(lin)] = torch._weight_norm(lin.weight_v, lin.weight_g, 0)
            return hit
        # the op torch.nn.utils.weight_norm itself evaluates (fused forward and backward: one launch each, against three
        # and a dozen for g * v / |v| spelled out)
        return torch._weight_norm(lin.weight_v, lin.weight_g, 0)
    return lin.weight


class Deformer(nn.Module):
    """Forward-skinning MLP 3 -> 128 x4 -> 25, Softplus(beta=100)
    (reference metaavatar/models/decoder.py:133-233; ARAH configs use no skip / no cond)."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(), cond_in=(), cond_dim=96,
                 multires=0, bias=1.0, geometric_init=True, weight_norm=True, **kwargs):
        super().__init__()
        if multires > 0 or len(skip_in) or len(cond_in) or geometric_init:
            raise NotImplementedError("only the plain weight-normed MLP used by the ARAH configs")
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            setattr(self, "lin%d" % l, _wn_linear(dims[l], dims[l + 1], weight_norm))
        self.activation = nn.Softplus(beta=100)

    def forward(self, p, c=None, **kwargs):
        B, n, _ = p.shape
        x = p.reshape(-1, p.shape[-1])
        from .tall import tall_linear
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin%d" % l)
            x = tall_linear(x, folded_weight(lin), lin.bias)   # weight gradients over 1e5 samples: see tall.py
            if l < self.num_layers - 2:
                x = self.activation(x)
        return x.reshape(B, n, -1)

    def export_weights(self):
        """[(W (out,in), b (out,)), ...] fp32 contiguous, weight-norm folded."""
        out = []
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin%d" % l)
            out.append((folded_weight(lin).detach().float().contiguous(), lin.bias.detach().float().contiguous()))
        return out


class SkinningModel(nn.Module):
    """Thin holder, ``decode_w(p, c, forward=True)`` (reference skinning_model.py:23-35)."""

    def __init__(self, skinning_decoder_fwd=None, **kwargs):
        super().__init__()
        self.skinning_decoder_fwd = skinning_decoder_fwd

    def forward(self):
        raise NotImplementedError("You should not call the forward function of the skinning model.")

    def decode_w(self, p, c=None, forward=True, **kwargs):
        if not forward:
            raise ValueError("This skinning model does not have backward networks.")
        return self.skinning_decoder_fwd(p, c=c, **kwargs)


def positional_encoding(x, n_freqs):
    """NeRF embedding [x, sin(2^k x), cos(2^k x)]_k (reference embedder.py:6-51)."""
    out = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, dim=-1)


class RenderingNetwork(nn.Module):
    """Colour MLP (reference metaavatar_render/models/decoder.py:10-124)."""

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires=0,
                 multires_view=0, skips=(), squeeze_out=True, rel_joints=True, pose_encoder="leap"):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        self.multires = multires
        self.multires_view = multires_view
        self.pose_encoder_type = pose_encoder
        dims = [d_in + d_feature] + [d_hidden] * n_layers + [d_out]
        if multires > 0:
            dims[0] += 6 * multires
        if multires_view > 0:
            dims[0] += 6 * multires_view
        if pose_encoder == "leap":
            self.pose_encoder = HierarchicalPoseEncoder(rel_joints=rel_joints)
        self.skips = list(skips)
        for s in self.skips:
            dims[s] = dims[s] // 2 + dims[0]
        self.dims = dims
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            n_out = dims[l + 1] - dims[0] if (l + 1) in self.skips else dims[l + 1]
            setattr(self, "lin%d" % l, _wn_linear(dims[l], n_out, weight_norm))
        self.relu = nn.ReLU()

    def pose_vector(self, pose_feature):
        """The per-frame constant tail of the input (1, n_pose) or None."""
        t = self.pose_encoder_type
        if t == "leap":
            return self.pose_encoder(pose_feature["rots_full"][:1], pose_feature["Jtrs_posed"][:1])
        if t in ("root", "hybrid"):
            rot = pose_feature["rots_full"][:1, :1].reshape(1, 9)
            tr = pose_feature["Jtrs_posed"][:1, :1].reshape(1, 3)
            if "rot_noise" in pose_feature and "trans_noise" in pose_feature:
                rot = rot + pose_feature["rot_noise"]
                tr = tr + pose_feature["trans_noise"]
            vec = torch.cat([rot, tr], dim=-1)
            if t == "hybrid":
                vec = torch.cat([vec, pose_feature["latent_code"]], dim=-1)
            return vec
        if t == "latent":
            return pose_feature["latent_code"]
        return None

    def forward(self, points, normals, view_dirs, sdf_feature, pose_feature):
        if self.multires > 0:
            points = positional_encoding(points, self.multires)
        if self.multires_view > 0:
            view_dirs = positional_encoding(view_dirs, self.multires_view)
        feat = sdf_feature
        pv = self.pose_vector(pose_feature)
        if pv is not None:
            feat = torch.cat([feat, pv.expand(feat.shape[0], -1)], dim=-1)
        if self.mode == "idr":
            inp = torch.cat([points, view_dirs, normals, feat], dim=-1)
        elif self.mode == "no_view_dir":
            inp = torch.cat([points, normals, feat], dim=-1)
        elif self.mode == "no_normal":
            inp = torch.cat([points, view_dirs, feat], dim=-1)
        else:
            raise ValueError("unknown rendering mode %r" % self.mode)
        x = inp
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin%d" % l)
            x = lin(torch.cat([inp, x], dim=-1)) if l in self.skips else lin(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return torch.sigmoid(x) if self.squeeze_out else x

    def export_weights(self):
        out = []
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin%d" % l)
            out.append((folded_weight(lin).detach().float().contiguous(), lin.bias.detach().float().contiguous()))
        return out


class SingleVarianceNetwork(nn.Module):
    """Learnable VolSDF beta = |variance| (reference decoder.py:127-133)."""

    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        return torch.ones_like(x) * torch.linalg.norm(self.variance)


decoder_dict = {
    "hyper_bvp": HyperBVPNet,
    "deformer_mlp": Deformer,
}
