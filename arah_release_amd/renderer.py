"""Drop-in counterparts of the reference's model / renderer classes, backed by the HIP kernels.

Same constructors, same ``forward`` signatures, same dict-in / dict-out contract as

    MetaAvatarRender   im2mesh/metaavatar_render/models/__init__.py:17-315
    IDHRNetwork        im2mesh/metaavatar_render/renderer/implicit_differentiable_renderer.py:15-259
    BodyRayTracing     im2mesh/metaavatar_render/renderer/ray_tracing.py:13-172

(paths relative to the reference root).  The per-frame hypernetwork stays PyTorch; everything
evaluated per ray / per sample goes through the C ABI (hip.py -> libarah_hip.so).  There is no
torch implementation of the hot loops in this package: without the HIP library or without a GPU
the forward raises.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import hip, training
from .nets import folded_weight


def draw_uniform(shape, device, tag):
    """torch.rand draws of the training path, in the order the reference makes them ('steps', 'near', 'far' in
    ray_tracing.py:304, then 'eikonal' in implicit_differentiable_renderer.py:127).  Tests replace this hook
    to replay the reference's draws."""
    return torch.rand(shape, device=device)


def _emitted_sdf_layers(sdf_network):
    """nn.Sequential of 6 x [EmittedFiLMLinear, Sine] + EmittedLinear -> raw row-major tensors."""
    n = len(sdf_network)
    layers, freqs, phases = [], [], []
    for i in range(n - 1):
        lin = sdf_network[i][0]
        layers.append((lin.weights[0], lin.biases.reshape(-1)))
        freqs.append(lin.freq.reshape(-1))
        phases.append(lin.phase_shift.reshape(-1))
    last = sdf_network[n - 1]
    layers.append((last.weights[0], last.biases.reshape(-1)))
    return layers, torch.cat(freqs), torch.cat(phases)


# Inference caches live OUTSIDE the modules (weakly keyed by them): a torch.cuda.Event or a CUDAGraph in a module's __dict__ made
# copy.deepcopy / pickle / torch.save(model) of a model that had rendered a frame raise.
import weakref
_FOLDED = weakref.WeakKeyDictionary()          # weight-normed MLP -> (key, folded layers, event, device, stream)
_GRAPHED = weakref.WeakKeyDictionary()         # MetaAvatarRender -> {"eval": _GraphedDecoder, "train": _TrainGraphedDecoder}


def invalidate_caches(module):
    """Forget the folded weights / captured hypernetworks of `module` and everything below it.  The caches notice a parameter
    that was re-allocated or written in place through autograd's version counter, NOT a write through `p.data` (a broadcast
    into `p.data`, an EMA update, manual surgery): callers that write that way call this (train.broadcast_state does; a
    load_state_dict post-hook of MetaAvatarRender does)."""
    for m in module.modules():
        _FOLDED.pop(m, None)
        _GRAPHED.pop(m, None)


def _mlp_layers(module):
    """Row-major (W, b) of a weight-normed MLP, W = g v / |v| folded.  Without gradients (inference) the folded weights are
    kept (weakly keyed by the module) until a parameter changes (data pointer or in-place version; see invalidate_caches for
    writes through .data): a test sequence folds the skinning and colour networks once instead of eleven launches per frame.
    An event orders other streams behind the fold."""
    cache = not torch.is_grad_enabled()
    if cache:
        key = tuple((p.data_ptr(), p._version) for p in module.parameters())
        hit = _FOLDED.get(module)
        if hit is not None and hit[0] == key:
            if hit[2] is not None and torch.cuda.current_stream(hit[3]).cuda_stream != hit[4]:
                torch.cuda.current_stream(hit[3]).wait_event(hit[2])
            return hit[1]
    out = []
    for l in range(module.num_layers - 1):
        lin = getattr(module, "lin%d" % l)
        out.append((folded_weight(lin), lin.bias))
    if cache:
        out = [(w.detach().float().contiguous(), b.detach().float().contiguous()) for w, b in out]
        dev = out[0][0].device
        ev = sid = None
        if dev.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            sid = torch.cuda.current_stream(dev).cuda_stream
        _FOLDED[module] = (key, out, ev, dev, sid)
    return out


class _GraphedDecoder:
    """The pose encoder + hypernetwork of one frame (sdf_decoder: ~110 small launches, 2.5 ms of a launch-bound GPU in front of
    every frame) captured once per device as a graph and replayed: inference only (no gradients, no input noise).  The
    graph's inputs are copied into its static tensors, its outputs -- the emitted layers -- are copied out into fresh
    tensors in one multi-tensor launch, so that what the caller sees (inputs['sdf_network'], the returned sdf_params) is its
    own, as with the eager call; the kernels and their order are the eager call's: same bits.  Parameters are read in place,
    an update is seen by the next replay; a re-allocated parameter re-captures.  ARAH_HYPERNET_GRAPH=0 switches it off."""

    def __init__(self, decoder):
        self.decoder = decoder
        self.entries = {}
        self.broken = False

    def _key(self, dev):
        ptrs = 0
        for p in self.decoder.parameters():
            ptrs = (ptrs * 1000003 + p.data_ptr()) & 0xFFFFFFFFFFFF
        return (dev, ptrs)   # ONE graph per device: a replay runs on whatever stream the frame is on, an event hands the
                             # static tensors from one frame to the next (the decoders of frames in flight take turns, ~0.5 ms each)

    def __call__(self, decoder_input):
        dev = decoder_input["rots"].device
        key = self._key(dev)
        e = self.entries.get(key)
        if e is None:
            if len(self.entries) >= 8:
                self.entries.pop(next(iter(self.entries)))
            static_in = {k: v.detach().clone() for k, v in decoder_input.items()}
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                       # warm-up outside the capture (library handles, workspaces, attribute setup)
                    self.decoder(dict(static_in))
            cur.wait_stream(side)
            torch.cuda.current_stream(dev).synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.decoder(dict(static_in))
            srcs = []
            for i in range(len(out["decoder"])):
                lin = out["decoder"][i][0] if i < len(out["decoder"]) - 1 else out["decoder"][i]
                srcs += [lin.weights, lin.biases] + ([lin.freq, lin.phase_shift] if hasattr(lin, "freq") else [])
            e = self.entries[key] = {"graph": graph, "in": static_in, "out": out, "srcs": srcs, "done": None}
        cur = torch.cuda.current_stream(dev)
        if e["done"] is not None:
            cur.wait_event(e["done"])
        for k, v in decoder_input.items():
            e["in"][k].copy_(v, non_blocking=True)
        e["graph"].replay()
        from .nets import EmittedFiLMLinear, EmittedLinear, Sine
        fresh = [torch.empty_like(t) for t in e["srcs"]]
        torch._foreach_copy_(fresh, e["srcs"])
        model_out = e["out"]["model_out"].clone()
        e["done"] = torch.cuda.Event()
        e["done"].record(cur)
        mods, it = [], iter(fresh)
        n = len(e["out"]["decoder"])
        for i in range(n - 1):
            w, b, f, ph = next(it), next(it), next(it), next(it)
            mods.append(nn.Sequential(EmittedFiLMLinear(w, b, f, ph), Sine()))
        w, b = next(it), next(it)
        mods.append(EmittedLinear(w, b))
        decoder = nn.Sequential(*mods)
        B = e["in"]["coords"].shape[0]
        params = [decoder[i][0].weights.reshape(B, -1) for i in range(n - 1)] + [decoder[-1].weights.reshape(B, -1)]
        return {"model_in": e["out"]["model_in"], "model_out": model_out, "params": params, "decoder": decoder}


class _FlatDecoder(nn.Module):
    """sdf_decoder as a function of tensors to a tuple of tensors (what torch.cuda.make_graphed_callables captures)."""

    def __init__(self, decoder):
        super().__init__()
        self.decoder = decoder

    def forward(self, rots, Jtrs, latent):
        out = self.decoder({"coords": torch.zeros(1, 1, 3, dtype=torch.float32, device=rots.device), "rots": rots, "Jtrs": Jtrs,
                            "latent": latent})
        d = out["decoder"]
        flat = []
        for i in range(len(d) - 1):
            lin = d[i][0]
            flat += [lin.weights, lin.biases, lin.freq, lin.phase_shift]
        flat += [d[len(d) - 1].weights, d[len(d) - 1].biases, out["model_out"]]
        return tuple(flat)


class _TrainGraphedDecoder:
    """The pose encoder + hypernetwork of a TRAINING step -- ~95 launches forward, ~300 in backward, a third of the step's
    1100 and of the ~28 ms of host time they cost (profiles/r05_train_regions.txt; on a box with a slower host the step is
    host-bound: 60-65 ms instead of 25) -- captured once as a forward and a backward graph (torch.cuda.make_graphed_callables)
    and replayed inside autograd: same kernels, same order, same bits; the gradients of the 85.6 M hypernetwork parameters
    arrive in the graph's static buffers.  Used when only the latent code of the call's inputs requires gradients (the
    reference's training configuration without train_smpl); a re-allocated parameter re-captures.
    OFF by default (ARAH_TRAIN_HYPERNET_GRAPH=1 switches it on): measured on the MI355X box it does not pay -- 26.2 ms per
    step against 24.7 eager, and the process spends MORE host time per step (37 against 27 ms: replaying a 300-node graph is not
    cheaper for this runtime than enqueueing its kernels); beside a busy loop on the same core (a host half as fast) 41.6
    against 41.0 ms (tools/probes/train_host.py, profiles/r05_train_host.txt).  Gradients are the eager step's
    (tests/test_hip_parity.py::test_graphed_training_hypernetwork_is_the_eager_one)."""

    def __init__(self, decoder):
        self.decoder = decoder
        self.key = None
        self.fn = None
        self.broken = False

    def _key(self, dev):
        ptrs = 0
        for p in self.decoder.parameters():
            ptrs = (ptrs * 1000003 + p.data_ptr() + (7 if p.requires_grad else 0)) & 0xFFFFFFFFFFFF
        return (dev, ptrs)

    def __call__(self, decoder_input):
        rots, Jtrs, latent = decoder_input["rots"], decoder_input["Jtrs"], decoder_input["latent"]
        if "rots_noise" in decoder_input:   # siren_modules.py:288-289: outside the graph, one launch
            rots = rots + decoder_input["rots_noise"]
        dev = rots.device
        key = self._key(dev)
        if self.fn is None or self.key != key:
            flat = _FlatDecoder(self.decoder)
            sample = (rots.detach().clone(), Jtrs.detach().clone(), latent.detach().clone().requires_grad_(True))
            torch.cuda.current_stream(dev).synchronize()
            self.fn = torch.cuda.make_graphed_callables(flat, sample, allow_unused_input=True)
            self.key = key
        out = self.fn(rots.detach(), Jtrs.detach(), latent)
        from .nets import EmittedFiLMLinear, EmittedLinear, Sine
        mods, it = [], iter(out[:-1])
        n = (len(out) - 3) // 4 + 1
        for _ in range(n - 1):
            w, b, f, ph = next(it), next(it), next(it), next(it)
            mods.append(nn.Sequential(EmittedFiLMLinear(w, b, f, ph), Sine()))
        w, b = next(it), next(it)
        mods.append(EmittedLinear(w, b))
        decoder = nn.Sequential(*mods)
        B = 1
        params = [decoder[i][0].weights.reshape(B, -1) for i in range(n - 1)] + [decoder[-1].weights.reshape(B, -1)]
        return {"model_in": decoder_input["coords"], "model_out": out[-1], "params": params, "decoder": decoder}


def build_frame(sdf_network, skinning_model, rendering_network, deviation_network, pose_cond, smpl_verts,
                skinning_weights, bone_transforms, trans, coord_min, coord_max, center, precision=None, body_tables=None):
    """Pack one temporal frame for the kernels (weights emitted by the hypernetwork + body).
    precision: hip.PRECISION_SPLIT_F16 / hip.PRECISION_FP32; None = env ARAH_PRECISION (default split)."""
    with torch.no_grad():
        sdf_layers, freq, phase = _emitted_sdf_layers(sdf_network)
        skin_layers = _mlp_layers(skinning_model.skinning_decoder_fwd)
        color_layers, mode, pose_vec, beta = None, hip.COLOR_NO_VIEW_DIR, None, 1e-3
        if rendering_network is not None:
            rn = rendering_network
            if rn.mode == "idr":
                if rn.multires_view != 4 or rn.multires != 0:
                    raise ValueError("kernels are built for multires_view=4, multires=0 (ARAH configs)")
                mode = hip.COLOR_IDR
            elif rn.mode == "no_view_dir":
                if rn.multires != 0:
                    raise ValueError("kernels are built for multires=0 (ARAH configs)")
                mode = hip.COLOR_NO_VIEW_DIR
            else:
                raise ValueError("rendering mode %r is not used by any ARAH config" % rn.mode)
            if list(rn.skips) != [3] or rn.num_layers != 7:
                raise ValueError("kernels are built for the 5-hidden-layer colour MLP with a skip at 3")
            color_layers = _mlp_layers(rn)
            pose_vec = rn.pose_vector(pose_cond)
            beta = torch.linalg.norm(deviation_network.variance).reshape(1)   # decoder.py:132-133; stays on the device
        # trans / center / coord_min / coord_max go down as device tensors: no host copy, no stream drain per frame
        return hip.Frame(sdf_layers, freq, phase, skin_layers, color_layers, mode, pose_vec, beta,
                         smpl_verts[0], skinning_weights[0], bone_transforms[0], trans.reshape(-1)[:3],
                         center.reshape(-1)[:3], coord_min.reshape(-1)[:1], coord_max.reshape(-1)[:1],
                         precision=precision, body_tables=body_tables)


class BodyRayTracing(nn.Module):
    """Ray tracer for the articulated SDF: sphere tracing + joint root finding, then depth sampling
    and per-sample canonicalisation (reference ray_tracing.py:13-172)."""

    def __init__(self, root_finding_threshold=1.0e-5, sphere_tracing_iters=50, n_steps=64,
                 near_surface_vol_samples=16, far_surface_vol_samples=16, surface_vol_range=0.05,
                 sample_bg_pts=0, low_vram=False):
        super().__init__()
        if (root_finding_threshold, sphere_tracing_iters, surface_vol_range) != (1.0e-5, 50, 0.05):
            raise ValueError("the kernels hard-wire threshold 1e-5, 50 sphere-tracing steps and a 5 cm "
                             "surface range, like every ARAH config")
        self.root_finding_threshold = root_finding_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.n_steps = n_steps
        self.near_surface_vol_samples = near_surface_vol_samples
        self.far_surface_vol_samples = far_surface_vol_samples
        self.surface_vol_range = surface_vol_range
        self.sample_bg_pts = sample_bg_pts
        self.low_vram = low_vram
        self.pending_bounds_ok = None   # training: the near <= far verdict of the last call, still on the device (see forward)
        self._ws = {}          # one caller-owned scratch per (device, stream): frames in flight on different streams
        self._sampling = {}
        # False: exact lazy shading (normals/colours only where the VolSDF density is > 0); True: shade every
        # valid sample like the reference.  Same image either way (bit for bit); see DESIGN.md section 4.
        self.full_shading = os.environ.get("ARAH_FULL_SHADING", "0") == "1"
        # per-call engine switches of the C ABI (None: the process default, hip.default_shade_engine / default_canon_kernel)
        self.shade_engine = None
        self.canon_kernel = None
        self._events = {}

    def workspace(self, device):
        """Scratch of the C ABI for the CURRENT stream of `device` (render_sequence keeps several frames in flight, each
        on its own stream; they must not share scratch)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (device, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
        if key not in self._ws:
            if len(self._ws) >= 8:   # streams come and go: keep the eight most recent scratches (each is ~1 GB at 512 x 512 x 64;
                self._ws.pop(next(iter(self._ws)))   # a dropped one returns to the pool of the stream it was allocated on)
            self._ws[key] = hip.Workspace(device)
        return self._ws[key]

    def workspaces(self):
        return list(self._ws.values())

    def sampling(self, device, cano_view_dirs=True, render_last_pt=False, full_shading=None):
        full = bool(self.full_shading if full_shading is None else full_shading)
        key = (str(device), bool(cano_view_dirs), bool(render_last_pt), full, self.shade_engine, self.canon_kernel)
        if key not in self._sampling:
            sm = hip.Sampling(device, self.n_steps, self.near_surface_vol_samples, self.far_surface_vol_samples,
                              cano_view_dirs, render_last_pt, full, self.shade_engine, self.canon_kernel)
            for which, (a, b) in self._events.items():
                sm.set_events(which, a, b)
            self._sampling[key] = sm
        return self._sampling[key]

    def set_events(self, which, start=None, stop=None):
        """Profiling events ("canon" / "density" / "shade", hip.Sampling.set_events) on every sampling object of this tracer,
        present and future: bench.py's kernel timing.  They belong to the objects, not to the library."""
        if start is None or stop is None:
            self._events.pop(which, None)
        else:
            self._events[which] = (start, stop)
        for sm in self._sampling.values():
            sm.set_events(which, start, stop)

    def forward(self, sdf_network, skinning_model, cam_loc, ray_directions, body_bounds_intersections, loc,
                sc_factor, smpl_verts, smpl_verts_cano, skinning_weights, vol_feat, bone_transforms, trans,
                coord_min, coord_max, center, eval_mode=False, frame=None):
        assert self.near_surface_vol_samples > 0 or self.far_surface_vol_samples > 0
        B, N, _ = ray_directions.shape
        if N == 0:
            raise ValueError("No valid depth.")
        # RT:182 asserts near <= far; the test is a device -> host round trip.  Eval: here, like the reference.  Training: the
        # verdict is looked at behind the step's first compaction (forward_train), where the host waits for the device anyway --
        # here it would hold the tracer's launches back until the hypernetwork has drained
        bounds_ok = (body_bounds_intersections[..., 0] <= body_bounds_intersections[..., 1]).all()
        if eval_mode or os.environ.get("ARAH_TRAIN_LATE_BOUNDS_CHECK", "1") == "0":
            if not bool(bounds_ok):
                raise AssertionError("near bound exceeds far bound")
            self.pending_bounds_ok = None
        else:
            self.pending_bounds_ok = bounds_ok
        dev = ray_directions.device
        if frame is None:
            frame = build_frame(sdf_network, skinning_model, None, None, None, smpl_verts, skinning_weights,
                                bone_transforms, trans, coord_min, coord_max, center)
        ws = self.workspace(dev)
        cam = cam_loc.reshape(B, 3)
        d = ray_directions.reshape(B * N, 3)
        nf = body_bounds_intersections.reshape(B * N, 2)
        # training: joint root find on every ray (RT:249) and stratified jitter of the depth samples (RT:319-346)
        xn, T, conv, start, end = hip.trace(frame, ws, cam, d, nf, root_find_all=not eval_mode)
        rand = None
        if not eval_mode:
            S_, nn_, nf_ = self.n_steps, self.near_surface_vol_samples, self.far_surface_vol_samples
            rand = (draw_uniform((B * N, S_), dev, "steps"), draw_uniform((B * N, nn_ + 1), dev, "near"),
                    draw_uniform((B * N, max(nf_, 1)), dev, "far"))
        z, pts, Ts, mask = hip.sample_canonicalize(frame, ws, self.sampling(dev), cam, d, nf, conv, start, end, rand)
        S = self.n_steps
        return (xn.reshape(B, N, 3), conv.bool().reshape(B, N), start.reshape(B, N), pts.reshape(B, N, S, 3),
                z.reshape(B, N, S), Ts.reshape(B, N, S, 4, 4), mask.bool().reshape(B, N, S))


class IDHRNetwork(nn.Module):
    """Implicit differentiable human renderer (reference implicit_differentiable_renderer.py:15-259)."""

    def __init__(self, deviation_network, rendering_network, skinning_model, ray_tracer, cano_view_dirs=True,
                 train_skinning_net=False, render_last_pt=False, low_vram=False):
        super().__init__()
        self.deviation_network = deviation_network
        self.rendering_network = rendering_network
        self.skinning_model = skinning_model
        self.ray_tracer = ray_tracer
        self.cano_view_dirs = cano_view_dirs
        self.train_skinning_net = train_skinning_net
        self.render_last_pt = render_last_pt
        self.low_vram = low_vram
        self.last_counters = None
        # range guard of the split engine (ArahCounters.n_split_nonfinite): see _split_guard
        self.split_nonfinite = 0
        self._guard = {}
        # "lazy" (default): the count of frame k is looked at when frame k + 1 starts -- no stream drain, frame k itself is
        # returned as rendered (with a warning).  "strict" (ARAH_SPLIT_GUARD=strict): the count is read behind every frame
        # (one synchronisation per frame) and a frame that tripped it is rendered AGAIN on the exact fp32 engine before it
        # is returned.
        self.guard_mode = os.environ.get("ARAH_SPLIT_GUARD", "lazy")
        # Lazy or full shading, decided per frame from what earlier frames needed (both give the same image bit for bit,
        # DESIGN.md section 4).  The density pre-pass of lazy shading costs D = one SDF trunk per valid sample and saves the
        # normal sweep + colour MLP of the samples whose density is exactly 0; once the share r of samples with density > 0
        # exceeds 1 - D / S (S = shading every sample: ~0.7 on the MI355X) the pre-pass is pure overhead -- that happens when
        # the LEARNED VolSDF beta is large (r = 0.04 at the reference's initial 1e-3, 0.95 at 3e-2).  r of frame k travels to
        # the host with the range guard's counters (no stream drain) and picks the mode of the frames that follow; while in
        # full mode every 16th frame runs lazily again to keep r current.  ARAH_ADAPTIVE_SHADING=0 / tracer.full_shading
        # pin the mode.
        self.adaptive_shading = os.environ.get("ARAH_ADAPTIVE_SHADING", "1") != "0"
        self.shade_ratio = None          # last measured r
        self._shade_full = False
        self._shade_since_probe = 0
        # Tiered evaluation (csrc/tier.hpp): samples outside the frame's posed fat body are certified sigma = +0 without loops C
        # and D; same image and masks bit for bit (tests/test_tiered.py).  ARAH_TIERING=0 / .tiering = False: every sample
        # through the exact kernels, like the reference.  Lazy shading only: a frame that shades everything (large beta) has a
        # band as wide as the body's box and nothing to skip.
        self.tiering = os.environ.get("ARAH_TIERING", "1") != "0"
        # ... and only while it pays: the band is 18 beta wide, and with a large learned beta (or a subject whose bitmap overflows
        # its buffers) hardly a sample lies outside it.  The share of never-evaluated samples of earlier tiered frames arrives with
        # the range guard's counters; below a third the following frames run untiered (one phase, no bitmap), every 16th one
        # tiered again to keep the share current.  Follows `adaptive_shading` (bench.py pins both for its named passes).
        self.tier_share = None
        self._tier_off = False
        self._tier_since_probe = 0
        self.precision = None    # GEMM engine frames are prepared for: None = the process default (ARAH_PRECISION, split),
                                 # hip.PRECISION_FP32 / PRECISION_SPLIT_F16 = this renderer's own choice (bench.py's passes)
        self._precision = None   # becomes hip.PRECISION_FP32 once the range guard has fired: overrides `precision`

    def _regulariser_probe(self, input, frame, sdf_network, B, n_reg, dev):
        """The regulariser queries (IDR:104-128) through the training kernel without its colour half: value and gradient in one
        launch, the second-order path of the eikonal term in one more (training.SdfNormal) -- an autograd SIREN is ~100 launches
        forward and ~150 backward for the same numbers.  Uses the scratch of the CURRENT stream.
        -> off-surface sdf (B, n_reg, 1), eikonal gradients (B n_reg, 3), inside sdf or None."""
        eik = ((draw_uniform((B, n_reg, 3), dev, "eikonal") - 0.5) * 2).reshape(-1, 3)
        pts_in = input["points_inside"].reshape(-1, 3) if "points_inside" in input else eik[:0]
        probe = torch.cat([eik, input["points_uniform"].reshape(-1, 3), pts_in], dim=0)
        sdf_probe, n_probe = training.sdf_normal_hip(frame, self.ray_tracer.workspace(dev), sdf_network, probe)
        uniform_sdf = sdf_probe[B * n_reg:2 * B * n_reg, :].reshape(B, n_reg, 1)
        grad_eik = n_probe[:B * n_reg]
        inside_sdf = sdf_probe[2 * B * n_reg:].reshape(input["points_inside"].shape[:-1] + (1,)).squeeze(0) \
            if "points_inside" in input else None
        return uniform_sdf, grad_eik, inside_sdf

    def forward_train(self, input):
        """Training forward (IDR:42-248): HIP kernels for the ray tracer (no_grad, like the reference), autograd
        for loop D and the regulariser queries (training.py)."""
        from .nets import fold_cache
        with fold_cache():   # a weight-normed layer evaluated several times in this step is folded once
            return self._forward_train(input)

    def _late_bounds_check(self):
        """The ray tracer's near <= far assertion (RT:182) of a training step, behind the step's compaction."""
        ok, self.ray_tracer.pending_bounds_ok = getattr(self.ray_tracer, "pending_bounds_ok", None), None
        if ok is not None and not bool(ok):
            raise AssertionError("near bound exceeds far bound")

    def _forward_train(self, input):
        ray_dirs, cam_loc = input["ray_dirs"], input["cam_loc"]
        sdf_network, pose_cond = input["sdf_network"], input["pose_cond"]
        cmin, cmax, center = input["coord_min"], input["coord_max"], input["center"]
        B, N, _ = ray_dirs.shape
        dev = ray_dirs.device
        pred_weights = None
        use_hip_shading = os.environ.get("ARAH_TRAIN_AUTOGRAD", "0") != "1"
        # (Round 6 tried the skinning-weight query of the loss and the regulariser queries on a side stream next to the ray
        # tracer's dependent chains: no gain -- 23.4-23.5 ms per step either way, and 12 ms more CPU in the waits between the
        # streams of the backward; removed.  What does pay is the ORDER on the one stream: the query's ~50 launches are issued
        # behind the tracer's, while the GPU works through loops A-C and this thread would otherwise wait for them at the
        # compaction below -- see further down.)
        # one packed frame per step serves the ray tracer (loops A-C) and the hand-written loop-D op
        frame = None
        if ray_dirs.is_cuda:   # (CPU: only reachable with a stubbed ray tracer, e.g. the gloo DDP test; autograd loop D)
            frame = build_frame(sdf_network, self.skinning_model, self.rendering_network, self.deviation_network,
                                pose_cond, input["smpl_verts"], input["skinning_weights"], input["bone_transforms"],
                                input["trans"], cmin, cmax, center, body_tables=input.get("_body_tables"))
        n_reg = 1024
        with torch.no_grad():
            xn, _, _, s_pts, s_z, s_T, s_mask = self.ray_tracer(
                sdf_network, self.skinning_model, cam_loc=cam_loc, ray_directions=ray_dirs,
                body_bounds_intersections=input["body_bounds_intersections"], loc=input["loc"],
                sc_factor=input["sc_factor"], smpl_verts=input["smpl_verts"], smpl_verts_cano=input["minimal_shape"],
                skinning_weights=input["skinning_weights"], vol_feat=input["vol_feat"],
                bone_transforms=input["bone_transforms"], trans=input["trans"], coord_min=cmin, coord_max=cmax,
                center=center, eval_mode=False, frame=frame)
        if frame is not None and use_hip_shading and os.environ.get("ARAH_TRAIN_PROBE_OP", "1") != "0":
            uniform_sdf, grad_eik, inside_sdf = self._regulariser_probe(input, frame, sdf_network, B, n_reg, dev)
        else:
            eik = ((draw_uniform((B, n_reg, 3), dev, "eikonal") - 0.5) * 2).reshape(-1, 3)
            inside_sdf = sdf_network(input["points_inside"]).squeeze(0) if "points_inside" in input else None
            probe = torch.cat([eik, input["points_uniform"].reshape(-1, 3)], dim=0).requires_grad_(True)
            sdf_probe = sdf_network(probe).squeeze(0)
            uniform_sdf = sdf_probe[B * n_reg:2 * B * n_reg, :].reshape(B, n_reg, 1)
            grad_eik = torch.autograd.grad(sdf_probe, probe, torch.ones_like(sdf_probe), create_graph=True,
                                           retain_graph=True)[0][:B * n_reg]
        if "points_skinning" in input:   # (the loss' skinning term, models/__init__.py:228-231; independent of everything above)
            pred_weights = training.query_weights(input["points_skinning"], cmin, cmax, center, self.skinning_model)
        vol_mask = s_mask.any(-1)
        dirs_in, ray_augm = ray_dirs, False
        if "view_noise" in pose_cond:
            vn = pose_cond["view_noise"]
            if vn is None:
                dirs_in = torch.zeros_like(ray_dirs)
            elif vn.shape[-2:] == (3, 3):
                dirs_in = torch.matmul(vn, ray_dirs.transpose(1, 2)).transpose(1, 2)
                ray_augm = True
            else:
                dirs_in = ray_dirs + vn
        from .nets import SingleVarianceNetwork
        if (frame is not None and use_hip_shading and isinstance(self.deviation_network, SingleVarianceNetwork)
                and os.environ.get("ARAH_TRAIN_COMPOSITE_OP", "1") != "0" and os.environ.get("ARAH_TRAIN_RAY_COMPACTION", "0") != "1"):
            # Round 6: no compaction of the RAYS.  The reference shades the rays that own a valid sample and leaves the others
            # at zero; the compositing op gives exactly that zero (colour and accumulation) for a ray without samples, so all
            # rays go down as they are: one device -> host round trip, seven gathers and two scatters less (~25 launches with
            # their backward).  The samples are still compacted once, inside.
            rgb_all, w_all = training.shade_composite_train(
                self, sdf_network, s_pts.reshape(B * N, *s_pts.shape[2:]), s_z.reshape(B * N, -1),
                s_T.reshape(B * N, *s_T.shape[2:]), s_mask.reshape(B * N, -1), dirs_in.reshape(B * N, 3),
                ray_dirs.reshape(B * N, 3), pose_cond, input["bone_transforms"][:1], cmin[:1], cmax[:1], center[:1],
                self.ray_tracer.n_steps, ray_augm=ray_augm, frame=frame, ws=self.ray_tracer.workspace(dev))
            self._late_bounds_check()
            out = {"rgb_values": rgb_all.reshape(B, N, 3), "sdf_output": w_all.reshape(B, N), "network_body_mask": vol_mask,
                   "body_mask": input["body_mask"], "off_surface_mask": vol_mask, "off_surface_sdf": uniform_sdf,
                   "grad_theta": grad_eik, "surface_normals": None}
            if pred_weights is not None:
                out["pred_weights"] = pred_weights
            if inside_sdf is not None:
                out["inside_sdf"] = inside_sdf
            return out
        vb, vr = vol_mask.nonzero(as_tuple=True)   # one compaction for the seven gathers and the two scatters
        rgb_hit, w_hit = training.shade_composite_train(
            self, sdf_network, s_pts[vb, vr], s_z[vb, vr], s_T[vb, vr], s_mask[vb, vr], dirs_in[vb, vr],
            ray_dirs[vb, vr], pose_cond, input["bone_transforms"][:1], cmin[:1], cmax[:1], center[:1],
            self.ray_tracer.n_steps, ray_augm=ray_augm, frame=frame if use_hip_shading else None,
            ws=self.ray_tracer.workspace(dev) if frame is not None else None)
        self._late_bounds_check()
        rgb = torch.zeros(B * N, 3, device=dev, dtype=xn.dtype).index_copy(0, vb * N + vr, rgb_hit).reshape(B, N, 3)
        acc = torch.zeros(B * N, device=dev).index_copy(0, vb * N + vr, w_hit.reshape(-1)).reshape(B, N)
        out = {"rgb_values": rgb, "sdf_output": acc, "network_body_mask": vol_mask, "body_mask": input["body_mask"],
               "off_surface_mask": vol_mask, "off_surface_sdf": uniform_sdf, "grad_theta": grad_eik,
               "surface_normals": None}
        if pred_weights is not None:
            out["pred_weights"] = pred_weights
        if inside_sdf is not None:
            out["inside_sdf"] = inside_sdf
        return out

    def _split_guard(self, ws, dev):
        """The split engine carries activations as f16 pairs; a network whose activations leave the f16 range makes loop C's
        first residual of a sample non-finite, the kernel counts those (ArahCounters.n_split_nonfinite) and retires the
        sample at its start state.  The count of frame k is copied to pinned host memory behind frame k (eight bytes, no
        stream drain) and looked at when frame k + 1 starts: if it grew, every later frame of this renderer is prepared
        for the exact fp32 engine, with a warning (`split_nonfinite` holds the total; a caller that needs frame k itself
        exact re-renders it)."""
        key = (dev, id(ws))   # one record per scratch = per stream with a frame in flight: each has its own counter
        g = self._guard.get(key)
        if g is not None and self.guard_mode != "strict" and g["event"].query():   # (strict: read behind the frame itself)
            ctr = g["host"].tolist()                                 # the nine counters of ArahCounters behind an earlier frame
            now = int(ctr[8])
            grew = now - g["seen"] if now >= g["seen"] else now      # the counters may have been reset in between
            g["seen"] = now
            d_col, d_den = ctr[4] - g["col"], ctr[6] - g["den"]      # n_col, n_density since the last look
            g["col"], g["den"] = ctr[4], ctr[6]
            win_full, g["win_full"] = g["win_full"], False
            if d_den > 0 and d_col >= 0 and not win_full:            # lazily shaded frame(s) went by: their share of sigma > 0 samples
                self.shade_ratio = d_col / d_den
                self._shade_full = self.shade_ratio > 0.7
            tier_now = (ctr[13], ctr[14], ctr[15])                   # n_tier_samples_p1 / _p2 / _skipped
            d_t = [a - b for a, b in zip(tier_now, g["tier"])]
            g["tier"] = tier_now
            if min(d_t) >= 0 and sum(d_t) > 0:                       # tiered frame(s) went by
                self.tier_share = d_t[2] / sum(d_t)
                self._tier_off = self.tier_share < 1.0 / 3.0
            if grew > 0:
                self.split_nonfinite += grew
                if self._precision != hip.PRECISION_FP32:
                    import warnings
                    warnings.warn("split-f16 engine: %d loop-C samples met activations outside the f16 range; following frames "
                                  "use the exact fp32 engine (ARAH_PRECISION=fp32)" % grew)
                    self._precision = hip.PRECISION_FP32
        if g is None:
            if len(self._guard) >= 8:
                self._guard.pop(next(iter(self._guard)))
            g = self._guard[key] = {"host": torch.zeros(hip.COUNTER_BYTES // 8, dtype=torch.int64).pin_memory(), "event": torch.cuda.Event(),
                                    "seen": 0, "col": 0, "den": 0, "win_full": False, "tier": (0, 0, 0)}
        return g

    def _split_guard_arm(self, g, ws, full=False):
        g["win_full"] = g["win_full"] or bool(full)   # a frame that shaded everything says nothing about the share
        # n_split_nonfinite is the ninth 64-bit counter at the head of the workspace (include/arah_hip.h: ArahCounters)
        g["host"].copy_(ws.buf[0:hip.COUNTER_BYTES].view(torch.int64), non_blocking=True)
        g["event"].record()

    def forward(self, input):
        if self.training:
            return self.forward_train(input)
        ray_dirs = input["ray_dirs"]
        cam_loc = input["cam_loc"]
        pose = input["pose"]
        nf = input["body_bounds_intersections"]
        B, N, _ = ray_dirs.shape
        if N == 0:
            raise ValueError("No valid depth.")
        dev = ray_dirs.device
        ws = self.ray_tracer.workspace(dev)
        guard = self._split_guard(ws, dev) if dev.type == "cuda" else None
        frame = build_frame(input["sdf_network"], self.skinning_model, self.rendering_network,
                            self.deviation_network, input["pose_cond"], input["smpl_verts"],
                            input["skinning_weights"], input["bone_transforms"], input["trans"],
                            input["coord_min"], input["coord_max"], input["center"],
                            precision=self._precision if self._precision is not None else self.precision,
                            body_tables=input.get("_body_tables"))
        self.last_frame = frame   # the gen_cano_mesh branch of the model entry meshes the same emitted network
        var = getattr(self.deviation_network, "variance", None)
        if var is not None:   # the measured shares (sigma > 0 samples, samples the tiers skip) belong to ONE beta
            vkey = (var.data_ptr(), var._version)
            if getattr(self, "_beta_key", vkey) != vkey:
                self._shade_full, self.shade_ratio, self._tier_off, self.tier_share = False, None, False, None
            self._beta_key = vkey
        full = self.ray_tracer.full_shading
        if not full and self.adaptive_shading and self._shade_full and guard is not None and self.guard_mode != "strict":
            self._shade_since_probe += 1
            full = self._shade_since_probe % 16 != 0       # every 16th frame lazily: refreshes the measured share
        samp = self.ray_tracer.sampling(dev, self.cano_view_dirs, self.render_last_pt, full_shading=full)
        pose34 = pose[0, :3, :4].detach().float().contiguous()
        tiered = self.tiering and not full
        if tiered and self.adaptive_shading and self._tier_off and guard is not None and self.guard_mode != "strict":
            self._tier_since_probe += 1
            tiered = self._tier_since_probe % 16 == 0          # every 16th frame tiered: refreshes the measured share
        rgb, pcam, vol, acc, dists, conv = hip.render(frame, ws, samp, cam_loc.reshape(B, 3),
                                                      ray_dirs.reshape(B * N, 3), nf.reshape(B * N, 2), pose34,
                                                      tiered=tiered)
        if guard is not None and self.guard_mode == "strict" and frame.precision != hip.PRECISION_FP32:
            now = int(ws.buf[64:72].view(torch.int64).item())   # ArahCounters.n_split_nonfinite; synchronises the stream
            grew = now - guard["seen"] if now >= guard["seen"] else now
            guard["seen"] = now
            if grew > 0:
                import warnings
                warnings.warn("split-f16 engine: %d loop-C samples met activations outside the f16 range; this frame is rendered "
                              "again on the exact fp32 engine, and so are the following ones" % grew)
                self.split_nonfinite += grew
                self._precision = hip.PRECISION_FP32
                frame = build_frame(input["sdf_network"], self.skinning_model, self.rendering_network,
                                    self.deviation_network, input["pose_cond"], input["smpl_verts"],
                                    input["skinning_weights"], input["bone_transforms"], input["trans"],
                                    input["coord_min"], input["coord_max"], input["center"],
                                    precision=hip.PRECISION_FP32, body_tables=input.get("_body_tables"))
                self.last_frame = frame
                rgb, pcam, vol, acc, dists, conv = hip.render(frame, ws, samp, cam_loc.reshape(B, 3),
                                                              ray_dirs.reshape(B * N, 3), nf.reshape(B * N, 2), pose34,
                                                              tiered=tiered)
        elif guard is not None and self.guard_mode != "strict":
            self._split_guard_arm(guard, ws, full)
        pcam = pcam.reshape(B, N, 3)
        if B > 1:   # per-view camera pose for the remaining batch elements (IDR:114-115)
            pw = cam_loc.reshape(B, 1, 3) + dists.reshape(B, N, 1) * ray_dirs
            pc = torch.matmul(pw, pose[:, :3, :3].transpose(1, 2)) + pose[:, :3, 3].unsqueeze(1)
            pcam = torch.where((pcam.abs().sum(-1, keepdim=True) > 0), pc, torch.zeros_like(pc))
        return {"points_cam": pcam, "network_body_mask": vol.bool().reshape(B, N), "rgb_values": rgb.reshape(B, N, 3)}


class MetaAvatarRender(nn.Module):
    """Model entry (reference models/__init__.py:17-354): hypernetwork -> SDF MLP, then the renderer."""

    def __init__(self, sdf_decoder=None, skinning_model=None, color_decoder=None, deviation_decoder=None,
                 train_cameras=False, train_smpl=False, train_latent_code=False, train_geo_latent_code=False,
                 cano_view_dirs=True, near_surface_samples=16, far_surface_samples=16, train_skinning_net=False,
                 render_last_pt=False, pose_input_noise=True, view_input_noise=True, nv_noise_type="rotation",
                 low_vram=False, n_steps=64, **kwargs):
        super().__init__()
        self.sdf_decoder = sdf_decoder
        self.skinning_model = skinning_model
        self.color_decoder = color_decoder
        self.deviation_decoder = deviation_decoder
        self.pose_input_noise = pose_input_noise
        self.view_input_noise = view_input_noise
        self.nv_noise_type = nv_noise_type
        ray_tracer = BodyRayTracing(root_finding_threshold=1e-5, n_steps=n_steps,
                                    near_surface_vol_samples=near_surface_samples,
                                    far_surface_vol_samples=far_surface_samples, sample_bg_pts=0, low_vram=low_vram)
        self.idhr_network = IDHRNetwork(deviation_decoder, color_decoder, skinning_model, ray_tracer,
                                        cano_view_dirs=cano_view_dirs, train_skinning_net=train_skinning_net,
                                        render_last_pt=render_last_pt, low_vram=low_vram)
        self.register_load_state_dict_post_hook(lambda module, incompatible: invalidate_caches(module))
        self.train_cameras = train_cameras
        if train_cameras:   # models/__init__.py:81-89: per-camera extrinsics (XYZW quaternion + translation) become parameters
            cam_trans, cam_rots = kwargs.get("cam_trans"), kwargs.get("cam_rots")
            assert cam_trans is not None and cam_rots is not None
            self.register_parameter("cam_trans", nn.Parameter(torch.as_tensor(np.asarray(cam_trans, np.float32))))
            self.register_parameter("cam_rots", nn.Parameter(torch.as_tensor(np.asarray(cam_rots, np.float32))))
        self.train_smpl = train_smpl
        if train_smpl:      # models/__init__.py:91-123: SMPL templates as buffers, per-frame poses / translation + betas as parameters
            from . import smpl
            body = kwargs.get("body_model")   # reference: np.load('body_models/misc/*.npz')[gender] (licence-gated files)
            if body is None:
                body = smpl.BodyModel.from_files(kwargs.get("gender"))
            f32 = lambda a: torch.as_tensor(np.asarray(a, np.float32))
            self.register_buffer("v_template", f32(body.v_template).unsqueeze(0))
            self.register_buffer("posedirs", f32(body.posedirs))
            self.register_buffer("shapedirs", f32(body.shapedirs))
            self.register_buffer("J_regressor", f32(body.J_regressor))
            self.register_buffer("lbs_weights", f32(body.lbs_weights))
            self.register_buffer("kintree_table", torch.as_tensor(np.asarray(body.kintree_table, np.int32)))
            frames = kwargs.get("frames")
            self.frames = frames
            params = {}
            for key in ("root_orient", "pose_body", "pose_hand", "trans"):
                vals = kwargs.get(key)
                params.update({"%s_%s" % (key, fr): nn.Parameter(f32(vals[i])) for i, fr in enumerate(frames)})
            self.body_poses = nn.ParameterDict(params)
            self.register_parameter("betas", nn.Parameter(f32(kwargs.get("betas"))))
        self.train_latent_code = train_latent_code
        self.train_geo_latent_code = train_geo_latent_code
        if train_latent_code or train_geo_latent_code:
            self.latent = nn.Embedding(kwargs.get("n_data_points"), 128)
            if not train_smpl:
                self.frames = kwargs.get("frames")

    def forward_smpl(self, betas, root_orient, pose_body, pose_hand, trans):
        """SMPL linear blend skinning of the template (models/__init__.py:317-339):
        -> verts_posed (B,V,3) without translation, Jtrs (rest), Jtrs_posed, bone_transforms (B,24,4,4), minimal_shape."""
        from . import smpl
        full_pose = torch.cat([root_orient, pose_body, pose_hand], dim=-1)
        verts, J_posed, J, A, _, v_posed = smpl.lbs(betas, full_pose, self.v_template, self.shapedirs, self.posedirs,
                                                    self.J_regressor, self.kintree_table[0].long(), self.lbs_weights)
        return verts, J, J_posed, A, v_posed

    def network_parameters(self):
        for name, param in self.named_parameters():
            if name not in ["cam_rots", "cam_trans", "root_orient", "pose_body", "pose_hand", "trans", "betas"]:
                yield param

    def camera_parameters(self):
        for name, param in self.named_parameters():
            if name in ["cam_rots", "cam_trans"]:
                yield param

    def smpl_parameters(self):
        for name, param in self.named_parameters():
            if name.startswith(("body_poses", "betas")):
                yield param

    def forward(self, inputs, gen_cano_mesh=False, eval=False):
        rots, Jtrs = inputs["rots"], inputs["Jtrs"]
        B, dev = rots.size(0), rots.device
        decoder_input = {"coords": torch.zeros(1, 1, 3, dtype=torch.float32, device=dev),
                         "rots": rots[0].unsqueeze(0), "Jtrs": Jtrs[0].unsqueeze(0)}
        if "geo_latent_code_idx" in inputs:
            decoder_input["latent"] = self.latent(inputs["geo_latent_code_idx"])
        if dev.type == "cuda" and os.environ.get("ARAH_EARLY_BODY_TABLES", "1") != "0":
            # the nearest-vertex tables need only the posed body: build them now, on a side stream, next to the pose
            # encoder and the hypernetwork (hip.BodyTables)
            inputs["_body_tables"] = hip.BodyTables(inputs["smpl_verts"][0])
        if (self.pose_input_noise or self.view_input_noise) and not eval:
            if np.random.uniform() <= 0.5:   # models/__init__.py:157-174
                if self.pose_input_noise:
                    decoder_input["rots_noise"] = torch.normal(mean=0, std=0.1, size=rots.shape, dtype=rots.dtype, device=dev)
                    inputs["pose_cond"]["rot_noise"] = torch.normal(mean=0, std=0.1, size=(B, 9), dtype=rots.dtype, device=dev)
                    inputs["pose_cond"]["trans_noise"] = torch.normal(mean=0, std=0.1, size=(B, 3), dtype=rots.dtype, device=dev)
                if self.view_input_noise:
                    if self.nv_noise_type == "gaussian":
                        inputs["pose_cond"]["view_noise"] = torch.normal(mean=0, std=0.1, size=inputs["ray_dirs"].shape,
                                                                          dtype=rots.dtype, device=dev)
                    elif self.nv_noise_type == "rotation":
                        inputs["pose_cond"]["view_noise"] = torch.tensor(training.augm_rots(45, 45, 45), dtype=torch.float32,
                                                                          device=dev).unsqueeze(0)
                    else:
                        raise ValueError("wrong nv_noise_type, expected either gaussian or rotation, got %s"
                                         % self.nv_noise_type)
        if (eval and dev.type == "cuda" and not torch.is_grad_enabled() and "rots_noise" not in decoder_input
                and "latent" in decoder_input and os.environ.get("ARAH_HYPERNET_GRAPH", "1") != "0"):
            graphed = _GRAPHED.setdefault(self, {}).get("eval")
            if graphed is None or graphed.decoder is not self.sdf_decoder:
                graphed = _GRAPHED[self]["eval"] = _GraphedDecoder(self.sdf_decoder)
            try:
                out = graphed(decoder_input) if not graphed.broken else self.sdf_decoder(decoder_input)
            except RuntimeError as err:   # a capture the runtime refuses: the same launches, eagerly, from here on
                import warnings
                warnings.warn("hypernetwork graph capture failed (%s); this model runs the eager call from now on" % err)
                graphed.broken = True
                out = self.sdf_decoder(decoder_input)
        elif (not eval and dev.type == "cuda" and torch.is_grad_enabled() and "latent" in decoder_input
              and not rots.requires_grad and not Jtrs.requires_grad and decoder_input["latent"].requires_grad
              and os.environ.get("ARAH_TRAIN_HYPERNET_GRAPH", "0") == "1"):
            tg = _GRAPHED.setdefault(self, {}).get("train")
            if tg is None or tg.decoder is not self.sdf_decoder:
                tg = _GRAPHED[self]["train"] = _TrainGraphedDecoder(self.sdf_decoder)
            try:
                out = tg(decoder_input) if not tg.broken else self.sdf_decoder(decoder_input)
            except RuntimeError as err:   # a capture the runtime refuses: the eager call from here on
                import warnings
                warnings.warn("training hypernetwork graph capture failed (%s); this model runs the eager call from now on" % err)
                tg.broken = True
                out = self.sdf_decoder(decoder_input)
        else:
            if not eval and torch.is_grad_enabled():
                decoder_input["skip_model_out"] = True     # the value of the emitted network at the dummy point is never used
            out = self.sdf_decoder(decoder_input)
        inputs.update({"loc": torch.zeros(B, 1, 3, device=dev), "sc_factor": torch.ones(B, 1, 1, device=dev),
                       "vol_feat": torch.empty(B, 0, device=dev), "sdf_network": out["decoder"]})
        if "latent_code_idx" in inputs["pose_cond"]:
            inputs["pose_cond"]["latent_code"] = self.latent(inputs["pose_cond"]["latent_code_idx"])
        try:
            model_outputs = self.idhr_network(inputs)
        finally:
            inputs.pop("_body_tables", None)   # they belong to this call's vertices, not to the caller's dict
        model_outputs.update({"sdf_params": out["params"]})
        if gen_cano_mesh:   # models/__init__.py:203-311: canonical mesh + the three normal maps, all on the device
            from . import meshing
            frame = getattr(self.idhr_network, "last_frame", None)
            if frame is None:   # training-mode forward: pack the emitted network once for the meshing kernels
                frame = build_frame(inputs["sdf_network"], self.skinning_model, None, None, None, inputs["smpl_verts"],
                                    inputs["skinning_weights"], inputs["bone_transforms"], inputs["trans"],
                                    inputs["coord_min"], inputs["coord_max"], inputs["center"])
            maps, _ = meshing.canonical_mesh_outputs(frame, self.idhr_network.ray_tracer.workspace(dev), inputs, want_tri=False)
            model_outputs.update(maps)
        return model_outputs


def _walk_tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _walk_tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _walk_tensors(v)


_FREE_STREAMS = {}   # streams of map_in_flight calls without an owner, per (device, count)


def frames_in_flight(n_frames):
    """How many frames of a sequence render_sequence keeps in flight by default: every further frame hides more of the
    kernels' tails and of the latency-bound finishers under other frames' wide kernels, but a short sequence must still fill the
    streams evenly.  Round 4 (untiered frames of 35 ms): 36.4 / 35.4 / 34.7 / 34.3 ms per frame with 3 / 4 / 5 / 6 in flight on
    20-frame passes, 36.2 / 34.9 / 35.4 with 3 / 4 / 5 on eight frames -- five from fifteen frames on, four below.  Round 6
    (tiered frames of 13 ms, whose kernels are short): 20-frame passes 12.1 / 12.9 / 12.3 / 12.2 ms with 4 / 5 / 6 / 8 in flight
    (three alternations on one box, within 0.05 ms of each other; why five is the slow one was not looked into), eight-frame
    passes 12.9 / 13.0 / 13.0 / 12.8 / 13.1 with 3 / 4 / 5 / 6 / 8: FOUR for every length.  Soaks with 4, 5 and 6 in flight came
    back clean and bit-identical (tools/stress_streams.py, profiles/r04d_streams_soak.txt, r06e_streams_soak.txt)."""
    return max(1, min(4, n_frames))


def render_sequence(model, frames, n_streams=None, **forward_kwargs):
    """Render independent frames (a test sequence, reference test.py / lightning_model.py:320) with `n_streams` of them in
    flight: frame k runs on HIP stream k mod n_streams with its own scratch, so that the latency-bound stretches of one
    frame (the tails of sphere tracing and of the joint root find: a few hundred live rays, ~60 us of kernel latency per
    step) run under the other frames' wide kernels.  Per-frame results are bit-identical to one-at-a-time rendering
    (tests/test_zz_render_sequence.py); ~46 -> ~40 ms per 512x512 frame on one MI355X.  n_streams=1: one frame at a time
    on the caller's stream.
    Round 2 saw the HIP runtime of this image (ROCm 7.2) stop accepting launches with a thousand launches queued behind
    a cross-stream wait; the loop below never lets that state arise (the caller's stream is drained once at the start,
    a stream takes its next frame only when its previous one has finished: at most n_streams frames, ~270 launches
    each, are ever queued), and three frames in flight became the default after 200 consecutive eight-frame passes
    (1600 frames, bit-identical images) came back clean (tools/stress_streams.py, profiles/r03_streams_soak.txt);
    round 4: n_streams=None = frames_in_flight(len(frames)), four or five.
    frames: iterable of input dicts (resident on one GPU); returns the list of output dicts, usable on the caller's
    current stream.  model(inputs, **forward_kwargs) is called under torch.no_grad()."""
    return map_in_flight(lambda f: model(f, **forward_kwargs), frames, n_streams=n_streams, owner=model)


def map_in_flight(fn, items, n_streams=None, owner=None):
    """[fn(item) for item in items] with `n_streams` of the calls in flight: call k runs on HIP stream k mod n_streams (the
    renderer's scratch is per stream), under torch.no_grad().  The loop of render_sequence, for any per-frame function --
    the test sequence's test_step (render + canonical mesh + normal maps) goes through it too.  `owner`: the object the
    streams are cached on."""
    items = list(items)
    if not items:
        return []
    dev = next((t.device for t in _walk_tensors(items[0]) if t.is_cuda), None)
    n_streams = frames_in_flight(len(items)) if n_streams is None else max(1, int(n_streams))
    if n_streams == 1 or dev is None:   # (host-resident inputs: the model raises for want of a GPU, as it always does)
        with torch.no_grad():
            return [fn(f) for f in items]
    cache = owner.__dict__.setdefault("_sequence_streams", {}) if owner is not None else _FREE_STREAMS
    if (dev, n_streams) not in cache:
        cache[(dev, n_streams)] = [torch.cuda.Stream(dev) for _ in range(n_streams)]
    streams = cache[(dev, n_streams)]
    cur = torch.cuda.current_stream(dev)
    # The caller's stream is drained once, here, and a stream takes its next frame only when its previous one has
    # finished: at most n_streams frames (~250 launches each) are ever queued.  Not a nicety -- with the caller's stream
    # still busy and a thousand launches queued behind a cross-stream wait, the HIP runtime (ROCm 7.2) stops accepting
    # launches and never resumes (the host blocks inside hipLaunchKernel; reproduced by tools/probes/seq_debug.py "prenosync").
    cur.synchronize()
    outs, done = [], []
    with torch.no_grad():
        for k, inp in enumerate(items):
            st = streams[k % n_streams]
            if k >= n_streams:
                done[k - n_streams].synchronize()
            for t in _walk_tensors(inp):
                if t.is_cuda:
                    t.record_stream(st)
            with torch.cuda.stream(st):
                outs.append(fn(inp))
                ev = torch.cuda.Event()
                ev.record(st)
                done.append(ev)
    for st in streams:
        cur.wait_stream(st)
    for out in outs:
        for t in _walk_tensors(out):
            if t.is_cuda:
                t.record_stream(cur)
    return outs
