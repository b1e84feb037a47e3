"""Probe: which fraction of the valid samples of a frame sits behind the point where the ray's transmittance is exactly
0.0f (their compositing weight is exactly zero whatever they evaluate to)?"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from arah_release_amd import config, hip, synthetic, training

dev = torch.device("cuda:0")
scene = synthetic.SyntheticScene(0)
model, cfg = config.build_synthetic_model("zju377_mono", device=dev)
idhr = model.idhr_network
for f in (2, 5, 7):
    inp = scene.make_inputs(512, 512, frame_idx=f, device=dev)
    with torch.no_grad():
        model(inp, eval=True)
        frame = idhr.last_frame
        ws = idhr.ray_tracer.workspace(dev)
        samp = idhr.ray_tracer.sampling(dev, idhr.cano_view_dirs, idhr.render_last_pt)
        cam, d, nf = inp["cam_loc"].reshape(1, 3), inp["ray_dirs"][0], inp["body_bounds_intersections"][0]
        xn, T, conv, start, end = hip.trace(frame, ws, cam, d, nf)
        z, pts, Ts, mask = hip.sample_canonicalize(frame, ws, samp, cam, d, nf, conv, start, end, None)
        N, S = d.shape[0], 64
        z = z.reshape(N, S); mask = mask.reshape(N, S).bool(); pts = pts.reshape(N, S, 3)
        cmin, cmax, cen = inp["coord_min"].reshape(()), inp["coord_max"].reshape(()), inp["center"].reshape(1, 1, 3)
        xnorm = training.normalize_canonical_points(pts, cmin, cmax, cen)
        sdf = hip.sdf_eval(frame, ws, xnorm.reshape(-1, 3).contiguous())
        sdf = sdf[0] if isinstance(sdf, (tuple, list)) else sdf
        sdf = sdf.reshape(N, S) / 2.0 * 1.1 * (cmax - cmin)
        beta = idhr.deviation_network(sdf.reshape(-1, 1)).clip(1e-6, 1e6).reshape(N, S) if False else torch.linalg.norm(idhr.deviation_network.variance).clip(1e-6, 1e6)
        ib = 1.0 / beta
        dens = torch.relu(ib * (0.5 + 0.5 * torch.sign(-sdf) * (1 - torch.exp(-sdf.abs() * ib))))
        # left-packed compositing per ray, vectorised: sort valid first keeping order
        big = torch.where(mask, z, torch.full_like(z, 1e30))
        order = torch.argsort(big, dim=1, stable=True)
        zs = torch.gather(z, 1, order); ms = torch.gather(mask, 1, order); ds = torch.gather(dens, 1, order)
        cnt = ms.sum(1)
        delta = torch.cat([zs[:, 1:] - zs[:, :-1], torch.full((N, 1), 1.0 / S, device=dev)], 1)
        idx = torch.arange(S, device=dev)[None]
        delta = torch.where(idx == (cnt[:, None] - 1), torch.full_like(delta, 1.0 / S), delta)
        alpha = 1 - torch.exp(-ds * delta)
        fac = torch.where(ms, 1 - alpha + 1e-7, torch.ones_like(alpha))
        trans = torch.cumprod(torch.cat([torch.ones(N, 1, device=dev), fac[:, :-1]], 1), 1)
        dead = ms & (trans == 0)
        valid = int(ms.sum())
        print("frame %d: beta %.2e, rays %d, valid samples %d (%.1f/ray), behind exact-zero transmittance %d (%.1f %%), sigma>0 %d (%.1f %%), "
              "rays that reach T==0: %.1f %%; first dead slot (mean over those rays) %.1f of %.1f valid"
              % (f, float(beta), N, valid, valid / N, int(dead.sum()), 100.0 * int(dead.sum()) / valid,
                 int((ms & (ds > 0)).sum()), 100.0 * int((ms & (ds > 0)).sum()) / valid,
                 100.0 * float((dead.any(1)).float().mean()),
                 float(torch.where(dead.any(1), dead.float().argmax(1).float(), torch.zeros(N, device=dev)).sum() / dead.any(1).sum().clamp(min=1)),
                 float(cnt[dead.any(1)].float().mean())))
