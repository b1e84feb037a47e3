"""Headline benchmark: rendered rays/s of the articulated-SDF volume renderer (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one ``MetaAvatarRender.forward(inputs, eval=True)`` (per-frame hypernetwork included,
gen_cano_mesh off, data generation excluded: the input dict is resident in HBM before the timed
region) on one synthetic 512x512 frame of the ZJUMOCAP-377-mono configuration with 64 samples/ray
(BASELINE.json configs[1]).  Frames are independent, so N GPUs render disjoint frames (frame
i -> rank i mod N): no data-path collective, weak scaling, value = all rays of all ranks
/ max-over-ranks time.

Extra objects on the JSON line:
  roofline      largest single launch of the default path: k_canon_wave, loop C (every Broyden iteration of every
                valid sample in one resident kernel).  Algorithmic MFMA flops = skinning-MLP evaluations x 105 472
                / HIP-event duration of the launch.  Peak: the default GEMM engine carries each fp32 operand as two
                f16 and spends three v_mfma_f32_16x16x32_f16 per product, so its ceiling in algorithmic fp32 flops is
                the dense f16 MFMA peak / 3 = 833 TFLOP/s; the exact engine (ARAH_PRECISION=fp32) is priced against
                the fp32 MFMA peak, 157.3 TFLOP/s (MI355X_MICROARCH.md).
  roofline_k_density  the same for the second largest launch (round 1's dominant kernel): the SDF MLP forward on
                every valid sample.
  exact_fp32_engine  the same frames with v_mfma_f32_16x16x4_f32 everywhere.
  parity_vs_reference_frame  frame 0 of the workload, every ray, against the reference's OWN render of it (a committed fixture:
                tests/golden/make_golden.py f7full ran taconite/arah-release on the CPU in the build container).
  beta_sweep    the same frames with the VolSDF beta overridden (what exact lazy shading saves depends on it).
  cpu_baseline  the oracle (a torch-CPU restatement of the reference, pinned against it) on a bounded
                sample of the same frame's rays, on the host cores of this box; rank 0, N == 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F_SDF = 657408          # 2*(3*256 + 5*256^2 + 256)
F_SDF_GRAD = 657408
F_COL = {"no_view_dir": 794112, "idr": 821760}
F_SKIN = 105472         # 2*(3*128 + 3*128^2 + 128*25), one skinning-MLP evaluation
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
PEAK_SPLIT_TFLOPS = PEAK_F16_MFMA_TFLOPS / 3.0   # three f16 MFMAs per fp32 product
# what a loop of nothing but v_mfma_f32_16x16x32_f16 on random data sustains on an MI355X under its power budget, two waves per SIMD
# (tools/ubench/mfma_shape.hip, profiles/r05_mfma_shape.txt: 1968 TFLOP/s at 2.0 GHz; one wave per SIMD 1693) -- context for
# `frac`, which stays priced against the nominal peak
MEASURED_F16_MFMA_TFLOPS = 1968.0


def mixed_peak(parts):
    """Time-weighted peak of a kernel whose GEMM classes run on different engines: parts = [(flops, peak), ...]."""
    return sum(f for f, _ in parts) / sum(f / p for f, p in parts)


def shard_frames(rank, world, steps, warmup):
    """Frame indices of this rank: (warmup frames, timed frames). Frame i belongs to rank i mod world."""
    per_rank = warmup + steps
    mine = [rank + world * k for k in range(per_rank)]
    return mine[:warmup], mine[warmup:]


def aggregate(local_rays, local_seconds, dist=None):
    """Whole-job rays and the max-over-ranks time."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_rays), float(local_seconds)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    r = torch.tensor([float(local_rays)], dtype=torch.float64, device=dev)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(r, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(r.item()), float(t.item())


def pmc_traffic(kernel, full_shading=False):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json,
    made by tools/rocpd_pmc.py from separate FETCH_SIZE / WRITE_SIZE passes of this same command; ..._pmc_traffic_full.json
    for the passes run with ARAH_FULL_SHADING=1); None if absent.
    PMC counters cannot be read from inside the process, so this is the one figure that is not measured live."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_full.json" if full_shading else "r*_pmc_traffic.json")))
    files = [f for f in files if full_shading or not f.endswith("_full.json")]
    if not files:
        return None, None
    try:
        ks = json.load(open(files[-1]))["kernels"]
        k = ks.get(kernel) or next((v for n, v in sorted(ks.items()) if n.startswith(kernel)), None)
        return (k["hbm_bytes_avg"], os.path.relpath(files[-1], ROOT)) if k else (None, None)
    except Exception:
        return None, None


def pmc_db_per_kernel(path, counter):
    """{kernel name without namespace and arguments: (average bytes per launch, launches)} of one counter (FETCH_SIZE or
    WRITE_SIZE, both in KiB) in a rocprofv3 rocpd database -- the reduction of tools/rocpd_pmc.py."""
    import re, sqlite3
    db = sqlite3.connect(path)
    try:
        rows = db.execute("select kernel_name, count(*), avg(value) from counters_collection where counter_name=? "
                          "group by kernel_name", (counter,)).fetchall()
    finally:
        db.close()
    out = {}
    for name, n, avg in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short).replace("void ", "")
        out[short] = (avg * 1024.0, n)
    return out


def live_traffic(args, kernels, limit_s=150.0):
    """HBM bytes per launch of `kernels` (name prefixes) measured on THIS box: two child runs of this file under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE, then WRITE_SIZE: separate passes, no other tracing domain), one
    warm-up + one timed + one event-timed frame of the default path each, reduced as tools/rocpd_pmc.py does it
    (MI355X_MICROARCH.md, HBM section: both counters in KiB; FETCH_SIZE doubled on gfx950, WRITE_SIZE as is).  Returns
    ({prefix: bytes}, note) or (None, why)."""
    import glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_TOOL", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself under a profiler: no nested rocprofv3 passes"
    tmp = tempfile.mkdtemp(prefix="arah_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "1", "--streams", "1", "--no-cpu-baseline",
             "--no-gpu-baseline", "--no-train", "--passes", "default", "--no-live-traffic", "--size", str(args.size), "--n-steps", str(args.n_steps),
             "--config", args.config]
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--"] + child, cwd="/tmp",
                               env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=limit_s)
            dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True), key=os.path.getsize)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s pass failed (rc %d)" % (counter, r.returncode)
            for short, v in pmc_db_per_kernel(dbs[-1], counter).items():
                per.setdefault(short, {})[counter] = v
    except Exception as e:   # a profiler that hangs or a database that is not there: the committed figure stays
        return None, "live PMC passes failed: %s" % e
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for prefix in kernels:
        k = next((v for n, v in sorted(per.items()) if n.startswith(prefix)), None)
        if k and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            out[prefix] = 2.0 * k["FETCH_SIZE"][0] + k["WRITE_SIZE"][0]
    return out, ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, two child runs of bench.py (--steps 1 --warmup 1 --streams 1 "
                 "--passes default) spawned by this run on this box; FETCH_SIZE x 2 (gfx950), KiB -> bytes")


def _cpu_worker_init(cfg_name, n_steps, near, far, threads):
    # one process of the multi-process CPU baseline: its own model, its own thread pool
    global _CPU_WORKER
    torch.set_num_threads(threads)
    from arah_release_amd import config, synthetic
    from oracle import arah_oracle as O
    O.KDTREE_WORKERS = threads          # the 1-NN search's own pool: this process's share of the cores, not all of them
    model, cfg = config.build_synthetic_model(cfg_name, n_steps, near, far, device="cpu")
    _CPU_WORKER = (model, cfg, synthetic.SyntheticScene(0), n_steps, near, far)


def _slice_rays(inputs, lo, hi):
    n = inputs["ray_dirs"].shape[1]
    out = {}
    for k, v in inputs.items():
        if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == n and k not in ("smpl_verts", "skinning_weights", "minimal_shape"):
            out[k] = v[:, lo:hi].contiguous()
        else:
            out[k] = v
    return out


def _cpu_worker_render(job):
    size, sample_rays, lo, hi = job
    from oracle import arah_oracle as O
    model, cfg, scene, n_steps, near, far = _CPU_WORKER
    inputs = _slice_rays(scene.make_inputs(size, size, frame_idx=0, max_rays=sample_rays), lo, hi)
    t0 = time.perf_counter()
    ref = O.render_inputs(model, inputs, cfg["model"]["cano_view_dirs"], n_steps, near, far)
    return (lo, hi, ref["rgb_values"].numpy(), ref["network_body_mask"].numpy(), dict(ref["frame"].counters), time.perf_counter() - t0)


def cpu_baseline_multiprocess(scene, cfg_name, size, n_steps, near, far, sample_rays, model_gpu, dev, threads_per_worker=4):
    """The oracle on ALL host cores: the oracle is a Python loop around torch operators on shrinking active sets and does not
    scale past a handful of threads (round 5: 208 rays/s on 128 threads against the reference's own 348 rays/s on 8), but rays are
    independent -- worker processes render disjoint slices of the same evenly subsampled rays of frame 0.  The split was
    measured on the MI355X box's host (256 logical cores, 16 384 rays, idle OpenMP threads sleeping): 16 x 8 threads 2 226 rays/s,
    32 x 4 2 661, 64 x 2 1 990, 128 x 1 1 122, 64 x 4 1 709 (and 32 x 8 with spinning OpenMP threads: 912) -- min(32, cores / 8)
    processes of 4 threads.  Timed: the pool's map over the slices, after every worker has built its model and rendered a
    warm-up slice."""
    import multiprocessing as mp
    import numpy as np
    cores = os.cpu_count() or 8
    workers = int(os.environ.get("ARAH_CPU_BASELINE_WORKERS", "0")) or max(1, min(32, cores // 8))
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # inherited by the workers: idle OpenMP threads sleep instead of spinning
    n = int(scene.make_inputs(size, size, frame_idx=0, max_rays=sample_rays)["ray_dirs"].shape[1])
    per = (n + workers - 1) // workers
    jobs = [(size, sample_rays, lo, min(n, lo + per)) for lo in range(0, n, per)]
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers, initializer=_cpu_worker_init, initargs=(cfg_name, n_steps, near, far, threads_per_worker)) as pool:
        pool.map(_cpu_worker_render, [(size, sample_rays, 0, 64)] * workers)          # warm-up: every worker is up
        t0 = time.perf_counter()
        parts = pool.map(_cpu_worker_render, jobs)
        dt = time.perf_counter() - t0
    rgb_ref = np.zeros((n, 3), np.float64)
    mask_ref = np.zeros(n, bool)
    work = {}
    for lo, hi, rgb, mask, ctr, _ in parts:
        rgb_ref[lo:hi], mask_ref[lo:hi] = rgb, mask
        for k, v in ctr.items():
            work[k] = work.get(k, 0) + v
    out = {"value": n / dt, "unit": "rays/s", "cores": workers * threads_per_worker, "kind": "port",
           "sample": "%d rays evenly subsampled from frame 0 of the %dx%dx%d workload, oracle/arah_oracle.py (torch CPU fp32, cKDTree "
                     "1-NN) in %d processes x %d threads on disjoint ray slices, %.1f s (slowest slice %.1f s)"
                     % (n, size, size, n_steps, workers, threads_per_worker, dt, max(p[5] for p in parts)),
           "work": {"per_ray": {k: v / max(n, 1) for k, v in work.items()}}}
    if model_gpu is not None:
        with torch.no_grad():
            got = model_gpu(scene.make_inputs(size, size, frame_idx=0, max_rays=sample_rays, device=dev), eval=True)
        rgb = got["rgb_values"][0].double().cpu().numpy()
        mask = got["network_body_mask"][0].cpu().numpy()
        mse = float(np.mean((rgb - rgb_ref) ** 2))
        out["psnr_vs_oracle_db"] = None if mse == 0 else -10.0 * float(np.log10(mse))
        out["mask_agreement"] = float((mask == mask_ref).mean())
        out["parity_sample"] = "HIP render of the same %d rays vs the oracle's image" % n
    return out


def cpu_baseline(scene, cfg_name, size, n_steps, near, far, sample_rays, model_gpu=None, dev=None):
    """Oracle on `sample_rays` rays spread evenly over frame 0 of the benchmark workload, timed on the host cores;
    the same rays are rendered by the HIP path and compared (BASELINE.json's "+ PSNR vs ref": the oracle is the
    pinned CPU restatement of the reference, the reference itself does not travel to this box)."""
    import numpy as np
    from arah_release_amd import config
    from oracle import arah_oracle as O
    model, cfg = config.build_synthetic_model(cfg_name, n_steps, near, far, device="cpu")
    inputs = scene.make_inputs(size, size, frame_idx=0, max_rays=sample_rays)
    n = inputs["ray_dirs"].shape[1]
    t0 = time.perf_counter()
    ref = O.render_inputs(model, inputs, cfg["model"]["cano_view_dirs"], n_steps, near, far)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d rays evenly subsampled from frame 0 of the %dx%dx%d workload, oracle/arah_oracle.py "
                     "(torch CPU fp32, cKDTree 1-NN), %.1f s" % (n, size, size, n_steps, dt),
           # the oracle's own work counters per ray (tests/test_hip_parity.py::test_work_counters_against_oracle
           # holds the kernels' counters against them): the reference shades every valid sample and evaluates the start
           # point of every loop-C sample twice, so n_col / n_sdf_grad / n_skin_fwd are larger than the default path's
           "work": {"per_ray": {k: v / max(n, 1) for k, v in ref["frame"].counters.items()}}}
    if model_gpu is not None:
        with torch.no_grad():
            got = model_gpu(scene.make_inputs(size, size, frame_idx=0, max_rays=sample_rays, device=dev), eval=True)
        rgb = got["rgb_values"][0].double().cpu().numpy()
        mask = got["network_body_mask"][0].cpu().numpy()
        rgb_ref = ref["rgb_values"].double().numpy()
        mask_ref = ref["network_body_mask"].numpy()
        mse = float(np.mean((rgb - rgb_ref) ** 2))
        out["psnr_vs_oracle_db"] = None if mse == 0 else -10.0 * float(np.log10(mse))   # im2mesh/utils/eval.py:6-9
        out["mask_agreement"] = float((mask == mask_ref).mean())
        out["parity_sample"] = "HIP render of the same %d rays vs the oracle's image" % n
    return out


def gpu_torch_baseline(scene, cfg_name, size, n_steps, near, far, sample_rays, model_gpu, dev):
    """The stand-in for "the reference's single-GPU rays/s" (BASELINE.json: ">= 10x the reference single-GPU
    rays/sec"): the reference's op sequence -- the oracle, pinned against the reference's own outputs -- on PyTorch-ROCm tensors
    on THIS GPU, with the reference's chunk sizes and its per-batch host round trips of eval_sdf (oracle/gpu_standin.py), on
    `sample_rays` rays spread evenly over frame 0 of the workload; the HIP path renders the same rays for the parity figures.
    Outside the timed region of `value`, like cpu_baseline."""
    import numpy as np
    from arah_release_amd import config
    from oracle import gpu_standin as G
    cfg = config.builtin_config(cfg_name, n_steps, near, far)
    cvd = cfg["model"]["cano_view_dirs"]
    mk = lambda n: scene.make_inputs(size, size, frame_idx=0, max_rays=n if n else None, device=dev)   # noqa: E731
    ref, dt, n = G.timed(model_gpu, mk, cvd, sample_rays, 2048, n_steps, near, far)
    with torch.no_grad():
        got = model_gpu(mk(sample_rays), eval=True)
    rgb = got["rgb_values"][0].double().cpu().numpy()
    mask = got["network_body_mask"][0].cpu().numpy()
    rgb_ref = ref["rgb_values"].double().cpu().numpy()
    mask_ref = ref["network_body_mask"].cpu().numpy()
    mse = float(np.mean((rgb - rgb_ref) ** 2))
    return {"value": n / dt, "unit": "rays/s", "kind": "stand-in (BASELINE.md section 2): the reference's op sequence on PyTorch-ROCm on this "
                                                       "GPU -- oracle/arah_oracle.py on cuda tensors, exact brute-force 1-NN on the device for "
                                                       "pytorch3d.knn_points, eval_sdf in 1e5-point batches with a host round trip each "
                                                       "(root_finding_utils.py:116-144), loop C in 1e6-point chunks, loop D in 20 480-ray chunks; "
                                                       "the reference itself publishes no rays/s and cannot run here",
            "sample": "%d rays %s frame 0 of the %dx%dx%d workload, one render after a 2048-ray warm-up, %.2f s"
                      % (n, "evenly subsampled from" if sample_rays else "= every ray of", size, size, n_steps, dt),
            "seconds": dt, "torch": torch.__version__,
            "psnr_hip_vs_standin_db": None if mse == 0 else -10.0 * float(np.log10(mse)),
            "mask_agreement_hip_vs_standin": float((mask == mask_ref).mean())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--n-steps", type=int, default=64)
    ap.add_argument("--config", default="zju377_mono")
    ap.add_argument("--cpu-sample-rays", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gpu-baseline-rays", type=int, default=0,
                    help="rays of frame 0 the PyTorch-ROCm stand-in of the reference's GPU path renders (vs_baseline's denominator); "
                         "0 = the whole frame (the reference's own batch sizes then see a full frame's points)")
    ap.add_argument("--no-gpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step line (configs[2], one GPU)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not spawn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic on "
                         "THIS box; the figure then comes from the newest committed profiles/*_pmc_traffic.json")
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight in the timed region of every pass (renderer.render_sequence: frame k on HIP stream "
                         "k mod N, own scratch each); 0 = the product's default for a sequence of --steps frames "
                         "(renderer.frames_in_flight: four; soaks in profiles/r03_streams_soak.txt, "
                         "r04d_streams_soak.txt); 1 = strictly one frame after the other (also measured, first, and reported "
                         "as 'one_frame_at_a_time')")
    ap.add_argument("--pipelined-streams", type=int, default=0,
                    help="with --streams 1 only: after everything else, the default path once more with this many frames in "
                         "flight, under a watchdog (object 'frames_in_flight'); 0 = skip")
    ap.add_argument("--beta", type=float, default=None,
                    help="override the VolSDF beta of the synthetic subject (|deviation_decoder.variance|, default 1e-3 = the "
                         "reference's initial value): the share of samples with density > 0, hence what exact lazy shading "
                         "saves, depends on it")
    ap.add_argument("--passes", default="all", choices=["all", "default"],
                    help="'default': only the product's default path (profiling runs); 'all' adds the full-shading, "
                         "exact-engine and strict passes over the same frames")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    assert world == max(args.gpus, 1), "launch with --nproc-per-node == --gpus"

    import __graft_entry__
    if rank == 0:
        __graft_entry__.build()
    if world > 1:
        dist.barrier()
    from arah_release_amd import config, hip, renderer, synthetic

    if args.streams <= 0:
        args.streams = renderer.frames_in_flight(args.steps)
    near = far = args.n_steps // 4
    model, cfg = config.build_synthetic_model(args.config, args.n_steps, near, far, device=dev)
    if args.beta is not None:
        with torch.no_grad():
            model.deviation_decoder.variance.fill_(args.beta)
    rt = GpuRuntime(world, rank, dev, dist if world > 1 else None, model, cfg, synthetic.SyntheticScene(0), hip)
    rt.partial_line = None
    if world == 1:   # last line of defence: a stalled runtime must not cost the whole measurement
        import threading

        def bail():
            if rt.partial_line is not None:
                print(json.dumps(rt.partial_line), flush=True)
            os._exit(0 if rt.partial_line is not None else 3)

        global_dog = threading.Timer(480.0, bail)
        global_dog.daemon = True
        global_dog.start()
    line = run(args, rt)
    if world == 1:
        global_dog.cancel()
        global_dog.join()   # the timer thread holds the runtime (model, tensors): it must be gone before the interpreter winds down
    if rank == 0 and world == 1 and args.pipelined_streams > 1 and args.streams == 1:
        line = pipelined_extra(args, rt, line)
    if rank == 0 and world == 1 and args.passes == "all" and not args.no_live_traffic and "roofline" in line:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got, note = live_traffic(args, [line["roofline"]["kernel"] + "<", "k_density<"])
        for key, prefix in (("roofline", line["roofline"]["kernel"] + "<"), ("roofline_k_density", "k_density<")):
            if got and prefix in got and key in line:
                line[key].update({"traffic": got[prefix], "traffic_live": True, "traffic_source": note,
                                  "traffic_committed_profile": line[key].get("traffic")})
        line["live_traffic"] = {"ok": bool(got), "note": note, "seconds": time.perf_counter() - t0}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


WORKLOAD_NAMES = {"zju377_mono": "ZJUMOCAP-377-mono", "zju313": "ZJUMOCAP-313", "h36m": "H36M-S9 (idr colour mode)"}


class GpuRuntime:
    """What run() needs from the machine: the model, the scene, device synchronisation, events around the dominant
    kernel.  tests/test_bench_sharding.py drives the same run() on two gloo ranks with a stub of this class."""

    def __init__(self, world, rank, dev, dist, model, cfg, scene, hip):
        self.world, self.rank, self.dev, self.dist = world, rank, dev, dist
        self.model, self.cfg, self.scene, self.hip = model, cfg, scene, hip
        self.tracer = model.idhr_network.ray_tracer
        self.ev0 = torch.cuda.Event(enable_timing=True)
        self.ev1 = torch.cuda.Event(enable_timing=True)
        self.cv0 = torch.cuda.Event(enable_timing=True)   # around loop C's solver (k_canon_solve)
        self.cv1 = torch.cuda.Event(enable_timing=True)
        # the tiered forward launches loop C's solver and the density pass twice per frame (phase 1 / phase 2)
        self.cw0, self.cw1, self.dw0, self.dw1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))

    def make_inputs(self, size, frame_idx):
        return self.scene.make_inputs(size, size, frame_idx=frame_idx, device=self.dev)

    def render(self, inputs):
        return self.model(inputs, eval=True)

    def render_many(self, frames, n_streams):
        """The timed region: independent frames, n_streams of them in flight (renderer.render_sequence)."""
        from arah_release_amd import renderer
        return renderer.render_sequence(self.model, frames, n_streams=n_streams, eval=True)

    def prepare(self, n_rays, n_steps):
        """Size every scratch that exists (one per stream that has rendered) before anything is timed or counted."""
        self.tracer.workspace(self.dev)
        for ws in self.tracer.workspaces():
            ws.ensure(n_rays, n_steps)

    def reset_counters(self):
        for ws in self.tracer.workspaces():
            ws.reset_counters()

    def counters(self):
        tot = {}
        for ws in self.tracer.workspaces():
            for k, v in ws.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def device_sync(self):
        torch.cuda.synchronize()

    def set_events(self, full_shading, on):
        """HIP events around the dominant kernels, on the tracer's sampling objects (fields of the calls: the library keeps no
        process-wide hooks since round 4)."""
        which = "shade" if full_shading else "density"
        self.tracer.set_events(which, self.ev0 if on else None, self.ev1 if on else None)
        self.tracer.set_events("canon", self.cv0 if on else None, self.cv1 if on else None)
        self.tracer.set_events("canon2", self.cw0 if on else None, self.cw1 if on else None)
        self.tracer.set_events("density2", self.dw0 if on else None, self.dw1 if on else None)
        if on:   # a frame that does not reach the second launches leaves these at zero
            for e in (self.cw0, self.cw1, self.dw0, self.dw1):
                e.record()

    def set_tiering(self, on):
        """Tiered evaluation (csrc/tier.hpp) on / off for the following frames; returns the previous setting."""
        idhr = self.model.idhr_network
        was, idhr.tiering = idhr.tiering, bool(on)
        return was

    def tiered(self):
        return bool(self.model.idhr_network.tiering)

    def phase2_ms(self):
        """(loop C, density) durations of the tiered forward's second launches of the last event-timed frame."""
        return self.cw0.elapsed_time(self.cw1), self.dw0.elapsed_time(self.dw1)

    def set_adaptive(self, on):
        """Lazy / full shading chosen per frame from the measured share of sigma > 0 samples (renderer.IDHRNetwork) or pinned."""
        self.model.idhr_network.adaptive_shading = bool(on) and os.environ.get("ARAH_ADAPTIVE_SHADING", "1") != "0"

    def set_precision(self, name):
        """GEMM engine the following frames are prepared for: an attribute of this renderer, not the process environment."""
        self.model.idhr_network.precision = {"split": self.hip.PRECISION_SPLIT_F16, "fp32": self.hip.PRECISION_FP32}[name]

    def event_ms(self):
        return self.ev0.elapsed_time(self.ev1)

    def canon_ms(self):
        return self.cv0.elapsed_time(self.cv1)

    def split_engine(self):
        return self.hip.default_precision() == self.hip.PRECISION_SPLIT_F16

    def beta_sweep(self, timed_inputs, warm_inputs, n_streams, n_steps, betas=(1e-3, 1e-2, 3e-2)):
        """What exact lazy shading saves depends on the VolSDF beta, a LEARNED parameter (|deviation_decoder.variance|; 1e-3
        is the reference's initial value, metaavatar_render/models/decoder.py:127-133): density is exactly 0 beyond 16.6 beta
        from the surface, so the share of samples that need a normal and a colour grows with beta.  The same frames, the
        product's default path (frames in flight), one short pass per beta; rays/s and shaded samples per ray."""
        var = self.model.deviation_decoder.variance
        keep = var.detach().clone()
        rays = sum(int(i["ray_dirs"].shape[1]) for i in timed_inputs)
        rows = []
        idhr = self.model.idhr_network

        def one_pass():
            self.render_many(warm_inputs[:1] + timed_inputs[:n_streams], n_streams)   # also lets the adaptive choice settle
            self.device_sync()
            self.reset_counters()
            self.device_sync()
            t0 = time.perf_counter()
            self.render_many(timed_inputs, n_streams)
            self.device_sync()
            dt = time.perf_counter() - t0
            c = self.counters()
            return {"value": rays / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / max(len(timed_inputs), 1),
                    "shaded_samples_per_ray": c["n_col"] / max(rays, 1),
                    "samples_through_the_density_pre_pass_per_ray": c["n_density"] / max(rays, 1)}
        try:
            for b in betas:
                with torch.no_grad():
                    var.fill_(b)
                    self.set_adaptive(False)
                    row = {"beta": b, "lazy_shading_pinned": one_pass()}
                    self.set_adaptive(True)
                    idhr._shade_full, idhr.shade_ratio, idhr._tier_off, idhr.tier_share = False, None, False, None
                    row["product_default"] = one_pass()       # lazy or full per frame, from the measured share
                    row["product_default"]["measured_share_of_shaded_samples"] = idhr.shade_ratio
                    row["product_default"]["measured_share_of_samples_the_tiers_skip"] = getattr(idhr, "tier_share", None)
                rows.append(row)
        finally:
            with torch.no_grad():
                var.copy_(keep)
            self.set_adaptive(True)
            idhr._shade_full, idhr.shade_ratio, idhr._tier_off, idhr.tier_share = False, None, False, None
        return {"note": "the same %d frames (%d in flight) with the VolSDF beta overridden; beta = 1e-3 is the subject's own value "
                        "(the reference's initial value).  lazy_shading_pinned: the density pre-pass + normal / colour for the "
                        "sigma > 0 samples only (what `value` measures); product_default: the renderer picks lazy or full "
                        "shading per frame from the share of sigma > 0 samples earlier frames reported (same image either way, "
                        "bit for bit).  With every valid sample shaded (value_full_shading) the figure does not depend on beta"
                        % (len(timed_inputs), n_streams),
                "sweep": rows}

    def reference_frame_parity(self, args):
        """Frame 0 of the benchmark workload against the REFERENCE's own render of it (tests/golden/
        f7_forward_zju377_mono_512x512_s64.npz: taconite/arah-release on the CPU, all 155 572 rays, written by
        tests/golden/make_golden.py f7full in the build container; data, not code).  BASELINE.json's "+ PSNR vs ref"."""
        import numpy as np
        path = os.path.join(ROOT, "tests", "golden", "f7_forward_%s_%dx%d_s%d.npz" % (args.config, args.size, args.size, args.n_steps))
        if not os.path.exists(path):
            return None
        g = np.load(path)
        with torch.no_grad():
            out = self.model(self.make_inputs(args.size, int(g["frame_idx"])), eval=True)
        rgb = out["rgb_values"][0].double().cpu().numpy()
        mask = out["network_body_mask"][0].cpu().numpy()
        if rgb.shape != g["rgb_values"].shape:
            return None
        mse = float(np.mean((rgb - g["rgb_values"].astype(np.float64)) ** 2))
        pc = out["points_cam"][0].cpu().numpy()
        hit, hit_ref = np.abs(pc).sum(-1) > 0, np.abs(g["points_cam"]).sum(-1) > 0
        both = hit & hit_ref
        return {"note": "frame %d of this workload, every ray, against the reference's own render of it (fixture written by "
                        "tests/golden/make_golden.py f7full: the reference on %d CPU threads, %.0f s)"
                        % (int(g["frame_idx"]), int(g["reference_threads"]), float(g["reference_seconds"])),
                "rays": int(rgb.shape[0]), "psnr_db": None if mse == 0 else -10.0 * float(np.log10(mse)),
                "mask_agreement": float((mask == g["network_body_mask"]).mean()),
                "surface_hit_agreement": float((hit == hit_ref).mean()),
                "surface_points_within_2e-4": float((np.abs(pc[both] - g["points_cam"][both]).max(-1) <= 2e-4).mean()),
                "reference_cpu_rays_per_s": float(rgb.shape[0] / float(g["reference_seconds"])),
                "reference_cpu_threads": int(g["reference_threads"])}

    def gpu_torch_baseline(self, args, near, far):
        return gpu_torch_baseline(self.scene, args.config, args.size, args.n_steps, near, far, args.gpu_baseline_rays,
                                  self.model, self.dev)

    def cpu_baseline(self, args, near, far):
        if (os.cpu_count() or 1) >= 16 and os.environ.get("ARAH_CPU_BASELINE_PROCESSES", "1") != "0":
            try:   # all host cores: processes x threads (a Python-loop oracle does not scale with threads alone)
                return cpu_baseline_multiprocess(self.scene, args.config, args.size, args.n_steps, near, far,
                                                 max(args.cpu_sample_rays, 32768), self.model, self.dev)
            except Exception as e:   # a box that cannot spawn: the single-process figure
                sys.stderr.write("multi-process CPU baseline failed (%s: %s); single process\n" % (type(e).__name__, e))
        return cpu_baseline(self.scene, args.config, args.size, args.n_steps, near, far, args.cpu_sample_rays,
                            model_gpu=self.model, dev=self.dev)

    def test_py_frame(self, size, steps=4, warmup=1):
        """The frame test.py itself asks for (lightning_model.py:320: gen_cano_mesh=True): the render PLUS the canonical mesh
        of the emitted SDF (256^3 lattice, marching cubes, forward skinning) and the three 512 x 512 normal maps, one frame
        after the other on one stream.  Not `value`: BASELINE.json's metric is the rays of model.forward without the mesh."""
        frames = [self.make_inputs(size, k) for k in range(steps + warmup)]
        with torch.no_grad():
            for k in range(warmup):
                self.model(frames[k], gen_cano_mesh=True, eval=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(warmup, warmup + steps):
                out = self.model(frames[k], gen_cano_mesh=True, eval=True)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rays = sum(int(f["ray_dirs"].shape[1]) for f in frames[warmup:])
        line = {"note": "MetaAvatarRender.forward(gen_cano_mesh=True, eval=True) as test.py calls it: render + canonical mesh "
                        "(arah_sdf_grid 256^3, marching cubes, arah_skin_lbs) + output_normal / normal_cano_front / "
                        "normal_cano_back (arah_rasterize), one frame at a time",
                "value": rays / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                "outputs": sorted(k for k in out if k != "sdf_params")}
        # the same frames the way arah_release_amd.test_sequence renders them: several in flight (renderer.map_in_flight)
        from arah_release_amd import renderer
        fn = lambda f: self.model(f, gen_cano_mesh=True, eval=True)   # noqa: E731
        renderer.map_in_flight(fn, frames[:warmup], owner=self.model)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        renderer.map_in_flight(fn, frames[warmup:], owner=self.model)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line["in_flight"] = {"frames_in_flight": renderer.frames_in_flight(steps), "value": rays / dt, "unit": "rays/s",
                             "ms_per_step": 1e3 * dt / steps}
        return line

    def training_line(self, steps=5, warmup=4):
        """Training step of BASELINE.json configs[2] (ZJUMOCAP-313 shapes, one view of 2048 rays on this GPU): forward
        (HIP ray tracer + hand-written loop D) + IDHRLoss + backward + Adam."""
        from arah_release_amd import config, training
        # the inference passes before this line leave tens of GB of cached blocks (scratches of five streams, the full-shading
        # slabs) in torch's allocator; the step's own 350 MB gradient blocks then came out of device allocations / frees in the
        # timed steps every other run (64 against 25 ms per step on the same box, same build): start from an empty cache and
        # let the allocator settle during the warm-up steps
        import gc
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        model, cfg = config.build_synthetic_model("zju313", device=self.dev)
        model.train()
        opt = training.configure_optimizers(model, cfg)
        crit = training.build_loss(cfg)
        batches = [self.scene.make_inputs(512, 512, frame_idx=k, max_rays=2048, eval_mode=False, device=self.dev)
                   for k in range(steps + warmup)]

        def step(inp):
            opt.zero_grad(set_to_none=True)
            losses = training.training_step(model, crit, inp)
            losses["loss"].backward()
            opt.step()

        with torch.enable_grad():
            for k in range(warmup):
                step(batches[k])
            torch.cuda.synchronize()
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            stats0 = torch.cuda.memory_stats(self.dev)
            t0, c0 = time.perf_counter(), time.process_time()
            marks[0].record()
            for k in range(warmup, warmup + steps):
                step(batches[k])
                marks[k - warmup + 1].record()
            c1 = time.process_time()
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stats1 = torch.cuda.memory_stats(self.dev)
        each = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        return {"note": "ZJUMOCAP-313 training step, 1 view x 2048 rays on one GPU: HIP ray tracer (no_grad) + hand-written "
                        "loop-D forward/backward (k_shade_train, on the samples whose density can carry a gradient: sdf / beta <= 110) and "
                        "regulariser queries + compositing / loss / hypernetwork on autograd + fused Adam",
                "value": 2048 * steps / dt, "unit": "rays/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
                "host_cpu_ms_per_step": 1e3 * (c1 - c0) / steps,
                "ms_each_step": each,
                "device_allocations_in_the_timed_steps": int(stats1.get("num_device_alloc", 0) - stats0.get("num_device_alloc", 0)),
                "device_frees_in_the_timed_steps": int(stats1.get("num_device_free", 0) - stats0.get("num_device_free", 0)),
                "host_note": "CPU time of this process until the last step was enqueued (round 6: the step is ~720 launches, 15 ms of "
                             "kernels and host-bound; round 5: 1100 launches, 24.9 ms of kernels).  Earlier in round 5 this line came back at 60-65 ms every other run: the step's "
                             "350 MB gradient blocks were device allocations / frees inside the timed steps when the inference passes "
                             "had left torch's allocator full of other sizes; the line now starts from an empty cache with four warm-up "
                             "steps (4 of 4 runs 24.6-25.8 ms, profiles/r05_train_host.txt)"}


def pipelined_extra(args, rt, line, limit_s=90.0):
    """The default path once more with several frames of the sequence in flight (renderer.render_sequence).  Kept out of
    `value`: with several HIP streams and thousands of queued launches the runtime of this image (ROCm 7.2) has twice
    stopped accepting launches for good, so this pass runs last, under a watchdog that prints the
    line without it and ends the process if it does not come back."""
    import threading

    def give_up():
        line["frames_in_flight"] = {"note": "pass did not return within %.0f s (HIP runtime stall); not measured" % limit_s}
        print(json.dumps(line), flush=True)
        os._exit(0)

    dog = threading.Timer(limit_s, give_up)
    dog.daemon = True
    dog.start()
    n = args.pipelined_streams
    warm = [rt.make_inputs(args.size, f) for f in shard_frames(0, 1, args.steps, args.warmup)[0]]
    timed = [rt.make_inputs(args.size, f) for f in shard_frames(0, 1, args.steps, args.warmup)[1]]
    rays = sum(int(i["ray_dirs"].shape[1]) for i in timed)
    with torch.no_grad():
        rt.render_many(warm + timed[:n], n)
        rt.device_sync()
        rt.prepare(max(int(i["ray_dirs"].shape[1]) for i in warm + timed), args.n_steps)
        rt.device_sync()
        t0 = time.perf_counter()
        rt.render_many(timed, n)
        rt.device_sync()
        dt = time.perf_counter() - t0
    dog.cancel()
    dog.join()
    line["frames_in_flight"] = {"note": "same frames, same path, %d frames of the sequence in flight on %d HIP streams "
                                        "(renderer.render_sequence; per-frame results bit-identical)" % (n, n),
                                "streams": n, "value": rays / dt, "unit": "rays/s",
                                "ms_per_step": 1e3 * dt / max(args.steps, 1)}
    return line


def run(args, rt):
    """Everything after the model exists: frame sharding, the timed passes, whole-job aggregation, the JSON line."""
    world, rank, dist = rt.world, rt.rank, rt.dist
    tracer = rt.tracer
    near = far = args.n_steps // 4
    cfg = rt.cfg
    warm_frames, timed_frames = shard_frames(rank, world, args.steps, args.warmup)
    warm_inputs = [rt.make_inputs(args.size, f) for f in warm_frames]
    timed_inputs = [rt.make_inputs(args.size, f) for f in timed_frames]
    n_rays_local = sum(int(i["ray_dirs"].shape[1]) for i in timed_inputs)

    def sync():
        if world > 1:
            dist.barrier()
        rt.device_sync()

    n_rays_max = max(int(i["ray_dirs"].shape[1]) for i in warm_inputs + timed_inputs)

    canon_ms = {}

    phase2_ms = {}
    has_tiers = hasattr(rt, "set_tiering")

    def timed_pass(full_shading, precision="split", n_streams=None, tiering=None):
        """K timed steps (barrier + sync on both sides), then the same K steps again with HIP events
        around the dominant kernel (reading an event needs a sync per step, so it stays outside).
        tiering: None = the product's setting, False = every sample through loops C and D (the reference's amount of work)."""
        # ARAH_FULL_SHADING=1 forces the shade-everything path in every pass (the profiler's full-shading PMC passes)
        tracer.full_shading = full_shading or os.environ.get("ARAH_FULL_SHADING") == "1"
        was_tiering = rt.set_tiering(rt.tiered() if tiering is None else tiering) if has_tiers else None
        rt.set_adaptive(False)   # the passes measure the path they name; the product's own choice is reported by beta_sweep
        rt.set_precision(precision)
        n_streams = args.streams if n_streams is None else n_streams
        # every stream of the timed region sees a warm-up frame (torch's caching allocator keeps a pool per stream: with W < the
        # frames in flight the cold streams' first frames went to the driver for memory inside the timed region, 16 device
        # allocations and +1.3 ms per frame over an eight-frame pass, tools/probes/alloc_probe.py): the W frames are repeated
        warm = list(warm_inputs)
        while warm_inputs and len(warm) < n_streams:
            warm += list(warm_inputs)
        rt.render_many(warm[:max(len(warm_inputs), n_streams)], n_streams)
        sync()
        rt.prepare(n_rays_max, args.n_steps)   # every scratch sized for the largest frame before anything is timed or counted
        rt.reset_counters()
        sync()
        t0 = time.perf_counter()
        rt.render_many(timed_inputs, n_streams)
        sync()
        dt = time.perf_counter() - t0
        ctr = rt.counters()
        rt.set_events(full_shading, True)
        ms, cms, p2 = [], [], []
        for inp in timed_inputs:
            rt.render(inp)
            rt.device_sync()
            ms.append(rt.event_ms())
            cms.append(rt.canon_ms())
            if has_tiers and rt.tiered() and not tracer.full_shading:
                p2.append(rt.phase2_ms())
        rt.set_events(full_shading, False)
        rt.set_adaptive(True)
        rt.set_precision(default_engine)
        canon_ms[(full_shading, precision, tiering)] = cms
        phase2_ms[(full_shading, precision, tiering)] = p2
        if has_tiers:
            rt.set_tiering(was_tiering)
        return dt, ctr, ms

    default_engine = os.environ.get("ARAH_PRECISION", "split")
    split = rt.split_engine()
    with torch.no_grad():
        elapsed_one = None
        if args.streams > 1:                                    # the same frames strictly one after the other, FIRST:
            elapsed_one, _, _ = timed_pass(False, default_engine, n_streams=1)   # it is what the watchdog falls back on
        first = elapsed_one
        if first is None:
            elapsed, counters, dens_ms = timed_pass(False, default_engine)      # the product's default path
            first = elapsed
        if world == 1:   # what the watchdog of main() prints if a LATER pass never returns
            rt.partial_line = {
                "metric": "rendered rays/sec", "value": n_rays_local / first, "unit": "rays/s", "n_gpus": 1,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * first / max(args.steps, 1),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": WORKLOAD_NAMES.get(args.config, args.config) + " test.py inference, %dx%d, %d samples/ray"
                                       % (args.size, args.size, args.n_steps), "config": args.config},
                "note": "PARTIAL LINE: a pass after the first one did not return (watchdog); roofline / cpu_baseline objects "
                        "were not reached" + ("; value = one frame at a time" if elapsed_one else "")}
        if elapsed_one is not None:
            elapsed, counters, dens_ms = timed_pass(False, default_engine)      # the product's default path: frames in flight
        elapsed_full = elapsed_exact = elapsed_strict = elapsed_untiered = None
        if args.passes == "all" and has_tiers and rt.tiered():   # every sample through loops C and D, like the reference
            elapsed_untiered, counters_untiered, dens_ms_untiered = timed_pass(False, default_engine, tiering=False)
        if args.passes == "all":
            elapsed_full, counters_full, shade_ms = timed_pass(True, default_engine)  # shade every valid sample, like the reference
        if split and args.passes == "all":                      # same frames on the exact fp32 MFMA engine
            elapsed_exact, _, _ = timed_pass(False, "fp32")
            elapsed_strict, counters_strict, strict_ms = timed_pass(True, "fp32")   # all of the reference's work, fp32 MFMA only
        tracer.full_shading = False

    total_rays, t_max = aggregate(n_rays_local, elapsed, dist)
    t_max_one = aggregate(n_rays_local, elapsed_one, dist)[1] if elapsed_one else None
    t_max_full = aggregate(n_rays_local, elapsed_full, dist)[1] if elapsed_full else None
    t_max_exact = aggregate(n_rays_local, elapsed_exact, dist)[1] if elapsed_exact else None
    t_max_strict = aggregate(n_rays_local, elapsed_strict, dist)[1] if elapsed_strict else None
    t_max_untiered = aggregate(n_rays_local, elapsed_untiered, dist)[1] if elapsed_untiered else None
    line = None
    if rank == 0:
        mode = cfg["model"]["renderer_kwargs"]["mode"]

        def path_flops(c):
            return (F_SDF * c["n_sdf_fwd"] + F_SDF_GRAD * c["n_sdf_grad"] + F_SKIN * (c["n_skin_fwd"] + 3 * c["n_skin_jac"]) +
                    F_COL[mode] * c["n_col"] + 55120 * c["n_knn"])

        # largest launch of the default path: k_canon_solve = loop C, every Broyden iteration of every valid sample in one
        # resident kernel (skinning MLP 3 -> 128 x4 -> 25 on the split engine + softmax tree + LBS blend + update)
        # Tiered forward (csrc/tier.hpp): the solver and the density pass are launched twice per frame, over the phase-1 samples
        # (surface rays, samples inside the posed fat body, witnesses) and over the rest of the promoted rays; `achieved` is
        # the evaluations of both launches over the time of both, avg_launch_ms the mean of the two.
        tiers_on = has_tiers and rt.tiered()
        cms = canon_ms[(False, default_engine, None)]
        p2 = phase2_ms.get((False, default_engine, None)) or []
        n_frames_ev = max(len(cms), 1)
        canon_launches = 2 if p2 else 1
        canon_total_ms = sum(cms) + sum(a for a, _ in p2)
        canon_avg_ms = canon_total_ms / (n_frames_ev * canon_launches)
        canon_evals = counters["n_canon"] / (n_frames_ev * canon_launches)
        canon_achieved = counters["n_canon"] * F_SKIN / (canon_total_ms * 1e-3) / 1e12
        # second largest: k_density = the SDF MLP forward on every valid sample that is not certified sigma = +0 (one launch
        # per frame: phase 2's samples and the witnesses lie outside the posed fat body)
        n_launch = max(len(dens_ms), 1)
        dens_total_ms = sum(dens_ms)
        dens_samples = counters["n_density"] / n_launch           # == number of valid (converged) samples that were evaluated
        dens_avg_ms = dens_total_ms / n_launch
        achieved = counters["n_density"] * F_SDF / (dens_total_ms * 1e-3) / 1e12
        flops_per_sample = F_SDF + F_SDF_GRAD + F_COL[mode]
        total_flops = path_flops(counters)
        tf = {True: "true", False: "false"}
        canon_kernel = "k_canon_wave" if split and os.environ.get("ARAH_CANON_KERNEL", "wave") != "tile" else "k_canon_solve"
        dens_traffic, traffic_src = pmc_traffic("k_density<%s" % tf[split])
        canon_traffic, canon_src = pmc_traffic(canon_kernel + "<")
        shade_traffic, shade_src = pmc_traffic("k_shade<%s, %s" % (tf[mode == "idr"], tf[split]), full_shading=True)
        peak_fwd = PEAK_SPLIT_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        # k_shade: forward trunk on the default engine; reverse sweep and colour MLP on the bf16 x 3 engine (three bf16
        # MFMAs per fp32 product: the same rate as the f16 split) unless ARAH_SHADE_ENGINE=fp32 / the exact engine
        shade_b3 = split and os.environ.get("ARAH_SHADE_ENGINE", "b3") != "fp32"
        peak_shade = mixed_peak([(F_SDF, peak_fwd), (F_SDF_GRAD + F_COL[mode], peak_fwd if shade_b3 else PEAK_F32_MFMA_TFLOPS)])
        engine = ("fp32 operands as hi+lo f16 pairs, 3 x v_mfma_f32_16x16x32_f16 per product, fp32 accumulate "
                  "(forward SDF trunks, loop-C skinning MLP); " +
                  ("hi+lo bf16 pairs, 3 x v_mfma_f32_16x16x32_bf16 (2^-16 per product) for loop D's normal sweep and "
                   "colour MLP" if shade_b3 else "v_mfma_f32_16x16x4_f32 for reverse sweeps and the colour MLP")
                  if split else "v_mfma_f32_16x16x4_f32 everywhere")
        line = {
            "metric": "rendered rays/sec", "value": total_rays / t_max, "unit": "rays/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * t_max / max(args.steps, 1),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (operands as hi+lo f16 pairs on the f16 MFMA pipe, fp32 accumulate)" if split else "f32",
            "data": "synthetic" if getattr(args, "beta", None) is None else "synthetic, VolSDF beta overridden to %g" % args.beta,
            "precision": engine,
            "config": {"workload": WORKLOAD_NAMES.get(args.config, args.config) + " test.py inference, %dx%d, %d samples/ray (near %d / far %d), "
                                   "synthetic capsule body + fitted SIREN, one frame per step" %
                                   (args.size, args.size, args.n_steps, near, far),
                       "config": args.config, "rays_per_frame": n_rays_local / max(args.steps, 1),
                       "pixels_per_frame": args.size * args.size, "parallelism": "frame-parallel x%d" % world,
                       "frames_in_flight_per_gpu": args.streams},
            "roofline": {"bound": "mfma", "kernel": canon_kernel, "achieved": canon_achieved, "peak": peak_fwd,
                         "unit": "TFLOP/s", "frac": canon_achieved / peak_fwd, "traffic": canon_traffic,
                         "traffic_unit": "bytes/launch", "traffic_source": canon_src, "traffic_live": False,
                         "avg_launch_ms": canon_avg_ms, "evaluations_per_launch": canon_evals,
                         "launches_per_frame": canon_launches, "ms_per_frame_in_this_kernel": canon_total_ms / n_frames_ev,
                         "flops_per_evaluation": F_SKIN,
                         "peak_note": ("dense f16 MFMA peak / 3: three v_mfma_f32_16x16x32_f16 per fp32 product "
                                       "(MI355X_MICROARCH.md)" if split else "dense fp32 MFMA peak"),
                         "frac_of_measured_mfma_ceiling": (canon_achieved / (MEASURED_F16_MFMA_TFLOPS / 3.0)) if split else None,
                         "measured_mfma_ceiling_note": "a loop of nothing but v_mfma_f32_16x16x32_f16 on random data sustains "
                                                       "1968 TFLOP/s on this part (power-limited clock; tools/ubench/mfma_shape.hip, "
                                                       "profiles/r05_mfma_shape.txt) = 656 TFLOP/s in this engine's algorithmic "
                                                       "flops; context only, `frac` is against the nominal peak"},
            "roofline_k_density": {"bound": "mfma", "kernel": "k_density", "achieved": achieved, "peak": peak_fwd,
                                   "unit": "TFLOP/s", "frac": achieved / peak_fwd, "traffic": dens_traffic,
                                   "traffic_unit": "bytes/launch", "traffic_source": traffic_src, "traffic_live": False,
                                   "avg_launch_ms": dens_avg_ms, "samples_per_launch": dens_samples,
                                   "flops_per_sample": F_SDF,
                                   "frac_of_measured_mfma_ceiling": (achieved / (MEASURED_F16_MFMA_TFLOPS / 3.0)) if split else None,
                                   "note": "second largest launch (the dominant one of round 1): SDF MLP forward on every "
                                           "valid sample"},
            "work": {"per_ray": {k: v / max(n_rays_local, 1) for k, v in counters.items()},
                     "algorithmic_mflop_per_ray": total_flops / max(n_rays_local, 1) / 1e6,
                     "whole_path_tflops_rank0": total_flops / elapsed / 1e12},
        }
        if t_max_one:
            line["one_frame_at_a_time"] = {"note": "same frames, same path, a single HIP stream (no frame in flight while "
                                                   "another one renders): the latency figure",
                                           "value": total_rays / t_max_one, "unit": "rays/s",
                                           "ms_per_step": 1e3 * t_max_one / max(args.steps, 1)}
        line["roofline"]["k_density_frac"] = achieved / peak_fwd
        if t_max_untiered:
            # the same frames with every sample of every ray through loops C and D, as the reference runs them
            # (ray_tracing.py:313-380, implicit_differentiable_renderer.py:261-396): one launch of each kernel per frame
            ucms = canon_ms[(False, default_engine, False)]
            u_ms = sum(ucms) / max(len(ucms), 1)
            u_ach = counters_untiered["n_canon"] / max(len(ucms), 1) * F_SKIN / (u_ms * 1e-3) / 1e12
            d_ms = sum(dens_ms_untiered) / max(len(dens_ms_untiered), 1)
            d_ach = counters_untiered["n_density"] / max(len(dens_ms_untiered), 1) * F_SDF / (d_ms * 1e-3) / 1e12
            line["untiered"] = {"note": "same frames with tiered evaluation off (ARAH_TIERING=0): loops C and D over every depth sample "
                                        "of every ray, like the reference; bit-identical images and masks",
                                "value": total_rays / t_max_untiered, "unit": "rays/s",
                                "ms_per_step": 1e3 * t_max_untiered / max(args.steps, 1),
                                "algorithmic_mflop_per_ray": path_flops(counters_untiered) / max(n_rays_local, 1) / 1e6,
                                "roofline": {"bound": "mfma", "kernel": canon_kernel, "achieved": u_ach, "peak": peak_fwd,
                                             "unit": "TFLOP/s", "frac": u_ach / peak_fwd, "avg_launch_ms": u_ms,
                                             "evaluations_per_launch": counters_untiered["n_canon"] / max(len(ucms), 1)},
                                "roofline_k_density": {"bound": "mfma", "kernel": "k_density", "achieved": d_ach, "peak": peak_fwd,
                                                       "unit": "TFLOP/s", "frac": d_ach / peak_fwd, "avg_launch_ms": d_ms}}
            line["roofline"]["untiered_frac"] = u_ach / peak_fwd
            line["roofline"]["untiered_k_density_frac"] = d_ach / peak_fwd
            if tiers_on:
                line["roofline"]["frac_note"] = (
                    "tiered frames launch this kernel twice on SHORT lists (a fifth of the untiered evaluations): a launch lasts as long "
                    "as the chain of its slowest samples (51 dependent Broyden steps, ~2.4 ms) whatever it holds, so `frac` measures "
                    "that chain, not the kernel -- the same kernel on the untiered list of the same frames: `untiered_frac`; the "
                    "frame's second MFMA kernel: `k_density_frac`")
        if tiers_on:
            per = max(n_rays_local, 1)
            line["tiers"] = {"note": "tiered evaluation (csrc/tier.hpp): rays whose segment misses the posed fat body skip loops A+B; "
                                     "samples outside it are certified sigma = +0 and never evaluated; a non-surface ray is promoted "
                                     "to full evaluation when a phase-1 sample shows density > 0 or none converged",
                             "share_of_rays": {k[7:]: counters[k] / max(counters["n_tier_rays"], 1) for k in
                                               ("n_tier_rays_surface", "n_tier_rays_promoted", "n_tier_rays_skipped", "n_tier_rays_untraced")},
                             "samples_per_ray": {"phase1": counters["n_tier_samples_p1"] / per, "phase2": counters["n_tier_samples_p2"] / per,
                                                 "never_evaluated": counters["n_tier_samples_skipped"] / per}}
        if t_max_full:
            # dominant kernel of the shade-everything path: k_shade (forward trunk on the default engine, reverse sweep
            # and colour MLP on the bf16 x 3 engine)
            n_l = max(len(shade_ms), 1)
            samples_per_launch = counters_full["n_col"] / n_l
            avg_ms = sum(shade_ms) / n_l
            achieved_full = samples_per_launch * flops_per_sample / (avg_ms * 1e-3) / 1e12
            line["full_shading"] = {"note": "same frames with lazy shading off (normal + colour for EVERY valid sample, as "
                                            "the reference does); bit-identical images",
                                    "value": total_rays / t_max_full, "unit": "rays/s",
                                    "ms_per_step": 1e3 * t_max_full / max(args.steps, 1),
                                    "algorithmic_mflop_per_ray": path_flops(counters_full) / max(n_rays_local, 1) / 1e6,
                                    "roofline": {"bound": "mfma", "kernel": "k_shade", "achieved": achieved_full,
                                                 "peak": peak_shade, "unit": "TFLOP/s",
                                                 "frac": achieved_full / peak_shade, "traffic": shade_traffic,
                                                 "traffic_source": shade_src, "traffic_live": False,
                                                 "peak_note": "time-weighted over the kernel's GEMM classes and their engines",
                                                 "avg_launch_ms": avg_ms, "samples_per_launch": samples_per_launch,
                                                 "flops_per_sample": flops_per_sample}}
        if t_max_exact:
            line["exact_fp32_engine"] = {"note": "same frames with ARAH_PRECISION=fp32 (v_mfma_f32_16x16x4_f32 for every "
                                                 "GEMM, two workgroups per CU)",
                                         "value": total_rays / t_max_exact, "unit": "rays/s",
                                         "ms_per_step": 1e3 * t_max_exact / max(args.steps, 1)}
        if t_max_strict:
            n_l = max(len(strict_ms), 1)
            ach = counters_strict["n_col"] / n_l * flops_per_sample / (sum(strict_ms) / n_l * 1e-3) / 1e12
            line["strict"] = {"note": "reference-equivalent work: exact fp32 MFMA engine (v_mfma_f32_16x16x4_f32 for every "
                                      "GEMM) AND normal + colour for every valid sample",
                              "value": total_rays / t_max_strict, "unit": "rays/s",
                              "ms_per_step": 1e3 * t_max_strict / max(args.steps, 1),
                              "roofline": {"bound": "mfma", "kernel": "k_shade", "achieved": ach,
                                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                           "frac": ach / PEAK_F32_MFMA_TFLOPS,
                                           "avg_launch_ms": sum(strict_ms) / n_l}}
        # the reference shades every valid sample in fp32: the two reference-equivalent figures next to the headline
        line["value_full_shading"] = line["full_shading"]["value"] if "full_shading" in line else None
        line["value_strict"] = line["strict"]["value"] if "strict" in line else None
        line["value_untiered"] = line["untiered"]["value"] if "untiered" in line else None
        # the conditions of the headline as scalars inside `config` (the driver's record keeps config / roofline / cpu_baseline)
        line["config"].update({
            "tiered_evaluation": bool(tiers_on), "lazy_shading": True,
            "value_untiered": line["value_untiered"], "value_full_shading": line["value_full_shading"],
            "value_strict": line["value_strict"],
            "one_frame_at_a_time_ms": (1e3 * t_max_one / max(args.steps, 1)) if t_max_one else None,
            "beta": getattr(args, "beta", None) or 1e-3,
            "shaded_samples_per_ray": counters["n_col"] / max(n_rays_local, 1),
            "evaluated_samples_per_ray": counters["n_density"] / max(n_rays_local, 1),
            "skinning_evaluations_per_ray": counters["n_canon"] / max(n_rays_local, 1)})
        if world == 1 and args.passes == "all" and getattr(args, "beta", None) is None and hasattr(rt, "beta_sweep"):
            line["beta_sweep"] = rt.beta_sweep(timed_inputs, warm_inputs, args.streams, args.n_steps)
        if world == 1 and not args.no_train:
            line["training"] = rt.training_line()
            line["config"]["training_ms_per_step"] = line["training"]["ms_per_step"]
            line["test_py_frame"] = rt.test_py_frame(args.size)
        if world == 1 and hasattr(rt, "reference_frame_parity"):
            line["parity_vs_reference_frame"] = rt.reference_frame_parity(args)
        if world == 1 and not getattr(args, "no_gpu_baseline", True) and hasattr(rt, "gpu_torch_baseline"):
            try:
                gb = rt.gpu_torch_baseline(args, near, far)
            except Exception as e:   # the baseline leg must not cost the measurement
                gb = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
            line["gpu_torch_baseline"] = gb
            if gb.get("value"):
                line["vs_baseline"] = line["value"] / world / gb["value"]   # per GPU, like the denominator
                line["vs_baseline_kind"] = ("value per GPU / gpu_torch_baseline.value: a measured stand-in (BASELINE.md section 2), "
                                            "not a number the reference publishes")
                line["config"]["gpu_torch_baseline_rays_per_s"] = gb["value"]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = rt.cpu_baseline(args, near, far)
            line["psnr_vs_oracle_db"] = line["cpu_baseline"].get("psnr_vs_oracle_db")
            line["mask_agreement"] = line["cpu_baseline"].get("mask_agreement")
    return line


if __name__ == "__main__":
    main()
