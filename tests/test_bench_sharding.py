"""Multi-GPU path of bench.py on CPU: frame sharding + whole-job aggregation with a 2-rank gloo group
(the data path has no collective: frames are independent)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bench


def test_shard_frames_partition():
    for world in (1, 2, 4, 8):
        seen = []
        for r in range(world):
            warm, timed = bench.shard_frames(r, world, steps=5, warmup=2)
            assert len(warm) == 2 and len(timed) == 5
            assert all(f % world == r for f in warm + timed)      # frame i -> rank i mod world
            seen += warm + timed
        assert sorted(seen) == list(range(world * 7))               # disjoint, dense cover


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rays, secs = bench.aggregate(1000 * (rank + 1), 0.5 * (rank + 1), dist)
    out[rank] = (rays, secs)
    dist.barrier()
    dist.destroy_process_group()


def test_aggregate_two_ranks_gloo():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        rays, secs = out[r]
        assert rays == 3000.0          # sum over ranks
        assert secs == 1.0             # max over ranks
    assert bench.aggregate(10, 2.0, None) == (10.0, 2.0)


# ---- the real bench.run() on two gloo ranks with a stub of the GPU runtime --------------------------------------------
class _StubWorkspace:
    def __init__(self):
        self.n = 0

    def ensure(self, n_rays, n_steps):
        pass

    def reset_counters(self):
        self.n = 0

    def counters(self):
        keys = ("n_sdf_fwd", "n_sdf_grad", "n_skin_fwd", "n_skin_jac", "n_col", "n_knn", "n_density", "n_canon")
        return {k: 100 * self.n for k in keys}


class _StubTracer:
    def __init__(self):
        self.full_shading = False
        self.ws = _StubWorkspace()

    def workspace(self, dev):
        return self.ws

    def workspaces(self):
        return [self.ws]


class _StubRuntime:
    """Same surface as bench.GpuRuntime; a 'render' costs 2 ms per frame on rank 0 and 4 ms on rank 1, a frame has
    1000 + frame_idx rays."""

    def __init__(self, world, rank):
        self.world, self.rank, self.dev, self.dist = world, rank, "cpu", dist
        self.cfg = {"model": {"renderer_kwargs": {"mode": "no_view_dir"}}}
        self.tracer = _StubTracer()
        self.rendered = []

    def make_inputs(self, size, frame_idx):
        return {"ray_dirs": torch.zeros(1, 1000 + frame_idx, 3), "frame": frame_idx}

    def render(self, inputs):
        import time
        time.sleep(0.002 * (self.rank + 1))
        self.tracer.ws.n += 1
        self.rendered.append(inputs["frame"])

    def render_many(self, frames, n_streams):
        for f in frames:
            self.render(f)

    def prepare(self, n_rays, n_steps):
        pass

    def reset_counters(self):
        self.tracer.ws.reset_counters()

    def counters(self):
        return self.tracer.ws.counters()

    def device_sync(self):
        pass

    def set_events(self, full_shading, on):
        pass

    def set_adaptive(self, on):
        pass

    def set_precision(self, name):
        pass

    def event_ms(self):
        return 1.0

    def canon_ms(self):
        return 2.0

    def split_engine(self):
        return False


def _run_worker(rank, world, port, out, backend="gloo"):
    import argparse
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":   # RCCL: one GPU per rank, the aggregation's two scalar all-reduces on the device
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    args = argparse.Namespace(gpus=world, steps=3, warmup=1, size=64, n_steps=64, config="zju377_mono",
                              cpu_sample_rays=16, no_cpu_baseline=True, no_train=True, passes="default", streams=1, beta=None)
    rt = _StubRuntime(world, rank)
    line = bench.run(args, rt)
    out[rank] = (line, list(rt.rendered))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_run_two_ranks_rccl():
    """bench.run()'s sharding and aggregation over RCCL (stub renderer, real collectives): needs two visible GPUs."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _check_bench_run("nccl")


def test_bench_run_two_ranks_gloo():
    """bench.run() end to end on two ranks: disjoint frames per rank, value = all ranks' rays / the SLOWER rank's
    time, one JSON line on rank 0 only, with the fields the driver parses."""
    _check_bench_run("gloo")


def _check_bench_run(backend):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_run_worker, args=(world, _free_port(), out, backend), nprocs=world, join=True)
    line0, frames0 = out[0]
    line1, frames1 = out[1]
    assert line1 is None and line0 is not None
    # frame i -> rank i mod 2; each pass renders warm-up + timed + the event-timing repeat of the timed frames
    assert set(frames0) == {0, 2, 4, 6} and set(frames1) == {1, 3, 5, 7}
    assert frames0 == [0, 2, 4, 6, 2, 4, 6]
    rays = sum(1000 + f for f in (2, 4, 6)) + sum(1000 + f for f in (3, 5, 7))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line0, key
    assert line0["n_gpus"] == 2 and line0["steps"] == 3 and line0["scaling"] == "weak" and line0["unit"] == "rays/s"
    t = line0["ms_per_step"] * 3 / 1e3
    assert abs(line0["value"] - rays / t) < 1e-6 * line0["value"]
    assert 3 * 0.004 <= t < 3 * 0.004 + 0.2          # the slower rank (4 ms per frame) sets the time
    assert "workload" in line0["config"] and "model" not in line0["config"]


def test_pipelined_extra_adds_its_object_and_leaves_the_line_alone():
    """The frames-in-flight pass of bench.py (N = 1, after everything else): same frames as the timed region, its own
    object, `value` untouched."""
    import argparse
    args = argparse.Namespace(steps=3, warmup=1, size=64, n_steps=64, pipelined_streams=3)
    rt = _StubRuntime(1, 0)
    line = {"value": 123.0, "ms_per_step": 4.0}
    out = bench.pipelined_extra(args, rt, dict(line), limit_s=30.0)
    assert out["value"] == 123.0 and out["ms_per_step"] == 4.0
    fi = out["frames_in_flight"]
    assert fi["streams"] == 3 and fi["unit"] == "rays/s"
    rays = sum(1000 + f for f in (1, 2, 3))                      # frames 1..3 are the timed ones of rank 0 of 1
    assert abs(fi["value"] - rays / (fi["ms_per_step"] * 3 / 1e3)) < 1e-6 * fi["value"]
    # warm-up + one round of the streams, then the timed frames
    assert rt.rendered == [0, 1, 2, 3, 1, 2, 3]


def test_frames_in_flight_default():
    """renderer.frames_in_flight: never more streams than frames, four from four frames on (round 6: the tiered frames' optimum for
    every sequence length; five from fifteen frames on until round 5)."""
    from arah_release_amd import renderer
    assert [renderer.frames_in_flight(n) for n in (0, 1, 3, 4, 8, 14, 15, 20, 400)] == [1, 1, 3, 4, 4, 4, 4, 4, 4]
    assert renderer.map_in_flight(lambda x: x + 1, []) == []
    assert renderer.map_in_flight(lambda x: x + 1, [1, 2, 3]) == [2, 3, 4]      # host-resident items: plain map


def test_pmc_database_reduction(tmp_path):
    """bench.live_traffic reduces the rocprofv3 --pmc databases of its child runs as tools/rocpd_pmc.py does: KiB -> bytes,
    average per launch, kernel names without namespace and argument list."""
    import sqlite3
    import bench
    path = str(tmp_path / "pmc_results.db")
    db = sqlite3.connect(path)
    db.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    rows = [("void (anonymous namespace)::k_canon_wave<true, false>(FrameDev, int const*)", "FETCH_SIZE", 1000.0),
            ("void (anonymous namespace)::k_canon_wave<true, false>(FrameDev, int const*)", "FETCH_SIZE", 3000.0),
            ("void (anonymous namespace)::k_density<true, 8>(FrameDev, float const*)", "FETCH_SIZE", 10.0),
            ("void (anonymous namespace)::k_density<true, 8>(FrameDev, float const*)", "WRITE_SIZE", 7.0)]
    db.executemany("insert into counters_collection values (?, ?, ?)", rows)
    db.commit()
    db.close()
    f = bench.pmc_db_per_kernel(path, "FETCH_SIZE")
    assert f["k_canon_wave<true, false>"] == (2000.0 * 1024.0, 2)
    assert f["k_density<true, 8>"] == (10.0 * 1024.0, 1)
    assert bench.pmc_db_per_kernel(path, "WRITE_SIZE") == {"k_density<true, 8>": (7.0 * 1024.0, 1)}
