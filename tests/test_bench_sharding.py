"""Multi-GPU path of bench.py on CPU: frame sharding + whole-job aggregation with a 2-rank gloo group
(the data path has no collective: frames are independent)."""
import os
import socket

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
