"""world_size-2 gloo test of the molecule-sharded sampling path (CPU; the per-shard sampler is the oracle on a
reduced-width net because HIP kernels cannot run here)."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import synth
from oracle import gcdm_oracle as O

par = importlib.import_module("bio-diffusion_amd.parallel")


def test_shard_ranges_cover_everything():
    for B in (1, 2, 7, 256, 1023):
        for w in (1, 2, 3, 8):
            got = []
            for r in range(w):
                lo, hi = par.shard_range(B, r, w)
                got += list(range(lo, hi))
            assert got == list(range(B))
            sizes = [par.shard_range(B, r, w)[1] - par.shard_range(B, r, w)[0] for r in range(w)]
            assert max(sizes) - min(sizes) <= 1


def _small_model():
    shapes = synth.dynamics_shapes(S=32, V=8, Se=16, Ve=4, L=2, h_in=7)
    W = synth.make_weights(shapes, seed=5, scale_2d=0.5)
    cfg = O.OracleConfig(num_layers=2)
    return W, cfg


def _sample_shard(nn_local, seed):
    W, cfg = _small_model()
    out, _ = O.mol_gen_sample(W, cfg, nn_local, O.TapeNoise(seed), num_timesteps=4)
    return out


def _worker(rank, world, port, nn_all, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    nn_local = par.shard_num_nodes(nn_all, rank, world)
    out = _sample_shard(nn_local, 100 + rank)
    full, nn_g = par.gather_samples(out, nn_local)
    if rank == 0:
        q.put((full.numpy(), nn_g.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sampling_equals_independent_shards():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nn_all = torch.tensor([5, 7, 3, 6, 4])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nn_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, nn_g = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nn_g.tolist() == nn_all.tolist()
    want = torch.cat([_sample_shard(par.shard_num_nodes(nn_all, r, 2), 100 + r) for r in range(2)], dim=0).numpy()
    assert full.shape == want.shape and np.array_equal(full, want)
