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


class _OracleDDPM:
    """Stand-in with the `mol_gen_sample` signature `parallel.sample_sharded` calls; samples with the CPU oracle (test only)."""

    def __init__(self):
        self.calls = []

    def mol_gen_sample(self, num_samples, num_nodes, device, num_timesteps=None, context=None, seed=1234, lanes=1):
        self.calls.append((int(num_samples), [int(v) for v in num_nodes], None if context is None else context.clone(), seed, lanes))
        out = _sample_shard(torch.as_tensor(num_nodes), seed)
        if context is not None:                     # make the result depend on the context rows this rank was given
            bi = torch.repeat_interleave(torch.arange(len(num_nodes)), torch.as_tensor(num_nodes))
            out = out + context.to(out.dtype)[bi].sum(-1, keepdim=True)
        return out, None, None


def _worker_sharded(rank, world, port, nn_all, ctx_all, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    ddpm = _OracleDDPM()
    xh, nn_g = par.sample_sharded(ddpm, nn_all, "cpu", context=ctx_all, num_timesteps=4, seed=50, lanes=2)
    q.put((rank, xh.numpy(), nn_g.numpy(), ddpm.calls[0][1], ddpm.calls[0][3], ddpm.calls[0][4]))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_sharded_two_ranks_with_context():
    """`parallel.sample_sharded` end to end under gloo: every rank samples its contiguous block (own context rows, seed + rank), and every rank
    receives all samples in the original molecule order."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nn_all = torch.tensor([5, 7, 3, 6, 4])
    ctx_all = torch.tensor([[0.5], [-1.0], [2.0], [0.25], [-0.75]])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_sharded, args=(r, 2, port, nn_all, ctx_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = []
    for r in range(2):
        lo, hi = par.shard_range(5, r, 2)
        o, _, _ = _OracleDDPM().mol_gen_sample(hi - lo, nn_all[lo:hi], "cpu", 4, ctx_all[lo:hi], 50 + r)
        want.append(o)
        assert res[r][3] == nn_all[lo:hi].tolist() and res[r][4] == 50 + r and res[r][5] == 2
    want = torch.cat(want).numpy()
    for r in range(2):
        assert np.array_equal(res[r][1], want) and res[r][2].tolist() == nn_all.tolist()


def test_bench_plain_command_launches_its_own_ranks():
    """`python bench.py --gpus 2` without WORLD_SIZE (the driver's 1-GPU command shape with N > 1) re-executes itself through torch.distributed.run on
    127.0.0.1 and prints ONE line from rank 0 (GCDM_BENCH_LAUNCH_ONLY: the ranks stop before they touch a GPU -- the rest of the multi-rank branch is
    covered on the GPU box by tests/test_bench_gpu.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["GCDM_BENCH_LAUNCH_ONLY"] = "1"
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3"], cwd=root, env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    assert json.loads(lines[0]) == {"launched_world": 2, "gpus": 2, "master_addr": "127.0.0.1"}
