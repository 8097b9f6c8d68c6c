"""Molecule-sharded sampling across the GPUs of one node.

Molecules are independent trajectories, so the sampling batch is cut into contiguous blocks of molecules, one block
per rank (one process per GPU, ``torch.distributed`` backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests).
There is no collective inside the 1000-step loop; the only communication is one gather of the final samples.
Parity definition: the N-rank result equals N independent single-GPU runs of the shards (each shard is its own flat
batch -- the flat-batch orientation quirk, SURVEY A.6.2, is defined per flat batch).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(num_molecules: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced block [lo, hi) of molecule indices owned by ``rank``."""
    base, rem = divmod(num_molecules, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_num_nodes(num_nodes: torch.Tensor, rank: int, world_size: int) -> torch.Tensor:
    lo, hi = shard_range(len(num_nodes), rank, world_size)
    return num_nodes[lo:hi]


def gather_samples(out: torch.Tensor, num_nodes_local: torch.Tensor, group: Optional[dist.ProcessGroup] = None
                   ) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gathers the ragged per-rank results ``out`` [N_r, D] (+ their molecule sizes) in rank order.

    Two fixed-size collectives (sizes, then padded payload): bucket size is the largest shard, which for the
    GEOM config is < 1 MB per rank -- latency-bound on xGMI, bandwidth irrelevant.
    """
    world = dist.get_world_size(group)
    dev = out.device
    meta = torch.tensor([out.shape[0], len(num_nodes_local)], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    max_n = int(max(m[0].item() for m in metas))
    max_b = int(max(m[1].item() for m in metas))
    pad = torch.zeros((max_n, out.shape[1]), dtype=out.dtype, device=dev)
    pad[: out.shape[0]] = out
    nn_pad = torch.zeros(max_b, dtype=torch.int64, device=dev)
    nn_pad[: len(num_nodes_local)] = num_nodes_local.to(dev, torch.int64)
    outs = [torch.empty_like(pad) for _ in range(world)]
    nns = [torch.empty_like(nn_pad) for _ in range(world)]
    dist.all_gather(outs, pad, group=group)
    dist.all_gather(nns, nn_pad, group=group)
    xs: List[torch.Tensor] = [o[: int(m[0].item())] for o, m in zip(outs, metas)]
    ns: List[torch.Tensor] = [n[: int(m[1].item())] for n, m in zip(nns, metas)]
    return torch.cat(xs, dim=0), torch.cat(ns, dim=0)


def sample_sharded(ddpm, num_nodes: torch.Tensor, device, context: Optional[torch.Tensor] = None, num_timesteps: Optional[int] = None,
                   seed: int = 1234, lanes: int = 2, group: Optional[dist.ProcessGroup] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The whole multi-GPU recipe: this rank samples its block of molecules (``EquivariantVariationalDiffusion.mol_gen_sample`` on
    ``device``, its own Philox stream ``seed + rank``), then one gather.  Returns (xh [N_total, 3+F], num_nodes [B_total]) on every rank,
    molecules in the original order.  Works without an initialised process group (single GPU)."""
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(len(num_nodes), rank, world)
    local = num_nodes[lo:hi]
    ctx = None if context is None else context[lo:hi]
    xh, _, _ = ddpm.mol_gen_sample(num_samples=len(local), num_nodes=local, device=device, num_timesteps=num_timesteps, context=ctx,
                                   seed=seed + rank, lanes=lanes)
    if world == 1:
        return xh, torch.as_tensor(local).to(torch.int64)
    return gather_samples(xh, torch.as_tensor(local), group)
