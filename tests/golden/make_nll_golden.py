"""Golden fixtures of the evaluation-mode likelihood terms (validation / test NLL), produced by the REFERENCE itself.

Run in the build container only (needs /root/reference; CPU):

    python tests/golden/make_nll_golden.py        ->  tests/golden/nll_full_{qm9,qm9cond,geom}.npz

The unmodified ``EquivariantVariationalDiffusion.forward`` (variational_diffusion.py:948-1160, eval mode) runs on a small data-like batch
with full-width synthetic weights, in fp32 and -- as the high-precision adjudicator -- in fp64.  Its two sources of randomness are pinned:
``torch.randint`` (the timesteps) returns the stored ``t_int``, ``torch.randn`` draws from a seeded tape (ref_harness.NoiseTape).
Stored: the inputs, ``t_int``, the noise seed, every term the method returns and the loss info.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_harness as rh  # noqa: E402
import synth  # noqa: E402
from make_golden import cfgs_for, npz  # noqa: E402

torch.set_num_threads(8)
NAMES = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior", "log_pN", "t_int")


def make_case(case, weight_seed=29, noise_seed=4321):
    ds, cond, cfgs = cfgs_for(case)
    d = synth.DATASET_DIMS[case]
    net = rh.build_reference_dynamics(cfgs, seed=0)
    net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=weight_seed))
    ddpm = rh.build_reference_ddpm(cfgs, net, ds)
    include_charges = bool(cfgs["dataloader_cfg"]["include_charges"])
    nt = int(cfgs["dataloader_cfg"]["num_atom_types"])
    sizes = [5, 19, 3, 11] if case != "geom" else [5, 44, 3, 30]
    nn_ = torch.tensor(sizes)
    B, N = len(sizes), sum(sizes)
    bi = torch.repeat_interleave(torch.arange(B), nn_)
    g = torch.Generator().manual_seed(61)
    x = torch.randn((N, 3), generator=g) * 1.5
    for b in range(B):
        x[bi == b] -= x[bi == b].mean(0, keepdim=True)
    types = torch.randint(0, nt, (N,), generator=g)
    one_hot = torch.nn.functional.one_hot(types, nt).float()
    charges = (torch.randint(1, 10, (N,), generator=g).float() if include_charges else torch.zeros((N, 0)))
    ctx = torch.randn((B, 1), generator=g) if d["n_ctx"] else None
    t_int = torch.tensor([[1], [517], [1000], [36]])
    mask = torch.ones(N, dtype=torch.bool)

    def run(dtype):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        orig_randint = torch.randint
        try:
            ddpm.to(dtype)
            ddpm.eval()
            batch = rh.make_batch(bi, mask, None if ctx is None else ctx[bi].to(dtype))
            batch.x = x.to(dtype).clone()
            batch.h = {"categorical": one_hot.to(dtype).clone(), "integer": charges.to(dtype).clone()}
            batch.num_graphs = B
            batch.num_nodes_present = nn_.clone()
            torch.randint = lambda *a, **k: t_int.clone()
            with rh.NoiseTape(noise_seed) as tape, torch.no_grad():
                out = ddpm(batch, return_loss_info=True)
            assert [c[0] for c in tape.calls] == [N] * 4, tape.calls          # x / h noise of z_t, then of z_0
            return out
        finally:
            torch.randint = orig_randint
            torch.set_default_dtype(prev)
            ddpm.to(prev)

    o32, o64 = run(torch.float32), run(torch.float64)
    arrs = dict(num_nodes=nn_, x=x, one_hot=one_hot, charges=charges, ctx=ctx, t_int=t_int.flatten(), weight_seed=weight_seed, noise_seed=noise_seed)
    for tag, o in (("32", o32), ("64", o64)):
        for name, v in zip(NAMES, o[:9]):
            arrs[f"{name}_{tag}"] = v.double() if tag == "64" else v
        for k, v in o[9].items():
            arrs[f"{k}_{tag}"] = v.double() if tag == "64" else v
    npz(f"nll_full_{case}", **arrs)
    worst = max(float(((o32[i].double() - o64[i].double()).abs() / (1.0 + o64[i].double().abs())).max()) for i in range(8))
    print(f"{case}: worst relative fp32-vs-fp64 gap over the terms = {worst:.2e}; error_t = {o32[1].tolist()}")


if __name__ == "__main__":
    assert rh.reference_available(), "reference checkout not found"
    for case in ("qm9", "qm9cond", "geom"):
        make_case(case)
