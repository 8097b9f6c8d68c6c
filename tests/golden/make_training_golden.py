"""Golden fixtures of the TRAINING objective and its gradients, produced by the REFERENCE itself (build container only):

    python tests/golden/make_training_golden.py        ->  tests/golden/train_full_{qm9,geom}.npz

The unmodified ``EquivariantVariationalDiffusion.forward`` in TRAINING mode (variational_diffusion.py:948-1160: one evaluation of the network at
t >= 0, the t = 0 terms masked in) runs on a data-like ragged batch with full-width seed-recreated weights, its two sources of randomness pinned
(``torch.randint`` returns the stored t_int -- which includes a 0 -- and ``torch.randn`` draws from ref_harness.NoiseTape); the terms are
assembled into the L2 training loss exactly as ``QM9MoleculeGenerationDDPM.forward`` does (qm9_mol_gen_ddpm.py:222-262, loss_type "l2",
norm_training_by_max_nodes false) and ``loss = nll.mean(0)`` (``training_step`` :352) is back-propagated by torch autograd through the
reference's modules.  Once in fp32 and once in fp64.  Stored: the batch, t_int, seeds, every term, nll, loss, the gradient NORM and absolute
maximum of every parameter tensor, and a handful of full gradients.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402
from make_golden import cfgs_for  # noqa: E402

torch.set_num_threads(4)
NAMES = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior", "log_pN", "t_int")
FULL_GRADS = ("gcp_embedding.edge_embedding.scalar_out.weight", "gcp_embedding.node_embedding.vector_down.weight",
              "interaction_layers.0.interaction.message_fusion.0.vector_down_frames.weight", "interaction_layers.0.interaction.message_fusion.2.vector_up.weight",
              "interaction_layers.1.interaction.scalar_message_attention.0.weight", "interaction_layers.1.feedforward_network.0.scalar_out.2.bias",
              "interaction_layers.2.node_position_update_gcp.vector_up.weight", "scalar_node_projection_gcp.scalar_out.weight")


def make_case(case, weight_seed=31, noise_seed=2468):
    ds, cond, cfgs = cfgs_for(case)
    d = synth.DATASET_DIMS[case]
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    include_charges = bool(cfgs["dataloader_cfg"]["include_charges"])
    nt = int(cfgs["dataloader_cfg"]["num_atom_types"])
    sizes = [5, 19, 3, 11, 16, 9] if case != "geom" else [5, 44, 3, 30]
    nn_ = torch.tensor(sizes)
    B, N = len(sizes), sum(sizes)
    bi = torch.repeat_interleave(torch.arange(B), nn_)
    g = torch.Generator().manual_seed(67)
    x = torch.randn((N, 3), generator=g) * 1.5
    for b in range(B):
        x[bi == b] -= x[bi == b].mean(0, keepdim=True)
    one_hot = torch.nn.functional.one_hot(torch.randint(0, nt, (N,), generator=g), nt).float()
    charges = (torch.randint(1, 10, (N,), generator=g).float() if include_charges else torch.zeros((N, 0)))
    t_int = torch.tensor([[0], [517], [1000], [36], [1], [250]][:B])
    mask = torch.ones(N, dtype=torch.bool)
    F = nt + int(include_charges)

    def run(dtype):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        orig_randint = torch.randint
        try:
            net = rh.build_reference_dynamics(cfgs, seed=0)
            net.load_state_dict(synth.make_weights(shapes, seed=weight_seed, scale_2d=0.5))
            ddpm = rh.build_reference_ddpm(cfgs, net, ds).to(dtype)
            ddpm.train()
            batch = rh.make_batch(bi, mask, None)
            batch.x = x.to(dtype).clone()
            batch.h = {"categorical": one_hot.to(dtype).clone(), "integer": charges.to(dtype).clone()}
            batch.num_graphs = B
            batch.num_nodes_present = nn_.clone()
            torch.randint = lambda *a, **k: t_int.clone()
            with rh.NoiseTape(noise_seed) as tape:
                terms = ddpm(batch, return_loss_info=True)
            assert [c[0] for c in tape.calls] == [N] * 2, tape.calls           # x / h noise of z_t: ONE evaluation in training mode
            delta_log_px, error_t, SNR_weight, loss_0_x, loss_0_h, neg_log_const_0, kl_prior, log_pN, _, _ = terms
            # qm9_mol_gen_ddpm.py:222-262 (training, loss_type l2, norm_training_by_max_nodes false)
            denom = (3 + F) * nn_
            loss_t = 0.5 * (error_t / denom)
            loss_0 = loss_0_x / denom + loss_0_h
            nll = loss_t + loss_0 + kl_prior - delta_log_px - log_pN
            loss = nll.mean(0)
            loss.backward()
            grads = {k: p.grad.detach().clone() for k, p in ddpm.dynamics_network.named_parameters()}
            return terms, nll.detach(), loss.detach(), grads
        finally:
            torch.randint = orig_randint
            torch.set_default_dtype(prev)

    arrs = dict(num_nodes=nn_.numpy(), x=x.numpy(), one_hot=one_hot.numpy(), charges=charges.numpy(), t_int=t_int.flatten().numpy(),
                weight_seed=weight_seed, weight_scale=0.5, noise_seed=noise_seed, keys=np.array(list(shapes)))
    res = {}
    for tag, dtype in (("32", torch.float32), ("64", torch.float64)):
        terms, nll, loss, grads = run(dtype)
        res[tag] = (loss, grads)
        cast = (lambda v: v.detach().double().numpy()) if tag == "64" else (lambda v: v.detach().float().numpy())
        for name, v in zip(NAMES, terms[:9]):
            arrs[f"{name}_{tag}"] = cast(v)
        arrs[f"nll_{tag}"], arrs[f"loss_{tag}"] = cast(nll), cast(loss)
        arrs[f"grad_norm_{tag}"] = np.array([float(grads[k].double().norm()) for k in shapes])
        arrs[f"grad_absmax_{tag}"] = np.array([float(grads[k].double().abs().max()) for k in shapes])
        for k in FULL_GRADS:
            if k in grads:
                arrs[f"grad_{tag}::{k}"] = cast(grads[k])
    l32, g32 = res["32"]
    l64, g64 = res["64"]
    worst = max(float((g32[k].double() - g64[k]).norm() / max(float(g64[k].norm()), 1e-30)) for k in shapes)
    print(f"{case}: loss {float(l32):.6f} / {float(l64):.6f}; worst relative fp32-vs-fp64 gradient gap = {worst:.2e}; "
          f"zero-gradient tensors: {sum(1 for k in shapes if float(g64[k].norm()) == 0.0)}", flush=True)
    np.savez_compressed(os.path.join(HERE, f"train_full_{case}.npz"), **arrs)


if __name__ == "__main__":
    assert rh.reference_available(), "reference checkout not found"
    for case in ("qm9", "geom"):
        make_case(case)
