"""Golden vectors for the property-guided optimisation loop, produced by the REFERENCE itself.

    python tests/golden/make_optimize_golden.py        (build container only: needs /root/reference; CPU)

Runs `EquivariantVariationalDiffusion.mol_gen_optimize` (src/models/components/variational_diffusion.py:1416-1546) of the
unmodified reference on the reduced-width alpha-conditional QM9 model of `sampler_small_qm9cond.npz` (same weight seed, so the
weights are NOT stored again) for both time normalisations, under a recorded noise tape.  -> tests/golden/optimize_small_qm9cond.npz
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

torch.set_num_threads(8)


def main():
    case = "qm9cond"
    cfgs = rh.shrink_cfgs(rh.load_reference_cfgs("qm9", ("alpha",)))
    net = rh.build_reference_dynamics(cfgs, seed=4, weight_scale=0.5)       # == the weights stored in sampler_small_qm9cond.npz
    ddpm = rh.build_reference_ddpm(cfgs, net, "qm9")
    d = synth.DATASET_DIMS[case]
    F = d["num_atom_types"]
    nn_ = torch.tensor([5, 7, 3, 6])
    B = len(nn_)
    g = torch.Generator().manual_seed(77)
    ctx_b = torch.randn((B, 1), generator=g)
    samples = []
    for n in nn_.tolist():
        x = torch.randn((n, 3), generator=g) * 1.3
        x = x - x.mean(0, keepdim=True)                                     # the reference asserts CoM-free inputs (:1464)
        h = torch.nn.functional.one_hot(torch.randint(0, F, (n,), generator=g), F).float()
        samples.append((x, h))
    out = dict(num_nodes=nn_, ctx=ctx_b, x=torch.cat([s[0] for s in samples]), h=torch.cat([s[1] for s in samples]),
               weight_check=net.state_dict()["gcp_embedding.node_embedding.scalar_out.0.weight"].float()
               if "gcp_embedding.node_embedding.scalar_out.0.weight" in net.state_dict() else next(iter(net.state_dict().values())).float())
    for tag, T, orig, frames in (("a", 10, False, 1), ("b", 8, True, 1), ("c", 10, False, 5)):      # c: chain frames (return_frames = 5, :1490-1497, 1540-1546)
        with rh.NoiseTape(4321) as tape, torch.no_grad():
            xh, bi, _ = ddpm.mol_gen_optimize(samples=[(x.clone(), h.clone()) for x, h in samples], num_nodes=nn_, device="cpu",
                                              num_timesteps=T, context=ctx_b, norm_with_original_timesteps=orig, return_frames=frames)
        out[f"{tag}_T"], out[f"{tag}_orig"], out[f"{tag}_out"], out[f"{tag}_frames"] = T, int(orig), xh, frames
        out[f"{tag}_calls"] = np.array(tape.calls, dtype=np.int64)
        print(tag, "T", T, "orig", orig, "out", tuple(xh.shape), "randn calls", len(tape.calls), tape.calls[:3])
    out["noise_seed"] = 4321
    path = os.path.join(HERE, "optimize_small_qm9cond.npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
