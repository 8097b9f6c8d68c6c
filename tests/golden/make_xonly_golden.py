"""Position-only diffusion (`generate_x_only=True`) by the REFERENCE itself (build container only):

    python tests/golden/make_xonly_golden.py      ->  tests/golden/sampler_xonly_qm9.npz

The unmodified ``mol_gen_sample(..., generate_x_only=True, num_timesteps=12)`` (variational_diffusion.py:1282-1412 with the x-only branches of
:795-836, 840-907, 735-793, 1204-1278) around a GCPNetDynamics built WITHOUT node features (dataloader_cfg.num_atom_types = 0,
include_charges = False: the only way the reference's network accepts the [N, 3] latent), reduced width, on a noise tape, fp32 and fp64.
Stored: sizes, seeds, z after every step, the final positions.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(4)
SIZES, STEPS, WEIGHT_SEED, NOISE_SEED = [5, 9, 3, 12], 12, 73, 2468


def run(dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cfgs = rh.load_reference_cfgs("qm9", ())
        synth.apply_variant(cfgs, None)
        cfgs["dataloader_cfg"]["num_atom_types"] = 0
        cfgs["dataloader_cfg"]["include_charges"] = False
        net = rh.build_reference_dynamics(cfgs, seed=0)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(synth.make_weights(shapes, seed=WEIGHT_SEED, scale_2d=0.25))
        net = net.to(dtype)
        ddpm = rh.build_reference_ddpm(cfgs, net, "qm9").to(dtype)
        zs = []
        orig = ddpm.sample_p_zs_given_zt

        def spy(*a, **kw):
            out = orig(*a, **kw)
            zs.append(out.detach().clone())
            return out

        ddpm.sample_p_zs_given_zt = spy
        with rh.NoiseTape(NOISE_SEED), torch.no_grad():
            x, bi, _ = ddpm.mol_gen_sample(num_samples=len(SIZES), num_nodes=torch.tensor(SIZES), device="cpu", num_timesteps=STEPS, generate_x_only=True)
        return x, torch.stack(zs), shapes
    finally:
        torch.set_default_dtype(prev)


def main():
    assert rh.reference_available()
    x32, z32, shapes = run(torch.float32)
    x64, z64, _ = run(torch.float64)
    assert x32.shape == (sum(SIZES), 3) and z32.shape[-1] == 3
    np.savez_compressed(os.path.join(HERE, "sampler_xonly_qm9.npz"), num_nodes=np.array(SIZES), steps=STEPS, weight_seed=WEIGHT_SEED, weight_scale=0.25,
                        noise_seed=NOISE_SEED, final32=x32.float().numpy(), final64=x64.double().numpy(), z32=z32.float().numpy(), z64=z64.double().numpy(),
                        keys=np.array(list(shapes)), shapes=np.array([",".join(str(d) for d in s) for s in shapes.values()]))
    print("max|z|", z64.abs().max().item(), "|ref32-ref64| z", (z32.double() - z64).abs().max().item(), "final x", (x32.double() - x64).abs().max().item())


if __name__ == "__main__":
    main()
