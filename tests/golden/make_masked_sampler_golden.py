"""Sampling WITH MASKED NODES inside the loop, by the REFERENCE itself (build container only):

    python tests/golden/make_masked_sampler_golden.py      ->  tests/golden/sampler_masked_{full,small}_qm9.npz

The unmodified ``mol_gen_sample(..., node_mask=<partial mask>, num_timesteps=10)`` (variational_diffusion.py:1282-1412; the mask enters the
noise, the CoM projections, the network and the decode) on a noise tape, at FULL width (the fused kernels' masked plan inside the loop) and at
reduced width (the module-path loop), fp32 and fp64.  Stored: sizes, mask, seeds, z after every step, the final decode.
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
SIZES, STEPS, WEIGHT_SEED, NOISE_SEED = [5, 9, 3, 12], 10, 71, 1357


def run(dtype, small, mask):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        cfgs = rh.load_reference_cfgs("qm9", ())
        if small:
            synth.apply_variant(cfgs, None)
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
            xh, bi, _ = ddpm.mol_gen_sample(num_samples=len(SIZES), num_nodes=torch.tensor(SIZES), device="cpu", num_timesteps=STEPS, node_mask=mask)
        return xh, torch.stack(zs)
    finally:
        torch.set_default_dtype(prev)


def main():
    assert rh.reference_available()
    N = sum(SIZES)
    mask = torch.ones(N, dtype=torch.bool)
    mask[[2, 7, 8, 20]] = False
    for small in (False, True):
        x32, z32 = run(torch.float32, small, mask)
        x64, z64 = run(torch.float64, small, mask)
        name = f"sampler_masked_{'small' if small else 'full'}_qm9.npz"
        np.savez_compressed(os.path.join(HERE, name), num_nodes=np.array(SIZES), mask=mask.numpy(), steps=STEPS, weight_seed=WEIGHT_SEED, weight_scale=0.25,
                            noise_seed=NOISE_SEED, final32=x32.float().numpy(), final64=x64.double().numpy(), z32=z32.float().numpy(), z64=z64.double().numpy())
        print(name, "max|z|", z64.abs().max().item(), "|ref32-ref64| z", (z32.double() - z64).abs().max().item(), "final x", (x32[:, :3].double() - x64[:, :3]).abs().max().item(),
              "masked rows zero:", bool((x64[~mask] == 0).all()), "discrete equal:", bool((x32[:, 3:].double() == x64[:, 3:]).all()))


if __name__ == "__main__":
    main()
