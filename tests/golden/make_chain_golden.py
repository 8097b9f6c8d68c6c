"""Golden vectors for sampling with intermediate frames (`return_frames > 1`, the chain-visualisation mode of
`mol_gen_sample`, src/models/components/variational_diffusion.py:1282-1412) and with `fix_noise=True` (:1323-1325, 832-834), produced by the REFERENCE itself on the reduced-width QM9
model of `sampler_small_qm9.npz` (same weight seed; weights not stored again).  -> tests/golden/chain_small_qm9.npz

    python tests/golden/make_chain_golden.py        (build container only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402

torch.set_num_threads(8)


def main():
    cfgs = rh.shrink_cfgs(rh.load_reference_cfgs("qm9", ()))
    net = rh.build_reference_dynamics(cfgs, seed=4, weight_scale=0.5)
    ddpm = rh.build_reference_ddpm(cfgs, net, "qm9")
    nn_ = torch.tensor([5, 7, 3, 6])
    with rh.NoiseTape(1234) as tape, torch.no_grad():
        frames, bi, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cpu", num_timesteps=12, return_frames=4)
    with rh.NoiseTape(1234) as tape2, torch.no_grad():
        fixed, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cpu", num_timesteps=12, fix_noise=True)
    out = dict(num_nodes=nn_.numpy(), T=12, return_frames=4, seed=1234, frames=frames.numpy(), fix_noise_out=fixed.numpy(),
               weight_check=next(iter(net.state_dict().values())).float().numpy())
    np.savez_compressed(os.path.join(HERE, "chain_small_qm9.npz"), **out)
    print("frames", tuple(frames.shape), "calls", len(tape.calls))


if __name__ == "__main__":
    main()
