"""RePaint inpainting of a position-only diffusion model (`inpaint(..., generate_x_only=True)`, variational_diffusion.py:1582-1789) by the reference
with the two crashing tokens of make_inpaint_golden.py repaired in memory -- around a GCPNetDynamics built WITHOUT node features (the only kind
that accepts the [N, 3] latent; make_xonly_golden.py), full width, seed-recreated weights, on a noise tape, fp32 and fp64.

    python tests/golden/make_inpaint_xonly_golden.py      ->  tests/golden/inpaint_xonly_qm9.npz        (build container only)

Two runs: a jump schedule (2 resamplings, jump length 2, 6 steps) and chain frames (3 resamplings, jump length 1, 6 steps, 3 frames)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402
from make_inpaint_golden import repaired_method  # noqa: E402

torch.set_num_threads(4)
SIZES, WEIGHT_SEED, NOISE_SEED = [5, 9, 3, 12], 79, 1357
RUNS = [("jump", dict(num_resamplings=2, jump_length=2, num_timesteps=6, return_frames=1)),
        ("frames", dict(num_resamplings=3, jump_length=1, num_timesteps=6, return_frames=3))]


def inputs():
    nn_ = torch.tensor(SIZES)
    N = int(nn_.sum())
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
    g = torch.Generator().manual_seed(31)
    x = torch.randn((N, 3), generator=g, dtype=torch.float32) * 1.5 + torch.tensor([0.3, -2.0, 1.0], dtype=torch.float32)   # deliberately not centred (fp32 draws whatever the default dtype)
    fixed = torch.zeros(N, dtype=torch.bool)
    fixed[[0, 1, 2, 5, 8, 9, 14, 15, 16, 17, 20, 28]] = True                                  # every molecule has >= 1 fixed node; molecule 2 is all fixed
    return nn_, bi, x, fixed


def run(dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        _, vd, _ = rh.import_reference()
        inpaint = repaired_method(vd, "inpaint", r"(s_array_self_cond = [^\n]*?) / num_denoise_steps", r"\1")
        jump = repaired_method(vd, "sample_p_zt_given_zs", r"alpha_t_given_s\[node_mask\]", "alpha_t_given_s[batch_index]")
        cfgs = rh.load_reference_cfgs("qm9", ())
        synth.apply_variant(cfgs, None)
        cfgs["dataloader_cfg"]["num_atom_types"] = 0
        cfgs["dataloader_cfg"]["include_charges"] = False
        net = rh.build_reference_dynamics(cfgs, seed=0)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(synth.make_weights(shapes, seed=WEIGHT_SEED, scale_2d=0.25))
        net = net.to(dtype)
        ddpm = rh.build_reference_ddpm(cfgs, net, "qm9").to(dtype)
        ddpm.inpaint = types.MethodType(inpaint, ddpm)
        ddpm.sample_p_zt_given_zs = types.MethodType(jump, ddpm)
        nn_, bi, x, fixed = inputs()
        res = {}
        for name, kw in RUNS:
            with rh.NoiseTape(NOISE_SEED) as tape, torch.no_grad():
                res[name] = ddpm.inpaint(molecule=dict(x=x.clone().to(dtype), num_nodes=nn_, batch_index=bi), node_mask_fixed=fixed, generate_x_only=True, **kw)
            res[name + "_draws"] = len(tape.calls)
        return res, shapes
    finally:
        torch.set_default_dtype(prev)


def main():
    assert rh.reference_available()
    r32, shapes = run(torch.float32)
    r64, _ = run(torch.float64)
    nn_, bi, x, fixed = inputs()
    out = dict(num_nodes=nn_.numpy(), x=x.numpy(), fixed=fixed.numpy(), weight_seed=WEIGHT_SEED, weight_scale=0.25, noise_seed=NOISE_SEED,
               keys=np.array(list(shapes)), shapes=np.array([",".join(str(d) for d in s) for s in shapes.values()]))
    for name, kw in RUNS:
        out[f"{name}_out32"], out[f"{name}_out64"] = r32[name].float().numpy(), r64[name].double().numpy()
        out[f"{name}_kw"] = np.array([kw["num_resamplings"], kw["jump_length"], kw["num_timesteps"], kw["return_frames"]])
        out[f"{name}_draws"] = r32[name + "_draws"]
        print(name, tuple(r32[name].shape), "draws", r32[name + "_draws"], "|ref32 - ref64|", (r32[name].double() - r64[name]).abs().max().item(),
              "max|x|", r64[name].abs().max().item())
    np.savez_compressed(os.path.join(HERE, "inpaint_xonly_qm9.npz"), **out)


if __name__ == "__main__":
    main()
