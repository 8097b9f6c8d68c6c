"""Golden vectors for RePaint inpainting (`EquivariantVariationalDiffusion.inpaint`, src/models/components/variational_diffusion.py:1582-1789,
with `sample_p_zt_given_zs` :1163-1201), on the reduced-width QM9 models of `sampler_small_qm9.npz` / `sampler_small_qm9sc.npz` (same weight
seeds; weights not stored again).  -> tests/golden/inpaint_small_qm9.npz

    python tests/golden/make_inpaint_golden.py        (build container only)

NOT a pure reference output.  The reference's `inpaint` raises on every call (UnboundLocalError, :1650: `num_denoise_steps` is used before the
loop that defines it) and `sample_p_zt_given_zs` raises on every call with more nodes than molecules (IndexError, :1177: a [B,1] tensor
indexed with the [N] node mask).  This script imports the reference and repairs exactly those two tokens IN MEMORY before running it:
  :1650  `<zeros> / num_denoise_steps`      ->  `<zeros>`                         (0 / anything; the jump target of the estimate is t = 0)
  :1177  `alpha_t_given_s[node_mask]`       ->  `alpha_t_given_s[batch_index]`    (the gather used by every sibling function)
Everything else -- the schedule, the three noise draws per step and their order, the CoM matching, the frames -- is the reference's own code.
"""
import inspect
import os
import re
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402

torch.set_num_threads(8)


def repaired_method(vd, name, pattern, repl):
    fn = getattr(vd.EquivariantVariationalDiffusion, name)
    src = textwrap.dedent(inspect.getsource(fn))
    src, n = re.subn(pattern, repl, src)
    assert n == 1, (name, n)
    src = "\n".join(line for line in src.split("\n") if not line.startswith("@"))        # typechecked / inference_mode decorators
    ns = dict(vd.__dict__)
    exec(src, ns)
    return ns[name]


def main():
    _, vd, _ = rh.import_reference()
    inpaint = repaired_method(vd, "inpaint", r"(s_array_self_cond = [^\n]*?) / num_denoise_steps", r"\1")
    jump = repaired_method(vd, "sample_p_zt_given_zs", r"alpha_t_given_s\[node_mask\]", "alpha_t_given_s[batch_index]")
    out = {}
    for tag, sc in (("plain", False), ("sc", True)):
        cfgs = rh.shrink_cfgs(rh.load_reference_cfgs("qm9", ()))
        cfgs["diffusion_cfg"]["self_condition"] = sc
        net = rh.build_reference_dynamics(cfgs, seed=4, weight_scale=0.5)
        ddpm = rh.build_reference_ddpm(cfgs, net, "qm9")
        ddpm.inpaint = types.MethodType(inpaint, ddpm)
        ddpm.sample_p_zt_given_zs = types.MethodType(jump, ddpm)
        nn_ = torch.tensor([5, 7, 3, 6])
        N = int(nn_.sum())
        bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
        g = torch.Generator().manual_seed(21)
        x = torch.randn((N, 3), generator=g) * 1.5 + torch.tensor([0.3, -2.0, 1.0])           # deliberately not centred
        one_hot = torch.nn.functional.one_hot(torch.randint(0, 5, (N,), generator=g), 5).float()
        charges = torch.randint(0, 9, (N, 1), generator=g).float()
        fixed = torch.zeros(N, dtype=torch.bool)
        fixed[[0, 1, 2, 5, 8, 11, 12, 13, 14, 15, 20]] = True                                  # every molecule has >= 1 fixed node; molecule 2 is all fixed
        mol = lambda: dict(x=x.clone(), one_hot=one_hot.clone(), charges=charges.clone(), num_nodes=nn_, batch_index=bi)  # noqa: E731
        runs = [("jump", dict(num_resamplings=2, jump_length=2, num_timesteps=6, return_frames=1)),
                ("frames", dict(num_resamplings=3, jump_length=1, num_timesteps=6, return_frames=3))]
        if sc:
            runs = runs[:1]
        for name, kw in runs:
            with rh.NoiseTape(1234) as tape, torch.no_grad():
                res = ddpm.inpaint(molecule=mol(), node_mask_fixed=fixed, **kw)
            out[f"{tag}_{name}_out"] = res.numpy()
            out[f"{tag}_{name}_kw"] = np.array([kw["num_resamplings"], kw["jump_length"], kw["num_timesteps"], kw["return_frames"]])
            out[f"{tag}_{name}_draws"] = len(tape.calls) // 2
            print(tag, name, tuple(res.shape), "draws", len(tape.calls) // 2, "schedule",
                  ddpm.get_repaint_schedule(kw["num_resamplings"], kw["jump_length"], kw["num_timesteps"]))
        out[f"{tag}_weight_check"] = next(iter(net.state_dict().values())).float().numpy()
    out.update(num_nodes=nn_.numpy(), x=x.numpy(), one_hot=one_hot.numpy(), charges=charges.numpy(), fixed=fixed.numpy(), seed=1234)
    np.savez_compressed(os.path.join(HERE, "inpaint_small_qm9.npz"), **out)


if __name__ == "__main__":
    main()
