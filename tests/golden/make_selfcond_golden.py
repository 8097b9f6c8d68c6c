"""Golden vectors for the self-conditioning branch (diffusion_cfg.self_condition=True; gcpnet.py:1112-1139, variational_diffusion.py:1363-1386),
produced by the REFERENCE itself.  Off in both production configs, so these are the only pins of that branch.

    python tests/golden/make_selfcond_golden.py       (build container only)

  dyn_full_qm9sc.npz      full-width QM9 network with self-conditioning inputs (weights re-created from tests/synth.py, seed 17)
  sampler_small_qm9sc.npz reduced-width model (weights stored): a teacher-forced forward + a free-running 6-step sample on a noise tape
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(8)


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrs.items() if v is not None}
    p = os.path.join(HERE, name + ".npz")
    np.savez_compressed(p, **out)
    print("wrote", p, os.path.getsize(p) // 1024, "KiB")


def fwd(net, xh, t, bi, sc, dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        net = net.to(dtype)
        mask = torch.ones(len(bi), dtype=torch.bool)
        kw = {} if sc is None else dict(xh_self_cond=sc.to(dtype), x_self_cond=sc.to(dtype))
        with torch.no_grad():
            _, out = net(rh.make_batch(bi, mask), xh.to(dtype), t.to(dtype), **kw)
        return out.float()
    finally:
        torch.set_default_dtype(prev)
        net.to(prev)


def main():
    d = synth.DATASET_DIMS["qm9"]
    F = synth.dims_feat(d)
    # ---- full width ----
    cfgs = rh.load_reference_cfgs("qm9", ())
    cfgs["diffusion_cfg"]["self_condition"] = True
    net = rh.build_reference_dynamics(cfgs, seed=0)
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F)
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref_shapes == shapes and list(ref_shapes) == list(shapes), "synth.dynamics_shapes disagrees with the reference state_dict"
    net.load_state_dict(synth.make_weights(shapes, seed=17))
    xh, t, bi, nn_, _ = synth.make_inputs([5, 19, 3, 11], F, seed=9)
    g = torch.Generator().manual_seed(10)
    sc = torch.randn(xh.shape, generator=g) * 0.7                      # previous estimate: neither centred nor masked in the reference
    save("dyn_full_qm9sc", num_nodes=nn_, xh=xh, t=t, sc=sc, weight_seed=17,
         out32=fwd(net, xh, t, bi, sc, torch.float32), out64=fwd(net, xh, t, bi, sc, torch.float64),
         out32_nosc=fwd(net, xh, t, bi, None, torch.float32))
    # ---- reduced width, with the sampler ----
    cfgs = rh.shrink_cfgs(rh.load_reference_cfgs("qm9", ()))
    cfgs["diffusion_cfg"]["self_condition"] = True
    net = rh.build_reference_dynamics(cfgs, seed=4, weight_scale=0.5)
    ddpm = rh.build_reference_ddpm(cfgs, net, "qm9")
    nn2 = torch.tensor([5, 7, 3, 6])
    bi2 = torch.repeat_interleave(torch.arange(len(nn2)), nn2)
    xh2, t2, _, _, _ = synth.make_inputs(nn2.tolist(), F, seed=5)
    sc2 = torch.randn(xh2.shape, generator=g)
    sd = {("w:" + k): v.clone().float() for k, v in net.state_dict().items()}
    with rh.NoiseTape(1234) as tape, torch.no_grad():
        free, _, _ = ddpm.mol_gen_sample(num_samples=len(nn2), num_nodes=nn2, device="cpu", num_timesteps=6)
    save("sampler_small_qm9sc", num_nodes=nn2, xh=xh2, t=t2, sc=sc2, out32=fwd(net, xh2, t2, bi2, sc2, torch.float32),
         free_T=6, free_seed=1234, free_out=free, free_calls=np.array(tape.calls, dtype=np.int64), **sd)


if __name__ == "__main__":
    main()
