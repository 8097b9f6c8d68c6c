"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE itself.

Run in the build container only (needs /root/reference; CPU):

    python tests/golden/make_golden.py

The reference modules are imported unmodified under the stubs of ``ref_harness.py``.  The fixtures are
data only: inputs, (for reduced-width cases) weights, and the reference's outputs in fp32 and, as the
high-precision adjudicator, the same code run in fp64.
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

torch.set_num_threads(8)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if v is None:
            continue
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def cfgs_for(case):
    ds, cond = {"qm9": ("qm9", ()), "qm9cond": ("qm9", ("alpha",)), "geom": ("geom", ())}[case]
    return ds, cond, rh.load_reference_cfgs(ds, cond)


def run_ref_forward(net, xh, t, bi, ctx, dtype):
    # the reference hard-wires the *default* dtype in localize/scalarize
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        return _run_ref_forward(net, xh, t, bi, ctx, dtype)
    finally:
        torch.set_default_dtype(prev)
        net.to(prev)


def _run_ref_forward(net, xh, t, bi, ctx, dtype):
    net = net.to(dtype)
    mask = torch.ones(len(bi), dtype=torch.bool)
    caps = {}
    hooks = []

    def cap(name):
        def fn(mod, inp, out):
            caps[name] = out
        return fn

    hooks.append(net.gcp_embedding.register_forward_hook(cap("embed")))
    for l, layer in enumerate(net.interaction_layers):
        hooks.append(layer.register_forward_hook(cap(f"layer{l}")))
        hooks.append(layer.interaction.register_forward_hook(cap(f"mp{l}")))
    with torch.no_grad():
        batch = rh.make_batch(bi, mask, None if ctx is None else ctx.to(dtype))
        _, out = net(batch, xh.to(dtype), t.to(dtype))
    for h in hooks:
        h.remove()
    return out, caps, batch


def function_level():
    """Golden I/O of the small geometric helpers and of single GCP2 / message-passing / interaction calls."""
    gcp, vd, comps = rh.import_reference()
    import importlib

    edm = importlib.import_module("src.datamodules.components.edm_dataset")
    torch.manual_seed(7)
    nn_ = torch.tensor([5, 4, 3])
    bi = torch.repeat_interleave(torch.arange(3), nn_)
    N = int(nn_.sum())
    mask = torch.ones(N, dtype=torch.bool)
    x = torch.randn(N, 3)
    ei = gcp.GCPNetDynamics.get_fully_connected_edge_index(bi, mask)
    b = rh.make_batch(bi, mask)
    b.x = x
    b.edge_index = ei
    e, xi = edm._edge_features(b)
    _, chi0 = edm._node_features(b, edm_sampling=True)
    _, xc = comps.centralize(b, "x", bi, mask, edm=True)
    fr = comps.localize(xc, ei, norm_x_diff=True, node_mask=mask)
    u_e = torch.randn(ei.shape[1], 3, 3)
    u_n = torch.randn(N, 3, 3)
    q_e = comps.scalarize(u_e, ei, fr, node_inputs=False, dim_size=ei.shape[1], node_mask=mask)
    q_n = comps.scalarize(u_n, ei, fr, node_inputs=True, dim_size=N, node_mask=mask)
    vv = torch.randn(N, 6, 3) * 1e-5
    sn = comps.safe_norm(vv.transpose(-1, -2), dim=-2)
    npz("fn_geometry", num_nodes=nn_, x=x, edge_index=ei, e=e, xi=xi, chi0=chi0, x_central=xc, frames=fr,
        u_edge=u_e, u_node=u_n, q_edge=q_e, q_node=q_n, sn_in=vv, sn_out=sn)

    # single GCP2s (edge mode with SiLU, node mode without act, node mode ff) with random weights
    torch.manual_seed(11)
    out = dict(num_nodes=nn_, x=x)
    specs = {
        "edge": dict(dims=((12, 8), (16, 8)), kw=dict(nonlinearities=("silu", "silu"), bottleneck=4), node=False),
        "node": dict(dims=((10, 2), (16, 4)), kw=dict(nonlinearities=(None, None), bottleneck=1), node=True),
        "nodeff": dict(dims=((24, 8), (16, 4)), kw=dict(nonlinearities=(None, None), bottleneck=4, feedforward_out=True), node=True),
        "proj": dict(dims=((16, 4), (7, 0)), kw=dict(nonlinearities=(None, None), bottleneck=1), node=True),
    }
    for name, sp in specs.items():
        (si, vi), (so, vo) = sp["dims"]
        mod = gcp.GCP2((si, vi), (so, vo), vector_gate=True, **sp["kw"]).eval()
        M = N if sp["node"] else ei.shape[1]
        s = torch.randn(M, si)
        v = torch.randn(M, vi, 3)
        with torch.no_grad():
            r = mod(comps.ScalarVector(s, v), ei, fr, node_inputs=sp["node"], node_mask=mask)
        out[f"{name}_s"], out[f"{name}_v"] = s, v
        if vo:
            out[f"{name}_os"], out[f"{name}_ov"] = r[0], r[1]
        else:
            out[f"{name}_os"] = r
        for k, w in mod.state_dict().items():
            out[f"{name}_w_{k}"] = w
    npz("fn_gcp2", **out)


def dyn_small(case):
    ds, cond, cfgs = cfgs_for(case)
    cfgs = rh.shrink_cfgs(cfgs)
    net = rh.build_reference_dynamics(cfgs, seed=3)
    d = synth.DATASET_DIMS[case]
    xh, t, bi, nn_, ctx = synth.make_inputs([5, 7, 3, 6], synth.dims_feat(d), seed=5, n_ctx=d["n_ctx"])
    out32, caps, _ = run_ref_forward(net, xh, t, bi, ctx, torch.float32)
    sd = {("w:" + k): v.clone().float() for k, v in net.state_dict().items()}
    (h_e, chi_e), (e_e, xi_e) = caps["embed"]
    (h0, chi0), x0 = caps["layer0"]
    a_s, a_v = caps["mp0"]
    out64, _, _ = run_ref_forward(net, xh, t, bi, ctx, torch.float64)
    npz(f"dyn_small_{case}", num_nodes=nn_, xh=xh, t=t, ctx=ctx, out32=out32, out64=out64.float(),
        h_embed=h_e, chi_embed=chi_e, e_embed=e_e, xi_embed=xi_e, agg_s0=a_s, agg_v0=a_v,
        h_l0=h0, chi_l0=chi0, x_l0=x0, **sd)


def dyn_full(case):
    ds, cond, cfgs = cfgs_for(case)
    d = synth.DATASET_DIMS[case]
    net = rh.build_reference_dynamics(cfgs, seed=0)
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    ref_shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert ref_shapes == shapes, "synth.dynamics_shapes disagrees with the reference state_dict"
    assert list(ref_shapes) == list(shapes)
    W = synth.make_weights(shapes, seed=17)
    net.load_state_dict(W)
    xh, t, bi, nn_, ctx = synth.make_inputs([5, 19, 3, 11], synth.dims_feat(d), seed=9, n_ctx=d["n_ctx"])
    out32, caps, _ = run_ref_forward(net, xh, t, bi, ctx, torch.float32)
    (h_e, chi_e), (e_e, xi_e) = caps["embed"]
    (h0, chi0), x0 = caps["layer0"]
    a_s, a_v = caps["mp0"]
    L = d["L"]
    (hL, chiL), xL = caps[f"layer{L - 1}"]
    out64, _, _ = run_ref_forward(net, xh, t, bi, ctx, torch.float64)
    npz(f"dyn_full_{case}", num_nodes=nn_, xh=xh, t=t, ctx=ctx, weight_seed=17, out32=out32, out64=out64.float(),
        h_embed=h_e, chi_embed=chi_e, e_embed=e_e, xi_embed=xi_e, agg_s0=a_s, agg_v0=a_v, h_l0=h0, chi_l0=chi0,
        x_l0=x0, h_last=hL, chi_last=chiL, x_last=xL)


def dyn_full_masked(case):
    """Full-width forward with masked nodes (`batch.mask` with False entries: gcpnet.py:1081-1099 zeroed inputs, :1062-1065 no edges,
    components/__init__.py:53-92 masked centroid, gcpnet.py:914-928 re-masking after every layer), run by the REFERENCE in fp32 and fp64."""
    ds, cond, cfgs = cfgs_for(case)
    d = synth.DATASET_DIMS[case]
    net = rh.build_reference_dynamics(cfgs, seed=0)
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    net.load_state_dict(synth.make_weights(shapes, seed=19))
    sizes = [5, 19, 3, 11, 40] if case != "geom" else [5, 44, 3, 70]
    xh, t, bi, nn_, ctx = synth.make_inputs(sizes, synth.dims_feat(d), seed=11, n_ctx=d["n_ctx"])
    g = torch.Generator().manual_seed(23)
    mask = torch.rand(len(bi), generator=g) > 0.25
    for b in range(len(nn_)):                        # every molecule keeps an unmasked atom; one molecule is fully unmasked, one has its last atoms masked
        sel = (bi == b).nonzero().flatten()
        mask[sel[0]] = True
    mask[(bi == 1)] = True
    mask[(bi == 3).nonzero().flatten()[-3:]] = False
    # inputs as a caller provides them: masked rows arbitrary (the network zeroes them), unmasked positions CoM-free per molecule
    for b in range(len(nn_)):
        sel = (bi == b) & mask
        xh[sel, :3] -= xh[sel, :3].mean(0, keepdim=True)
    xin = xh.clone()
    xin[~mask] = 0.0                                  # the reference asserts masked positions ~ 0 inside centralize(edm=True)

    def run(dtype):
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            n2 = net.to(dtype)
            with torch.no_grad():
                batch = rh.make_batch(bi, mask, None if ctx is None else ctx.to(dtype))
                _, out = n2(batch, xin.to(dtype), t.to(dtype))
            return out
        finally:
            torch.set_default_dtype(prev)
            net.to(prev)

    out32 = run(torch.float32)
    out64 = run(torch.float64)
    npz(f"dyn_masked_{case}", num_nodes=nn_, xh=xin, t=t, ctx=ctx, mask=mask, weight_seed=19, out32=out32, out64=out64.float())


def sampler_small(case):
    ds, cond, cfgs = cfgs_for(case)
    cfgs = rh.shrink_cfgs(cfgs)
    net = rh.build_reference_dynamics(cfgs, seed=4, weight_scale=0.5)
    ddpm = rh.build_reference_ddpm(cfgs, net, ds)
    d = synth.DATASET_DIMS[case]
    nn_ = torch.tensor([5, 7, 3, 6])
    B = len(nn_)
    bi = torch.repeat_interleave(torch.arange(B), nn_)
    N = int(nn_.sum())
    mask = torch.ones(N, dtype=torch.bool)
    g = torch.Generator().manual_seed(21)
    ctx_b = torch.randn((B, 1), generator=g) if d["n_ctx"] else None
    sd = {("w:" + k): v.clone().float() for k, v in net.state_dict().items()}
    out = dict(num_nodes=nn_, ctx=ctx_b, gamma=ddpm.gamma.gamma.detach(), **sd)

    # known-answer pins of the schedule algebra
    t1, s1 = torch.full((1, 1), 1.0), torch.full((1, 1), 0.999)
    gt, gs = ddpm.gamma(t1), ddpm.gamma(s1)
    s2, s_, a_ = ddpm.sigma_and_alpha_t_given_s(gt, gs, gt)
    out["pins"] = torch.tensor([s2.item(), s_.item(), a_.item(), ddpm.sigma(gs, gs).item(), ddpm.sigma(gt, gt).item(),
                                ddpm.SNR(-0.5 * ddpm.gamma(torch.zeros(1, 1))).item()])

    # teacher-forced single steps at three points of the schedule
    T = 1000
    for idx, s in enumerate([999, 500, 0]):
        gz = torch.Generator().manual_seed(100 + s)
        z = torch.randn((N, 3 + synth.dims_feat(d)), generator=gz) * (1.0 if s > 100 else 0.3)
        for b in range(B):
            z[bi == b, :3] -= z[bi == b, :3].mean(0, keepdim=True)
        ctx = None if ctx_b is None else ctx_b[bi]
        with rh.NoiseTape(300 + s) as tape, torch.no_grad():
            zs = ddpm.sample_p_zs_given_zt(s=torch.full((B, 1), s / T), t=torch.full((B, 1), (s + 1) / T), z=z,
                                           batch_index=bi, node_mask=mask, context=ctx)
        out[f"tf{idx}_s"], out[f"tf{idx}_z"], out[f"tf{idx}_zs"] = s, z, zs
        out[f"tf{idx}_noise_seed"] = 300 + s

    # free-running: 12 coarse steps + final decode, same noise tape
    with rh.NoiseTape(1234) as tape, torch.no_grad():
        xh, bi2, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cpu", num_timesteps=12, context=ctx_b)
    out["free_T"], out["free_seed"], out["free_out"] = 12, 1234, xh
    out["free_calls"] = np.array(tape.calls, dtype=np.int64)
    npz(f"sampler_small_{case}", **out)


if __name__ == "__main__":
    assert rh.reference_available(), "reference checkout not found"
    if len(sys.argv) > 1 and sys.argv[1] == "masked":          # only the masked-node fixtures (added in round 2)
        for case in ("qm9", "qm9cond", "geom"):
            dyn_full_masked(case)
        sys.exit(0)
    function_level()
    for case in ("qm9", "qm9cond", "geom"):
        dyn_small(case)
        sampler_small(case)
        dyn_full(case)
        dyn_full_masked(case)
