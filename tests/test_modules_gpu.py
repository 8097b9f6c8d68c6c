"""GPU parity tests of the MODULE path: libgcdm_ops.so operators composed by the callable GCP / GCP2 / GCPEmbedding / GCPMessagePassing /
GCPInteractions mirrors (bio-diffusion_amd/gcp_modules.py) -- plug point 3, the non-production configurations of the path's Hydra surface,
and the backward pass.  References: fixtures the imported reference produced (tests/golden/fn_gcp2.npz, dyn_variant_*.npz,
train_full_*.npz) and, for seeded inputs of other sizes, the CPU oracle and its torch autograd."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402
from oracle import gcdm_oracle as O  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
pytestmark = pytest.mark.gpu
DEV = "cuda"


def _graph(num_nodes, x):
    nn_ = torch.as_tensor(num_nodes)
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_).to(DEV)
    ei = pkg.ops.fully_connected_edge_index(nn_, DEV)
    xc = pkg.ops.centralize(x.to(DEV), bi, None)
    return bi, ei, pkg.ops.localize(xc, ei, True)


def test_geometry_operators_match_reference_golden(golden_dir):
    """fully-connected edge list, centralize, localize, scalarize (edge / node mode), safe_norm against the reference's own outputs (fn_geometry.npz)."""
    g = {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, "fn_geometry.npz")).items()}
    bi, ei, fr = _graph(g["num_nodes"], g["x"])
    row, col = O.fully_connected_edges(bi.cpu(), torch.ones(len(bi), dtype=torch.bool))
    assert torch.equal(ei.cpu(), torch.stack((row, col))) and torch.equal(ei.cpu(), g["edge_index"])
    e, xi = pkg.ops.edge_features(g["x"].to(DEV), ei)
    assert (e.cpu() - g["e"]).abs().max().item() <= 1e-6 and (xi.cpu() - g["xi"]).abs().max().item() <= 1e-6
    assert (pkg.ops.orientations(g["x"].to(DEV)).cpu() - g["chi0"]).abs().max().item() <= 1e-6
    assert (pkg.ops.centralize(g["x"].to(DEV), bi, None).cpu() - g["x_central"]).abs().max().item() <= 1e-6
    assert (fr.cpu() - g["frames"]).abs().max().item() <= 1e-6
    graph = pkg.ops.graph_of(ei, len(bi))
    q_edge = pkg.ops.scalarize(g["u_edge"].transpose(-1, -2).contiguous().to(DEV), fr)
    assert (q_edge.cpu() - g["q_edge"]).abs().max().item() <= 2e-6
    q_node = pkg.ops.scalarize(g["u_node"].transpose(-1, -2).contiguous().to(DEV), pkg.ops.mean_frames(fr, graph))
    assert (q_node.cpu() - g["q_node"]).abs().max().item() <= 2e-6
    sn = pkg.ops.safe_norm_pre(g["sn_in"].transpose(-1, -2).contiguous().to(DEV))
    assert torch.allclose(sn.cpu(), g["sn_out"], rtol=2e-6, atol=0)


@pytest.mark.parametrize("name,node,act,ff,vout", [("edge", False, "silu", False, True), ("node", True, None, False, True),
                                                    ("nodeff", True, None, True, True), ("proj", True, None, False, False)])
def test_plug_point_3_gcp2_matches_reference_golden(golden_dir, name, node, act, ff, vout):
    """`module_cfg.selected_GCP(input_dims, output_dims, **flags)(s_maybe_v, edge_index, frames, node_inputs=...)` -- the reference's third plug
    point (gcpnet.py:418-491, called at :522, 615, 1028) -- against single GCP2 evaluations of the reference itself (tests/golden/fn_gcp2.npz:
    edge mode, node mode, node mode with the feed-forward scalar_out, the scalar-only projection)."""
    g = {k: torch.tensor(v) for k, v in np.load(os.path.join(golden_dir, "fn_gcp2.npz")).items()}
    _, ei, fr = _graph(g["num_nodes"], g["x"])
    P = {k[len(name) + 3:]: v for k, v in g.items() if k.startswith(name + "_w_")}
    s, v = g[name + "_s"], g[name + "_v"]
    s_out = g[name + "_os"].shape[1]
    v_out = g[name + "_ov"].shape[1] if vout else 0
    H = P["vector_down.weight"].shape[0]
    bottleneck = v.shape[1] // H if H < v.shape[1] else 1
    mod = pkg.GCP2((s.shape[1], v.shape[1]), (s_out, v_out), nonlinearities=(act, act), feedforward_out=ff, bottleneck=bottleneck)
    mod.load_state_dict(P)
    mod = mod.to(DEV)
    with torch.no_grad():
        r = mod((s.to(DEV), v.to(DEV)), ei, fr, node_inputs=node)
    if vout:
        assert (r[0].cpu() - g[name + "_os"]).abs().max().item() <= 2e-6
        assert (r[1].cpu() - g[name + "_ov"]).abs().max().item() <= 2e-6
    else:
        assert (r.cpu() - g[name + "_os"]).abs().max().item() <= 2e-6


def _variant_net(name):
    cfgs = synth.apply_variant(pkg.default_cfgs("qm9"), name)
    return pkg.GCPNetDynamics(**cfgs), cfgs


@pytest.mark.parametrize("name", list(synth.VARIANTS))
def test_variant_forward_matches_reference_golden(name, golden_dir):
    """Every non-production setting of the path's Hydra surface the reference's modules implement -- GCP v1, frame_gate, sigma_frame_gate,
    vector / frame residuals, no vector gate, ablated frame updates, GCPLayerNorm (pre / post), vector-sum position updates, other numbers
    of message and feed-forward GCPs, other nonlinearities, widths and bottlenecks -- against the reference's own outputs on the same inputs
    and seed-recreated weights (tests/golden/dyn_variant_<name>.npz, fp32 + fp64), with an all-True mask and with masked nodes.
    Bar: |hip - ref32| <= 4 |ref32 - ref64| + 1e-5 max|out|."""
    g = np.load(os.path.join(golden_dir, f"dyn_variant_{name}.npz"))
    net, _ = _variant_net(name)
    shapes = {k: tuple(int(x) for x in s.split(",")) if s else () for k, s in zip(g["keys"].tolist(), g["shapes"].tolist())}
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == shapes and list(net.state_dict()) == list(shapes)
    assert net.fused_unsupported is not None, "a non-production variant must not be routed to the fused kernels"
    net.load_state_dict(synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"])))
    net = net.to(DEV).eval()
    nn_ = torch.tensor(g["num_nodes"])
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_).to(DEV)
    xh, t = torch.tensor(g["xh"]).to(DEV), torch.tensor(g["t"]).to(DEV)
    for tag, mask in (("full", torch.ones(len(bi), dtype=torch.bool)), ("part", torch.tensor(g["mask_part"]))):
        with torch.no_grad():
            _, out = net(dict(batch=bi, mask=mask.to(DEV), props_context=None), xh, t)
        r32, r64 = torch.tensor(g[f"out32_{tag}"]).double(), torch.tensor(g[f"out64_{tag}"])
        bound = 4.0 * (r32 - r64).abs().max().item() + 1e-5 * r64.abs().max().item()
        err = (out.cpu().double() - r32).abs().max().item()
        assert err <= bound, f"{name} / {tag}: |hip - ref32| = {err:.3e} > {bound:.3e}"


@pytest.mark.parametrize("case", ["qm9", "geom"])
def test_module_path_equals_fused_path_and_oracle(case):
    """Production configuration at full width: the module path, the fused kernels and the CPU oracle agree (1e-4 bar; observed ~1e-7)."""
    d = synth.DATASET_DIMS[case]
    net = pkg.GCPNetDynamics(**pkg.default_cfgs(case))
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=3, scale_2d=0.5)
    net.load_state_dict(W)
    net = net.to(DEV).eval()
    xh, t, bi, nn_, _ = synth.make_inputs([5, 9, 3, 12] if case == "qm9" else [7, 44, 3], synth.dims_feat(d), seed=2)
    batch = dict(batch=bi.to(DEV), mask=torch.ones(len(bi), dtype=torch.bool, device=DEV), props_context=None)
    with torch.no_grad():
        net.path = "fused"
        _, out_f = net(batch, xh.to(DEV), t.to(DEV))
        net.path = "modules"
        _, out_m = net(batch, xh.to(DEV), t.to(DEV))
    ocfg = O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"], num_layers=d["L"])
    ref = O.dynamics_forward(W, ocfg, xh, t, bi)
    scale = max(1.0, ref.abs().max().item())
    assert (out_m.cpu() - ref).abs().max().item() <= 1e-4 * scale and (out_f.cpu() - ref).abs().max().item() <= 1e-4 * scale
    assert (out_m - out_f).abs().max().item() <= 1e-5 * scale


def test_module_path_gradients_match_oracle_autograd():
    """The backward pass: d(sum(out * r)) / d(every parameter) of the full-width QM9 network on the HIP operators against torch's
    autograd through the CPU oracle (the same restatement the forward parity rests on) -- relative error per tensor <= 1e-4."""
    d = synth.DATASET_DIMS["qm9"]
    net = pkg.GCPNetDynamics(**pkg.default_cfgs("qm9"))
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=3, scale_2d=0.5)
    net.load_state_dict(W)
    net = net.to(DEV).train()
    xh, t, bi, nn_, _ = synth.make_inputs([5, 9, 3, 12], synth.dims_feat(d), seed=2)
    torch.manual_seed(0)
    r = torch.randn(len(bi), xh.shape[1])
    Wg = {k: v.clone().requires_grad_(True) for k, v in W.items()}
    lo = (O.dynamics_forward(Wg, O.OracleConfig(num_layers=d["L"]), xh, t, bi) * r).sum()
    lo.backward()
    batch = dict(batch=bi.to(DEV), mask=torch.ones(len(bi), dtype=torch.bool, device=DEV), props_context=None)
    _, out = net(batch, xh.to(DEV), t.to(DEV))
    lh = (out * r.to(DEV)).sum()
    lh.backward()
    assert abs(lh.item() - lo.item()) <= 1e-5 * max(1.0, abs(lo.item()))
    params = dict(net.named_parameters())
    worst = 0.0
    for k, v in Wg.items():
        assert params[k].grad is not None, k
        rel = (params[k].grad.cpu() - v.grad).abs().max().item() / max(v.grad.abs().max().item(), 1e-12)
        worst = max(worst, rel)
        assert rel <= 1e-4, (k, rel)
    print(f"worst relative gradient error over {len(Wg)} tensors: {worst:.2e}")


def test_atom_type_table_of_the_embedding_matches_reference_golden(golden_dir):
    """`GCPEmbedding(num_atom_types > 0)` (gcpnet.py:509-512, 569-572): integer atom types through an nn.Embedding table, then the embedding GCPs and the
    GCP layer norms -- outputs and the gradient of a fixed scalar w.r.t. the table against the reference's own (fixture from the imported reference)."""
    g = np.load(os.path.join(golden_dir, "fn_atom_embedding.npz"))
    T = int(g["w_atom_embedding.weight"].shape[0])
    cfg = pkg.default_cfgs("qm9")["module_cfg"]
    mod = pkg.GCPEmbedding((1, 1), (T, 2), (8, 4), (16, 4), num_atom_types=T, cfg=cfg, pre_norm=False, use_gcp_norm=True)
    sd = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")}
    assert set(sd) == set(mod.state_dict()), (sorted(set(sd) ^ set(mod.state_dict())))
    mod.load_state_dict(sd)
    mod = mod.to(DEV).train()

    class B:
        pass
    b = B()
    b.h = torch.from_numpy(g["types"]).to(DEV)
    b.chi, b.e, b.xi, b.f_ij = (torch.from_numpy(g[k]).to(DEV) for k in ("chi", "e", "xi", "frames"))
    b.edge_index = torch.from_numpy(g["edge_index"]).to(DEV)
    (ns, nv), (es, ev) = mod(b)
    for got, key in ((ns, "node_s"), (nv, "node_v"), (es, "edge_s"), (ev, "edge_v")):
        want = torch.from_numpy(g[key])
        assert (got.detach().cpu() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()), key
    (ns * torch.from_numpy(g["r"]).to(DEV)).sum().backward()
    want = torch.from_numpy(g["grad_table"])
    assert (mod.atom_embedding.weight.grad.cpu() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(IndexError):
        pkg.ops.embedding(mod.atom_embedding.weight, torch.tensor([T], device=DEV))


def test_operators_refuse_cpu_tensors():
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.ops.linear(torch.zeros(2, 3), torch.zeros(4, 3))


TRAIN_TERMS = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior", "log_pN")


@pytest.mark.parametrize("case", ["qm9", "geom"])
def test_training_loss_and_gradients_match_reference_autograd(case, golden_dir):
    """the TRAINING objective and its backward pass.  `model.train(); model.training_step(batch)["loss"].backward()` -- the
    mirror of qm9_mol_gen_ddpm.py:340-360 over EquivariantVariationalDiffusion.forward in training mode (variational_diffusion.py:948-1160) --
    on a ragged data-like batch at full width, against the REFERENCE's own loss terms, loss and torch-autograd gradients on the same batch,
    timesteps (one of them 0: the masked-in L_0 branch) and noise tape (tests/golden/train_full_{qm9,geom}.npz, fp32 + fp64):
    every term and the loss within 4 x |ref32 - ref64| + 1e-4 relative; for EVERY parameter tensor the gradient norm and absolute maximum within
    1e-4 relative (+ 4 x the reference's own fp32-vs-fp64 gap); eight tensors compared element by element."""
    g = np.load(os.path.join(golden_dir, f"train_full_{case}.npz"), allow_pickle=False)
    d = synth.DATASET_DIMS[case]
    cfgs = pkg.default_cfgs(case)
    cls = pkg.GEOMMoleculeGenerationDDPM if case == "geom" else pkg.QM9MoleculeGenerationDDPM
    model = cls(**cfgs)
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    assert list(shapes) == g["keys"].tolist()
    model.ddpm.dynamics_network.load_state_dict(synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"])))
    model = model.to(DEV).train()
    nn_ = torch.tensor(g["num_nodes"])
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_).to(DEV)
    N, F = int(nn_.sum()), synth.dims_feat(d)
    tape = O.TapeNoise(int(g["noise_seed"]))
    noise = [torch.cat((tape(N, 3), tape(N, F)), dim=-1)]
    t_int = torch.tensor(g["t_int"]).view(-1, 1)

    def batch():
        return pkg.config.AttrDict(x=torch.tensor(g["x"]).to(DEV), one_hot=torch.tensor(g["one_hot"]).to(DEV), charges=torch.tensor(g["charges"]).to(DEV),
                                   batch=bi, mask=torch.ones(N, dtype=torch.bool, device=DEV), props_context=None)

    # the terms of the diffusion model in training mode
    b = batch()
    b.h = {"categorical": b.one_hot, "integer": b.charges}
    b.num_graphs, b.num_nodes_present = len(nn_), nn_.to(DEV)
    terms = model.ddpm(b, return_loss_info=True, t_int=t_int, noise=noise)
    for name, got in zip(TRAIN_TERMS, terms[:8]):
        w32, w64 = torch.tensor(g[f"{name}_32"]).double(), torch.tensor(g[f"{name}_64"]).double()
        bar = 4 * (w32 - w64).abs() + 1e-4 * w64.abs().clamp(min=1.0)
        assert ((got.detach().double().cpu() - w64).abs() <= bar).all(), name
    # the training step and its backward pass
    model.zero_grad()
    metrics = model.training_step(batch(), t_int=t_int, noise=noise)
    loss = metrics["loss"]
    l32, l64 = float(g["loss_32"]), float(g["loss_64"])
    assert abs(loss.item() - l64) <= 4 * abs(l32 - l64) + 1e-4 * abs(l64), (loss.item(), l64)
    assert all(not v.requires_grad for k, v in metrics.items() if k != "loss")
    loss.backward()
    params = dict(model.ddpm.dynamics_network.named_parameters())
    worst = 0.0
    for i, k in enumerate(shapes):
        gr = params[k].grad
        assert gr is not None and torch.isfinite(gr).all(), k
        for stat, fn in (("grad_norm", lambda v: float(v.double().norm())), ("grad_absmax", lambda v: float(v.double().abs().max()))):
            w32, w64 = float(g[f"{stat}_32"][i]), float(g[f"{stat}_64"][i])
            rel = abs(fn(gr) - w64) / max(w64, 1e-30)
            worst = max(worst, rel)
            assert abs(fn(gr) - w64) <= 4 * abs(w32 - w64) + 1e-4 * w64, (k, stat, fn(gr), w64)
    full = [k[len("grad_64::"):] for k in g.files if k.startswith("grad_64::")]
    assert len(full) >= 6
    for k in full:
        w32, w64 = torch.tensor(g[f"grad_32::{k}"]).double(), torch.tensor(g[f"grad_64::{k}"])
        bar = 4 * (w32 - w64).abs().max().item() + 1e-4 * w64.abs().max().item()
        assert (params[k].grad.double().cpu() - w64).abs().max().item() <= bar, k
    print(f"training step {case}: loss {loss.item():.6f} (reference {l64:.6f}); worst relative error of a gradient norm / abs-max over {len(shapes)} tensors: {worst:.2e}")


def test_training_step_updates_the_weights_the_sampler_uses():
    """One optimiser step on the module path is seen by the fused sampling kernels (the packed weights are re-uploaded when a parameter's
    version changes): the evaluation-mode forward before and after the step differ, and the new one equals the module path's."""
    cfgs = pkg.default_cfgs("qm9")
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    d = synth.DATASET_DIMS["qm9"]
    model.ddpm.dynamics_network.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=5, scale_2d=0.5))
    model = model.to(DEV)
    net = model.ddpm.dynamics_network
    xh, t, bi, nn_, _ = synth.make_inputs([6, 11, 4], synth.dims_feat(d), seed=4)
    batch = dict(batch=bi.to(DEV), mask=torch.ones(len(bi), dtype=torch.bool, device=DEV), props_context=None)
    model.eval()
    with torch.no_grad():
        before = net(batch, xh.to(DEV), t.to(DEV))[1].clone()
    model.train()
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    N = len(bi)
    data = pkg.config.AttrDict(x=xh[:, :3].to(DEV), one_hot=torch.nn.functional.one_hot(torch.arange(N) % 5, 5).float().to(DEV),
                               charges=torch.ones(N, device=DEV), batch=bi.to(DEV), mask=torch.ones(N, dtype=torch.bool, device=DEV))
    model.training_step(data, t_int=torch.tensor([[500], [20], [900]]))["loss"].backward()
    opt.step()
    model.eval()
    with torch.no_grad():
        after = net(batch, xh.to(DEV), t.to(DEV))[1]
        net.path = "modules"
        after_m = net(batch, xh.to(DEV), t.to(DEV))[1]
    assert (after - before).abs().max().item() > 1e-5
    assert (after - after_m).abs().max().item() <= 1e-5 * max(1.0, after.abs().max().item())


@pytest.mark.parametrize("width", ["full", "small"])
def test_sampling_with_masked_nodes_matches_reference_golden(width, golden_dir):
    """`mol_gen_sample(..., node_mask=<partial mask>)` -- masked nodes INSIDE the sampling loop (noise, CoM projections, network, decode;
    variational_diffusion.py:1282-1412) -- against the reference's own 10-step run on the same noise tape (make_masked_sampler_golden.py).
    "full": production width, the network evaluations run on the fused kernels with the masked plan; "small": reduced width, the module path.
    z after every step (all rows) within 4 |ref32 - ref64| + 1e-4 max|z|; the decode has exactly zero masked rows and equal discrete outputs."""
    g = np.load(os.path.join(golden_dir, f"sampler_masked_{width}_qm9.npz"))
    cfgs = pkg.default_cfgs("qm9")
    if width == "small":
        synth.apply_variant(cfgs, None)
    net = pkg.GCPNetDynamics(**cfgs)
    assert (net.fused_unsupported is None) == (width == "full")
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    net.load_state_dict(synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"])))
    net = net.to(DEV)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(DEV).eval()
    nn_, mask, steps = torch.tensor(g["num_nodes"]), torch.tensor(g["mask"]), int(g["steps"])
    N, F = int(nn_.sum()), 6
    tape = O.TapeNoise(int(g["noise_seed"]))
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1).to(DEV) for _ in range(steps + 2)]
    zs = []
    out, bi, m2 = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=DEV, num_timesteps=steps, node_mask=mask.to(DEV),
                                      noise_fn=lambda k: draws[k], step_callback=lambda s, z: zs.append(z.detach().cpu().clone()))
    z32, z64 = torch.tensor(g["z32"]).double(), torch.tensor(g["z64"])
    assert len(zs) == steps
    for i in range(steps):
        bound = 4.0 * (z32[i] - z64[i]).abs().max().item() + 1e-4 * z64[i].abs().max().item()
        assert (zs[i].double() - z32[i]).abs().max().item() <= bound, i        # (masked rows included: the reference's latent is NOT zero there, only its decode)
    f32, f64 = torch.tensor(g["final32"]).double(), torch.tensor(g["final64"])
    out = out.cpu()
    bound = 4.0 * (f32[:, :3] - f64[:, :3]).abs().max().item() + 1e-4 * f64[:, :3].abs().max().item()
    assert (out[:, :3].double() - f32[:, :3]).abs().max().item() <= bound
    agree = f32[:, 3:] == f64[:, 3:]
    assert torch.equal(out[:, 3:].double()[agree], f32[:, 3:][agree]) and bool((out[~mask] == 0).all())


def _xonly_cfgs():
    cfgs = pkg.default_cfgs("qm9")
    synth.apply_variant(cfgs, None)
    cfgs["dataloader_cfg"]["num_atom_types"] = 0
    cfgs["dataloader_cfg"]["include_charges"] = False
    return cfgs


def test_position_only_sampling_matches_reference_golden(golden_dir):
    """`mol_gen_sample(..., generate_x_only=True)` -- z = z_x, centred x-noise only, [N, 3] positions out (variational_diffusion.py:795-836, 840-907,
    1282-1412) -- around a dynamics network built without node features, against the reference's own 12-step run on the same noise tape
    (make_xonly_golden.py): z after every step within 4 |ref32 - ref64| + 1e-4 max|z|, the final positions likewise."""
    g = np.load(os.path.join(golden_dir, "sampler_xonly_qm9.npz"))
    cfgs = _xonly_cfgs()
    net = pkg.GCPNetDynamics(**cfgs)
    assert net.fused_unsupported is not None
    want = {k: tuple(int(x) for x in sh.split(",")) if sh else () for k, sh in zip(g["keys"].tolist(), g["shapes"].tolist())}
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert shapes == want and list(shapes) == list(want)
    net.load_state_dict(synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"])))
    net = net.to(DEV)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(DEV).eval()
    nn_, steps = torch.tensor(g["num_nodes"]), int(g["steps"])
    N = int(nn_.sum())
    tape = O.TapeNoise(int(g["noise_seed"]))
    draws = [tape(N, 3).to(DEV) for _ in range(steps + 2)]
    zs = []
    out, bi, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=DEV, num_timesteps=steps, generate_x_only=True,
                                     noise_fn=lambda k: draws[k], step_callback=lambda s, z: zs.append(z.detach().cpu().clone()))
    z32, z64 = torch.tensor(g["z32"]).double(), torch.tensor(g["z64"])
    assert len(zs) == steps and tuple(out.shape) == (N, 3)
    for i in range(steps):
        bound = 4.0 * (z32[i] - z64[i]).abs().max().item() + 1e-4 * z64[i].abs().max().item()
        assert (zs[i].double() - z32[i]).abs().max().item() <= bound, i
    f32, f64 = torch.tensor(g["final32"]).double(), torch.tensor(g["final64"])
    assert (out.cpu().double() - f32).abs().max().item() <= 4.0 * (f32 - f64).abs().max().item() + 1e-4 * f64.abs().max().item()
    # a network WITH node features cannot take the [N, 3] latent (neither can the reference's)
    full = pkg.default_cfgs("qm9")
    ddpm_full = pkg.EquivariantVariationalDiffusion(pkg.GCPNetDynamics(**full).to(DEV), full["diffusion_cfg"], full["dataloader_cfg"], pkg.dataset_info("qm9")).to(DEV).eval()
    with pytest.raises(ValueError, match="without node features"):
        ddpm_full.mol_gen_sample(num_samples=2, num_nodes=torch.tensor([3, 4]), device=DEV, num_timesteps=2, generate_x_only=True)


def test_position_only_inpainting_matches_reference_golden(golden_dir):
    """`inpaint(..., generate_x_only=True)` (variational_diffusion.py:1582-1789 with the two repairs of make_inpaint_golden.py) around a dynamics network
    built without node features, against the repaired reference's own runs on the same noise tape (make_inpaint_xonly_golden.py): a jump schedule and
    chain frames, within 4 |ref32 - ref64| + 1e-4 max|x|; the caller's molecule is left untouched (the reference shifts it in place)."""
    g = np.load(os.path.join(golden_dir, "inpaint_xonly_qm9.npz"))
    cfgs = _xonly_cfgs()
    net = pkg.GCPNetDynamics(**cfgs)
    want = {k: tuple(int(x) for x in sh.split(",")) if sh else () for k, sh in zip(g["keys"].tolist(), g["shapes"].tolist())}
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert shapes == want and list(shapes) == list(want)
    net.load_state_dict(synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"])))
    net = net.to(DEV)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(DEV).eval()
    nn_ = torch.tensor(g["num_nodes"])
    N = int(nn_.sum())
    x, fixed = torch.tensor(g["x"]), torch.tensor(g["fixed"])
    for name in ("jump", "frames"):
        R, J, T, Fr = (int(v) for v in g[f"{name}_kw"])
        tape = O.TapeNoise(int(g["noise_seed"]))
        seen = []
        def noise_fn(k, _t=tape, _s=seen):
            assert k == len(_s)
            _s.append(k)
            return _t(N, 3)
        mol = dict(x=x.to(DEV), num_nodes=nn_)
        out = ddpm.inpaint(mol, fixed.to(DEV), num_resamplings=R, jump_length=J, num_timesteps=T, return_frames=Fr, generate_x_only=True, noise_fn=noise_fn).cpu().double()
        r32, r64 = torch.tensor(g[f"{name}_out32"]).double(), torch.tensor(g[f"{name}_out64"])
        assert out.shape == r32.shape and len(seen) == int(g[f"{name}_draws"])          # the reference's number of randn calls
        bound = 4.0 * (r32 - r64).abs().max().item() + 1e-4 * r64.abs().max().item()
        assert (out - r32).abs().max().item() <= bound, name
        assert torch.equal(mol["x"].cpu(), x)
    # seeded device noise: reproducible, seed-dependent
    mol = dict(x=x.to(DEV), num_nodes=nn_)
    a = ddpm.inpaint(mol, fixed.to(DEV), num_resamplings=2, jump_length=2, num_timesteps=6, generate_x_only=True, seed=3)
    b = ddpm.inpaint(mol, fixed.to(DEV), num_resamplings=2, jump_length=2, num_timesteps=6, generate_x_only=True, seed=3)
    c = ddpm.inpaint(mol, fixed.to(DEV), num_resamplings=2, jump_length=2, num_timesteps=6, generate_x_only=True, seed=4)
    assert torch.equal(a, b) and torch.isfinite(a).all() and not torch.equal(a, c) and tuple(a.shape) == (N, 3)


def test_module_path_sampling_loop_matches_oracle():
    """The general sampling loop (reference-signature sample_p_zs_given_zt / sample_p_xh_given_z0 of this package, network on the module
    path) against the oracle's mol_gen_sample on the same tape: production configuration forced onto the module path, 10 steps."""
    d = synth.DATASET_DIMS["qm9"]
    cfgs = pkg.default_cfgs("qm9")
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=19, scale_2d=0.25)
    net.load_state_dict(W)
    net = net.to(DEV)
    net.path = "modules"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(DEV).eval()
    nn_ = torch.tensor([6, 11, 4])
    N, F, Tp = int(nn_.sum()), 6, 10
    want, bi = O.mol_gen_sample(W, O.OracleConfig(num_layers=d["L"]), nn_, O.TapeNoise(5), num_timesteps=Tp)
    tape = O.TapeNoise(5)
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1).to(DEV) for _ in range(Tp + 2)]
    out, bi2, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=DEV, num_timesteps=Tp, noise_fn=lambda k: draws[k])
    out = out.cpu()
    scale = max(1.0, want[:, :3].abs().max().item())
    assert torch.equal(bi2.cpu(), bi)
    assert (out[:, :3] - want[:, :3]).abs().max().item() <= 1e-4 * scale and torch.equal(out[:, 3:], want[:, 3:])
