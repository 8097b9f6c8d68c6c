"""Pins the CPU oracle (oracle/gcdm_oracle.py) to golden vectors produced by the REFERENCE itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import synth
from oracle import gcdm_oracle as O

CASES = ["qm9", "qm9cond", "geom"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.tensor(z[k]) for k in z.files}


def weights_of(g):
    return {k[2:]: v for k, v in g.items() if k.startswith("w:")}


def cfg_for(case, L):
    d = synth.DATASET_DIMS[case]
    return O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"],
                          num_context=d["n_ctx"], num_layers=L, norm_values=d["norm_values"])


def test_geometry_functions(golden_dir):
    g = load(golden_dir, "fn_geometry")
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    mask = torch.ones(len(bi), dtype=torch.bool)
    row, col = O.fully_connected_edges(bi, mask)
    assert torch.equal(torch.stack((row, col)), g["edge_index"])
    e, xi = O.edge_features(g["x"], row, col)
    assert torch.equal(e, g["e"]) and torch.allclose(xi, g["xi"], atol=1e-7)
    assert torch.allclose(O.orientations(g["x"]), g["chi0"], atol=1e-7)
    xc = O.centralize(g["x"], bi, len(g["num_nodes"]), mask)
    assert torch.allclose(xc, g["x_central"], atol=1e-7)
    fr = O.localize(xc, row, col)
    assert torch.allclose(fr, g["frames"], atol=1e-7)
    assert torch.allclose(O.scalarize(g["u_edge"], row, fr, False, len(row)), g["q_edge"], atol=1e-6)
    assert torch.allclose(O.scalarize(g["u_node"], row, fr, True, len(bi)), g["q_node"], atol=1e-6)
    assert torch.allclose(O.safe_norm(g["sn_in"].transpose(-1, -2), dim=-2), g["sn_out"], rtol=1e-6, atol=0)


@pytest.mark.parametrize("name,node,act,ff,vout", [("edge", False, "silu", False, True), ("node", True, None, False, True),
                                                    ("nodeff", True, None, True, True), ("proj", True, None, False, False)])
def test_single_gcp2(golden_dir, name, node, act, ff, vout):
    g = load(golden_dir, "fn_gcp2")
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    mask = torch.ones(len(bi), dtype=torch.bool)
    row, col = O.fully_connected_edges(bi, mask)
    fr = O.localize(O.centralize(g["x"], bi, len(g["num_nodes"]), mask), row, col)
    P = {k[len(name) + 3:]: v for k, v in g.items() if k.startswith(name + "_w_")}
    r = O.gcp2(P, "", g[name + "_s"], g[name + "_v"], row, fr, node, act, vout, feedforward_out=ff)
    if vout:
        assert torch.allclose(r[0], g[name + "_os"], atol=2e-6)
        assert torch.allclose(r[1], g[name + "_ov"], atol=2e-6)
    else:
        assert torch.allclose(r, g[name + "_os"], atol=2e-6)


@pytest.mark.parametrize("case", CASES)
def test_dynamics_small(golden_dir, case):
    g = load(golden_dir, f"dyn_small_{case}")
    P = weights_of(g)
    cfg = cfg_for(case, O.infer_num_layers(P))
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    out, inter = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi, None, g.get("ctx"), return_intermediates=True)
    assert torch.allclose(inter["h_embed"], g["h_embed"], atol=1e-5)
    assert torch.allclose(inter["e"], g["e_embed"], atol=1e-5)
    assert torch.allclose(inter["xi"], g["xi_embed"], atol=1e-5)
    assert torch.allclose(inter["h_0"], g["h_l0"], atol=1e-5)
    assert torch.allclose(inter["x_0"], g["x_l0"], atol=1e-5)
    assert (out - g["out32"]).abs().max().item() <= 1e-5
    # fp64 twin: the oracle in fp64 reproduces the reference in fp64 to (stored) fp32 precision
    P64 = {k: v.double() for k, v in P.items()}
    ctx = g.get("ctx")
    out64 = O.dynamics_forward(P64, cfg, g["xh"].double(), g["t"].double(), bi, None, None if ctx is None else ctx.double())
    assert (out64.float() - g["out64"]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("case", CASES)
def test_schedule_and_sampler_small(golden_dir, case):
    g = load(golden_dir, f"sampler_small_{case}")
    P = weights_of(g)
    cfg = cfg_for(case, O.infer_num_layers(P))
    gam = O.gamma_table(cfg)
    assert torch.equal(gam, g["gamma"])
    # SURVEY A.5 known answers
    kn = {0: -11.5129156, 1: -11.330595, 2: -10.9251308, 500: -0.251309335, 998: 10.5586309, 999: 11.1767302, 1000: 11.512516}
    for i, v in kn.items():
        assert abs(gam[i].item() - v) < 2e-6 * max(1, abs(v))
    s2, s_, a_ = O.sigma_and_alpha_t_given_s(gam[1000].view(1, 1), gam[999].view(1, 1))
    pins = g["pins"]
    assert torch.allclose(torch.stack([s2.squeeze(), s_.squeeze(), a_.squeeze()]), pins[:3], rtol=1e-6)
    assert abs(pins[0].item() - 0.285220444) < 1e-7 and abs(pins[5].item() - 0.00316229323) < 1e-9

    nn_ = g["num_nodes"]
    bi = O.num_nodes_to_batch_index(nn_)
    B = len(nn_)
    mask = torch.ones(len(bi), dtype=torch.bool)
    ctx_b = g.get("ctx")
    ctx = None if ctx_b is None else ctx_b[bi]
    for idx in range(3):
        s = int(g[f"tf{idx}_s"])
        zs, _ = O.sample_p_zs_given_zt(P, cfg, gam, s / 1000, (s + 1) / 1000, g[f"tf{idx}_z"], bi, B, mask, ctx,
                                       O.TapeNoise(int(g[f"tf{idx}_noise_seed"])))
        assert (zs - g[f"tf{idx}_zs"]).abs().max().item() <= 1e-5
    out, _ = O.mol_gen_sample(P, cfg, nn_, O.TapeNoise(int(g["free_seed"])), context=ctx_b, num_timesteps=int(g["free_T"]))
    ref = g["free_out"]
    assert (out[:, :3] - ref[:, :3]).abs().max().item() <= 1e-4 * max(1.0, ref[:, :3].abs().max().item())
    assert torch.equal(out[:, 3:], ref[:, 3:])


def test_sampling_chain_frames_small(golden_dir):
    """mol_gen_sample(return_frames=4): intermediate un-normalised frames + final decode vs the reference's own output."""
    gw = load(golden_dir, "sampler_small_qm9")
    g = load(golden_dir, "chain_small_qm9")
    P = weights_of(gw)
    assert torch.equal(P["gcp_embedding.edge_embedding.vector_down.weight"], g["weight_check"])
    cfg = cfg_for("qm9", O.infer_num_layers(P))
    frames, _ = O.mol_gen_sample(P, cfg, g["num_nodes"], O.TapeNoise(int(g["seed"])), num_timesteps=int(g["T"]), return_frames=int(g["return_frames"]))
    ref = g["frames"]
    assert frames.shape == ref.shape
    scale = max(1.0, ref.abs().max().item())
    assert (frames[1:] - ref[1:]).abs().max().item() <= 1e-4 * scale
    assert (frames[0, :, :3] - ref[0, :, :3]).abs().max().item() <= 1e-4 * scale and torch.equal(frames[0, :, 3:], ref[0, :, 3:])
    # fix_noise=True: x-noise centred over the whole flat batch
    fixed, _ = O.mol_gen_sample(P, cfg, g["num_nodes"], O.TapeNoise(int(g["seed"])), num_timesteps=int(g["T"]), fix_noise=True)
    rf = g["fix_noise_out"]
    assert (fixed[:, :3] - rf[:, :3]).abs().max().item() <= 1e-4 * max(1.0, rf[:, :3].abs().max().item()) and torch.equal(fixed[:, 3:], rf[:, 3:])
    assert (rf[:, :3] - ref[0, :, :3]).abs().max().item() > 1e-3       # and it is a different sample than the per-molecule centring gives


def test_self_conditioning_branch(golden_dir):
    """diffusion_cfg.self_condition=True (gcpnet.py:1112-1139; sampler :1363-1386) vs the reference's own outputs: full-width forward with a
    previous estimate, without one (zeros), and a free-running sample of the reduced-width model (two network evaluations per step)."""
    g = load(golden_dir, "dyn_full_qm9sc")
    d = synth.DATASET_DIMS["qm9"]
    F_ = synth.dims_feat(d)
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F_), seed=int(g["weight_seed"]))
    cfg = cfg_for("qm9", d["L"])
    cfg.self_condition = True
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    out = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi, xh_self_cond=g["sc"])
    assert (out - g["out32"]).abs().max().item() <= 1e-5 and (out - g["out64"]).abs().max().item() <= 1e-5
    out0 = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi)
    assert (out0 - g["out32_nosc"]).abs().max().item() <= 1e-5
    assert (g["out32"] - g["out32_nosc"]).abs().max().item() > 1e-3              # the self-conditioning input matters
    gs = load(golden_dir, "sampler_small_qm9sc")
    Ps = weights_of(gs)
    cfgs_ = cfg_for("qm9", O.infer_num_layers(Ps))
    cfgs_.self_condition = True
    bis = O.num_nodes_to_batch_index(gs["num_nodes"])
    outs = O.dynamics_forward(Ps, cfgs_, gs["xh"], gs["t"], bis, xh_self_cond=gs["sc"])
    assert (outs - gs["out32"]).abs().max().item() <= 1e-5
    free, _ = O.mol_gen_sample(Ps, cfgs_, gs["num_nodes"], O.TapeNoise(int(gs["free_seed"])), num_timesteps=int(gs["free_T"]))
    ref = gs["free_out"]
    assert (free[:, :3] - ref[:, :3]).abs().max().item() <= 1e-4 * max(1.0, ref[:, :3].abs().max().item()) and torch.equal(free[:, 3:], ref[:, 3:])


def test_mol_gen_optimize_small(golden_dir):
    """Property-guided optimisation loop (variational_diffusion.py:1416-1546) vs the reference's own outputs, both time normalisations."""
    gw = load(golden_dir, "sampler_small_qm9cond")          # same reduced-width weights (weight seed 4)
    g = load(golden_dir, "optimize_small_qm9cond")
    P = weights_of(gw)
    assert torch.equal(P["gcp_embedding.edge_embedding.vector_down.weight"], g["weight_check"])     # first state-dict entry of the same model
    cfg = cfg_for("qm9cond", O.infer_num_layers(P))
    for tag in ("a", "b"):
        out, _ = O.mol_gen_optimize(P, cfg, g["x"], g["h"], g["num_nodes"], O.TapeNoise(int(g["noise_seed"])), context=g["ctx"],
                                    num_timesteps=int(g[f"{tag}_T"]), norm_with_original_timesteps=bool(int(g[f"{tag}_orig"])))
        ref = g[f"{tag}_out"]
        assert (out[:, :3] - ref[:, :3]).abs().max().item() <= 1e-4 * max(1.0, ref[:, :3].abs().max().item())
        assert torch.equal(out[:, 3:], ref[:, 3:])
    # the two normalisations really are different trajectories
    assert (g["a_out"][:, :3] - g["b_out"][:, :3]).abs().max().item() > 1e-3
    # chain frames (return_frames = 5, :1490-1497, 1540-1546): every frame of the reference's [5, N, 3 + F] result; frame 0 = the decoded sample
    fr, _ = O.mol_gen_optimize(P, cfg, g["x"], g["h"], g["num_nodes"], O.TapeNoise(int(g["noise_seed"])), context=g["ctx"],
                               num_timesteps=int(g["c_T"]), norm_with_original_timesteps=bool(int(g["c_orig"])), return_frames=int(g["c_frames"]))
    ref = g["c_out"]
    assert fr.shape == ref.shape == (5, int(g["num_nodes"].sum()), 8)
    assert (fr - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert torch.equal(fr[0, :, 3:], ref[0, :, 3:])                                 # one-hot of the decode
    # (frame 0 of a chain is the decoded sample WITHOUT the CoG re-projection the frame-less run a_out gets, reference :1389: the two may differ)


def test_repaint_schedule_matches_reference(golden_dir):
    """get_repaint_schedule (variational_diffusion.py:1548-1578) on a grid of 192 (resamplings, jump_length, T) triples."""
    import json
    cases = json.load(open(os.path.join(golden_dir, "repaint_schedule.json")))
    assert len(cases) == 192
    for c in cases:
        assert O.get_repaint_schedule(c["resamplings"], c["jump_length"], c["num_timesteps"]) == c["schedule"], c


def test_inpaint_small(golden_dir):
    """RePaint inpainting loop (variational_diffusion.py:1582-1789) vs the reference run with its two crashing tokens repaired in memory
    (tests/golden/make_inpaint_golden.py): jump_length 2, chain frames with jump_length 1, and the self-conditioning variant."""
    g = load(golden_dir, "inpaint_small_qm9")
    for tag, wfile, runs in (("plain", "sampler_small_qm9", ("jump", "frames")), ("sc", "sampler_small_qm9sc", ("jump",))):
        P = weights_of(load(golden_dir, wfile))
        assert torch.equal(P["gcp_embedding.edge_embedding.vector_down.weight"], g[f"{tag}_weight_check"])
        cfg = cfg_for("qm9", O.infer_num_layers(P))
        cfg.self_condition = tag == "sc"
        for run in runs:
            r, j, T, frames = (int(v) for v in g[f"{tag}_{run}_kw"])
            tape = O.TapeNoise(int(g["seed"]))
            draws = [0]
            def noise(n, k, dtype=torch.float32, _t=tape, _d=draws):
                _d[0] += 1
                return _t(n, k, dtype)
            out = O.inpaint(P, cfg, g["x"], g["one_hot"], g["charges"], g["num_nodes"], g["fixed"], noise, num_resamplings=r, jump_length=j,
                            return_frames=frames, num_timesteps=T)
            ref = g[f"{tag}_{run}_out"]
            assert out.shape == ref.shape and draws[0] // 2 == int(g[f"{tag}_{run}_draws"])
            last, lref = (out, ref) if frames == 1 else (out[0], ref[0])
            scale = max(1.0, ref.abs().max().item())
            assert (last[:, :3] - lref[:, :3]).abs().max().item() <= 1e-4 * scale and torch.equal(last[:, 3:], lref[:, 3:])
            if frames > 1:
                assert (out[1:] - ref[1:]).abs().max().item() <= 1e-4 * scale and ref[1:].abs().max().item() > 0


@pytest.mark.parametrize("case", CASES)
def test_dynamics_full_width(golden_dir, case):
    """Full-width production architecture; weights re-created from the seed recipe (tests/synth.py)."""
    g = load(golden_dir, f"dyn_full_{case}")
    d = synth.DATASET_DIMS[case]
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)),
                           seed=int(g["weight_seed"]))
    cfg = cfg_for(case, d["L"])
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    out, inter = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi, None, g.get("ctx"), return_intermediates=True)
    assert torch.allclose(inter["h_embed"], g["h_embed"], atol=1e-5)
    assert torch.allclose(inter["h_0"], g["h_l0"], atol=2e-5)
    assert (out - g["out32"]).abs().max().item() <= 2e-5
    assert (out - g["out64"]).abs().max().item() <= 1e-4


def test_f16x3_emulation_matches_fp32_accuracy(golden_dir):
    """Numerics of the split-precision scheme the HIP edge kernel uses by default (x = hi + 2^-11 lo', three f16 products, fp32
    accumulation), emulated inside the oracle: against the reference's fp64 outputs it is as accurate as plain fp32."""
    import torch.nn.functional as F
    g = load(golden_dir, "dyn_full_qm9")
    d = synth.DATASET_DIMS["qm9"]
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]))
    cfg = cfg_for("qm9", d["L"])
    bi = O.num_nodes_to_batch_index(g["num_nodes"])

    def split(x):
        hi = x.to(torch.float16)
        return hi.float(), ((x - hi.float()) * 2048.0).to(torch.float16).float()

    orig = F.linear

    def linear_x3(x, w, b=None):
        if w.shape[0] < 32:
            return orig(x, w, b)
        xh, xl = split(x)
        wh, wl = split(w)
        y = xh @ wh.T + (xh @ wl.T + xl @ wh.T) * (1.0 / 2048.0)
        return y if b is None else y + b

    O.F.linear = linear_x3
    try:
        out = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi)
    finally:
        O.F.linear = orig
    err_x3 = (out - g["out64"]).abs().max().item()
    err_f32 = (g["out32"] - g["out64"]).abs().max().item()
    assert err_x3 <= 3e-6 and err_x3 <= 4 * err_f32 + 1e-6


def test_oracle_follows_long_horizon_golden_first_50_steps(golden_dir):
    """The oracle's free-running sampler against the reference's own 1000-step run (tests/golden/long_full_qm9.npz, make_long_golden.py):
    the first 50 of the 1000 steps at full width on the same tape (the whole run is compared on the GPU, tests/test_gpu_parity.py)."""
    g = np.load(os.path.join(golden_dir, "long_full_qm9.npz"))
    d = synth.DATASET_DIMS["qm9"]
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]),
                           scale_2d=float(g["weight_scale"]))
    cfg = O.OracleConfig(num_layers=d["L"])
    nn_ = torch.tensor(g["num_nodes"])
    B = len(nn_)
    bi = O.num_nodes_to_batch_index(nn_)
    mask = torch.ones_like(bi).bool()
    gam = O.gamma_table(cfg)
    noise = O.TapeNoise(int(g["noise_seed"]))
    z = O.sample_combined_noise(noise, bi, B, mask, cfg.num_node_scalar_features, torch.float32)
    for s in range(999, 949, -1):
        z, _ = O.sample_p_zs_given_zt(W, cfg, gam, s / 1000, (s + 1) / 1000, z, bi, B, mask, None, noise)
        if s in (999, 990, 950):
            r32, r64 = torch.tensor(g[f"z32_{s}"]).double(), torch.tensor(g[f"z64_{s}"])
            bound = 4.0 * (r32 - r64).abs().max().item() + 1e-4 * r64.abs().max().item()
            assert (z.double() - r32).abs().max().item() <= bound, s


@pytest.mark.parametrize("fixture,steps", [("long_geom8.npz", 3), ("long_config0_qm9.npz", 1), ("long_cond6_qm9.npz", 3)])
def test_oracle_follows_the_round4_long_goldens_first_steps(fixture, steps, golden_dir):
    """The first steps of the reference's own 1000-step runs of 8 GEOM-Drugs-sized molecules (GEOM architecture) and of BASELINE.json configs[0]
    (64 QM9 molecules x 19 atoms), and of 6 molecules on the alpha-conditional model -- tests/golden/make_long_golden.py geom8 / config0 / cond6 -- on the
    same tape (and the stored per-molecule context): the oracle lands on z32_999 (the whole
    runs are compared on the GPU, tests/test_gpu_parity.py::test_long_horizon_sampling_matches_reference_golden)."""
    path = os.path.join(golden_dir, fixture)
    if not os.path.exists(path):
        pytest.skip(f"{fixture} is not generated yet")
    g = np.load(path)
    case = str(g["dataset"])
    d = synth.DATASET_DIMS[case]
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]),
                           scale_2d=float(g["weight_scale"]))
    cfg = cfg_for(case, d["L"])
    nn_ = torch.tensor(g["num_nodes"])
    B = len(nn_)
    bi = O.num_nodes_to_batch_index(nn_)
    mask = torch.ones_like(bi).bool()
    gam = O.gamma_table(cfg)
    noise = O.TapeNoise(int(g["noise_seed"]))
    ctx = torch.tensor(g["context"])[bi] if "context" in g.files else None
    z = O.sample_combined_noise(noise, bi, B, mask, cfg.num_node_scalar_features, torch.float32)
    for s in range(999, 999 - steps, -1):
        z, _ = O.sample_p_zs_given_zt(W, cfg, gam, s / 1000, (s + 1) / 1000, z, bi, B, mask, ctx, noise)
        if s == 999:
            r32, r64 = torch.tensor(g["z32_999"]).double(), torch.tensor(g["z64_999"])
            bound = 4.0 * (r32 - r64).abs().max().item() + 1e-4 * r64.abs().max().item()
            assert z.shape == r32.shape and (z.double() - r32).abs().max().item() <= bound


@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_oracle_masked_nodes_match_reference_golden(case, golden_dir):
    """Masked nodes (`batch.mask` with False entries): the oracle against the reference's own full-width outputs (dyn_masked_*.npz)."""
    g = load(golden_dir, f"dyn_masked_{case}")
    d = synth.DATASET_DIMS[case]
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]))
    cfg = cfg_for(case, d["L"])
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    mask = torch.as_tensor(g["mask"]).bool()
    assert (~mask).sum() > 5 and mask.sum() > 5
    out = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi, mask, g.get("ctx"))
    assert (out - g["out32"]).abs().max().item() <= 2e-5
    assert (out - g["out64"]).abs().max().item() <= 1e-4
    assert out[~mask][:, :3].abs().max().item() == 0.0


NLL_TERMS = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior")


@pytest.mark.parametrize("case", CASES)
def test_nll_terms_match_reference_golden(golden_dir, case):
    """Evaluation-mode likelihood terms (variational_diffusion.py:948-1160; two evaluations of the network per batch) vs the terms the
    REFERENCE's own forward returned on the same batch, timesteps and noise tape (tests/golden/make_nll_golden.py), in fp32 and fp64, and the
    NLL assembled from them (qm9_mol_gen_ddpm.py:246-262)."""
    g = load(golden_dir, f"nll_full_{case}")
    d = synth.DATASET_DIMS[case]
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]))
    cfg = cfg_for(case, d["L"])
    for dtype, tag, tol in ((torch.float32, "32", 2e-5), (torch.float64, "64", 1e-9)):
        Pd = {k: v.to(dtype) for k, v in P.items()}
        out = O.nll_terms(Pd, cfg, g["x"], g["one_hot"], g["charges"] if d["include_charges"] else None, g["num_nodes"], g["t_int"],
                          O.TapeNoise(int(g["noise_seed"])), context=g.get("ctx"), log_pN=g[f"log_pN_{tag}"], dtype=dtype)
        for name in NLL_TERMS + ("eps_hat_x", "eps_hat_h"):
            want = g[f"{name}_{tag}"].to(torch.float64)
            err = (out[name].to(torch.float64) - want).abs().max().item()
            assert err <= tol * max(1.0, want.abs().max().item()), (name, tag, err)
        ref_terms = {k: g[f"{k}_{tag}"].to(torch.float64) for k in NLL_TERMS + ("log_pN",)}
        nll = O.nll_from_terms({k: v.to(torch.float64) for k, v in out.items()}, cfg.num_timesteps)
        want = O.nll_from_terms(ref_terms, cfg.num_timesteps)
        assert (nll - want).abs().max().item() <= 10 * tol * max(1.0, want.abs().max().item())
    assert torch.isfinite(want).all() and want.abs().max().item() > 10.0          # the fixture is not degenerate


def _variant_oracle_cfg(name):
    """OracleConfig of synth.VARIANTS[name] at the reduced width of the variant fixtures."""
    import dataclasses
    kw = dict(num_layers=synth.SHRINK["num_encoder_layers"])
    v = synth.VARIANTS[name]
    fields = {f.name for f in dataclasses.fields(O.OracleConfig)}
    for grp in ("module_cfg", "layer_cfg", "mp_cfg"):
        for k, val in v.get(grp, {}).items():
            if k in fields:
                kw[k] = tuple(val) if k == "nonlinearities" else val
    if "num_encoder_layers" in v.get("model_cfg", {}):
        kw["num_layers"] = v["model_cfg"]["num_encoder_layers"]
    return O.OracleConfig(**kw)


@pytest.mark.parametrize("name", list(synth.VARIANTS))
def test_general_forward_matches_reference_variant_golden(golden_dir, name):
    """The oracle's general forward (GCP v1, frame / sigma gates, residuals, ablated frame updates, GCPLayerNorm, other depths / widths /
    nonlinearities, vector-sum position updates) against the REFERENCE's own outputs, fp32 and fp64, all-True and partial masks."""
    g = np.load(os.path.join(golden_dir, f"dyn_variant_{name}.npz"))
    shapes = {k: tuple(int(x) for x in s.split(",")) if s else () for k, s in zip(g["keys"].tolist(), g["shapes"].tolist())}
    W = synth.make_weights(shapes, seed=int(g["weight_seed"]), scale_2d=float(g["weight_scale"]))
    cfg = _variant_oracle_cfg(name)
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    xh, t = torch.tensor(g["xh"]), torch.tensor(g["t"])
    for tag, mask in (("full", None), ("part", torch.tensor(g["mask_part"]))):
        out = O.dynamics_forward_general(W, cfg, xh, t, bi, mask)
        assert (out.double() - torch.tensor(g[f"out32_{tag}"]).double()).abs().max().item() <= 2e-6, (name, tag)
        W64 = {k: v.double() for k, v in W.items()}
        out64 = O.dynamics_forward_general(W64, cfg, xh.double(), t.double(), bi, mask)
        assert (out64 - torch.tensor(g[f"out64_{tag}"])).abs().max().item() <= 1e-12, (name, tag)


def test_general_forward_equals_production_forward(golden_dir):
    g = load(golden_dir, "dyn_small_qm9")
    P = weights_of(g)
    cfg = cfg_for("qm9", O.infer_num_layers(P))
    bi = O.num_nodes_to_batch_index(g["num_nodes"])
    a = O.dynamics_forward(P, cfg, g["xh"], g["t"], bi)
    b = O.dynamics_forward_general(P, cfg, g["xh"], g["t"], bi)
    assert torch.equal(a, b) or (a - b).abs().max().item() <= 1e-7
