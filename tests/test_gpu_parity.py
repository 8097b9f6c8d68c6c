"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's golden vectors.  MI355X only.

Tolerance (BASELINE.json north_star): 1e-4 on output coordinates / features.  Where a synthetic net drives
activations far above O(1) the same bar is applied relative to the output scale (the reference's own
fp32-vs-fp64 gap scales the same way).
"""
import ctypes as C
import importlib
import math
import os

import numpy as np
import pytest
import torch

import synth
from oracle import gcdm_oracle as O

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("bio-diffusion_amd")
TOL = 1e-4


def _dims(case):
    return synth.DATASET_DIMS[case]


def _ocfg(case):
    d = _dims(case)
    return O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"], num_context=d["n_ctx"],
                          num_layers=d["L"], norm_values=d["norm_values"])


MODES = [pytest.param(1, id="f16x3"), pytest.param(0, id="f32")]


def _net(case, seed=17, scale=1.0, mode=None):
    d = _dims(case)
    ds = "geom" if case == "geom" else "qm9"
    cond = ("alpha",) if d["n_ctx"] else ()
    cfgs = pkg.default_cfgs(ds, cond)
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=seed, scale_2d=scale)
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    if mode is not None:
        net._ensure_handle(torch.device("cuda"))
        net.set_mfma_mode(mode)
    return net, W, cfgs


def _fwd(net, xh, t, bi, ctx=None):
    dev = torch.device("cuda")
    batch = dict(batch=bi.to(dev), mask=torch.ones(len(bi), dtype=torch.bool, device=dev), props_context=None if ctx is None else ctx.to(dev))
    _, out = net(batch, xh.to(dev), t.to(dev))
    torch.cuda.synchronize()
    return out.cpu()


def test_native_library_is_loaded():
    net, _, _ = _net("geom")
    net._ensure_handle(torch.device("cuda"))
    maps = open("/proc/self/maps").read()
    assert "libgcdm_hip.so" in maps


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_forward_matches_reference_golden(case, mode, golden_dir):
    """Full-width production architecture vs the outputs of the REFERENCE itself (tests/golden/dyn_full_*.npz), in both
    matrix modes (split-precision f16x3 = default, fp32 MFMA)."""
    g = np.load(os.path.join(golden_dir, f"dyn_full_{case}.npz"))
    net, W, _ = _net(case, seed=int(g["weight_seed"]), mode=mode)
    assert net.mfma_mode == mode
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    ctx = torch.tensor(g["ctx"]) if "ctx" in g.files else None
    out = _fwd(net, torch.tensor(g["xh"]), torch.tensor(g["t"]), bi, ctx)
    assert (out - torch.tensor(g["out32"])).abs().max().item() <= TOL
    assert (out - torch.tensor(g["out64"])).abs().max().item() <= TOL
    # internal state after the last layer (layouts: DESIGN.md section 2)
    N = len(bi)
    h = net.debug_read("h").view(64, N, 4).permute(1, 0, 2).reshape(N, 256)
    chi = net.debug_read("chi").view(32, 3, N).permute(2, 0, 1)
    assert (h - torch.tensor(g["h_last"])).abs().max().item() <= TOL
    assert (chi - torch.tensor(g["chi_last"])).abs().max().item() <= TOL
    assert net.read_flags() == 0


def _un_g4(buf, groups, n):
    """[groups][n] float4 channel groups -> [n, 4 * groups]"""
    return buf.view(groups, n, 4).permute(1, 0, 2).reshape(n, 4 * groups)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_stage_outputs_match_reference_golden(case, mode, golden_dir):
    """Stage-local pins against the REFERENCE's own intermediates (tests/golden/dyn_full_*.npz, captured by forward hooks in
    make_golden.py): embeddings (gcpnet.py:551-603, 1081-1190; a3/a4/a5/a10 of SURVEY 8), the first layer's aggregated messages
    (message passing, :676-737; a6/a7/a8) and the first layer's node update (:834-930; a9)."""
    g = np.load(os.path.join(golden_dir, f"dyn_full_{case}.npz"))
    d = _dims(case)
    net, W, _ = _net(case, seed=int(g["weight_seed"]), mode=mode)
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    N, E = len(bi), int((nn_ * nn_).sum())
    ctx = torch.tensor(g["ctx"]) if "ctx" in g.files else None
    dev = torch.device("cuda")
    net.sync_weights()
    net.plan(nn_)
    args = (torch.tensor(g["xh"]).to(dev), torch.tensor(g["t"]).to(dev), None if ctx is None else ctx.to(dev))

    def close(name, got, want):
        want = torch.tensor(want)
        err = (got - want).abs().max().item()
        assert err <= TOL * max(1.0, want.abs().max().item()), f"{name}: {err:.3e}"

    try:
        net.debug_set_layer_limit(0)                 # embedding stage only
        net.native_forward(*args)
        torch.cuda.synchronize()
        close("h_embed", _un_g4(net.debug_read("h"), 64, N), g["h_embed"])
        close("chi_embed", net.debug_read("chi").view(32, 3, N).permute(2, 0, 1), g["chi_embed"])
        close("e_embed", _un_g4(net.debug_read("ep"), d["Se"] // 4, E), g["e_embed"])
        u = net.debug_read("u").view(3, E).t()
        al = net.debug_read("alpha").view(d["Ve"], E).t()
        close("xi_embed", al[:, :, None] * u[:, None, :], g["xi_embed"])          # embedded edge vectors are rank 1: xi'_c = alpha_c * u
        net.debug_set_layer_limit(1)                 # + first interaction layer
        net.native_forward(*args)
        torch.cuda.synchronize()
        agg = net.debug_read("agg").view(N, 352)
        close("agg_s0", agg[:, :256], g["agg_s0"])
        close("agg_v0", agg[:, 256:].reshape(N, 32, 3), g["agg_v0"])
        close("h_l0", _un_g4(net.debug_read("h"), 64, N), g["h_l0"])
        close("chi_l0", net.debug_read("chi").view(32, 3, N).permute(2, 0, 1), g["chi_l0"])
    finally:
        net.debug_set_layer_limit(-1)
    assert net.read_flags() == 0


def test_edge_list_and_geometry_match_reference_golden(golden_dir):
    """The plan's edge list is bit-identical to the reference's get_fully_connected_edge_index (gcpnet.py:1054-1066) and the per-edge /
    per-node geometry (frames: localize, components/__init__.py:122-171; orientations: protein_graph_dataset.py:217-225; centralize)
    matches the reference's own outputs (tests/golden/fn_geometry.npz)."""
    g = np.load(os.path.join(golden_dir, "fn_geometry.npz"))
    d = _dims("qm9")
    net, W, _ = _net("qm9", seed=3)
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    N, E = len(bi), g["edge_index"].shape[1]
    dev = torch.device("cuda")
    net._ensure_handle(dev)
    net.sync_weights()
    net.plan(nn_)
    ei = torch.stack((net.debug_read("erow").view(torch.int32), net.debug_read("ecol").view(torch.int32))).long()
    assert torch.equal(ei, torch.tensor(g["edge_index"]))
    xh = torch.zeros(N, 3 + synth.dims_feat(d))
    xh[:, :3] = torch.tensor(g["x"])
    xh[:, 3] = 1.0
    try:
        net.debug_set_layer_limit(0)
        net.native_forward(xh.to(dev), torch.full((N, 1), 0.5, device=dev))
        torch.cuda.synchronize()
        assert (net.debug_read("x").view(3, N).t() - torch.tensor(g["x_central"])).abs().max().item() <= 1e-6
        assert (net.debug_read("frames").view(9, E).t().reshape(E, 3, 3) - torch.tensor(g["frames"])).abs().max().item() <= 1e-6
        assert (net.debug_read("chi0").view(2, 3, N).permute(2, 0, 1) - torch.tensor(g["chi0"])).abs().max().item() <= 1e-6
    finally:
        net.debug_set_layer_limit(-1)


# (fixture, matrix mode) -> (atom-type, charge) near-tie differences observed with the kernels of this tree; everything not listed: (0, 0)
NEAR_TIES = {("long_ragged16_qm9.npz", 0): (0, 1)}
NEAR_TIE_SLACK = 2


@pytest.mark.parametrize("fixture", ["long_full_qm9.npz", "long_ragged16_qm9.npz", "long_geom8.npz", "long_config0_qm9.npz", "long_cond6_qm9.npz"])
@pytest.mark.parametrize("mode", MODES)
def test_long_horizon_sampling_matches_reference_golden(mode, fixture, golden_dir):
    """SURVEY section 7 contract (iii): the FULL 1000-step free-running sample on the noise tape of tests/golden/long_full_qm9.npz
    (the reference's own mol_gen_sample, variational_diffusion.py:1282-1412, at full width in fp32 and fp64, make_long_golden.py).
    At every stored checkpoint |hip - ref32| <= 4 |ref32 - ref64| + 1e-4 max|z|; same for the decoded positions; decoded discrete
    outputs equal the reference's wherever its fp32 and fp64 runs agree.  Both matrix modes (f16x3 must hold this WITHOUT the fp32 re-run).
    Four fixtures: 4 molecules (n = 5, 19, 3, 11); a ragged batch of 16 molecules of 5 ... 27 atoms (275 atoms, rows cut by tile boundaries); round 4:
    8 GEOM-Drugs-sized molecules of 18 ... 72 atoms on the GEOM architecture (342 atoms, 16 802 edges), and BASELINE.json configs[0] itself -- 64 QM9
    molecules x 19 atoms (the reference's own full run of that shape, hours of CPU in the build container; no CPU oracle time on the GPU box); and
    6 molecules on the alpha-CONDITIONAL model (configs[2]'s architecture) with the stored per-molecule context."""
    path = os.path.join(golden_dir, fixture)
    if not os.path.exists(path):
        pytest.skip(f"{fixture} is not generated yet (tests/golden/make_long_golden.py)")
    g = np.load(path)
    case = str(g["dataset"]) if "dataset" in g.files else "qm9"
    net, W, cfgs = _net(case, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]), mode=mode)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("geom" if case == "geom" else "qm9")).cuda()
    nn_ = torch.tensor(g["num_nodes"])
    N, F, T = int(nn_.sum()), _ocfg(case).num_node_scalar_features, int(g["T"])
    ctx = torch.tensor(g["context"]).cuda() if "context" in g.files else None       # [B, 1] per molecule; mol_gen_sample expands it (:1317-1320)
    tape = O.TapeNoise(int(g["noise_seed"]))
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1).cuda() for _ in range(T + 2)]
    want = {int(s) for s in g["checkpoints"]}
    got = {}

    def cb(s, z):
        if s in want:
            got[s] = z.detach().cpu().clone()

    out, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", context=ctx, noise_fn=lambda k: draws[k], step_callback=cb)
    out = out.cpu()
    assert net.mfma_mode == mode and (ddpm.last_flags & pkg._native.FLAG_F16_RANGE) == 0        # no fp32 re-run behind the scenes
    assert set(got) == want
    worst = 0.0
    for s in sorted(want, reverse=True):
        r32, r64 = torch.tensor(g[f"z32_{s}"]).double(), torch.tensor(g[f"z64_{s}"])
        bound = 4.0 * (r32 - r64).abs().max().item() + 1e-4 * r64.abs().max().item()
        err = (got[s].double() - r32).abs().max().item()
        worst = max(worst, err / bound)
        assert err <= bound, f"s={s}: |hip - ref32| = {err:.3e} > {bound:.3e}"
    f32, f64 = torch.tensor(g["final32"]).double(), torch.tensor(g["final64"])
    bound = 4.0 * (f32[:, :3] - f64[:, :3]).abs().max().item() + 1e-4 * f64[:, :3].abs().max().item()
    assert (out[:, :3].double() - f32[:, :3]).abs().max().item() <= bound
    # discrete outputs: equal to the reference's on every atom its own fp32 and fp64 runs decide alike -- except rounding near-ties: the
    # untrained weights drive the charge channel to O(5e3), where the allowed 1e-4 * max|z| deviation of the latent is a fraction of the
    # rounding unit; such atoms (charge off by at most 1) are counted, not hidden, and their number is pinned (NEAR_TIES)
    nt = _ocfg(case).num_atom_types
    dec_t = f32[:, 3:3 + nt].argmax(1) == f64[:, 3:3 + nt].argmax(1)
    bad_t = int((out[:, 3:3 + nt].argmax(1)[dec_t] != f32[:, 3:3 + nt].argmax(1)[dec_t]).sum())
    if _ocfg(case).include_charges:
        dec_q = f32[:, 3 + nt] == f64[:, 3 + nt]
        dq = (out[:, 3 + nt].double() - f32[:, 3 + nt])[dec_q].abs()
    else:                                              # GEOM: no charge column
        dec_q, dq = torch.zeros(0, dtype=torch.bool), torch.zeros(0, dtype=torch.float64)
    bad_q = int((dq != 0).sum())
    # pinned to what the kernels of this tree produce (round 5, GPUTEST log: every fixture 0 / 0 in both modes, except ONE charge near-tie of the ragged
    # fixture in the fp32-MFMA mode): a regression from 0 to a handful of differing atoms fails here instead of hiding below a 1 % allowance
    # ... with a slack of NEAR_TIE_SLACK atoms: a benign re-ordering of a contraction flips a rounding near-tie somewhere else without being a
    # regression (round 5's msg0 K-layout did); such atoms stay off by at most one charge unit, and going beyond the pin is printed
    allowed_t, allowed_q = NEAR_TIES.get((fixture, mode), (0, 0))
    if bad_t > allowed_t or bad_q > allowed_q:
        print(f"NOTE {fixture} mode {mode}: {bad_t} / {bad_q} near-tie differences exceed the pin {allowed_t} / {allowed_q} (inside the slack of {NEAR_TIE_SLACK})")
    assert bad_t <= allowed_t + NEAR_TIE_SLACK and bad_q <= allowed_q + NEAR_TIE_SLACK and (dq.max().item() if len(dq) else 0.0) <= 1.0, \
        (bad_t, bad_q, dq.max().item() if len(dq) else 0.0)
    print(f"long horizon {fixture}: {bad_t} type / {bad_q} charge near-tie differences on {int(dec_t.sum())} / {int(dec_q.sum())} decided atoms")
    print(f"long horizon ({'f16x3' if mode else 'f32'}): worst err / bound over the checkpoints = {worst:.3f}")


# (test_free_running_sampling_config0_size -- 24 free-running steps of the configs[0] shape against the CPU oracle, with a relaxed discrete check -- is folded
#  into test_long_horizon_sampling_matches_reference_golden[long_config0_qm9.npz]: the reference's own full 1000-step run of that shape, no oracle time on
#  the GPU box, discrete outputs equal on every atom the reference's fp32 and fp64 runs decide alike.)


@pytest.mark.parametrize("case,num_nodes", [
    ("qm9", [1]), ("qm9", [2, 1, 1, 3]), ("qm9", [29] * 7 + [3]), ("qm9", [64, 65, 63, 1, 130]),
    ("geom", [44] * 5), ("geom", [181, 3, 90]), ("qm9cond", [19] * 9),
])
@pytest.mark.parametrize("mode", MODES)
def test_forward_matches_oracle_ragged(case, num_nodes, mode):
    """Edge cases of the tiling: single atoms, rows shorter / equal / longer than the 64-edge tile, rows spanning 3+ tiles
    (atomic accumulation), last tile partially filled, molecule sizes at the dataset maxima (29 / 181)."""
    d = _dims(case)
    net, W, _ = _net(case, seed=23, scale=0.5, mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=31, t_value=0.63, n_ctx=d["n_ctx"])
    ref = O.dynamics_forward(W, _ocfg(case), xh, t, bi, None, ctx)
    out = _fwd(net, xh, t, bi, ctx)
    scale = max(1.0, ref.abs().max().item())
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max().item() <= TOL * scale


def test_edge_embedding_of_both_modes_agrees_on_a_large_ragged_batch():
    """The split-precision edge-embedding kernel feeds MFMAs straight from inline-asm 16-bit partial writes; without the settle fence
    (x3_settle, gcdm_edge_x3.hip.h) a few waves of a grid larger than one round of workgroups read stale B lanes, different waves from
    run to run.  40 870 edges (320 workgroups), 6 repeats, every edge within 1e-5 of the fp32 kernel's e' and alpha."""
    d = _dims("geom")
    num_nodes = [181, 3, 90]
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=31, t_value=0.63, n_ctx=d["n_ctx"])
    want = None
    for mode, reps in ((0, 1), (1, 6)):
        net, _, _ = _net("geom", seed=23, scale=0.5, mode=mode)
        for _ in range(reps):
            _fwd(net, xh, t, bi, ctx)
            E = net.debug_read("u").numel() // 3
            got = (_un_g4(net.debug_read("ep"), d["Se"] // 4, E), net.debug_read("alpha").view(d["Ve"], E).t().clone())
            if want is None:
                want = got
                continue
            for g, w in zip(got, want):
                assert (g - w).abs().max().item() <= 1e-5 * max(1.0, w.abs().max().item())


@pytest.mark.parametrize("tile", [64, 32])
@pytest.mark.parametrize("case", ["qm9", "geom"])
def test_persistent_workgroups_walk_a_ragged_batch(case, tile):
    """More tiles than the chip holds workgroups: the split-precision edge-message kernel runs as persistent workgroups that walk the
    tile list with the next tile's operands prefetched (option "persistent", default on).  Same bits as one workgroup per tile
    (persistent = 0), and the reference's numbers within the forward tolerance."""
    d = _dims(case)
    g = torch.Generator().manual_seed(77)
    if case == "qm9":
        num_nodes = [int(v) for v in torch.randint(3, 30, (150,), generator=g)]
    else:
        num_nodes = [int(v) for v in torch.randint(3, 120, (14,), generator=g)] + [181, 3]
    E = sum(n * n for n in num_nodes)
    assert E // tile > 2 * 256 * (64 // tile) + 64, "the batch must give every workgroup more than one tile"
    net, W, _ = _net(case, seed=29, scale=0.5, mode=1)
    lib, h = net._lib, net._handle
    assert lib.gcdm_set_option(h, b"edge_tile", tile) == 0
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=37, t_value=0.41, n_ctx=d["n_ctx"])
    assert lib.gcdm_get_option(h, b"persistent") == 1
    out = _fwd(net, xh, t, bi, ctx)
    again = _fwd(net, xh, t, bi, ctx)
    assert lib.gcdm_set_option(h, b"persistent", 0) == 0
    one_per_tile = _fwd(net, xh, t, bi, ctx)
    assert lib.gcdm_set_option(h, b"persistent", 1) == 0 and lib.gcdm_set_option(h, b"edge_tile", 0) == 0
    assert torch.equal(out, again) and torch.equal(out, one_per_tile)
    if tile == 64:                   # (the 32-edge tiling against the oracle: test_forward_32_edge_tiles)
        ref = O.dynamics_forward(W, _ocfg(case), xh, t, bi, None, ctx)
        assert (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case,num_nodes", [("qm9", [29] * 7 + [3]), ("geom", [181, 3, 90])])
def test_forward_32_edge_tiles(case, num_nodes, mode):
    """The 32-edge tiling of the edge-message kernels (option "edge_tile"; two workgroups per CU) gives the same result."""
    d = _dims(case)
    net, W, _ = _net(case, seed=23, scale=0.5, mode=mode)
    lib, h = net._lib, net._handle
    assert lib.gcdm_set_option(h, b"edge_tile", 32) == 0 and lib.gcdm_get_option(h, b"edge_tile") == 32
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=31, t_value=0.63, n_ctx=d["n_ctx"])
    ref = O.dynamics_forward(W, _ocfg(case), xh, t, bi, None, ctx)
    out = _fwd(net, xh, t, bi, ctx)
    assert (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert lib.gcdm_set_option(h, b"edge_tile", 64) == 0
    out64 = _fwd(net, xh, t, bi, ctx)
    assert (out - out64).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("tile", [32, 64])
def test_ragged_geom_is_bit_reproducible(mode, tile):
    """Rows of 100+ edges are cut into >= 3 pieces by the 32- / 64-edge tiles; the pieces are summed from per-tile partials in tile order
    (no atomics), so repeated launches are bit-identical -- and the result does not depend on the tile size beyond the summation order."""
    d = _dims("geom")
    net, W, _ = _net("geom", seed=23, scale=0.5, mode=mode)
    lib, h = net._lib, net._handle
    sizes = [181, 3, 90, 44, 130, 65, 64, 63, 1, 101] * 3
    xh, t, bi, nn_, _ = synth.make_inputs(sizes, synth.dims_feat(d), seed=31, t_value=0.63)
    assert lib.gcdm_set_option(h, b"edge_tile", tile) == 0
    outs = [_fwd(net, xh, t, bi) for _ in range(4)]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = O.dynamics_forward(W, _ocfg("geom"), xh[:184], t[:184], bi[:184])          # the first two molecules (181 + 3 atoms) alone: same flat neighbours
    sub = _fwd(net, xh[:184], t[:184], bi[:184])
    assert (sub - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert lib.gcdm_set_option(h, b"edge_tile", 0) == 0


@pytest.mark.parametrize("tile", [32, 64])
def test_repeat_launch_bit_identity_full_batch(tile):
    """Run-to-run bit reproducibility of the split-precision kernels with co-resident workgroups (32-edge tiles: two per CU) over many
    launches of the benchmark batch: a regression of the packed-fp32 code-generation hazard (the library is built with
    -target-feature -packed-fp32-ops) or any hidden LDS / VMEM race shows up as a differing hash."""
    d = _dims("qm9")
    net, W, _ = _net("qm9", seed=51, scale=0.5, mode=1)
    lib, h = net._lib, net._handle
    assert lib.gcdm_set_option(h, b"edge_tile", tile) == 0
    xh, t, bi, nn_, _ = synth.make_inputs([19] * 1024, synth.dims_feat(d), seed=77, t_value=0.41)
    first = _fwd(net, xh, t, bi)
    for _ in range(12):
        assert torch.equal(_fwd(net, xh, t, bi), first)
    # the 64-node tiles of the layer node kernel (round 4; this batch picks 32 on its own): same bits, launch after launch
    assert lib.gcdm_set_option(h, b"node_tile", 64) == 0
    for _ in range(8):
        assert torch.equal(_fwd(net, xh, t, bi), first)
    assert lib.gcdm_set_option(h, b"node_tile", 0) == 0
    assert lib.gcdm_set_option(h, b"edge_tile", 0) == 0


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_masked_nodes_match_reference_golden(case, mode, golden_dir):
    """`batch.mask` with False entries (gcpnet.py:1062-1065, 1081-1099, 914-928; components/__init__.py:53-92): full-width forward against the
    REFERENCE's own outputs (tests/golden/dyn_masked_*.npz: random masks, a fully unmasked molecule, masked tails), both matrix modes and both
    tile sizes; masked rows of the input are arbitrary (the network zeroes them); masked plans are refused by the sampler entry points."""
    g = np.load(os.path.join(golden_dir, f"dyn_masked_{case}.npz"))
    net, W, _ = _net(case, seed=int(g["weight_seed"]), mode=mode)
    lib, h = net._lib, net._handle
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    mask = torch.tensor(g["mask"]).bool()
    ctx = torch.tensor(g["ctx"]) if "ctx" in g.files else None
    xh = torch.tensor(g["xh"]).clone()
    xh[~mask] = 7.5                                     # garbage in the masked rows must not matter
    dev = torch.device("cuda")
    for tile in (32, 64):
        assert lib.gcdm_set_option(h, b"edge_tile", tile) == 0
        batch = dict(batch=bi.to(dev), mask=mask.to(dev), props_context=None if ctx is None else ctx.to(dev))
        _, out = net(batch, xh.to(dev), torch.tensor(g["t"]).to(dev))
        torch.cuda.synchronize()
        out = out.cpu()
        assert (out - torch.tensor(g["out32"])).abs().max().item() <= TOL
        assert (out - torch.tensor(g["out64"])).abs().max().item() <= TOL
        assert out[~mask][:, :3].abs().max().item() == 0.0
    assert lib.gcdm_set_option(h, b"edge_tile", 0) == 0
    assert net.read_flags() == 0
    z = torch.zeros_like(xh).to(dev)
    st = lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, C.c_uint64(1), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st < 0 and b"all-True" in lib.gcdm_last_error(h)
    # back to an unmasked plan on the same handle
    out2 = _fwd(net, torch.tensor(g["xh"]), torch.tensor(g["t"]), bi, ctx)
    assert torch.isfinite(out2).all()


def test_f16_range_flag_and_fp32_fallback():
    """Activations beyond the f16 range: the split-precision kernel raises GCDM_FLAG_F16_RANGE and the module-level call
    transparently recomputes with fp32 MFMA (bit-identical to fp32 mode)."""
    d = _dims("qm9")
    net, W, _ = _net("qm9", seed=29, scale=1.0)
    xh, t, bi, nn_, _ = synth.make_inputs([19, 7, 30], synth.dims_feat(d), seed=5)
    xh[:, 3:] *= 1e7          # node features far outside the trained range: activations reach ~5e8 (finite in fp32, beyond the 1.2e8 of the f16 images)
    net._ensure_handle(torch.device("cuda"))
    net.set_mfma_mode(0)
    want = _fwd(net, xh, t, bi)
    scale = want.abs().max().item()
    net.set_mfma_mode(1)
    net.check_f16_range = False
    raw = _fwd(net, xh, t, bi)
    fl = net.read_flags()
    assert torch.isfinite(want).all() and scale > 1.0
    assert fl & pkg._native.FLAG_F16_RANGE, "test input does not leave the f16 range"
    net.check_f16_range = True
    got = _fwd(net, xh, t, bi)
    assert net.mfma_mode == 1
    assert torch.equal(got, want) or (got - want).abs().max().item() <= 1e-6 * scale
    del raw
    # deferred guard (the default of the module-level call): no host sync per call, the flag surfaces at the next look and the handle falls back
    net.check_f16_range = "deferred"
    _fwd(net, xh, t, bi)
    with pytest.raises(pkg.F16RangeError):
        net.check_deferred_flags()
    assert net.mfma_mode == 0
    net.set_mfma_mode(1)
    net.read_flags()


@pytest.mark.parametrize("big,shift", [(300.0, 4), (40.0, 1), (1500.0, 6)])
def test_large_weights_keep_the_split_precision_mode(big, shift):
    """A checkpoint with a matrix weight >= 31.9 used to demote the whole handle to fp32 MFMA (2.7x slower).  The exponent split of the f16
    images is now chosen per checkpoint (packed weights 2^(11-k) W, activation images 2^(k-11) x; k = 0 for ordinary models): the
    split-precision mode is retained, the forward matches the oracle, and ordinary weights return to k = 0."""
    d = _dims("qm9")
    cfgs = pkg.default_cfgs("qm9", ())
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=23, scale_2d=0.5)
    key = "interaction_layers.0.interaction.message_fusion.1.scalar_out.weight"
    W[key][3, 5] = big
    W["interaction_layers.2.feedforward_network.0.vector_up.weight"][1, 2] = -0.5 * big          # a vector-path matrix too
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    net._ensure_handle(torch.device("cuda"))
    net.sync_weights()
    assert net.mfma_mode == 1 and net._lib.gcdm_get_option(net._handle, b"x3_shift") == shift
    xh, t, bi, nn_, _ = synth.make_inputs([19, 7, 30], synth.dims_feat(d), seed=5)
    ref = O.dynamics_forward(W, _ocfg("qm9"), xh, t, bi, None, None)
    out = _fwd(net, xh, t, bi)
    assert (net.read_flags() & pkg._native.FLAG_F16_RANGE) == 0
    assert torch.isfinite(out).all() and (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    net.set_mfma_mode(0)
    out32 = _fwd(net, xh, t, bi)
    net.set_mfma_mode(1)
    assert (out - out32).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # back to ordinary weights: k = 0 again
    W[key][3, 5] = 0.01
    W["interaction_layers.2.feedforward_network.0.vector_up.weight"][1, 2] = 0.01
    net.load_state_dict(W)
    net.sync_weights(force=True)
    assert net.mfma_mode == 1 and net._lib.gcdm_get_option(net._handle, b"x3_shift") == 0


@pytest.mark.parametrize("case,num_nodes", [("qm9", [19] * 70 + [3, 29, 1]), ("geom", [44, 181, 3, 90, 17, 64, 65, 63]), ("qm9cond", [7, 19, 4, 12, 23])])
def test_node_tile_sizes_agree_bitwise(case, num_nodes):
    """The 64-node tiles of the split-precision layer node kernel (k_node_x3w, round 4) run every contraction in the k-block order of the 32-node kernel:
    outputs of a full forward are bit-identical for both tile sizes -- node counts that are no multiple of 64, rows cut by edge-tile boundaries, a
    partial node mask, the context-conditional width -- and the automatic choice is one of the two."""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=33, scale=0.5, mode=1)
    lib, h = net._lib, net._handle
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=8, n_ctx=d["n_ctx"])
    dev = torch.device("cuda")
    mask = torch.ones(len(bi), dtype=torch.bool)
    outs = {}
    for masked in (False, True):
        if masked:
            mask[1::5] = False
            mask[0] = True
        batch = dict(batch=bi.to(dev), mask=mask.to(dev), props_context=None if ctx is None else ctx.to(dev))
        for nt in (32, 64, 0):
            assert lib.gcdm_set_option(h, b"node_tile", nt) == 0 and lib.gcdm_get_option(h, b"node_tile") == nt
            _, out = net(batch, xh.to(dev), t.to(dev))
            torch.cuda.synchronize()
            outs[(masked, nt)] = out.cpu()
        assert torch.isfinite(outs[(masked, 32)]).all()
        assert torch.equal(outs[(masked, 32)], outs[(masked, 64)]) and torch.equal(outs[(masked, 32)], outs[(masked, 0)])
    assert lib.gcdm_set_option(h, b"node_tile", 48) != 0
    lib.gcdm_set_option(h, b"node_tile", 0)


@pytest.mark.parametrize("case,num_nodes", [("qm9", [19] * 9 + [3, 29, 1]), ("geom", [44, 181, 3]), ("qm9cond", [7, 19, 4, 12, 23])])
@pytest.mark.parametrize("mode", MODES)
def test_captured_step_equals_direct_launches(case, num_nodes, mode):
    """gcdm_sample_step serves Philox-noise steps from ONE instantiated hipGraph per handle (round 4: the ~25 launches of a step differ between
    steps by four scalars and the draw index, which the captured kernels read from a device table at a device cursor).  Same kernels, same
    arguments: the latent after 40 steps -- with a rewind in the middle, as the range-checkpoint logic of the sampler makes -- is bit-identical to
    direct launches; a new z buffer, seed or option re-captures; caller-supplied noise, and handles that profile, launch directly."""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=23, scale=0.25, mode=mode)
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.tensor(num_nodes)
    bi = O.num_nodes_to_batch_index(nn_)
    N, D = len(bi), 3 + _ocfg(case).num_node_scalar_features
    dyn.plan(nn_)
    g = torch.Generator().manual_seed(5)
    z0 = (0.3 * torch.randn((N, D), generator=g)).to(dev)
    ctx = torch.randn((len(nn_), 1), generator=g)[bi].to(dev).contiguous() if d["n_ctx"] else None
    cp = None if ctx is None else C.c_void_p(ctx.data_ptr())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    seq = list(range(999, 979, -1)) + list(range(989, 969, -1))

    def run(graph, seed=7):
        assert lib.gcdm_set_option(h, b"step_graph", graph) == 0 and lib.gcdm_get_option(h, b"step_graph") == graph
        z = z0.clone()
        before = lib.gcdm_get_option(h, b"graph_launches")
        for s in seq:
            assert lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), cp, s, 1000, None, C.c_uint64(seed), C.c_void_p(fl.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        return z, lib.gcdm_get_option(h, b"graph_launches") - before

    direct, n0 = run(0)
    graphed, n1 = run(1)
    assert n0 == 0 and n1 == len(seq) and lib.gcdm_get_option(h, b"step_graph") == 1, lib.gcdm_last_error(h)
    assert torch.isfinite(direct).all() and torch.equal(direct, graphed)
    again, n2 = run(1)                               # another z buffer: re-captured, same result
    assert n2 == len(seq) and torch.equal(direct, again)
    other, _ = run(1, seed=8)                        # the seed is baked into the captured k_sample: a new seed must not reuse it
    assert not torch.equal(other, direct)
    other0, _ = run(0, seed=8)
    assert torch.equal(other, other0)
    # caller-supplied noise is not captured (the tape pointer changes every step)
    lib.gcdm_set_option(h, b"step_graph", 1)
    z = z0.clone()
    raw = torch.randn((N, D), generator=g).to(dev)
    before = lib.gcdm_get_option(h, b"graph_launches")
    assert lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), cp, 500, 1000, C.c_void_p(raw.data_ptr()), C.c_uint64(0), None, stream) == 0
    torch.cuda.synchronize()
    assert lib.gcdm_get_option(h, b"graph_launches") == before and torch.isfinite(z).all()


def test_step_refused_during_capture_keeps_the_callers_error_and_the_graph():
    """A call that fails the step's own argument checks WHILE the step is being captured (a context-conditioned model called without a context) must return
    that diagnostic -- what the direct path returns for the same call -- and must not switch the step graph off for the handle: the next valid call is
    captured and served by the graph (rounds 3-4 reported "capture of one step: no error" and latched the graph off)."""
    case = "qm9cond"
    d = _dims(case)
    net, W, cfgs = _net(case, seed=23, scale=0.25, mode=1)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.tensor([5, 19, 3, 11])
    bi = O.num_nodes_to_batch_index(nn_)
    N, D = len(bi), 3 + _ocfg(case).num_node_scalar_features
    dyn.plan(nn_)
    g = torch.Generator().manual_seed(5)
    z = (0.3 * torch.randn((N, D), generator=g)).to(dev)
    ctx = torch.randn((len(nn_), 1), generator=g)[bi].to(dev).contiguous()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    args = lambda cp, s: (h, C.c_void_p(z.data_ptr()), cp, s, 1000, None, C.c_uint64(7), C.c_void_p(fl.data_ptr()), stream)
    assert lib.gcdm_set_option(h, b"step_graph", 1) == 0
    before = lib.gcdm_get_option(h, b"graph_launches")
    assert lib.gcdm_sample_step(*args(None, 999)) != 0                      # refused: no context
    assert b"context required" in lib.gcdm_last_error(h), lib.gcdm_last_error(h)
    assert lib.gcdm_get_option(h, b"step_graph") == 1                        # ... and the graph is still enabled
    for s in (999, 998, 997):
        assert lib.gcdm_sample_step(*args(C.c_void_p(ctx.data_ptr()), s)) == 0, lib.gcdm_last_error(h)
    torch.cuda.synchronize()
    assert lib.gcdm_get_option(h, b"graph_launches") - before == 3 and torch.isfinite(z).all()
    # cog_fix never invalidates the captured step, nor does an option set to the value it already has (edge_tile below); mfma_mode ALWAYS drops the exec (the next
    # step re-captures) -- either way the steps keep being served by a graph: the launches keep counting
    assert lib.gcdm_set_option(h, b"cog_fix", 0) == 0 and lib.gcdm_set_option(h, b"cog_fix", 1) == 0
    assert lib.gcdm_set_option(h, b"edge_tile", lib.gcdm_get_option(h, b"edge_tile")) == 0
    assert lib.gcdm_set_option(h, b"mfma_mode", 1) == 0
    assert lib.gcdm_sample_step(*args(C.c_void_p(ctx.data_ptr()), 996)) == 0
    torch.cuda.synchronize()
    assert lib.gcdm_get_option(h, b"graph_launches") - before == 4


@pytest.mark.parametrize("bias", [1.0e4, 1.0e5])
def test_split_precision_envelope_at_large_activations(bias):
    """The worst clean point of the round-4 envelope sweep (tests/gpu_envelope.py): with no LayerNorm in the production configuration
    (use_gcp_norm false) a trained checkpoint's activations are bounded by nothing but its weights, so the last feed-forward bias of EVERY layer is
    set to +-1e4 / +-1e5 (h grows to ~1e6, the feed-forward hidden activations beyond): the exponent split stays k = 0 (a bias that never enters an
    f16 image must not cost activation range -- round 3 counted it), the range flag stays clear, and against the oracle in fp64 the split-precision
    forward is as accurate as plain fp32 (bar: 4 x the fp32 oracle's own distance from fp64 + 1e-6 max|out|)."""
    d = _dims("qm9")
    cfgs = pkg.default_cfgs("qm9", ())
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=29, scale_2d=0.25)
    for l in range(d["L"]):
        b = W[f"interaction_layers.{l}.feedforward_network.0.scalar_out.2.bias"]
        b.copy_(torch.where(torch.arange(b.numel()) % 2 == 0, bias, -bias).to(b.dtype))
    net.load_state_dict(W)
    net = net.cuda().eval()
    net._ensure_handle(torch.device("cuda"))
    net.sync_weights()
    assert net.mfma_mode == 1 and net._lib.gcdm_get_option(net._handle, b"x3_shift") == 0
    xh, t, bi, nn_, _ = synth.make_inputs([19, 7, 30, 12], synth.dims_feat(d), seed=6)
    ocfg = _ocfg("qm9")
    r32 = O.dynamics_forward(W, ocfg, xh, t, bi, None, None).double()
    r64 = O.dynamics_forward({k_: v.double() for k_, v in W.items()}, ocfg, xh.double(), t.double(), bi, None, None)
    out = _fwd(net, xh, t, bi).double()
    assert (net.read_flags() & pkg._native.FLAG_F16_RANGE) == 0 and torch.isfinite(out).all()
    scale = r64.abs().max().item()
    e_hip, e_ref = (out - r64).abs().max().item(), (r32 - r64).abs().max().item()
    print(f"bias {bias:g}: max|out| {scale:.3e}  |f16x3 - fp64| {e_hip:.2e}  |fp32 oracle - fp64| {e_ref:.2e}")
    assert e_hip <= 4.0 * e_ref + 1e-6 * scale
    net.set_mfma_mode(0)
    e32 = (_fwd(net, xh, t, bi).double() - r64).abs().max().item()
    net.set_mfma_mode(1)
    assert e32 <= 4.0 * e_ref + 1e-6 * scale


def test_weights_outside_every_exponent_split_use_fp32_mfma():
    """Beyond the largest shift (|W| >= ~1400 with the head room of the folded constants) or a non-finite weight: gcdm_finalize_weights switches
    the handle to fp32 MFMA, refuses mode 1, and the forward still matches the oracle."""
    d = _dims("qm9")
    cfgs = pkg.default_cfgs("qm9", ())
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=23, scale_2d=0.5)
    W["interaction_layers.0.interaction.message_fusion.1.scalar_out.weight"][3, 5] = 3.0e4
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    net._ensure_handle(torch.device("cuda"))
    net.sync_weights()
    assert net.mfma_mode == 0
    with pytest.raises(pkg._native.NativeError, match="outside the split-precision images"):
        net.set_mfma_mode(1)
    xh, t, bi, nn_, _ = synth.make_inputs([19, 7, 30], synth.dims_feat(d), seed=5)
    ref = O.dynamics_forward(W, _ocfg("qm9"), xh, t, bi, None, None)
    out = _fwd(net, xh, t, bi)
    assert torch.isfinite(out).all() and (out - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    W["interaction_layers.0.interaction.message_fusion.1.scalar_out.weight"][3, 5] = 0.01
    net.load_state_dict(W)
    net.sync_weights(force=True)
    assert net.mfma_mode == 1


def test_forward_input_validation():
    net, _, _ = _net("qm9")
    d = _dims("qm9")
    xh, t, bi, nn_, _ = synth.make_inputs([4, 5], synth.dims_feat(d))
    dev = torch.device("cuda")
    mask = torch.ones(len(bi), dtype=torch.bool, device=dev)
    with pytest.raises(ValueError):
        net(dict(batch=bi.to(dev), mask=mask), xh[:, :-1].to(dev), t.to(dev))
    mask2 = mask.clone(); mask2[:4] = False           # a molecule without an unmasked atom: the reference's centroid would be 0 / 0
    bi2 = bi.clone().to(dev)
    with pytest.raises(pkg._native.NativeError, match="unmasked"):
        net(dict(batch=bi2, mask=mask2), xh.to(dev), t.to(dev))
    # without diffusion_cfg.self_condition the reference ignores a self-conditioning input (gcpnet.py:1112); so does the mirror
    _, o1 = net(dict(batch=bi.to(dev), mask=mask), xh.to(dev), t.to(dev), xh_self_cond=xh.to(dev))
    _, o2 = net(dict(batch=bi.to(dev), mask=mask), xh.to(dev), t.to(dev))
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("mode", MODES)
def test_self_conditioning_forward_matches_reference_golden(mode, golden_dir):
    """diffusion_cfg.self_condition=True (gcpnet.py:1112-1139): full-width forward with a previous estimate and with none (zeros) vs the
    outputs of the REFERENCE itself (tests/golden/dyn_full_qm9sc.npz), both matrix modes; plus ragged tiles vs the oracle."""
    g = np.load(os.path.join(golden_dir, "dyn_full_qm9sc.npz"))
    d = _dims("qm9")
    F_ = synth.dims_feat(d)
    cfgs = pkg.default_cfgs("qm9")
    cfgs["diffusion_cfg"]["self_condition"] = True
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F_), seed=int(g["weight_seed"]))
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    net._ensure_handle(torch.device("cuda"))
    net.set_mfma_mode(mode)
    dev = torch.device("cuda")
    bi = O.num_nodes_to_batch_index(torch.tensor(g["num_nodes"]))
    batch = dict(batch=bi.to(dev), mask=torch.ones(len(bi), dtype=torch.bool, device=dev), props_context=None)
    xh, t, sc = torch.tensor(g["xh"]).to(dev), torch.tensor(g["t"]).to(dev), torch.tensor(g["sc"]).to(dev)
    _, out = net(batch, xh, t, xh_self_cond=sc, x_self_cond=sc)
    assert (out.cpu() - torch.tensor(g["out32"])).abs().max().item() <= TOL
    assert (out.cpu() - torch.tensor(g["out64"])).abs().max().item() <= TOL
    _, out0 = net(batch, xh, t)
    assert (out0.cpu() - torch.tensor(g["out32_nosc"])).abs().max().item() <= TOL
    # ragged tiling against the oracle
    ocfg = _ocfg("qm9")
    ocfg.self_condition = True
    Ws = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F_), seed=23, scale_2d=0.5)
    net.load_state_dict(Ws)
    xh2, t2, bi2, _, _ = synth.make_inputs([29] * 3 + [1, 70, 2], F_, seed=31, t_value=0.63)
    g2 = torch.Generator().manual_seed(8)
    sc2 = torch.randn(xh2.shape, generator=g2)
    ref = O.dynamics_forward(Ws, ocfg, xh2, t2, bi2, xh_self_cond=sc2)
    b2 = dict(batch=bi2.to(dev), mask=torch.ones(len(bi2), dtype=torch.bool, device=dev), props_context=None)
    _, o2 = net(b2, xh2.to(dev), t2.to(dev), xh_self_cond=sc2.to(dev))
    assert (o2.cpu() - ref).abs().max().item() <= TOL * max(1.0, ref.abs().max().item())
    assert net.read_flags() == 0
    # GEOM widths (33 embedding inputs, Se = 16, Ve = 8): same code, checked against the oracle
    dg = _dims("geom")
    Fg = synth.dims_feat(dg)
    cg = pkg.default_cfgs("geom")
    cg["diffusion_cfg"]["self_condition"] = True
    ng = pkg.GCPNetDynamics(**cg)
    Wg = synth.make_weights(synth.dynamics_shapes(dg["S"], dg["V"], dg["Se"], dg["Ve"], dg["L"], synth.dims_h_in(dg), self_cond_feats=Fg), seed=29, scale_2d=0.5)
    ng.load_state_dict(Wg)
    ng = ng.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    ng._ensure_handle(dev)
    ng.set_mfma_mode(mode)
    og = _ocfg("geom")
    og.self_condition = True
    xh3, t3, bi3, _, _ = synth.make_inputs([44, 5, 91, 17], Fg, seed=33, t_value=0.21)
    sc3 = torch.randn(xh3.shape, generator=g2)
    ref3 = O.dynamics_forward(Wg, og, xh3, t3, bi3, xh_self_cond=sc3)
    b3 = dict(batch=bi3.to(dev), mask=torch.ones(len(bi3), dtype=torch.bool, device=dev), props_context=None)
    _, o3 = ng(b3, xh3.to(dev), t3.to(dev), xh_self_cond=sc3.to(dev))
    assert (o3.cpu() - ref3).abs().max().item() <= TOL * max(1.0, ref3.abs().max().item())


def test_self_conditioned_sampling_matches_oracle():
    """mol_gen_sample with diffusion_cfg.self_condition=True (two network evaluations per step, the estimate fed back; oracle pinned by
    tests/golden/sampler_small_qm9sc.npz): free-running sample on the oracle's noise tape."""
    d = _dims("qm9")
    F_ = synth.dims_feat(d)
    cfgs = pkg.default_cfgs("qm9")
    cfgs["diffusion_cfg"]["self_condition"] = True
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F_), seed=43, scale_2d=0.25)
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    ocfg = _ocfg("qm9")
    ocfg.self_condition = True
    nn_ = torch.tensor([7, 19, 4, 12])
    N = int(nn_.sum())
    Tp = 6
    want, bi = O.mol_gen_sample(W, ocfg, nn_, O.TapeNoise(1234), num_timesteps=Tp)
    tape = O.TapeNoise(1234)
    draws = [torch.cat((tape(N, 3), tape(N, F_)), dim=-1) for _ in range(2 * Tp + 2)]
    out, bi2, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, noise_fn=lambda k: draws[k])
    out = out.cpu()
    assert torch.equal(bi2.cpu(), bi)
    scale = max(1.0, want[:, :3].abs().max().item())
    assert (out[:, :3] - want[:, :3]).abs().max().item() <= TOL * scale
    assert torch.equal(out[:, 3:], want[:, 3:])
    # Philox noise: runs, deterministic, finite
    a, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, norm_with_original_timesteps=True, seed=5)
    a = a.clone()
    b, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, norm_with_original_timesteps=True, seed=5)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    # the optimisation loop with self-conditioning (variational_diffusion.py:1500-1512) on the alpha-conditional model
    dc = _dims("qm9cond")
    Fc = synth.dims_feat(dc)
    cc = pkg.default_cfgs("qm9", ("alpha",))
    cc["diffusion_cfg"]["self_condition"] = True
    nc = pkg.GCPNetDynamics(**cc)
    Wc = synth.make_weights(synth.dynamics_shapes(dc["S"], dc["V"], dc["Se"], dc["Ve"], dc["L"], synth.dims_h_in(dc), self_cond_feats=Fc), seed=47, scale_2d=0.25)
    nc.load_state_dict(Wc)
    nc = nc.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    dd = pkg.EquivariantVariationalDiffusion(nc, cc["diffusion_cfg"], cc["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    oc = _ocfg("qm9cond")
    oc.self_condition = True
    gq = torch.Generator().manual_seed(9)
    ctx_b = torch.randn((len(nn_), 1), generator=gq)
    samples = []
    for n in nn_.tolist():
        x = torch.randn((n, 3), generator=gq) * 1.2
        samples.append((x - x.mean(0, keepdim=True), torch.nn.functional.one_hot(torch.randint(0, Fc, (n,), generator=gq), Fc).float()))
    wo, _ = O.mol_gen_optimize(Wc, oc, torch.cat([s_[0] for s_ in samples]), torch.cat([s_[1] for s_ in samples]), nn_, O.TapeNoise(77), context=ctx_b, num_timesteps=5)
    tp = O.TapeNoise(77)
    dr = [torch.cat((tp(N, 3), tp(N, Fc)), dim=-1) for _ in range(2 * 5 + 1)]
    oo, _, _ = dd.mol_gen_optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_nodes=nn_, device="cuda", num_timesteps=5, context=ctx_b.cuda(),
                                   noise_fn=lambda k: dr[k])
    oo = oo.cpu()
    assert (oo[:, :3] - wo[:, :3]).abs().max().item() <= TOL * max(1.0, wo[:, :3].abs().max().item()) and torch.equal(oo[:, 3:], wo[:, 3:])


def _inpaint_inputs(nn_, F_, charges, seed=21):
    N = int(nn_.sum())
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, 3), generator=g) * 1.5 + torch.tensor([0.3, -2.0, 1.0])            # not centred
    oh = torch.nn.functional.one_hot(torch.randint(0, F_, (N,), generator=g), F_).float()
    ch = torch.randint(0, 9, (N, 1), generator=g).float() if charges else None
    fixed = torch.rand(N, generator=g) < 0.4
    off = [0] + nn_.cumsum(0).tolist()
    fixed[off[1]:off[2]] = False                                                           # a molecule generated freely
    fixed[off[2]:off[3]] = True                                                            # a molecule kept whole
    fixed[0] = True
    return x, oh, ch, fixed


@pytest.mark.parametrize("selfcond", [False, True])
def test_inpaint_matches_oracle(selfcond):
    """RePaint inpainting (variational_diffusion.py:1582-1789 with its two crashing tokens repaired; oracle pinned by
    tests/golden/inpaint_small_qm9.npz): known part re-noised each step, model step, CoM matching, jumps back -- on the oracle's noise tape."""
    d = _dims("qm9")
    F_ = synth.dims_feat(d)
    cfgs = pkg.default_cfgs("qm9")
    cfgs["diffusion_cfg"]["self_condition"] = selfcond
    net = pkg.GCPNetDynamics(**cfgs)
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d), self_cond_feats=F_ if selfcond else 0),
                           seed=43, scale_2d=0.25)
    net.load_state_dict(W)
    net = net.cuda().eval()          # inference: the fused kernels ("auto" takes the module path while autograd records a training step)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    ocfg = _ocfg("qm9")
    ocfg.self_condition = selfcond
    nn_ = torch.tensor([7, 19, 4, 12])
    N = int(nn_.sum())
    x, oh, ch, fixed = _inpaint_inputs(nn_, d["num_atom_types"], True)
    mol = dict(x=x.cuda(), one_hot=oh.cuda(), charges=ch.cuda(), num_nodes=nn_)
    runs = [dict(num_resamplings=2, jump_length=2, num_timesteps=6, return_frames=1)]
    if not selfcond:
        runs.append(dict(num_resamplings=3, jump_length=1, num_timesteps=6, return_frames=3))
    for kw in runs:
        want = O.inpaint(W, ocfg, x, oh, ch, nn_, fixed, O.TapeNoise(1234), **kw)
        tape = O.TapeNoise(1234)
        cache = {}
        def noise_fn(k, _t=tape, _c=cache):
            assert k == len(_c)                                    # the draws are requested in order, each once
            _c[k] = torch.cat((_t(N, 3), _t(N, F_)), dim=-1)
            return _c[k]
        out = ddpm.inpaint(mol, fixed.cuda(), noise_fn=noise_fn, **kw).cpu()
        assert out.shape == want.shape
        last, lw = (out, want) if kw["return_frames"] == 1 else (out[0], want[0])
        scale = max(1.0, want.abs().max().item())
        assert (last[:, :3] - lw[:, :3]).abs().max().item() <= TOL * scale and torch.equal(last[:, 3:], lw[:, 3:])
        if kw["return_frames"] > 1:
            assert (out[1:] - want[1:]).abs().max().item() <= TOL * scale and want[1:].abs().max().item() > 0
    assert torch.equal(mol["x"].cpu(), x)                          # the caller's molecule is not modified
    # the general loop (module path: what configurations off the fused kernels and generate_x_only take) gives the oracle's numbers too
    net.path = "modules"
    try:
        for kw in runs:
            want = O.inpaint(W, ocfg, x, oh, ch, nn_, fixed, O.TapeNoise(1234), **kw)
            tape = O.TapeNoise(1234)
            out = ddpm.inpaint(mol, fixed.cuda(), noise_fn=lambda k, _t=tape: torch.cat((_t(N, 3), _t(N, F_)), dim=-1), **kw).cpu()
            scale = max(1.0, want.abs().max().item())
            last, lw = (out, want) if kw["return_frames"] == 1 else (out[0], want[0])
            assert out.shape == want.shape and (last[:, :3] - lw[:, :3]).abs().max().item() <= TOL * scale and torch.equal(last[:, 3:], lw[:, 3:])
            if kw["return_frames"] > 1:
                assert (out[1:] - want[1:]).abs().max().item() <= TOL * scale
    finally:
        net.path = "auto"
    # Philox noise: deterministic, finite, seed-dependent; the schedule method is the reference's
    a = ddpm.inpaint(mol, fixed.cuda(), num_resamplings=2, jump_length=2, num_timesteps=6, seed=5).clone()
    b = ddpm.inpaint(mol, fixed.cuda(), num_resamplings=2, jump_length=2, num_timesteps=6, seed=5)
    c = ddpm.inpaint(mol, fixed.cuda(), num_resamplings=2, jump_length=2, num_timesteps=6, seed=6)
    assert torch.equal(a, b) and torch.isfinite(a).all() and not torch.equal(a[:, :3], c[:, :3])
    assert ddpm.get_repaint_schedule(2, 2, 6) == [4, 4, 2]
    with pytest.raises(AssertionError):
        ddpm.inpaint(mol, fixed.cuda(), jump_length=2, num_timesteps=6, return_frames=2)
    if not selfcond:
        # the whole-model entry (generate_molecules(ddpm_mode="inpainting"), qm9_mol_gen_ddpm.py:1130-1181): zero molecule, first node fixed,
        # every molecule moved back to the given centre of mass (the origin)
        model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
        model.ddpm.dynamics_network.load_state_dict(W)
        model = model.cuda()
        mols = model.generate_molecules(ddpm_mode="inpainting", num_samples=3, num_nodes=torch.tensor([5, 9, 3]), num_timesteps=4, num_resamplings=2,
                                        jump_length=2)
        assert [len(m[0]) for m in mols] == [5, 9, 3]
        for pos, at, chg in mols:
            assert torch.isfinite(pos).all() and pos.mean(0).abs().max().item() <= 1e-3 * max(1.0, pos.abs().max().item())
            assert at.min() >= 0 and at.max() < d["num_atom_types"]
        with pytest.raises(NotImplementedError):
            model.generate_molecules(ddpm_mode="conditional", num_samples=1)


def test_cabi_error_paths():
    """Every misuse of the C ABI returns a negative status with a message (no exception, no crash, no silent fallback)."""
    native = pkg._native
    lib = native.load()
    H = C.c_void_p
    good = native.GcdmConfig(native.ABI_VERSION, 5, 1, 0, 1, 9, 256, 32, 64, 16, 4, 1000, 1.0, (C.c_float * 3)(1, 4, 10), (C.c_float * 3)(0, 0, 0), 0, 0)
    for field, bad in (("abi_version", 99), ("h_hidden_dim", 128), ("chi_hidden_dim", 16), ("e_hidden_dim", 48), ("bottleneck", 2),
                       ("condition_on_time", 0), ("num_layers", 0)):
        cfg = native.GcdmConfig.from_buffer_copy(good)
        setattr(cfg, field, bad)
        h = H()
        assert lib.gcdm_create(C.byref(cfg), C.byref(h)) < 0, field
    h = H()
    assert lib.gcdm_create(C.byref(good), C.byref(h)) == 0
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = torch.zeros((8, 9), device="cuda")
    t = torch.zeros(8, device="cuda")
    ptr = lambda a: C.c_void_p(a.data_ptr())
    # forward / sampling before weights and plan
    assert lib.gcdm_forward(h, ptr(x), ptr(t), None, ptr(x), None, stream) < 0 and lib.gcdm_last_error(h)
    assert lib.gcdm_sample_init(h, ptr(x), None, C.c_uint64(1), stream) < 0
    assert lib.gcdm_finalize_weights(h) < 0 and b"missing" in lib.gcdm_last_error(h).lower()
    w = torch.zeros(10)
    assert lib.gcdm_set_weight(h, b"no.such.key", ptr(w), 10) < 0 or lib.gcdm_finalize_weights(h) < 0
    assert lib.gcdm_plan_batch(h, 0, None) < 0
    nn_bad = torch.tensor([3, 0, 2], dtype=torch.int32)
    assert lib.gcdm_plan_batch(h, 3, C.c_void_p(nn_bad.data_ptr())) < 0
    assert lib.gcdm_set_option(h, b"nonsense", 1) < 0 and lib.gcdm_set_option(h, b"edge_tile", 48) < 0 and lib.gcdm_get_option(h, b"nonsense") < 0
    assert lib.gcdm_set_gamma(h, ptr(w), 10) < 0                                   # wrong table length
    lib.gcdm_destroy(h)
    # a working handle: bad step indices, missing gamma table, bad buffers
    net, W, cfgs = _net("qm9", seed=3, scale=0.25)
    dev = torch.device("cuda")
    net._ensure_handle(dev); net.sync_weights(); net.plan(torch.tensor([4, 5]))
    lib, h = net._lib, net._handle
    z = torch.zeros((9, 9), device=dev)
    assert lib.gcdm_sample_step(h, ptr(z), None, 5, 1000, None, C.c_uint64(1), None, stream) < 0 and b"gamma" in lib.gcdm_last_error(h)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dyn, lib, h = ddpm._native(dev)
    dyn.plan(torch.tensor([4, 5]))
    assert lib.gcdm_sample_step(h, ptr(z), None, 10, 10, None, C.c_uint64(1), None, stream) < 0
    assert lib.gcdm_sample_step(h, ptr(z), None, -1, 10, None, C.c_uint64(1), None, stream) < 0
    assert lib.gcdm_sample_final(h, None, None, None, C.c_uint64(1), ptr(z), None, stream) < 0
    assert lib.gcdm_encode_samples(h, None, ptr(z), None, stream) < 0 and lib.gcdm_unnormalize_z(h, ptr(z), None, stream) < 0
    assert lib.gcdm_debug_read(h, b"no_such_buffer", None, 0) < 0
    tb = native.GcdmBondTables()
    tb.num_types = 99
    assert lib.gcdm_check_stability(tb, ptr(z), 9, ptr(z), ptr(z), 1, ptr(z), stream) == -1
    # and the handle still works afterwards
    assert lib.gcdm_sample_step(h, ptr(z), None, 3, 10, None, C.c_uint64(1), None, stream) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(z).all()


def _raw_noise(seed, N, F):
    """One draw in the reference's randn order: x-part [N,3] then h-part [N,F] (variational_diffusion.py:804-817)."""
    tape = O.TapeNoise(seed)
    return torch.cat((tape(N, 3), tape(N, F)), dim=-1)


@pytest.mark.parametrize("case", ["qm9", "geom", "qm9cond"])
def test_teacher_forced_sampler_steps(case):
    """One ancestral step z_t -> z_s from the same z_t and the same noise: fused C-ABI step, the reference-signature
    Python method, and the oracle agree (contract (ii))."""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=41, scale=0.25)
    ocfg = _ocfg(case)
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.tensor([6, 19, 9, 3])
    B = len(nn_)
    bi = O.num_nodes_to_batch_index(nn_)
    N, F = len(bi), ocfg.num_node_scalar_features
    mask = torch.ones(N, dtype=torch.bool)
    gam = O.gamma_table(ocfg)
    g = torch.Generator().manual_seed(3)
    ctx_b = torch.randn((B, 1), generator=g) if d["n_ctx"] else None
    ctx = None if ctx_b is None else ctx_b[bi]
    dyn.plan(nn_)
    for s in (999, 500, 1, 0):
        gz = torch.Generator().manual_seed(100 + s)
        z = torch.randn((N, 3 + F), generator=gz) * (1.0 if s > 100 else 0.3)
        z[:, :3] = O.centralize(z[:, :3], bi, B, mask)
        want, eps = O.sample_p_zs_given_zt(W, ocfg, gam, s / 1000, (s + 1) / 1000, z, bi, B, mask, ctx, O.TapeNoise(300 + s))
        raw = _raw_noise(300 + s, N, F).to(dev)
        # (a) fused step through the C ABI
        zc = z.to(dev).contiguous()
        cdev = None if ctx is None else ctx.to(dev).contiguous()
        st = lib.gcdm_sample_step(h, C.c_void_p(zc.data_ptr()), None if cdev is None else C.c_void_p(cdev.data_ptr()), s, 1000,
                                  C.c_void_p(raw.data_ptr()), C.c_uint64(0), None, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert st == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        scale = max(1.0, want.abs().max().item())
        assert (zc.cpu() - want).abs().max().item() <= TOL * scale, f"fused step s={s}"
        # (b) reference-signature method (torch algebra + HIP forward)
        zs = ddpm.sample_p_zs_given_zt(s=torch.full((B, 1), s / 1000, device=dev), t=torch.full((B, 1), (s + 1) / 1000, device=dev),
                                       z=z.to(dev), batch_index=bi.to(dev), node_mask=mask.to(dev), context=cdev, noise=raw)
        assert (zs.cpu() - want).abs().max().item() <= TOL * scale, f"python step s={s}"
        # x stays CoM-free
        for b in range(B):
            assert zc.cpu()[bi == b, :3].sum(0).abs().max().item() < 1e-4 * scale


def test_nan_velocity_is_zeroed_for_the_whole_batch_in_forward_and_in_a_sampler_step():
    """gcpnet.py:1212-1220: a NaN anywhere in vel zeroes vel for the WHOLE batch (the reference warns and carries on).  A model whose last position update has one NaN weight
    -- only vel sees it -- through gcdm_forward (k_finish) and through gcdm_sample_step, whose k_sample launch does k_finish's work since round 6: eps_x = 0 for every node,
    the feature columns as the oracle has them, GCDM flag 1 raised in both.  Exact fp32 MFMA mode (the split-precision mode answers a non-finite value with its range flag and
    the fp32 re-run, which is this path)."""
    d = _dims("qm9")
    cfgs = pkg.default_cfgs("qm9", ())
    W = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=53, scale_2d=0.25)
    W["interaction_layers.8.node_position_update_gcp.vector_up.weight"][0, 0] = float("nan")
    net = pkg.GCPNetDynamics(**cfgs)
    net.load_state_dict(W)
    net = net.cuda().eval()
    ocfg = _ocfg("qm9")
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    dyn.set_mfma_mode(0)
    nn_ = torch.tensor([6, 19, 9])
    B = len(nn_)
    bi = O.num_nodes_to_batch_index(nn_)
    N, F = len(bi), ocfg.num_node_scalar_features
    mask = torch.ones(N, dtype=torch.bool)
    gam = O.gamma_table(ocfg)
    dyn.plan(nn_)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z = torch.randn((N, 3 + F), generator=torch.Generator().manual_seed(5))
    z[:, :3] = O.centralize(z[:, :3], bi, B, mask)
    s = 600
    t = torch.full((N, 1), (s + 1) / 1000)
    want_eps = O.dynamics_forward(W, ocfg, z, t, bi, mask, None)
    assert torch.equal(want_eps[:, :3], torch.zeros(N, 3)) and bool(torch.isfinite(want_eps).all())
    # (a) the network by itself
    zc, tc, out = z.to(dev).contiguous(), t.to(dev).reshape(-1).contiguous(), torch.empty((N, 3 + F), device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    assert lib.gcdm_forward(h, C.c_void_p(zc.data_ptr()), C.c_void_p(tc.data_ptr()), None, C.c_void_p(out.data_ptr()), C.c_void_p(flags.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
    torch.cuda.synchronize()
    assert int(flags.item()) == pkg._native.FLAG_NAN_VEL
    assert torch.equal(out[:, :3].cpu(), torch.zeros(N, 3))
    assert (out[:, 3:].cpu() - want_eps[:, 3:]).abs().max().item() <= TOL * max(1.0, want_eps.abs().max().item())
    # (b) one ancestral step: direct launches (noise tape) and the captured step graph (Philox noise: compared with the direct Philox step, bitwise)
    want, _ = O.sample_p_zs_given_zt(W, ocfg, gam, s / 1000, (s + 1) / 1000, z, bi, B, mask, None, O.TapeNoise(11))
    raw = _raw_noise(11, N, F).to(dev)
    z1 = z.to(dev).contiguous()
    flags.zero_()
    assert lib.gcdm_sample_step(h, C.c_void_p(z1.data_ptr()), None, s, 1000, C.c_void_p(raw.data_ptr()), C.c_uint64(0), C.c_void_p(flags.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
    torch.cuda.synchronize()
    assert int(flags.item()) == pkg._native.FLAG_NAN_VEL
    assert (z1.cpu() - want).abs().max().item() <= TOL * max(1.0, want.abs().max().item())
    res = []
    for graph in (1, 0):
        assert lib.gcdm_set_option(h, b"step_graph", graph) == 0
        z2 = z.to(dev).contiguous()
        flags.zero_()
        assert lib.gcdm_sample_step(h, C.c_void_p(z2.data_ptr()), None, s, 1000, None, C.c_uint64(123), C.c_void_p(flags.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        assert int(flags.item()) == pkg._native.FLAG_NAN_VEL
        res.append(z2.cpu())
    assert torch.equal(res[0], res[1]) and bool(torch.isfinite(res[0]).all())


@pytest.mark.parametrize("shift", [0.0, 0.75])
def test_final_decode_cog_drift_flag_and_reprojection(shift):
    """sample_p_xh_given_z0 + the whole-batch CoG re-projection of mol_gen_sample (variational_diffusion.py:840-907, 1389-1402) through gcdm_sample_final, whose k_sample
    launch also does the network's last stage since round 6: a z_0 whose centroid is off in ONE molecule raises GCDM flag 4 (and nothing else) and comes back with every molecule
    re-centred, exactly as the oracle decides; a centred z_0 raises nothing and is left alone.  Discrete outputs identical."""
    net, W, cfgs = _net("qm9", seed=47, scale=0.25)
    ocfg = _ocfg("qm9")
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.tensor([5, 19, 8])
    B = len(nn_)
    bi = O.num_nodes_to_batch_index(nn_)
    N, F = len(bi), ocfg.num_node_scalar_features
    mask = torch.ones(N, dtype=torch.bool)
    gam = O.gamma_table(ocfg)
    dyn.plan(nn_)
    assert lib.gcdm_set_option(h, b"cog_fix", 1) == 0
    z = torch.randn((N, 3 + F), generator=torch.Generator().manual_seed(9)) * 0.3
    z[:, :3] = O.centralize(z[:, :3], bi, B, mask)
    z[bi == 1, 0] += shift                                    # molecule 1 drifts along x
    x, one_hot, charges = O.sample_p_xh_given_z0(W, ocfg, gam, z, bi, B, mask, None, O.TapeNoise(77))
    cog = torch.zeros(B, 3).index_add_(0, bi, x).abs().max().item()
    assert (cog > 5e-2) == (shift > 0)
    if cog > 5e-2:
        x = O.centralize(x, bi, B, mask)
    raw = _raw_noise(77, N, F).to(dev)
    zc, out = z.to(dev).contiguous(), torch.empty((N, 3 + F), device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    st = lib.gcdm_sample_final(h, C.c_void_p(zc.data_ptr()), None, C.c_void_p(raw.data_ptr()), C.c_uint64(0), C.c_void_p(out.data_ptr()), C.c_void_p(flags.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0, lib.gcdm_last_error(h)
    torch.cuda.synchronize()
    out = out.cpu()
    assert int(flags.item()) == (pkg._native.FLAG_COG_DRIFT if shift > 0 else 0)
    scale = max(1.0, x.abs().max().item())
    assert (out[:, :3] - x).abs().max().item() <= TOL * scale
    assert torch.equal(out[:, 3:3 + ocfg.num_atom_types], one_hot.float())
    if ocfg.include_charges:
        assert torch.equal(out[:, -1:], charges.float())
    if shift > 0:                                             # EVERY molecule re-centred, not only the one that drifted
        assert torch.zeros(B, 3).index_add_(0, bi, out[:, :3]).abs().max().item() < 1e-4 * scale


@pytest.mark.parametrize("case", ["qm9", "geom"])
def test_free_running_sampling_short(case):
    """mol_gen_sample (init + T' steps + final decode) on the same noise tape as the oracle: continuous outputs within the
    relative bar, discrete outputs (one-hot atom types, rounded charges) identical (contract (iii)/(iv))."""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=43, scale=0.25)
    ocfg = _ocfg(case)
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    nn_ = torch.tensor([7, 19, 4, 12])
    N, F = int(nn_.sum()), ocfg.num_node_scalar_features
    Tp = 10
    want, bi = O.mol_gen_sample(W, ocfg, nn_, O.TapeNoise(1234), num_timesteps=Tp)
    tape = O.TapeNoise(1234)
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1) for _ in range(Tp + 2)]
    out, bi2, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, noise_fn=lambda k: draws[k])
    out = out.cpu()
    assert torch.equal(bi2.cpu(), bi)
    scale = max(1.0, want[:, :3].abs().max().item())
    assert (out[:, :3] - want[:, :3]).abs().max().item() <= TOL * scale
    assert torch.equal(out[:, 3:], want[:, 3:])
    assert (ddpm.last_flags & 1) == 0


@pytest.mark.parametrize("case,sizes,K", [("qm9", [19] * 40, 2), ("qm9", [5, 19, 7, 29, 3, 12, 19, 8, 21, 4, 17], 3), ("geom", [44] * 12 + [7, 91], 2)])
def test_sampling_in_lanes_equals_single_handle(case, sizes, K):
    """One flat batch sampled as K slices on K handles / streams (mol_gen_sample(lanes=K)): same Philox noise, flat-batch neighbours read
    across the seams -> the samples of the single-handle run (continuous part within the bar: only the edge-tile boundaries move)."""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=43, scale=0.25)
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    nn_ = torch.tensor(sizes)
    kw = dict(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=8, norm_with_original_timesteps=True, seed=99)
    one, bi1, _ = ddpm.mol_gen_sample(**kw)
    one = one.clone()
    assert (ddpm.last_flags & pkg._native.FLAG_F16_RANGE) == 0
    many, bi2, _ = ddpm.mol_gen_sample(lanes=K, **kw)
    assert (ddpm.last_flags & pkg._native.FLAG_F16_RANGE) == 0 and len(ddpm._lanes) == K      # really ran in K split-precision lanes
    assert torch.equal(bi1, bi2)
    scale = max(1.0, one[:, :3].abs().max().item())
    assert (one[:, :3] - many[:, :3]).abs().max().item() <= TOL * scale
    assert torch.equal(one[:, 3:], many[:, 3:])
    # a single forward through a slice handle reproduces the whole-batch forward on its rows, including the seam nodes' orientations
    dyn, lib, h = ddpm._native(torch.device("cuda"))
    xh, t, bi, _, _ = synth.make_inputs(sizes, synth.dims_feat(d), seed=5, t_value=0.37)
    want = _fwd(net, xh, t, bi)
    ln = ddpm._lanes[0]
    cut = len(sizes) // 2
    n0 = int(sum(sizes[:cut]))
    part = torch.tensor(sizes[cut:], dtype=torch.int32)
    assert ln.lib.gcdm_plan_batch(ln.h, len(part), C.c_void_p(part.data_ptr())) == 0
    for name, val in ((b"flat_prev", 1), (b"flat_next", 0), (b"node_base", n0)):
        assert ln.lib.gcdm_set_option(ln.h, name, val) == 0
    xd, td = xh.cuda().contiguous(), t.reshape(-1).cuda().contiguous()
    od = torch.zeros_like(xd)
    D = xd.shape[1]
    st = ln.lib.gcdm_forward(ln.h, C.c_void_p(xd.data_ptr() + 4 * n0 * D), C.c_void_p(td.data_ptr() + 4 * n0), None, C.c_void_p(od.data_ptr() + 4 * n0 * D), None,
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert st == 0
    torch.cuda.synchronize()
    for name in (b"flat_prev", b"flat_next", b"node_base"):
        ln.lib.gcdm_set_option(ln.h, name, 0)
    assert (od[n0:].cpu() - want[n0:]).abs().max().item() <= TOL * max(1.0, want.abs().max().item())
    ddpm.release_lanes()


def test_sampling_with_chain_frames():
    """mol_gen_sample(return_frames=5): un-normalised intermediate frames (gcdm_unnormalize_z) + final decode without CoG re-projection,
    against the oracle (pinned by tests/golden/chain_small_qm9.npz) on the same noise tape."""
    net, W, cfgs = _net("qm9", seed=43, scale=0.25)
    ocfg = _ocfg("qm9")
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    nn_ = torch.tensor([7, 19, 4, 12])
    N, F = int(nn_.sum()), ocfg.num_node_scalar_features
    Tp, RF = 10, 5
    want, bi = O.mol_gen_sample(W, ocfg, nn_, O.TapeNoise(1234), num_timesteps=Tp, return_frames=RF)
    tape = O.TapeNoise(1234)
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1) for _ in range(Tp + 2)]
    out, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, return_frames=RF, noise_fn=lambda k: draws[k])
    out = out.cpu()
    assert out.shape == want.shape == (RF, N, 3 + F)
    scale = max(1.0, want.abs().max().item())
    assert (out[1:] - want[1:]).abs().max().item() <= TOL * scale
    assert (out[0, :, :3] - want[0, :, :3]).abs().max().item() <= TOL * scale and torch.equal(out[0, :, 3:], want[0, :, 3:])
    with pytest.raises(AssertionError):
        ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=10, return_frames=3)
    # fix_noise=True (sample_sweep_conditionally): every draw's x-part centred over the whole flat batch, tape and Philox noise
    wantf, _ = O.mol_gen_sample(W, ocfg, nn_, O.TapeNoise(1234), num_timesteps=Tp, fix_noise=True)
    outf, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=Tp, fix_noise=True, noise_fn=lambda k: draws[k])
    outf = outf.cpu()
    sc = max(1.0, wantf[:, :3].abs().max().item())
    assert (outf[:, :3] - wantf[:, :3]).abs().max().item() <= TOL * sc and torch.equal(outf[:, 3:], wantf[:, 3:])
    dyn, lib, h = ddpm._native(torch.device("cuda"))
    dyn.plan(nn_)
    z = torch.empty((N, 3 + F), device="cuda")
    assert lib.gcdm_set_option(h, b"fix_noise", 1) == 0
    assert lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, C.c_uint64(3), C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
    assert lib.gcdm_set_option(h, b"fix_noise", 0) == 0
    torch.cuda.synchronize()
    assert z[:, :3].sum(0).abs().max().item() < 1e-4 and z[:7, :3].sum(0).abs().max().item() > 1e-2     # zero mean over the batch, not per molecule


@pytest.mark.parametrize("orig", [False, True])
def test_mol_gen_optimize_matches_oracle(orig):
    """Property-guided optimisation loop (variational_diffusion.py:1416-1546; oracle pinned by tests/golden/optimize_small_qm9cond.npz):
    normalize -> T' steps -> decode on the alpha-conditional QM9 model, both time normalisations, same noise tape."""
    d = _dims("qm9cond")
    net, W, cfgs = _net("qm9cond", seed=47, scale=0.25)
    ocfg = _ocfg("qm9cond")
    model_ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    nn_ = torch.tensor([7, 19, 4, 12])
    B, N, F = len(nn_), int(nn_.sum()), ocfg.num_node_scalar_features
    g = torch.Generator().manual_seed(9)
    ctx_b = torch.randn((B, 1), generator=g)
    samples = []
    for n in nn_.tolist():
        x = torch.randn((n, 3), generator=g) * 1.2
        x = x - x.mean(0, keepdim=True)
        samples.append((x, torch.nn.functional.one_hot(torch.randint(0, F, (n,), generator=g), F).float()))
    Tp = 9
    X, Hc = torch.cat([s_[0] for s_ in samples]), torch.cat([s_[1] for s_ in samples])
    want, bi = O.mol_gen_optimize(W, ocfg, X, Hc, nn_, O.TapeNoise(77), context=ctx_b, num_timesteps=Tp, norm_with_original_timesteps=orig)
    tape = O.TapeNoise(77)
    draws = [torch.cat((tape(N, 3), tape(N, F)), dim=-1) for _ in range(Tp + 1)]
    out, bi2, _ = model_ddpm.mol_gen_optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_nodes=nn_, device="cuda", num_timesteps=Tp,
                                              context=ctx_b.cuda(), norm_with_original_timesteps=orig, noise_fn=lambda k: draws[k])
    out = out.cpu()
    assert torch.equal(bi2.cpu(), bi)
    scale = max(1.0, want[:, :3].abs().max().item())
    assert (out[:, :3] - want[:, :3]).abs().max().item() <= TOL * scale
    assert torch.equal(out[:, 3:], want[:, 3:])
    # un-centred input: the reference's assert_mean_zero_with_mask
    bad = [(x.cuda() + 0.5, h_.cuda()) for x, h_ in samples]
    with pytest.raises(AssertionError):
        model_ddpm.mol_gen_optimize(samples=bad, num_nodes=nn_, device="cuda", num_timesteps=2, context=ctx_b.cuda())
    # the whole-model stand-in
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    model.ddpm.dynamics_network.load_state_dict(W)
    model = model.cuda()
    x2, oh2, ch2, bi3 = model.optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_timesteps=Tp, num_nodes=nn_, context=ctx_b.cuda(),
                                       norm_with_original_timesteps=orig, noise_fn=lambda k: draws[k])
    assert (x2.cpu() - want[:, :3]).abs().max().item() <= TOL * scale and torch.equal(oh2.cpu(), want[:, 3:]) and ch2.numel() == 0
    wantf, _ = O.mol_gen_optimize(W, ocfg, X, Hc, nn_, O.TapeNoise(77), context=ctx_b, num_timesteps=Tp, norm_with_original_timesteps=orig, return_frames=3)
    # the whole-model driver with chain frames (:674-731): XYZ files of molecule 1's chain (3 frames + the last one 10 more times), frame 0 returned
    import tempfile, glob as _glob
    with tempfile.TemporaryDirectory() as td:
        x3, oh3, ch3, _ = model.optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_timesteps=Tp, num_nodes=nn_, context=ctx_b.cuda(),
                                         norm_with_original_timesteps=orig, noise_fn=lambda k: draws[k], return_frames=3, sampling_output_dir=td,
                                         optim_property="alpha", iteration_index=0, chain_viz_batch_element_idx=1, verbose=False)
        files = sorted(_glob.glob(os.path.join(model.last_chain_dir, "*.xyz")))
        assert len(files) == 13 and ch3.numel() == 0
        assert len(open(files[0]).read().splitlines()) == int(nn_[1]) + 2            # XYZ: count, blank, one line per atom
    # chain frames (return_frames = 3 of 9 steps, :1490-1497, 1540-1546; the oracle's frames are pinned by the reference's own in
    # optimize_small_qm9cond.npz, run c): fused loop, and the general loop on the module path
    assert (x3.cpu() - wantf[0, :, :3]).abs().max().item() <= TOL * max(1.0, wantf.abs().max().item()) and torch.equal(oh3.cpu(), wantf[0, :, 3:])
    for path in ("auto", "modules"):
        net.path = path
        try:
            fr, _, _ = model_ddpm.mol_gen_optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_nodes=nn_, device="cuda", num_timesteps=Tp,
                                                   context=ctx_b.cuda(), norm_with_original_timesteps=orig, noise_fn=lambda k: draws[k], return_frames=3)
        finally:
            net.path = "auto"
        fr = fr.cpu()
        assert fr.shape == wantf.shape == (3, N, 3 + F)
        assert (fr - wantf).abs().max().item() <= TOL * max(1.0, wantf.abs().max().item()), path
        assert torch.equal(fr[0, :, 3:], wantf[0, :, 3:])
        if path == "modules":                      # ... and the frame-less call on the module path equals the fused one
            net.path = "modules"
            try:
                o2, _, _ = model_ddpm.mol_gen_optimize(samples=[(x.cuda(), h_.cuda()) for x, h_ in samples], num_nodes=nn_, device="cuda", num_timesteps=Tp,
                                                       context=ctx_b.cuda(), norm_with_original_timesteps=orig, noise_fn=lambda k: draws[k])
            finally:
                net.path = "auto"
            assert (o2.cpu()[:, :3] - want[:, :3]).abs().max().item() <= TOL * scale and torch.equal(o2.cpu()[:, 3:], want[:, 3:])


def test_philox_noise_statistics_and_determinism():
    net, W, cfgs = _net("qm9", seed=45, scale=0.25)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.full((512,), 19)
    dyn.plan(nn_)
    N = int(nn_.sum())
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z1 = torch.empty((N, 9), device=dev); z2 = torch.empty((N, 9), device=dev); z3 = torch.empty((N, 9), device=dev)
    for z, seed in ((z1, 7), (z2, 7), (z3, 8)):
        assert lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, C.c_uint64(seed), stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(z1, z2) and not torch.equal(z1, z3)
    hpart = z1[:, 3:]
    assert abs(hpart.mean().item()) < 0.02 and abs(hpart.std().item() - 1.0) < 0.02
    assert abs(torch.mean(hpart ** 4).item() - 3.0) < 0.15              # normal kurtosis
    xs = z1[:, :3].view(512, 19, 3).sum(1)
    assert xs.abs().max().item() < 1e-5                                  # CoM-free x-noise
    assert abs(z1[:, :3].std().item() - math.sqrt(18 / 19)) < 0.02


# ---- BASELINE.json full sizes: size-independent properties ------------------------------------------------------
def _rot(seed=0):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case,B,n", [("qm9", 1024, 19), ("qm9cond", 1024, 19), ("geom", 256, 44), ("qm9", 64, 19)])
def test_full_size_properties(case, B, n, mode):
    """At the benchmark configurations (C2 / C4): (1) run-to-run determinism, (2) per-molecule zero CoM of vel,
    (3) SE(3) equivariance (rotation + translation; reflections are not a symmetry of the frames),
    (4) batch-composition independence: a molecule's output in the 1024-batch equals its output in the 3-molecule flat
    batch of itself and its two neighbours (which fixes the flat-batch orientation context) -- exercises tile boundaries,
    and (5) agreement with the CPU oracle on a 3-molecule slice."""
    d = _dims(case)
    net, W, _ = _net(case, seed=51, scale=0.5, mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41, n_ctx=d["n_ctx"])
    out = _fwd(net, xh, t, bi, ctx)
    out2 = _fwd(net, xh, t, bi, ctx)
    assert torch.isfinite(out).all()
    assert torch.equal(out, out2)                                        # (1)  n <= 64: rows split over at most 2 tiles
    scale = max(1.0, out.abs().max().item())
    assert out[:, :3].view(B, n, 3).sum(1).abs().max().item() <= 1e-4 * scale   # (2)
    R = _rot(5)
    xr = xh.clone()
    xr[:, :3] = xh[:, :3] @ R.T + torch.tensor([0.3, -1.1, 2.0])
    outr = _fwd(net, xr, t, bi, ctx)
    assert (outr[:, :3] - out[:, :3] @ R.T).abs().max().item() <= TOL * scale      # (3)
    assert (outr[:, 3:] - out[:, 3:]).abs().max().item() <= TOL * scale
    for b in (1, B // 2, B - 2):                                                  # (4) + (5)
        lo, hi = (b - 1) * n, (b + 2) * n
        cs = None if ctx is None else ctx[lo:hi]
        sub = _fwd(net, xh[lo:hi], t[lo:hi], bi[lo:hi] - (b - 1), cs)
        assert (sub[n:2 * n] - out[b * n:(b + 1) * n]).abs().max().item() <= TOL * scale
        ref = O.dynamics_forward(W, _ocfg(case), xh[lo:hi], t[lo:hi], bi[lo:hi] - (b - 1), None, cs)
        assert (sub - ref).abs().max().item() <= TOL * scale


_FULLSIZE_OUT = {}


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_full_size_every_row_matches_reference_golden(case, mode, golden_dir):
    """BASELINE.json configs[1] / [2] / [3] at FULL size (QM9 1024 x 19, alpha-conditional QM9 1024 x 19, GEOM-Drugs 256 x 44): EVERY row of the forward -- all
    19 456 / 19 456 / 11 264 nodes, i.e. every tile of every persistent workgroup, every XCD range boundary -- against the REFERENCE's own forward of the same
    batch (tests/golden/fullsize_<case>.npz: the reference run in fp64 in the build container, make_fullsize_golden.py; weights / inputs are synth seeds 51 / 77, as
    in test_full_size_properties).  Bar: 1e-4 * max(1, |out|) (north_star); both matrix modes; and the two modes -- two independent kernel families -- against
    each other on the full batch."""
    g = np.load(os.path.join(golden_dir, f"fullsize_{case}.npz"))
    B, n = int(g["B"]), int(g["n"])
    d = _dims(case)
    net, W, _ = _net(case, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]), mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs([n] * B, synth.dims_feat(d), seed=int(g["input_seed"]), t_value=float(g["t_value"]), n_ctx=d["n_ctx"])
    assert abs(float(xh.double().sum().item()) - float(g["xh_checksum"])) <= 1e-6 * max(1.0, abs(float(g["xh_checksum"])))      # the inputs the fixture was made from
    out = _fwd(net, xh, t, bi, ctx)
    ref = torch.tensor(g["out64"])
    assert out.shape == ref.shape == (B * n, 3 + synth.dims_feat(d))
    bar = TOL * max(1.0, ref.abs().max().item())
    err = (out - ref).abs()
    worst_row = int(err.max(dim=1).values.argmax())
    print(f"full size {case} ({'f16x3' if mode else 'f32'}): max |hip - ref64| over {out.shape[0]} rows = {err.max().item():.3e} (row {worst_row}, molecule {worst_row // n}); "
          f"reference's own |fp32 - fp64| = {float(g['ref32_vs_ref64_maxabs']):.3e}; bar {bar:.1e}")
    assert torch.isfinite(out).all() and err.max().item() <= bar
    # fp32-class, not merely inside the bar: within 20 x the reference's own fp32-vs-fp64 distance on this batch
    assert err.max().item() <= 20.0 * float(g["ref32_vs_ref64_maxabs"]) + 1e-6
    _FULLSIZE_OUT[(case, mode)] = out
    other = _FULLSIZE_OUT.get((case, 1 - mode))
    if other is not None:
        assert (out - other).abs().max().item() <= bar


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_full_size_every_row_through_the_two_slice_handles_matches_reference_golden(case, mode, golden_dir):
    """The configuration bench.py TIMES is not one handle: `mol_gen_sample(lanes=2)` / `_SlicedBatch` cut the flat batch into two slices of molecules, each planned on
    its own handle with the seam options (flat_prev / flat_next / node_base) set, and every step evaluates the network once per slice.  Here the same full-size inputs
    as test_full_size_every_row_matches_reference_golden go through exactly those two handles (`_SlicedBatch` builds them; one gcdm_forward per slice on its own
    stream, rows addressed as the sampler addresses them) and EVERY row -- the seam rows 9 727 / 9 728 (C2, C3) and 5 631 / 5 632 (C4) included -- is held to the
    reference's own fp64 forward of the flat batch.  (gcpnet.py:1069-1232: the orientations of a node read its flat-batch neighbours.)"""
    g = np.load(os.path.join(golden_dir, f"fullsize_{case}.npz"))
    B, n = int(g["B"]), int(g["n"])
    d = _dims(case)
    net, W, cfgs = _net(case, seed=int(g["weight_seed"]), scale=float(g["weight_scale"]), mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs([n] * B, synth.dims_feat(d), seed=int(g["input_seed"]), t_value=float(g["t_value"]), n_ctx=d["n_ctx"])
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    dev = torch.device("cuda")
    ctx_mol = None if ctx is None else ctx.view(B, n, -1)[:, 0].contiguous()            # per molecule, as mol_gen_sample takes it
    sb = ddpm._SlicedBatch(ddpm, torch.full((B,), n, dtype=torch.int32), dev, ctx_mol, 11, 2)
    assert len(sb.sl) == 2 and sb.sl[1]["n0"] == (B // 2) * n
    xd, td = xh.to(dev).contiguous(), t.reshape(-1).to(dev).contiguous()
    od = torch.full_like(xd, float("nan"))
    start = torch.cuda.Event()
    start.record(torch.cuda.current_stream())
    for k, w in enumerate(sb.sl):
        ln, n0 = w["lane"], w["n0"]
        assert ln.lib.gcdm_get_option(ln.h, b"flat_prev") == int(k > 0) and ln.lib.gcdm_get_option(ln.h, b"flat_next") == int(k < 1)
        assert ln.lib.gcdm_get_option(ln.h, b"node_base") == n0 and ln.lib.gcdm_get_option(ln.h, b"mfma_mode") == mode
        ln.stream.wait_event(start)
        st = ln.lib.gcdm_forward(ln.h, sb._row(xd, n0), C.c_void_p(td.data_ptr() + 4 * n0), sb._cptr(n0), sb._row(od, n0), None, w["stream"])
        assert st == 0, ln.lib.gcdm_last_error(ln.h)
    torch.cuda.synchronize()
    sb.close()
    out = od.cpu()
    ref = torch.tensor(g["out64"])
    bar = TOL * max(1.0, ref.abs().max().item())
    err = (out - ref).abs()
    seam = sb.sl[1]["n0"]
    print(f"full size {case} ({'f16x3' if mode else 'f32'}) through 2 slice handles: max |hip - ref64| = {err.max().item():.3e}; seam rows {seam - 1}, {seam}: "
          f"{err[seam - 1].max().item():.3e}, {err[seam].max().item():.3e}; bar {bar:.1e}")
    assert torch.isfinite(out).all() and err.max().item() <= bar
    assert err.max().item() <= 20.0 * float(g["ref32_vs_ref64_maxabs"]) + 1e-6
    one = _FULLSIZE_OUT.get((case, mode))
    if one is not None:                                  # and against the one-handle forward of the same build (only the edge-tile boundaries move)
        assert (out - one).abs().max().item() <= bar
    ddpm.release_lanes()


@pytest.mark.parametrize("case,B,n", [("qm9", 1024, 19), ("geom", 256, 44)])
def test_full_size_captured_step_and_sliced_steps_equal_direct_single_handle(case, B, n):
    """At the benchmark sizes (round 5 checked these at <= 40 molecules): (a) gcdm_sample_step served by the handle's captured hipGraph is BITWISE the direct-launch
    step (device cursor table, every kernel of a step, 4 steps with on-device Philox noise); (b) the same 4 steps taken by the two slice handles of `_SlicedBatch`
    (double-buffered latent, seam options, global Philox index) give the single-handle latent up to the edge-tile boundaries (1e-4 bar) -- the stepper bench.py times.
    (variational_diffusion.py:1204-1278)"""
    d = _dims(case)
    net, W, cfgs = _net(case, seed=51, scale=0.25, mode=1)
    ds = "geom" if case == "geom" else "qm9"
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    nn_ = torch.full((B,), n, dtype=torch.int32)
    dyn.plan(nn_)
    N, D = B * n, 3 + _ocfg(case).num_node_scalar_features
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    seed = C.c_uint64(77)
    steps = (999, 998, 997, 996)

    def run(graph):
        assert lib.gcdm_set_option(h, b"step_graph", graph) == 0
        z = torch.empty((N, D), device=dev)
        assert lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, seed, stream) == 0
        before = lib.gcdm_get_option(h, b"graph_launches")
        for s in steps:
            assert lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), None, s, 1000, None, seed, C.c_void_p(fl.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        return z, lib.gcdm_get_option(h, b"graph_launches") - before

    direct, n0 = run(0)
    graphed, n1 = run(1)
    assert n0 == 0 and n1 == len(steps)
    assert torch.isfinite(direct).all() and torch.equal(direct, graphed) and int(fl.item()) == 0
    sb = ddpm._SlicedBatch(ddpm, nn_, dev, None, 77, 2)
    sb.init()
    for s in steps:
        sb.step(s, 1000)
    sb.wait()
    torch.cuda.synchronize()
    zs = sb.bufs[sb.cur]
    scale = max(1.0, direct.abs().max().item())
    err = (zs - direct).abs().max().item()
    print(f"{case} {B} x {n}: captured == direct bitwise over {len(steps)} steps; 2 slices vs one handle: max |dz| = {err:.3e} (bar {TOL * scale:.1e})")
    assert err <= TOL * scale and int(sb.flags.max().item()) == 0
    sb.close()
    ddpm.release_lanes()


@pytest.mark.parametrize("dataset,B,n", [("qm9", 1024, 19), ("geom", 256, 44)])
def test_full_sample_at_benchmark_size(dataset, B, n):
    """BASELINE.json configs[1] / configs[3] END TO END: one complete 1000-step sample + decode (Philox noise, the default f16x3 matrix mode, the
    batch sampled as 2 slices like bench.py does) through `EquivariantVariationalDiffusion.mol_gen_sample` -- what bench.py extrapolates from a few timed
    steps.  Asserts: no device flag (in particular no f16-range overflow, i.e. no hidden fp32 re-run), finite outputs, one-hot atom types,
    zero centre of mass per molecule, and a host-clock wall time within 5 % of 1001 x the per-evaluation cost measured in THIS test on a
    200-step run of the same loop (free-running, untrained SURVEY-8(d) weights: |x| grows to ~1e3, the hardest case for the range guard)."""
    import time
    cfgs = pkg.default_cfgs(dataset)
    torch.manual_seed(0)
    model = (pkg.GEOMMoleculeGenerationDDPM if dataset == "geom" else pkg.QM9MoleculeGenerationDDPM)(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda()
    ddpm = model.ddpm
    nn_ = torch.full((B,), n)
    ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", num_timesteps=5, lanes=2)         # handles, plans, slices
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", num_timesteps=200, seed=3, lanes=2)
    torch.cuda.synchronize()
    per_eval = (time.perf_counter() - t0) / 201
    t0 = time.perf_counter()
    xh, bi, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", seed=7, lanes=2)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    assert ddpm.last_flags == 0, f"device flags {ddpm.last_flags}"
    assert ddpm.dynamics_network.mfma_mode == 1
    assert torch.isfinite(xh).all()
    F = model.num_atom_types
    assert bool((xh[:, 3:3 + F].sum(1) == 1).all()) and bool(((xh[:, 3:3 + F] == 0) | (xh[:, 3:3 + F] == 1)).all())
    com = torch.zeros(B, 3, device=xh.device).index_add_(0, bi, xh[:, :3]).abs().max().item() / n
    assert com <= 1e-4 * max(1.0, xh[:, :3].abs().max().item())
    print(f"full sample {dataset} {B} x {n}: {wall:.2f} s = {B / wall:.1f} molecules/s; 1001 x per-evaluation cost of a 200-step run = {1001 * per_eval:.2f} s")
    assert abs(wall - 1001 * per_eval) <= 0.05 * 1001 * per_eval, (wall, 1001 * per_eval)


NLL_TERMS = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior", "log_pN")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_nll_terms_match_reference_golden(case, mode, golden_dir):
    """Evaluation-mode likelihood terms (EquivariantVariationalDiffusion.forward, variational_diffusion.py:948-1160: two evaluations of the
    network per batch) and the NLL the module assembles from them (qm9_mol_gen_ddpm.py:184-272) against what the REFERENCE's own forward
    returned on the same batch, timesteps and noise tape (tests/golden/nll_full_*.npz), both matrix modes.  Bar per term: 4 x the reference's
    own fp32-vs-fp64 gap + 1e-4 x its magnitude."""
    g = np.load(os.path.join(golden_dir, f"nll_full_{case}.npz"))
    d = _dims(case)
    net, W, cfgs = _net(case, seed=int(g["weight_seed"]), mode=mode)
    ds = {"qm9": "qm9", "qm9cond": "qm9_second_half", "geom": "geom"}[case]       # the conditional experiment trains on the second half of QM9
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info(ds)).cuda().eval()
    dev = torch.device("cuda")
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_).to(dev)
    N, F = int(nn_.sum()), synth.dims_feat(d)
    tape = O.TapeNoise(int(g["noise_seed"]))
    noise = [torch.cat((tape(N, 3), tape(N, F)), dim=-1) for _ in range(2)]          # the reference's call order: x-part, h-part; z_t then z_0
    ctx = torch.tensor(g["ctx"])[bi.cpu()].to(dev) if "ctx" in g.files else None

    def batch():
        return pkg.config.AttrDict(x=torch.tensor(g["x"]).to(dev), one_hot=torch.tensor(g["one_hot"]).to(dev), charges=torch.tensor(g["charges"]).to(dev),
                                   batch=bi, mask=torch.ones(N, dtype=torch.bool, device=dev), props_context=ctx)

    b = batch()
    b.h = {"categorical": b.one_hot, "integer": b.charges}
    b.num_graphs, b.num_nodes_present = len(nn_), nn_.to(dev)
    out = ddpm(b, return_loss_info=True, t_int=torch.tensor(g["t_int"]).view(-1, 1), noise=noise)
    assert (net.read_flags() & pkg._native.FLAG_F16_RANGE) == 0
    for name, got in zip(NLL_TERMS, out[:8]):
        w32, w64 = torch.tensor(g[f"{name}_32"]).double(), torch.tensor(g[f"{name}_64"]).double()
        bar = 4 * (w32 - w64).abs() + 1e-4 * w64.abs().clamp(min=1.0)
        err = (got.double().cpu() - w64).abs()
        assert (err <= bar).all(), (name, err.tolist(), bar.tolist())
    assert torch.equal(out[8].cpu(), torch.tensor(g["t_int"]))
    for k in ("eps_hat_x", "eps_hat_h"):
        assert abs(out[9][k].item() - float(g[f"{k}_64"])) <= 1e-4 * max(1.0, abs(float(g[f"{k}_64"])))
    # the module-level forward: NLL per molecule from the same terms
    cls = pkg.GEOMMoleculeGenerationDDPM if case == "geom" else pkg.QM9MoleculeGenerationDDPM
    model = cls(**cfgs)
    model.ddpm.dynamics_network.load_state_dict(W)
    model = model.cuda().eval()
    model.ddpm.dynamics_network._ensure_handle(dev)
    model.ddpm.dynamics_network.set_mfma_mode(mode)
    nll, info = model(batch(), t_int=torch.tensor(g["t_int"]).view(-1, 1), noise=noise)
    want = O.nll_from_terms({k: torch.tensor(g[f"{k}_64"]).double() for k in NLL_TERMS}, int(cfgs["diffusion_cfg"]["num_timesteps"]))
    assert (nll.double().cpu() - want).abs().max().item() <= 2e-4 * want.abs().max().item()
    assert abs(info["kl_prior"].item() - float(torch.tensor(g["kl_prior_64"]).mean())) <= 1e-4 * max(1.0, abs(float(torch.tensor(g["kl_prior_64"]).mean())))
    metrics = model.validation_step(batch(), t_int=torch.tensor(g["t_int"]).view(-1, 1), noise=noise)
    assert abs(metrics["loss"].item() - want.mean().item()) <= 2e-4 * abs(want.mean().item()) and metrics["log_SNR_max"] > metrics["log_SNR_min"]


def test_overflow_late_in_a_run_resumes_from_a_checkpoint():
    """Range guard inside the sampling loop: an activation that leaves the f16 images at step 900 of a 1000-step run no longer costs a whole
    second trajectory in fp32 (1 + 2.7 runs).  The loop looks at the flag word every RANGE_CHECK_EVERY steps (asynchronous copy, no GPU stall),
    resumes from the last clean snapshot with fp32 MFMA, reports it in `last_flags`, and takes <= 1.3x the clean run; the result is the one an
    all-fp32 run produces (same Philox noise, same perturbation)."""
    import time
    cfgs = pkg.default_cfgs("qm9")
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda().eval()
    ddpm, dyn = model.ddpm, model.ddpm.dynamics_network
    B, n = 256, 19
    nn_ = torch.full((B,), n)
    hit = []

    def poke(s, z):                      # one huge (finite) node feature from step 900 on: beyond 1.3e8 x 2^-11 ... of the f16 images, fine in fp32
        if s == 99:
            z[7, 5] = 3.0e8
            hit.append(s)

    ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", num_timesteps=5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    clean, _, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", seed=11, step_callback=lambda s, z: None)
    torch.cuda.synchronize()
    t_clean = time.perf_counter() - t0
    assert ddpm.last_flags == 0 and ddpm.last_range_rewinds == 0
    t0 = time.perf_counter()
    out, _, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", seed=11, step_callback=poke)
    torch.cuda.synchronize()
    t_poked = time.perf_counter() - t0
    assert ddpm.last_flags & pkg._native.FLAG_F16_RANGE and ddpm.last_range_rewinds == 1 and len(hit) == 2      # s = 99 ran twice: x3, then fp32
    assert dyn.mfma_mode == 1                                  # the handle is back in its default mode
    assert torch.isfinite(out).all()
    print(f"clean run {t_clean:.2f} s, run with an overflow at step 900: {t_poked:.2f} s = {t_poked / t_clean:.2f} x")
    assert t_poked <= 1.3 * t_clean
    dyn.set_mfma_mode(0)
    try:
        hit.clear()
        ref, _, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=nn_, device="cuda", seed=11, step_callback=poke)
    finally:
        dyn.set_mfma_mode(1)
    ok = torch.ones(B * n, dtype=torch.bool, device=out.device)
    ok[7 // n * n:(7 // n + 1) * n] = False                   # the poked molecule itself lives at |z| ~ 1e8: compared relatively below
    scale = max(1.0, ref[ok][:, :3].abs().max().item())
    assert (out[ok][:, :3] - ref[ok][:, :3]).abs().max().item() <= 1e-3 * scale
    rel = (out[~ok][:, :3] - ref[~ok][:, :3]).abs().max().item() / max(1.0, ref[~ok][:, :3].abs().max().item())
    assert rel <= 1e-3
