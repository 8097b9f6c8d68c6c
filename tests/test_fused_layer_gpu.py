"""The fused layer launch (round 6; csrc/gcdm_layer_x3.hip.h): the layer's node tiles as a tail role of the persistent edge-message workgroups.  Same arithmetic in
the same order as the two launches per layer of rounds 1-5 -> every test here asks for BITWISE equality with the un-fused path, which the rest of the suite holds to
the reference (gcpnet.py:834-930 behind :676-737).  Covered: the benchmark sizes, ragged batches, molecules whose node tiles span many edge tiles and XCD
boundaries, repeated launches (self-resetting counters), the captured step graph, two fused handles sharing the GPU (the arrival gate), the plans / modes that do
not qualify, and the host's reaction to GCDM_FLAG_TAIL."""
import ctypes as C
import importlib
import os
import sys
import warnings

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
pytestmark = pytest.mark.gpu


def _net(case, seed=51, scale=0.5):
    d = synth.DATASET_DIMS[case]
    ds = "geom" if case == "geom" else "qm9"
    cond = ("alpha",) if d["n_ctx"] else ()
    cfgs = pkg.default_cfgs(ds, cond)
    net = pkg.GCPNetDynamics(**cfgs)
    net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=seed, scale_2d=scale))
    net = net.cuda().eval()
    net._ensure_handle(torch.device("cuda"))
    return net, d, cfgs


def _fwd(net, xh, t, bi, ctx=None, mask=None):
    dev = torch.device("cuda")
    batch = dict(batch=bi.to(dev), mask=torch.ones(len(bi), dtype=torch.bool, device=dev) if mask is None else mask.to(dev),
                 props_context=None if ctx is None else ctx.to(dev))
    _, out = net(batch, xh.to(dev), t.to(dev))
    torch.cuda.synchronize()
    return out.clone()


def _opt(net, name, value=None):
    if value is None:
        return net._lib.gcdm_get_option(net._handle, name)
    assert net._lib.gcdm_set_option(net._handle, name, value) == 0, net._lib.gcdm_last_error(net._handle)


CASES = [
    ("qm9", [19] * 1024),                                     # BASELINE.json configs[1]: 5 776 edge tiles, 608 node tiles
    ("geom", [44] * 256),                                     # configs[3]
    ("qm9", None),                                            # ragged: 700 molecule sizes from the dataset histogram
    ("geom", [181] * 24 + [3, 90, 7, 181]),                   # 32 nodes of a 181-atom molecule are 91 edge tiles: node tiles span XCD ranges, rows span tiles
    ("qm9cond", [19] * 512 + [5, 29, 1, 12] * 8),             # context-conditioned, single-atom molecules in the mix
]


@pytest.mark.parametrize("case,sizes", CASES, ids=["qm9_1024x19", "geom_256x44", "qm9_ragged", "geom_181", "qm9cond_mixed"])
def test_fused_layer_equals_two_launches_bitwise(case, sizes):
    net, d, cfgs = _net(case)
    if sizes is None:
        ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9"))
        torch.manual_seed(5)
        sizes = ddpm.num_nodes_distribution.sample(700).tolist()
    xh, t, bi, nn_, ctx = synth.make_inputs(sizes, synth.dims_feat(d), seed=77, t_value=0.41, n_ctx=d["n_ctx"])
    assert _opt(net, b"fuse_node") == 1                       # the default of a primary handle
    _opt(net, b"fuse_node", 0)
    want = _fwd(net, xh, t, bi, ctx)
    assert _opt(net, b"fuse_active") == 0
    _opt(net, b"fuse_node", 1)
    for rep in range(3):                                      # counters and cursors reset themselves: every launch starts from zero
        got = _fwd(net, xh, t, bi, ctx)
        assert _opt(net, b"fuse_active") == 1, "the plan should qualify for the fused launch"
        assert net.read_flags() == 0
        assert torch.isfinite(got).all() and torch.equal(got, want), f"repeat {rep}: max |d| = {(got - want).abs().max().item():.3e}"


def test_plans_and_modes_that_do_not_qualify_use_two_launches():
    net, d, _ = _net("qm9")
    xh, t, bi, nn_, _ = synth.make_inputs([19] * 40, synth.dims_feat(d), seed=3, t_value=0.3)     # 226 tiles: not a persistent launch
    small = _fwd(net, xh, t, bi)
    assert _opt(net, b"fuse_active") == 0 and torch.isfinite(small).all()
    xh, t, bi, nn_, _ = synth.make_inputs([19] * 600, synth.dims_feat(d), seed=3, t_value=0.3)
    big = _fwd(net, xh, t, bi)
    assert _opt(net, b"fuse_active") == 1
    mask = torch.ones(len(bi), dtype=torch.bool)
    mask[5::19] = False                                       # masked plans (gcdm_plan_batch_masked): rows without edges
    _fwd(net, xh, t, bi, mask=mask)
    assert _opt(net, b"fuse_active") == 0
    for name, val, back in ((b"mfma_mode", 0, 1), (b"edge_tile", 32, 0), (b"persistent", 0, 1)):
        _opt(net, name, val)
        _fwd(net, xh, t, bi)
        assert _opt(net, b"fuse_active") == 0, name
        _opt(net, name, back)
    again = _fwd(net, xh, t, bi)
    assert _opt(net, b"fuse_active") == 1 and torch.equal(again, big)
    assert net._lib.gcdm_set_option(net._handle, b"fuse_tile", 48) != 0


def test_two_fused_handles_share_the_gpu():
    """Two handles, two streams, both with the fused launch, enqueued back to back so that their grids are dispatched interleaved: the node role is entered only
    when every workgroup of an XCD group has arrived, so nothing waits for a workgroup that holds no CU.  Results bitwise those of the un-fused runs, no flag."""
    dev = torch.device("cuda")
    nets, ins, wants = [], [], []
    for i, (case, sizes) in enumerate((("qm9", [19] * 700), ("geom", [44] * 200))):
        net, d, _ = _net(case, seed=51 + i)
        xh, t, bi, nn_, _ = synth.make_inputs(sizes, synth.dims_feat(d), seed=70 + i, t_value=0.4)
        _opt(net, b"fuse_node", 0)
        wants.append(_fwd(net, xh, t, bi))
        _opt(net, b"fuse_node", 1)
        net.plan(nn_)
        nets.append(net)
        ins.append((xh.to(dev), t.to(dev)))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [[], []]
    torch.cuda.synchronize()
    for rep in range(6):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs[i].append(nets[i].native_forward(*ins[i]).clone())
    torch.cuda.synchronize()
    for i in (0, 1):
        assert nets[i].read_flags() == 0 and _opt(nets[i], b"fuse_active") == 1
        for o in outs[i]:
            assert torch.equal(o, wants[i])


@pytest.mark.parametrize("case,B,n", [("qm9", 1024, 19)])
def test_captured_step_graph_with_the_fused_launch(case, B, n):
    """gcdm_sample_step on the primary handle (fused layer launches inside the captured step graph: the counters reset themselves, the graph replays the same
    arguments): 6 steps bitwise equal to the same steps with two launches per layer."""
    net, d, cfgs = _net(case, scale=0.25)
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).cuda()
    dev = torch.device("cuda")
    dyn, lib, h = ddpm._native(dev)
    dyn.plan(torch.full((B,), n, dtype=torch.int32))
    N, D = B * n, 3 + d["num_atom_types"] + int(d["include_charges"])
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    seed = C.c_uint64(9)

    def run(fuse):
        assert lib.gcdm_set_option(h, b"fuse_node", fuse) == 0
        z = torch.empty((N, D), device=dev)
        assert lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, seed, stream) == 0
        before = lib.gcdm_get_option(h, b"graph_launches")
        for s in range(999, 993, -1):
            assert lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), None, s, 1000, None, seed, C.c_void_p(fl.data_ptr()), stream) == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        assert lib.gcdm_get_option(h, b"graph_launches") - before == 6 and lib.gcdm_get_option(h, b"fuse_active") == fuse
        return z

    two = run(0)
    one = run(1)
    assert torch.isfinite(two).all() and torch.equal(one, two) and int(fl.item()) == 0


def test_tail_flag_switches_the_handle_to_two_launches():
    """GCDM_FLAG_TAIL (a failed placement check or a bounded wait; never observed) comes together with GCDM_FLAG_F16_RANGE, so the callers' fp32 re-run repairs the
    result; the Python side additionally turns the fused launch off on the handle.  The flag word is forged here."""
    net, d, _ = _net("qm9")
    xh, t, bi, nn_, _ = synth.make_inputs([19] * 600, synth.dims_feat(d), seed=3, t_value=0.3)
    want = _fwd(net, xh, t, bi)
    assert _opt(net, b"fuse_node") == 1 and net.read_flags() == 0
    net._flags.fill_(pkg._native.FLAG_TAIL | pkg._native.FLAG_F16_RANGE)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        v = net.read_flags()
    assert v & pkg._native.FLAG_TAIL and _opt(net, b"fuse_node") == 0 and any("fuse_node" in str(x.message) for x in w)
    again = _fwd(net, xh, t, bi)
    assert _opt(net, b"fuse_active") == 0 and torch.equal(again, want)
