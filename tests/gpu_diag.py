"""Stage-by-stage comparison of the HIP path with the CPU oracle on a real MI355X (run via gpurun).
Prints one line per internal buffer with the max abs difference; does not stop at the first mismatch.

    python tests/gpu_diag.py [qm9|geom|qm9cond ...]
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import synth  # noqa: E402
from oracle import gcdm_oracle as O  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")


def line(name, got, want):
    got, want = got.float().cpu(), want.float().cpu()
    if got.shape != want.shape:
        print(f"  {name:<22s} SHAPE MISMATCH got {tuple(got.shape)} want {tuple(want.shape)}")
        return 1e9
    d = (got - want).abs()
    bad = int((~torch.isfinite(got)).sum())
    print(f"  {name:<22s} max|d|={d.max().item():.3e}  mean|d|={d.mean().item():.3e}  max|ref|={want.abs().max().item():.3e}  nonfinite={bad}")
    return d.max().item()


def un_g4(buf, groups, n):     # [groups][n][4] -> [n][4*groups]
    return buf.view(groups, n, 4).permute(1, 0, 2).reshape(n, 4 * groups)


def run_case(case, num_nodes=(5, 19, 3, 11)):
    d = synth.DATASET_DIMS[case]
    ds = "geom" if case == "geom" else "qm9"
    cond = ("alpha",) if d["n_ctx"] else ()
    print(f"=== {case}: num_nodes={list(num_nodes)}")
    cfgs = pkg.default_cfgs(ds, cond)
    net = pkg.GCPNetDynamics(**cfgs)
    shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    W = synth.make_weights(shapes, seed=17)
    net.load_state_dict(W)
    net = net.cuda().eval()
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=9, n_ctx=d["n_ctx"])
    ocfg = O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"], num_context=d["n_ctx"],
                          num_layers=d["L"], norm_values=d["norm_values"])
    ref, it = O.dynamics_forward(W, ocfg, xh, t, bi, None, ctx, return_intermediates=True)
    N, E = len(bi), len(it["row"])
    dev = torch.device("cuda")
    batch = dict(batch=bi.to(dev), mask=torch.ones(N, dtype=torch.bool, device=dev), props_context=None if ctx is None else ctx.to(dev))
    net._ensure_handle(dev)
    net.sync_weights()
    net.plan(nn_)
    worst = 0.0
    # --- embedding stage only
    net.debug_set_layer_limit(0)
    net.native_forward(xh.to(dev), t.to(dev), None if ctx is None else ctx.to(dev))
    torch.cuda.synchronize()
    fr = net.debug_read("frames").view(9, E).t().reshape(E, 3, 3)
    worst = max(worst, line("frames", fr, it["frames"]))
    u = net.debug_read("u").view(3, E).t()
    al = net.debug_read("alpha").view(d["Ve"], E).t()
    worst = max(worst, line("xi' = alpha*u", al[:, :, None] * u[:, None, :], it["xi"]))
    worst = max(worst, line("e'", un_g4(net.debug_read("ep"), d["Se"] // 4, E), it["e"]))
    fbar = net.debug_read("fbar").view(9, N).t()
    fm = torch.zeros(N, 9).index_add_(0, it["row"], it["frames"].reshape(E, 9)) / torch.bincount(it["row"]).float()[:, None]
    worst = max(worst, line("fbar", fbar, fm))
    worst = max(worst, line("chi0", net.debug_read("chi0").view(2, 3, N).permute(2, 0, 1), O.orientations(xh[:, :3])))
    worst = max(worst, line("x_central", net.debug_read("x").view(3, N).t(), it["x_central"]))
    worst = max(worst, line("h_embed", un_g4(net.debug_read("h"), 64, N), it["h_embed"]))
    worst = max(worst, line("chi_embed", net.debug_read("chi").view(32, 3, N).permute(2, 0, 1), it["chi_embed"]))
    # PQ halves of layer 0
    S, Se = 256, d["Se"]
    ws = W["interaction_layers.0.interaction.message_fusion.0.scalar_out.weight"]
    bs = W["interaction_layers.0.interaction.message_fusion.0.scalar_out.bias"]
    Pw = it["h_embed"] @ ws[:, :S].T + bs
    Qw = it["h_embed"] @ ws[:, S + Se:2 * S + Se].T
    worst = max(worst, line("PQ (layer0 halves)", un_g4(net.debug_read("pq"), 128, N), torch.cat((Pw, Qw), dim=1)))
    # --- layer by layer
    h, chi, x = it["h_embed"], it["chi_embed"], it["x_central"]
    for l in range(d["L"]):
        net.debug_set_layer_limit(l + 1)
        net.native_forward(xh.to(dev), t.to(dev), None if ctx is None else ctx.to(dev))
        torch.cuda.synchronize()
        a_s, a_v = O.message_passing(W, f"interaction_layers.{l}.interaction.", h, chi, it["e"], it["xi"], it["row"], it["col"], it["frames"], ocfg)
        agg = net.debug_read("agg").view(N, 352)
        worst = max(worst, line(f"L{l} agg.s", agg[:, :256], a_s))
        worst = max(worst, line(f"L{l} agg.v", agg[:, 256:].reshape(N, 32, 3), a_v))
        h, chi, x = it[f"h_{l}"], it[f"chi_{l}"], it[f"x_{l}"]
        worst = max(worst, line(f"L{l} h", un_g4(net.debug_read("h"), 64, N), h))
        worst = max(worst, line(f"L{l} chi", net.debug_read("chi").view(32, 3, N).permute(2, 0, 1), chi))
        worst = max(worst, line(f"L{l} x", net.debug_read("x").view(3, N).t(), x))
        if l >= 1 and "-v" not in sys.argv:
            pass
    net.debug_set_layer_limit(-1)
    _, out = net(batch, xh.to(dev), t.to(dev).view(-1, 1))
    torch.cuda.synchronize()
    worst = max(worst, line("net_out (vs oracle)", out, ref))
    gpath = os.path.join(ROOT, "tests", "golden", f"dyn_full_{case}.npz")
    if os.path.exists(gpath) and tuple(num_nodes) == (5, 19, 3, 11):
        g = np.load(gpath)
        line("net_out (vs ref fp32)", out, torch.tensor(g["out32"]))
        line("net_out (vs ref fp64)", out, torch.tensor(g["out64"]))
    print(f"  flags={net.read_flags()}  worst={worst:.3e}")
    # timing of one forward on this small batch (launch-bound) for orientation
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        net.native_forward(xh.to(dev), t.to(dev), None if ctx is None else ctx.to(dev))
    torch.cuda.synchronize()
    print(f"  small-batch forward: {(time.time() - t0) / 5 * 1e3:.2f} ms")
    return worst


if __name__ == "__main__":
    cases = [a for a in sys.argv[1:] if not a.startswith("-")] or ["qm9", "geom", "qm9cond"]
    print("device:", torch.cuda.get_device_name(0))
    w = 0.0
    for c in cases:
        w = max(w, run_case(c))
        if c == "qm9":
            w = max(w, run_case(c, num_nodes=(70, 2, 1, 33, 64, 29)))   # rows longer than a 64-edge tile, tiny molecules
    print("WORST", w)
