"""Diagnostic (run via gpurun): error of one ragged forward against the CPU oracle per molecule, both matrix modes; then the edge
embedding (e') of the split-precision kernel against the fp32 kernel over 6 repeats, listing the waves (32 edges) that differ.

    python tests/gpu_ragged_diag.py [case] [n1,n2,...]      (default: geom 181,3,90 -- 320 workgroups of the embedding kernel)
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import test_gpu_parity as T
import synth
O = T.O
case = sys.argv[1] if len(sys.argv) > 1 else "geom"
num_nodes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [181, 3, 90]
d = T._dims(case)
for mode in (1, 0):
    net, W, _ = T._net(case, seed=23, scale=0.5, mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=31, t_value=0.63, n_ctx=d["n_ctx"])
    ref = O.dynamics_forward(W, T._ocfg(case), xh, t, bi, None, ctx)
    out = T._fwd(net, xh, t, bi, ctx)
    err = (out - ref).abs()
    scale = max(1.0, ref.abs().max().item())
    per_node = err.max(dim=1).values
    worst = per_node.argmax().item()
    print(f"mode={mode} max err={err.max().item():.3e} scale={scale:.3e} rel={err.max().item()/scale:.3e} finite={torch.isfinite(out).all().item()} worst node={worst} (mol {bi[worst].item()})",
          "err by molecule:", [f"{per_node[bi == m].max().item():.2e}" for m in range(len(num_nodes))], flush=True)


keep = {}
Se, Ve = d["Se"], d["Ve"]
for mode in (0, 1):
    net, W, _ = T._net(case, seed=23, scale=0.5, mode=mode)
    xh, t, bi, nn_, ctx = synth.make_inputs(num_nodes, synth.dims_feat(d), seed=31, t_value=0.63, n_ctx=d["n_ctx"])
    for rep in range(6 if mode else 1):
        T._fwd(net, xh, t, bi, ctx)
        E = net.debug_read("u").numel() // 3
        ep = T._un_g4(net.debug_read("ep"), Se // 4, E)
        if mode == 0:
            ep0 = ep.clone()
        else:
            bad_e = ((ep - ep0).abs().max(dim=1).values > 1e-4).nonzero().flatten()
            print("x3 repeat", rep, "bad edges", bad_e.numel(), "bad waves:", sorted(set((bad_e // 32).tolist()))[:24], flush=True)
