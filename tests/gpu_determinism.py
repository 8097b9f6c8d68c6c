"""Run-to-run bitwise comparison of the per-layer buffers of one forward (debug helper; run via gpurun).

    python tests/gpu_determinism.py [qm9|geom] [B]          (GCDM_EDGE_TILE=32|64, GCDM_MFMA=f16x3|f32 select the kernels)

Prints nothing between the header and "done" when 8 repetitions are bit-identical.
"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
d = synth.DATASET_DIMS[case]
n = 44 if case == "geom" else 19
net = pkg.GCPNetDynamics(**pkg.default_cfgs("geom" if case == "geom" else "qm9"))
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.5))
net = net.cuda().eval()
dev = torch.device("cuda")
xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
xh, t = xh.to(dev), t.to(dev)
net._ensure_handle(dev); net.sync_weights(); net.plan(nn_)
ET = net._lib.gcdm_get_option(net._handle, b"edge_tile")
print("edge_tile", ET, "mfma_mode", net.mfma_mode)
for lim in (1, 2, d["L"]):
    net.debug_set_layer_limit(lim)
    snaps = []
    for rep in range(8):
        net.native_forward(xh, t)
        torch.cuda.synchronize()
        snaps.append({k: net.debug_read(k).clone() for k in ("agg", "h", "chi", "x", "pq")})
    for k in snaps[0]:
        for r in range(1, 8):
            dlt = (snaps[0][k] - snaps[r][k]).abs()
            nz = int((dlt > 0).sum())
            if nz:
                msg = f"layers={lim} {k}: run0 vs run{r}: {nz} differing, max {dlt.max().item():.3e}"
                if k == "agg":
                    nodes = torch.unique(torch.nonzero(dlt.view(-1, 352) > 0)[:, 0])
                    e0 = (nodes // n) * n * n + (nodes % n) * n
                    split = ((e0 // ET) != ((e0 + n - 1) // ET))
                    msg += f"; {len(nodes)} rows, {int(split.sum())} of them cut by a tile boundary; first rows {nodes[:6].tolist()}"
                print(msg)
print("done")
