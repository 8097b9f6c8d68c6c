"""Phase time stamps of the per-layer node kernel for both tile sizes (needs a -DGCDM_STAMPS build as bio-diffusion_amd/libgcdm_hip.so):
python tests/gpu_node_phases.py [qm9|geom] [B]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
d = synth.DATASET_DIMS[case]
n = 44 if case == "geom" else 19
B = int(sys.argv[2]) if len(sys.argv) > 2 else (256 if case == "geom" else 1024)
net = pkg.GCPNetDynamics(**pkg.default_cfgs("geom" if case == "geom" else "qm9"))
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.25))
net = net.cuda().eval()
dev = torch.device("cuda")
net._ensure_handle(dev); net.sync_weights()
lib, h = net._lib, net._handle
xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
net.plan(nn_)
xh, t = xh.to(dev), t.to(dev)
names = {1: "load agg/h/chi", 2: "barrier", 3: "ff vecmat (mfma)", 4: "ff pre tail", 5: "barrier", 6: "ff GEMM1 (34 kb)", 7: "silu+store+2 barriers",
         8: "ff GEMM2 (16 kb)", 9: "gate fold+barriers", 10: "h update+vec_finish", 11: "barrier", 12: "pos vecmat+tail+barriers", 13: "pos GEMM (18 kb)",
         14: "pos gate+finish+x", 15: "state write-back", 16: "PQ GEMMs (2x16 kb)", 17: "VDI/VDJ vecmat"}
for nt in (32, 64):
    assert lib.gcdm_set_option(h, b"node_tile", nt) == 0
    for _ in range(3):
        net.native_forward(xh, t)
    assert lib.gcdm_profile_enable(h, 3) == 0
    net.native_forward(xh, t)
    torch.cuda.synchronize()
    tiles = (B * n + nt - 1) // nt
    pn = net.debug_read("phase_node").view(-1, 8, 24)[:tiles]
    lib.gcdm_profile_enable(h, 0)
    mean, pw = pn.mean(dim=(0, 1)), pn.mean(dim=0)
    print(f"{case} B={B}: node kernel, {nt}-node tiles ({tiles} tiles; shader cycles, cumulative -> delta); per-wave deltas in brackets:")
    prev, prevw = 0.0, torch.zeros(8)
    for i in sorted(names):
        dw = pw[:, i] - prevw
        print(f"  {i:2d} {names[i]:<26s} delta={mean[i]-prev:9.0f}  cum={mean[i]:9.0f}   [" + " ".join(f"{x:6.0f}" for x in dw.tolist()) + "]")
        prev, prevw = mean[i], pw[:, i]
