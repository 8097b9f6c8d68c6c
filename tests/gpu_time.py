"""Wall-clock of one forward at several batch sizes (orientation; run via gpurun)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 64, 256, 1024]
d = synth.DATASET_DIMS[case]
n = 44 if case == "geom" else 19
net = pkg.GCPNetDynamics(**pkg.default_cfgs("geom" if case == "geom" else "qm9"))
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=1, scale_2d=0.5))
net = net.cuda().eval()
dev = torch.device("cuda")
print("cpu_count", os.cpu_count(), torch.cuda.get_device_name(0), flush=True)
for B in sizes:
    xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=1)
    xh, t = xh.to(dev), t.to(dev)
    net._ensure_handle(dev); net.sync_weights(); net.plan(nn_)
    for _ in range(2):
        net.native_forward(xh, t)
    torch.cuda.synchronize()
    reps = 5
    t0 = time.time()
    for _ in range(reps):
        net.native_forward(xh, t)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    E = B * n * n
    print(f"B={B:5d} N={B*n:6d} E={E:8d} tiles={E//64:6d} forward {dt*1e3:9.3f} ms   {dt/(E)*1e9:8.2f} ns/edge", flush=True)

# in-kernel phase breakdown of the edge-message kernel (last layer of the last forward)
import ctypes as C
lib, h = net._lib, net._handle
lib.gcdm_profile_enable(h, 2)
net.native_forward(xh, t)
torch.cuda.synchronize()
nw = 4 if lib.gcdm_get_option(h, b"edge_tile") == 32 else 8
ph = net.debug_read("phase").view(-1, 8, 24)[:, :nw]
lib.gcdm_profile_enable(h, 0)
names = {1: "P1 msg0 pre", 2: "barrier", 3: "PQ add", 4: "GEMM0", 5: "silu", 6: "gate+PG", 7: "barrier", 8: "state store", 9: "barrier",
         12: "k1 GEMM(+vec)+mid barrier", 13: "k1 silu", 14: "k1 gate+PG", 15: "barrier", 16: "k1 state store", 17: "barrier",
         18: "k2,k3 (all) + attention + fp32 image", 19: "(barrier)", 20: "aggregate"}
mean = ph.mean(dim=(0, 1))
print("phase breakdown (shader cycles, mean over tiles x waves; cumulative -> delta); per-wave deltas in brackets:")
prev, prevw = 0.0, torch.zeros(nw)
pw = ph.mean(dim=0)          # [nw, 24]
for i in sorted(names):
    dw = pw[:, i] - prevw
    print(f"  {i:2d} {names[i]:<18s} delta={mean[i]-prev:10.0f}  cum={mean[i]:10.0f}   [" + " ".join(f"{x:7.0f}" for x in dw.tolist()) + "]")
    prev, prevw = mean[i], pw[:, i]

# the same for the per-layer node kernel (last layer with a successor)
lib.gcdm_profile_enable(h, 3)
net.native_forward(xh, t)
torch.cuda.synchronize()
pn = net.debug_read("phase_node").view(-1, 8, 24)
lib.gcdm_profile_enable(h, 0)
nnames = {1: "load agg/h/chi", 2: "barrier", 3: "ff vecmat (mfma)", 4: "barrier+ff pre tail", 5: "barrier", 6: "ff GEMM1 (34 kb)", 7: "silu+store+2 barriers",
          8: "ff GEMM2 (16 kb)", 9: "gate fold+barriers", 10: "h update+vec_finish", 11: "barrier", 12: "pos vecmat+tail+barriers", 13: "pos GEMM (18 kb)",
          14: "pos gate+finish+x", 15: "state write-back", 16: "PQ GEMMs (2x16 kb)", 17: "VDI/VDJ vecmat"}
mean = pn.mean(dim=(0, 1))
pw = pn.mean(dim=0)
print(f"node kernel phases ({pn.shape[0]} tiles; shader cycles, cumulative -> delta); per-wave deltas in brackets:")
prev, prevw = 0.0, torch.zeros(8)
for i in sorted(nnames):
    dw = pw[:, i] - prevw
    print(f"  {i:2d} {nnames[i]:<26s} delta={mean[i]-prev:9.0f}  cum={mean[i]:9.0f}   [" + " ".join(f"{x:6.0f}" for x in dw.tolist()) + "]")
    prev, prevw = mean[i], pw[:, i]
