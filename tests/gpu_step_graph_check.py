"""Captured step (hipGraph) against direct launches: bit equality over a run with a rewind, and the per-step host / wall cost of both.
python tests/gpu_step_graph_check.py [qm9|geom] [B ...]"""
import ctypes as C, importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
sizes = [int(x) for x in sys.argv[2:]] or [2, 64, 100]
d = synth.DATASET_DIMS[case]
n = 44 if case == "geom" else 19
net = pkg.GCPNetDynamics(**pkg.default_cfgs("geom" if case == "geom" else "qm9"))
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.25))
net = net.cuda().eval()
dev = torch.device("cuda")
net._ensure_handle(dev); net.sync_weights()
lib, h = net._lib, net._handle
gamma = torch.linspace(-6.0, 6.0, 1001)
assert lib.gcdm_set_gamma(h, C.c_void_p(gamma.data_ptr()), 1001) == 0
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in sizes:
    xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
    net.plan(nn_)
    z0 = (0.1 * torch.randn(B * n, xh.shape[1])).to(dev)
    fl = torch.zeros(1, dtype=torch.int32, device=dev)
    outs = {}
    for g in (0, 1):
        assert lib.gcdm_set_option(h, b"step_graph", g) == 0
        z = z0.clone()
        zp = C.c_void_p(z.data_ptr())
        seq = list(range(999, 979, -1)) + list(range(990, 970, -1))       # 20 steps, a rewind of 10, 20 more
        for s in seq:
            assert lib.gcdm_sample_step(h, zp, None, s, 1000, None, 7, C.c_void_p(fl.data_ptr()), st) == 0, lib.gcdm_last_error(h)
        torch.cuda.synchronize()
        outs[g] = z.clone()
        K = 200
        t0 = time.perf_counter()
        for s in range(K):
            lib.gcdm_sample_step(h, zp, None, 800 - s, 1000, None, 7, C.c_void_p(fl.data_ptr()), st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{case} B={B:4d} step_graph={g} (effective {lib.gcdm_get_option(h, b'step_graph')}, graph launches {lib.gcdm_get_option(h, b'graph_launches')}): "
              f"host enqueue {(t1 - t0) / K * 1e3:.3f} ms/step, wall {(t2 - t0) / K * 1e3:.3f} ms/step", flush=True)
    same = torch.equal(outs[0], outs[1])
    print(f"   bit-identical after 40 steps with a rewind: {same}; finite: {bool(torch.isfinite(outs[1]).all())}; flags {int(fl.item())}", flush=True)
    assert same
