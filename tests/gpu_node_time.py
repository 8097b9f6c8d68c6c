"""Launch time of the per-layer node kernel for both tile sizes (HIP events of the library): python tests/gpu_node_time.py [qm9|geom]"""
import ctypes as C
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
d = synth.DATASET_DIMS[case]
n = 44 if case == "geom" else 19
cfgs = pkg.default_cfgs("geom" if case == "geom" else "qm9")
net = pkg.GCPNetDynamics(**cfgs)
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.25))
net = net.cuda()
dev = torch.device("cuda")
net._ensure_handle(dev)
net.sync_weights()
lib, h = net._lib, net._handle
for B in ((1024, 862, 431, 215) if case == "qm9" else (256, 186, 93)):
    xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
    net.plan(nn_)
    xh, t = xh.to(dev), t.to(dev)
    line = f"{case} B={B:5d} N={B * n:6d}: "
    for nt in (32, 64):
        assert lib.gcdm_set_option(h, b"node_tile", nt) == 0
        for _ in range(3):
            net.native_forward(xh, t)
        lib.gcdm_profile_enable(h, 1)
        tot_e = tot_n = 0.0
        cnt = 0
        for _ in range(5):
            net.native_forward(xh, t)
            ms, nl = C.c_double(), C.c_int32()
            lib.gcdm_profile_edge_kernel_ms(h, C.byref(ms), C.byref(nl)); tot_e += ms.value; cnt += nl.value
            lib.gcdm_profile_node_kernel_ms(h, C.byref(ms), C.byref(nl)); tot_n += ms.value
        lib.gcdm_profile_enable(h, 0)
        line += f"  node_tile {nt}: node kernel {tot_n / cnt * 1e3:7.1f} us ({(B * n + nt - 1) // nt} tiles), edge kernel {tot_e / cnt * 1e3:7.1f} us;"
    print(line, flush=True)
