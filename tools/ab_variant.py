"""A/B helper for kernel variants (run via gpurun; the library under test is whatever bio-diffusion_amd/libgcdm_hip.so is at the moment):

    python tools/ab_variant.py <tag> [qm9|geom] [compare-to-tag]

Prints one line: sha256 of a full-size forward output and of the latent after 3 Philox sampler steps (bit-identity check between
variants), and the per-step time of 40 sampler steps on one handle (HIP events on the launch stream).  The forward output is also
kept in /tmp/ab_<tag>_<case>.pt; with a third argument the line additionally carries max |out - out_of_that_tag| / max |out| (for
variants that change the summation order: expect ~1e-6, the parity bar is 1e-4).

Typical call (libraries pre-built in the build container under build/ab/, which travels to the GPU box):
    for v in base new; do cp build/ab/libgcdm_$v.so bio-diffusion_amd/libgcdm_hip.so; python tools/ab_variant.py $v qm9 base; done
"""
import ctypes as C
import hashlib
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
tag = sys.argv[1]
case = sys.argv[2] if len(sys.argv) > 2 else "qm9"
d = synth.DATASET_DIMS[case]
B, n = (256, 44) if case == "geom" else (1024, 19)
cfgs = pkg.default_cfgs("geom" if case == "geom" else "qm9")
net = pkg.GCPNetDynamics(**cfgs)
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.25))
net = net.cuda()
ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("geom" if case == "geom" else "qm9")).cuda()
dev = torch.device("cuda")
xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
dyn, lib, h = ddpm._native(dev)
dyn.plan(nn_)
out = dyn.native_forward(xh.to(dev), t.to(dev))
torch.cuda.synchronize()
h_fwd = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
torch.save(out.cpu(), f"/tmp/ab_{tag}_{case}.pt")
rel = ""
if len(sys.argv) > 3 and os.path.exists(f"/tmp/ab_{sys.argv[3]}_{case}.pt"):
    other = torch.load(f"/tmp/ab_{sys.argv[3]}_{case}.pt")
    rel = f" rel_vs_{sys.argv[3]}={(out.cpu() - other).abs().max().item() / other.abs().max().item():.2e}"
N, D = xh.shape
z = torch.empty((N, D), device=dev)
flags = torch.zeros(1, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
zp, fp, sd = C.c_void_p(z.data_ptr()), C.c_void_p(flags.data_ptr()), C.c_uint64(7)
assert lib.gcdm_sample_init(h, zp, None, sd, st) == 0
for s in (999, 998, 997):
    assert lib.gcdm_sample_step(h, zp, None, s, 1000, None, sd, fp, st) == 0
torch.cuda.synchronize()
h_z = hashlib.sha256(z.cpu().numpy().tobytes()).hexdigest()[:16]
for s in range(996, 986, -1):
    lib.gcdm_sample_step(h, zp, None, s, 1000, None, sd, fp, st)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
torch.cuda.synchronize()
ev[0].record()
K = 40
for i in range(K):
    lib.gcdm_sample_step(h, zp, None, 986 - i, 1000, None, sd, fp, st)
ev[1].record()
torch.cuda.synchronize()
# in-kernel phase stamps of the edge kernel (shader cycles: independent of the clock the box settles at)
cyc = ""
if os.environ.get("GCDM_AB_STAMPS", "1") != "0" and lib.gcdm_profile_enable(h, 2) == 0:
    dyn.native_forward(xh.to(dev), t.to(dev))
    torch.cuda.synchronize()
    nw = 4 if lib.gcdm_get_option(h, b"edge_tile") == 32 else 8
    ph = dyn.debug_read("phase").view(-1, 8, 24)[:, :nw].mean(dim=(0, 1))
    lib.gcdm_profile_enable(h, 0)
    cyc = " tile_cycles=%d phases=[%s]" % (ph[20], " ".join("%d:%d" % (i, ph[i]) for i in (1, 2, 4, 6, 7, 8, 9, 12, 14, 15, 16, 17, 18, 19, 20)))
# launch durations of the two kernel families (HIP events recorded by the library on the launch stream; clock dependent: compare alternating runs on one box)
kms = ""
if lib.gcdm_profile_enable(h, 1) == 0:
    te = tn = 0.0
    ne = 0
    for i in range(5):
        lib.gcdm_sample_step(h, zp, None, 940 - i, 1000, None, sd, fp, st)
        ms_, nl_ = C.c_double(), C.c_int32()
        lib.gcdm_profile_edge_kernel_ms(h, C.byref(ms_), C.byref(nl_)); te += ms_.value; ne += nl_.value
        lib.gcdm_profile_node_kernel_ms(h, C.byref(ms_), C.byref(nl_)); tn += ms_.value
    lib.gcdm_profile_enable(h, 0)
    kms = f" edge_ms={te / max(ne, 1):.4f} node_ms={tn / max(ne, 1):.4f}"
print(f"AB {tag} {case} edge_tile={lib.gcdm_get_option(h, b'edge_tile')} fwd={h_fwd} z3={h_z} ms_per_step={ev[0].elapsed_time(ev[1]) / K:.4f} flags={int(flags.item())}{rel}{kms}{cyc}")
