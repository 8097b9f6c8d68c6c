"""Is the FIRST tile of a persistent edge-message workgroup slower than its later ones (cold weights in the XCD's L2 at the start of every layer's launch)?
Needs a -DGCDM_STAMPS build as bio-diffusion_amd/libgcdm_hip.so (tools/build_variants.sh stamps:-DGCDM_STAMPS).  python tests/gpu_first_tile.py [qm9|geom]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
case = sys.argv[1] if len(sys.argv) > 1 else "qm9"
d = synth.DATASET_DIMS[case]
B, n = (256, 44) if case == "geom" else (1024, 19)
net = pkg.GCPNetDynamics(**pkg.default_cfgs("geom" if case == "geom" else "qm9"))
net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=51, scale_2d=0.25))
net = net.cuda().eval()
dev = torch.device("cuda")
net._ensure_handle(dev); net.sync_weights()
lib, h = net._lib, net._handle
xh, t, bi, nn_, _ = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41)
net.plan(nn_)
xh, t = xh.to(dev), t.to(dev)
for _ in range(3):
    net.native_forward(xh, t)
assert lib.gcdm_profile_enable(h, 2) == 0, "needs a -DGCDM_STAMPS build"
net.native_forward(xh, t)
torch.cuda.synchronize()
ph = net.debug_read("phase").view(-1, 8, 24)          # the LAST layer's launch: [tiles][waves][stamps]
lib.gcdm_profile_enable(h, 0)
tiles = ph.shape[0]
G, cus = tiles, 256
base_, rem_ = G >> 3, G & 7
stride = cus // 8
end = ph[:, :, 20].mean(1)                             # end-of-tile stamp (cycles since the tile's start)
gemm1 = (ph[:, :, 12] - ph[:, :, 9]).mean(1)
rows = {}
for x in range(8):
    cnt = base_ + (1 if x < rem_ else 0)
    start = x * base_ + min(x, rem_)
    for tt in range(cnt):
        j = tt // stride
        rows.setdefault(j, []).append(start + tt)
print(f"{case}: {tiles} tiles, {stride} workgroups per XCD")
for j in sorted(rows):
    idx = torch.tensor(rows[j])
    print(f"  tile #{j:2d} of a workgroup ({len(idx):4d} tiles): end-of-tile {end[idx].mean().item():8.0f} cycles   GEMM k=1 phase {gemm1[idx].mean().item():7.0f}")
