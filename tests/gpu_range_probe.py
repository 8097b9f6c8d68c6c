import importlib, os, sys, ctypes as C
import torch
sys.path[:0] = ["/root/repo", "/root/repo/tests"]
pkg = importlib.import_module("bio-diffusion_amd")
cfgs = pkg.default_cfgs("qm9")
torch.manual_seed(0)
net = pkg.GCPNetDynamics(**cfgs)
with torch.no_grad():
    for p in net.parameters():
        if p.dim() == 2: p.mul_(0.25)
net = net.cuda().eval()
ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9"))
dev = torch.device("cuda")
dyn, lib, h = ddpm._native(dev)
B = 256
dyn.plan(torch.full((B,), 19, dtype=torch.int32))
N = B * 19
z = torch.empty((N, 9), device=dev)
fl = torch.zeros(1, dtype=torch.int32, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), None, C.c_uint64(1), st)
for s in reversed(range(1000)):
    lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), None, s, 1000, None, C.c_uint64(1), C.c_void_p(fl.data_ptr()), st)
    if s % 50 == 0 or int(fl.item()) & 8:
        print("s", s, "flags", int(fl.item()), "max|z_x|", round(z[:, :3].abs().max().item(), 2), "max|z_h|", round(z[:, 3:].abs().max().item(), 2), flush=True)
        if int(fl.item()) & 8:
            break
