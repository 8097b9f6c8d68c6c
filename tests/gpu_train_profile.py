"""Where the time of one training step on the module path goes: torch profiler over forward + loss + backward of a 64-molecule batch (run via gpurun)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("bio-diffusion_amd")
dev = torch.device("cuda")
cfgs = pkg.default_cfgs("qm9")
torch.manual_seed(0)
net = pkg.GCPNetDynamics(**cfgs)
with torch.no_grad():
    for p in net.parameters():
        if p.dim() == 2:
            p.mul_(0.25)
net = net.to(dev)
ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(dev).train()
B, n = 64, 19
nt = torch.full((B,), n, device=dev)
N = B * n
bt = torch.repeat_interleave(torch.arange(B, device=dev), nt)
x = torch.randn(N, 3, device=dev)
cat = torch.nn.functional.one_hot(torch.randint(0, 5, (N,), device=dev), 5).float()
integ = torch.randint(0, 9, (N,), device=dev).float()
tb = pkg.config.AttrDict(x=x, batch=bt, mask=torch.ones(N, dtype=torch.bool, device=dev), props_context=None, h={"categorical": cat, "integer": integ}, num_graphs=B, num_nodes_present=nt)
def once():
    for p_ in ddpm.parameters():
        p_.grad = None
    terms = ddpm(tb)
    (terms[1] + terms[3] + terms[4]).mean().backward()
for _ in range(3):
    once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    once()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / 5 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(2):
        once()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=22, max_name_column_width=70))
