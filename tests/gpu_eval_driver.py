"""The reference's evaluation-driver shape end to end: sample_and_analyze with batches of 100, full 1000 steps, host clock (run via gpurun).
    python tests/gpu_eval_driver.py [num_samples] [concurrent_batches]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
pkg = importlib.import_module("bio-diffusion_amd")
num = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfgs = pkg.default_cfgs("qm9")
torch.manual_seed(0)
model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
with torch.no_grad():
    for p in model.ddpm.dynamics_network.parameters():
        if p.dim() == 2:
            p.mul_(0.25)
model = model.cuda()
model.sample_and_analyze(num_samples=2 * K, batch_size=2, num_timesteps=3, concurrent_batches=K)       # warm-up: handles, weights
torch.cuda.synchronize()
torch.manual_seed(1)
t0 = time.perf_counter()
res = model.sample_and_analyze(num_samples=num, batch_size=100, concurrent_batches=K)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"sample_and_analyze: {num} molecules in batches of 100, 1000 steps, {K} batches in flight: {dt:.1f} s -> {num / dt:.1f} molecules/s "
      f"(10 000 samples: {10000 / (num / dt):.0f} s)")
print(res)
