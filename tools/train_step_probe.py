"""Where a training step of the module path spends its time (GPU box): wall clock per step against the GPU-side kernel time of the same steps.

    python tools/train_step_probe.py [steps]                       -> wall ms / step, HIP-event ms / step, launches per step (torch profiler-free: counted by rocprofv3)
    rocprofv3 --kernel-trace --stats -- python tools/train_step_probe.py 5      (sum of kernel durations / steps = GPU busy time per step)

The step is bench.py's `training_step`: forward in training mode + loss + backward of one 64-molecule QM9 batch through libgcdm_ops.so's operators."""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def build_step():
    import synth
    pkg = importlib.import_module("bio-diffusion_amd")
    dev = torch.device("cuda", 0)
    d = synth.DATASET_DIMS["qm9"]
    cfgs = pkg.default_cfgs("qm9", ())
    torch.manual_seed(0)
    net = pkg.GCPNetDynamics(**cfgs)
    with torch.no_grad():
        for p in net.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    net = net.to(dev)
    info = pkg.dataset_info("qm9")
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], info).to(dev)
    ddpm._native(dev)
    Bt, n = 64, 19
    nt_ = torch.full((Bt,), n, dtype=torch.long, device=dev)
    Nt = Bt * n
    bt = torch.repeat_interleave(torch.arange(Bt, device=dev), nt_)
    g = torch.Generator().manual_seed(5)
    types = torch.randint(0, d["num_atom_types"], (Nt,), generator=g)
    tb = pkg.config.AttrDict(x=torch.randn((Nt, 3), generator=g).to(dev), batch=bt, mask=torch.ones(Nt, dtype=torch.bool, device=dev), props_context=None,
                             h={"categorical": torch.nn.functional.one_hot(types, d["num_atom_types"]).float().to(dev),
                                "integer": (torch.randint(1, 10, (Nt,), generator=g).float().to(dev) if d["include_charges"] else torch.zeros((Nt, 0), device=dev))},
                             num_graphs=Bt, num_nodes_present=nt_)
    ddpm.train()

    def once():
        for p_ in ddpm.parameters():
            p_.grad = None
        terms = ddpm(tb)
        (terms[1] + terms[3] + terms[4]).mean().backward()

    return once, dev


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    once, dev = build_step()
    for _ in range(3):
        once()
    torch.cuda.synchronize(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    t0 = time.perf_counter()
    ev[0].record()
    for _ in range(steps):
        once()
    ev[1].record()
    t_host = (time.perf_counter() - t0) / steps * 1e3          # host done enqueueing
    torch.cuda.synchronize(dev)
    wall = (time.perf_counter() - t0) / steps * 1e3
    print(f"training step, 64 x 19: wall {wall:.2f} ms / step, host enqueue {t_host:.2f} ms / step, HIP events {ev[0].elapsed_time(ev[1]) / steps:.2f} ms / step")


if __name__ == "__main__":
    main()
