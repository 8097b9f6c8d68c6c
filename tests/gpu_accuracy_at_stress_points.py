"""Accuracy of the two matrix modes at the stress points of tests/gpu_envelope.py, against the CPU oracle in fp64 on the SAME input (one network
evaluation on the latent of step 500 of the fp32-mode trajectory): is a split-precision / fp32 discrepancy rounding noise of the same class as
plain fp32's, or a loss of accuracy?   python tests/gpu_accuracy_at_stress_points.py"""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402
from oracle import gcdm_oracle as O  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
src = open(os.path.join(ROOT, "tests", "gpu_envelope.py")).read()
exec(src[src.index("native = pkg._native"):src.index("def run(")])

torch.set_num_threads(min(32, os.cpu_count() or 1))
for kind, s in (("all", 0.25), ("layer4", 1.0), ("layer4", 4.0), ("layer4", 8.0), ("bias", 100.0), ("bias", 3000.0)):
    net, ddpm = model(kind, s)
    dyn, lib, h = ddpm._native(dev)
    got = {}
    dyn.set_mfma_mode(0)
    ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=dev, seed=5, num_timesteps=1000,
                        step_callback=lambda s_, z: got.__setitem__(s_, z.detach().clone()) if s_ == 500 else None)
    z = got[500]
    bi = torch.repeat_interleave(torch.arange(len(nn_), device=dev), nn_.to(dev).long())
    t = torch.full((z.shape[0], 1), 0.5, device=dev)
    outs = {}
    for m in (1, 0):
        dyn.set_mfma_mode(m)
        lib.gcdm_clear_flags(h) if hasattr(lib, "gcdm_clear_flags") else None
        outs[m] = dyn.native_forward(z, t).cpu().double()
    fl = dyn.read_flags()
    dyn.set_mfma_mode(1)
    W = {k_: v.detach().cpu() for k_, v in net.state_dict().items()}
    ocfg = O.OracleConfig(num_layers=d["L"])
    r32 = O.dynamics_forward(W, ocfg, z.cpu(), t.cpu(), bi.cpu()).double()
    r64 = O.dynamics_forward({k_: v.double() for k_, v in W.items()}, ocfg, z.cpu().double(), t.cpu().double(), bi.cpu())
    sc = r64.abs().max().item()
    e = lambda a: (a - r64).abs().max().item() / sc
    print(f"{kind:7s} {s:7.4g}  k={int(lib.gcdm_get_option(h, b'x3_shift'))} flags={fl}  max|z| {z.abs().max().item():.2e} max|out| {sc:.2e}   error vs fp64 oracle / max|out|:  "
          f"f16x3 {e(outs[1]):.2e}   fp32 MFMA {e(outs[0]):.2e}   fp32 oracle (torch CPU) {e(r32):.2e}", flush=True)
    net.release()
