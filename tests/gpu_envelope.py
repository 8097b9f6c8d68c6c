"""Where does the f16x3 (split-precision) envelope end for weights a trained checkpoint might have?  (round-3 verdict item 9; no checkpoint is
available offline, README.md:120-127.)  QM9 production architecture, 64 molecules x 19 atoms, full 1000-step Philox sampling per point:

  * "all":    every 2-D weight drawn at  s x  the default initialisation (the benchmarks use s = 0.25: SURVEY 8d)
  * "layer4": interaction layer 4 alone at s x default, the rest at 0.25 (a whole layer's matrices, not one outlier element)
  * "bias":   activations pushed up instead: the last feed-forward bias of every layer set to +-b (LayerNorm-free: use_gcp_norm false)

Per point: the exponent split k gcdm_finalize_weights chose, the largest |z| of the final latent, whether the range flag fired, the step the loop
resumed from with fp32 MFMA, the wall time against a clean f16x3 run of the same weights' shape, NaN-in-vel (the reference's own guard:
the network output is not finite in fp32 either).   python tests/gpu_envelope.py  -> table on stdout (gpurun_out/r4_envelope.txt)
"""
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
native = pkg._native
dev = torch.device("cuda")
d = synth.DATASET_DIMS["qm9"]
cfgs = pkg.default_cfgs("qm9")
nn_ = torch.full((64,), 19, dtype=torch.int32)


def model(kind, s):
    torch.manual_seed(0)
    net = pkg.GCPNetDynamics(**cfgs)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() == 2:
                if kind == "all":
                    p.mul_(s)
                elif kind == "layer4":
                    p.mul_(s if name.startswith("interaction_layers.4.") else 0.25)
                else:
                    p.mul_(0.25)
            elif kind == "bias" and name.endswith("feedforward_network.0.scalar_out.2.bias"):
                p.copy_(torch.where(torch.arange(p.numel()) % 2 == 0, s, -s).to(p.dtype))
    net = net.to(dev).eval()
    ddpm = pkg.EquivariantVariationalDiffusion(net, cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9")).to(dev)
    return net, ddpm


def run(kind, s):
    net, ddpm = model(kind, s)
    wmax = max(float(p.detach().abs().max()) for p in net.parameters() if p.dim() == 2)
    dyn, lib, h = ddpm._native(dev)
    k = int(lib.gcdm_get_option(h, b"x3_shift"))
    mode = dyn.mfma_mode
    res = {}
    for tag, m in (("x3", 1), ("f32", 0)):
        if m == 1 and mode == 0:
            res[tag] = None
            continue
        dyn.set_mfma_mode(m)
        ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=dev, num_timesteps=20, seed=5)      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device=dev, seed=5)
        torch.cuda.synchronize()
        res[tag] = dict(s=time.perf_counter() - t0, flags=int(ddpm.last_flags), resume=getattr(ddpm, "last_range_resume_step", None),
                        rewinds=int(getattr(ddpm, "last_range_rewinds", 0)), finite=bool(torch.isfinite(out).all()), zmax=float(out[:, :3].abs().max()), out=out)
    dyn.set_mfma_mode(1 if mode == 1 else 0)
    x3, f32 = res["x3"], res["f32"]
    agree = None
    if x3 is not None and x3["finite"] and f32["finite"]:
        agree = float((x3["out"][:, :3] - f32["out"][:, :3]).abs().max() / max(1e-30, f32["out"][:, :3].abs().max()))
    fl = lambda r: "-" if r is None else "".join(c for c, b in (("N", 1), ("M", 2), ("C", 4), ("R", 8)) if r["flags"] & b) or "0"
    print(f"{kind:7s} {s:8.3g}  max|W| {wmax:8.3g}  k={k}  mode={'f16x3' if mode else 'f32 only'}   "
          f"x3: flags {fl(x3):3s} resume@{('-' if x3 is None or x3['resume'] is None else x3['resume'])!s:>4} "
          f"{('-' if x3 is None else '%.2f s' % x3['s']):>7}   f32: flags {fl(f32):3s} {f32['s']:.2f} s   max|x| {f32['zmax']:9.3g}   "
          f"|x3 - f32| / max|x| {('-' if agree is None else '%.1e' % agree)}", flush=True)
    net.release()


if __name__ == "__main__":
    print("flags: N = NaN in vel (zeroed, as the reference does), C = CoG drift re-projected, R = f16 range left: part of the run in fp32 MFMA; "
          "resume@s = the step the loop went back to", flush=True)
    for s in (0.25, 0.35, 0.5, 0.7, 1.0, 2.0, 4.0, 16.0, 64.0):
        run("all", s)
    for s in (1.0, 4.0, 16.0, 64.0, 256.0):
        run("layer4", s)
    for s in (1.0, 1e2, 1e4, 1e5, 3e5, 1e6, 1e7, 1e8):
        run("bias", s)
