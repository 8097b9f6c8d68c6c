"""BASELINE.json configs[0] as specified, on the CPU, in full: 64 QM9 molecules x 19 atoms, 1000 DDPM steps + decode (1001 network evaluations).

    python tests/cpu_full_config0.py reference     # the UNMODIFIED reference imported under the test stubs (build container only)
    python tests/cpu_full_config0.py oracle        # the CPU restatement (what bench.py's cpu_baseline leg samples 21 steps of)

Weights: default init under torch.manual_seed(0), 2-D parameters x 0.25.  Prints wall-clock, ms / step and molecules / s.
Takes ~20-40 minutes; not part of any test run (the numbers are quoted in DESIGN.md section 4).
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
which = sys.argv[1] if len(sys.argv) > 1 else "oracle"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
torch.set_num_threads(threads)
B, n, T = 64, 19, 1000
num_nodes = torch.full((B,), n, dtype=torch.long)

if which == "reference":
    import ref_harness as rh
    from make_golden import cfgs_for
    ds, cond, cfgs = cfgs_for("qm9")
    net = rh.build_reference_dynamics(cfgs, seed=0, weight_scale=0.25)
    ddpm = rh.build_reference_ddpm(cfgs, net, ds)
    torch.manual_seed(1)
    t0 = time.time()
    with torch.no_grad():
        xh, bi, _ = ddpm.mol_gen_sample(num_samples=B, num_nodes=num_nodes, device="cpu", num_timesteps=T)
    dt = time.time() - t0
else:
    import synth
    from oracle import gcdm_oracle as O
    import importlib
    pkg = importlib.import_module("bio-diffusion_amd")
    d = synth.DATASET_DIMS["qm9"]
    torch.manual_seed(0)
    net = pkg.GCPNetDynamics(**pkg.default_cfgs("qm9"))
    W = {k: (v * 0.25 if v.dim() == 2 else v).clone() for k, v in net.state_dict().items()}
    cfg = O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"], num_context=0, num_layers=d["L"], norm_values=d["norm_values"])
    t0 = time.time()
    xh, bi = O.mol_gen_sample(W, cfg, num_nodes, O.TapeNoise(1), num_timesteps=T)
    dt = time.time() - t0
ok = bool(torch.isfinite(xh).all())
print(f"{which}: {B} molecules x {n} atoms, {T} steps + decode on {threads} CPU threads ({os.cpu_count()} cores visible): {dt:.1f} s = {dt / (T + 1) * 1e3:.1f} ms / step "
      f"= {B / dt:.4f} molecules / s; outputs finite: {ok}")
