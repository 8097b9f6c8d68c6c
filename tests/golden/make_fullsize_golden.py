"""Golden outputs of the REFERENCE's own forward at the BENCHMARK sizes (BASELINE.json configs[1] / [2] / [3]: QM9 1024 x 19, QM9 alpha-conditional
1024 x 19, GEOM-Drugs 256 x 44), so that the GPU suite can compare EVERY row of a full-size batch -- every tile of every persistent workgroup, every XCD
range boundary, the seam of the two slices -- with something reference-derived (missing #2).

Run in the build container only (needs /root/reference; CPU, ~3 min):

    python tests/golden/make_fullsize_golden.py

Weights and inputs are NOT stored: they are `synth.make_weights(seed=51, scale_2d=0.5)` / `synth.make_inputs([n] * B, seed=77, t_value=0.41)`, the ones
tests/test_gpu_parity.py::test_full_size_properties has always used.  Stored: the reference's output run in fp64 (rounded to fp32: 6e-8 relative, three orders
below the 1e-4 bar) and the largest |fp32 run - fp64 run| of the reference itself, for scale.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import make_golden as mg  # noqa: E402
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "16")))
CASES = [("qm9", 1024, 19), ("qm9cond", 1024, 19), ("geom", 256, 44)]


def main():
    for case, B, n in CASES:
        ds, cond, cfgs = mg.cfgs_for(case)
        d = synth.DATASET_DIMS[case]
        net = rh.build_reference_dynamics(cfgs, seed=0)
        shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
        net.load_state_dict(synth.make_weights(shapes, seed=51, scale_2d=0.5))
        xh, t, bi, nn_, ctx = synth.make_inputs([n] * B, synth.dims_feat(d), seed=77, t_value=0.41, n_ctx=d["n_ctx"])
        t0 = time.time()
        out32, _, _ = mg.run_ref_forward(net, xh, t, bi, ctx, torch.float32)
        t1 = time.time()
        out64, _, _ = mg.run_ref_forward(net, xh, t, bi, ctx, torch.float64)
        t2 = time.time()
        gap = (out32.double() - out64).abs().max().item()
        print(f"{case}: N={len(bi)} fp32 {t1 - t0:.1f} s, fp64 {t2 - t1:.1f} s, max|out|={out64.abs().max().item():.4g}, max|ref32-ref64|={gap:.3g}")
        mg.npz(f"fullsize_{case}", B=B, n=n, weight_seed=51, weight_scale=0.5, input_seed=77, t_value=0.41, out64=out64.float(), ref32_vs_ref64_maxabs=gap,
               xh_checksum=float(xh.double().sum().item()))


if __name__ == "__main__":
    main()
