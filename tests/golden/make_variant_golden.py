"""Golden fixtures for the NON-PRODUCTION configurations of the path's Hydra surface, produced by the REFERENCE itself (build container only):

    python tests/golden/make_variant_golden.py        ->  tests/golden/dyn_variant_<name>.npz   (synth.VARIANTS)

For every variant the unmodified ``GCPNetDynamics`` (src/models/components/gcpnet.py: GCP :33-262 / GCP2 :265-491 with frame_gate, sigma_frame_gate,
residuals, ablations; GCPLayerNorm; vector-sum position updates; other message / feed-forward depths, widths and nonlinearities) is built at reduced
width under the stubs of ref_harness, loaded with seed-recreated weights (synth.make_weights over ITS OWN state-dict shapes), and evaluated on a
ragged 4-molecule batch in fp32 and fp64; one extra case per variant has masked nodes.  Stored: inputs and outputs only.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(4)
WEIGHT_SEED = 83


def run(name, dtype, xh, t, bi, mask):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        gcp, _, _ = rh.import_reference()
        cfgs = rh.load_reference_cfgs("qm9", ())
        synth.apply_variant(cfgs, name, gcp_classes={"GCP": gcp.GCP, "GCP2": gcp.GCP2})
        net = rh.build_reference_dynamics(cfgs, seed=0)
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        net.load_state_dict(synth.make_weights(shapes, seed=WEIGHT_SEED, scale_2d=0.7))
        net = net.to(dtype).eval()
        with torch.no_grad():
            _, out = net(rh.make_batch(bi, mask, None), xh.to(dtype), t.to(dtype))
        return out, shapes
    finally:
        torch.set_default_dtype(prev)


def main():
    assert rh.reference_available(), "reference checkout not found"
    sizes = [5, 9, 3, 12]
    xh, t, bi, nn_, _ = synth.make_inputs(sizes, 6, seed=21, t_value=0.63)
    mask_full = torch.ones(len(bi), dtype=torch.bool)
    mask_part = mask_full.clone()
    mask_part[[2, 7, 8, 20]] = False
    for name in synth.VARIANTS:
        arrs = dict(num_nodes=nn_.numpy(), xh=xh.numpy(), t=t.numpy(), weight_seed=WEIGHT_SEED, weight_scale=0.7, mask_part=mask_part.numpy())
        for tag, mask in (("full", mask_full), ("part", mask_part)):
            o32, shapes = run(name, torch.float32, xh, t, bi, mask)
            o64, _ = run(name, torch.float64, xh, t, bi, mask)
            arrs[f"out32_{tag}"], arrs[f"out64_{tag}"] = o32.float().numpy(), o64.double().numpy()
            gap = (o32.double() - o64).abs().max().item()
            print(f"{name:18s} {tag}: {len(shapes)} tensors, max|out| = {o64.abs().max().item():.3e}, |ref32 - ref64| = {gap:.2e}", flush=True)
        arrs["keys"] = np.array(list(shapes))
        arrs["shapes"] = np.array([",".join(map(str, s)) for s in shapes.values()])
        np.savez_compressed(os.path.join(HERE, f"dyn_variant_{name}.npz"), **arrs)


if __name__ == "__main__":
    main()
