"""Long-horizon golden: the REFERENCE's own `mol_gen_sample` (src/models/components/variational_diffusion.py:1282-1412), free-running on a
noise tape for the FULL 1000 steps at full width (QM9 production architecture, 4 molecules n = 5, 19, 3, 11, seed-recreated weights x 0.25),
once in fp32 and once with the same code in fp64 (the adjudicator).  Stored: the latent z after the step to s = 900, 800, ..., 0 and
after selected early steps, and the final decode.  -> tests/golden/long_full_qm9.npz   (contract (iii))

    python tests/golden/make_long_golden.py              (build container only; ~10 min of CPU)
    python tests/golden/make_long_golden.py ragged16     -> tests/golden/long_ragged16_qm9.npz: 16 molecules of 5 ... 27 atoms (~40 min of CPU)
    python tests/golden/make_long_golden.py config0      -> tests/golden/long_config0_qm9.npz: BASELINE.json configs[0], 64 molecules x 19 atoms (hours of CPU)
    python tests/golden/make_long_golden.py geom8        -> tests/golden/long_geom8.npz: 8 GEOM-Drugs-sized molecules of 18 ... 72 atoms, GEOM architecture
    python tests/golden/make_long_golden.py cond6        -> tests/golden/long_cond6_qm9.npz: 6 molecules on the alpha-CONDITIONAL QM9 model (BASELINE.json configs[2]'s architecture), context per molecule stored

Only data is stored (inputs = num_nodes + seeds, outputs); the weights are re-created from `synth.make_weights(..., seed=LONG_WEIGHT_SEED,
scale_2d=0.25)` and the noise from `TapeNoise(LONG_NOISE_SEED)` wherever the fixture is used.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402
import synth  # noqa: E402

torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))

LONG_WEIGHT_SEED = 61
LONG_NOISE_SEED = 4321
SIZES = [5, 19, 3, 11]
CHECKPOINTS = [999, 990, 950] + list(range(900, -1, -100))      # z after the step that lands on s
OUT_NAME = "long_full_qm9.npz"
if len(sys.argv) > 1 and sys.argv[1] == "ragged16":             # second fixture: a ragged batch of 16 QM9-sized molecules (275 atoms, 5 165 edges)
    LONG_WEIGHT_SEED, LONG_NOISE_SEED = 67, 9753
    SIZES = [18, 19, 17, 23, 9, 16, 21, 12, 19, 14, 27, 5, 20, 18, 15, 22]
    CHECKPOINTS = [999, 900, 500, 100, 0]
    OUT_NAME = "long_ragged16_qm9.npz"
DATASET = "qm9"
if len(sys.argv) > 1 and sys.argv[1] == "config0":              # BASELINE.json configs[0]: 64 QM9 molecules x 19 atoms (1 216 atoms, 23 104 edges)
    LONG_WEIGHT_SEED, LONG_NOISE_SEED = 47, 77
    SIZES = [19] * 64
    CHECKPOINTS = [999] + list(range(900, -1, -100))
    OUT_NAME = "long_config0_qm9.npz"
if len(sys.argv) > 1 and sys.argv[1] == "geom8":                # 8 GEOM-Drugs-sized molecules (342 atoms, 17 034 edges, rows of up to 72 edges), GEOM architecture
    DATASET = "geom"
    LONG_WEIGHT_SEED, LONG_NOISE_SEED = 71, 2468
    SIZES = [44, 31, 58, 18, 72, 40, 27, 52]
    CHECKPOINTS = [999, 900, 700, 500, 300, 100, 0]
    OUT_NAME = "long_geom8.npz"
COND, CASE, CONTEXT_SEED = (), None, None
if len(sys.argv) > 1 and sys.argv[1] == "cond6":                # the property-conditional model (configs[2]): context [B, 1] ~ N(0, 1) per molecule, broadcast to the atoms
    COND, CASE, CONTEXT_SEED = ("alpha",), "qm9cond", 2
    LONG_WEIGHT_SEED, LONG_NOISE_SEED = 83, 1357
    SIZES = [9, 19, 4, 23, 14, 17]
    CHECKPOINTS = [999, 900, 700, 500, 300, 100, 0]
    OUT_NAME = "long_cond6_qm9.npz"


def run(dtype):
    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)          # the reference hard-wires the default dtype in localize / scalarize
    try:
        cfgs = rh.load_reference_cfgs(DATASET, COND)
        d = synth.DATASET_DIMS[CASE or DATASET]
        net = rh.build_reference_dynamics(cfgs, seed=0)
        shapes = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
        net.load_state_dict(synth.make_weights(shapes, seed=LONG_WEIGHT_SEED, scale_2d=0.25))
        net = net.to(dtype)
        ddpm = rh.build_reference_ddpm(cfgs, net, DATASET).to(dtype)
        nn_ = torch.tensor(SIZES)
        zs = {}
        orig = ddpm.sample_p_zs_given_zt

        def spy(*a, **kw):                  # observes the reference's own step (no change of behaviour)
            out = orig(*a, **kw)
            s = int(round(float(kw["s"].flatten()[0]) * 1000))
            if s in CHECKPOINTS:
                zs[s] = out.detach().clone()
            return out

        ddpm.sample_p_zs_given_zt = spy
        t0 = time.time()
        ctx = None
        if COND:
            # [B, 1] per molecule: mol_gen_sample expands it itself, context[batch_index] (variational_diffusion.py:1317-1320)
            ctx = torch.randn((len(SIZES), 1), generator=torch.Generator().manual_seed(CONTEXT_SEED), dtype=torch.float32).to(dtype)
        with rh.NoiseTape(LONG_NOISE_SEED) as tape, torch.no_grad():
            xh, bi, _ = ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cpu", context=ctx)
        print(f"{dtype}: {time.time() - t0:.0f} s, {len(tape.calls)} randn calls, max|x| = {xh[:, :3].abs().max().item():.3e}", flush=True)
        return xh, zs
    finally:
        torch.set_default_dtype(prev)


def main():
    assert rh.reference_available(), "reference checkout not found"
    x32, z32 = run(torch.float32)
    x64, z64 = run(torch.float64)
    out = dict(dataset=CASE or DATASET, num_nodes=np.array(SIZES), weight_seed=LONG_WEIGHT_SEED, weight_scale=0.25, noise_seed=LONG_NOISE_SEED, T=1000,
               checkpoints=np.array(CHECKPOINTS), final32=x32.float().numpy(), final64=x64.double().numpy())
    if COND:
        out["context"] = torch.randn((len(SIZES), 1), generator=torch.Generator().manual_seed(CONTEXT_SEED), dtype=torch.float32).numpy()
    for s in CHECKPOINTS:
        out[f"z32_{s}"] = z32[s].float().numpy()
        out[f"z64_{s}"] = z64[s].double().numpy()
        gap = (z32[s].double() - z64[s]).abs().max().item()
        print(f"s={s:4d}  max|z| = {z64[s].abs().max().item():.4e}   |ref32 - ref64| = {gap:.3e}")
    print("final: |ref32 - ref64| x =", (x32[:, :3].double() - x64[:, :3]).abs().max().item(), " discrete equal:", bool((x32[:, 3:].double() == x64[:, 3:]).all()))
    path = os.path.join(HERE, OUT_NAME)
    np.savez_compressed(path, **out)
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
