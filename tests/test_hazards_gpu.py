"""The two code-generation hazards this code base works AROUND rather than root-causes, pinned.

(a) stale MFMA B operands behind inline-asm operand splits (csrc/gcdm_edge_x3.hip.h, x3_settle): found in k_edge_embed_x3 in round 2, when the splits
    ended in 16-bit partial writes (v_fma_mix{lo,hi}_f16; stand-alone reproducer of that precondition: tools/mfma_partial_write_hazard.hip).  Since
    round 3 every split ends in v_cvt_pk_f16_f32, a full 32-bit write.  Round 6 priced the fence (58 825 -> 58 680 cycles per QM9 tile, 57 290 -> 57 115
    GEOM: -0.25 % / -0.3 %, same bits, profiles/r06_hazards.txt) and ships the UN-fenced code; the fenced build (-DGCDM_X3_SETTLE) is the pinned fallback:
    built and run here, it must give the shipped build's bits.
(b) run-to-run differences of the 32-edge kernel when the compiler's SLP vectoriser packs fp32 FMAs (v_pk_fma_f32) whose operands are destinations of
    in-flight per-lane loads (round 1; the library is built with -packed-fp32-ops).

Each test builds the library WITHOUT the workaround (hipcc on the GPU box, into /tmp; ~40 s, both in parallel, cached per session), runs the configuration
that exposed the fault several times in a fresh process, and records what happens (gpurun_out/hazards_<name>.txt + the test's output).  The variant's outcome
is recorded, not asserted -- the faults are timing dependent; what IS asserted is that the shipped build passes the same script, so a compiler upgrade that
re-introduces either fault in the shipped code fails here (and in test_gpu_parity's determinism tests) by construction, not by luck.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
SRC = os.path.join(ROOT, "bio-diffusion_amd", "csrc", "gcdm_api.hip")
BASE = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]
NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
VARIANTS = {
    "with_settle": NOPK + ["-DGCDM_X3_SETTLE"],       # (a): the shipped flags + the x3_settle operand fence of rounds 2-5 (the fallback)
    "slp_packed": [],                                   # (b): the compiler's default -- packed-fp32 feature on, SLP vectoriser on
}

# the probe: hashes of repeated launches (both tile sizes, a ragged GEOM batch whose edge-embedding launch runs a second round of workgroups), and the
# edge embedding of the split-precision mode against the fp32 kernels on the batch that exposed (a)
PROBE = r'''
import hashlib, importlib, json, os, sys
import torch
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import synth
pkg = importlib.import_module("bio-diffusion_amd")
res = {"lib": pkg._native.LIB_PATH, "hashes": {}, "embed_max_diff": []}
def net_for(case, mode, tile):
    d = synth.DATASET_DIMS[case]
    cfgs = pkg.default_cfgs("geom" if case == "geom" else "qm9")
    net = pkg.GCPNetDynamics(**cfgs)
    net.load_state_dict(synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=17, scale_2d=0.5))
    net = net.cuda().eval()
    net._ensure_handle(torch.device("cuda"))
    net.set_mfma_mode(mode)
    net._lib.gcdm_set_option(net._handle, b"edge_tile", tile)
    return net, d
def fwd(net, xh, t, bi):
    dev = torch.device("cuda")
    batch = dict(batch=bi.to(dev), mask=torch.ones(len(bi), dtype=torch.bool, device=dev), props_context=None)
    _, out = net(batch, xh.to(dev), t.to(dev))
    torch.cuda.synchronize()
    return out
for case, sizes in (("geom", [181, 3, 90]), ("qm9", [19] * 256)):
    for tile in (32, 64):
        net, d = net_for(case, 1, tile)
        xh, t, bi, nn_, _ = synth.make_inputs(sizes, synth.dims_feat(d), seed=3, t_value=0.3)
        hs = []
        for rep in range(REPS):
            hs.append(hashlib.sha256(fwd(net, xh, t, bi).cpu().numpy().tobytes()).hexdigest()[:12])
        res["hashes"][f"{case}/tile{tile}"] = hs
        if tile == 64:
            n32, _ = net_for(case, 0, tile)
            fwd(n32, xh, t, bi)
            ref = {k: n32.debug_read(k) for k in ("ep", "alpha")}
            for rep in range(REPS):
                fwd(net, xh, t, bi)
                res["embed_max_diff"].append(max(float((net.debug_read(k) - ref[k]).abs().max() / ref[k].abs().max().clamp(min=1e-30)) for k in ref))
print("PROBE " + json.dumps(res))
'''


@pytest.fixture(scope="session")
def variant_libs():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc is not on this box")
    out = tempfile.mkdtemp(prefix="gcdm_hazard_")
    procs = {n: subprocess.Popen([hipcc] + BASE + fl + ["-o", os.path.join(out, f"libgcdm_{n}.so"), SRC], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for n, fl in VARIANTS.items()}
    libs = {}
    for n, p in procs.items():
        log, _ = p.communicate(timeout=900)
        assert p.returncode == 0, log[-2000:]
        libs[n] = os.path.join(out, f"libgcdm_{n}.so")
    yield libs
    shutil.rmtree(out, ignore_errors=True)


def _probe(lib, reps):
    env = dict(os.environ)
    if lib:
        env["GCDM_HIP_LIB"] = lib
    else:
        env.pop("GCDM_HIP_LIB", None)
    code = PROBE.replace("ROOT", repr(ROOT)).replace("REPS", str(reps))
    p = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("PROBE ")][-1]
    return json.loads(line[6:])


def _verdict(r):
    repro = all(len(set(h)) == 1 for h in r["hashes"].values())
    embed = max(r["embed_max_diff"]) if r["embed_max_diff"] else 0.0
    return repro, embed


def _record(name, text):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"hazards_{name}.txt"), "w") as f:
        f.write(text + "\n")
    print(text)


def test_shipped_build_passes_the_hazard_probe():
    r = _probe(None, 4)
    repro, embed = _verdict(r)
    _record("shipped", f"shipped build {os.path.basename(r['lib'])}: repeated launches identical = {repro}; edge embedding f16x3 vs fp32 kernels, worst of {len(r['embed_max_diff'])} "
                       f"runs = {embed:.2e}; hashes {json.dumps(r['hashes'])}")
    assert repro, r["hashes"]
    assert embed <= 1e-5, r["embed_max_diff"]                        # a stale B column is an O(1) error in 16 edges of a wave


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_variant_build_is_on_record(name, variant_libs):
    r = _probe(variant_libs[name], 6)
    repro, embed = _verdict(r)
    ref = _probe(None, 1)
    same_bits = all(r["hashes"][k][0] == ref["hashes"][k][0] for k in r["hashes"])
    _record(name, f"variant {name} ({' '.join(VARIANTS[name]) or 'compiler defaults: packed-fp32 feature + SLP vectoriser'}): repeated launches identical = {repro}; "
                  f"edge embedding f16x3 vs fp32 kernels, worst of {len(r['embed_max_diff'])} runs = {embed:.2e}; first hash equals the shipped build's = {same_bits}; "
                  f"hashes {json.dumps(r['hashes'])}\n"
                  f"  reading: {'the fault does NOT reproduce with this compiler on this box' if (repro and embed <= 1e-5) else 'FAULT REPRODUCED -- the workaround is load-bearing'}")
    # recorded, not asserted (timing dependent); the probe itself must have run
    assert r["hashes"] and len(r["embed_max_diff"]) == 12
    if name == "with_settle":          # the fenced FALLBACK of the un-fenced shipped build (round 6): same bits, repeatable
        assert repro and same_bits and embed <= 1e-5, (r["hashes"], ref["hashes"])
