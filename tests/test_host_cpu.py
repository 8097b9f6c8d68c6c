"""CPU-only checks: module/state-dict parity with the reference layout, config surface, C-ABI exports,
and a NumPy emulation of the MFMA operand mapping that the weight packing relies on."""
import ctypes
import importlib
import os
import re

import numpy as np
import pytest
import torch

import synth

pkg = importlib.import_module("bio-diffusion_amd")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("case,ds,cond", [("qm9", "qm9", ()), ("qm9cond", "qm9", ("alpha",)), ("geom", "geom", ())])
def test_state_dict_matches_reference_layout(case, ds, cond):
    cfgs = pkg.default_cfgs(ds, cond)
    net = pkg.GCPNetDynamics(**cfgs)
    d = synth.DATASET_DIMS[case]
    want = synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d))
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want
    assert list(got) == list(want)          # registration order too
    n_params = sum(int(np.prod(s)) for s in want.values())
    assert n_params == {"qm9": 6213433, "qm9cond": 6213433, "geom": 2727387}[case]
    net.load_state_dict(synth.make_weights(want, seed=1))


@pytest.mark.needs_reference
@pytest.mark.parametrize("ds,cond", [("qm9", ()), ("qm9", ("alpha",)), ("geom", ())])
def test_state_dict_matches_imported_reference(ds, cond):
    import ref_harness as rh
    ref = rh.build_reference_dynamics(rh.load_reference_cfgs(ds, cond))
    net = pkg.GCPNetDynamics(**pkg.default_cfgs(ds, cond))
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert a == b and list(a) == list(b)
    # config values: our defaults == the reference YAML tree
    refc = rh.load_reference_cfgs(ds, cond)
    ours = pkg.default_cfgs(ds, cond)
    tree = pkg.load_cfg_tree(os.path.join(rh.REFERENCE_ROOT, "configs"), ds, cond)
    for grp in ("model_cfg", "layer_cfg", "diffusion_cfg"):
        for k, v in refc[grp].items():
            if k in ours[grp] and not isinstance(v, dict):
                assert ours[grp][k] == v, (grp, k)
                assert tree[grp][k] == v, (grp, k)
    for k in ("num_atom_types", "include_charges", "num_x_dims"):
        assert ours["dataloader_cfg"][k] == refc["dataloader_cfg"][k]
    for k in ("bottleneck", "vector_gate", "frame_gate", "norm_x_diff", "node_positions_weight"):
        assert ours["module_cfg"][k] == refc["module_cfg"][k]
    info = pkg.dataset_info("geom" if ds == "geom" else ("qm9_second_half" if cond else "qm9"))
    rinfo = rh.dataset_info(ds, cond)
    assert info["n_nodes"] == {int(k): int(v) for k, v in rinfo["n_nodes"].items()}
    assert info["atom_decoder"] == rinfo["atom_decoder"] and info["max_n_nodes"] == rinfo["max_n_nodes"]


def test_predefined_noise_schedules_equal_the_reference_tables(golden_dir):
    """cosine, polynomial_2 (production), polynomial_3 at T = 1000 / 250: the gamma lookup tables are BIT-equal to the ones the reference's
    PredefinedNoiseSchedule built (tests/golden/gamma_tables.npz, make_schedule_golden.py); anything else raises like there."""
    g = np.load(os.path.join(golden_dir, "gamma_tables.npz"))
    for key in g.files:
        name, T = key.split("__")
        assert np.array_equal(pkg.PredefinedNoiseSchedule(name, int(T), 1e-5).gamma.numpy(), g[key]), key
    with pytest.raises(ValueError):
        pkg.PredefinedNoiseSchedule("linear", 1000, 1e-5)


def test_forward_refuses_cpu():
    net = pkg.GCPNetDynamics(**pkg.default_cfgs("qm9"))
    bi = torch.zeros(3, dtype=torch.long)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(dict(batch=bi, mask=torch.ones(3, dtype=torch.bool)), torch.zeros(3, 9), torch.zeros(3, 1))


def test_gamma_and_ddpm_host_logic():
    from oracle import gcdm_oracle as O
    cfgs = pkg.default_cfgs("qm9")
    ddpm = pkg.EquivariantVariationalDiffusion(pkg.GCPNetDynamics(**cfgs), cfgs["diffusion_cfg"], cfgs["dataloader_cfg"], pkg.dataset_info("qm9"))
    assert torch.equal(ddpm.gamma.gamma, O.gamma_table(O.OracleConfig()))
    n = ddpm.num_nodes_distribution.sample(64)
    assert n.min() >= 3 and n.max() <= 29
    keys = set(ddpm.state_dict())
    assert {"gamma.gamma", "num_nodes_distribution.num_nodes", "num_nodes_distribution.prob"} <= keys
    g = torch.Generator().manual_seed(0)
    bi = torch.repeat_interleave(torch.arange(3), torch.tensor([4, 5, 3]))
    z = ddpm.sample_combined_position_feature_noise(bi, torch.ones(12, dtype=torch.bool), generator=g)
    for b in range(3):
        assert z[bi == b, :3].sum(0).abs().max() < 1e-5


def test_whole_model_plug_point_and_checkpoint_roundtrip(tmp_path):
    cfgs = pkg.default_cfgs("geom")
    model = pkg.GEOMMoleculeGenerationDDPM(optimizer=None, scheduler=None, **cfgs)
    assert model.dataset_info["max_n_nodes"] == 181
    sd = model.state_dict()
    assert any(k.startswith("ddpm.dynamics_network.interaction_layers.3.") for k in sd)
    path = os.path.join(tmp_path, "m.ckpt")
    torch.save({"state_dict": sd, "hyper_parameters": {"x": 1}}, path)
    m2 = model.load_from_checkpoint(checkpoint_path=path, map_location="cpu", **cfgs)
    for k, v in m2.state_dict().items():
        assert torch.equal(v, sd[k])


def test_cabi_exports_every_declared_symbol():
    native = pkg._native
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("libgcdm_hip.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(native.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "gcdm_hip.h")).read()
    declared = set(re.findall(r"\b(gcdm_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(native.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_ops_cabi_exports_every_declared_symbol():
    """include/gcdm_ops.h (module-level operators) <-> libgcdm_ops.so <-> the ctypes signatures."""
    native = pkg._native
    if not os.path.exists(native.OPS_LIB_PATH):
        pytest.skip("libgcdm_ops.so not built (run __graft_entry__.build())")
    lib = ctypes.CDLL(native.OPS_LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "gcdm_ops.h")).read()
    hdr = hdr[hdr.index("#ifndef GCDM_OPS_H"):]
    declared = set(re.findall(r"\b(gcdm_op_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(native.OPS_EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
        n_args = hdr[hdr.index(name + "("):].split(")")[0].count(",") + 1
        assert n_args == len(native.OPS_SIGNATURES[name]), name


def test_mfma_operand_mapping_of_the_packing():
    """Emulates v_mfma_f32_32x32x2_f32 lane semantics (cdna guide section 3) on the host-side packing formula:
    D[i][j] = sum_k A[i][k] B[k][j], A: lane l holds A[l&31][l>>5], B: lane l holds B[l>>5][l&31],
    C/D: lane l reg r holds D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]."""
    rng = np.random.default_rng(0)
    M, K, T = 64, 24, 32
    W = rng.standard_normal((M, K)).astype(np.float64)
    X = rng.standard_normal((K, T)).astype(np.float64)
    MT, G = M // 32, K // 8
    packed = np.zeros((MT, G, 64, 4))
    for mt in range(MT):
        for g in range(G):
            for lane in range(64):
                for t in range(4):
                    packed[mt, g, lane, t] = W[32 * mt + (lane & 31), 8 * g + 4 * (lane >> 5) + t]
    xs4 = X.reshape(K // 4, 4, T).transpose(0, 2, 1)          # XS4[group][entity][4]
    out = np.zeros((M, T))
    for mt in range(MT):
        acc = np.zeros((64, 16))
        for g in range(G):
            for t in range(4):
                A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
                for lane in range(64):
                    A[lane & 31, lane >> 5] = packed[mt, g, lane, t]
                    Bm[lane >> 5, lane & 31] = xs4[2 * g + (lane >> 5), lane & 31, t]
                Dm = A @ Bm
                for lane in range(64):
                    for r in range(16):
                        acc[lane, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31]
        for lane in range(64):
            for r in range(16):
                out[32 * mt + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), lane & 31] = acc[lane, r]
    assert np.allclose(out, W @ X)
    # accumulator registers 4q..4q+3 of a lane are exactly channel group 8 mt + 2 q + half
    for lane in (0, 31, 32, 63):
        for q in range(4):
            ch = [(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) for r in range(4 * q, 4 * q + 4)]
            assert ch == list(range(4 * (2 * q + (lane >> 5)), 4 * (2 * q + (lane >> 5)) + 4))


def test_xyz_writers_match_reference_files(tmp_path):
    """save_xyz_file / write_xyz_file produce the reference's files byte for byte (tests/golden/xyz.npz, written by the reference)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "xyz.npz"))
    nn_ = torch.tensor(g["num_nodes"])
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
    d = str(tmp_path) + "/out/"
    pkg.save_xyz_file(d, torch.tensor(g["pos"]), torch.tensor(g["one_hot"]), torch.zeros(0), pkg.dataset_info("qm9"), id_from=7, name="mol",
                      batch_index=bi)
    assert sorted(os.listdir(d)) == list(g["names"])
    for nme, txt in zip(g["names"], g["texts"]):
        assert open(os.path.join(d, str(nme))).read() == str(txt)
    types = torch.tensor(g["one_hot"]).argmax(-1)
    pkg.write_xyz_file(torch.tensor(g["pos"])[:3], types[:3], str(tmp_path) + "/single.xyz")
    assert open(str(tmp_path) + "/single.xyz").read() == str(g["single"])


def test_bench_flop_count_matches_oracle_closed_form():
    """bench.py carries its own copy of the SURVEY A.4 closed form (the product side must not import oracle/ outside the cpu_baseline leg)."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle import gcdm_oracle as O
    for N, E, dims in ((19456, 369664, (256, 32, 64, 16, 9, 7)), (11264, 495616, (256, 32, 16, 8, 4, 17)), (1216, 23104, (256, 32, 64, 16, 9, 7))):
        total, edge, _node = bench.algorithmic_flops(N, E, dims)
        S, V, Se, Ve, L, h_in = dims
        assert total == O.forward_flops(N, E, S, V, Se, Ve, L, h_in)
        assert edge == O._gcp2_flops(E, 2 * S + Se, 2 * V + Ve, S, V, 4) + 3 * O._gcp2_flops(E, S, V, S, V, 4) + 2 * E * S
    assert abs(bench.algorithmic_flops(19456, 369664, (256, 32, 64, 16, 9, 7))[0] / 2836.5e9 - 1) < 1e-3      # C2 = 2 836.5 GFLOP


def test_slice_cuts_are_contiguous_nonempty_and_balanced():
    from importlib import import_module
    vd = import_module("bio-diffusion_amd.variational_diffusion")
    g = torch.Generator().manual_seed(0)
    for B, K in ((2, 2), (3, 3), (5, 2), (40, 2), (40, 3), (1024, 2), (257, 4)):
        for trial in range(3):
            nn_ = torch.randint(1, 60, (B,), generator=g) if trial else torch.full((B,), 19)
            cuts = vd.slice_cuts(nn_, K)
            assert cuts[0] == 0 and cuts[-1] == B and len(cuts) == K + 1
            assert all(b > a for a, b in zip(cuts[:-1], cuts[1:]))
            if B >= 40:
                w = (nn_.long() ** 2)
                parts = [int(w[a:b].sum()) for a, b in zip(cuts[:-1], cuts[1:])]
                assert max(parts) <= 1.35 * (sum(parts) / K) + int(w.max())
    with pytest.raises(ValueError):
        vd.slice_cuts(torch.tensor([5]), 2)


def test_repaint_schedule_matches_reference_and_covers_every_step(golden_dir):
    """Product-side `repaint_schedule` (closed form) vs the reference's outputs (tests/golden/repaint_schedule.json, 192 triples), plus the
    invariant the loop relies on: walking the schedule with jumps of `jump_length` ends exactly at s = -1."""
    import json
    vd = importlib.import_module("bio-diffusion_amd.variational_diffusion")
    cases = json.load(open(os.path.join(golden_dir, "repaint_schedule.json")))
    for c in cases:
        r, j, T = c["resamplings"], c["jump_length"], c["num_timesteps"]
        sched = vd.repaint_schedule(r, j, T)
        assert sched == c["schedule"], c
        s = T - 1
        for i, n in enumerate(sched):
            s -= n
            assert s >= -1
            if i < len(sched) - 1:
                s += j
                assert s <= T - 1
        assert s == -1
    assert vd.repaint_schedule(2, 5, 0) == [] and vd.repaint_schedule(0, 2, 7) == [1]
    with pytest.raises(ValueError):
        vd.repaint_schedule(2, 0, 10)


def test_sample_sweep_conditionally_builds_the_reference_context():
    """src/models/__init__.py:200-226: linspace of the normalised property range for the given size, fix_noise=True, one size for all frames."""
    pkg = importlib.import_module("bio-diffusion_amd")

    class Props:
        distributions = {"alpha": {19: {"params": (torch.tensor(40.0), torch.tensor(100.0))}}, "gap": {19: {"params": (torch.tensor(0.1), torch.tensor(0.5))}}}
        normalizer = {"alpha": {"mean": torch.tensor(70.0), "mad": torch.tensor(10.0)}, "gap": {"mean": torch.tensor(0.3), "mad": torch.tensor(0.1)}}

    class Model:
        device = torch.device("cpu")

        def sample(self, **kw):
            self.kw = kw
            return "x", "one_hot", "charges", "batch_index"

    m = Model()
    out = pkg.sample_sweep_conditionally(m, Props(), num_nodes=19, num_frames=5)
    assert out == ("x", "one_hot", "charges", "batch_index")
    assert m.kw["fix_noise"] is True and m.kw["num_samples"] == 5 and m.kw["num_nodes"].tolist() == [19] * 5
    ctx = m.kw["context"]
    assert ctx.dtype == torch.float32 and ctx.shape == (5, 2)
    assert torch.allclose(ctx[:, 0], torch.linspace(-3.0, 3.0, 5)) and torch.allclose(ctx[:, 1], torch.linspace(-2.0, 2.0, 5), atol=1e-6)


def test_header_is_plain_c_and_matches_the_ctypes_binding(tmp_path):
    """include/gcdm_hip.h must be usable from C (the drop-in boundary is a C ABI): compile it with gcc -std=c99 -pedantic and compare the struct
    layouts the C compiler sees with those of the ctypes binding (bio-diffusion_amd/_native.py)."""
    import subprocess
    native = importlib.import_module("bio-diffusion_amd._native")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    fields_cfg = [f[0] for f in native.GcdmConfig._fields_]
    fields_tab = [f[0] for f in native.GcdmBondTables._fields_]
    lines = ['#include "gcdm_hip.h"', "#include <stddef.h>", "#include <stdio.h>", "int main(void) {",
             '  printf("%zu %zu %d\\n", sizeof(GcdmConfig), sizeof(GcdmBondTables), GCDM_ABI_VERSION);']
    lines += [f'  printf("%zu\\n", offsetof(GcdmConfig, {f}));' for f in fields_cfg]
    lines += [f'  printf("%zu\\n", offsetof(GcdmBondTables, {f}));' for f in fields_tab]
    lines += ["  return 0;", "}"]
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True).stdout.split("\n")
    size_cfg, size_tab, abi = (int(v) for v in out[0].split())
    assert (size_cfg, size_tab, abi) == (ctypes.sizeof(native.GcdmConfig), ctypes.sizeof(native.GcdmBondTables), native.ABI_VERSION)
    offs = [int(v) for v in out[1:1 + len(fields_cfg) + len(fields_tab)]]
    want = [getattr(native.GcdmConfig, f).offset for f in fields_cfg] + [getattr(native.GcdmBondTables, f).offset for f in fields_tab]
    assert offs == want


def test_library_links_from_c_without_torch(tmp_path):
    """A C program (no Python, no torch) links libgcdm_hip.so through the header and calls entry points that need no GPU: the null-handle
    error paths.  This is what a cgo / JNI / plain-C host would do."""
    import subprocess
    native = importlib.import_module("bio-diffusion_amd._native")
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("library not built")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "link.c"
    src.write_text("""
#include "gcdm_hip.h"
#include <stdio.h>
#include <string.h>
int main(void) {
    GcdmConfig cfg; gcdm_handle* h = NULL;
    memset(&cfg, 0, sizeof cfg);
    if (strcmp(gcdm_last_error(NULL), "null handle") != 0) return 1;
    if (gcdm_num_nodes(NULL) != -1 || gcdm_num_edges(NULL) != -1) return 2;
    if (gcdm_create(NULL, &h) >= 0) return 3;
    if (gcdm_create(&cfg, NULL) >= 0) return 4;
    if (gcdm_destroy(NULL) != 0) return 5;
    if (gcdm_get_option(NULL, "mfma_mode") != -1) return 6;
    puts("ok");
    return 0;
}
""")
    exe = tmp_path / "link"
    libdir = os.path.dirname(native.LIB_PATH)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L" + libdir, "-lgcdm_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and run.stdout.strip() == "ok", (run.returncode, run.stdout, run.stderr)


def test_timestep_index_rounds_like_torch():
    """The gamma-table index of every sampler entry point (gcdm_timestep_index, pure host code) against the reference's formula
    `torch.round(t * T).long()` (variational_diffusion.py:252-255) for t = s / n in fp32 -- including the step counts whose products are exact
    ties (n = 16, 80, 400, 2000 on the T = 1000 table), which round to EVEN in torch."""
    native = importlib.import_module("bio-diffusion_amd._native")
    if not os.path.exists(native.LIB_PATH):
        pytest.skip("library not built")
    lib = native.load()
    ties = 0
    for T in (1000, 500, 8):
        for n in (1, 2, 5, 6, 12, 16, 37, 80, 125, 400, 1000, 2000):
            s = torch.arange(0, n + 1, dtype=torch.float32)
            t = s / n
            want = torch.round(t * T).long().clamp(0, T).tolist()
            got = [lib.gcdm_timestep_index(float(v), T) for v in t.tolist()]
            assert got == want, (T, n)
            prod = (t * T)
            ties += int(((prod - torch.floor(prod)) == 0.5).sum())
    assert ties > 100                                         # the tie cases really occur
    assert lib.gcdm_timestep_index(-0.3, 1000) == 0 and lib.gcdm_timestep_index(1.7, 1000) == 1000


def test_checkpoint_unpickler_resolves_names_not_modules(tmp_path):
    """A crafted .ckpt must not reach code-executing globals (builtins.eval / exec / getattr, os.system, torch callables, ...): the loader
    whitelists (module, name) pairs, everything else unpickles to an inert placeholder; a genuine state-dict still loads."""
    import pickle
    mg = importlib.import_module("bio-diffusion_amd.mol_gen_ddpm")
    marker = tmp_path / "pwned"

    class Evil:
        def __reduce__(self):
            return (eval, (f"open({str(marker)!r}, 'w').write('x')",))

    class Evil2:
        def __reduce__(self):
            import os as _os
            return (_os.system, (f"touch {marker}",))

    for payload in (Evil(), Evil2()):
        p = tmp_path / "evil.ckpt"
        with open(p, "wb") as f:
            pickle.dump({"state_dict": {"w": payload}}, f)
        with open(p, "rb") as f:
            obj = mg._StateDictUnpickler(f).load()
        assert not marker.exists()
        assert type(obj["state_dict"]["w"]).__name__ == "_Dummy"
    for mod, name in (("builtins", "eval"), ("builtins", "exec"), ("builtins", "getattr"), ("builtins", "__import__"), ("os", "system"),
                      ("torch", "load"), ("torch.serialization", "load"), ("subprocess", "Popen"), ("numpy", "fromfile")):
        assert mg._StateDictUnpickler(open(os.devnull, "rb")).find_class(mod, name) is mg._Dummy, (mod, name)
    # nested pickle: torch.storage._load_from_bytes is torch.load(BytesIO(b), weights_only=False) with the DEFAULT unpickler -- REDUCEd onto a
    # bytes payload it would run an unrestricted inner pickle.  It is not on the whitelist (zip-format checkpoints never need it).
    import io
    inner = io.BytesIO()
    pickle.dump(Evil(), inner)

    class Nested:
        def __reduce__(self):
            return (torch.storage._load_from_bytes, (inner.getvalue(),))

    for how in ("unpickler", "loader"):
        p = tmp_path / "nested.ckpt"
        with open(p, "wb") as f:
            pickle.dump({"state_dict": {"w": Nested()}}, f)
        if how == "unpickler":
            with open(p, "rb") as f:
                obj = mg._StateDictUnpickler(f).load()
        else:
            try:                       # (a bare pickle is not a torch file: torch.load may refuse it before any unpickling -- also fine)
                obj = {"state_dict": mg.load_lightning_state_dict(str(p))}
            except RuntimeError:
                obj = None
        assert not marker.exists(), how
        assert obj is None or type(obj["state_dict"]["w"]).__name__ == "_Dummy"
    assert mg._StateDictUnpickler(open(os.devnull, "rb")).find_class("torch.storage", "_load_from_bytes") is mg._Dummy
    good = tmp_path / "good.ckpt"
    sd = {"ddpm.x": torch.arange(6, dtype=torch.float32).view(2, 3), "ddpm.y": torch.nn.Parameter(torch.ones(3)), "n": torch.tensor(3)}
    torch.save({"state_dict": sd, "hyper_parameters": {"a": 1}}, good)
    got = mg.load_lightning_state_dict(str(good))
    assert torch.equal(got["ddpm.x"], sd["ddpm.x"]) and torch.equal(got["ddpm.y"].data, sd["ddpm.y"].data) and int(got["n"]) == 3


def test_config_tree_loads_gcp_v1(tmp_path):
    """module_cfg.selected_GCP = GCP (the first-generation module, gcpnet.py:33-262) loads and builds the module-path network (it used to be
    refused); anything else still raises."""
    import shutil
    import yaml
    src = "/root/reference/configs"
    if not os.path.isdir(src):
        pytest.skip("reference configs not present")
    dst = tmp_path / "configs"
    shutil.copytree(src, dst)
    path = dst / "model" / "module_cfg" / "qm9_mol_gen_ddpm_gcp_module.yaml"
    d = yaml.safe_load(open(path))
    assert d["selected_GCP"]["_target_"].endswith(".GCP2")
    assert pkg.GCPNetDynamics(**pkg.load_cfg_tree(str(dst), "qm9", ())).fused_unsupported is None       # the production tree: fused kernels
    d["selected_GCP"]["_target_"] = "src.models.components.gcpnet.GCP"
    yaml.safe_dump(d, open(path, "w"))
    net = pkg.GCPNetDynamics(**pkg.load_cfg_tree(str(dst), "qm9", ()))
    assert isinstance(net.gcp_embedding.edge_embedding, pkg.GCP) and "selected_GCP" in net.fused_unsupported
    d["selected_GCP"]["_target_"] = "src.models.components.gcpnet.GVP"
    yaml.safe_dump(d, open(path, "w"))
    with pytest.raises(NotImplementedError, match="selected_GCP"):
        pkg.load_cfg_tree(str(dst), "qm9", ())


@pytest.mark.parametrize("name", list(synth.VARIANTS))
def test_variant_state_dict_layout_equals_the_reference(name, golden_dir):
    """Keys, shapes and registration order of every non-production variant equal what the REFERENCE's GCPNetDynamics registered when the
    fixture was made (tests/golden/dyn_variant_<name>.npz stores them), so its checkpoints load."""
    g = np.load(os.path.join(golden_dir, f"dyn_variant_{name}.npz"))
    net = pkg.GCPNetDynamics(**synth.apply_variant(pkg.default_cfgs("qm9"), name))
    want = {k: tuple(int(x) for x in s.split(",")) if s else () for k, s in zip(g["keys"].tolist(), g["shapes"].tolist())}
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    assert net.fused_unsupported is not None


def test_position_only_network_has_the_reference_layout_and_feature_networks_refuse_x_only(golden_dir):
    """A dynamics network built without node features (num_atom_types = 0, include_charges = False: what `generate_x_only` needs) registers the
    reference's keys and shapes (fixture: sampler_xonly_qm9.npz) and runs on the module path; a network WITH features refuses the [N, 3] latent."""
    g = np.load(os.path.join(golden_dir, "sampler_xonly_qm9.npz"))
    cfgs = pkg.default_cfgs("qm9")
    synth.apply_variant(cfgs, None)
    cfgs["dataloader_cfg"]["num_atom_types"] = 0
    cfgs["dataloader_cfg"]["include_charges"] = False
    net = pkg.GCPNetDynamics(**cfgs)
    want = {k: tuple(int(x) for x in sh.split(",")) if sh else () for k, sh in zip(g["keys"].tolist(), g["shapes"].tolist())}
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == want and list(got) == list(want)
    full = pkg.default_cfgs("qm9")
    net_full = pkg.GCPNetDynamics(**full)
    assert "no node features" in pkg.GCPNetDynamics(**{**full, "dataloader_cfg": {**full["dataloader_cfg"], "num_atom_types": 0, "include_charges": False}}).fused_unsupported
    ddpm = pkg.EquivariantVariationalDiffusion(net_full, full["diffusion_cfg"], full["dataloader_cfg"], pkg.dataset_info("qm9"))
    with pytest.raises(ValueError, match="without node features"):
        ddpm.mol_gen_sample(num_samples=2, num_nodes=torch.tensor([3, 4]), device="cpu", num_timesteps=2, generate_x_only=True)


def test_embedding_with_an_atom_type_table_has_the_reference_layout(golden_dir):
    """`GCPEmbedding(num_atom_types > 0)` registers the reference's keys and shapes in its order (fixture: the reference's own state dict)."""
    g = np.load(os.path.join(golden_dir, "fn_atom_embedding.npz"))
    want = {k[2:]: tuple(g[k].shape) for k in g.files if k.startswith("w_")}
    T = want["atom_embedding.weight"][0]
    mod = pkg.GCPEmbedding((1, 1), (T, 2), (8, 4), (16, 4), num_atom_types=T, cfg=pkg.default_cfgs("qm9")["module_cfg"], pre_norm=False, use_gcp_norm=True)
    got = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
    assert got == want and list(got) == list(want)


@pytest.mark.needs_reference
@pytest.mark.parametrize("name", ["gcp1_frame_gate", "three_ff", "gcp_norm"])
def test_variant_state_dict_matches_imported_reference(name):
    import ref_harness as rh
    gcp, _, _ = rh.import_reference()
    ref = rh.build_reference_dynamics(synth.apply_variant(rh.load_reference_cfgs("qm9", ()), name, gcp_classes={"GCP": gcp.GCP, "GCP2": gcp.GCP2}))
    net = pkg.GCPNetDynamics(**synth.apply_variant(pkg.default_cfgs("qm9"), name))
    a = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert a == b and list(a) == list(b)


def test_training_forward_needs_the_gpu():
    """Training mode routes the network through the module path (HIP operators with autograd); like every other path it has no CPU fallback."""
    cfgs = pkg.default_cfgs("qm9")
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    batch = pkg.config.AttrDict(x=torch.zeros(3, 3), one_hot=torch.zeros(3, 5), charges=torch.zeros(3), batch=torch.zeros(3, dtype=torch.long),
                                mask=torch.ones(3, dtype=torch.bool))
    model.train()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(batch)
    model.eval()
    with pytest.raises(RuntimeError, match="training_step needs"):
        model.training_step(batch)


@pytest.mark.parametrize("case", ["qm9", "qm9cond", "geom"])
def test_likelihood_algebra_of_the_mirror_matches_reference_golden(case):
    """The host-side algebra of EquivariantVariationalDiffusion.forward and of the module's forward (everything but the two network
    evaluations, which the CPU oracle stands in for HERE ONLY) against the terms the reference returned (tests/golden/nll_full_*.npz)."""
    from oracle import gcdm_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", f"nll_full_{case}.npz"))
    d = synth.DATASET_DIMS[case]
    P = synth.make_weights(synth.dynamics_shapes(d["S"], d["V"], d["Se"], d["Ve"], d["L"], synth.dims_h_in(d)), seed=int(g["weight_seed"]))
    ocfg = O.OracleConfig(num_atom_types=d["num_atom_types"], include_charges=d["include_charges"], num_context=d["n_ctx"], num_layers=d["L"],
                          norm_values=d["norm_values"])

    class OracleNet(torch.nn.Module):
        def forward(self, batch, xh, t, x_self_cond=None, xh_self_cond=None):
            return None, O.dynamics_forward(P, ocfg, xh, t, batch.batch, None, batch.props_context)

    ds = "geom" if case == "geom" else "qm9"
    cfgs = pkg.default_cfgs(ds, ("alpha",) if d["n_ctx"] else ())
    cls = pkg.GEOMMoleculeGenerationDDPM if case == "geom" else pkg.QM9MoleculeGenerationDDPM
    model = cls(**cfgs).eval()
    model.ddpm.dynamics_network = OracleNet()
    nn_ = torch.tensor(g["num_nodes"])
    bi = O.num_nodes_to_batch_index(nn_)
    N, F = int(nn_.sum()), synth.dims_feat(d)
    tape = O.TapeNoise(int(g["noise_seed"]))
    noise = [torch.cat((tape(N, 3), tape(N, F)), dim=-1) for _ in range(2)]
    ctx = torch.tensor(g["ctx"])[bi] if "ctx" in g.files else None
    batch = pkg.config.AttrDict(x=torch.tensor(g["x"]), one_hot=torch.tensor(g["one_hot"]), charges=torch.tensor(g["charges"]), batch=bi,
                                mask=torch.ones(N, dtype=torch.bool), props_context=ctx)
    nll, info = model(batch, t_int=torch.tensor(g["t_int"]).view(-1, 1), noise=noise)
    names = ("delta_log_px", "error_t", "SNR_weight", "loss_0_x", "loss_0_h", "neg_log_constants", "kl_prior", "log_pN")
    want = O.nll_from_terms({k: torch.tensor(g[f"{k}_32"]).double() for k in names}, int(cfgs["diffusion_cfg"]["num_timesteps"]))
    assert (nll.double() - want).abs().max().item() <= 5e-5 * want.abs().max().item()
    for k in ("kl_prior", "delta_log_px", "log_pN", "SNR_weight"):
        assert abs(info[k].item() - float(torch.tensor(g[f"{k}_32"]).double().mean())) <= 1e-5 * max(1.0, abs(float(torch.tensor(g[f"{k}_32"]).double().mean())))
    for k in ("eps_hat_x", "eps_hat_h"):
        assert abs(info[k].item() - float(g[f"{k}_32"])) <= 1e-5


def test_source_fingerprint_ignores_comments_only():
    """bench.csrc_sha16 (the staleness stamp of the committed PMC summaries) is blind to comments and blank lines and to nothing else."""
    import bench
    a = 'int f(int x) { return x + 1; }   // add one\n/* block\n comment */\nconst char* s = "// not a comment";\n\n'
    b = 'int f(int x) { return x + 1; }\nconst char* s = "// not a comment";\n'
    c = 'int f(int x) { return x + 2; }\nconst char* s = "// not a comment";\n'
    assert bench.strip_cxx_comments(a) == bench.strip_cxx_comments(b) != bench.strip_cxx_comments(c)
    assert '"// not a comment"' in bench.strip_cxx_comments(a)
    assert len(bench.csrc_sha16()) == 16


def test_persistent_tile_schedule_visits_every_tile_once():
    """The contract between gcdm_api.hip (grid size, `wg_stride`) and the persistent loop of k_edge_msg_x3 (gcdm_edge_x3.hip.h): workgroup b works on
    XCD b % 8's contiguous range [start, start + cnt) and takes local tiles (b >> 3), (b >> 3) + stride, ...  Restated here and checked for every
    tile count that can occur around the interesting boundaries: each tile exactly once, never a tile >= G, for both workgroup budgets."""
    def schedule(G, budget):
        wgs = budget // 8 * 8
        if G <= wgs or wgs < 8:
            nwg, stride = G, G                      # one tile per workgroup (the round-2 launch)
        else:
            nwg, stride = wgs, wgs // 8
        seen = [0] * G
        for b in range(nwg):
            xcd, base, rem = b & 7, G >> 3, G & 7
            cnt, start, it = base + (1 if xcd < rem else 0), xcd * base + min(xcd, rem), b >> 3
            while it < cnt:
                seen[start + it] += 1
                it += stride
        return seen
    for budget in (256, 512, 304, 32):              # CUs x workgroups per CU (64- / 32-edge tiles), a non-multiple of 8, a partition of 32 CUs
        for G in list(range(1, 40)) + list(range(250, 270)) + list(range(505, 530)) + [1023, 1024, 1025, 2888, 5776, 5777, 7744, 40001]:
            seen = schedule(G, budget)
            assert all(c == 1 for c in seen), (budget, G, [i for i, c in enumerate(seen) if c != 1][:5])


def test_committed_pmc_summaries_belong_to_the_kernels_in_the_tree():
    """The PMC summaries bench.py takes `roofline.traffic` / `mfma_busy_frac_pmc` from carry the fingerprint of the libgcdm_hip.so sources they
    were collected on; a kernel change without a new measurement round would make the bench line report nulls (`pmc_stale`).  The operator
    library (gcdm_ops.*) is a separate build and outside the fingerprint."""
    import bench
    for workload in ("qm9", "geom"):
        pmc = bench.load_pmc_summary(workload, True)
        assert pmc and pmc["stale"] is False and pmc["hbm_bytes_per_launch"] and pmc["mfma_busy_frac"], workload
    import tempfile, shutil
    src = os.path.join(ROOT, "bio-diffusion_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        for f in os.listdir(src):
            shutil.copy(os.path.join(src, f), tmp)
        base = bench.csrc_sha16(tmp)
        assert base == bench.csrc_sha16()
        with open(os.path.join(tmp, "gcdm_ops.hip.h"), "a") as f:
            f.write("\nint not_a_kernel_of_the_hip_library;\n")
        assert bench.csrc_sha16(tmp) == base
        with open(os.path.join(tmp, "gcdm_edge_x3.hip.h"), "a") as f:
            f.write("\nint a_change;\n")
        assert bench.csrc_sha16(tmp) != base


def test_persistent_edge_kernels_do_not_touch_scratch():
    """Static check (no GPU): no instantiation of the persistent split-precision edge kernel may spill.  One spilled register cost 2.4 % of every tile in
    rounds 3-4: a scratch reload is the youngest load in flight and waits behind every prefetch of the next tile."""
    import shutil
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_census
    fake = ("_Z13k_edge_msg_x3ILi64ELi16ELi64EEv13EdgeMsgX3Args:                                  ; @_Z13k_edge_msg_x3ILi64ELi16ELi64EEv13EdgeMsgX3Args\n"
            "\tscratch_load_dword v1, off, off\n\ts_endpgm\n\t.end_amdhsa_kernel\n")
    assert isa_census.scratch_users(fake) == ["_Z13k_edge_msg_x3ILi64ELi16ELi64EEv13EdgeMsgX3Args"]
    assert isa_census.scratch_users(fake.replace("scratch_load_dword v1, off, off", "v_mov_b32_e32 v1, v2")) == []
    if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("hipcc not available")
    assert isa_census.scratch_users() == []


def test_no_mfma_directly_behind_a_partial_write_of_its_source():
    """Static check of the compiled gfx950 kernels (no GPU): an MFMA issued with no wait state behind a v_fma_mix{lo,hi}_f16 write of one
    of its source registers reads the old register on gfx950 (tools/mfma_partial_write_hazard.hip); the compiler is expected to separate the
    two -- this fails the moment a build comes out without the separation."""
    import shutil
    sys_path = os.path.join(ROOT, "tools")
    import sys
    sys.path.insert(0, sys_path)
    import isa_census
    fake = ("_Z1kv:                                  ; @_Z1kv\n\tv_fma_mixhi_f16 v17, v37, s6, v22 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
            "\tv_mfma_f32_32x32x16_f16 a[16:31], v[12:15], v[16:19], a[16:31]\n\ts_endpgm\n\t.end_amdhsa_kernel\n")
    assert len(isa_census.lint_partial_writes(fake)) == 1
    assert isa_census.lint_partial_writes(fake.replace("\tv_mfma", "\ts_nop 0\n\tv_mfma")) == []
    assert isa_census.lint_partial_writes(fake.replace("v[16:19]", "v[20:23]")) == []
    # the LAST writer counts, wherever it is: an unrelated instruction in between is one wait state, a full rewrite of the register clears it
    assert isa_census.lint_partial_writes(fake.replace("\tv_mfma", "\tv_add_f32_e32 v40, v41, v42\n\tv_mfma")) == []
    assert len(isa_census.lint_partial_writes(fake.replace("\tv_mfma", "\tv_add_f32_e32 v40, v41, v42\n\tv_mfma"), min_states=2)) == 1
    assert isa_census.lint_partial_writes(fake.replace("\tv_mfma", "\ts_nop 1\n\tv_mfma"), min_states=2) == []
    assert isa_census.lint_partial_writes(fake.replace("\tv_mfma", "\tv_mov_b32_e32 v17, v3\n\tv_mfma")) == []
    # rule 2: no split between two MFMAs of k_edge_embed_x3 (an s_nop fence in between makes it a new burst)
    emb = ("_Z15k_edge_embed_x3ILi64ELi16EEv15EdgeEmbedX3Args:                                  ; @_Z15k_edge_embed_x3ILi64ELi16EEv15EdgeEmbedX3Args\n"
           "\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[4:7], 0\n\tv_fma_mixlo_f16 v40, v45, s6, 0 op_sel_hi:[0,0,0]\n"
           "\tv_mfma_f32_32x32x16_f16 a[0:15], v[0:3], v[8:11], a[0:15]\n\ts_endpgm\n\t.end_amdhsa_kernel\n")
    assert len(isa_census.lint_partial_writes(emb)) == 1
    assert isa_census.lint_partial_writes(emb.replace("\tv_fma_mixlo", "\ts_nop 1\n\tv_fma_mixlo")) == []
    assert isa_census.lint_partial_writes(emb.replace("k_edge_embed_x3", "k_other_kernelx")) == []
    if shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None:
        pytest.skip("hipcc not available")
    assert isa_census.lint_partial_writes() == []
