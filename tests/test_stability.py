"""Molecular-stability row: oracle vs the reference's golden outputs (CPU), HIP kernel vs both (GPU, through the C ABI)."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import stability_oracle as so  # noqa: E402

pkg = importlib.import_module("bio-diffusion_amd")
G = np.load(os.path.join(ROOT, "tests", "golden", "stability.npz"))
TABLES = json.load(open(os.path.join(ROOT, "bio-diffusion_amd", "data", "bond_tables.json")))


def molecules(ds):
    sizes = G[f"{ds}_sizes"]
    off = np.r_[0, np.cumsum(sizes)]
    return [(G[f"{ds}_x"][a:b], G[f"{ds}_types"][a:b]) for a, b in zip(off[:-1], off[1:])]


@pytest.mark.parametrize("ds", ["qm9", "geom"])
def test_oracle_matches_reference_golden(ds):
    info = pkg.dataset_info(ds)
    bonds = so.bond_length_arrays(TABLES, info["atom_encoder"])
    np.testing.assert_array_equal(np.stack(bonds), G[f"{ds}_bonds"])                      # get_bond_length_arrays
    res = [so.check_molecular_stability(x, t, info["atom_decoder"], TABLES, bonds) for x, t in molecules(ds)]
    np.testing.assert_array_equal(np.asarray([[int(a), b, c] for a, b, c in res], np.int32), G[f"{ds}_result"])
    assert G[f"{ds}_result"][:, 0].sum() >= 5                                             # both outcomes are exercised
    x, t = molecules(ds)[6]
    a1, a2 = np.meshgrid(t.astype(np.int64), t.astype(np.int64), indexing="xy")
    d = so.pair_distances(x).reshape(-1)
    np.testing.assert_array_equal(so.bond_order_batch(a1.reshape(-1), a2.reshape(-1), d, bonds, TABLES["margins"]), G[f"{ds}_order_mol6"])
    np.testing.assert_array_equal(so.bond_order_batch(a1.reshape(-1), a2.reshape(-1), d, bonds, TABLES["margins"], True), G[f"{ds}_order1_mol6"])
    assert set(np.unique(G[f"{ds}_order_mol6"])) >= {0, 1, 2}
    kl = so.kl_divergence(info["atom_types"], len(info["atom_decoder"]), G[f"{ds}_types"])
    assert abs(kl - float(G[f"{ds}_kl"])) < 1e-12


@pytest.mark.parametrize("ds", ["qm9", "geom"])
def test_host_tables_and_kl(ds):
    """Product-side host logic: table packing and the categorical KL (no GPU, no oracle arithmetic in the product)."""
    info = pkg.dataset_info(ds)
    np.testing.assert_array_equal(np.stack(pkg.get_bond_length_arrays(info["atom_encoder"])), G[f"{ds}_bonds"])
    tb = pkg.stability.bond_tables(info)
    T = len(info["atom_decoder"])
    assert tb.num_types == T and tb.limit_bonds_to_one == 0
    thr1 = np.asarray(list(tb.thr1)).reshape(16, 16)[:T, :T]
    np.testing.assert_array_equal(thr1, G[f"{ds}_bonds"][0] + 10)
    thr3 = np.asarray(list(tb.thr3)).reshape(16, 16)[:T, :T]
    np.testing.assert_array_equal(thr3, G[f"{ds}_bonds"][2] + 3)
    dec = info["atom_decoder"]
    assert tb.allowed_mask[dec.index("C")] == 1 << 4 and tb.allowed_mask[dec.index("H")] == 1 << 1
    if "P" in dec:
        assert tb.allowed_mask[dec.index("P")] == (1 << 3) | (1 << 5)
    cat = pkg.CategoricalDistribution(info["atom_types"], info["atom_encoder"])
    assert abs(cat.kl_divergence(G[f"{ds}_types"]) - float(G[f"{ds}_kl"])) < 1e-12
    with pytest.raises(RuntimeError):
        pkg.check_molecular_stability_batch(torch.zeros(3, 3), torch.zeros(3, dtype=torch.int64), torch.tensor([3]), info)


@pytest.mark.gpu
@pytest.mark.parametrize("ds", ["qm9", "geom"])
def test_hip_matches_reference_golden(ds):
    info = pkg.dataset_info(ds)
    sizes = torch.from_numpy(G[f"{ds}_sizes"].astype(np.int64))
    x = torch.from_numpy(G[f"{ds}_x"]).cuda()
    t = torch.from_numpy(G[f"{ds}_types"]).cuda()
    out = pkg.check_molecular_stability_batch(x, t, sizes, info).cpu().numpy()
    np.testing.assert_array_equal(out, G[f"{ds}_result"])
    # strided view of a sampler-shaped buffer [N, 3+F] and the single-molecule mirror
    xh = torch.randn(x.shape[0], 9, device="cuda")
    xh[:, :3] = x
    np.testing.assert_array_equal(pkg.check_molecular_stability_batch(xh, t, sizes, info).cpu().numpy(), G[f"{ds}_result"])
    o = int(sizes[:6].sum())
    n6 = int(sizes[6])
    s, k, n = pkg.check_molecular_stability(x[o:o + n6].contiguous(), t[o:o + n6], info)
    assert [int(s), k, n] == list(G[f"{ds}_result"][6])


@pytest.mark.gpu
@pytest.mark.parametrize("ds,B", [("qm9", 1024), ("geom", 256)])
def test_hip_matches_oracle_on_sampler_shaped_batches(ds, B):
    """Benchmark-sized ragged batch from the dataset histogram; molecules whose closest distance-to-threshold is below 1e-3 pm
    (where the reference's own two cdist code paths may disagree) are excluded from the comparison."""
    info = pkg.dataset_info(ds)
    bonds = so.bond_length_arrays(TABLES, info["atom_encoder"])
    torch.manual_seed(1)
    sizes = pkg.NumNodesDistribution(info["n_nodes"]).sample(B)
    g = np.random.default_rng(3)
    xs, ts = [], []
    for n in sizes.tolist():
        pos = np.zeros((n, 3), np.float32)
        for i in range(1, n):
            d = g.normal(size=3)
            pos[i] = pos[int(g.integers(max(0, i - 3), i))] + d / np.linalg.norm(d) * g.uniform(0.9, 1.6)
        xs.append(pos)
        ts.append(g.integers(0, len(info["atom_decoder"]), size=n))
    x = torch.from_numpy(np.concatenate(xs)).cuda()
    t = torch.from_numpy(np.concatenate(ts)).cuda()
    out = pkg.check_molecular_stability_batch(x, t, sizes, info).cpu().numpy()
    checked = 0
    for m in range(0, B, max(1, B // 128)):
        if so.threshold_gap(xs[m], ts[m], bonds, TABLES["margins"]) < 1e-3:
            continue
        ref = so.check_molecular_stability(xs[m], ts[m], info["atom_decoder"], TABLES, bonds)
        assert list(out[m]) == [int(ref[0]), ref[1], ref[2]], m
        checked += 1
    assert checked > 64
    assert (out[:, 2] == sizes.numpy()).all()
    # empty batch and bad arguments
    assert pkg.check_molecular_stability_batch(x[:0], t[:0], sizes[:0], info).shape == (0, 3)
    with pytest.raises(ValueError):
        pkg.check_molecular_stability_batch(x, t[:-1], sizes, info)


@pytest.mark.gpu
def test_evaluation_driver_shape():
    """sample_and_analyze (qm9_mol_gen_ddpm.py:747-885): ragged batches from the size histogram, device-side statistics that agree
    with the oracle applied to the same samples."""
    cfgs = pkg.default_cfgs("qm9")
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda()
    seen = []
    orig = pkg.mol_gen_ddpm.check_molecular_stability_batch

    def spy(xh, types, num_nodes, info, *a, **k):
        r = orig(xh, types, num_nodes, info, *a, **k)
        seen.append((xh.cpu().numpy().copy(), types.cpu().numpy().copy(), num_nodes.cpu().numpy().copy(), r.cpu().numpy().copy()))
        return r

    pkg.mol_gen_ddpm.check_molecular_stability_batch = spy
    try:
        torch.manual_seed(5)
        res = model.sample_and_analyze(num_samples=23, batch_size=10, num_timesteps=25)
    finally:
        pkg.mol_gen_ddpm.check_molecular_stability_batch = orig
    assert [len(s[2]) for s in seen] == [10, 10, 3]
    info = pkg.dataset_info("qm9")
    bonds = so.bond_length_arrays(TABLES, info["atom_encoder"])
    st, atoms, stable_atoms, types_all = 0, 0, 0, []
    for xh, types, nn_, r in seen:
        off = np.r_[0, np.cumsum(nn_)]
        for m, (a, b) in enumerate(zip(off[:-1], off[1:])):
            x = np.ascontiguousarray(xh[a:b, :3])
            ref = so.check_molecular_stability(x, types[a:b], info["atom_decoder"], TABLES, bonds)
            if so.threshold_gap(x, types[a:b], bonds, TABLES["margins"]) > 1e-3:
                assert list(r[m]) == [int(ref[0]), ref[1], ref[2]]
            st += int(r[m][0]); stable_atoms += int(r[m][1]); atoms += int(r[m][2])
        types_all += list(types)
    assert res["mol_stable"] == st / 23 and res["atm_stable"] == stable_atoms / atoms
    kl = so.kl_divergence(info["atom_types"], 5, types_all)
    assert (np.isnan(kl) and np.isnan(res["kl_div_atom_types"])) or abs(kl - res["kl_div_atom_types"]) < 1e-9
    assert res["validity"] is None


@pytest.mark.gpu
def test_concurrent_batches_equal_sequential():
    """mol_gen_sample_concurrent (one handle + stream per batch, interleaved launches) gives bit-identical samples to the sequential
    calls with the same seeds, and the evaluation driver's statistics do not depend on `concurrent_batches`."""
    cfgs = pkg.default_cfgs("qm9")
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda()
    lists = [torch.tensor([5, 19, 7]), torch.tensor([12, 3]), torch.tensor([19, 19, 19, 4])]
    seq = [model.ddpm.mol_gen_sample(len(nn_), nn_, "cuda", num_timesteps=6, seed=77 + b)[0].clone() for b, nn_ in enumerate(lists)]
    con = model.ddpm.mol_gen_sample_concurrent(lists, "cuda", num_timesteps=6, seeds=[77, 78, 79])
    for a, (b_, bi, _) in zip(seq, con):
        assert torch.equal(a, b_)
    torch.manual_seed(3)
    r1 = model.sample_and_analyze(num_samples=17, batch_size=5, num_timesteps=8)
    torch.manual_seed(3)
    r3 = model.sample_and_analyze(num_samples=17, batch_size=5, num_timesteps=8, concurrent_batches=3)
    assert r1["mol_stable"] == r3["mol_stable"] and r1["atm_stable"] == r3["atm_stable"]
    assert (r1["kl_div_atom_types"] == r3["kl_div_atom_types"]) or (np.isnan(r1["kl_div_atom_types"]) and np.isnan(r3["kl_div_atom_types"]))
    model.ddpm.release_lanes()


@pytest.mark.gpu
def test_sample_sharded_single_rank_equals_direct_call():
    """parallel.sample_sharded without a process group is the plain single-GPU sample (rank 0 of 1, seed + 0)."""
    par = importlib.import_module("bio-diffusion_amd.parallel")
    cfgs = pkg.default_cfgs("qm9")
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda()
    nn_ = torch.tensor([19, 5, 12, 7, 19, 3, 9, 11])
    a, nn_out = par.sample_sharded(model.ddpm, nn_, "cuda", num_timesteps=6, seed=5, lanes=2)
    b, _, _ = model.ddpm.mol_gen_sample(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=6, seed=5, lanes=2)
    assert torch.equal(a, b) and torch.equal(nn_out.cpu(), nn_.long())
    model.ddpm.release_lanes()


@pytest.mark.gpu
def test_chain_sampling_entry_points(tmp_path):
    """sample_chain_and_save (qm9_mol_gen_ddpm.py:957-1060) and generate_molecules(sample_chain=True) (:1119-1128, 1183-1207): frames in generation
    order, the last one repeated 10 times, one XYZ file per frame; the frames are those of mol_gen_sample(return_frames=...) (parity:
    tests/test_gpu_parity.py::test_sampling_with_chain_frames)."""
    cfgs = pkg.default_cfgs("qm9")
    cfgs["diffusion_cfg"]["num_timesteps"] = 8
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    model = model.cuda()
    stable = model.sample_chain_and_save(keep_frames=4, num_tries=1, sampling_output_dir=str(tmp_path), seed=3)
    files = sorted(os.listdir(tmp_path / "chain"))
    assert files == ["chain_%03d.xyz" % i for i in range(14)] and isinstance(stable, bool)
    texts = [open(tmp_path / "chain" / f).read() for f in files]
    assert all(t.startswith("19\n\n") and len(t.splitlines()) == 21 for t in texts)
    assert len(set(texts[:4])) == 4 and len(set(texts[3:])) == 1               # 4 distinct frames, then the final one 11 times
    want, _, _ = model.ddpm.mol_gen_sample(num_samples=1, num_nodes=torch.tensor([19]), device="cuda", return_frames=4, seed=3)
    last = want[0].cpu()
    lines = texts[-1].splitlines()[2:]
    dec = pkg.dataset_info("qm9")["atom_decoder"]
    for i, ln in enumerate(lines):
        sym, xs, ys, zs = ln.split()
        assert sym == dec[int(last[i, 3:8].argmax())] and abs(float(xs) - last[i, 0].item()) <= 1e-6 * max(1.0, abs(last[i, 0].item()))
    mols = model.generate_molecules(ddpm_mode="unconditional", num_samples=1, num_nodes=torch.tensor([7]), sample_chain=True, seed=3)
    assert len(mols) == 8 and all(m[0].shape == (7, 3) and m[1].shape == (7,) for m in mols)
    fr, _, _ = model.ddpm.mol_gen_sample(num_samples=1, num_nodes=torch.tensor([7]), device="cuda", return_frames=8, seed=3)
    assert torch.equal(mols[-1][0], fr[0, :, :3].cpu()) and torch.equal(mols[0][0], fr[7, :, :3].cpu())
    with pytest.raises(AssertionError):
        model.generate_molecules(ddpm_mode="unconditional", num_samples=2, sample_chain=True)


def _two_rank_model(cond):
    cfgs = pkg.default_cfgs("qm9", ("alpha",) if cond else ())
    torch.manual_seed(0)
    model = pkg.QM9MoleculeGenerationDDPM(**cfgs)
    with torch.no_grad():
        for p in model.ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(0.25)
    return model.cuda()


def _two_rank_worker(rank, world, port, nn_all, ctx_all, q):
    """One rank of the multi-GPU recipe, all ranks on GPU 0 (one-GPU box): HIP path + `parallel.sample_sharded` + gloo gather."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    par = importlib.import_module("bio-diffusion_amd.parallel")
    model = _two_rank_model(ctx_all is not None)
    xh, nn_g = par.sample_sharded(model.ddpm, nn_all, "cuda", context=ctx_all, num_timesteps=6, seed=40, lanes=1)
    maps = open("/proc/self/maps").read()
    q.put((rank, xh.cpu().numpy(), nn_g.cpu().numpy(), "libgcdm_hip.so" in maps))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("cond", [False, True])
def test_two_ranks_on_one_gpu_equal_independent_shard_runs(cond):
    """SURVEY 8(e) on hardware: 2 processes (both on GPU 0: this pool has one-GPU boxes) run `parallel.sample_sharded` with the HIP path;
    every rank ends up with all samples, in the original molecule order, bit-identical to the two independent single-handle runs of the
    shards (seed + rank, own context rows)."""
    import socket
    import torch.multiprocessing as mp
    par = importlib.import_module("bio-diffusion_amd.parallel")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    nn_all = torch.tensor([19, 5, 12, 7, 19, 3, 9])
    ctx_all = torch.tensor([[0.5], [-1.0], [2.0], [0.25], [-0.75], [1.5], [0.0]]) if cond else None
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, nn_all, ctx_all, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    model = _two_rank_model(cond)
    want = []
    for r in range(2):
        lo, hi = par.shard_range(len(nn_all), r, 2)
        o, _, _ = model.ddpm.mol_gen_sample(num_samples=hi - lo, num_nodes=nn_all[lo:hi], device="cuda", num_timesteps=6,
                                            context=None if ctx_all is None else ctx_all[lo:hi], seed=40 + r, lanes=1)
        want.append(o.cpu().clone())
    want = torch.cat(want).numpy()
    for r in range(2):
        assert res[r][3], "the rank did not load the HIP library"
        assert np.array_equal(res[r][1], want) and res[r][2].tolist() == nn_all.tolist()


@pytest.mark.gpu
def test_lanes_follow_weight_updates():
    """Extra handles (lanes) are copies of the weights: after the parameters change, a sliced run must use the NEW weights."""
    model = _two_rank_model(False)
    ddpm = model.ddpm
    nn_ = torch.tensor([19, 5, 12, 7, 19, 3, 9, 11])
    kw = dict(num_samples=len(nn_), num_nodes=nn_, device="cuda", num_timesteps=5, seed=9)
    a2, _, _ = ddpm.mol_gen_sample(lanes=2, **kw)
    a2 = a2.clone()
    with torch.no_grad():
        for p in ddpm.dynamics_network.parameters():
            if p.dim() == 2:
                p.mul_(1.5)
    b1, _, _ = ddpm.mol_gen_sample(lanes=1, **kw)
    b1 = b1.clone()
    b2, _, _ = ddpm.mol_gen_sample(lanes=2, **kw)
    scale = max(1.0, b1[:, :3].abs().max().item())
    assert (b2[:, :3] - b1[:, :3]).abs().max().item() <= 1e-4 * scale          # new weights in the lanes ...
    assert (b2[:, :3] - a2[:, :3]).abs().max().item() > 3e-4 * scale           # ... and they do differ from the old ones
    ddpm.release_lanes()


def test_sdf_records_round_trip(tmp_path):
    """V2000 records of the SDF writer (mirror of write_sdf_file, src/models/components/__init__.py:372-378): layout and round trip (host only)."""
    sdf = pkg.sdf
    mols = [sdf.Molecule(["C", "O", "H", "H"], np.array([[0.0, 0.0, 0.0], [1.2, 0.0, 0.0], [-0.55, 0.94, 0.0], [-0.55, -0.94, 0.0001]]),
                         [(1, 0, 2), (2, 0, 1), (3, 0, 1)]),
            None,
            sdf.Molecule(["N", "H", "H", "H", "H"], np.random.default_rng(0).normal(size=(5, 3)), [(1, 0, 1), (2, 0, 1), (3, 0, 1), (4, 0, 1)],
                         np.array([1, 0, 0, 0, 0]))]
    path = tmp_path / "m.sdf"
    sdf.write_sdf_file(path, mols)
    text = open(path).read()
    recs = text.split("$$$$\n")
    assert len(recs) == 3 and recs[2] == ""                       # None is skipped, every record ends with $$$$
    l = recs[0].split("\n")
    assert l[0] == "" and l[1] == "     RDKit          3D" and l[2] == "" and l[3] == "  4  3  0  0  0  0  0  0  0  0999 V2000"
    assert l[4] == "    0.0000    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0" and len(l[4]) == 69
    assert l[8] == "  2  1  2  0" and l[11] == "M  END"
    assert "M  CHG  1   1   1" in recs[1]
    back = sdf.read_sdf_file(path)
    assert len(back) == 2
    for m, b in zip([mols[0], mols[2]], back):
        assert b.symbols == m.symbols and b.bonds == m.bonds
        np.testing.assert_allclose(b.positions, m.positions, atol=5e-5)
        assert (b.charges is None and m.charges is None) or list(b.charges) == list(m.charges)


@pytest.mark.gpu
@pytest.mark.parametrize("ds", ["qm9", "geom"])
def test_sdf_bond_tables_from_the_device_match_the_oracle(ds, tmp_path):
    """Bond orders of every pair from the device kernel (gcdm_bond_orders) == the oracle's get_bond_order_batch on the golden molecules
    (incl. the matrix the reference itself produced for molecule 6), and the SDF written from them parses back to the same connection tables."""
    info = pkg.dataset_info(ds)
    bonds = so.bond_length_arrays(TABLES, info["atom_encoder"])
    sizes = torch.from_numpy(G[f"{ds}_sizes"].astype(np.int64))
    x = torch.from_numpy(G[f"{ds}_x"]).cuda()
    t = torch.from_numpy(G[f"{ds}_types"]).cuda()
    limit = ds == "geom"                                               # make_mol_edm: limit_bonds_to_one = "GEOM" in dataset_info["name"]
    E = pkg.bond_order_matrices(x, t, sizes, info)
    for m, (xm, tm) in enumerate(molecules(ds)):
        n = len(tm)
        if so.threshold_gap(xm, tm, bonds, TABLES["margins"]) < 1e-3:
            continue                                                   # a distance within 1e-3 pm of a threshold: cdist's last bits decide
        a1, a2 = np.meshgrid(tm.astype(np.int64), tm.astype(np.int64), indexing="xy")
        want = so.bond_order_batch(a1.reshape(-1), a2.reshape(-1), so.pair_distances(xm).reshape(-1), bonds, TABLES["margins"], limit).reshape(n, n)
        np.fill_diagonal(want, 0)
        np.testing.assert_array_equal(E[m].astype(np.int64), want)
    ref6 = G[f"{ds}_order1_mol6" if limit else f"{ds}_order_mol6"].reshape(E[6].shape).copy()
    np.fill_diagonal(ref6, 0)
    np.testing.assert_array_equal(E[6].astype(np.int64), ref6)
    mols = pkg.build_molecules(x, t, sizes, info)
    path = tmp_path / f"{ds}.sdf"
    pkg.write_sdf_file(path, [m for m in mols if len(m.symbols) <= 999 and len(m.bonds) <= 999])
    back = pkg.sdf.read_sdf_file(path)
    k = 0
    for m, e in zip(mols, E):
        if len(m.symbols) > 999 or len(m.bonds) > 999:
            continue
        tri = np.tril(e, -1)
        assert back[k].bonds == [(int(i), int(j), int(tri[i, j])) for i, j in zip(*np.nonzero(tri))]
        assert back[k].symbols == m.symbols
        k += 1
