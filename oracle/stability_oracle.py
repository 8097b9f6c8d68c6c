"""CPU restatement of the reference's molecular-stability check -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product path
(bio-diffusion_amd/) never does.  Pinned by tests/golden/stability.npz, which holds the outputs of the reference's own
functions on synthetic molecules (tests/golden/make_stability_golden.py).

Reference (file:line into /root/reference):
  * get_bond_length_arrays        src/datamodules/components/edm/__init__.py:25-41
  * get_bond_order_batch          src/datamodules/components/edm/__init__.py:61-88
  * check_molecular_stability     src/datamodules/components/edm/__init__.py:91-122
  * CategoricalDistribution       src/models/__init__.py:418-439
  * analyze_samples (fractions)   src/models/qm9_mol_gen_ddpm.py:846-885
"""
from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np
import torch


def bond_length_arrays(tables: Dict[str, Any], atom_encoder: Dict[str, int]) -> List[np.ndarray]:
    """edm/__init__.py:25-41 -- [T, T] float64 arrays of single / double / triple bond lengths in pm, 0 where the pair has none."""
    out = []
    for i in range(3):
        bd = tables[f"bonds{i + 1}"]
        arr = np.zeros((len(atom_encoder), len(atom_encoder)))
        for a1, i1 in atom_encoder.items():
            for a2, i2 in atom_encoder.items():
                arr[i1, i2] = bd[a1][a2] if a1 in bd and a2 in bd[a1] else 0
        assert np.all(arr == arr.T)
        out.append(arr)
    return out


def bond_order_batch(atoms1: np.ndarray, atoms2: np.ndarray, distances: np.ndarray, bonds: Sequence[np.ndarray],
                     margins: Sequence[float], limit_bonds_to_one: bool = False) -> np.ndarray:
    """edm/__init__.py:61-88.  `distances` fp32 in Angstrom; compared (after the fp32 multiplication by 100) against float64
    `length + margin`; later assignments overwrite earlier ones; a missing pair has length 0, so it still bonds below the margin."""
    d = (np.float32(100.0) * distances.astype(np.float32)).astype(np.float64)
    order = np.zeros(atoms1.shape, np.int64)
    order[d < bonds[0][atoms1, atoms2] + margins[0]] = 1
    order[d < bonds[1][atoms1, atoms2] + margins[1]] = 2
    order[d < bonds[2][atoms1, atoms2] + margins[2]] = 3
    if limit_bonds_to_one:
        order[order > 1] = 1
    return order


def pair_distances(positions: np.ndarray, direct: bool = False) -> np.ndarray:
    """`torch.cdist(x, x, p=2)` as the reference calls it (:104).  ATen evaluates sqrt(sum d^2) directly for n <= 25 and switches
    to the |x|^2 + |y|^2 - 2 x.y expansion above that (`use_mm_for_euclid_dist_if_necessary`); `direct=True` forces the former."""
    x = torch.from_numpy(np.ascontiguousarray(positions, dtype=np.float32))
    mode = "donot_use_mm_for_euclid_dist" if direct else "use_mm_for_euclid_dist_if_necessary"
    return torch.cdist(x, x, p=2.0, compute_mode=mode).numpy()


def allowed(tables: Dict[str, Any], symbol: str, nr_bonds: int) -> bool:
    """:111-116 -- `allowed_bonds` entries are an int or a list of ints."""
    pb = tables["allowed_bonds"][symbol]
    return (pb == nr_bonds) if isinstance(pb, int) else (nr_bonds in pb)


def check_molecular_stability(positions: np.ndarray, atom_types: np.ndarray, atom_decoder: Sequence[str], tables: Dict[str, Any],
                              bonds: Sequence[np.ndarray], direct: bool = False) -> Tuple[bool, int, int]:
    """edm/__init__.py:91-122 -> (molecule_stable, nr_stable_atoms, n)."""
    n = len(positions)
    dist = pair_distances(positions, direct).reshape(-1)
    t = np.asarray(atom_types, np.int64)
    a1, a2 = np.meshgrid(t, t, indexing="xy")
    order = bond_order_batch(a1.reshape(-1), a2.reshape(-1), dist, bonds, tables["margins"]).reshape(n, n)
    np.fill_diagonal(order, 0)
    nr_bonds = order.sum(axis=1)
    stable = sum(int(allowed(tables, atom_decoder[int(ti)], int(nb))) for ti, nb in zip(t, nr_bonds))
    return stable == n, stable, n


def threshold_gap(positions: np.ndarray, atom_types: np.ndarray, bonds: Sequence[np.ndarray], margins: Sequence[float]) -> float:
    """Smallest |100 d - threshold| in pm over all off-diagonal pairs and the three thresholds (float64): how far the molecule is from
    a bond-order decision that rounding could flip.  Test helper, no reference counterpart."""
    x = np.asarray(positions, np.float64)
    n = len(x)
    if n < 2:
        return np.inf
    d = 100.0 * np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    t = np.asarray(atom_types, np.int64)
    gap = np.inf
    off = ~np.eye(n, dtype=bool)
    for b, m in zip(bonds, margins):
        gap = min(gap, float(np.abs(d - (b[t][:, t] + m))[off].min()))
    return gap


def kl_divergence(histogram: Dict[int, int], num_types: int, samples: Sequence[int], eps: float = 1e-10) -> float:
    """src/models/__init__.py:418-439."""
    h = np.zeros(num_types)
    for k, v in histogram.items():
        h[int(k)] = v
    p = h / h.sum()
    q = np.zeros(num_types)
    for s in samples:
        q[int(s)] += 1
    q = q / q.sum()
    with np.errstate(divide="ignore", invalid="ignore"):
        return float(-np.sum(p * np.log(q / p + eps)))
