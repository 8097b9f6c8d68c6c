"""Molecular stability of sampled batches on the device, and the node-type KL that goes with it.

Interface mirror of the reference's evaluation helpers:
  * get_bond_length_arrays      src/datamodules/components/edm/__init__.py:25-41
  * check_molecular_stability   src/datamodules/components/edm/__init__.py:91-122   (single molecule; same signature and result)
  * CategoricalDistribution     src/models/__init__.py:418-439
plus `check_molecular_stability_batch`, which evaluates a whole flat batch in one launch of `gcdm_check_stability`
(include/gcdm_hip.h) -- what `analyze_samples` (src/models/qm9_mol_gen_ddpm.py:859-868) loops over on the host.
No arithmetic happens here: this module builds the constant tables and calls the C ABI; CPU tensors are rejected.
"""
from __future__ import annotations

import json
import os
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _native

_TABLES: Optional[Dict[str, Any]] = None


def bond_constants() -> Dict[str, Any]:
    """`margin1..3`, `allowed_bonds`, `bonds1..3` of src/datamodules/components/edm/constants.py:20-72 (data/bond_tables.json)."""
    global _TABLES
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "bond_tables.json")) as f:
            _TABLES = json.load(f)
    return _TABLES


def get_bond_length_arrays(atom_mapping: Dict[str, int]) -> List[np.ndarray]:
    """edm/__init__.py:25-41."""
    const = bond_constants()
    arrays = []
    for i in range(3):
        bond_dict = const[f"bonds{i + 1}"]
        arr = np.zeros((len(atom_mapping), len(atom_mapping)))
        for a1, i1 in atom_mapping.items():
            for a2, i2 in atom_mapping.items():
                arr[i1, i2] = bond_dict[a1][a2] if a1 in bond_dict and a2 in bond_dict[a1] else 0
        assert np.all(arr == arr.T)
        arrays.append(arr)
    return arrays


def ensure_bond_arrays(dataset_info: Dict[str, Any]) -> Dict[str, Any]:
    """What the reference's `on_validation_start` / mol_gen_eval.py:113-122 do before the first stability check."""
    if any(dataset_info.get(k) is None for k in ("bonds1", "bonds2", "bonds3")):
        b = get_bond_length_arrays(dataset_info["atom_encoder"])
        dataset_info["bonds1"], dataset_info["bonds2"], dataset_info["bonds3"] = b
    return dataset_info


def bond_tables(dataset_info: Dict[str, Any], limit_bonds_to_one: bool = False) -> "_native.GcdmBondTables":
    """Packs `dataset_info["bonds1..3"]` (+ margins) and `allowed_bonds` into the ABI struct."""
    ensure_bond_arrays(dataset_info)
    const = bond_constants()
    decoder = dataset_info["atom_decoder"]
    T, M = len(decoder), _native.STABILITY_MAX_TYPES
    if T > M:
        raise ValueError(f"at most {M} atom types")
    tb = _native.GcdmBondTables()
    tb.num_types, tb.limit_bonds_to_one = T, int(bool(limit_bonds_to_one))
    for name, key, margin in (("thr1", "bonds1", const["margins"][0]), ("thr2", "bonds2", const["margins"][1]),
                              ("thr3", "bonds3", const["margins"][2])):
        arr = np.asarray(dataset_info[key], np.float64)
        dst = getattr(tb, name)
        for a in range(T):
            for b in range(T):
                dst[a * M + b] = float(arr[a, b] + margin)
    for a, sym in enumerate(decoder):
        allowed = const["allowed_bonds"][sym]
        mask = 0
        for nb in ([allowed] if isinstance(allowed, int) else allowed):
            mask |= 1 << int(nb)
        tb.allowed_mask[a] = mask
    return tb


def _offsets(num_nodes: torch.Tensor, device) -> torch.Tensor:
    off = torch.zeros(len(num_nodes) + 1, dtype=torch.int32)
    off[1:] = torch.cumsum(num_nodes.to(torch.int64).cpu(), 0).to(torch.int32)
    return off.to(device)


@torch.inference_mode()
def check_molecular_stability_batch(positions: torch.Tensor, atom_types: torch.Tensor, num_nodes: torch.Tensor,
                                    dataset_info: Dict[str, Any], limit_bonds_to_one: bool = False) -> torch.Tensor:
    """`positions` [N, >=3] fp32 (row stride arbitrary: a view `xh[:, :3]` of the sampler output is fine), `atom_types` [N] integer,
    `num_nodes` [B] atoms per molecule (flat batch order).  Returns int32 [B, 3] on the device: (molecule_stable, nr_stable_atoms, n)."""
    if not positions.is_cuda:
        raise RuntimeError("check_molecular_stability_batch runs on the GPU only (no CPU fallback)")
    if positions.dtype != torch.float32 or positions.dim() != 2 or positions.shape[1] < 3 or positions.stride(1) != 1:
        raise ValueError("positions must be fp32 [N, >=3] with unit column stride")
    N = positions.shape[0]
    if int(num_nodes.sum()) != N or atom_types.shape[0] != N:
        raise ValueError("num_nodes / atom_types do not match positions")
    lib = _native.load()
    dev = positions.device
    types = atom_types.to(device=dev, dtype=torch.int32).contiguous()
    off = _offsets(num_nodes, dev)
    B = len(num_nodes)
    out = torch.empty((B, 3), dtype=torch.int32, device=dev)
    tb = bond_tables(dataset_info, limit_bonds_to_one)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        st = lib.gcdm_check_stability(tb, positions.data_ptr(), positions.stride(0) if N > 1 else max(positions.shape[1], 3),
                                      types.data_ptr(), off.data_ptr(), B, out.data_ptr(), stream)
    if st != 0:
        raise _native.NativeError(f"gcdm_check_stability failed with status {st}")
    return out


def check_molecular_stability(positions: torch.Tensor, atom_types: torch.Tensor, dataset_info: Dict[str, Any],
                              verbose: bool = False) -> Tuple[bool, int, int]:
    """edm/__init__.py:91-122 for one molecule (device tensors)."""
    assert positions.dim() == 2 and positions.shape[1] == 3
    r = check_molecular_stability_batch(positions.contiguous(), atom_types, torch.tensor([positions.shape[0]]), dataset_info)
    s, k, n = (int(v) for v in r[0].tolist())
    return bool(s), k, n


class CategoricalDistribution:
    """src/models/__init__.py:418-439 (host statistics over at most 16 categories)."""
    EPS = 1e-10

    def __init__(self, histogram_dict: Dict[int, int], mapping: Dict[str, int]):
        histogram = np.zeros(len(mapping))
        for k, v in histogram_dict.items():
            histogram[int(k)] = v
        self.p = histogram / histogram.sum()
        self.mapping = mapping

    def kl_divergence(self, other_samples: Union[Sequence[int], torch.Tensor, np.ndarray]) -> float:
        counts = np.bincount(np.asarray(torch.as_tensor(other_samples).cpu(), np.int64).reshape(-1), minlength=len(self.mapping))
        q = counts / counts.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(-np.sum(self.p * np.log(q / self.p + self.EPS)))
