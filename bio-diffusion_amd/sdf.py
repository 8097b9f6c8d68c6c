"""SDF / V2000 molfile emission of generated molecules without RDKit.

Mirror of ``write_sdf_file`` (src/models/components/__init__.py:372-378) fed by ``build_molecule`` /
``make_mol_edm`` (src/datamodules/components/edm/rdkit_functions.py:209-320): atoms from the decoded atom types, bonds from the
EDM length tables -- the pairwise bond orders come from the device (``gcdm_bond_orders``: the kernel that also does the stability
check), the lower triangle is kept (``torch.tril(E, -1)``, "the graph should be DIRECTED") and bonds are listed in ``torch.nonzero``
order with the larger atom index first, exactly as ``mol.AddBond(bond[0], bond[1], ...)`` receives them.  The text follows the CTfile
V2000 layout RDKit's ``SDWriter`` emits for a molecule with one 3D conformer (name line empty, program line ``     RDKit          3D``);
RDKit is not in the image, so byte parity with its writer is not pinned -- the tests parse the records back and compare the
connection tables with the oracle's bond orders.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native
from .stability import _offsets, bond_tables


@dataclass
class Molecule:
    """What the reference holds in a ``Chem.RWMol`` after ``make_mol_edm(..., add_coords=True)``."""
    symbols: List[str]
    positions: np.ndarray                                   # [n, 3] float
    bonds: List[Tuple[int, int, int]] = field(default_factory=list)   # (begin, end, order), begin > end, nonzero order of tril(E, -1)
    charges: Optional[np.ndarray] = None                    # [n] int formal charges (None: all zero)


@torch.inference_mode()
def bond_order_matrices(positions: torch.Tensor, atom_types: torch.Tensor, num_nodes: torch.Tensor, dataset_info: Dict[str, Any]) -> List[np.ndarray]:
    """Per molecule the n x n uint8 bond-order matrix of ``get_bond_order_batch`` (device kernel; GEOM limits bonds to single ones
    like ``make_mol_edm``: ``"GEOM" in dataset_info["name"]``)."""
    if not positions.is_cuda:
        raise RuntimeError("bond_order_matrices runs on the GPU only (no CPU fallback)")
    if positions.dtype != torch.float32 or positions.dim() != 2 or positions.shape[1] < 3 or positions.stride(1) != 1:
        raise ValueError("positions must be fp32 [N, >=3] with unit column stride")
    lib = _native.load()
    dev = positions.device
    nn_ = num_nodes.to(torch.int64).cpu()
    N, B = positions.shape[0], len(nn_)
    if int(nn_.sum()) != N or atom_types.shape[0] != N:
        raise ValueError("num_nodes / atom_types do not match positions")
    poff = torch.zeros(B + 1, dtype=torch.int64)
    poff[1:] = torch.cumsum(nn_ * nn_, 0)
    orders = torch.zeros(int(poff[-1]), dtype=torch.uint8, device=dev)
    types = atom_types.to(device=dev, dtype=torch.int32).contiguous()
    off = _offsets(nn_, dev)
    tb = bond_tables(dataset_info, "GEOM" in str(dataset_info.get("name", "")).upper())
    pdev = poff[:-1].to(dev)
    with torch.cuda.device(dev):
        st = lib.gcdm_bond_orders(tb, positions.data_ptr(), positions.stride(0) if N > 1 else max(positions.shape[1], 3), types.data_ptr(),
                                  off.data_ptr(), B, pdev.data_ptr(), orders.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
    if st != 0:
        raise _native.NativeError(f"gcdm_bond_orders failed with status {st}")
    flat = orders.cpu().numpy()
    return [flat[int(poff[m]):int(poff[m + 1])].reshape(int(nn_[m]), int(nn_[m])) for m in range(B)]


def build_molecules(positions: torch.Tensor, atom_types: torch.Tensor, num_nodes: torch.Tensor, dataset_info: Dict[str, Any],
                    charges: Optional[torch.Tensor] = None) -> List[Molecule]:
    """``build_molecule`` (rdkit_functions.py:209-234, EDM bond rules) for every molecule of a flat batch."""
    E = bond_order_matrices(positions, atom_types, num_nodes, dataset_info)
    dec = dataset_info["atom_decoder"]
    pos = positions[:, :3].detach().cpu().numpy()
    typ = atom_types.detach().cpu().numpy()
    chg = None if charges is None or charges.numel() == 0 else charges.detach().reshape(-1).round().to(torch.int64).cpu().numpy()
    out, o = [], 0
    for m, e in enumerate(E):
        n = e.shape[0]
        tri = np.tril(e, -1)
        ii, jj = np.nonzero(tri)                     # row-major, like torch.nonzero
        out.append(Molecule(symbols=[dec[int(t)] for t in typ[o:o + n]], positions=pos[o:o + n].copy(),
                            bonds=[(int(i), int(j), int(tri[i, j])) for i, j in zip(ii, jj)], charges=None if chg is None else chg[o:o + n].copy()))
        o += n
    return out


_CHG_CODE = {0: 0, 3: 1, 2: 2, 1: 3, -1: 5, -2: 6, -3: 7}       # CTfile atom-block charge field


def molblock(mol: Molecule, name: str = "") -> str:
    """One V2000 connection table (counts line, atom block, bond block, M  CHG, M  END)."""
    n, nb = len(mol.symbols), len(mol.bonds)
    if n > 999 or nb > 999:
        raise ValueError("V2000 holds at most 999 atoms / bonds")
    lines = [name, "     RDKit          3D", "", f"{n:3d}{nb:3d}  0  0  0  0  0  0  0  0999 V2000"]
    for i, s in enumerate(mol.symbols):
        x, y, z = (float(v) for v in mol.positions[i])
        c = 0 if mol.charges is None else _CHG_CODE.get(int(mol.charges[i]), 0)
        lines.append(f"{x:10.4f}{y:10.4f}{z:10.4f} {s:<3s} 0{c:3d}  0  0  0  0  0  0  0  0  0  0")
    for b, e, o in mol.bonds:
        lines.append(f"{b + 1:3d}{e + 1:3d}{o:3d}  0")
    if mol.charges is not None:
        ch = [(i + 1, int(q)) for i, q in enumerate(mol.charges) if int(q) != 0]
        for k in range(0, len(ch), 8):
            part = ch[k:k + 8]
            lines.append(f"M  CHG{len(part):3d}" + "".join(f" {i:3d} {q:3d}" for i, q in part))
    lines.append("M  END")
    return "\n".join(lines) + "\n"


def write_sdf_file(sdf_path: Any, molecules: Sequence[Optional[Molecule]], verbose: bool = False) -> None:
    """src/models/components/__init__.py:372-378: one record per molecule that is not None."""
    with open(str(sdf_path), "w") as f:
        for m in molecules:
            if m is not None:
                f.write(molblock(m))
                f.write("$$$$\n")


def read_sdf_file(sdf_path: Any) -> List[Molecule]:
    """Parser of the records above (tests; round trips of generated files)."""
    inv = {v: k for k, v in _CHG_CODE.items()}
    out: List[Molecule] = []
    with open(str(sdf_path)) as f:
        recs = f.read().split("$$$$\n")
    for rec in recs:
        ls = rec.split("\n")
        if len(ls) < 5 or not ls[3].rstrip().endswith("V2000"):
            continue
        n, nb = int(ls[3][0:3]), int(ls[3][3:6])
        sym, pos, chg = [], [], []
        for l in ls[4:4 + n]:
            pos.append([float(l[0:10]), float(l[10:20]), float(l[20:30])])
            sym.append(l[31:34].strip())
            chg.append(inv.get(int(l[36:39]), 0))
        bonds = [(int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])) for l in ls[4 + n:4 + n + nb]]
        for l in ls[4 + n + nb:]:
            if l.startswith("M  CHG"):
                k = int(l[6:9])
                for t in range(k):
                    chg[int(l[10 + 8 * t:13 + 8 * t]) - 1] = int(l[14 + 8 * t:17 + 8 * t])
        out.append(Molecule(sym, np.array(pos), bonds, np.array(chg) if any(chg) else None))
    return out
