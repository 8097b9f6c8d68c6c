"""XYZ emission of sampled batches -- host-side file formatting mirroring the reference's writers (byte-identical files):
  * save_xyz_file    src/models/components/__init__.py:325-356   (one `<name>_%03d.xyz` per molecule, "%s %.9f %.9f %.9f")
  * write_xyz_file   src/models/components/__init__.py:359-370   (single molecule, atom-type column as given, "%.3f")
One device->host copy per batch (positions + arg-max atom types); no per-atom tensor indexing as in the reference's loop.
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Sequence

import numpy as np
import torch


def save_xyz_file(path: str, positions: torch.Tensor, one_hot: torch.Tensor, charges: Optional[torch.Tensor], dataset_info: Dict[str, Any],
                  id_from: int = 0, name: str = "molecule", batch_index: Optional[torch.Tensor] = None) -> None:
    os.makedirs(path, exist_ok=True)
    if batch_index is None:
        batch_index = torch.zeros(len(one_hot))
    pos = positions.detach().to(torch.float32).cpu().numpy()
    atoms = torch.argmax(one_hot, dim=-1).cpu().numpy()
    bi = batch_index.detach().cpu().numpy()
    decoder = dataset_info["atom_decoder"]
    for b in np.unique(bi):                                   # sorted, like torch.unique
        sel = bi == b
        p, a = pos[sel], atoms[sel]
        lines = ["%d\n\n" % int(sel.sum())]
        lines += ["%s %.9f %.9f %.9f\n" % (decoder[int(a[i])], p[i, 0], p[i, 1], p[i, 2]) for i in range(len(a))]
        with open(path + name + "_" + "%03d.xyz" % (int(b) + id_from), "w") as f:
            f.write("".join(lines))


def write_xyz_file(positions: torch.Tensor, atom_types: Sequence[Any], filename: str) -> None:
    assert len(positions) == len(atom_types)
    pos = torch.as_tensor(positions).detach().cpu().numpy()
    out = f"{len(pos)}\n\n"
    for i in range(len(pos)):
        out += f"{atom_types[i]} {pos[i, 0]:.3f} {pos[i, 1]:.3f} {pos[i, 2]:.3f}\n"
    with open(filename, "w") as f:
        f.write(out)
