"""Regenerates bio-diffusion_amd/data/dataset_info.json -- DATA the sampler consults (atom vocabulary, molecule-size histogram, atom-type
counts) -- from the reference's src/datamodules/components/edm/datasets_config.py, PRESERVING the insertion order of the size histogram:
`NumNodesDistribution` (src/models/__init__.py:264-308) numbers its categories in that order, so the order decides both the layout of the
`num_nodes_distribution.*` buffers in a checkpoint and which sizes a given torch seed draws.

    python tests/golden/make_dataset_info.py        (build container only)
"""
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [HERE, os.path.dirname(HERE), ROOT]
import ref_harness as rh  # noqa: E402


def main():
    rh.install_stubs()
    dc = importlib.import_module("src.datamodules.components.edm.datasets_config")
    out = {}
    for key, (name, remove_h) in {"qm9": ("QM9", False), "qm9_second_half": ("QM9_second_half", False), "geom": ("GEOM", False)}.items():
        d = dc.get_dataset_info(name, remove_h)
        dec = list(d["atom_decoder"])
        assert {a: i for i, a in enumerate(dec)} == dict(d["atom_encoder"])
        types = d["atom_types"]
        assert sorted(types) == list(range(len(dec)))        # consulted by key (CategoricalDistribution): order irrelevant, stored by index
        out[key] = {"name": d["name"], "with_h": bool(d["with_h"]), "max_n_nodes": int(d["max_n_nodes"]), "atom_decoder": dec,
                    "n_nodes_hist": [[int(k), int(v)] for k, v in d["n_nodes"].items()],           # insertion order kept
                    "atom_type_counts": [int(types[i]) for i in range(len(dec))]}
    path = os.path.join(ROOT, "bio-diffusion_amd", "data", "dataset_info.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, {k: v["n_nodes_hist"][:4] for k, v in out.items()})


if __name__ == "__main__":
    main()
