"""Golden XYZ files written by the REFERENCE's save_xyz_file / write_xyz_file (src/models/components/__init__.py:325-370).

    python tests/golden/make_xyz_golden.py       (build container only)   ->  tests/golden/xyz.npz (inputs + the files' text)
"""
import glob
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402


def main():
    _, _, comps = rh.import_reference()
    info = rh.dataset_info("qm9")
    g = torch.Generator().manual_seed(11)
    nn_ = torch.tensor([3, 1, 5, 2])
    N = int(nn_.sum())
    pos = torch.randn((N, 3), generator=g) * torch.tensor([1.0, 10.0, 1e-4])
    pos[0, 0], pos[1, 1] = 0.0, -0.0
    types = torch.randint(0, 5, (N,), generator=g)
    one_hot = torch.nn.functional.one_hot(types, 5).float()
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
    out = dict(num_nodes=nn_.numpy(), pos=pos.numpy(), one_hot=one_hot.numpy())
    with tempfile.TemporaryDirectory() as d:
        comps.save_xyz_file(d + "/", pos, one_hot, torch.zeros(0), info, id_from=7, name="mol", batch_index=bi)
        files = sorted(glob.glob(d + "/*.xyz"))
        out["names"] = np.array([os.path.basename(f) for f in files])
        out["texts"] = np.array([open(f).read() for f in files])
        comps.write_xyz_file(pos[:3], types[:3], d + "/single.xyz")
        out["single"] = np.array(open(d + "/single.xyz").read())
    np.savez_compressed(os.path.join(HERE, "xyz.npz"), **out)
    print(out["names"], repr(str(out["texts"][0])), repr(str(out["single"])))


if __name__ == "__main__":
    main()
