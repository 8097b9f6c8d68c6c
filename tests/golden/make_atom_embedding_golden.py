"""Golden fixture for the atom-type table of GCPEmbedding (num_atom_types > 0, gcpnet.py:509-512, 569-572), produced by the REFERENCE itself.
Run in the build container only (needs /root/reference; CPU):
    python tests/golden/make_atom_embedding_golden.py
Data only: inputs (integer atom types, orientations, edge features, frames), the module's weights, its outputs and the gradient of a
fixed scalar with respect to the table.
"""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [HERE, os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE))]
import ref_harness as rh  # noqa: E402

torch.set_num_threads(8)
gcp, vd, comps = rh.import_reference()
edm = importlib.import_module("src.datamodules.components.edm_dataset")
cfg = rh.to_dictconfig(rh.load_reference_cfgs("qm9", ())["module_cfg"])

torch.manual_seed(19)
nn_ = torch.tensor([5, 4, 3, 6])
bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
N = int(nn_.sum())
mask = torch.ones(N, dtype=torch.bool)
x = torch.randn(N, 3)
ei = gcp.GCPNetDynamics.get_fully_connected_edge_index(bi, mask)
b = rh.make_batch(bi, mask)
b.x = x
b.edge_index = ei
e, xi = edm._edge_features(b)
_, chi0 = edm._node_features(b, edm_sampling=True)
_, xc = comps.centralize(b, "x", bi, mask, edm=True)
fr = comps.localize(xc, ei, norm_x_diff=True, node_mask=mask)
T = 5
types = torch.randint(0, T, (N,))
mod = gcp.GCPEmbedding(comps.ScalarVector(1, 1), comps.ScalarVector(T, 2), comps.ScalarVector(8, 4), comps.ScalarVector(16, 4), num_atom_types=T,
                       cfg=cfg, pre_norm=False, use_gcp_norm=True)
mod.train()
b.h, b.chi, b.e, b.xi, b.f_ij = types, chi0, e, xi, fr
(ns, nv), (es, ev) = mod(b)
r = torch.randn_like(ns)
(ns * r).sum().backward()
out = dict(num_nodes=nn_, x=x, types=types, chi=chi0, e=e, xi=xi, frames=fr, edge_index=ei, node_s=ns, node_v=nv, edge_s=es, edge_v=ev, r=r,
           grad_table=mod.atom_embedding.weight.grad)
for k, w in mod.state_dict().items():
    out["w_" + k] = w
path = os.path.join(HERE, "fn_atom_embedding.npz")
np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
print("wrote", path, os.path.getsize(path) // 1024, "KiB")
