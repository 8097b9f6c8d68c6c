"""Deterministic synthetic weights / inputs shared by the tests, the golden-fixture generator and smoke().

Weights are drawn per state-dict key from a numpy PCG64 stream seeded by (seed, crc32(key)) so that the
same tensors can be re-created anywhere (build container, GPU box) from the key->shape map alone; the
full-width golden fixtures therefore only need to store inputs and expected outputs.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch


def gcp2_shapes(pre: str, s_in: int, v_in: int, s_out: int, v_out: int, bottleneck: int, ff: bool = False):
    """Parameter names/shapes of one reference GCP2 (gcpnet.py:286-342) in the production configuration."""
    H = v_in // bottleneck if bottleneck > 1 else max(v_in, v_out)
    sh = {pre + "vector_down.weight": (H, v_in)}
    k = s_in + H + 9
    if ff:
        sh[pre + "scalar_out.0.weight"] = (s_out, k)
        sh[pre + "scalar_out.0.bias"] = (s_out,)
        sh[pre + "scalar_out.2.weight"] = (s_out, s_out)
        sh[pre + "scalar_out.2.bias"] = (s_out,)
    else:
        sh[pre + "scalar_out.weight"] = (s_out, k)
        sh[pre + "scalar_out.bias"] = (s_out,)
    sh[pre + "vector_down_frames.weight"] = (3, v_in)
    if v_out:
        sh[pre + "vector_up.weight"] = (v_out, H)
        sh[pre + "vector_out_scale.weight"] = (v_out, s_out)
        sh[pre + "vector_out_scale.bias"] = (v_out,)
    return sh


def dynamics_shapes(S=256, V=32, Se=64, Ve=16, L=9, h_in=7, bottleneck=4, self_cond_feats: int = 0) -> Dict[str, Tuple[int, ...]]:
    """All state-dict keys of GCPNetDynamics, in the reference's registration order.  ``self_cond_feats`` = F (number of
    diffused node scalars) when diffusion_cfg.self_condition is on: the embeddings then take [h | h_sc | t | ctx], 4 node vectors,
    2 edge scalars and 2 edge vectors (gcpnet.py:955-975); the projection keeps h_in outputs (:1027)."""
    sh: Dict[str, Tuple[int, ...]] = {}
    sc = self_cond_feats > 0
    sh.update(gcp2_shapes("gcp_embedding.edge_embedding.", 2 if sc else 1, 2 if sc else 1, Se, Ve, 1))
    sh.update(gcp2_shapes("gcp_embedding.node_embedding.", h_in + self_cond_feats, 4 if sc else 2, S, V, 1))
    for l in range(L):
        p = f"interaction_layers.{l}."
        sh.update(gcp2_shapes(p + "interaction.message_fusion.0.", 2 * S + Se, 2 * V + Ve, S, V, bottleneck))
        for k in (1, 2, 3):
            sh.update(gcp2_shapes(p + f"interaction.message_fusion.{k}.", S, V, S, V, bottleneck))
        sh[p + "interaction.scalar_message_attention.0.weight"] = (1, S)
        sh[p + "interaction.scalar_message_attention.0.bias"] = (1,)
        sh.update(gcp2_shapes(p + "feedforward_network.0.", 2 * S, 2 * V, S, V, bottleneck, ff=True))
        sh.update(gcp2_shapes(p + "node_position_update_gcp.", S, V, S, 1, bottleneck))
    sh.update(gcp2_shapes("scalar_node_projection_gcp.", S, V, h_in, 0, 1))
    return sh


def make_weights(shapes: Dict[str, Tuple[int, ...]], seed: int = 0, scale_2d: float = 1.0) -> Dict[str, torch.Tensor]:
    """U(-1/sqrt(fan_in), 1/sqrt(fan_in)) per tensor (PyTorch-Linear-like magnitudes); 2-D tensors x ``scale_2d``."""
    out = {}
    for key, shp in shapes.items():
        rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(key.encode())]))
        if len(shp) == 2:
            bound = 1.0 / np.sqrt(shp[1])
            w = rng.uniform(-bound, bound, size=shp) * scale_2d
        else:
            wkey = key.replace(".bias", ".weight")
            fan_in = shapes[wkey][1] if wkey in shapes and len(shapes[wkey]) == 2 else shp[0]
            bound = 1.0 / np.sqrt(fan_in)
            w = rng.uniform(-bound, bound, size=shp)
            if len(shp) == 1 and key.endswith(".weight"):        # LayerNorm scale (GCPLayerNorm.scalar_norm): around 1
                w = 1.0 + w
        out[key] = torch.tensor(w, dtype=torch.float32)
    return out


def make_inputs(num_nodes: Sequence[int], n_feat: int, seed: int = 1, t_value: float = 0.37,
                n_ctx: int = 0):
    """xh [N,3+F] with a CoM-free x-part, t [N,1], batch_index, context [N,C] or None."""
    g = torch.Generator().manual_seed(seed)
    nn_ = torch.tensor(list(num_nodes), dtype=torch.long)
    bi = torch.repeat_interleave(torch.arange(len(nn_)), nn_)
    N = int(nn_.sum())
    xh = torch.randn((N, 3 + n_feat), generator=g, dtype=torch.float32)
    for b in range(len(nn_)):
        sel = bi == b
        xh[sel, :3] -= xh[sel, :3].mean(0, keepdim=True)
    t = torch.full((N, 1), t_value, dtype=torch.float32)
    ctx = None
    if n_ctx:
        ctx = torch.randn((len(nn_), n_ctx), generator=g, dtype=torch.float32)[bi]
    return xh, t, bi, nn_, ctx


DATASET_DIMS = {
    # name: (S, V, Se, Ve, L, num_atom_types, include_charges, n_ctx, norm_values)
    "qm9": dict(S=256, V=32, Se=64, Ve=16, L=9, num_atom_types=5, include_charges=True, n_ctx=0,
                norm_values=(1.0, 4.0, 10.0)),
    "qm9cond": dict(S=256, V=32, Se=64, Ve=16, L=9, num_atom_types=5, include_charges=False, n_ctx=1,
                    norm_values=(1.0, 8.0, 1.0)),
    "geom": dict(S=256, V=32, Se=16, Ve=8, L=4, num_atom_types=16, include_charges=False, n_ctx=0,
                 norm_values=(1.0, 4.0, 10.0)),
}


def dims_h_in(d) -> int:
    return d["num_atom_types"] + int(d["include_charges"]) + 1 + d["n_ctx"]


def dims_feat(d) -> int:
    return d["num_atom_types"] + int(d["include_charges"])


# ---- non-production configurations of the path's Hydra surface (module path; tests/golden/make_variant_golden.py) ---------------------------------
# group -> {key: value}; "mp_cfg" is layer_cfg.mp_cfg.  selected_GCP is given by NAME and resolved by the user (the reference's class / ours).
VARIANTS = {
    "gcp1": dict(module_cfg=dict(selected_GCP="GCP")),
    "gcp1_frame_gate": dict(module_cfg=dict(selected_GCP="GCP", frame_gate=True, vector_frame_residual=True)),
    "gcp1_sigma_gate": dict(module_cfg=dict(selected_GCP="GCP", sigma_frame_gate=True, vector_gate=False)),
    "frame_gate": dict(module_cfg=dict(frame_gate=True)),
    "no_vector_gate": dict(module_cfg=dict(vector_gate=False, vector_residual=True)),
    "ablate_frames": dict(module_cfg=dict(ablate_frame_updates=True)),
    "gcp_norm": dict(layer_cfg=dict(use_gcp_norm=True, pre_norm=True)),
    "gcp_norm_post": dict(layer_cfg=dict(use_gcp_norm=True, pre_norm=False)),
    "vector_sum": dict(module_cfg=dict(update_positions_with_vector_sum=True, node_positions_weight=0.5)),
    "two_messages": dict(mp_cfg=dict(num_message_layers=2, use_residual_message_gcp=False), layer_cfg=dict(use_scalar_message_attention=False)),
    "three_ff": dict(layer_cfg=dict(num_feedforward_layers=3)),
    "relu_widths": dict(module_cfg=dict(scalar_nonlinearity="relu", vector_nonlinearity="leakyrelu", nonlinearities=["relu", "leakyrelu"], bottleneck=2,
                                        default_bottleneck=2),
                        model_cfg=dict(h_hidden_dim=48, chi_hidden_dim=12, e_hidden_dim=24, xi_hidden_dim=6, num_encoder_layers=3)),
}
SHRINK = dict(h_hidden_dim=32, chi_hidden_dim=8, e_hidden_dim=16, xi_hidden_dim=4, num_encoder_layers=2)      # ref_harness.shrink_cfgs


def apply_variant(cfgs, name: Optional[str], gcp_classes=None, shrink: bool = True):
    """Applies SHRINK and VARIANTS[name] in place to a dict of five config groups (attribute- or item-style containers both work)."""
    def put(group, key, value):
        try:
            group[key] = value
        except Exception:
            setattr(group, key, value)
    if shrink:
        for k, v in SHRINK.items():
            put(cfgs["model_cfg"], k, v)
    for grp, kv in (VARIANTS[name] if name else {}).items():
        target = cfgs["layer_cfg"]["mp_cfg"] if grp == "mp_cfg" else cfgs[grp]
        for k, v in kv.items():
            if k == "selected_GCP" and gcp_classes is not None:
                v = gcp_classes[v]
            put(target, k, v)
    return cfgs
