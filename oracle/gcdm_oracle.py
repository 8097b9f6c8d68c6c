"""CPU oracle for the GCDM denoising inner loop -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch (CPU, fp32 or fp64) restatement of the reference algorithm for the one hot path this
repository accelerates: ``GCPNetDynamics.forward`` evaluated at each DDPM step of
``EquivariantVariationalDiffusion.mol_gen_sample``.  It follows the reference's own formulation (explicit
edge lists, gather / index_add) so that it can be compared with the reference function by function, and
it consumes a ``state_dict`` with the *reference's* parameter names.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file,
and only as the checker / reported baseline.  The product package (``bio-diffusion_amd``) never imports it.

Parity pin: the reference has no numeric tests for this path, so this oracle is
pinned against golden vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``)
and, when ``/root/reference`` is present, directly against the imported reference
(``tests/test_oracle_vs_reference.py``).

All "file:line" citations are relative to the reference checkout (``/root/reference``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ------------------------------------------------------------------------------------------------
# configuration
# ------------------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    """The handful of hyper-parameters the hot path depends on (production values as defaults).

    Sources: configs/model/module_cfg/*_gcp_module.yaml, configs/model/diffusion_cfg/*.yaml,
    configs/datamodule/dataloader_cfg/edm_*_dataloader.yaml.
    """

    num_atom_types: int = 5
    include_charges: bool = True
    num_context: int = 0                 # len(module_cfg.conditioning)
    num_layers: int = 9                  # model_cfg.num_encoder_layers
    bottleneck: int = 4                  # module_cfg.bottleneck == default_bottleneck
    num_message_layers: int = 4          # layer_cfg.mp_cfg.num_message_layers
    node_positions_weight: float = 1.0
    condition_on_time: bool = True
    num_timesteps: int = 1000
    noise_schedule: str = "polynomial_2"
    noise_precision: float = 1e-5
    norm_values: Sequence[float] = (1.0, 4.0, 10.0)
    norm_biases: Sequence[Optional[float]] = (None, 0.0, 0.0)
    self_condition: bool = False         # diffusion_cfg.self_condition (False in both production configs)
    # ---- the rest of the path's Hydra surface (production values as defaults; module_cfg / layer_cfg / mp_cfg keys of the same names) ----
    selected_GCP: str = "GCP2"           # "GCP" = the first-generation module (gcpnet.py:33-262)
    nonlinearities: Sequence[Optional[str]] = ("silu", "silu")
    vector_gate: bool = True
    frame_gate: bool = False
    sigma_frame_gate: bool = False
    vector_residual: bool = False
    vector_frame_residual: bool = False
    ablate_frame_updates: bool = False
    use_residual_message_gcp: bool = True
    use_scalar_message_attention: bool = True
    num_feedforward_layers: int = 1
    use_gcp_norm: bool = False
    pre_norm: bool = False               # layer_cfg.pre_norm (GCPEmbedding.pre_norm is always True, gcpnet.py:504)
    update_positions_with_vector_sum: bool = False

    @property
    def num_node_scalar_features(self) -> int:
        return self.num_atom_types + int(self.include_charges)


# ------------------------------------------------------------------------------------------------
# graph + geometry helpers
# ------------------------------------------------------------------------------------------------
def num_nodes_to_batch_index(num_nodes: Tensor) -> Tensor:
    """src/models/components/__init__.py:314-321."""
    return torch.repeat_interleave(torch.arange(len(num_nodes)), num_nodes)


def fully_connected_edges(batch_index: Tensor, mask: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """gcpnet.py:1054-1066 -- all (i, j) with batch[i] == batch[j] INCLUDING i == j, sorted by (i, j).

    Built from per-molecule offsets instead of the reference's N x N boolean matrix (same result:
    ``batch_index`` is sorted, so each molecule is one contiguous block).
    """
    counts = torch.bincount(batch_index)
    counts = counts[counts > 0]
    offs = torch.cumsum(counts, 0) - counts
    rows, cols = [], []
    for o, n in zip(offs.tolist(), counts.tolist()):
        idx = torch.arange(o, o + n)
        rows.append(idx.repeat_interleave(n))
        cols.append(idx.repeat(n))
    row, col = torch.cat(rows), torch.cat(cols)
    if mask is not None and not bool(mask.all()):
        keep = mask[row] & mask[col]
        row, col = row[keep], col[keep]
    return row, col


def _normalize(t: Tensor, dim: int = -1) -> Tensor:
    """src/datamodules/components/helper.py:15-24."""
    return torch.nan_to_num(t / torch.norm(t, dim=dim, keepdim=True))


def orientations(x: Tensor) -> Tensor:
    """protein_graph_dataset.py:217-225 via edm_dataset.py:41-76 (``edm_sampling=True``).

    chi0[i] = [nrm(x[i+1]-x[i]), nrm(x[i-1]-x[i])] over the FLAT batch (crosses molecule boundaries,
    SURVEY A.6.2); zero-padded at the global ends; NaN -> 0.
    """
    fwd = _normalize(x[1:] - x[:-1])
    bwd = _normalize(x[:-1] - x[1:])
    fwd = F.pad(fwd, [0, 0, 0, 1])
    bwd = F.pad(bwd, [0, 0, 1, 0])
    return torch.nan_to_num(torch.stack((fwd, bwd), dim=-2))


def edge_features(x: Tensor, row: Tensor, col: Tensor) -> Tuple[Tensor, Tensor]:
    """edm_dataset.py:22-38: e = |x_i-x_j|^2 [E,1]; xi = nrm(x_i-x_j) [E,1,3]; NaN -> 0."""
    d = x[row] - x[col]
    e = torch.sum(d ** 2, dim=1, keepdim=True)
    xi = _normalize(d).unsqueeze(-2)
    return torch.nan_to_num(e), torch.nan_to_num(xi)


def centralize(x: Tensor, batch_index: Tensor, num_graphs: int, mask: Tensor) -> Tensor:
    """components/__init__.py:45-98, ``edm=True`` branch: subtract the per-molecule mean (sum / count)."""
    cnt = torch.zeros(num_graphs, dtype=x.dtype).index_add_(0, batch_index, mask.to(x.dtype)).unsqueeze(-1)
    s = torch.zeros(num_graphs, x.shape[1], dtype=x.dtype).index_add_(0, batch_index, x)
    cen = s / cnt
    return x - cen[batch_index] * mask.to(x.dtype).unsqueeze(-1)


def localize(x: Tensor, row: Tensor, col: Tensor) -> Tensor:
    """components/__init__.py:122-171 (norm_x_diff=True): rows [a; b; c],
    a = (x_i-x_j)/(|.|+1), b = (x_i x x_j)/(|.|+1), c = a x b."""
    xd = x[row] - x[col]
    xc = torch.linalg.cross(x[row], x[col], dim=-1)
    xd = xd / (torch.sqrt(torch.sum(xd ** 2, dim=1, keepdim=True)) + 1)
    xc = xc / (torch.sqrt(torch.sum(xc ** 2, dim=1, keepdim=True)) + 1)
    xv = torch.linalg.cross(xd, xc, dim=-1)
    return torch.stack((xd, xc, xv), dim=1)


def scalarize(u: Tensor, row: Tensor, frames: Tensor, node_inputs: bool, dim_size: int) -> Tensor:
    """components/__init__.py:174-219.  ``u`` is [M, 3ch, 3xyz]; q[3c+r] = f[r,:].u[c,:];
    node mode: scatter-MEAN over ``row`` (count clamped >= 1)."""
    ui = u[row] if node_inputs else u
    loc = torch.matmul(frames, ui.transpose(-1, -2)).transpose(-1, -2).reshape(ui.shape[0], 9)
    if not node_inputs:
        return loc
    out = torch.zeros(dim_size, 9, dtype=u.dtype).index_add_(0, row, loc)
    cnt = torch.zeros(dim_size, dtype=u.dtype).index_add_(0, row, torch.ones(row.shape[0], dtype=u.dtype))
    return out / cnt.clamp(min=1).unsqueeze(-1)


def safe_norm(x: Tensor, dim: int, eps: float = 1e-8) -> Tensor:
    """components/__init__.py:275-286 (eps added twice)."""
    return torch.sqrt(torch.sum(x ** 2, dim=dim) + eps) + eps


def _act(name: Optional[str]) -> Callable[[Tensor], Tensor]:
    """get_nonlinearity, src/models/__init__.py:30-45."""
    if name is None:
        return lambda t: t
    table = {"silu": F.silu, "relu": F.relu, "leakyrelu": lambda t: F.leaky_relu(t, negative_slope=1e-2), "selu": F.selu, "sigmoid": torch.sigmoid}
    if name.lower().strip() not in table:
        raise NotImplementedError(name)
    return table[name.lower().strip()]


def vectorize(gate: Tensor, row: Tensor, frames: Tensor, node_inputs: bool, dim_size: int) -> Tensor:
    """components/__init__.py:227-272: gate [M, 9] -> vectors [M, 3, 3], gv[k] = gate[3k] a + gate[3k+1] b + gate[3k+2] c with (a, b, c) the rows of
    the edge's frame; node mode: gate of the edge's source node, scatter-MEAN over ``row``."""
    g = (gate[row] if node_inputs else gate).reshape(-1, 3, 3)
    gv = torch.matmul(g, frames)                                # [E, k, xyz]
    if not node_inputs:
        return gv
    out = torch.zeros(dim_size, 3, 3, dtype=gate.dtype).index_add_(0, row, gv)
    cnt = torch.zeros(dim_size, dtype=gate.dtype).index_add_(0, row, torch.ones(row.shape[0], dtype=gate.dtype))
    return out / cnt.clamp(min=1).reshape(-1, 1, 1)


def gcp_layernorm(P: Params, pre: str, s: Tensor, v: Optional[Tensor], use_gcp_norm: bool, eps: float = 1e-8):
    """GCPLayerNorm, components/__init__.py:779-808: LayerNorm on the scalars, vectors / sqrt(mean_c clamp(|v_c|^2, eps)); identity without use_gcp_norm."""
    if not use_gcp_norm:
        return s, v
    s = F.layer_norm(s, (s.shape[-1],), P[pre + "scalar_norm.weight"], P[pre + "scalar_norm.bias"])
    if v is not None:
        vn = torch.clamp(torch.sum(v ** 2, dim=-1, keepdim=True), min=eps)
        v = v / torch.sqrt(torch.mean(vn, dim=-2, keepdim=True))
    return s, v


def gcp_any(P: Params, pre: str, s: Tensor, v: Tensor, row: Tensor, frames: Tensor, node_inputs: bool, acts: Sequence[Optional[str]],
            vector_out: bool, cfg: "OracleConfig", feedforward_out: bool = False, vector_residual: bool = False):
    """One GCP / GCP2 with the flags of ``cfg`` (module_cfg): GCP2.forward gcpnet.py:418-491 with process_vector_with_frames :378-408 /
    process_vector_without_frames :357-375; GCP.forward :190-262 with process_vector :121-140 and process_vector_frames :150-188."""
    a_s, a_v = _act(acts[0]), _act(acts[1])
    ident_v = acts[1] is None
    M = s.shape[0]
    lin = lambda key, t: F.linear(t, P[pre + key + ".weight"], P.get(pre + key + ".bias"))

    def scalar_out(merged):
        if feedforward_out:
            return lin("scalar_out.2", F.silu(lin("scalar_out.0", merged)))
        return lin("scalar_out", merged)

    def self_gate(vr):
        return vr if ident_v else vr * a_v(safe_norm(vr, dim=-1).unsqueeze(-1))

    def frame_gate(p, vr):
        gate = lin("vector_out_scale_frames", a_v(p))
        gvr = (vectorize(gate, row, frames, node_inputs, M).transpose(-1, -2) @ P[pre + "vector_up_frames.weight"].T).transpose(-1, -2)
        return vr * a_v(safe_norm(gvr, dim=-1).unsqueeze(-1))

    vt = v.transpose(-1, -2)                                    # [M, 3, V_in]
    vh = vt @ P[pre + "vector_down.weight"].T
    merged = torch.cat((s, safe_norm(vh, dim=-2)), dim=-1)
    if cfg.selected_GCP == "GCP2":
        if not cfg.ablate_frame_updates:
            u = (vt @ P[pre + "vector_down_frames.weight"].T).transpose(-1, -2)
            merged = torch.cat((merged, scalarize(u, row, frames, node_inputs, M)), dim=-1)
        p = scalar_out(merged)
        if not vector_out:
            return a_s(p)
        up = vh @ P[pre + "vector_up.weight"].T
        vr = ((up + vt) if vector_residual else up).transpose(-1, -2)
        if cfg.frame_gate and not cfg.ablate_frame_updates:
            vr = frame_gate(p, vr)
        elif cfg.vector_gate:
            vr = vr * torch.sigmoid(lin("vector_out_scale", a_v(p))).unsqueeze(-1)
        else:
            vr = self_gate(vr)
        return a_s(p), vr
    # ---- GCP (v1) ----
    p = scalar_out(merged)
    vr = v
    if vector_out:
        up = vh @ P[pre + "vector_up.weight"].T
        vr = ((up + vt) if vector_residual else up).transpose(-1, -2)
        vr = vr * torch.sigmoid(lin("vector_out_scale", a_v(p))).unsqueeze(-1) if cfg.vector_gate else self_gate(vr)
    p = a_s(p)
    if cfg.ablate_frame_updates:
        return (p, vr) if vector_out else p
    u = (vr.transpose(-1, -2) @ P[pre + "vector_down_frames.weight"].T).transpose(-1, -2)
    p = lin("scalar_out_frames", torch.cat((p, scalarize(u, row, frames, node_inputs, M)), dim=-1))
    if not vector_out:
        return a_s(p)
    if cfg.sigma_frame_gate:
        vr = vr * torch.sigmoid(lin("vector_out_scale_sigma_frames", a_v(p))).unsqueeze(-1)
    elif cfg.frame_gate:
        gated = frame_gate(p, vr)
        vr = gated + vr if cfg.vector_frame_residual else gated
    else:
        vr = self_gate(vr)
    return a_s(p), vr


# ------------------------------------------------------------------------------------------------
# GCP2
# ------------------------------------------------------------------------------------------------
def gcp2(P: Params, pre: str, s: Tensor, v: Tensor, row: Tensor, frames: Tensor, node_inputs: bool,
         act: Optional[str], vector_out: bool, feedforward_out: bool = False):
    """gcpnet.py:418-491 + process_vector_with_frames :378-415 in the production configuration
    (vector_gate=True, frame_gate=False, no residuals, no ablations).

    s [M,S_in], v [M,V_in,3] -> (act(p) [M,S_out], v' [M,V_out,3]) or act(p) if not ``vector_out``.
    Note: the vector gate sees ``act(p)`` *computed from the pre-activation p*.
    """
    a = _act(act)
    vt = v.transpose(-1, -2)                                   # [M,3,V_in]
    vh = vt @ P[pre + "vector_down.weight"].T                  # [M,3,H]
    merged = torch.cat((s, safe_norm(vh, dim=-2)), dim=-1)
    u = (vt @ P[pre + "vector_down_frames.weight"].T).transpose(-1, -2)  # [M,3ch,3xyz]
    merged = torch.cat((merged, scalarize(u, row, frames, node_inputs, u.shape[0])), dim=-1)
    if feedforward_out:   # Linear - SiLU - Linear (gcpnet.py:321-325; scalar_out_nonlinearity default "silu")
        p = F.linear(merged, P[pre + "scalar_out.0.weight"], P[pre + "scalar_out.0.bias"])
        p = F.linear(F.silu(p), P[pre + "scalar_out.2.weight"], P[pre + "scalar_out.2.bias"])
    else:
        p = F.linear(merged, P[pre + "scalar_out.weight"], P[pre + "scalar_out.bias"])
    if not vector_out:
        return a(p)
    vo = (vh @ P[pre + "vector_up.weight"].T).transpose(-1, -2)         # [M,V_out,3]
    gate = F.linear(a(p), P[pre + "vector_out_scale.weight"], P[pre + "vector_out_scale.bias"])
    vo = vo * torch.sigmoid(gate).unsqueeze(-1)
    return a(p), vo


# ------------------------------------------------------------------------------------------------
# message passing / interaction layer / dynamics
# ------------------------------------------------------------------------------------------------
def message_passing(P: Params, pre: str, h: Tensor, chi: Tensor, e: Tensor, xi: Tensor, row: Tensor,
                    col: Tensor, frames: Tensor, cfg: OracleConfig) -> Tuple[Tensor, Tensor]:
    """GCPMessagePassing.message/aggregate/forward, gcpnet.py:676-737 (residual message GCPs, scalar
    message attention, sum aggregation over ``row``)."""
    s = torch.cat((h[row], e, h[col]), dim=-1)
    v = torch.cat((chi[row], xi, chi[col]), dim=1)
    ms, mv = gcp2(P, pre + "message_fusion.0.", s, v, row, frames, False, "silu", True)
    for k in range(1, cfg.num_message_layers):
        ns, nv = gcp2(P, pre + f"message_fusion.{k}.", ms, mv, row, frames, False, "silu", True)
        ms, mv = ms + ns, mv + nv
    attn = torch.sigmoid(F.linear(ms, P[pre + "scalar_message_attention.0.weight"],
                                  P[pre + "scalar_message_attention.0.bias"]))
    ms = ms * attn
    flat = torch.cat((ms, mv.reshape(mv.shape[0], -1)), dim=-1)
    agg = torch.zeros(h.shape[0], flat.shape[1], dtype=h.dtype).index_add_(0, row, flat)
    V = chi.shape[1]
    return agg[:, : -3 * V], agg[:, -3 * V:].reshape(-1, V, 3)


def interaction_layer(P: Params, pre: str, h: Tensor, chi: Tensor, e: Tensor, xi: Tensor, row: Tensor,
                      col: Tensor, frames: Tensor, x: Tensor, maskf: Tensor, cfg: OracleConfig):
    """GCPInteractions.forward + derive_x_update, gcpnet.py:834-930 (pre_norm False, norms/dropout identity,
    one feed-forward GCP2 with feedforward_out, position update from a (S,1) GCP2)."""
    a_s, a_v = message_passing(P, pre + "interaction.", h, chi, e, xi, row, col, frames, cfg)
    hs = torch.cat((a_s, h), dim=-1)
    hv = torch.cat((a_v, chi), dim=1)
    fs, fv = gcp2(P, pre + "feedforward_network.0.", hs, hv, row, frames, True, None, True, feedforward_out=True)
    h = (h + fs) * maskf[:, None]
    chi = (chi + fv) * maskf[:, None, None]
    _, pv = gcp2(P, pre + "node_position_update_gcp.", h, chi, row, frames, True, "silu", True)
    x = (x + pv[:, 0, :] * cfg.node_positions_weight) * maskf[:, None]
    return h, chi, x


def dynamics_forward_general(P: Params, cfg: OracleConfig, xh: Tensor, t: Tensor, batch_index: Tensor, mask: Optional[Tensor] = None,
                             context: Optional[Tensor] = None) -> Tensor:
    """GCPNetDynamics.atom_types_and_coords_forward (gcpnet.py:1069-1232) for ANY setting of the module / layer / mp configuration groups that
    OracleConfig carries -- GCP or GCP2, gates, residuals, ablated frame updates, GCPLayerNorm, message / feed-forward depths, vector-sum
    position updates (GCPEmbedding :551-603, GCPMessagePassing :676-737, GCPInteractions :834-930).  ``dynamics_forward`` is the same
    function specialised to the production flags; pinned by tests/golden/dyn_variant_*.npz."""
    N = xh.shape[0]
    mask = torch.ones(N, dtype=torch.bool) if mask is None else mask
    maskf = mask.to(xh.dtype)
    B = int(batch_index.max().item()) + 1
    xh = xh * maskf[:, None]
    x0, h = xh[:, :3].clone(), xh[:, 3:].clone()
    row, col = fully_connected_edges(batch_index, mask)
    chi = orientations(x0)
    e, xi = edge_features(x0, row, col)
    if cfg.condition_on_time:
        h = torch.cat((h, t.view(N, 1)), dim=-1)
    if cfg.num_context:
        h = torch.cat((h, context.view(N, cfg.num_context)), dim=-1)
    x = centralize(x0, batch_index, B, mask)
    frames = localize(x, row, col)
    nl = tuple(cfg.nonlinearities)
    # GCPEmbedding (pre_norm True; its edge GCP keeps ("silu", "silu"), gcpnet.py:502, 1006-1014)
    e, xi = gcp_layernorm(P, "gcp_embedding.edge_normalization.", e, xi, cfg.use_gcp_norm)
    h, chi = gcp_layernorm(P, "gcp_embedding.node_normalization.", h, chi, cfg.use_gcp_norm)
    e, xi = gcp_any(P, "gcp_embedding.edge_embedding.", e, xi, row, frames, False, ("silu", "silu"), True, cfg)
    h, chi = gcp_any(P, "gcp_embedding.node_embedding.", h, chi, row, frames, True, (None, None), True, cfg)
    L = cfg.num_layers if cfg.num_layers else infer_num_layers(P)
    for l in range(L):
        pre = f"interaction_layers.{l}."
        if cfg.pre_norm:
            h, chi = gcp_layernorm(P, pre + "gcp_norm.0.", h, chi, cfg.use_gcp_norm)
        # message (:676-713): primary GCPs (first / last) use default_bottleneck and no residual, secondary ones module_cfg.vector_residual
        ms = torch.cat((h[row], e, h[col]), dim=-1)
        mv = torch.cat((chi[row], xi, chi[col]), dim=1)
        n_msg = cfg.num_message_layers
        for k in range(n_msg):
            res = cfg.vector_residual and 0 < k < n_msg - 1
            ns, nv = gcp_any(P, pre + f"interaction.message_fusion.{k}.", ms, mv, row, frames, False, nl, True, cfg, vector_residual=res)
            ms, mv = (ms + ns, mv + nv) if (cfg.use_residual_message_gcp and k > 0) else (ns, nv)
        if cfg.use_scalar_message_attention:
            ms = ms * torch.sigmoid(F.linear(ms, P[pre + "interaction.scalar_message_attention.0.weight"], P[pre + "interaction.scalar_message_attention.0.bias"]))
        flat = torch.cat((ms, mv.reshape(mv.shape[0], -1)), dim=-1)
        agg = torch.zeros(N, flat.shape[1], dtype=h.dtype).index_add_(0, row, flat)
        V = chi.shape[1]
        hs, hv = torch.cat((agg[:, : -3 * V], h), dim=-1), torch.cat((agg[:, -3 * V:].reshape(-1, V, 3), chi), dim=1)
        # feed-forward GCPs (:789-823): first (no residual; Linear-SiLU-Linear if it is the only one), middle (module_cfg flags), last (no residual, ff out)
        n_ff = cfg.num_feedforward_layers
        for k in range(n_ff):
            first, last = k == 0, k == n_ff - 1 and n_ff > 1
            acts = (None, None) if (first and n_ff == 1) or last else nl
            hs, hv = gcp_any(P, pre + f"feedforward_network.{k}.", hs, hv, row, frames, True, acts, True, cfg,
                             feedforward_out=(first and n_ff == 1) or last, vector_residual=cfg.vector_residual and not first and not last)
        h, chi = h + hs, chi + hv
        if not cfg.pre_norm:
            h, chi = gcp_layernorm(P, pre + "gcp_norm.0.", h, chi, cfg.use_gcp_norm)
        h, chi = h * maskf[:, None], chi * maskf[:, None, None]
        _, pv = gcp_any(P, pre + "node_position_update_gcp.", h, chi, row, frames, True, nl, True, cfg)
        upd = pv.sum(1) if cfg.update_positions_with_vector_sum else pv[:, 0, :]
        x = (x + upd * cfg.node_positions_weight) * maskf[:, None]
    hout = gcp_any(P, "scalar_node_projection_gcp.", h, chi, row, frames, True, (None, None), False, cfg)
    vel = (x - x0) * maskf[:, None]
    if cfg.num_context:
        hout = hout[:, : -cfg.num_context]
    if cfg.condition_on_time:
        hout = hout[:, :-1]
    if bool(vel.isnan().any()):
        vel = torch.zeros_like(vel)
    return torch.cat((centralize(vel, batch_index, B, mask), hout), dim=-1)


def infer_num_layers(P: Params) -> int:
    n = 0
    while f"interaction_layers.{n}.interaction.message_fusion.0.vector_down.weight" in P:
        n += 1
    return n


def dynamics_forward(P: Params, cfg: OracleConfig, xh: Tensor, t: Tensor, batch_index: Tensor,
                     mask: Optional[Tensor] = None, context: Optional[Tensor] = None,
                     return_intermediates: bool = False, xh_self_cond: Optional[Tensor] = None):
    """GCPNetDynamics.atom_types_and_coords_forward, gcpnet.py:1069-1232.

    xh [N,3+F], t [N,1], batch_index [N] sorted, context [N,C] or None  ->  net_out [N,3+F].
    With cfg.self_condition (:1112-1139) the previous estimate `xh_self_cond` (zeros if None) contributes its features, orientations and
    edge features to the embedding inputs: h -> [h | h_sc], chi -> 4 vectors, e -> 2 scalars, xi -> 2 vectors.
    """
    N = xh.shape[0]
    mask = torch.ones(N, dtype=torch.bool) if mask is None else mask
    maskf = mask.to(xh.dtype)
    B = int(batch_index.max().item()) + 1
    xh = xh * maskf[:, None]
    x0, h0 = xh[:, :3].clone(), xh[:, 3:].clone()
    row, col = fully_connected_edges(batch_index, mask)
    chi = orientations(x0)                                      # :1105
    e, xi = edge_features(x0, row, col)                         # :1109 (un-centralised x)
    h = h0
    if cfg.self_condition:                                      # :1112-1139 (x_self_cond is NOT masked / centralised)
        x_sc = xh_self_cond[:, :3].clone() if xh_self_cond is not None else torch.zeros_like(x0)
        h_sc = xh_self_cond[:, 3:].clone() if xh_self_cond is not None else torch.zeros_like(h0)
        e_sc, xi_sc = edge_features(x_sc, row, col)
        h = torch.cat((h, h_sc), dim=-1)
        chi = torch.cat((chi, orientations(x_sc)), dim=1)
        e = torch.cat((e, e_sc), dim=-1)
        xi = torch.cat((xi, xi_sc), dim=1)
    if cfg.condition_on_time:
        h = torch.cat((h, t.view(N, 1)), dim=-1)               # :1142-1150
    if cfg.num_context:
        h = torch.cat((h, context.view(N, cfg.num_context)), dim=-1)   # :1153-1155
    x = centralize(x0, batch_index, B, mask)                    # :1160
    frames = localize(x, row, col)                              # :1169
    # GCPEmbedding.forward :551-603 (pre-norms are identity: use_gcp_norm False)
    e, xi = gcp2(P, "gcp_embedding.edge_embedding.", e, xi, row, frames, False, "silu", True)
    h, chi = gcp2(P, "gcp_embedding.node_embedding.", h, chi, row, frames, True, None, True)
    inter = {}
    if return_intermediates:
        inter.update(row=row, col=col, frames=frames, e=e, xi=xi, h_embed=h, chi_embed=chi, x_central=x)
    L = cfg.num_layers if cfg.num_layers else infer_num_layers(P)
    for l in range(L):
        h, chi, x = interaction_layer(P, f"interaction_layers.{l}.", h, chi, e, xi, row, col, frames, x, maskf, cfg)
        if return_intermediates:
            inter[f"h_{l}"], inter[f"chi_{l}"], inter[f"x_{l}"] = h, chi, x
    hout = gcp2(P, "scalar_node_projection_gcp.", h, chi, row, frames, True, None, False)   # :1191
    vel = (x - x0) * maskf[:, None]                             # :1204
    if cfg.num_context:
        hout = hout[:, : -cfg.num_context]
    if cfg.condition_on_time:
        hout = hout[:, :-1]
    if bool(vel.isnan().any()):                                 # :1213-1216 (whole batch zeroed)
        vel = torch.zeros_like(vel)
    vel = centralize(vel, batch_index, B, mask)                 # :1220
    out = torch.cat((vel, hout), dim=-1)
    return (out, inter) if return_intermediates else out


# ------------------------------------------------------------------------------------------------
# noise schedule + sampler
# ------------------------------------------------------------------------------------------------
def gamma_table(cfg: OracleConfig) -> Tensor:
    """PredefinedNoiseSchedule ("polynomial_<p>"), variational_diffusion.py:67-107, 206-250.
    Computed in float64 numpy and rounded to fp32 exactly as the reference does."""
    T = cfg.num_timesteps
    power = float(cfg.noise_schedule.split("_")[1])
    steps = T + 1
    x = np.linspace(0, steps, steps)
    a2 = (1 - np.power(x / steps, power)) ** 2
    a2 = np.concatenate([np.ones(1), a2], axis=0)
    step = np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0)
    a2 = np.cumprod(step, axis=0)
    a2 = (1 - 2 * cfg.noise_precision) * a2 + cfg.noise_precision
    g = -(np.log(a2) - np.log(1 - a2))
    return torch.tensor(g).float()


def gamma_at(gam: Tensor, t: Tensor, T: int) -> Tensor:
    """variational_diffusion.py:252-255."""
    return gam[torch.round(t * T).long()]


def sigma_and_alpha_t_given_s(gt: Tensor, gs: Tensor):
    """variational_diffusion.py:342-367."""
    s2 = -torch.expm1(F.softplus(gs) - F.softplus(gt))
    a = torch.exp(0.5 * (F.logsigmoid(-gt) - F.logsigmoid(-gs)))
    return s2, torch.sqrt(s2), a


class TapeNoise:
    """Noise source reproducing the reference's ``torch.randn`` call order: per draw an x-part [N,3]
    then an h-part [N,F] (variational_diffusion.py:804-817), fp32 from a seeded CPU generator."""

    def __init__(self, seed: int = 1234):
        self.gen = torch.Generator().manual_seed(seed)

    def __call__(self, n: int, k: int, dtype=torch.float32) -> Tensor:
        return torch.randn((n, k), generator=self.gen, dtype=torch.float32).to(dtype)


def sample_combined_noise(noise, batch_index: Tensor, B: int, mask: Tensor, F_: int, dtype, fix_noise: bool = False) -> Tensor:
    """variational_diffusion.py:795-819: CoM-free x-noise, plain h-noise.  `fix_noise` (:832-834, 1323-1325): the reference passes an all-zero
    batch index, i.e. the x-noise is centred over the whole flat batch as if it were one molecule."""
    N = batch_index.shape[0]
    zx = noise(N, 3, dtype) * mask.to(dtype)[:, None]
    zx = centralize(zx, torch.zeros_like(batch_index), 1, mask) if fix_noise else centralize(zx, batch_index, B, mask)
    zh = noise(N, F_, dtype) * mask.to(dtype)[:, None]
    return torch.cat((zx, zh), dim=-1)


def sample_p_zs_given_zt(P: Params, cfg: OracleConfig, gam: Tensor, s: float, t: float, z: Tensor,
                         batch_index: Tensor, B: int, mask: Tensor, context: Optional[Tensor], noise,
                         eps_override: Optional[Tensor] = None, fix_noise: bool = False,
                         xh_self_cond: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """variational_diffusion.py:1204-1278.  Returns (z_s, eps_t)."""
    dt = z.dtype
    sv = torch.full((B, 1), s, dtype=dt)
    tv = torch.full((B, 1), t, dtype=dt)
    gs, gt = gamma_at(gam, sv, cfg.num_timesteps).to(dt), gamma_at(gam, tv, cfg.num_timesteps).to(dt)
    s2ts, sts, ats = sigma_and_alpha_t_given_s(gt, gs)
    sig_s, sig_t = torch.sqrt(torch.sigmoid(gs)), torch.sqrt(torch.sigmoid(gt))
    eps = (dynamics_forward(P, cfg, z, tv[batch_index], batch_index, mask, context, xh_self_cond=xh_self_cond)
           if eps_override is None else eps_override)
    mu = z / ats[batch_index] - (s2ts[batch_index] / ats[batch_index] / sig_t[batch_index]) * eps
    sigma = sts * sig_s / sig_t
    zs = mu + sigma[batch_index] * sample_combined_noise(noise, batch_index, B, mask, cfg.num_node_scalar_features, dt, fix_noise)
    zs = torch.cat((centralize(zs[:, :3], batch_index, B, mask), zs[:, 3:]), dim=-1)
    return zs, eps


def sample_p_xh_given_z0(P: Params, cfg: OracleConfig, gam: Tensor, z0: Tensor, batch_index: Tensor, B: int,
                         mask: Tensor, context: Optional[Tensor], noise, fix_noise: bool = False, xh_self_cond: Optional[Tensor] = None):
    """variational_diffusion.py:840-907 (+ unnormalize :735-757)."""
    dt = z0.dtype
    t0 = torch.zeros((B, 1), dtype=dt)
    g0 = gamma_at(gam, t0, cfg.num_timesteps).to(dt)
    sigma_x = torch.exp(-(-0.5 * g0))                            # SNR(-0.5*gamma_0) = exp(0.5*gamma_0)
    eps = dynamics_forward(P, cfg, z0, t0[batch_index], batch_index, mask, context, xh_self_cond=xh_self_cond)
    sig0, alp0 = torch.sqrt(torch.sigmoid(g0)), torch.sqrt(torch.sigmoid(-g0))
    mu = 1.0 / alp0[batch_index] * (z0 - sig0[batch_index] * eps)
    xh = mu + sigma_x[batch_index] * sample_combined_noise(noise, batch_index, B, mask, cfg.num_node_scalar_features, dt, fix_noise)
    x = xh[:, :3] * cfg.norm_values[0]
    mf = mask.to(dt)[:, None]
    if cfg.include_charges:
        h_cat, h_int = xh[:, 3:-1], xh[:, -1:]
    else:
        h_cat, h_int = xh[:, 3:], torch.zeros(0, dtype=dt)
    h_cat = (h_cat * cfg.norm_values[1] + cfg.norm_biases[1]) * mf
    h_int = h_int * cfg.norm_values[2] + cfg.norm_biases[2]
    if cfg.include_charges:
        h_int = h_int * mf
    one_hot = F.one_hot(torch.argmax(h_cat, dim=-1), cfg.num_atom_types) * mask.long()[:, None]
    charges = torch.round(h_int).long() * mask.long()[:, None] if cfg.include_charges else h_int.long()
    return x, one_hot, charges


def unnormalize_z(cfg: OracleConfig, z: Tensor, mask: Tensor) -> Tensor:
    """variational_diffusion.py:735-792 (unnormalize + unnormalize_z): continuous values, no argmax / rounding."""
    nv, nb = cfg.norm_values, cfg.norm_biases
    mf = mask.to(z.dtype)[:, None]
    F_ = cfg.num_atom_types
    parts = [z[:, :3] * nv[0], (z[:, 3:3 + F_] * nv[1] + nb[1]) * mf]
    if cfg.include_charges:
        parts.append((z[:, 3 + F_:] * nv[2] + nb[2]) * mf)
    return torch.cat(parts, dim=-1)


def mol_gen_sample(P: Params, cfg: OracleConfig, num_nodes: Tensor, noise, context: Optional[Tensor] = None,
                   num_timesteps: Optional[int] = None, dtype=torch.float32,
                   record: Optional[List[Tensor]] = None, return_frames: int = 1, fix_noise: bool = False) -> Tuple[Tensor, Tensor]:
    """EquivariantVariationalDiffusion.mol_gen_sample, variational_diffusion.py:1282-1412
    (no self-conditioning, fix_noise False).  Returns (out [N,3+F] or [return_frames,N,3+F], batch_index)."""
    T = cfg.num_timesteps if num_timesteps is None else num_timesteps
    assert 0 < return_frames <= T and T % return_frames == 0
    B = len(num_nodes)
    bi = num_nodes_to_batch_index(num_nodes)
    mask = torch.ones_like(bi).bool()
    ctx = None
    if context is not None:
        ctx = context.to(dtype)[bi] * mask.to(dtype)[:, None]
    gam = gamma_table(cfg)
    z = sample_combined_noise(noise, bi, B, mask, cfg.num_node_scalar_features, dtype, fix_noise)
    frames = torch.zeros((return_frames,) + tuple(z.shape), dtype=dtype)
    self_cond = None
    for s in reversed(range(T)):
        z, _ = sample_p_zs_given_zt(P, cfg, gam, s / T, (s + 1) / T, z, bi, B, mask, ctx, noise, fix_noise=fix_noise, xh_self_cond=self_cond)
        if record is not None:
            record.append(z.clone())
        if (s * return_frames) % T == 0:                          # :1354-1361
            frames[(s * return_frames) // T] = unnormalize_z(cfg, z, mask)
        if cfg.self_condition:                                    # :1363-1375: a jump from t = s/T to 0 (without self-conditioning input) is the next estimate
            self_cond, _ = sample_p_zs_given_zt(P, cfg, gam, 0.0, s / T, z, bi, B, mask, ctx, noise)
    x, one_hot, charges = sample_p_xh_given_z0(P, cfg, gam, z, bi, B, mask, ctx, noise, fix_noise, xh_self_cond=self_cond)
    if return_frames == 1:
        cog = torch.zeros(B, 3, dtype=dtype).index_add_(0, bi, x).abs().max().item()
        if cog > 5e-2:                                            # :1392-1402
            x = centralize(x, bi, B, mask)
    parts = [x, one_hot.to(dtype)] + ([charges.to(dtype)] if cfg.include_charges else [])
    frames[0] = torch.cat(parts, dim=-1)
    return (frames[0] if return_frames == 1 else frames), bi


def normalize(cfg: OracleConfig, x: Tensor, h_cat: Tensor, mask: Tensor) -> Tuple[Tensor, Tensor]:
    """EquivariantVariationalDiffusion.normalize, variational_diffusion.py:702-732 (x and the categorical part)."""
    nv, nb = cfg.norm_values, cfg.norm_biases
    return x / nv[0], (h_cat.to(x.dtype) - nb[1]) / nv[1] * mask.to(x.dtype)[:, None]


def assert_mean_zero_with_mask(x: Tensor, batch_index: Tensor, B: int, eps: float = 1e-10) -> float:
    """variational_diffusion.py:465-474: relative CoM error of the whole batch (the reference asserts < 1e-2)."""
    largest = x.abs().max().item()
    err = torch.zeros(B, x.shape[1], dtype=x.dtype).index_add_(0, batch_index, x).abs().max().item()
    return err / (largest + eps)



def get_repaint_schedule(resamplings: int, jump_length: int, num_timesteps: int) -> List[int]:
    """EquivariantVariationalDiffusion.get_repaint_schedule, variational_diffusion.py:1548-1578: how many denoising steps to apply before
    each jump back (RePaint), listed from t = T downwards.  Pinned to the reference's outputs in tests/golden/repaint_schedule.json."""
    t, out = 0, []
    while t < num_timesteps:
        step = jump_length if t + jump_length < num_timesteps else num_timesteps - t
        if step == jump_length and t + jump_length < num_timesteps:
            if out:
                out[-1] += step
                out.extend([step] * (resamplings - 1))
            else:
                out.extend([step] * resamplings)
        elif out:
            out[-1] += step
        else:
            out.append(step)
        t += step
    return out[::-1]


def mol_gen_optimize(P: Params, cfg: OracleConfig, x: Tensor, h_cat: Tensor, num_nodes: Tensor, noise,
                     context: Optional[Tensor] = None, num_timesteps: Optional[int] = None,
                     norm_with_original_timesteps: bool = False, dtype=torch.float32, return_frames: int = 1) -> Tuple[Tensor, Tensor]:
    """EquivariantVariationalDiffusion.mol_gen_optimize, variational_diffusion.py:1416-1546:
    the given samples are normalised and used as z at t = num_timesteps / T_norm, then denoised for num_timesteps steps and decoded.
    As in the reference the charge column is not part of z (`"integer": torch.tensor([])`, :1457), so include_charges must be False.
    return_frames > 1 (:1490-1497, 1526, 1540-1546): frame (s * return_frames) // T holds unnormalize_z of the latent after the step to s
    whenever (s * return_frames) % T == 0, frame 0 is overwritten by the decoded sample, and the CoG re-projection is skipped; -> [frames, N, 3 + F]."""
    assert not cfg.include_charges
    T = cfg.num_timesteps if num_timesteps is None else num_timesteps
    assert 0 < return_frames <= T and T % return_frames == 0
    Tn = cfg.num_timesteps if norm_with_original_timesteps else T
    B = len(num_nodes)
    bi = num_nodes_to_batch_index(num_nodes)
    mask = torch.ones_like(bi).bool()
    ctx = None
    if context is not None:
        ctx = context.to(dtype)[bi] * mask.to(dtype)[:, None]
    gam = gamma_table(cfg)
    xn, hn = normalize(cfg, x.to(dtype), h_cat, mask)
    z = torch.cat((xn, hn), dim=-1)
    assert assert_mean_zero_with_mask(z[:, :3], bi, B) < 1e-2                       # :1464
    self_cond = None
    frames = torch.zeros((return_frames,) + tuple(z.shape), dtype=dtype)
    for s in reversed(range(T)):
        z, _ = sample_p_zs_given_zt(P, cfg, gam, s / Tn, (s + 1) / Tn, z, bi, B, mask, ctx, noise, xh_self_cond=self_cond)
        if (s * return_frames) % T == 0:                                            # :1490-1497
            frames[(s * return_frames) // T] = unnormalize_z(cfg, z, mask)
        if cfg.self_condition:                                                      # :1500-1512
            self_cond, _ = sample_p_zs_given_zt(P, cfg, gam, 0.0, s / Tn, z, bi, B, mask, ctx, noise)
    xo, one_hot, charges = sample_p_xh_given_z0(P, cfg, gam, z, bi, B, mask, ctx, noise, xh_self_cond=self_cond)
    if return_frames == 1:
        cog = torch.zeros(B, 3, dtype=dtype).index_add_(0, bi, xo).abs().max().item()
        if cog > 5e-2:                                                              # :1527-1537
            xo = centralize(xo, bi, B, mask)
    frames[0] = torch.cat((xo, one_hot.to(dtype)), dim=-1)
    return (frames[0] if return_frames == 1 else frames), bi



def sample_p_zt_given_zs(cfg: OracleConfig, gam: Tensor, zs: Tensor, s: float, t: float, batch_index: Tensor, B: int, mask: Tensor,
                         noise) -> Tensor:
    """variational_diffusion.py:1163-1201: the forward (noising) jump z_s -> z_t, s < t, x re-projected to zero CoM.
    REPAIR: the reference line :1177 reads `alpha_t_given_s[node_mask] * zs`, a boolean-mask index of a [B,1] tensor with an [N] mask, which
    raises IndexError unless N == B; every other per-molecule factor in the file is gathered with `[batch_index]` (e.g. :929, :1252), and
    that is what is restated here."""
    dt = zs.dtype
    gs = gamma_at(gam, torch.full((B, 1), s, dtype=dt), cfg.num_timesteps).to(dt)
    gt = gamma_at(gam, torch.full((B, 1), t, dtype=dt), cfg.num_timesteps).to(dt)
    _, sts, ats = sigma_and_alpha_t_given_s(gt, gs)
    zt = ats[batch_index] * zs + sts[batch_index] * sample_combined_noise(noise, batch_index, B, mask, cfg.num_node_scalar_features, dt)
    return torch.cat((centralize(zt[:, :3], batch_index, B, mask), zt[:, 3:]), dim=-1)


def _fixed_mean(x: Tensor, fixed: Tensor, batch_index: Tensor, B: int) -> Tensor:
    """scatter(x[fixed], batch_index[fixed], reduce="mean") with one row per molecule (a molecule without fixed nodes gets 0).  The reference
    omits dim_size (:1626, :1688, :1694), so its result is shorter than B -- and the following `[batch_index]` gather raises -- whenever the
    last molecules have no fixed node; for inputs the reference can process the two agree."""
    sums = torch.zeros((B, x.shape[1]), dtype=x.dtype).index_add_(0, batch_index[fixed], x[fixed])
    cnt = torch.zeros(B, dtype=x.dtype).index_add_(0, batch_index[fixed], torch.ones(int(fixed.sum()), dtype=x.dtype))
    return sums / cnt.clamp(min=1)[:, None]


def inpaint(P: Params, cfg: OracleConfig, x: Tensor, h_cat: Tensor, h_int: Optional[Tensor], num_nodes: Tensor, fixed: Tensor, noise,
            num_resamplings: int = 1, jump_length: int = 1, return_frames: int = 1, num_timesteps: Optional[int] = None,
            context: Optional[Tensor] = None, dtype=torch.float32) -> Tensor:
    """EquivariantVariationalDiffusion.inpaint (RePaint), variational_diffusion.py:1582-1789.  The reference method cannot run as written:
      * :1650 divides by `num_denoise_steps` before the loop that defines it (UnboundLocalError on every call); the value is 0 / anything,
        i.e. the self-conditioning jump targets t = 0 as in mol_gen_sample (:1364) -- restated as 0;
      * :1177 (see sample_p_zt_given_zs above) raises IndexError on the first jump back.
    tests/golden/make_inpaint_golden.py runs the reference with exactly these two tokens repaired at run time; this restatement is pinned
    to that output and is "reference + two repairs", not the reference.  As in the reference the known molecule is used as given (it is not
    passed through `normalize`), only shifted so that its fixed nodes have zero CoM (:1625-1633)."""
    T = cfg.num_timesteps if num_timesteps is None else num_timesteps
    assert 0 < return_frames <= T and T % return_frames == 0
    assert jump_length == 1 or return_frames == 1
    B = len(num_nodes)
    bi = num_nodes_to_batch_index(num_nodes)
    mask = torch.ones_like(bi).bool()
    fixed = fixed.bool()
    ctx = None if context is None else context.to(dtype)[bi]
    parts = [x.to(dtype), h_cat.to(dtype)] + ([h_int.to(dtype)] if cfg.include_charges else [])
    xh0 = torch.cat(parts, dim=-1)
    xh0[:, :3] = xh0[:, :3] - _fixed_mean(x.to(dtype), fixed, bi, B)[bi]
    gam = gamma_table(cfg)
    Fd = cfg.num_node_scalar_features
    z = sample_combined_noise(noise, bi, B, mask, Fd, dtype)
    out = torch.zeros((return_frames,) + tuple(z.shape), dtype=dtype)
    schedule = get_repaint_schedule(num_resamplings, jump_length, T)
    s = T - 1
    self_cond = None
    fm = fixed.to(dtype)[:, None]
    for i, steps in enumerate(schedule):
        for j in range(steps):
            sn, tn = s / T, (s + 1) / T
            g_s = gamma_at(gam, torch.full((B, 1), sn, dtype=dtype), cfg.num_timesteps).to(dtype)
            # compute_noised_representation (:910-931)
            eps_known = sample_combined_noise(noise, bi, B, mask, Fd, dtype)
            z_known = torch.sqrt(torch.sigmoid(-g_s))[bi] * xh0 + torch.sqrt(torch.sigmoid(g_s))[bi] * eps_known
            z_unknown, _ = sample_p_zs_given_zt(P, cfg, gam, sn, tn, z, bi, B, mask, ctx, noise, xh_self_cond=self_cond)
            if cfg.self_condition:                                    # :1664-1676
                self_cond, _ = sample_p_zs_given_zt(P, cfg, gam, 0.0, sn, z_unknown, bi, B, mask, ctx, noise)
            shift = _fixed_mean(z_unknown[:, :3], fixed, bi, B) - _fixed_mean(z_known[:, :3], fixed, bi, B)   # :1680-1697
            z_known = torch.cat((z_known[:, :3] + shift[bi], z_known[:, 3:]), dim=-1)
            z = z_known * fm + z_unknown * (1 - fm)
            assert assert_mean_zero_with_mask(z[:, :3], bi, B) < 1e-2
            if (steps > jump_length or i == len(schedule) - 1) and (s * return_frames) % T == 0:     # :1707-1715
                out[(s * return_frames) // T] = unnormalize_z(cfg, z, mask)
            if j == steps - 1 and i < len(schedule) - 1:              # :1717-1737: jump back
                t_back = s + jump_length
                z = sample_p_zt_given_zs(cfg, gam, z, sn, t_back / T, bi, B, mask, noise)
                s = t_back
            s -= 1
    xo, one_hot, charges = sample_p_xh_given_z0(P, cfg, gam, z, bi, B, mask, ctx, noise, xh_self_cond=self_cond)
    if return_frames == 1:
        cog = torch.zeros(B, 3, dtype=dtype).index_add_(0, bi, xo).abs().max().item()
        if cog > 5e-2:
            xo = centralize(xo, bi, B, mask)
    parts = [xo, one_hot.to(dtype)] + ([charges.to(dtype)] if cfg.include_charges else [])
    out[0] = torch.cat(parts, dim=-1)
    return out[0] if return_frames == 1 else out

# ------------------------------------------------------------------------------------------------
# algorithmic FLOP count -- used by bench.py for the roofline line
# ------------------------------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# likelihood terms of a data batch in evaluation mode (validation / test NLL: two evaluations of the network)
# ------------------------------------------------------------------------------------------------
def _per_graph_sum(v: Tensor, batch_index: Tensor, B: int) -> Tensor:
    """sum_node_features_except_batch, variational_diffusion.py:449-454."""
    return torch.zeros(B, dtype=v.dtype).index_add_(0, batch_index, v.sum(-1))


def _gaussian_kl(mu2: Tensor, q_sigma: Tensor, d) -> Tensor:
    """gaussian_KL against N(0, 1), variational_diffusion.py:371-391."""
    return d * torch.log(1.0 / q_sigma) + 0.5 * (d * q_sigma ** 2 + mu2) - 0.5 * d


def _std_normal_cdf(x: Tensor) -> Tensor:
    """variational_diffusion.py:395-396."""
    return 0.5 * (1.0 + torch.erf(x / math.sqrt(2)))


def log_pxh_given_z0(cfg: OracleConfig, gam0: Tensor, h_cat_n: Tensor, h_int_n: Optional[Tensor], z0: Tensor, eps: Tensor, net_out: Tensor,
                     batch_index: Tensor, B: int, mask: Tensor, epsilon: float = 1e-10) -> Tuple[Tensor, Tensor]:
    """log_pxh_given_z0_without_constants, variational_diffusion.py:598-699.  `h_cat_n` / `h_int_n`: NORMALISED one-hot / integer features."""
    nv, nb = cfg.norm_values, cfg.norm_biases
    nt = cfg.num_atom_types
    m = mask.to(z0.dtype)[:, None]
    log_px = -0.5 * _per_graph_sum((eps[:, :3] - net_out[:, :3]) ** 2, batch_index, B)
    sigma0 = torch.sqrt(torch.sigmoid(gam0))[batch_index][:, None]
    # categorical part: probability mass of [0.5, 1.5] under N(estimate, sigma0 * norm), normalised over the classes (:672-684)
    est_cat = z0[:, 3:3 + nt] * nv[1] + nb[1] - 1.0
    s_cat = sigma0 * nv[1]
    lp = torch.log(_std_normal_cdf((est_cat + 0.5) / s_cat) - _std_normal_cdf((est_cat - 0.5) / s_cat) + epsilon)
    lp = lp - torch.logsumexp(lp, dim=-1, keepdim=True)
    onehot = h_cat_n * nv[1] + nb[1]
    log_ph = _per_graph_sum(lp * onehot * m, batch_index, B)
    if cfg.include_charges:              # integer part: mass of [-0.5, 0.5] around (true - estimate) (:655-668)
        h_int = torch.round(h_int_n.reshape(-1, 1) * nv[2] + nb[2]).long()
        centred = h_int - (z0[:, -1:] * nv[2] + nb[2])
        s_int = sigma0 * nv[2]
        li = torch.log(_std_normal_cdf((centred + 0.5) / s_int) - _std_normal_cdf((centred - 0.5) / s_int) + epsilon)
        log_ph = log_ph + _per_graph_sum(li * m, batch_index, B)
    return log_px, log_ph


def nll_terms(P: Params, cfg: OracleConfig, x: Tensor, one_hot: Tensor, charges: Optional[Tensor], num_nodes: Tensor, t_int: Tensor, noise,
              context: Optional[Tensor] = None, log_pN: Optional[Tensor] = None, dtype=torch.float32) -> Dict[str, Tensor]:
    """EquivariantVariationalDiffusion.atom_types_and_coords_forward in EVALUATION mode (variational_diffusion.py:955-1160; helpers :493-557,
    580-596, 702-732, 910-931, 943-945): the per-molecule terms of the variational bound.  `x` CoM-free raw positions, `one_hot` / `charges` raw
    features, `t_int` [B] the drawn timesteps (the reference draws them with torch.randint(1, T + 1)), `noise(n, k)` the standard-normal source in the
    reference's call order (x-part, h-part for z_t, then the same for z_0), `context` per MOLECULE."""
    B, T = len(num_nodes), cfg.num_timesteps
    bi = num_nodes_to_batch_index(num_nodes)
    mask = torch.ones(len(bi), dtype=torch.bool)
    nv, nb = cfg.norm_values, cfg.norm_biases
    gam = gamma_table(cfg).to(dtype)
    x = x.to(dtype) / nv[0]
    h_cat = (one_hot.to(dtype) - nb[1]) / nv[1]
    h_int = ((charges.to(dtype) - nb[2]) / nv[2]) if cfg.include_charges else None
    xh = torch.cat([x, h_cat] + ([h_int.reshape(-1, 1)] if cfg.include_charges else []), dim=-1)
    Fh = cfg.num_node_scalar_features
    dof = ((num_nodes - 1) * 3).to(dtype)
    out: Dict[str, Tensor] = {"delta_log_px": -dof * math.log(nv[0])}
    t = t_int.to(dtype) / T
    s = (t_int - 1).to(dtype) / T
    g_t, g_s = gamma_at(gam, t, T), gamma_at(gam, s, T)
    ctx = None if context is None else context.to(dtype)[bi]

    def noised(g):                        # compute_noised_representation, :910-931
        eps = sample_combined_noise(noise, bi, B, mask, Fh, dtype)
        a, sg = torch.sqrt(torch.sigmoid(-g))[bi][:, None], torch.sqrt(torch.sigmoid(g))[bi][:, None]
        return a * xh + sg * eps, eps

    z_t, eps_t = noised(g_t)
    net_t = dynamics_forward(P, cfg, z_t, t[bi][:, None], bi, None, ctx)
    out["error_t"] = _per_graph_sum((eps_t - net_t) ** 2, bi, B)
    out["SNR_weight"] = torch.exp(-(g_s - g_t)) - 1.0
    g0 = gamma_at(gam, torch.zeros(B, dtype=dtype), T)
    out["neg_log_constants"] = -(dof * (-0.5 * g0 - 0.5 * math.log(2 * math.pi)))
    # KL(q(z_T | x) || N(0, 1)), :501-557
    g_T = gamma_at(gam, torch.ones(B, dtype=dtype), T)
    mu_T = torch.sqrt(torch.sigmoid(-g_T))[bi][:, None] * xh
    sig_T = torch.sqrt(torch.sigmoid(g_T))
    out["kl_prior"] = _gaussian_kl(_per_graph_sum(mu_T[:, :3] ** 2, bi, B), sig_T, dof) + _gaussian_kl(_per_graph_sum(mu_T[:, 3:] ** 2, bi, B), sig_T, 1)
    # L_0 from a separate draw at t = 0 (:1107-1128)
    z_0, eps_0 = noised(g0)
    net_0 = dynamics_forward(P, cfg, z_0, torch.zeros(len(bi), 1, dtype=dtype), bi, None, ctx)
    lx, lh = log_pxh_given_z0(cfg, g0, h_cat, h_int, z_0, eps_0, net_0, bi, B, mask)
    out["loss_0_x"], out["loss_0_h"] = -lx, -lh
    if log_pN is not None:
        out["log_pN"] = log_pN.to(dtype)
    out["eps_hat_x"] = (torch.zeros(B, dtype=dtype).index_add_(0, bi, net_t[:, :3].abs().mean(-1)) / num_nodes.to(dtype)).mean()
    out["eps_hat_h"] = (torch.zeros(B, dtype=dtype).index_add_(0, bi, net_t[:, 3:].abs().mean(-1)) / num_nodes.to(dtype)).mean()
    return out


def nll_from_terms(terms: Dict[str, Tensor], T: int) -> Tensor:
    """The evaluation branch of the module's forward, qm9_mol_gen_ddpm.py:246-262: NLL per molecule from the terms above."""
    loss_t = T * 0.5 * terms["SNR_weight"] * terms["error_t"]
    loss_0 = terms["loss_0_x"] + terms["loss_0_h"] + terms["neg_log_constants"]
    return loss_t + loss_0 + terms["kl_prior"] - terms["delta_log_px"] - terms["log_pN"]


def _gcp2_flops(M, S_in, V_in, S_out, V_out, bn, ff=False):
    H = V_in // bn if bn > 1 else max(V_in, V_out)
    f = 3 * V_in * H + 9 * V_in + 27 + (S_in + H + 9) * S_out
    if ff:
        f += S_out * S_out
    if V_out:
        f += 3 * H * V_out + S_out * V_out
    return 2 * M * f


def forward_flops(N: int, E: int, S: int, V: int, Se: int, Ve: int, L: int, h_in: int) -> int:
    per_layer = (_gcp2_flops(E, 2 * S + Se, 2 * V + Ve, S, V, 4) + 3 * _gcp2_flops(E, S, V, S, V, 4) + 2 * E * S
                 + _gcp2_flops(N, 2 * S, 2 * V, S, V, 4, ff=True) + _gcp2_flops(N, S, V, S, 1, 4))
    return (_gcp2_flops(E, 1, 1, Se, Ve, 1) + _gcp2_flops(N, h_in, 2, S, V, 1) + L * per_layer
            + _gcp2_flops(N, S, V, h_in, 0, 1))
