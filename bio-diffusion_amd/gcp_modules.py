"""The GCPNet modules with the reference's names, constructor keywords, state-dict keys and ``forward`` signatures -- callable.

Mirrors ``src/models/components/gcpnet.py``: ``GCP`` (:33-262), ``GCP2`` (:265-491), ``GCPEmbedding`` (:494-603), ``get_GCP_with_custom_cfg``
(:606-615), ``GCPMessagePassing`` (:618-737), ``GCPInteractions`` (:740-930) and ``GCPLayerNorm`` / ``GCPDropout`` of
``src/models/components/__init__.py`` (:745-808).  Each ``forward`` is a composition of the HIP operators in ``ops.py`` (libgcdm_ops.so):
every matrix product, norm, frame projection, gather and segment reduction is a kernel of this repository, forward and backward; what is
left to torch is tensor plumbing (views, ``cat``, residual adds, masks).  These modules are

  * plug point 3 -- ``module_cfg.selected_GCP(input_dims, output_dims, **flags)(s_maybe_v, edge_index, frames, node_inputs, node_mask)``;
  * the building blocks of ``GCPNetDynamics``' general path (any flag / width of the Hydra surface, training with autograd), next to the
    fused sampling kernels that evaluate the production configuration in one pass (gcpnet.py in this package decides which one runs).
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

import torch
from torch import nn

from . import ops
from .config import cfg_get

SV = Tuple[torch.Tensor, torch.Tensor]


def _is_identity(name) -> bool:
    return name is None or (isinstance(name, str) and name.lower() in ("none", "identity"))


def _entity_frames(edge_index: torch.Tensor, frames: torch.Tensor, node_inputs: bool, num_entities: int,
                   node_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """One frame per entity: the edge's own frame (edge inputs) or the mean of the frames of the node's edges (node inputs) -- scalarize and
    vectorize are linear in the frame, so the reference's scatter-mean of per-edge products (components/__init__.py:208-217, 256-270) is
    the product with the mean frame.  Edges with a masked end point contribute zeros (and still count), as there (:196-205)."""
    edge_mask = None
    if node_mask is not None and not bool(node_mask.all()):
        edge_mask = node_mask[edge_index[0]] & node_mask[edge_index[1]]
    if node_inputs:
        return ops.mean_frames(frames, ops.graph_of(edge_index, num_entities), edge_mask)
    f = frames.reshape(-1, 3, 3)
    return f if edge_mask is None else f * edge_mask.to(f.dtype).reshape(-1, 1, 1)


class _GCPBase(nn.Module):
    def _common(self, input_dims, output_dims, nonlinearities, scalar_gate, vector_gate, frame_gate, sigma_frame_gate, vector_residual,
                vector_frame_residual, ablate_frame_updates, ablate_scalars, ablate_vectors, bottleneck):
        if nonlinearities is None:
            nonlinearities = (None, None)
        self.scalar_input_dim, self.vector_input_dim = int(input_dims[0]), int(input_dims[1])
        self.scalar_output_dim, self.vector_output_dim = int(output_dims[0]), int(output_dims[1])
        self.nonlinearities = (nonlinearities[0], nonlinearities[1])
        self.scalar_gate, self.vector_gate, self.frame_gate, self.sigma_frame_gate = scalar_gate, bool(vector_gate), bool(frame_gate), bool(sigma_frame_gate)
        self.vector_residual, self.vector_frame_residual = bool(vector_residual), bool(vector_frame_residual)
        self.ablate_frame_updates, self.ablate_scalars, self.ablate_vectors = bool(ablate_frame_updates), bool(ablate_scalars), bool(ablate_vectors)
        self.bottleneck = int(bottleneck)
        if scalar_gate and scalar_gate > 0:
            self.norm = nn.LayerNorm(self.scalar_output_dim)          # declared by the reference (:73, :304), never applied in its forward
        if self.vector_input_dim:
            assert self.vector_input_dim % self.bottleneck == 0, \
                f"Input channel of vector ({self.vector_input_dim}) must be divisible with bottleneck factor ({self.bottleneck})"
            self.hidden_dim = (self.vector_input_dim // self.bottleneck if self.bottleneck > 1 else max(self.vector_input_dim, self.vector_output_dim))

    def _scalar_out(self, merged: torch.Tensor) -> torch.Tensor:
        if isinstance(self.scalar_out, nn.Sequential):
            h = ops.linear(merged, self.scalar_out[0].weight, self.scalar_out[0].bias)
            h = ops.act(h, self._ff_act)
            return ops.linear(h, self.scalar_out[2].weight, self.scalar_out[2].bias)
        return ops.linear(merged, self.scalar_out.weight, self.scalar_out.bias)

    def _make_scalar_out(self, in_dim: int, feedforward_out: bool, scalar_out_nonlinearity):
        self._ff_act = scalar_out_nonlinearity
        if feedforward_out:
            return nn.Sequential(nn.Linear(in_dim, self.scalar_output_dim), _act_module(scalar_out_nonlinearity),
                                 nn.Linear(self.scalar_output_dim, self.scalar_output_dim))
        return nn.Linear(in_dim, self.scalar_output_dim)

    def _inputs(self, s_maybe_v):
        s, v = s_maybe_v
        if self.ablate_scalars:
            s = torch.zeros_like(s)
        if self.ablate_vectors:
            v = torch.zeros_like(v)
        return s, v, v.transpose(-1, -2).contiguous()

    def _self_gate(self, vector_rep: torch.Tensor) -> torch.Tensor:
        """`vector_rep * nonlinearity(safe_norm(vector_rep))` -- the fall-back when no learnt gate is configured (:137-138, 405-406)."""
        if _is_identity(self.nonlinearities[1]):
            return vector_rep
        return ops.rowscale(vector_rep, ops.act(ops.safe_norm_rep(vector_rep), self.nonlinearities[1]))

    def _finish(self, s: torch.Tensor, v: torch.Tensor):
        s = ops.act(s, self.nonlinearities[0])
        if self.ablate_scalars:
            s = torch.zeros_like(s)
        if self.ablate_vectors:
            v = torch.zeros_like(v)
        return s, v


def _act_module(name) -> nn.Module:
    key = name.lower() if isinstance(name, str) else name
    table = {None: nn.Identity, "none": nn.Identity, "silu": nn.SiLU, "swish": nn.SiLU, "relu": nn.ReLU, "leakyrelu": nn.LeakyReLU, "selu": nn.SELU}
    if key not in table:
        raise NotImplementedError(f"nonlinearity {name!r}")
    return table[key]()          # parameter-free placeholder that keeps the Sequential's indices (state-dict keys `scalar_out.0`, `scalar_out.2`)


class GCP2(_GCPBase):
    """gcpnet.py:265-491."""

    def __init__(self, input_dims, output_dims, nonlinearities=("silu", "silu"), scalar_out_nonlinearity="silu", scalar_gate: int = 0,
                 vector_gate: bool = True, frame_gate: bool = False, sigma_frame_gate: bool = False, feedforward_out: bool = False,
                 bottleneck: int = 1, vector_residual: bool = False, vector_frame_residual: bool = False, ablate_frame_updates: bool = False,
                 ablate_scalars: bool = False, ablate_vectors: bool = False, scalarization_vectorization_output_dim: int = 3, **kwargs):
        super().__init__()
        self._common(input_dims, output_dims, nonlinearities, scalar_gate, vector_gate, frame_gate, sigma_frame_gate, vector_residual,
                     vector_frame_residual, ablate_frame_updates, ablate_scalars, ablate_vectors, bottleneck)
        self.feedforward_out = bool(feedforward_out)
        self.sv_dim = int(scalarization_vectorization_output_dim)
        if self.vector_input_dim:
            frame_dim = 0 if self.ablate_frame_updates else 3 * self.sv_dim
            self.vector_down = nn.Linear(self.vector_input_dim, self.hidden_dim, bias=False)
            self.scalar_out = self._make_scalar_out(self.hidden_dim + self.scalar_input_dim + frame_dim, feedforward_out, scalar_out_nonlinearity)
            if not self.ablate_frame_updates:
                self.vector_down_frames = nn.Linear(self.vector_input_dim, self.sv_dim, bias=False)
            if self.vector_output_dim:
                self.vector_up = nn.Linear(self.hidden_dim, self.vector_output_dim, bias=False)
                if not self.ablate_frame_updates and self.frame_gate:
                    self.vector_out_scale_frames = nn.Linear(self.scalar_output_dim, 3 * self.sv_dim)
                    self.vector_up_frames = nn.Linear(self.sv_dim, self.vector_output_dim, bias=False)
                elif self.vector_gate:
                    self.vector_out_scale = nn.Linear(self.scalar_output_dim, self.vector_output_dim)
        else:
            self.scalar_out = self._make_scalar_out(self.scalar_input_dim, feedforward_out, scalar_out_nonlinearity)

    def _vector_out(self, s_pre_act: torch.Tensor, v_pre: torch.Tensor, vh: torch.Tensor, F: Optional[torch.Tensor]) -> torch.Tensor:
        """process_vector_with_frames / process_vector_without_frames (:357-408)."""
        up = ops.linear(vh, self.vector_up.weight)                      # [M, 3, V_out]
        if self.vector_residual:
            up = up + v_pre
        vector_rep = up.transpose(-1, -2).contiguous()                   # [M, V_out, 3]
        gate_in = ops.act(s_pre_act, self.nonlinearities[1])
        if F is not None and self.frame_gate:
            gate = ops.linear(gate_in, self.vector_out_scale_frames.weight, self.vector_out_scale_frames.bias)
            gate_vector = ops.vectorize(gate, F)                         # [M, sv, 3]
            gvr = ops.linear(gate_vector.transpose(-1, -2).contiguous(), self.vector_up_frames.weight).transpose(-1, -2).contiguous()
            return ops.rowscale(vector_rep, ops.act(ops.safe_norm_rep(gvr), self.nonlinearities[1]))
        if self.vector_gate:
            gate = ops.linear(gate_in, self.vector_out_scale.weight, self.vector_out_scale.bias)
            return ops.rowscale(vector_rep, ops.act(gate, "sigmoid"))
        return self._self_gate(vector_rep)

    def forward(self, s_maybe_v, edge_index: torch.Tensor, frames: torch.Tensor, node_inputs: bool = False,
                node_mask: Optional[torch.Tensor] = None):
        F = None
        if self.vector_input_dim:
            s, v, v_pre = self._inputs(s_maybe_v)
            vh = ops.linear(v_pre, self.vector_down.weight)              # [M, 3, H]
            merged = torch.cat((s, ops.safe_norm_pre(vh)), dim=-1)
            if not self.ablate_frame_updates:
                F = _entity_frames(edge_index, frames, node_inputs, v.shape[0], node_mask)
                u = ops.linear(v_pre, self.vector_down_frames.weight)    # [M, 3, sv]
                merged = torch.cat((merged, ops.scalarize(u, F)), dim=-1)
        else:
            merged = s_maybe_v
        s_out = self._scalar_out(merged)
        if not self.vector_output_dim:
            if self.ablate_scalars:
                s_out = torch.zeros_like(s_out)
            return ops.act(s_out, self.nonlinearities[0])
        if not self.vector_input_dim:
            v_out = torch.zeros(s_out.shape[0], self.vector_output_dim, 3, device=s_out.device)
        else:
            v_out = self._vector_out(s_out, v_pre, vh, F)
        return self._finish(s_out, v_out)


class GCP(_GCPBase):
    """gcpnet.py:33-262 (the first-generation module: vectors are updated first, the frames enter through a second scalar mixing step)."""

    def __init__(self, input_dims, output_dims, nonlinearities=("silu", "silu"), scalar_out_nonlinearity="silu", scalar_gate: int = 0,
                 vector_gate: bool = True, frame_gate: bool = False, sigma_frame_gate: bool = False, feedforward_out: bool = False,
                 bottleneck: int = 1, vector_residual: bool = False, vector_frame_residual: bool = False, ablate_frame_updates: bool = False,
                 ablate_scalars: bool = False, ablate_vectors: bool = False, scalarization_vectorization_output_dim: int = 3, **kwargs):
        super().__init__()
        self._common(input_dims, output_dims, nonlinearities, scalar_gate, vector_gate, frame_gate, sigma_frame_gate, vector_residual,
                     vector_frame_residual, ablate_frame_updates, ablate_scalars, ablate_vectors, bottleneck)
        self.feedforward_out = bool(feedforward_out)
        self.sv_dim = int(scalarization_vectorization_output_dim)
        if self.vector_input_dim:
            self.vector_down = nn.Linear(self.vector_input_dim, self.hidden_dim, bias=False)
            self.scalar_out = self._make_scalar_out(self.hidden_dim + self.scalar_input_dim, feedforward_out, scalar_out_nonlinearity)
            if self.vector_output_dim:
                self.vector_up = nn.Linear(self.hidden_dim, self.vector_output_dim, bias=False)
                if self.vector_gate:
                    self.vector_out_scale = nn.Linear(self.scalar_output_dim, self.vector_output_dim)
            if not self.ablate_frame_updates:
                frames_in = self.hidden_dim if not self.vector_output_dim else self.vector_output_dim
                self.vector_down_frames = nn.Linear(frames_in, self.sv_dim, bias=False)
                self.scalar_out_frames = nn.Linear(self.scalar_output_dim + 3 * self.sv_dim, self.scalar_output_dim)
                if self.vector_output_dim and self.sigma_frame_gate:
                    self.vector_out_scale_sigma_frames = nn.Linear(self.scalar_output_dim, self.vector_output_dim)
                elif self.vector_output_dim and self.frame_gate:
                    self.vector_out_scale_frames = nn.Linear(self.scalar_output_dim, 3 * self.sv_dim)
                    self.vector_up_frames = nn.Linear(self.sv_dim, self.vector_output_dim, bias=False)
        else:
            self.scalar_out = self._make_scalar_out(self.scalar_input_dim, feedforward_out, scalar_out_nonlinearity)

    def forward(self, s_maybe_v, edge_index: torch.Tensor, frames: torch.Tensor, node_inputs: bool = False,
                node_mask: Optional[torch.Tensor] = None):
        vector_rep = None
        if self.vector_input_dim:
            s, v, v_pre = self._inputs(s_maybe_v)
            vector_rep = v
            vh = ops.linear(v_pre, self.vector_down.weight)
            merged = torch.cat((s, ops.safe_norm_pre(vh)), dim=-1)
        else:
            merged = torch.zeros_like(s_maybe_v) if self.ablate_scalars else s_maybe_v
        s_rep = self._scalar_out(merged)
        if self.vector_input_dim and self.vector_output_dim:          # process_vector (:121-140)
            up = ops.linear(vh, self.vector_up.weight)
            if self.vector_residual:
                up = up + v_pre
            vector_rep = up.transpose(-1, -2).contiguous()
            if self.vector_gate:
                gate = ops.linear(ops.act(s_rep, self.nonlinearities[1]), self.vector_out_scale.weight, self.vector_out_scale.bias)
                vector_rep = ops.rowscale(vector_rep, ops.act(gate, "sigmoid"))
            else:
                vector_rep = self._self_gate(vector_rep)
        s_rep = ops.act(s_rep, self.nonlinearities[0])
        if self.vector_output_dim and not self.vector_input_dim:
            vector_rep = torch.zeros(s_rep.shape[0], self.vector_output_dim, 3, device=s_rep.device)
        if self.ablate_frame_updates:
            return (s_rep, vector_rep) if self.vector_output_dim else s_rep
        if vector_rep is None:
            raise ValueError("GCP with frame updates needs vector-valued inputs or outputs (the reference fails here too: gcpnet.py:225)")
        # scalar features from the complete local frames (:224-238)
        v_pre2 = vector_rep.transpose(-1, -2).contiguous()
        F = _entity_frames(edge_index, frames, node_inputs, v_pre2.shape[0], node_mask)
        u = ops.linear(v_pre2, self.vector_down_frames.weight)
        merged = torch.cat((s_rep, ops.scalarize(u, F)), dim=-1)
        s_rep = ops.linear(merged, self.scalar_out_frames.weight, self.scalar_out_frames.bias)
        if not self.vector_output_dim:
            if self.ablate_scalars:
                s_rep = torch.zeros_like(s_rep)
            return ops.act(s_rep, self.nonlinearities[0])
        if self.vector_input_dim and self.vector_output_dim:          # process_vector_frames (:150-188), on the vectors just produced
            gate_in = ops.act(s_rep, self.nonlinearities[1])
            if self.sigma_frame_gate:
                gate = ops.linear(gate_in, self.vector_out_scale_sigma_frames.weight, self.vector_out_scale_sigma_frames.bias)
                vector_rep = ops.rowscale(vector_rep, ops.act(gate, "sigmoid"))
            elif self.frame_gate:
                gate = ops.linear(gate_in, self.vector_out_scale_frames.weight, self.vector_out_scale_frames.bias)
                gate_vector = ops.vectorize(gate, F)
                gvr = ops.linear(gate_vector.transpose(-1, -2).contiguous(), self.vector_up_frames.weight).transpose(-1, -2).contiguous()
                gated = ops.rowscale(vector_rep, ops.act(ops.safe_norm_rep(gvr), self.nonlinearities[1]))
                vector_rep = gated + vector_rep if self.vector_frame_residual else gated
            else:
                vector_rep = self._self_gate(vector_rep)
        return self._finish(s_rep, vector_rep)


GCP_VARIANTS = {"GCP": GCP, "GCP2": GCP2}


def selected_gcp_class(cfg) -> type:
    """module_cfg.selected_GCP: a class, a functools.partial / Hydra `_partial_` node of one, or its name ("GCP" | "GCP2", default GCP2)."""
    sel = cfg_get(cfg, "selected_GCP", None)
    if sel is None:
        return GCP2
    if isinstance(sel, type) and issubclass(sel, _GCPBase):
        return sel
    name = None
    if isinstance(sel, str):
        name = sel
    elif isinstance(sel, dict) or hasattr(sel, "get"):
        name = str(sel.get("_target_", ""))
    elif hasattr(sel, "func"):
        name = getattr(sel.func, "__name__", "")
    else:
        name = getattr(sel, "__name__", str(sel))
    name = name.rsplit(".", 1)[-1]
    if name not in GCP_VARIANTS:
        raise NotImplementedError(f"module_cfg.selected_GCP = {sel!r}: expected GCP or GCP2 (gcpnet.py:33, 265)")
    return GCP_VARIANTS[name]


_GCP_FLAG_KEYS = ("scalar_gate", "vector_gate", "frame_gate", "sigma_frame_gate", "vector_frame_residual", "ablate_frame_updates", "ablate_scalars",
                  "ablate_vectors")
_GCP_FLAG_DEFAULTS = dict(scalar_gate=0, vector_gate=True, frame_gate=False, sigma_frame_gate=False, vector_frame_residual=False,
                          ablate_frame_updates=False, ablate_scalars=False, ablate_vectors=False)


def get_GCP_with_custom_cfg(input_dims, output_dims, cfg, **kwargs):
    """gcpnet.py:606-615: every key of module_cfg is a constructor keyword (the constructors swallow the rest), `kwargs` override."""
    args = {k: cfg_get(cfg, k, d) for k, d in _GCP_FLAG_DEFAULTS.items()}
    args.update(nonlinearities=cfg_get(cfg, "nonlinearities"), bottleneck=cfg_get(cfg, "bottleneck", 1), vector_residual=cfg_get(cfg, "vector_residual", False))
    args.update(kwargs)
    return selected_gcp_class(cfg)(input_dims, output_dims, **args)


def _embedding_gcp(cfg, input_dims, output_dims, nonlinearities):
    """The embedding / projection GCPs are built with an explicit flag list (gcpnet.py:523-549, 1028-1039): bottleneck and vector_residual keep
    their constructor defaults (1, False)."""
    args = {k: cfg_get(cfg, k, d) for k, d in _GCP_FLAG_DEFAULTS.items()}
    return selected_gcp_class(cfg)(input_dims, output_dims, nonlinearities=nonlinearities, **args)


# ---- ScalarVector plumbing (components/__init__.py ScalarVector: idx / concat / flatten / recover / mask) -------------------------------------
def sv_flatten(s: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    return torch.cat((s, v.reshape(v.shape[0], -1)), dim=-1)


def sv_recover(x: torch.Tensor, vector_dim: int) -> SV:
    v = x[..., -3 * vector_dim:].reshape(x.shape[0], vector_dim, 3)
    return x[..., : -3 * vector_dim], v


class GCPLayerNorm(nn.Module):
    """components/__init__.py:779-808: LayerNorm on the scalars, vectors divided by the root of their mean squared norm over the channels.
    (O(N) element-wise work; runs as torch device ops around the HIP operators.)"""

    def __init__(self, dims, eps: float = 1e-8, use_gcp_norm: bool = True):
        super().__init__()
        self.scalar_dims, self.vector_dims = dims
        self.scalar_norm = nn.LayerNorm(self.scalar_dims) if use_gcp_norm else nn.Identity()
        self.use_gcp_norm = bool(use_gcp_norm)
        self.eps = eps

    @staticmethod
    def norm_vector(v: torch.Tensor, use_gcp_norm: bool = True, eps: float = 1e-8) -> torch.Tensor:
        if not use_gcp_norm:
            return v
        vn = torch.clamp(torch.sum(torch.square(v), dim=-1, keepdim=True), min=eps)
        return v / torch.sqrt(torch.mean(vn, dim=-2, keepdim=True))

    def forward(self, x):
        if isinstance(x, torch.Tensor):
            return x if x.shape[0] == 0 else self.scalar_norm(x)
        s, v = x
        if s.shape[0] == 0 or v.shape[0] == 0:
            return x
        return self.scalar_norm(s), self.norm_vector(v, use_gcp_norm=self.use_gcp_norm, eps=self.eps)


class GCPDropout(nn.Module):
    """components/__init__.py:733-776: element dropout on scalars, whole-vector dropout on vector channels; identity in eval mode."""

    def __init__(self, drop_rate: float, use_gcp_dropout: bool = True):
        super().__init__()
        self.drop_rate = float(drop_rate)
        self.use_gcp_dropout = bool(use_gcp_dropout)
        self.scalar_dropout = nn.Dropout(self.drop_rate) if use_gcp_dropout else nn.Identity()

    def _vector(self, v: torch.Tensor) -> torch.Tensor:
        if not self.use_gcp_dropout or not self.training:
            return v
        mask = torch.bernoulli((1 - self.drop_rate) * torch.ones(v.shape[:-1], device=v.device)).unsqueeze(-1)
        return mask * v / (1 - self.drop_rate)

    def forward(self, x):
        if isinstance(x, torch.Tensor):
            return x if x.shape[0] == 0 else self.scalar_dropout(x)
        s, v = x
        if s.shape[0] == 0 or v.shape[0] == 0:
            return x
        return self.scalar_dropout(s), self._vector(v)


class GCPEmbedding(nn.Module):
    """gcpnet.py:494-603."""

    def __init__(self, edge_input_dims, node_input_dims, edge_hidden_dims, node_hidden_dims, num_atom_types: int = 0,
                 nonlinearities=("silu", "silu"), cfg=None, pre_norm: bool = True, use_gcp_norm: bool = True):
        super().__init__()
        # integer atom types -> rows of a [num_atom_types, num_atom_types] table (gcpnet.py:540-549); GCPNetDynamics builds the embedding with
        # num_atom_types = 0 (the atom types arrive as one-hot floats, gcpnet.py:1011)
        self.atom_embedding = nn.Embedding(num_atom_types, num_atom_types) if num_atom_types > 0 else None
        self.pre_norm = bool(pre_norm)
        # parameter-free when use_gcp_norm is False (nn.Identity inside): the production state dict has no norm keys
        self.edge_normalization = GCPLayerNorm(edge_input_dims if pre_norm else edge_hidden_dims, use_gcp_norm=use_gcp_norm)
        self.node_normalization = GCPLayerNorm(node_input_dims if pre_norm else node_hidden_dims, use_gcp_norm=use_gcp_norm)
        self.edge_embedding = _embedding_gcp(cfg, edge_input_dims, edge_hidden_dims, nonlinearities)
        self.node_embedding = _embedding_gcp(cfg, node_input_dims, node_hidden_dims, (None, None))

    def forward(self, batch: Any):
        h = ops.embedding(self.atom_embedding.weight, batch.h) if self.atom_embedding is not None else batch.h      # gcpnet.py:569-572
        node_rep = (h, batch.chi)
        edge_rep = (batch.e, batch.xi)
        edge_rep = edge_rep[0] if not self.edge_embedding.vector_input_dim else edge_rep
        node_rep = node_rep[0] if not self.node_embedding.vector_input_dim else node_rep
        if self.pre_norm:
            edge_rep = self.edge_normalization(edge_rep)
            node_rep = self.node_normalization(node_rep)
        mask = getattr(batch, "mask", None)
        edge_rep = self.edge_embedding(edge_rep, batch.edge_index, batch.f_ij, node_inputs=False, node_mask=mask)
        node_rep = self.node_embedding(node_rep, batch.edge_index, batch.f_ij, node_inputs=True, node_mask=mask)
        if not self.pre_norm:
            edge_rep = self.edge_normalization(edge_rep)
            node_rep = self.node_normalization(node_rep)
        return node_rep, edge_rep


class GCPMessagePassing(nn.Module):
    """gcpnet.py:618-737."""

    def __init__(self, input_dims, output_dims, edge_dims, cfg, mp_cfg, reduce_function: str = "sum", use_scalar_message_attention: bool = True):
        super().__init__()
        self.scalar_input_dim, self.vector_input_dim = input_dims
        self.scalar_output_dim, self.vector_output_dim = output_dims
        self.edge_scalar_dim, self.edge_vector_dim = edge_dims
        self.self_message = cfg_get(mp_cfg, "self_message", True)
        self.reduce_function = reduce_function
        self.use_residual_message_gcp = bool(cfg_get(mp_cfg, "use_residual_message_gcp", True))
        self.use_scalar_message_attention = bool(use_scalar_message_attention)
        n_msg = int(cfg_get(mp_cfg, "num_message_layers", 4))
        s_in, v_in = 2 * self.scalar_input_dim + self.edge_scalar_dim, 2 * self.vector_input_dim + self.edge_vector_dim
        soft = dict(bottleneck=cfg_get(cfg, "default_bottleneck", 4), vector_residual=cfg_get(cfg, "default_vector_residual", False))
        nl = cfg_get(cfg, "nonlinearities")
        mods = [get_GCP_with_custom_cfg((s_in, v_in), output_dims, cfg, nonlinearities=nl, **soft)]
        for _ in range(n_msg - 2):
            mods.append(get_GCP_with_custom_cfg(output_dims, output_dims, cfg))
        if n_msg > 1:
            mods.append(get_GCP_with_custom_cfg(output_dims, output_dims, cfg, nonlinearities=nl, **soft))
        self.message_fusion = nn.ModuleList(mods)
        if self.use_scalar_message_attention:
            self.scalar_message_attention = nn.Sequential(nn.Linear(output_dims[0], 1), nn.Sigmoid())

    def message(self, node_rep: SV, edge_rep: SV, edge_index: torch.Tensor, frames: torch.Tensor, node_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        g = ops.graph_of(edge_index, node_rep[0].shape[0])
        s_row, s_col = ops.gather_row(node_rep[0], g), ops.gather_col(node_rep[0], g)
        v_row, v_col = ops.gather_row(node_rep[1], g), ops.gather_col(node_rep[1], g)
        ms = torch.cat((s_row, edge_rep[0], s_col), dim=-1)
        mv = torch.cat((v_row, edge_rep[1], v_col), dim=-2)
        if self.use_residual_message_gcp:
            ms, mv = self.message_fusion[0]((ms, mv), edge_index, frames, node_inputs=False, node_mask=node_mask)
            for module in self.message_fusion[1:]:
                ns, nv = module((ms, mv), edge_index, frames, node_inputs=False, node_mask=node_mask)
                ms, mv = ms + ns, mv + nv
        else:
            for module in self.message_fusion:
                ms, mv = module((ms, mv), edge_index, frames, node_inputs=False, node_mask=node_mask)
        if self.use_scalar_message_attention:
            lin = self.scalar_message_attention[0]
            ms = ms * ops.act(ops.linear(ms, lin.weight, lin.bias), "sigmoid")
        return sv_flatten(ms, mv)

    def aggregate(self, message: torch.Tensor, edge_index: torch.Tensor, dim_size: int) -> torch.Tensor:
        return ops.scatter_rows(message, ops.graph_of(edge_index, dim_size), self.reduce_function)

    def forward(self, node_rep: SV, edge_rep: SV, edge_index: torch.Tensor, frames: torch.Tensor, node_mask: Optional[torch.Tensor] = None) -> SV:
        message = self.message(node_rep, edge_rep, edge_index, frames, node_mask=node_mask)
        return sv_recover(self.aggregate(message, edge_index, dim_size=node_rep[0].shape[0]), self.vector_output_dim)


class GCPInteractions(nn.Module):
    """gcpnet.py:740-930."""

    def __init__(self, node_dims, edge_dims, cfg, layer_cfg, dropout: float = 0.0, nonlinearities=None, update_node_positions: bool = True):
        super().__init__()
        if nonlinearities is None:
            nonlinearities = cfg_get(cfg, "nonlinearities")
        self.pre_norm = bool(cfg_get(layer_cfg, "pre_norm", False))
        self.update_node_positions = bool(update_node_positions)
        self.node_positions_weight = float(cfg_get(cfg, "node_positions_weight", 1.0))
        self.update_positions_with_vector_sum = bool(cfg_get(cfg, "update_positions_with_vector_sum", False))
        self.interaction = GCPMessagePassing(node_dims, node_dims, edge_dims, cfg, cfg_get(layer_cfg, "mp_cfg"), reduce_function="sum",
                                             use_scalar_message_attention=cfg_get(layer_cfg, "use_scalar_message_attention", True))
        self.gcp_norm = nn.ModuleList([GCPLayerNorm(node_dims, use_gcp_norm=cfg_get(layer_cfg, "use_gcp_norm", False))])
        self.gcp_dropout = nn.ModuleList([GCPDropout(dropout, use_gcp_dropout=cfg_get(layer_cfg, "use_gcp_dropout", False))])
        s, v = node_dims
        n_ff = int(cfg_get(layer_cfg, "num_feedforward_layers", 1))
        hidden = (s, v) if n_ff == 1 else (4 * s, 2 * v)
        ff = [get_GCP_with_custom_cfg((2 * s, 2 * v), hidden, cfg, vector_residual=False,
                                      nonlinearities=(None, None) if n_ff == 1 else cfg_get(cfg, "nonlinearities"), feedforward_out=n_ff == 1)]
        ff.extend(get_GCP_with_custom_cfg(hidden, hidden, cfg, nonlinearities=nonlinearities) for _ in range(n_ff - 2))
        if n_ff > 1:
            ff.append(get_GCP_with_custom_cfg(hidden, node_dims, cfg, vector_residual=False, nonlinearities=(None, None), feedforward_out=True))
        self.feedforward_network = nn.ModuleList(ff)
        if self.update_node_positions:
            pos_out = node_dims if self.update_positions_with_vector_sum else (s, 1)
            self.node_position_update_gcp = get_GCP_with_custom_cfg(node_dims, pos_out, cfg, vector_residual=False, nonlinearities=cfg_get(cfg, "nonlinearities"))

    def derive_x_update(self, node_rep: SV, edge_index: torch.Tensor, f_ij: torch.Tensor, node_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        _, v = self.node_position_update_gcp(node_rep, edge_index, f_ij, node_inputs=True, node_mask=node_mask)
        x_update = v.sum(1) if self.update_positions_with_vector_sum else v.squeeze(1)
        return x_update * self.node_positions_weight

    def forward(self, node_rep: SV, edge_rep: SV, edge_index: torch.Tensor, frames: torch.Tensor, node_mask: Optional[torch.Tensor] = None,
                node_pos: Optional[torch.Tensor] = None):
        node_rep = (node_rep[0], node_rep[1])
        if self.pre_norm:
            node_rep = self.gcp_norm[0](node_rep)
        agg_s, agg_v = self.interaction(node_rep, edge_rep, edge_index, frames, node_mask=node_mask)
        hidden = (torch.cat((agg_s, node_rep[0]), dim=-1), torch.cat((agg_v, node_rep[1]), dim=-2))
        for module in self.feedforward_network:
            hidden = module(hidden, edge_index, frames, node_inputs=True, node_mask=node_mask)
        hidden = self.gcp_dropout[0](hidden)
        node_rep = (node_rep[0] + hidden[0], node_rep[1] + hidden[1])
        if not self.pre_norm:
            node_rep = self.gcp_norm[0](node_rep)
        if node_mask is not None:
            m = node_mask.float()
            node_rep = (node_rep[0] * m.unsqueeze(-1), node_rep[1] * m.reshape(-1, 1, 1))
        if not self.update_node_positions:
            return node_rep
        node_pos = node_pos + self.derive_x_update(node_rep, edge_index, frames, node_mask=node_mask)
        if node_mask is not None:
            node_pos = node_pos * node_mask.float().unsqueeze(-1)
        return node_rep, node_pos
