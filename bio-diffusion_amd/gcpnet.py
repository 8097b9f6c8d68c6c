"""GCPNet dynamics network on MI355X -- host-side mirror of the reference module interface.

Mirrors (names, constructor keywords, ``forward`` signature, ``state_dict`` keys) of
``src/models/components/gcpnet.py``: ``GCP2`` (:265-491), ``GCPEmbedding`` (:494-603), ``GCPMessagePassing``
(:618-737), ``GCPInteractions`` (:740-930) and ``GCPNetDynamics`` (:933-1232).  The sub-modules (gcp_modules.py) reproduce the
reference's state-dict keys (so a released ``*-EMA.ckpt`` loads under the prefix ``ddpm.dynamics_network.``) and are callable.
``GCPNetDynamics.forward`` has two HIP paths and no eager / CPU fallback:
  * FUSED (libgcdm_hip.so, C ABI include/gcdm_hip.h): the production configuration, evaluation / sampling -- ~25 launches per call;
  * MODULES (libgcdm_ops.so, C ABI include/gcdm_ops.h): the reference's module graph composed of this repository's forward / backward
    operators -- every flag and width of the Hydra surface, and training with autograd.
``GCPNetDynamics.path`` ("auto" | "fused" | "modules") selects; "auto" takes the fused path whenever the configuration allows it and no
gradient is being recorded.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional, Tuple

import torch
from torch import nn

from . import _native
from .config import cfg_get

NODE_FEATURE_DIFFUSION_TARGETS = ["atom_types_and_coords"]


from .gcp_modules import (GCP, GCP2, GCPDropout, GCPEmbedding, GCPInteractions, GCPLayerNorm, GCPMessagePassing,   # noqa: F401,E402
                          get_GCP_with_custom_cfg, selected_gcp_class, _embedding_gcp)


class F16RangeError(RuntimeError):
    """An activation left the range of the split-precision images (deferred range guard of GCPNetDynamics.forward)."""


class GCPNetDynamics(nn.Module):
    """Drop-in for the reference ``dynamics_networks["gcpnet"]`` (src/models/qm9_mol_gen_ddpm.py:101-131).

    ``forward(batch, xh [N,3+F], t [N,1], **kwargs) -> (batch, net_out [N,3+F])`` as gcpnet.py:1042-1052.
    ``batch`` needs ``.batch`` (sorted molecule index per node), ``.mask`` (all True, as in sampling) and, for
    a property-conditional model, ``.props_context`` [N,C].
    """

    def __init__(self, model_cfg, module_cfg, layer_cfg, diffusion_cfg, dataloader_cfg, **kwargs):
        super().__init__()
        if cfg_get(diffusion_cfg, "diffusion_target", "atom_types_and_coords") not in NODE_FEATURE_DIFFUSION_TARGETS:
            raise NotImplementedError("only diffusion_target=atom_types_and_coords is built")
        self.num_atom_types = int(cfg_get(dataloader_cfg, "num_atom_types"))
        self.include_charges = bool(cfg_get(dataloader_cfg, "include_charges"))
        self.num_x_dims = int(cfg_get(dataloader_cfg, "num_x_dims", 3))
        self.condition_on_time = bool(cfg_get(diffusion_cfg, "condition_on_time", True))
        self.num_context_node_features = len(cfg_get(module_cfg, "conditioning", []) or [])
        self.condition_on_context = self.num_context_node_features > 0
        # self-conditioning doubles the embedding inputs (gcpnet.py:955-975); the projection keeps h_in outputs (:1027)
        self.self_condition = bool(cfg_get(diffusion_cfg, "self_condition", False))
        h_diff = self.num_atom_types + int(self.include_charges)
        h_in = h_diff + int(self.condition_on_time) + self.num_context_node_features
        mult = 2 if self.self_condition else 1
        self.edge_input_dims = (int(cfg_get(model_cfg, "e_input_dim", 1)) * mult, int(cfg_get(model_cfg, "xi_input_dim", 1)) * mult)
        self.node_input_dims = (h_in + (h_diff if self.self_condition else 0), int(cfg_get(model_cfg, "chi_input_dim", 2)) * mult)
        if self.edge_input_dims != (mult, mult) or self.node_input_dims[1] != 2 * mult:
            raise NotImplementedError("e_input_dim = xi_input_dim = 1 and chi_input_dim = 2 are what the featuriser of the dynamics produces "
                                      "(squared distance, unit vector, two orientation vectors: gcpnet.py:1105-1109)")
        self.edge_dims = (int(cfg_get(model_cfg, "e_hidden_dim")), int(cfg_get(model_cfg, "xi_hidden_dim")))
        self.node_dims = (int(cfg_get(model_cfg, "h_hidden_dim")), int(cfg_get(model_cfg, "chi_hidden_dim")))
        self.num_layers = int(cfg_get(model_cfg, "num_encoder_layers"))
        self.bottleneck = int(cfg_get(module_cfg, "bottleneck", 4))
        self.node_positions_weight = float(cfg_get(module_cfg, "node_positions_weight", 1.0))
        self.norm_x_diff = bool(cfg_get(module_cfg, "norm_x_diff", True))
        self._diffusion_cfg = diffusion_cfg

        # (as gcpnet.py:1006-1014: the embedding keeps GCPEmbedding's own default nonlinearities ("silu", "silu") whatever module_cfg says)
        self.gcp_embedding = GCPEmbedding(self.edge_input_dims, self.node_input_dims, self.edge_dims, self.node_dims, num_atom_types=0,
                                          cfg=module_cfg, use_gcp_norm=cfg_get(layer_cfg, "use_gcp_norm", False))
        self.interaction_layers = nn.ModuleList(
            GCPInteractions(self.node_dims, self.edge_dims, cfg=module_cfg, layer_cfg=layer_cfg, dropout=float(cfg_get(model_cfg, "dropout", 0.0)),
                            update_node_positions=True) for _ in range(self.num_layers))
        self.scalar_node_projection_gcp = _embedding_gcp(module_cfg, self.node_dims, (h_in, 0), (None, None))
        # Which path evaluates forward(): "auto" = the fused sampling kernels when the configuration is the one they are built for and no
        # gradient is recorded, the module graph (HIP operators, autograd) otherwise; "fused" / "modules" force one.
        self.path = "auto"
        self.fused_unsupported = self._why_not_fused(model_cfg, module_cfg, layer_cfg)

        # native state (created lazily on the first forward, on the tensors' device)
        self._lib = None
        self._handle = None
        self._weights_version = None
        self._plan_key = None
        self._flags = None
        # Range guard of the split-precision mode on the module-level call (plug point 1):
        #   True (default): one host sync per call, the call recomputes itself with fp32 MFMA when an activation left the f16 images
        #                   (|x| > 1.2e8; trained models stay below 1e3) -- every output handed to a foreign caller has been checked;
        #   "deferred":     no host sync per call -- the device flag word of call k is copied to pinned host memory asynchronously and looked at
        #                   when call k+1 (or read_flags / check_deferred_flags) comes; an overflow then raises F16RangeError and the handle
        #                   falls back to fp32 MFMA.  The LAST call of a sequence is only checked by check_deferred_flags(wait=True): this
        #                   package's own drivers (ddpm.forward, sample_p_zs_given_zt + sample_p_xh_given_z0) select it per call
        #                   (`_range_check="deferred"`) and do that final check before they return anything;
        #   False:          no check.
        self.check_f16_range = True
        self._flags_host = None
        self._flags_event = None

    # ------------------------------------------------------------------------------------------
    def _why_not_fused(self, model_cfg, module_cfg, layer_cfg):
        """None if libgcdm_hip.so's fused kernels implement this configuration (the reference's production YAMLs and what varies between
        its datasets), else the first reason they do not -- such models run on the module path."""
        mp = cfg_get(layer_cfg, "mp_cfg")
        nl = cfg_get(module_cfg, "nonlinearities") or (None, None)
        checks = [
            (selected_gcp_class(module_cfg) is GCP2, "module_cfg.selected_GCP is not GCP2"),
            (bool(cfg_get(module_cfg, "vector_gate", True)) and not cfg_get(module_cfg, "frame_gate", False), "vector_gate / frame_gate"),
            (not cfg_get(module_cfg, "vector_residual", False) and not cfg_get(module_cfg, "default_vector_residual", False), "vector residuals"),
            (not (cfg_get(module_cfg, "ablate_frame_updates", False) or cfg_get(module_cfg, "ablate_scalars", False) or cfg_get(module_cfg, "ablate_vectors", False)),
             "ablation flags"),
            (tuple(str(n).lower() for n in nl) == ("silu", "silu"), "nonlinearities other than silu"),
            (int(cfg_get(module_cfg, "bottleneck", 4)) == 4 and int(cfg_get(module_cfg, "default_bottleneck", 4)) == 4, "bottleneck != 4"),
            (self.norm_x_diff, "norm_x_diff = False"),
            (not cfg_get(module_cfg, "update_positions_with_vector_sum", False), "update_positions_with_vector_sum"),
            (int(cfg_get(mp, "num_message_layers", 4)) == 4 and bool(cfg_get(mp, "use_residual_message_gcp", True)), "mp_cfg: 4 residual message GCPs"),
            (bool(cfg_get(layer_cfg, "use_scalar_message_attention", True)), "no scalar message attention"),
            (int(cfg_get(layer_cfg, "num_feedforward_layers", 1)) == 1, "num_feedforward_layers != 1"),
            (not cfg_get(layer_cfg, "use_gcp_norm", False), "use_gcp_norm"),
            (self.node_dims == (256, 32) and self.edge_dims in ((64, 16), (16, 8)), f"hidden sizes {self.node_dims} / {self.edge_dims}"),
            (self.num_atom_types + int(self.include_charges) > 0, "no node features (position-only diffusion, generate_x_only)"),
        ]
        for ok, why in checks:
            if not ok:
                return why
        return None

    def _dropout_active(self) -> bool:
        return self.training and any(l.gcp_dropout[0].use_gcp_dropout and l.gcp_dropout[0].drop_rate > 0 for l in self.interaction_layers)

    def _use_modules(self, xh: torch.Tensor) -> bool:
        if self.path == "modules":
            return True
        recording = torch.is_grad_enabled() and (xh.requires_grad or (self.training and any(p.requires_grad for p in self.parameters())))
        if self.path == "fused":
            if self.fused_unsupported is not None:
                raise NotImplementedError(f"path='fused': the fused kernels do not implement this configuration ({self.fused_unsupported})")
            if recording:
                raise RuntimeError("path='fused' has no backward pass; use path='auto' or 'modules' for training")
            return False
        return self.fused_unsupported is not None or recording or self._dropout_active()

    def forward_modules(self, batch: Any, xh: torch.Tensor, t: torch.Tensor, xh_self_cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        """atom_types_and_coords_forward (gcpnet.py:1069-1232) as the reference evaluates it -- module by module -- on this repository's HIP
        operators (ops.py), differentiable.  Any configuration; ~10^2 launches per layer, so sampling prefers the fused path."""
        from . import ops
        from .config import AttrDict
        if torch.is_grad_enabled() and xh.requires_grad:
            # frames, edge features, orientations and mean frames are functions of the input positions evaluated WITHOUT a backward twin
            # (ops.localize / edge_features / orientations / mean_frames): d(out)/d(xh) would silently miss those terms
            raise NotImplementedError("GCPNetDynamics (bio-diffusion_amd): gradients with respect to the input xh are not built (parameter gradients "
                                      "only -- the geometry operators of the network input have no backward); detach xh")
        nx = self.num_x_dims
        bi = cfg_get(batch, "batch")
        mask = cfg_get(batch, "mask")
        if mask is None:
            mask = torch.ones_like(bi, dtype=torch.bool)
        mf = mask.to(torch.float32).unsqueeze(-1)
        xh = xh.to(torch.float32) * mf
        x_init, h = xh[:, :nx].contiguous(), xh[:, nx:]
        N = xh.shape[0]
        counts = torch.unique_consecutive(bi, return_counts=True)[1]
        edge_index = ops.fully_connected_edge_index(counts.cpu(), xh.device)
        full_mask = bool(mask.all())
        if not full_mask:                                        # edges of masked nodes do not exist (gcpnet.py:1062-1065)
            edge_index = edge_index[:, mask[edge_index[0]] & mask[edge_index[1]]].contiguous()
        chi = ops.orientations(x_init)
        e, xi = ops.edge_features(x_init, edge_index)
        if self.self_condition:
            sc = torch.zeros_like(xh) if xh_self_cond is None else xh_self_cond.to(torch.float32)
            x_sc = sc[:, :nx].contiguous()
            e_sc, xi_sc = ops.edge_features(x_sc, edge_index)
            h = torch.cat((h, sc[:, nx:]), dim=-1)
            chi = torch.cat((chi, ops.orientations(x_sc)), dim=1)
            e, xi = torch.cat((e, e_sc), dim=-1), torch.cat((xi, xi_sc), dim=1)
        if self.condition_on_time:
            tt = t.to(torch.float32).reshape(-1, 1)
            h = torch.cat((h, tt.expand(N, 1) if tt.numel() == 1 else tt), dim=-1)
        if self.condition_on_context:
            h = torch.cat((h, cfg_get(batch, "props_context").to(torch.float32).reshape(N, self.num_context_node_features)), dim=-1)
        x = ops.centralize(x_init, bi, mask)
        f_ij = ops.localize(x, edge_index, self.norm_x_diff)
        node_mask = None if full_mask else mask
        b = AttrDict(h=h, chi=chi, e=e, xi=xi, edge_index=edge_index, f_ij=f_ij, mask=node_mask)
        (hh, cc), (ee, xx) = self.gcp_embedding(b)
        for layer in self.interaction_layers:
            (hh, cc), x = layer((hh, cc), (ee, xx), edge_index, f_ij, node_mask=node_mask, node_pos=x)
        h_out = self.scalar_node_projection_gcp((hh, cc), edge_index, f_ij, node_inputs=True, node_mask=node_mask)
        vel = (x - x_init) * mf
        if self.condition_on_context:
            h_out = h_out[:, : -self.num_context_node_features]
        if self.condition_on_time:
            h_out = h_out[:, :-1]
        if bool(vel.isnan().any()):                                  # gcpnet.py:1213-1216
            vel = torch.zeros_like(vel)
        vel = ops.centralize(vel, bi, mask)
        return torch.cat((vel, h_out), dim=-1)

    def _native_config(self, device_index: int) -> "_native.GcdmConfig":
        dc = self._diffusion_cfg
        nv = list(cfg_get(dc, "norm_values", [1.0, 4.0, 10.0]))
        nb = [0.0 if v is None else float(v) for v in cfg_get(dc, "norm_biases", [None, 0.0, 0.0])]
        cfg = _native.GcdmConfig()
        cfg.abi_version = _native.ABI_VERSION
        cfg.num_atom_types = self.num_atom_types
        cfg.include_charges = int(self.include_charges)
        cfg.num_context = self.num_context_node_features
        cfg.condition_on_time = int(self.condition_on_time)
        cfg.num_layers = self.num_layers
        cfg.h_hidden_dim, cfg.chi_hidden_dim = self.node_dims
        cfg.e_hidden_dim, cfg.xi_hidden_dim = self.edge_dims
        cfg.bottleneck = self.bottleneck
        cfg.num_timesteps = int(cfg_get(dc, "num_timesteps", 1000))
        cfg.node_positions_weight = self.node_positions_weight
        cfg.norm_values = (C.c_float * 3)(*[float(v) for v in nv])
        cfg.norm_biases = (C.c_float * 3)(*nb)
        cfg.device = device_index
        cfg.self_condition = int(self.self_condition)
        return cfg

    def _ensure_handle(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("GCPNetDynamics (bio-diffusion_amd) runs on an MI355X only: tensors must be on a HIP "
                               "('cuda') device; there is no CPU fallback")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is not None and self._device_index == idx:
            return
        self.release()
        self._lib = _native.load()
        h = C.c_void_p()
        cfg = self._native_config(idx)
        st = self._lib.gcdm_create(C.byref(cfg), C.byref(h))
        self._handle, self._device_index = h, idx
        _native.check(self._lib, h, st, "gcdm_create")
        self._weights_version = None
        self._plan_key = None
        with torch.inference_mode(False):       # normal tensors even when the handle is first created under inference_mode: the NLL path,
            # sample_p_zs_given_zt and sample() run under @torch.inference_mode(), a later plain no_grad call must still be able to write them
            self._flags = torch.zeros(1, dtype=torch.int32, device=device)
            self._flags_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._flags_event = torch.cuda.Event()
        self._flags_pending = False

    def _params_fingerprint(self):
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def sync_weights(self, force: bool = False):
        """Uploads (re-packs) the parameters if they changed since the last upload."""
        ver = self._params_fingerprint()
        if not force and ver == self._weights_version:
            return
        lib, h = self._lib, self._handle
        for key, val in self.state_dict().items():
            w = val.detach().to("cpu", torch.float32).contiguous()
            st = lib.gcdm_set_weight(h, key.encode(), C.c_void_p(w.data_ptr()), w.numel())
            _native.check(lib, h, st, f"gcdm_set_weight({key})")
        _native.check(lib, h, lib.gcdm_finalize_weights(h), "gcdm_finalize_weights")
        self._weights_version = ver

    def plan(self, num_nodes, node_mask: Optional[torch.Tensor] = None) -> None:
        """Builds the batch topology (molecule sizes, optionally a node mask with False = masked atoms) once; constant over the sampling loop."""
        nn_ = torch.as_tensor(num_nodes, dtype=torch.int32, device="cpu").contiguous()
        key = tuple(nn_.tolist())
        mk = None
        if node_mask is not None and not bool(node_mask.all()):
            mk = node_mask.detach().to("cpu", torch.uint8).contiguous()
            key = key + (mk.numpy().tobytes(),)
        if key == self._plan_key:
            return
        self._plan_key = None          # a rejected request leaves NO plan behind (the library drops the old one first)
        if mk is None:
            st = self._lib.gcdm_plan_batch(self._handle, len(nn_), C.c_void_p(nn_.data_ptr()))
        else:
            if mk.numel() != int(nn_.sum()):
                raise ValueError("node_mask must have one entry per node")
            st = self._lib.gcdm_plan_batch_masked(self._handle, len(nn_), C.c_void_p(nn_.data_ptr()), C.c_void_p(mk.data_ptr()))
        _native.check(self._lib, self._handle, st, "gcdm_plan_batch")
        self._plan_key = key
        self._plan_src = None
        self._plan_keep = None

    def _plan_from_batch_index(self, batch_index: torch.Tensor, mask: Optional[torch.Tensor]):
        # Short cut for the reference-signature step loops, which pass the SAME tensors 1000 times: (address, length, version) of both tensors.
        # The cache KEEPS the tensors alive (`_plan_keep`): while the key is held the caching allocator cannot hand their addresses to the
        # tensors of a different batch (a loop under inference_mode builds batch_index / mask afresh per call and frees them on return).
        # Inference tensors have no version counter (tensor_version == -1): for them the key is address + length only -- this package never
        # edits them in place; a caller who does must call plan() explicitly.
        src = (batch_index.data_ptr(), batch_index.shape[0], _native.tensor_version(batch_index),
               None if mask is None else (mask.data_ptr(), mask.shape[0], _native.tensor_version(mask)))
        if getattr(self, "_plan_src", None) == src and self._plan_key is not None and getattr(self, "_plan_keep", None) is not None:
            return
        counts = torch.unique_consecutive(batch_index, return_counts=True)[1]
        self.plan(counts.cpu(), mask)              # masked nodes (batch.mask with False entries): gcdm_plan_batch_masked
        self._plan_src = src
        self._plan_keep = (batch_index, mask)

    # ------------------------------------------------------------------------------------------
    def forward(self, batch: Any, xh: torch.Tensor, t: torch.Tensor, **kwargs: Any) -> Tuple[Any, torch.Tensor]:
        sc = kwargs.get("xh_self_cond")
        if sc is not None and not self.self_condition:
            sc = None                  # as the reference: the input is ignored unless diffusion_cfg.self_condition (gcpnet.py:1112)
        if self._use_modules(xh):
            if xh.device.type != "cuda":
                raise RuntimeError("GCPNetDynamics (bio-diffusion_amd) runs on an MI355X only: tensors must be on a HIP ('cuda') device; there is no CPU fallback")
            return batch, self.forward_modules(batch, xh, t, xh_self_cond=sc)
        self._ensure_handle(xh.device)
        self.sync_weights()
        self._plan_from_batch_index(cfg_get(batch, "batch"), cfg_get(batch, "mask"))
        ctx = cfg_get(batch, "props_context") if self.condition_on_context else None
        guard = kwargs.get("_range_check", self.check_f16_range)
        if guard == "deferred":
            self.check_deferred_flags(wait=False)          # the previous call's flag word, if it has arrived
        out = self.native_forward(xh, t, ctx, xh_self_cond=sc)
        if self.mfma_mode == 1 and guard == "deferred":
            self._flags_host.copy_(self._flags, non_blocking=True)
            self._flags_event.record(torch.cuda.current_stream(xh.device))
            self._flags_pending = True
        elif self.mfma_mode == 1 and guard:
            # one host sync to make the module-level call self-healing (the fused sampler loop reads the flag once per run instead):
            # an activation beyond the f16 range -> recompute this call with fp32 MFMA
            if self.read_flags() & _native.FLAG_F16_RANGE:
                self.set_mfma_mode(0)
                try:
                    out = self.native_forward(xh, t, ctx, xh_self_cond=sc)
                finally:
                    self.set_mfma_mode(1)
        return batch, out

    def check_deferred_flags(self, wait: bool = True) -> int:
        """Looks at the asynchronously copied flag word of the last module-level call (deferred range guard).  Raises F16RangeError -- after
        switching the handle to fp32 MFMA, so that a re-run is correct -- if an activation left the range of the split-precision images."""
        if not getattr(self, "_flags_pending", False) or self._flags_event is None:
            return 0
        if not wait and not self._flags_event.query():
            return 0
        self._flags_event.synchronize()
        self._flags_pending = False
        v = int(self._flags_host.item())
        if v & _native.FLAG_TAIL:
            self.disable_fused_layer("an earlier forward call")
        if v & _native.FLAG_F16_RANGE:
            self._flags.zero_()
            self.set_mfma_mode(0)
            raise F16RangeError("an activation exceeded the range of the split-precision (f16x3) images in an earlier forward call; its output "
                                "was invalid.  The handle now runs fp32 MFMA: re-run the computation, then call set_mfma_mode(1) to return to the "
                                "default mode (or set check_f16_range=True for a self-healing, host-synchronising call)")
        return v

    @property
    def mfma_mode(self) -> int:
        return int(self._lib.gcdm_get_option(self._handle, b"mfma_mode")) if self._handle is not None else -1

    def set_mfma_mode(self, mode: int) -> None:
        """0: fp32 MFMA everywhere; 1: split-precision f16x3 edge kernel (fp32-equivalent accuracy, default)."""
        _native.check(self._lib, self._handle, self._lib.gcdm_set_option(self._handle, b"mfma_mode", int(mode)), "gcdm_set_option")

    def native_forward(self, xh: torch.Tensor, t: torch.Tensor, context: Optional[torch.Tensor] = None,
                       out: Optional[torch.Tensor] = None, xh_self_cond: Optional[torch.Tensor] = None) -> torch.Tensor:
        N = int(self._lib.gcdm_num_nodes(self._handle))
        D = 3 + self.num_atom_types + int(self.include_charges)
        if xh.shape != (N, D):
            raise ValueError(f"xh has shape {tuple(xh.shape)}, plan expects {(N, D)}")
        xh = xh.detach().to(torch.float32).contiguous()
        t = t.detach().to(torch.float32).reshape(-1)
        if t.numel() == 1:
            t = t.expand(N)
        t = t.contiguous()
        if t.numel() != N:
            raise ValueError("t must have one entry per node")
        cptr = None
        if self.condition_on_context:
            if context is None:
                raise ValueError("props_context required by a context-conditioned model")
            context = context.detach().to(torch.float32).reshape(N, self.num_context_node_features).contiguous()
            cptr = C.c_void_p(context.data_ptr())
        if out is None:
            out = torch.empty_like(xh)
        stream = torch.cuda.current_stream(xh.device).cuda_stream
        sptr = None
        if xh_self_cond is not None:
            if xh_self_cond.shape != (N, D):
                raise ValueError(f"xh_self_cond has shape {tuple(xh_self_cond.shape)}, expected {(N, D)}")
            xh_self_cond = xh_self_cond.detach().to(xh.device, torch.float32).contiguous()
            sptr = C.c_void_p(xh_self_cond.data_ptr())
        st = self._lib.gcdm_forward_sc(self._handle, C.c_void_p(xh.data_ptr()), sptr, C.c_void_p(t.data_ptr()), cptr,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(self._flags.data_ptr()), C.c_void_p(stream))
        _native.check(self._lib, self._handle, st, "gcdm_forward_sc")
        return out

    # ------------------------------------------------------------------------------------------
    def disable_fused_layer(self, where: str) -> None:
        """GCDM_FLAG_TAIL was raised (fused layer launch: the workgroup-placement check or a bounded dependency wait failed -- never observed): from now on two
        launches per layer on this handle.  The flag comes together with GCDM_FLAG_F16_RANGE, so the caller's fp32 re-run has already repaired / will repair the result."""
        import warnings
        if self._handle is not None and self._lib.gcdm_get_option(self._handle, b"fuse_node") != 0:
            self._lib.gcdm_set_option(self._handle, b"fuse_node", 0)
            warnings.warn(f"bio-diffusion_amd: the fused layer launch reported GCDM_FLAG_TAIL in {where}; the handle now uses two launches per layer "
                          "(option fuse_node = 0) and the affected call is re-run", RuntimeWarning)

    def read_flags(self, reset: bool = True) -> int:
        """Device-side check word (host sync): bit 0 NaN in vel, bit 2 CoG drift re-projected."""
        self._flags_pending = False
        v = int(self._flags.item()) if self._flags is not None else 0
        if v & _native.FLAG_TAIL:
            self.disable_fused_layer("a forward call")
        if reset and self._flags is not None:
            self._flags.zero_()
        return v

    def debug_read(self, name: str) -> torch.Tensor:
        n = self._lib.gcdm_debug_read(self._handle, name.encode(), None, 0)
        _native.check(self._lib, self._handle, n, f"gcdm_debug_read({name})")
        buf = torch.empty(int(n), dtype=torch.float32)
        n2 = self._lib.gcdm_debug_read(self._handle, name.encode(), C.c_void_p(buf.data_ptr()), buf.numel())
        _native.check(self._lib, self._handle, n2, f"gcdm_debug_read({name})")
        return buf

    def debug_set_layer_limit(self, n: int):
        self._lib.gcdm_debug_set_layer_limit(self._handle, int(n))

    def flops_executed(self) -> float:
        return float(self._lib.gcdm_forward_flops_executed(self._handle))

    def release(self):
        if self._handle is not None and self._lib is not None:
            self._lib.gcdm_destroy(self._handle)
        self._handle = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass
