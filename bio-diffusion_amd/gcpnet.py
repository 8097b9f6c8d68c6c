"""GCPNet dynamics network on MI355X -- host-side mirror of the reference module interface.

Mirrors (names, constructor keywords, ``forward`` signature, ``state_dict`` keys) of
``src/models/components/gcpnet.py``: ``GCP2`` (:265-491), ``GCPEmbedding`` (:494-603), ``GCPMessagePassing``
(:618-737), ``GCPInteractions`` (:740-930) and ``GCPNetDynamics`` (:933-1232).  The sub-modules here are
*parameter containers* whose attribute names reproduce the reference's state-dict keys (so a released
``*-EMA.ckpt`` loads under the prefix ``ddpm.dynamics_network.``); all arithmetic of
``GCPNetDynamics.forward`` happens in ``libgcdm_hip.so`` (HIP, gfx950) through the C ABI in
``include/gcdm_hip.h``.  There is no eager / CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Optional, Tuple

import torch
from torch import nn

from . import _native
from .config import cfg_get

NODE_FEATURE_DIFFUSION_TARGETS = ["atom_types_and_coords"]


def _fused_only(name):
    raise NotImplementedError(
        f"{name}.forward is not a stand-alone op in bio-diffusion_amd: the GCP2 blocks are fused into the HIP kernels "
        "driven by GCPNetDynamics.forward (libgcdm_hip.so).")


class GCP2(nn.Module):
    """Parameter container with the reference's GCP2 layout (gcpnet.py:286-348); production flags only
    (vector_gate=True, frame_gate=False, no residuals, no ablations)."""

    def __init__(self, input_dims, output_dims, nonlinearities=("silu", "silu"), scalar_out_nonlinearity="silu",
                 scalar_gate: int = 0, vector_gate: bool = True, frame_gate: bool = False, sigma_frame_gate: bool = False,
                 feedforward_out: bool = False, bottleneck: int = 1, vector_residual: bool = False,
                 vector_frame_residual: bool = False, ablate_frame_updates: bool = False, ablate_scalars: bool = False,
                 ablate_vectors: bool = False, scalarization_vectorization_output_dim: int = 3, **kwargs):
        super().__init__()
        if (frame_gate or not vector_gate or vector_residual or ablate_frame_updates or ablate_scalars or ablate_vectors
                or scalar_gate or scalarization_vectorization_output_dim != 3):
            raise NotImplementedError("bio-diffusion_amd builds the production GCP2 variant only "
                                      "(vector_gate, no frame_gate / residual / ablation / scalar_gate)")
        self.scalar_input_dim, self.vector_input_dim = input_dims
        self.scalar_output_dim, self.vector_output_dim = output_dims
        self.nonlinearities = tuple(nonlinearities) if nonlinearities is not None else (None, None)
        self.feedforward_out = feedforward_out
        self.bottleneck = bottleneck
        if not self.vector_input_dim:
            raise NotImplementedError("scalar-only GCP2 is not on the GCDM sampling path")
        assert self.vector_input_dim % bottleneck == 0
        self.hidden_dim = (self.vector_input_dim // bottleneck if bottleneck > 1
                           else max(self.vector_input_dim, self.vector_output_dim))
        k = self.hidden_dim + self.scalar_input_dim + 9
        self.vector_down = nn.Linear(self.vector_input_dim, self.hidden_dim, bias=False)
        self.scalar_out = (nn.Sequential(nn.Linear(k, self.scalar_output_dim), nn.SiLU(),
                                         nn.Linear(self.scalar_output_dim, self.scalar_output_dim))
                           if feedforward_out else nn.Linear(k, self.scalar_output_dim))
        self.vector_down_frames = nn.Linear(self.vector_input_dim, 3, bias=False)
        if self.vector_output_dim:
            self.vector_up = nn.Linear(self.hidden_dim, self.vector_output_dim, bias=False)
            self.vector_out_scale = nn.Linear(self.scalar_output_dim, self.vector_output_dim)

    def forward(self, *a, **k):
        _fused_only("GCP2")


def _gcp_from_cfg(cfg, input_dims, output_dims, **kw) -> GCP2:
    """get_GCP_with_custom_cfg (gcpnet.py:606-615) for the production flags."""
    args = dict(nonlinearities=cfg_get(cfg, "nonlinearities"), scalar_gate=cfg_get(cfg, "scalar_gate", 0),
                vector_gate=cfg_get(cfg, "vector_gate", True), frame_gate=cfg_get(cfg, "frame_gate", False),
                sigma_frame_gate=cfg_get(cfg, "sigma_frame_gate", False), bottleneck=cfg_get(cfg, "bottleneck", 4),
                vector_residual=cfg_get(cfg, "vector_residual", False),
                vector_frame_residual=cfg_get(cfg, "vector_frame_residual", False),
                ablate_frame_updates=cfg_get(cfg, "ablate_frame_updates", False),
                ablate_scalars=cfg_get(cfg, "ablate_scalars", False), ablate_vectors=cfg_get(cfg, "ablate_vectors", False))
    args.update(kw)
    return GCP2(input_dims, output_dims, **args)


class GCPEmbedding(nn.Module):
    def __init__(self, edge_input_dims, node_input_dims, edge_hidden_dims, node_hidden_dims, cfg):
        super().__init__()
        nl = cfg_get(cfg, "nonlinearities")
        self.edge_embedding = _gcp_from_cfg(cfg, edge_input_dims, edge_hidden_dims, nonlinearities=nl, bottleneck=1)
        self.node_embedding = _gcp_from_cfg(cfg, node_input_dims, node_hidden_dims, nonlinearities=(None, None), bottleneck=1)

    def forward(self, *a, **k):
        _fused_only("GCPEmbedding")


class GCPMessagePassing(nn.Module):
    def __init__(self, input_dims, output_dims, edge_dims, cfg, mp_cfg):
        super().__init__()
        s_in, v_in = input_dims
        e_s, e_v = edge_dims
        n_msg = cfg_get(mp_cfg, "num_message_layers", 4)
        if n_msg != 4 or not cfg_get(mp_cfg, "use_residual_message_gcp", True):
            raise NotImplementedError("bio-diffusion_amd builds num_message_layers=4 with residual message GCPs")
        bn = cfg_get(cfg, "default_bottleneck", 4)
        mods = [_gcp_from_cfg(cfg, (2 * s_in + e_s, 2 * v_in + e_v), output_dims, bottleneck=bn)]
        for _ in range(n_msg - 2):
            mods.append(_gcp_from_cfg(cfg, output_dims, output_dims))
        mods.append(_gcp_from_cfg(cfg, output_dims, output_dims, bottleneck=bn))
        self.message_fusion = nn.ModuleList(mods)
        self.scalar_message_attention = nn.Sequential(nn.Linear(output_dims[0], 1), nn.Sigmoid())

    def forward(self, *a, **k):
        _fused_only("GCPMessagePassing")


class GCPInteractions(nn.Module):
    def __init__(self, node_dims, edge_dims, cfg, layer_cfg):
        super().__init__()
        if (cfg_get(layer_cfg, "pre_norm", False) or cfg_get(layer_cfg, "use_gcp_norm", False)
                or cfg_get(layer_cfg, "use_gcp_dropout", False) or cfg_get(layer_cfg, "num_feedforward_layers", 1) != 1
                or not cfg_get(layer_cfg, "use_scalar_message_attention", True)):
            raise NotImplementedError("bio-diffusion_amd builds the production interaction layer only "
                                      "(no norm / dropout, one feed-forward GCP2, scalar message attention)")
        if cfg_get(cfg, "update_positions_with_vector_sum", False):
            raise NotImplementedError("update_positions_with_vector_sum is not built")
        self.interaction = GCPMessagePassing(node_dims, node_dims, edge_dims, cfg, cfg_get(layer_cfg, "mp_cfg"))
        s, v = node_dims
        self.feedforward_network = nn.ModuleList([
            _gcp_from_cfg(cfg, (2 * s, 2 * v), (s, v), nonlinearities=(None, None), feedforward_out=True, vector_residual=False)])
        self.node_position_update_gcp = _gcp_from_cfg(cfg, node_dims, (s, 1), vector_residual=False)

    def forward(self, *a, **k):
        _fused_only("GCPInteractions")


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
            raise NotImplementedError("only e_input_dim = xi_input_dim = 1, chi_input_dim = 2 are built")
        self.edge_dims = (int(cfg_get(model_cfg, "e_hidden_dim")), int(cfg_get(model_cfg, "xi_hidden_dim")))
        self.node_dims = (int(cfg_get(model_cfg, "h_hidden_dim")), int(cfg_get(model_cfg, "chi_hidden_dim")))
        self.num_layers = int(cfg_get(model_cfg, "num_encoder_layers"))
        self.bottleneck = int(cfg_get(module_cfg, "bottleneck", 4))
        self.node_positions_weight = float(cfg_get(module_cfg, "node_positions_weight", 1.0))
        if not cfg_get(module_cfg, "norm_x_diff", True):
            raise NotImplementedError("norm_x_diff=False is not built")
        self._diffusion_cfg = diffusion_cfg

        self.gcp_embedding = GCPEmbedding(self.edge_input_dims, self.node_input_dims, self.edge_dims, self.node_dims, module_cfg)
        self.interaction_layers = nn.ModuleList(
            GCPInteractions(self.node_dims, self.edge_dims, module_cfg, layer_cfg) for _ in range(self.num_layers))
        self.scalar_node_projection_gcp = _gcp_from_cfg(module_cfg, self.node_dims, (h_in, 0), nonlinearities=(None, None), bottleneck=1)

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

    def _plan_from_batch_index(self, batch_index: torch.Tensor, mask: Optional[torch.Tensor]):
        src = (batch_index.data_ptr(), batch_index.shape[0], batch_index._version,
               None if mask is None else (mask.data_ptr(), mask._version))
        if getattr(self, "_plan_src", None) == src and self._plan_key is not None:
            return
        counts = torch.unique_consecutive(batch_index, return_counts=True)[1]
        self.plan(counts.cpu(), mask)              # masked nodes (batch.mask with False entries): gcdm_plan_batch_masked
        self._plan_src = src

    # ------------------------------------------------------------------------------------------
    def forward(self, batch: Any, xh: torch.Tensor, t: torch.Tensor, **kwargs: Any) -> Tuple[Any, torch.Tensor]:
        sc = kwargs.get("xh_self_cond")
        if sc is not None and not self.self_condition:
            sc = None                  # as the reference: the input is ignored unless diffusion_cfg.self_condition (gcpnet.py:1112)
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
        if v & _native.FLAG_F16_RANGE:
            self._flags.zero_()
            self.set_mfma_mode(0)
            raise F16RangeError("an activation exceeded the range of the split-precision (f16x3) images in an earlier forward call; its output "
                                "was invalid.  The handle now runs fp32 MFMA: re-run the computation (or set check_f16_range=True for a "
                                "self-healing, host-synchronising call)")
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
    def read_flags(self, reset: bool = True) -> int:
        """Device-side check word (host sync): bit 0 NaN in vel, bit 2 CoG drift re-projected."""
        self._flags_pending = False
        v = int(self._flags.item()) if self._flags is not None else 0
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
