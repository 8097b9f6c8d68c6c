"""Whole-model plug point: ``_target_`` stand-ins for the reference LightningModules.

Boundary only (plug point 2): the constructor keywords, ``load_from_checkpoint``, ``.to(device)``,
``.dataset_info``, ``.ddpm.num_nodes_distribution.sample(n)``, ``.sample(...)`` and ``.generate_molecules(...)`` that
``src/mol_gen_sample.py:81-182`` and ``src/mol_gen_eval_conditional_qm9.py:101-109`` use of
``src/models/qm9_mol_gen_ddpm.py`` / ``geom_mol_gen_ddpm.py``, plus the evaluation driver ``sample_and_analyze`` (``:747-885``)
with the stability statistics computed on the device.  Training / validation hooks, RDKit metrics and RDKit
post-processing are out of scope; ``generate_molecules`` returns raw (positions, atom-type indices, charges) per
molecule and hands them to an optional ``molecule_builder`` (the reference's ``build_molecule``) when given.
"""
from __future__ import annotations

import logging
import os
import pickle
import re
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import nn

from .config import cfg_get, dataset_info as _dataset_info
from .gcpnet import GCPNetDynamics
from .stability import CategoricalDistribution, check_molecular_stability_batch
from .xyz import save_xyz_file
from .variational_diffusion import EquivariantVariationalDiffusion, _segment_mean_sub

log = logging.getLogger(__name__)


DEFAULT_LANES_MIN_SAMPLES = 64     # sample(): plain batches of at least this many molecules are drawn as two slices (see sample)

class _Dummy:
    """Placeholder for globals (omegaconf containers, functools.partial(torch.optim.AdamW), ...) found in the
    ``hyper_parameters`` of a Lightning checkpoint; only ``state_dict`` is used."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        pass

    def __call__(self, *a, **k):
        return _Dummy()


class _StateDictUnpickler(pickle.Unpickler):
    """Resolves ONLY the globals a tensor state-dict needs, by (module, name); everything else in a Lightning checkpoint (omegaconf nodes,
    callbacks, optimizer objects, ...) becomes an inert placeholder.  In particular nothing from `builtins` that can run code (eval, exec,
    getattr, __import__, ...) and no `os` / `subprocess` / `torch.*` callables beyond the tensor rebuild helpers are reachable."""
    _ALLOWED = {
        ("collections", "OrderedDict"), ("collections", "defaultdict"),
        ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset"),
        ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"), ("builtins", "bytes"), ("builtins", "complex"),
        ("builtins", "slice"), ("builtins", "range"),
        ("_codecs", "encode"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
        ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"),
        ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"), ("torch.serialization", "_get_layout"),
        ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"),
    }
    _TORCH_TYPED = re.compile(r"^(Float|Double|Half|BFloat16|Long|Int|Short|Char|Byte|Bool|ComplexFloat|ComplexDouble)Storage$|^(float|double|half|bfloat16|"
                              r"float16|float32|float64|int8|int16|int32|int64|uint8|bool|long|int|short|complex64|complex128|strided)$")

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED or (module in ("torch", "torch.storage") and self._TORCH_TYPED.match(name)) or \
                (module == "torch.storage" and name in ("UntypedStorage", "TypedStorage")):
            # (torch.storage._load_from_bytes is NOT here: it is torch.load(BytesIO(b), weights_only=False) with the default pickle
            #  module, i.e. an unrestricted inner unpickler a crafted checkpoint can REDUCE onto a bytes payload.  Zip-format checkpoints
            #  -- everything torch >= 1.6 and Lightning write -- never reference it.)
            try:
                return super().find_class(module, name)
            except Exception:
                return _Dummy
        return _Dummy


class _PickleShim:
    Unpickler = _StateDictUnpickler
    __name__ = "pickle"

    @staticmethod
    def load(f, **kw):
        return _StateDictUnpickler(f, **kw).load()


def load_lightning_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """Reads ``state_dict`` from a reference ``*.ckpt`` without needing omegaconf / lightning importable."""
    try:
        ckpt = torch.load(path, map_location="cpu", weights_only=False, pickle_module=_PickleShim)
    except TypeError:
        ckpt = torch.load(path, map_location="cpu", pickle_module=_PickleShim)
    return ckpt["state_dict"] if "state_dict" in ckpt else ckpt


class _MoleculeGenerationDDPM(nn.Module):
    _DATASETS: Dict[str, str] = {}

    def __init__(self, optimizer: Any = None, scheduler: Any = None, model_cfg: Any = None, module_cfg: Any = None,
                 layer_cfg: Any = None, diffusion_cfg: Any = None, dataloader_cfg: Any = None, path_cfg: Any = None, **kwargs):
        super().__init__()
        if cfg_get(diffusion_cfg, "dynamics_network", "gcpnet") != "gcpnet":
            raise NotImplementedError("only dynamics_network=gcpnet is built (EGNN is the reference's ablation baseline)")
        if cfg_get(diffusion_cfg, "ddpm_mode", "unconditional") not in ("unconditional", "inpainting"):
            raise ValueError("unsupported ddpm_mode")
        self.ddpm_mode = cfg_get(diffusion_cfg, "ddpm_mode", "unconditional")
        self.T = int(cfg_get(diffusion_cfg, "num_timesteps"))
        self.num_atom_types = int(cfg_get(dataloader_cfg, "num_atom_types"))
        self.num_x_dims = int(cfg_get(dataloader_cfg, "num_x_dims", 3))
        self.include_charges = bool(cfg_get(dataloader_cfg, "include_charges"))
        self.condition_on_context = len(cfg_get(module_cfg, "conditioning", []) or []) > 0
        name = str(cfg_get(dataloader_cfg, "dataset"))
        if name not in self._DATASETS:
            raise ValueError(f"dataset {name!r} not supported by {type(self).__name__}")
        self.dataset_info = _dataset_info(self._DATASETS[name])
        dynamics_network = GCPNetDynamics(model_cfg=model_cfg, module_cfg=module_cfg, layer_cfg=layer_cfg,
                                          diffusion_cfg=diffusion_cfg, dataloader_cfg=dataloader_cfg)
        self.ddpm = EquivariantVariationalDiffusion(dynamics_network=dynamics_network, diffusion_cfg=diffusion_cfg,
                                                    dataloader_cfg=dataloader_cfg, dataset_info=self.dataset_info)
        self.props_distr = None   # set by the conditional-evaluation driver (PropertiesDistribution needs training data)
        # qm9_mol_gen_ddpm.py:196-199: node-type statistics of the training set, for the KL reported by analyze_samples
        self.node_type_distribution = CategoricalDistribution(self.dataset_info["atom_types"], self.dataset_info["atom_encoder"])
        self.molecular_metrics = None   # BasicMolecularMetrics needs RDKit + the training SMILES (out of scope); attach one to get validity etc.
        self._init_kwargs = dict(model_cfg=model_cfg, module_cfg=module_cfg, layer_cfg=layer_cfg, diffusion_cfg=diffusion_cfg,
                                 dataloader_cfg=dataloader_cfg, path_cfg=path_cfg)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    # the reference calls ``model.load_from_checkpoint(...)`` on an instance (mol_gen_sample.py:113-122)
    def load_from_checkpoint(self, checkpoint_path: str, map_location: Any = None, strict: bool = True, **kwargs):
        init = dict(self._init_kwargs)
        init.update({k: v for k, v in kwargs.items() if k in init})
        model = type(self)(**init)
        sd = load_lightning_state_dict(checkpoint_path)
        sd = {k: v for k, v in sd.items() if k.startswith("ddpm.")}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        bad = [k for k in missing if k.startswith("ddpm.dynamics_network.")]
        if strict and (bad or [k for k in unexpected if k.startswith("ddpm.dynamics_network.")]):
            raise RuntimeError(f"checkpoint does not match the dynamics network: missing {bad[:4]} unexpected {unexpected[:4]}")
        if map_location is not None:
            model = model.to(map_location)
        return model

    # ---- loss / likelihood of a data batch (training, validation, test step), qm9_mol_gen_ddpm.py:184-277, 340-360, 429-459 --------
    def forward(self, batch: Any, dtype: torch.dtype = torch.float32, t_int: Optional[torch.Tensor] = None,
                noise: Optional[List[torch.Tensor]] = None) -> Tuple[torch.Tensor, Dict[str, Any]]:
        """Loss per molecule and the batch means of the monitored terms.  Evaluation mode: the NLL (two evaluations of the network, fused
        kernels, inference mode).  Training mode: the L2 or VLB objective of ``diffusion_cfg.loss_type`` with gradients -- the network runs on
        the module path (HIP operators with autograd).  ``batch``: x, one_hot, charges, batch, mask and -- for a conditional model -- the
        per-node ``props_context`` (the reference derives it from the training set's property statistics, qm9utils.prepare_context; that data
        path is outside this package).  ``t_int`` / ``noise``: see EquivariantVariationalDiffusion.forward."""
        if self.training:
            return self._forward_impl(batch, dtype, t_int, noise)
        with torch.inference_mode():
            return self._forward_impl(batch, dtype, t_int, noise)

    def _forward_impl(self, batch, dtype, t_int, noise):
        bi, mask = batch.batch, batch.mask
        B = int(bi.max().item()) + 1
        batch.x = _segment_mean_sub(batch.x, bi, B, mask)                         # centralize(..., edm=True): translation-invariant positions
        batch.h = {"categorical": batch.one_hot, "integer": batch.charges}
        ctx = getattr(batch, "props_context", None)
        if self.condition_on_context:
            if ctx is None:
                raise ValueError("conditional model: batch.props_context (per node, normalised like the training set's) is required")
            batch.props_context = ctx.type(dtype)
        else:
            batch.props_context = None
        num_nodes = torch.zeros(B, dtype=torch.long, device=bi.device).index_add_(0, bi, mask.long())
        batch.num_nodes_present, batch.num_graphs = num_nodes, B
        (delta_log_px, error_t, SNR_weight, loss_0_x, loss_0_h, neg_log_const_0, kl_prior, log_pN, t_int, loss_info) = self.ddpm(
            batch, return_loss_info=True, t_int=t_int, noise=noise)
        if self.training and cfg_get(self._init_kwargs["diffusion_cfg"], "loss_type", "l2") == "l2":
            # L2 training objective (:222-234): the squared error per predicted number, the x part of L_0 normalised the same way
            eff = num_nodes.max() if cfg_get(self._init_kwargs["diffusion_cfg"], "norm_training_by_max_nodes", False) else num_nodes
            denom = (self.num_x_dims + self.ddpm.num_node_scalar_features) * eff
            error_t = error_t / denom
            loss_t = 0.5 * error_t
            loss_0_x = loss_0_x / denom
            loss_0 = loss_0_x + loss_0_h
        else:
            loss_t = self.T * 0.5 * SNR_weight * error_t                          # the variational bound (:236-240); evaluation always scores it
            loss_0 = loss_0_x + loss_0_h + neg_log_const_0
        nll = loss_t + loss_0 + kl_prior - delta_log_px - log_pN                  # normalisation of x undone, joint with the size prior (:243-252)
        for name, v in (("loss_t", loss_t), ("SNR_weight", SNR_weight), ("loss_0", loss_0), ("kl_prior", kl_prior), ("delta_log_px", delta_log_px),
                        ("neg_log_const_0", neg_log_const_0), ("log_pN", log_pN)):
            loss_info[name] = v.mean(0)
        return nll, loss_info

    def training_step(self, batch: Any, batch_idx: int = 0, **kw) -> Dict[str, Any]:
        """qm9_mol_gen_ddpm.py:340-360 without the Lightning metric objects: ``metrics["loss"]`` carries the graph (call ``.backward()`` on it),
        every other entry is detached.  The module must be in training mode."""
        if not self.training:
            raise RuntimeError("training_step needs .train() (evaluation mode scores the NLL without gradients)")
        nll, metrics = self.step(batch, **kw)
        metrics = {k: v.detach() for k, v in metrics.items()}
        metrics["loss"] = nll.mean(0)
        return metrics

    def step(self, batch: Any, **kw) -> Tuple[torch.Tensor, Dict[str, Any]]:
        return self.forward(batch, **kw)

    @torch.inference_mode()
    def validation_step(self, batch: Any, batch_idx: int = 0, **kw) -> Dict[str, Any]:
        """The metrics dictionary of the reference's validation / test step (without the Lightning logging around it)."""
        nll, metrics = self.step(batch, **kw)
        metrics["loss"] = nll.mean(0)
        g = self.ddpm.gamma.gamma
        metrics["log_SNR_max"], metrics["log_SNR_min"] = -g[0], -g[-1]            # -gamma(0), -gamma(1)
        return metrics

    test_step = validation_step

    @torch.inference_mode()
    def sample(self, num_samples: int, num_nodes: Optional[torch.Tensor] = None, node_mask: Optional[torch.Tensor] = None,
               context: Optional[torch.Tensor] = None, fix_noise: bool = False, num_timesteps: Optional[int] = None,
               **kw) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """qm9_mol_gen_ddpm.py:591-633."""
        if num_nodes is None:
            num_nodes = self.ddpm.num_nodes_distribution.sample(num_samples)
            assert int(num_nodes.max()) <= self.dataset_info.get("max_n_nodes", int(num_nodes.max()))
        if self.condition_on_context:
            if context is None:
                if self.props_distr is None:
                    raise ValueError("context required (no props_distr attached)")
                context = self.props_distr.sample_batch(num_nodes)
        else:
            context = None
        plain = not any(k in kw for k in ("lanes", "noise_fn", "step_callback", "_init_xh")) and kw.get("return_frames", 1) == 1
        if plain and not fix_noise and num_samples >= DEFAULT_LANES_MIN_SAMPLES and (node_mask is None or bool(node_mask.all())):
            # plain batches: two slices of the flat batch on two handles / streams (same samples up to fp summation order).  One slice's node kernels
            # and launch tails run under the other's edge kernels: +4-5 % at 1024 QM9 molecules, +17 % at the evaluation driver's 100, +23 % at 64;
            # GEOM-Drugs +-0 at 64, -5 % at 32 (tools/ab_lanes.sh) -- hence the threshold
            kw["lanes"] = 2
        xh, batch_index, _ = self.ddpm.mol_gen_sample(num_samples=num_samples, num_nodes=num_nodes, node_mask=node_mask,
                                                      context=context, fix_noise=fix_noise, fix_self_conditioning_noise=fix_noise,
                                                      device=self.device, num_timesteps=num_timesteps, **kw)
        x = xh[:, : self.num_x_dims]
        one_hot = xh[:, self.num_x_dims:-1] if self.include_charges else xh[:, self.num_x_dims:]
        charges = xh[:, -1:] if self.include_charges else torch.zeros(0, device=self.device)
        return x, one_hot, charges, batch_index

    @torch.inference_mode()
    def generate_molecules(self, ddpm_mode: str = "unconditional", num_samples: int = 1, num_nodes: Optional[torch.Tensor] = None,
                           sanitize: bool = False, largest_frag: bool = False, add_hydrogens: bool = False,
                           sample_chain: bool = False, relax_iter: int = 0, num_timesteps: Optional[int] = None,
                           node_mask: Optional[torch.Tensor] = None, context: Optional[torch.Tensor] = None,
                           num_resamplings: int = 1, jump_length: int = 1,
                           molecule_builder: Optional[Callable[[torch.Tensor, torch.Tensor, Dict[str, Any]], Any]] = None,
                           **kw) -> List[Any]:
        """qm9_mol_gen_ddpm.py:1063-1243 up to (not including) the RDKit post-processing.  ``ddpm_mode="inpainting"`` (:1130-1181): RePaint on an
        all-zero molecule with the nodes of ``node_mask`` fixed (default: the first node of the batch), then every generated molecule is
        moved back to the given one's centre of mass."""
        if sample_chain:
            assert num_samples == 1, "Chain sampling is only supported for single-molecule batches."
            if ddpm_mode != "unconditional":
                raise NotImplementedError("chains are built for ddpm_mode='unconditional' only")
            # every step is a frame (return_frames = num_timesteps, :1122), frames put in generation order (reverse_tensor, :1186)
            T = self.ddpm.T if num_timesteps is None else num_timesteps
            if num_nodes is None:
                num_nodes = self.ddpm.num_nodes_distribution.sample(num_samples)
            if self.condition_on_context and context is None:
                if self.props_distr is None:
                    raise ValueError("context required (no props_distr attached)")
                context = self.props_distr.sample_batch(num_nodes)
            frames, _, _ = self.ddpm.mol_gen_sample(num_samples=1, num_nodes=num_nodes, device=self.device, return_frames=T, num_timesteps=T,
                                                    node_mask=node_mask, context=context if self.condition_on_context else None, **kw)
            frames = frames.reshape(T, -1, frames.shape[-1]).flip(0)
            oh = frames[:, :, self.num_x_dims:-1] if self.include_charges else frames[:, :, self.num_x_dims:]
            mols = []
            for pos, at in zip(frames[:, :, : self.num_x_dims].cpu(), oh.argmax(-1).cpu()):
                mols.append(molecule_builder(pos, at, self.dataset_info) if molecule_builder is not None else (pos, at, None))
            return mols
        if ddpm_mode == "unconditional":
            x, one_hot, charges, batch_index = self.sample(num_samples, num_nodes=num_nodes, node_mask=node_mask, context=context,
                                                           num_timesteps=num_timesteps, **kw)
        elif ddpm_mode == "inpainting":
            if num_nodes is None:
                num_nodes = self.ddpm.num_nodes_distribution.sample(num_samples)
                assert int(num_nodes.max()) <= self.dataset_info.get("max_n_nodes", int(num_nodes.max()))
            if self.condition_on_context:
                if context is None:
                    if self.props_distr is None:
                        raise ValueError("context required (no props_distr attached)")
                    context = self.props_distr.sample_batch(num_nodes)
            else:
                context = None
            dev = self.device
            n_total = int(torch.as_tensor(num_nodes).sum())
            batch_index = torch.repeat_interleave(torch.arange(len(num_nodes), device=dev), torch.as_tensor(num_nodes).to(dev))
            molecule = {"x": torch.zeros((n_total, self.num_x_dims), device=dev), "one_hot": torch.zeros((n_total, self.num_atom_types), device=dev),
                        "charges": torch.zeros((n_total, 1), device=dev), "num_nodes": num_nodes, "batch_index": batch_index}
            if node_mask is None:                     # "largely disable inpainting": only the first node is a fixed point (:1159-1162)
                node_mask = torch.zeros(n_total, dtype=torch.bool, device=dev)
                node_mask[0] = True
            xh = self.ddpm.inpaint(molecule=molecule, node_mask_fixed=node_mask, num_resamplings=num_resamplings, jump_length=jump_length,
                                   num_timesteps=num_timesteps, context=context, **kw)
            cnt = torch.as_tensor(num_nodes).to(dev, torch.float32)[:, None]
            com_before = torch.zeros((len(cnt), self.num_x_dims), device=dev).index_add_(0, batch_index, molecule["x"]) / cnt
            com_after = torch.zeros((len(cnt), self.num_x_dims), device=dev).index_add_(0, batch_index, xh[:, : self.num_x_dims]) / cnt
            x = xh[:, : self.num_x_dims] + (com_before - com_after)[batch_index]
            one_hot = xh[:, self.num_x_dims:-1] if self.include_charges else xh[:, self.num_x_dims:]
            charges = xh[:, -1:] if self.include_charges else torch.zeros(0, device=dev)
        else:
            raise NotImplementedError(f"ddpm_mode {ddpm_mode!r} is not implemented (reference: 'unconditional' | 'inpainting')")
        atom_types = one_hot.argmax(dim=-1)
        counts = torch.unique_consecutive(batch_index, return_counts=True)[1].tolist()
        mols, o = [], 0
        for n in counts:
            pos, at = x[o:o + n].cpu(), atom_types[o:o + n].cpu()
            ch = charges[o:o + n].cpu() if self.include_charges else None
            mols.append(molecule_builder(pos, at, self.dataset_info) if molecule_builder is not None else (pos, at, ch))
            o += n
        return mols


    @torch.inference_mode()
    def sample_and_save(self, num_samples: int, num_nodes: Optional[torch.Tensor] = None, node_mask: Optional[torch.Tensor] = None,
                        context: Optional[torch.Tensor] = None, num_timesteps: Optional[int] = None, id_from: int = 0,
                        name: str = "molecule", sampling_output_dir: Optional[str] = None,
                        norm_with_original_timesteps: bool = False, **kw) -> None:
        """qm9_mol_gen_ddpm.py:887-947 without the matplotlib / wandb visualisation: sample, write one XYZ file per molecule."""
        x, one_hot, charges, batch_index = self.sample(num_samples, num_nodes=num_nodes, node_mask=node_mask, context=context,
                                                       num_timesteps=num_timesteps, norm_with_original_timesteps=norm_with_original_timesteps, **kw)
        out_dir = str(sampling_output_dir) if sampling_output_dir is not None else os.path.join("sampling_output", "epoch_0")
        save_xyz_file(path=out_dir + "/", positions=x, one_hot=one_hot, charges=charges, dataset_info=self.dataset_info,
                      id_from=id_from, name=name, batch_index=batch_index)

    @torch.inference_mode()
    def sample_chain_and_save(self, keep_frames: int, node_mask: Optional[torch.Tensor] = None, context: Optional[torch.Tensor] = None,
                              id_from: int = 0, num_tries: int = 1, name: str = os.sep + "chain", verbose: bool = True,
                              sampling_output_dir: Optional[str] = None, num_timesteps: Optional[int] = None, **kw) -> bool:
        """qm9_mol_gen_ddpm.py:957-1060 without the matplotlib / wandb visualisation: one molecule sampled with `keep_frames` intermediate
        frames (up to `num_tries` times, until the final molecule is stable -- checked on the device), frames put in generation order, the
        last one repeated 10 times, one XYZ file per frame.  Returns whether the molecule shown is stable."""
        if "QM9" in str(self.dataset_info.get("name", "")).upper():
            num_nodes = torch.tensor([19], dtype=torch.long)
        else:
            num_nodes = self.ddpm.num_nodes_distribution.sample(1)
            assert int(num_nodes.max()) <= self.dataset_info.get("max_n_nodes", int(num_nodes.max()))
        if self.condition_on_context:
            if context is None:
                if self.props_distr is None:
                    raise ValueError("context required (no props_distr attached)")
                context = self.props_distr.sample_batch(num_nodes)
        else:
            context = None
        x = one_hot = charges = None
        mol_stable = False
        for i in range(num_tries):
            chain, _, _ = self.ddpm.mol_gen_sample(num_samples=1, num_nodes=num_nodes, node_mask=node_mask, context=context,
                                                   return_frames=keep_frames, device=self.device, num_timesteps=num_timesteps,
                                                   seed=kw.get("seed", 1234) + i, **{k: v for k, v in kw.items() if k != "seed"})
            chain = chain.reshape(keep_frames, -1, chain.shape[-1]).flip(0)
            chain = torch.cat([chain, chain[-1:].repeat(10, 1, 1)], dim=0)           # repeat the last frame to see the final sample better
            oh_last = chain[-1, :, self.num_x_dims:-1] if self.include_charges else chain[-1, :, self.num_x_dims:]
            res = check_molecular_stability_batch(chain[-1].contiguous(), oh_last.argmax(-1), num_nodes, self.dataset_info)
            mol_stable = bool(int(res[0, 0]))
            x = chain[:, :, : self.num_x_dims]
            oh = chain[:, :, self.num_x_dims:-1] if self.include_charges else chain[:, :, self.num_x_dims:]
            one_hot = torch.nn.functional.one_hot(oh.argmax(-1), num_classes=self.num_atom_types)
            charges = torch.round(chain[:, :, -1:]).long() if self.include_charges else torch.zeros(0, dtype=torch.long, device=self.device)
            if mol_stable:
                break
        F_, n = x.shape[0], x.shape[1]
        out_dir = os.path.join(str(sampling_output_dir) if sampling_output_dir is not None else os.path.join("sampling_output", "epoch_0"), "chain")
        save_xyz_file(path=out_dir, positions=x.reshape(-1, x.shape[-1]), one_hot=one_hot.reshape(-1, one_hot.shape[-1]),
                      charges=charges.reshape(-1, 1) if charges.numel() > 0 else charges, dataset_info=self.dataset_info, id_from=id_from, name=name,
                      batch_index=torch.arange(F_).repeat_interleave(n))
        return mol_stable

    @torch.inference_mode()
    def optimize(self, samples: List[Tuple[torch.Tensor, torch.Tensor]], num_timesteps: int, num_nodes: torch.Tensor,
                 context: Optional[torch.Tensor], node_mask: Optional[torch.Tensor] = None, sampling_output_dir: Optional[str] = None,
                 optim_property: Optional[str] = None, iteration_index: Optional[int] = None, return_frames: int = 1, id_from: int = 0,
                 chain_viz_batch_element_idx: int = 0, name: str = os.sep + "chain", norm_with_original_timesteps: bool = False,
                 verbose: bool = True, **kw) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        """qm9_mol_gen_ddpm.py:636-743: property-guided optimisation of existing samples.  ``return_frames > 1`` (:674-731): the chain of molecule
        ``chain_viz_batch_element_idx`` -- frames in generation order, the last one repeated 10 times -- goes out as one XYZ file per frame under
        ``<sampling_output_dir>/<optim_property>/<time stamp>/iteration_<i>/chain`` (the reference then renders them with matplotlib / imageio:
        visualisation, not built), and the returned x / one_hot are frame 0 = the optimised molecules."""
        if not self.condition_on_context:
            raise Exception("Optimization requires a context conditional to optimize (e.g., `alpha`).")
        if context is None:
            if self.props_distr is None:
                raise ValueError("context required (no props_distr attached)")
            context = self.props_distr.sample_batch(num_nodes)
        xh, batch_index, _ = self.ddpm.mol_gen_optimize(samples=samples, num_nodes=num_nodes, node_mask=node_mask, context=context,
                                                        device=self.device, num_timesteps=num_timesteps, return_frames=return_frames,
                                                        norm_with_original_timesteps=norm_with_original_timesteps, **kw)
        if return_frames > 1:
            assert all(p is not None for p in (sampling_output_dir, optim_property, iteration_index)), \
                "Required parameters must be provided to visualize optimized molecules."
            import time as _time
            chain = xh[:, batch_index == chain_viz_batch_element_idx, :].flip(0)              # reverse_tensor (:679-680)
            chain = torch.cat([chain, chain[-1:].repeat(10, 1, 1)], dim=0)
            xs = chain[:, :, : self.num_x_dims]
            oh = chain[:, :, self.num_x_dims:-1] if self.include_charges else chain[:, :, self.num_x_dims:]
            one_hot_c = torch.nn.functional.one_hot(oh.argmax(-1), num_classes=self.num_atom_types)
            n_sel = torch.tensor([xs.shape[1]], dtype=torch.long)
            res = check_molecular_stability_batch(chain[-1].contiguous(), oh[-1].argmax(-1), n_sel, self.dataset_info)
            if verbose:
                log.info("Found stable molecule to visualize :)" if bool(int(res[0, 0])) else "Did not find stable molecule to visualize :(")
            out_dir = os.path.join(str(sampling_output_dir), str(optim_property), _time.strftime("%Y%m%d-%H%M%S"), f"iteration_{iteration_index}", "chain")
            save_xyz_file(path=out_dir, positions=xs.reshape(-1, xs.shape[-1]), one_hot=one_hot_c.reshape(-1, one_hot_c.shape[-1]),
                          charges=torch.tensor([]), dataset_info=self.dataset_info, id_from=id_from, name=name,
                          batch_index=torch.arange(xs.shape[0]).repeat_interleave(xs.shape[1]))
            self.last_chain_dir = out_dir
            x = xh[0, :, : self.num_x_dims]
            one_hot = xh[0, :, self.num_x_dims:-1] if self.include_charges else xh[0, :, self.num_x_dims:]
            charges = torch.round(chain[:, :, -1:]).long() if self.include_charges else torch.zeros(0, dtype=torch.long, device=self.device)   # (:705-709)
            return x, one_hot, charges, batch_index
        x = xh[:, : self.num_x_dims]
        one_hot = xh[:, self.num_x_dims:-1] if self.include_charges else xh[:, self.num_x_dims:]
        charges = xh[:, -1:] if self.include_charges else torch.zeros(0, device=self.device)
        return x, one_hot, charges, batch_index

    @torch.inference_mode()
    def sample_and_analyze(self, num_samples: int, node_mask: Optional[torch.Tensor] = None, context: Optional[torch.Tensor] = None,
                           batch_size: Optional[int] = None, max_num_nodes: Optional[int] = 100, num_timesteps: Optional[int] = None,
                           concurrent_batches: int = 1, **kw) -> Dict[str, Any]:
        """qm9_mol_gen_ddpm.py:747-843 -- the evaluation driver: batches of `batch_size` molecules with sizes drawn from the
        dataset histogram, one 1000-step sampling run per batch, stability statistics.  The per-molecule host loop of the
        reference (`check_molecular_stability` on CPU copies) is one device launch per batch here; only three integers per
        molecule and the atom-type histogram ever leave the GPU.  `concurrent_batches > 1` keeps that many batches in flight on
        separate handles / streams (`mol_gen_sample_concurrent`): same samples, better use of the chip at the reference's batch
        size of 100.  `save_molecules` (xyz files) is `sample_and_save`."""
        max_num_nodes = self.dataset_info.get("max_n_nodes", max_num_nodes)
        batch_size = int(cfg_get(self._init_kwargs["dataloader_cfg"], "batch_size", 64)) if batch_size is None else batch_size
        batch_size = min(batch_size, num_samples)
        results, type_counts = [], torch.zeros(self.num_atom_types, dtype=torch.int64)
        plan, done = [], 0
        while done < num_samples:                               # draw all sizes first: the same random stream as the sequential driver
            nb = min(batch_size, num_samples - done)
            num_nodes = self.ddpm.num_nodes_distribution.sample(nb)
            assert int(num_nodes.max()) <= max_num_nodes
            ctx = context
            if self.condition_on_context and ctx is None:
                if self.props_distr is None:
                    raise ValueError("context required (no props_distr attached)")
                ctx = self.props_distr.sample_batch(num_nodes)
            plan.append((nb, num_nodes, ctx if self.condition_on_context else None))
            done += nb

        def account(xh, num_nodes):
            oh = xh[:, self.num_x_dims:-1] if self.include_charges else xh[:, self.num_x_dims:]
            atom_types = oh.argmax(-1)
            results.append(check_molecular_stability_batch(xh, atom_types, num_nodes, self.dataset_info))
            type_counts.add_(torch.bincount(atom_types, minlength=self.num_atom_types).cpu())

        K = max(1, int(concurrent_batches))
        for i0 in range(0, len(plan), K):
            chunk = plan[i0:i0 + K]
            if K == 1:
                nb, num_nodes, ctx = chunk[0]
                xh, _, _ = self.ddpm.mol_gen_sample(num_samples=nb, num_nodes=num_nodes, node_mask=node_mask, context=ctx, device=self.device,
                                                    num_timesteps=num_timesteps, seed=1234 + i0, **kw)
                account(xh, num_nodes)
            else:
                outs = self.ddpm.mol_gen_sample_concurrent([c[1] for c in chunk], self.device, num_timesteps=num_timesteps,
                                                           contexts=[c[2] for c in chunk], seeds=[1234 + i0 + j for j in range(len(chunk))])
                for (xh, _, _), c in zip(outs, chunk):
                    account(xh, c[1])
        return self.analyze_samples(torch.cat(results).cpu(), type_counts)

    def analyze_samples(self, stability: torch.Tensor, atom_type_counts: torch.Tensor) -> Dict[str, Any]:
        """qm9_mol_gen_ddpm.py:846-885 on the device results: `stability` int [M, 3] rows (stable, nr_stable_atoms, n)."""
        st = stability.to(torch.int64)
        q = atom_type_counts.to(torch.float64).numpy()
        p = self.node_type_distribution.p
        with np.errstate(divide="ignore", invalid="ignore"):
            kl = float(-np.sum(p * np.log(q / q.sum() / p + self.node_type_distribution.EPS)))
        out = {
            "kl_div_atom_types": kl,
            "mol_stable": float(st[:, 0].sum()) / float(len(st)),
            "atm_stable": float(st[:, 1].sum()) / float(st[:, 2].sum()),
            "validity": None, "uniqueness": None, "novelty": None,    # RDKit metrics (BasicMolecularMetrics) are outside the path
        }
        return out


class QM9MoleculeGenerationDDPM(_MoleculeGenerationDDPM):
    """``_target_`` stand-in for src.models.qm9_mol_gen_ddpm.QM9MoleculeGenerationDDPM."""
    _DATASETS = {"QM9": "qm9", "QM9_second_half": "qm9_second_half"}


class GEOMMoleculeGenerationDDPM(_MoleculeGenerationDDPM):
    """``_target_`` stand-in for src.models.geom_mol_gen_ddpm.GEOMMoleculeGenerationDDPM."""
    _DATASETS = {"GEOM": "geom"}


def sample_sweep_conditionally(model: Any, props_distr: Any, num_nodes: int = 19, num_frames: int = 100
                               ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """src/models/__init__.py:200-226: `num_frames` molecules of `num_nodes` atoms whose (normalised) property context sweeps linearly from the
    smallest to the largest value seen for that size, all drawn with the same noise (`fix_noise=True`: option "fix_noise" of the library)."""
    num_nodes_ = torch.tensor([num_nodes] * num_frames, device=model.device)
    cols = []
    for key in props_distr.distributions:
        lo, hi = props_distr.distributions[key][num_nodes]["params"]
        mean, mad = props_distr.normalizer[key]["mean"], props_distr.normalizer[key]["mad"]
        lo_n, hi_n = float((torch.as_tensor(lo) - mean) / mad), float((torch.as_tensor(hi) - mean) / mad)
        cols.append(torch.tensor(np.linspace(lo_n, hi_n, num_frames)).unsqueeze(1))
    context = torch.cat(cols, dim=-1).float().to(model.device)
    return model.sample(num_samples=num_frames, num_nodes=num_nodes_, context=context, fix_noise=True)
