"""Configuration surface of the hot path.

The reference composes five Hydra groups (configs/model/{model_cfg,module_cfg,layer_cfg,diffusion_cfg}/*.yaml and
configs/datamodule/dataloader_cfg/*.yaml) into attribute-access DictConfigs and passes them as keyword
arguments (src/models/qm9_mol_gen_ddpm.py:125-131).  The same five groups, with the same key names, are
accepted here as any mapping / attribute object; ``default_cfgs`` supplies the production values and
``load_cfg_tree`` reads a reference-style ``configs/`` directory with plain PyYAML (coercing the ``1e-5``
style strings that OmegaConf would parse as floats).
"""
from __future__ import annotations

import copy
import json
import os
from typing import Any, Dict, Iterable, Optional


class AttrDict(dict):
    """dict with attribute access (the subset of omegaconf.DictConfig the hot path relies on)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_attr(obj):
    if isinstance(obj, dict):
        return AttrDict({k: to_attr(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_attr(v) for v in obj]
    return obj


def cfg_get(cfg: Any, key: str, default: Any = None) -> Any:
    """Reads ``key`` from a DictConfig / dict / namespace alike."""
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    try:
        return getattr(cfg, key)
    except Exception:
        try:
            return cfg[key]
        except Exception:
            return default


_MODEL = {
    "qm9": dict(chi_input_dim=2, e_input_dim=1, xi_input_dim=1, h_hidden_dim=256, chi_hidden_dim=32, e_hidden_dim=64,
                xi_hidden_dim=16, num_encoder_layers=9, num_decoder_layers=3, dropout=0.0),
    "geom": dict(chi_input_dim=2, e_input_dim=1, xi_input_dim=1, h_hidden_dim=256, chi_hidden_dim=32, e_hidden_dim=16,
                 xi_hidden_dim=8, num_encoder_layers=4, num_decoder_layers=3, dropout=0.0),
}
_MODULE = dict(norm_x_diff=True, scalar_gate=0, vector_gate=True, vector_residual=False, vector_frame_residual=False,
               frame_gate=False, sigma_frame_gate=False, scalar_nonlinearity="silu", vector_nonlinearity="silu",
               nonlinearities=["silu", "silu"], bottleneck=4, vector_linear=True, vector_identity=True,
               default_vector_residual=False, default_bottleneck=4, node_positions_weight=1.0,
               update_positions_with_vector_sum=False, ablate_frame_updates=False, ablate_scalars=False,
               ablate_vectors=False, conditioning=[], clip_gradients=True, log_grad_flow_steps=500)
_LAYER = dict(pre_norm=False, use_gcp_norm=False, use_gcp_dropout=False, use_scalar_message_attention=True,
              num_feedforward_layers=1, dropout=0.0, nonlinearity_slope=1e-2,
              mp_cfg=dict(edge_encoder=False, edge_gate=False, num_message_layers=4, message_residual=0,
                          message_ff_multiplier=1, self_message=True, use_residual_message_gcp=True))
_DIFFUSION = dict(ddpm_mode="unconditional", dynamics_network="gcpnet", diffusion_target="atom_types_and_coords",
                  num_timesteps=1000, parametrization="eps", noise_schedule="polynomial_2", noise_precision=1e-5,
                  loss_type="l2", norm_values=[1.0, 4.0, 10.0], norm_biases=[None, 0.0, 0.0], condition_on_time=True,
                  self_condition=False, norm_training_by_max_nodes=False, sample_during_training=True, eval_epochs=20,
                  visualize_sample_epochs=20, visualize_chain_epochs=20, num_eval_samples=1000, eval_batch_size=100,
                  num_visualization_samples=5, keep_frames=100)
_DATALOADER = {
    "qm9": dict(dataset="QM9", num_atom_types=5, num_x_dims=3, remove_h=False, include_charges=True, num_radials=1,
                batch_size=64, smiles_filepath=None, data_dir=None),
    "geom": dict(dataset="GEOM", num_atom_types=16, num_x_dims=3, remove_h=False, include_charges=False, num_radials=1,
                 batch_size=64, smiles_filepath=None, data_dir=None),
}


def default_cfgs(dataset: str = "qm9", conditioning: Iterable[str] = ()) -> Dict[str, AttrDict]:
    """Production values of the five config groups for ``dataset`` in {"qm9", "geom"}.

    ``conditioning`` (e.g. ("alpha",)) applies the property-conditional overrides of
    configs/experiment/qm9_mol_gen_conditional_ddpm.yaml:88,116,125.
    """
    dataset = dataset.lower()
    if dataset not in _MODEL:
        raise ValueError(f"unknown dataset {dataset!r}")
    cfgs = dict(model_cfg=copy.deepcopy(_MODEL[dataset]), module_cfg=copy.deepcopy(_MODULE), layer_cfg=copy.deepcopy(_LAYER),
                diffusion_cfg=copy.deepcopy(_DIFFUSION), dataloader_cfg=copy.deepcopy(_DATALOADER[dataset]))
    if dataset == "geom":   # evaluation cadence only; no effect on the sampling path
        cfgs["diffusion_cfg"].update(eval_epochs=1, visualize_sample_epochs=1, visualize_chain_epochs=1, num_eval_samples=500)
    conditioning = list(conditioning)
    if conditioning:
        cfgs["module_cfg"]["conditioning"] = conditioning
        cfgs["dataloader_cfg"]["include_charges"] = False
        cfgs["dataloader_cfg"]["dataset"] = "QM9_second_half"
        cfgs["diffusion_cfg"]["norm_values"] = [1.0, 8.0, 1.0]
    return {k: to_attr(v) for k, v in cfgs.items()}


def _coerce(v):
    if isinstance(v, str):
        try:
            return float(v)
        except ValueError:
            return v
    if isinstance(v, dict):
        return {k: _coerce(x) for k, x in v.items()}
    if isinstance(v, list):
        return [_coerce(x) for x in v]
    return v


def load_cfg_tree(config_root: str, dataset: str = "qm9", conditioning: Iterable[str] = ()) -> Dict[str, AttrDict]:
    """Reads the five groups from a reference-layout ``configs/`` directory (values override the defaults)."""
    import yaml

    tag = dataset.lower()
    cfgs = default_cfgs(tag, conditioning)

    def rd(rel):
        path = os.path.join(config_root, rel)
        if not os.path.exists(path):
            return {}
        with open(path) as f:
            d = yaml.safe_load(f) or {}
        d.pop("defaults", None)
        return {k: v for k, v in _coerce(d).items() if not (isinstance(v, str) and "${" in v)}

    cfgs["model_cfg"].update(to_attr(rd(f"model/model_cfg/{tag}_mol_gen_ddpm_gcp_model.yaml")))
    mod = rd(f"model/module_cfg/{tag}_mol_gen_ddpm_gcp_module.yaml")
    sel = mod.pop("selected_GCP", None)
    if sel is not None:          # configs/model/module_cfg/*.yaml: {_target_: src.models.components.gcpnet.GCP2, _partial_: true}
        target = sel.get("_target_", "") if isinstance(sel, dict) else str(getattr(sel, "__name__", sel))
        name = target.rsplit(".", 1)[-1]
        if name not in ("GCP", "GCP2"):
            raise NotImplementedError(f"module_cfg.selected_GCP = {target!r}: expected GCP or GCP2 (gcpnet.py:33, 265)")
        mod["selected_GCP"] = name   # GCP2 -> the fused kernels (production flags) or the module path; GCP -> the module path
    mod.pop("nonlinearities", None)
    keep_cond = cfgs["module_cfg"]["conditioning"]
    cfgs["module_cfg"].update(to_attr(mod))
    cfgs["module_cfg"]["conditioning"] = keep_cond or cfgs["module_cfg"].get("conditioning", [])
    cfgs["module_cfg"]["nonlinearities"] = [cfgs["module_cfg"]["scalar_nonlinearity"], cfgs["module_cfg"]["vector_nonlinearity"]]
    cfgs["layer_cfg"].update(to_attr(rd(f"model/layer_cfg/{tag}_mol_gen_ddpm_gcp_interaction_layer.yaml")))
    cfgs["layer_cfg"]["mp_cfg"].update(to_attr(rd(f"model/layer_cfg/mp_cfg/{tag}_mol_gen_ddpm_gcp_mp.yaml")))
    dif = rd(f"model/diffusion_cfg/{tag}_mol_gen_ddpm.yaml")
    if list(conditioning):
        dif.pop("norm_values", None)
    cfgs["diffusion_cfg"].update(to_attr(dif))
    dl = rd(f"datamodule/dataloader_cfg/edm_{tag}_dataloader.yaml")
    if list(conditioning):
        dl.pop("include_charges", None)
        dl.pop("dataset", None)
    cfgs["dataloader_cfg"].update(to_attr(dl))
    return cfgs


_INFO_CACHE: Optional[dict] = None


def dataset_info(name: str) -> Dict[str, Any]:
    """Atom vocabulary and node-count histogram consulted by sampling (the only parts of
    src/datamodules/components/edm/datasets_config.py the sampler touches); keys follow the reference dicts."""
    global _INFO_CACHE
    if _INFO_CACHE is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "dataset_info.json")) as f:
            _INFO_CACHE = json.load(f)
    key = {"qm9": "qm9", "qm9_second_half": "qm9_second_half", "geom": "geom"}[name.lower()]
    d = _INFO_CACHE[key]
    dec = d["atom_decoder"]
    return {
        "name": d["name"], "with_h": d["with_h"], "max_n_nodes": d["max_n_nodes"], "atom_decoder": dec,
        "atom_encoder": {a: i for i, a in enumerate(dec)},
        "n_nodes": {int(k): int(v) for k, v in d["n_nodes_hist"]},
        "atom_types": {i: c for i, c in enumerate(d["atom_type_counts"])},
    }
