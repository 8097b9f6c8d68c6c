"""Import the *reference* hot-path modules in THIS container (CPU, no GPU) under stub modules.

Test infrastructure only.  Used by ``make_golden.py`` (fixture generator) and by the
``needs_reference`` tests; it never ships to the GPU box (``/root/reference`` does not exist
there) and nothing in the product package imports it.

The reference is imported *unmodified* from ``/root/reference``; only third-party modules that
are absent from this image are replaced.  The stand-ins whose arithmetic matters:

* ``torch_scatter.scatter(src, index, dim=0, dim_size, reduce in {sum, mean})`` -> ``index_add_``
  (mean = sum / clamp(count, 1)), i.e. pytorch-scatter 2.1.0 semantics
  (reference pin: environment.yaml:227).
* ``torch_geometric.data.Batch`` -> attribute bag with ``__getitem__`` and ``num_nodes``.
* ``omegaconf.DictConfig`` -> dict with attribute access.
Everything else (lightning, wandb, rdkit, ...) is inert.
"""
from __future__ import annotations

import copy as _copy
import os
import sys
import types
from unittest.mock import MagicMock

import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("GCDM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "models", "components"))


class DictConfig(dict):
    """dict with attribute access (stands in for omegaconf.DictConfig)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # pragma: no cover
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def __copy__(self):
        return DictConfig(self)

    def __deepcopy__(self, memo):
        return DictConfig({k: _copy.deepcopy(v, memo) for k, v in self.items()})

    def copy(self):
        return DictConfig(self)


def to_dictconfig(obj):
    if isinstance(obj, dict):
        return DictConfig({k: to_dictconfig(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_dictconfig(v) for v in obj]
    return obj


class _Batch:
    """Attribute bag standing in for torch_geometric.data.Batch."""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def __contains__(self, k):
        return hasattr(self, k)

    @property
    def num_nodes(self):
        return self.x.shape[0]


def _scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    assert dim == 0 and out is None
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() else 0
    res = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
        cnt = cnt.clamp(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)


def _mock_module(name, **attrs):
    m = MagicMock(name=name)
    m.__name__ = name
    m.__path__ = []
    m.__spec__ = None
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


_INSTALLED = False


def install_stubs():
    global _INSTALLED
    if _INSTALLED:
        return
    sys.dont_write_bytecode = True
    ident = lambda f=None, *a, **k: f if callable(f) else (lambda g: g)

    # --- inert modules -------------------------------------------------------------------------
    for name in [
        "imageio", "wandb", "wandb.sdk", "wandb.sdk.wandb_run", "prody", "rdkit", "rdkit.Chem",
        "rdkit.Chem.rdchem", "rdkit.Chem.AllChem", "rdkit.Geometry", "pymol", "torchviz", "hydra",
        "hydra.core", "hydra.core.hydra_config", "hydra.utils", "torch_cluster", "torchmetrics",
        "pyrootutils", "dotenv", "openbabel", "posebusters", "rich", "rich.syntax", "rich.tree",
        "rich.prompt",
    ]:
        if name not in sys.modules:
            sys.modules[name] = _mock_module(name)
    sys.modules["wandb.sdk.wandb_run"].Run = type("Run", (), {})

    # --- pytorch_lightning with real base classes ----------------------------------------------
    pl = _mock_module("pytorch_lightning")
    pl.LightningModule = type("LightningModule", (nn.Module,), {})
    pl.LightningDataModule = type("LightningDataModule", (), {})
    pl.Callback = type("Callback", (), {})
    pl.Trainer = type("Trainer", (), {})
    cb = _mock_module("pytorch_lightning.callbacks")
    cb.ModelCheckpoint = type("ModelCheckpoint", (pl.Callback,), {})
    cb.Callback = pl.Callback
    util = _mock_module("pytorch_lightning.utilities")
    util.rank_zero_only = ident
    exc = _mock_module("pytorch_lightning.utilities.exceptions")
    exc.MisconfigurationException = type("MisconfigurationException", (Exception,), {})
    rz = _mock_module("pytorch_lightning.utilities.rank_zero")
    rz.rank_zero_only = ident
    rz.rank_zero_info = lambda *a, **k: None
    types_ = _mock_module("pytorch_lightning.utilities.types")
    loggers = _mock_module("pytorch_lightning.loggers")
    loggers.LightningLoggerBase = type("LightningLoggerBase", (), {})
    loggers.Logger = loggers.LightningLoggerBase
    lw = _mock_module("pytorch_lightning.loggers.wandb")
    for n, m in {
        "pytorch_lightning": pl, "pytorch_lightning.callbacks": cb, "pytorch_lightning.utilities": util,
        "pytorch_lightning.utilities.exceptions": exc, "pytorch_lightning.utilities.rank_zero": rz,
        "pytorch_lightning.utilities.types": types_, "pytorch_lightning.loggers": loggers,
        "pytorch_lightning.loggers.wandb": lw,
    }.items():
        sys.modules[n] = m

    # --- torchtyping / typeguard ---------------------------------------------------------------
    tt = types.ModuleType("torchtyping")

    class _TT:
        def __class_getitem__(cls, item):
            return torch.Tensor

    tt.TensorType = _TT
    tt.patch_typeguard = lambda *a, **k: None
    sys.modules["torchtyping"] = tt
    tg = types.ModuleType("typeguard")
    tg.typechecked = ident
    sys.modules["typeguard"] = tg

    # --- torch_scatter -------------------------------------------------------------------------
    ts = types.ModuleType("torch_scatter")
    ts.scatter = _scatter
    sys.modules["torch_scatter"] = ts

    # --- torch_geometric -----------------------------------------------------------------------
    tgm = types.ModuleType("torch_geometric")
    tgm.__path__ = []
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Batch = _Batch
    tgd.Data = _Batch
    tgd.Dataset = type("Dataset", (), {})
    tgm.data = tgd
    tgn = types.ModuleType("torch_geometric.nn")
    tgn.MessagePassing = type("MessagePassing", (nn.Module,), {})
    tgt = types.ModuleType("torch_geometric.typing")
    for n in ("Adj", "Size", "OptTensor", "Tensor"):
        setattr(tgt, n, torch.Tensor)
    tgl = types.ModuleType("torch_geometric.loader")
    tgl.DataLoader = type("DataLoader", (), {})
    for n, m in {"torch_geometric": tgm, "torch_geometric.data": tgd, "torch_geometric.nn": tgn,
                 "torch_geometric.typing": tgt, "torch_geometric.loader": tgl}.items():
        sys.modules[n] = m

    # --- omegaconf -----------------------------------------------------------------------------
    oc = types.ModuleType("omegaconf")
    oc.DictConfig = DictConfig

    class OmegaConf:
        @staticmethod
        def to_container(cfg, **kw):
            return dict(cfg)

        @staticmethod
        def create(x):
            return to_dictconfig(x)

    import contextlib

    oc.OmegaConf = OmegaConf
    oc.open_dict = lambda cfg: contextlib.nullcontext(cfg)
    sys.modules["omegaconf"] = oc

    # --- matplotlib private path moved in mpl >= 3.7 -------------------------------------------
    import matplotlib.axes

    sub = types.ModuleType("matplotlib.axes._subplots")
    sub.Axes = matplotlib.axes.Axes
    sub.AxesSubplot = matplotlib.axes.Axes
    sys.modules["matplotlib.axes._subplots"] = sub

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def import_reference():
    """Returns (gcpnet_module, variational_diffusion_module, components_module)."""
    install_stubs()
    import importlib

    comps = importlib.import_module("src.models.components")
    vd = importlib.import_module("src.models.components.variational_diffusion")
    gcp = importlib.import_module("src.models.components.gcpnet")
    return gcp, vd, comps


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


def load_reference_cfgs(dataset: str, conditioning=()):
    """Reads the reference YAML tree with PyYAML (coercing '1e-5'-style strings)."""
    import yaml

    gcp, _, _ = import_reference()
    tag = {"qm9": "qm9", "geom": "geom"}[dataset]
    root = os.path.join(REFERENCE_ROOT, "configs")

    def rd(rel):
        with open(os.path.join(root, rel)) as f:
            d = yaml.safe_load(f)
        d.pop("defaults", None)
        return _coerce(d)

    model_cfg = rd(f"model/model_cfg/{tag}_mol_gen_ddpm_gcp_model.yaml")
    module_cfg = rd(f"model/module_cfg/{tag}_mol_gen_ddpm_gcp_module.yaml")
    layer_cfg = rd(f"model/layer_cfg/{tag}_mol_gen_ddpm_gcp_interaction_layer.yaml")
    layer_cfg["mp_cfg"] = rd(f"model/layer_cfg/mp_cfg/{tag}_mol_gen_ddpm_gcp_mp.yaml")
    diffusion_cfg = rd(f"model/diffusion_cfg/{tag}_mol_gen_ddpm.yaml")
    dataloader_cfg = rd(f"datamodule/dataloader_cfg/edm_{tag}_dataloader.yaml")
    module_cfg["selected_GCP"] = gcp.GCP2
    module_cfg["nonlinearities"] = [module_cfg["scalar_nonlinearity"], module_cfg["vector_nonlinearity"]]
    module_cfg["conditioning"] = list(conditioning)
    if conditioning:  # configs/experiment/qm9_mol_gen_conditional_ddpm.yaml:88,116,125
        dataloader_cfg["include_charges"] = False
        diffusion_cfg["norm_values"] = [1.0, 8.0, 1.0]
    for k in ("visualize_sample_epochs", "visualize_chain_epochs"):
        diffusion_cfg[k] = diffusion_cfg.get("eval_epochs", 20)
    cfgs = dict(model_cfg=model_cfg, module_cfg=module_cfg, layer_cfg=layer_cfg,
                diffusion_cfg=diffusion_cfg, dataloader_cfg=dataloader_cfg)
    return {k: to_dictconfig(v) for k, v in cfgs.items()}


def shrink_cfgs(cfgs, h=32, chi=8, e=16, xi=4, layers=2):
    """Reduced-width variant (same code path, smaller fixtures)."""
    cfgs = _copy.deepcopy(cfgs)
    m = cfgs["model_cfg"]
    m.h_hidden_dim, m.chi_hidden_dim, m.e_hidden_dim, m.xi_hidden_dim = h, chi, e, xi
    m.num_encoder_layers = layers
    return cfgs


def build_reference_dynamics(cfgs, seed=0, weight_scale=1.0, dtype=torch.float32):
    gcp, _, _ = import_reference()
    torch.manual_seed(seed)
    net = gcp.GCPNetDynamics(**{k: _copy.deepcopy(v) for k, v in cfgs.items()})
    if weight_scale != 1.0:
        with torch.no_grad():
            for p in net.parameters():
                if p.dim() == 2:
                    p.mul_(weight_scale)
    net = net.to(dtype).eval()
    return net


def dataset_info(dataset: str, conditioning=()):
    install_stubs()
    import importlib

    dc = importlib.import_module("src.datamodules.components.edm.datasets_config")
    if dataset == "qm9":
        return dc.QM9_SECOND_HALF if conditioning else dc.QM9_WITH_H
    return dc.GEOM_WITH_H


def build_reference_ddpm(cfgs, net, dataset="qm9"):
    _, vd, _ = import_reference()
    import io
    import contextlib

    with contextlib.redirect_stdout(io.StringIO()):
        ddpm = vd.EquivariantVariationalDiffusion(
            dynamics_network=net, diffusion_cfg=_copy.deepcopy(cfgs["diffusion_cfg"]),
            dataloader_cfg=_copy.deepcopy(cfgs["dataloader_cfg"]),
            dataset_info=dataset_info(dataset, cfgs["module_cfg"]["conditioning"]))
    return ddpm.eval()


class NoiseTape:
    """Replaces ``torch.randn`` by draws from a seeded generator (fp32, cast to default dtype)."""

    def __init__(self, seed=1234):
        self.gen = torch.Generator().manual_seed(seed)
        self.calls = []

    def __enter__(self):
        self._orig = torch.randn

        def randn(*size, **kw):
            if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
                size = tuple(size[0])
            x = self._orig(size, generator=self.gen, dtype=torch.float32)
            self.calls.append(tuple(size))
            return x.to(torch.get_default_dtype())

        torch.randn = randn
        return self

    def __exit__(self, *a):
        torch.randn = self._orig


def make_batch(batch_index, mask, props_context=None):
    install_stubs()
    return _Batch(batch=batch_index, mask=mask, props_context=props_context)
