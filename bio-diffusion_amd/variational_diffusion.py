"""Ancestral DDPM sampler around the MI355X dynamics network.

Mirror of the *sampling subset* of ``src/models/components/variational_diffusion.py``:
``PredefinedNoiseSchedule`` (:206-255), ``EquivariantVariationalDiffusion`` with ``sigma/alpha/SNR`` (:318-338),
``sigma_and_alpha_t_given_s`` (:342-367), ``sample_combined_position_feature_noise`` (:795-819),
``sample_normal`` (:822-837), ``sample_p_zs_given_zt`` (:1204-1278), ``sample_p_xh_given_z0`` (:840-907) and
``mol_gen_sample`` (:1282-1412), plus ``NumNodesDistribution`` (src/models/__init__.py:264-308).
RePaint inpainting (``inpaint``, :1582-1789) and the property-guided optimisation loop (``mol_gen_optimize``, :1416-1546) are built
below on the same step / decode entry points.  ``forward`` (:948-1160) is built for EVALUATION mode -- the validation / test likelihood terms,
two evaluations of the network per batch; the training objective needs the backward pass of the network and raises.

Two ways to take a step:
  * ``sample_p_zs_given_zt`` -- the reference's method signature, torch ops on the device for the O(N) algebra
    and the HIP dynamics forward for the network call (used by the teacher-forced parity tests);
  * ``mol_gen_sample`` -- the production loop: ``gcdm_sample_init / gcdm_sample_step / gcdm_sample_final``
    (one fused HIP kernel per step after the network kernels; noise from a tape or on-device Philox;
    host-side asserts replaced by a device flag word read once at the end).
"""
from __future__ import annotations

import ctypes as C
import os
import logging
from random import random as _random
from typing import Any, Callable, Dict, List, Optional, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _native
from .config import AttrDict, cfg_get
from .gcpnet import F16RangeError

log = logging.getLogger(__name__)


def num_nodes_to_batch_index(num_samples: int, num_nodes, device) -> torch.Tensor:
    """src/models/components/__init__.py:314-321."""
    assert isinstance(num_nodes, int) or len(num_nodes) == num_samples
    idx = torch.arange(num_samples, device=device)
    return torch.repeat_interleave(idx, num_nodes if isinstance(num_nodes, int) else num_nodes.to(device))


def inflate_batch_array(array: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """src/models/__init__.py:80-86."""
    return array.view((array.shape[0],) + (1,) * (len(target.shape) - 1))


def _segment_mean_sub(x: torch.Tensor, batch_index: torch.Tensor, num_graphs: int, mask: torch.Tensor) -> torch.Tensor:
    """centralize(..., edm=True) (components/__init__.py:45-98) without torch_scatter."""
    mf = mask.to(x.dtype)
    cnt = torch.zeros(num_graphs, dtype=x.dtype, device=x.device).index_add_(0, batch_index, mf).unsqueeze(-1)
    s = torch.zeros(num_graphs, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, batch_index, x)
    return x - (s / cnt)[batch_index] * mf.unsqueeze(-1)


def polynomial_gamma(num_timesteps: int, noise_precision: float, power: float) -> np.ndarray:
    """gamma table of the "polynomial_<power>" schedule in float64 (variational_diffusion.py:67-107, 237-246)."""
    steps = num_timesteps + 1
    x = np.linspace(0, steps, steps)
    a2 = (1 - np.power(x / steps, power)) ** 2
    a2 = np.concatenate([np.ones(1), a2], axis=0)
    a2 = np.cumprod(np.clip(a2[1:] / a2[:-1], a_min=0.001, a_max=1.0), axis=0)
    a2 = (1 - 2 * noise_precision) * a2 + noise_precision
    return -(np.log(a2) - np.log(1 - a2))


def cosine_gamma(num_timesteps: int, s: float = 0.008) -> np.ndarray:
    """gamma table of the "cosine" schedule in float64 (cosine_beta_schedule + PredefinedNoiseSchedule, variational_diffusion.py:37-58, 220-243)."""
    steps = num_timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    a2 = np.cumprod(1.0 - np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999), axis=0)
    return -(np.log(a2) - np.log(1 - a2))


def slice_cuts(num_nodes: torch.Tensor, K: int) -> List[int]:
    """Molecule indices [c_0 = 0, c_1, ..., c_K = B] that cut a flat batch into K contiguous, non-empty slices of roughly equal work
    (edges, i.e. sum of n^2) -- used when one batch is sampled on K handles / streams."""
    nn_ = torch.as_tensor(num_nodes).long().cpu()
    Bm = len(nn_)
    if not 1 <= K <= Bm:
        raise ValueError(f"cannot cut {Bm} molecules into {K} non-empty slices")
    work_cum = (nn_ ** 2).cumsum(0)
    cuts = [0]
    # (round 5 measured unequal cuts -- the first slice's tile count a multiple of the persistent workgroup count, 0.4873 / 0.5127 / 0.45 / 0.531 of the work:
    #  6.93 / 6.92 / 6.95 / 6.85 ms per step against 6.88 for the equal cut, profiles/r05_slices.txt: the dispatcher already fills one slice's tails with the other)
    for k in range(1, K):
        c = int(torch.searchsorted(work_cum, work_cum[-1] * k // K).item()) + 1
        cuts.append(min(max(c, cuts[-1] + 1), Bm - (K - k)))
    cuts.append(Bm)
    return cuts


def repaint_schedule(resamplings: int, jump_length: int, num_timesteps: int) -> List[int]:
    """RePaint schedule (variational_diffusion.py:1548-1578, `get_repaint_schedule`): the number of denoising steps to apply before each
    jump back, from t = T downwards.  T is cut into ``(T-1) // jump_length`` stretches of ``jump_length`` steps plus a remainder; every
    stretch is visited ``resamplings`` times, the last visit running on through the following stretch, and the remainder joins the last
    entry.  (With one resampling there is never a jump: the schedule is [T].)"""
    if num_timesteps <= 0:
        return []
    if jump_length <= 0:
        raise ValueError("jump_length must be positive")
    stretches = (num_timesteps - 1) // jump_length
    rest = num_timesteps - stretches * jump_length
    if stretches == 0 or resamplings <= 0:
        return [rest]
    if resamplings == 1:
        return [num_timesteps]
    low_to_high = [jump_length] * (resamplings - 1)
    low_to_high += ([2 * jump_length] + [jump_length] * (resamplings - 2)) * (stretches - 1)
    low_to_high.append(jump_length + rest)
    return low_to_high[::-1]


class PredefinedNoiseSchedule(nn.Module):
    def __init__(self, noise_schedule: str, num_timesteps: int, noise_precision: float, verbose: bool = False, **kwargs):
        super().__init__()
        self.timesteps = num_timesteps
        if noise_schedule == "cosine":
            g = cosine_gamma(num_timesteps)
        elif "polynomial" in noise_schedule:
            splits = noise_schedule.split("_")
            assert len(splits) == 2
            g = polynomial_gamma(num_timesteps, float(noise_precision), float(splits[1]))
        else:
            raise ValueError(noise_schedule)
        self.gamma = nn.Parameter(torch.tensor(g).float(), requires_grad=False)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        return self.gamma[torch.round(t * self.timesteps).long()]


class NumNodesDistribution(nn.Module):
    """src/models/__init__.py:264-308."""

    def __init__(self, histogram: Dict[int, int], verbose: bool = False, eps: float = 1e-30):
        super().__init__()
        self.eps = eps
        sizes = list(histogram)                                  # insertion order = the order of the checkpoint's buffers
        counts = torch.tensor([histogram[n] for n in sizes])
        self.keys = {n: i for i, n in enumerate(sizes)}
        self.register_buffer("num_nodes", torch.tensor(sizes))
        self.register_buffer("prob", counts / counts.sum())      # same state-dict entries as the reference: sizes and normalised frequencies

    @property
    def m(self) -> torch.distributions.Categorical:
        """Built from the CURRENT `prob` buffer (a checkpoint may have replaced it after construction) and on the CPU, so that a torch seed
        draws the sizes the reference draws (its Categorical is created before the module moves to the GPU)."""
        return torch.distributions.Categorical(self.prob.detach().cpu())

    def sample(self, n_samples: int = 1) -> torch.Tensor:
        idx = self.m.sample((n_samples,))
        return self.num_nodes[idx.to(self.num_nodes.device)]

    def log_prob(self, batch_n_nodes: torch.Tensor) -> torch.Tensor:
        idcs = torch.tensor([self.keys[n] for n in batch_n_nodes.tolist()], device=batch_n_nodes.device)        # one host copy, not one per molecule
        return torch.log(self.prob + self.eps)[idcs]


RANGE_CHECK_EVERY = 25          # steps between two looks at the f16-range flag inside the fused sampling loops


class _RangeCheckpoints:
    """Range guard of the split-precision mode INSIDE a sampling loop.  Every RANGE_CHECK_EVERY steps the loop hands over a snapshot of its
    state (latent + counters) together with an asynchronous copy of the device flag word; one interval later that copy has long arrived and
    is looked at without stalling the GPU: clean -> the snapshot becomes the restart point; GCDM_FLAG_F16_RANGE -> the loop resumes from the
    previous restart point with fp32 MFMA instead of re-running the whole trajectory (an overflow at step 900 of 1000 costs <= 1.3x a
    clean run; it used to cost 1 + 2.7)."""

    def __init__(self, device):
        self.device = device
        self.good = None             # (state, tensors) verified clean
        self.pend = None             # (state, tensors, event, host flag) waiting for its flag copy
        self.rewinds = 0
        self.tail_flag = False       # GCDM_FLAG_TAIL seen (comes together with GCDM_FLAG_F16_RANGE: the rewind repairs the trajectory)

    def snapshot(self, state: Dict[str, Any], tensors: List[torch.Tensor], flags: torch.Tensor):
        """Called on the stream the latent is valid on.  Returns the restart point to rewind to if the PREVIOUS snapshot's flag is dirty."""
        rewind = self.resolve()
        if rewind is not None:
            return rewind
        host = torch.zeros(flags.numel(), dtype=torch.int32).pin_memory()
        copies = [t.clone() for t in tensors]
        host.copy_(flags, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self.pend = (dict(state), copies, ev, host)
        return None

    def resolve(self, final_flags: Optional[int] = None):
        """Looks at the pending snapshot's flag copy (or, at the end of the run, at the final flag word).  Returns (state, tensors) to rewind
        to, or None if the trajectory so far is clean."""
        if self.pend is not None:
            st, copies, ev, host = self.pend
            ev.synchronize()
            dirty = any(int(v) & _native.FLAG_F16_RANGE for v in host.tolist())
            self.tail_flag |= any(int(v) & _native.FLAG_TAIL for v in host.tolist())      # (fused layer launch; the caller of resolve() disables it on its handle)
            self.pend = None
            if dirty:
                return self._rewind()
            st["flags"] = [int(v) for v in host.tolist()]      # the flag word(s) AT the snapshot: what a rewind restores (bits raised during a
            self.good = (st, copies)                           # discarded f16 interval -- NaN in vel, CoG drift -- must not survive it)
        if final_flags is not None and final_flags & _native.FLAG_F16_RANGE:
            return self._rewind()
        return None

    def _rewind(self):
        self.rewinds += 1
        return self.good

    @staticmethod
    def restore_flags(flags: torch.Tensor, state: Dict[str, Any]) -> None:
        """Device flag word(s) back to their value at the restart point.  The first restart point (start of the loop) has no copy: only the
        bits a network evaluation / decode can raise are cleared there (a mean-not-zero flag of the encode step in front of it stays)."""
        saved = state.get("flags")
        if saved is not None:
            flags.copy_(torch.tensor(saved, dtype=flags.dtype).to(flags.device, non_blocking=True))
        else:
            flags.bitwise_and_(~(_native.FLAG_F16_RANGE | _native.FLAG_NAN_VEL | _native.FLAG_COG_DRIFT))


class _Batch(AttrDict):
    """Attribute bag with the three fields the dynamics network reads (stand-in for torch_geometric Batch)."""


class EquivariantVariationalDiffusion(nn.Module):
    def __init__(self, dynamics_network: nn.Module, diffusion_cfg: Any, dataloader_cfg: Any, dataset_info: Dict[str, Any]):
        super().__init__()
        assert cfg_get(diffusion_cfg, "parametrization", "eps") in ["eps"], "Epsilon is currently the only supported parametrization."
        if cfg_get(diffusion_cfg, "noise_schedule") == "learned":
            raise NotImplementedError("learned noise schedule (training) is out of scope")
        self.diffusion_cfg = diffusion_cfg
        self.diffusion_target = cfg_get(diffusion_cfg, "diffusion_target", "atom_types_and_coords")
        self.num_atom_types = int(cfg_get(dataloader_cfg, "num_atom_types"))
        self.num_x_dims = int(cfg_get(dataloader_cfg, "num_x_dims", 3))
        self.include_charges = bool(cfg_get(dataloader_cfg, "include_charges"))
        self.num_node_scalar_features = self.num_atom_types + int(self.include_charges)
        self.T = int(cfg_get(diffusion_cfg, "num_timesteps"))
        self.dynamics_network = dynamics_network
        histogram = {int(k): int(v) for k, v in dataset_info["n_nodes"].items()}
        self.num_nodes_distribution = NumNodesDistribution(histogram)
        self.gamma = PredefinedNoiseSchedule(noise_schedule=cfg_get(diffusion_cfg, "noise_schedule"), num_timesteps=self.T,
                                             noise_precision=float(cfg_get(diffusion_cfg, "noise_precision")))
        self._gamma_uploaded = None

    # ---- schedule algebra (:318-367) ---------------------------------------------------------------
    @staticmethod
    def sigma(gamma, target_tensor):
        return inflate_batch_array(torch.sqrt(torch.sigmoid(gamma)), target_tensor)

    @staticmethod
    def alpha(gamma, target_tensor):
        return inflate_batch_array(torch.sqrt(torch.sigmoid(-gamma)), target_tensor)

    @staticmethod
    def SNR(gamma):
        return torch.exp(-gamma)

    @staticmethod
    def sigma_and_alpha_t_given_s(gamma_t, gamma_s, target_tensor):
        sigma2_t_given_s = inflate_batch_array(-torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t)), target_tensor)
        log_alpha2_t_given_s = F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s)
        alpha_t_given_s = inflate_batch_array(torch.exp(0.5 * log_alpha2_t_given_s), target_tensor)
        return sigma2_t_given_s, torch.sqrt(sigma2_t_given_s), alpha_t_given_s

    # ---- noise (:396-437, 795-837) -----------------------------------------------------------------
    def sample_combined_position_feature_noise(self, batch_index, node_mask, generate_x_only: bool = False,
                                               generator: Optional[torch.Generator] = None, num_graphs: Optional[int] = None):
        # (num_graphs: the batch size when the caller knows it -- the reference's torch_scatter call derives it from batch_index.max(), a host
        #  sync per draw that leaves the GPU idle for ~0.6 ms per sampling step at the benchmark size)
        n, B = len(batch_index), (int(batch_index.max().item()) + 1 if num_graphs is None else int(num_graphs))
        dev = batch_index.device
        z_x = torch.randn((n, self.num_x_dims), device=dev, generator=generator) * node_mask.float().unsqueeze(-1)
        z_x = _segment_mean_sub(z_x, batch_index, B, node_mask)
        if generate_x_only:
            return z_x
        z_h = torch.randn((n, self.num_node_scalar_features), device=dev, generator=generator) * node_mask.float().unsqueeze(-1)
        return torch.cat([z_x, z_h], dim=-1)

    def sample_normal(self, mu, sigma, batch_index, node_mask, fix_noise: bool = False, generate_x_only: bool = False,
                      eps: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        if eps is None:
            bi = torch.zeros_like(batch_index) if fix_noise else batch_index
            eps = self.sample_combined_position_feature_noise(bi, node_mask, generate_x_only=generate_x_only, generator=generator,
                                                              num_graphs=1 if fix_noise else sigma.shape[0])
        return mu + sigma[batch_index] * eps

    def compute_x_pred(self, zt, net_out, gamma_t, batch_index):
        sigma_t = self.sigma(gamma_t, target_tensor=net_out)
        alpha_t = self.alpha(gamma_t, target_tensor=net_out)
        return 1.0 / alpha_t[batch_index] * (zt - sigma_t[batch_index] * net_out)

    @staticmethod
    def assert_mean_zero_with_mask(x, node_mask, eps: float = 1e-10):
        largest_value = x.abs().max().item()
        error = torch.sum(x, dim=0, keepdim=True).abs().max().item()
        rel_error = error / (largest_value + eps)
        assert rel_error < 1e-2, f"Mean is not zero, as relative_error {rel_error}"

    # ---- likelihood terms of a data batch, evaluation mode (:371-397, 449-454, 493-557, 580-732, 910-1160) ---------------------
    @staticmethod
    def sum_node_features_except_batch(values, batch_index, num_graphs: Optional[int] = None):
        B = int(batch_index.max().item()) + 1 if num_graphs is None else num_graphs
        return torch.zeros(B, dtype=values.dtype, device=values.device).index_add_(0, batch_index, values.sum(-1))

    @staticmethod
    def gaussian_KL(q_mu_minus_p_mu_squared, q_sigma, p_sigma, d):
        return d * torch.log(p_sigma / q_sigma) + 0.5 * (d * q_sigma ** 2 + q_mu_minus_p_mu_squared) / p_sigma ** 2 - 0.5 * d

    @staticmethod
    def cdf_standard_gaussian(x):
        return 0.5 * (1.0 + torch.erf(x * (0.5 ** 0.5)))

    def subspace_dimensionality(self, num_nodes):
        return (num_nodes - 1) * self.num_x_dims

    def delta_log_px(self, num_nodes):
        return -self.subspace_dimensionality(num_nodes) * float(np.log(cfg_get(self.diffusion_cfg, "norm_values")[0]))

    def log_pN(self, num_nodes):
        return self.num_nodes_distribution.log_prob(num_nodes)

    def normalize(self, x, h, node_mask, generate_x_only: bool = False):
        nv, nb = cfg_get(self.diffusion_cfg, "norm_values"), cfg_get(self.diffusion_cfg, "norm_biases")
        x = x / nv[0]
        if generate_x_only:
            return x, (h.float() - nb[1]) / nv[1]
        m = node_mask.float()
        h_cat = (h["categorical"].float() - nb[1]) / nv[1] * m.unsqueeze(-1)
        h_int = (h["integer"].float() - nb[2]) / nv[2]
        if self.include_charges:
            h_int = h_int * m
        return x, {"categorical": h_cat, "integer": h_int}

    def compute_noised_representation(self, xh, batch_index, node_mask, gamma_t, generate_x_only: bool = False, eps: Optional[torch.Tensor] = None):
        """z_t ~ q(z_t | x, h) (:910-931).  ``eps``: optional RAW standard-normal draws [N, 3 + F] (masked and CoM-projected here)."""
        if eps is None:
            eps = self.sample_combined_position_feature_noise(batch_index, node_mask, generate_x_only=generate_x_only, num_graphs=int(gamma_t.shape[0]))
        else:
            m = node_mask.float().unsqueeze(-1)
            ex = _segment_mean_sub(eps[:, : self.num_x_dims] * m, batch_index, int(gamma_t.shape[0]), node_mask)
            eps = torch.cat([ex, eps[:, self.num_x_dims:] * m], dim=-1)
        return self.alpha(gamma_t, xh)[batch_index] * xh + self.sigma(gamma_t, xh)[batch_index] * eps, eps

    def compute_kl_prior(self, xh, batch_index, node_mask, num_nodes, device=None, generate_x_only: bool = False):
        B = len(num_nodes)
        gamma_T = self.gamma(torch.ones((B, 1), device=xh.device))
        mu_T = self.alpha(gamma_T, xh)[batch_index] * xh
        sigma_T = self.sigma(gamma_T, xh).reshape(B)
        one = torch.ones_like(sigma_T)
        nx = self.num_x_dims
        kl = self.gaussian_KL(self.sum_node_features_except_batch(mu_T[:, :nx] ** 2, batch_index, B), sigma_T, one, self.subspace_dimensionality(num_nodes))
        if generate_x_only:
            return kl
        mu_h2 = self.sum_node_features_except_batch(mu_T[:, nx:] ** 2 * node_mask.float().unsqueeze(-1), batch_index, B)
        return kl + self.gaussian_KL(mu_h2, sigma_T, one, 1)

    def log_constants_p_x_given_z0(self, num_nodes, device=None):
        B = len(num_nodes)
        gamma_0 = self.gamma(torch.zeros((B, 1), device=self.gamma.gamma.device))
        return self.subspace_dimensionality(num_nodes).to(gamma_0.device) * (-0.5 * gamma_0.view(B) - 0.5 * float(np.log(2 * np.pi)))

    def log_pxh_given_z0_without_constants(self, h, z_0, eps, net_out, gamma_0, batch_index, node_mask, device=None,
                                           generate_x_only: bool = False, epsilon: float = 1e-10):
        nv, nb = cfg_get(self.diffusion_cfg, "norm_values"), cfg_get(self.diffusion_cfg, "norm_biases")
        nx, B = self.num_x_dims, int(gamma_0.shape[0])
        psum = lambda v: self.sum_node_features_except_batch(v, batch_index, B)
        log_px = -0.5 * psum((eps[:, :nx] - net_out[:, :nx]) ** 2)
        if generate_x_only:
            return log_px, None
        m = node_mask.float().unsqueeze(-1)
        sigma_0 = self.sigma(gamma_0[batch_index], target_tensor=z_0)

        def mass(centre, width):             # probability of [centre - 0.5, centre + 0.5] under N(0, width)
            return torch.log(self.cdf_standard_gaussian((centre + 0.5) / width) - self.cdf_standard_gaussian((centre - 0.5) / width) + epsilon)

        z_cat = z_0[:, nx:-1] if self.include_charges else z_0[:, nx:]
        lp = mass(z_cat * nv[1] + nb[1] - 1.0, sigma_0 * nv[1])
        lp = lp - torch.logsumexp(lp, dim=-1, keepdim=True)
        log_ph = psum(lp * (h["categorical"] * nv[1] + nb[1]) * m)
        if self.include_charges:
            h_int = torch.round(h["integer"].reshape(h["integer"].shape[0], -1) * nv[2] + nb[2]).long()
            log_ph = log_ph + psum(mass(h_int - (z_0[:, -1:] * nv[2] + nb[2]), sigma_0 * nv[2]) * m)
        return log_px, log_ph

    def forward(self, batch, return_loss_info: bool = False, t_int: Optional[torch.Tensor] = None, noise: Optional[List[torch.Tensor]] = None,
                self_conditioning_prob: float = 0.5, fix_self_conditioning_noise: bool = False):
        """Loss / NLL terms of a data batch (:948-1160): (delta_log_px, error_t, SNR_weight, loss_0_x, loss_0_h, neg_log_constants, kl_prior,
        log_pN, t_int[, loss_info]), each per molecule.  ``batch``: x (CoM-free), h = {categorical, integer}, batch, mask, num_graphs,
        num_nodes_present, props_context (per node or None).
          * evaluation mode: two evaluations of the network (t and 0) on the fused kernels, under inference mode;
          * TRAINING mode (``.train()``): one evaluation at t >= 0 on the module path (HIP operators with autograd), the t = 0 terms masked
            in as the reference does (:1078-1101); ``loss_type == "l2"`` drops the weights / constants (:978, 1050-1063).
        Extensions for reproducible runs: ``t_int`` [B, 1] instead of the torch.randint draw, ``noise`` = the raw standard-normal draws
        [N, 3 + F] (evaluation: two, for z_t and z_0; training: one)."""
        if self.diffusion_target != "atom_types_and_coords":
            raise NotImplementedError(f"diffusion_target {self.diffusion_target!r}")
        if self.training:
            return self._loss_terms(batch, return_loss_info, t_int, noise, self_conditioning_prob, fix_self_conditioning_noise)
        with torch.inference_mode():
            return self._loss_terms(batch, return_loss_info, t_int, noise, self_conditioning_prob, fix_self_conditioning_noise)

    def _loss_terms(self, batch, return_loss_info, t_int, noise, self_conditioning_prob, fix_self_conditioning_noise):
        training = self.training
        l2 = training and cfg_get(self.diffusion_cfg, "loss_type", "l2") == "l2"
        x, h = self.normalize(batch.x, batch.h, node_mask=batch.mask)
        bi, B, num_nodes, mask = batch.batch, int(batch.num_graphs), batch.num_nodes_present, batch.mask
        dev = x.device
        if t_int is None:
            t_int = torch.randint(0 if training else 1, self.T + 1, size=(B, 1), device=dev)     # lowest_t (:982-990)
        t_int = t_int.to(dev).reshape(B, 1)
        t_is_zero = (t_int == 0).float().squeeze(-1)
        s, t = (t_int - 1) / self.T, t_int / self.T
        gamma_s, gamma_t = inflate_batch_array(self.gamma(s), x), inflate_batch_array(self.gamma(t), x)
        xh = torch.cat([x, h["categorical"]] + ([h["integer"].reshape(-1, 1)] if self.include_charges else []), dim=-1)
        z_t, eps_t = self.compute_noised_representation(xh, bi, mask, gamma_t, eps=None if noise is None else noise[0].to(dev))
        num_nodes = num_nodes.to(dev)
        delta_log_px = self.delta_log_px(num_nodes)
        neg_log_constants = -self.log_constants_p_x_given_z0(num_nodes, dev)
        kl_prior = self.compute_kl_prior(xh, batch_index=bi, node_mask=mask, num_nodes=num_nodes, device=dev)
        if training:
            self_cond = None
            if bool(cfg_get(self.diffusion_cfg, "self_condition", False)) and not bool((t_int == self.T).any()) and _random() < self_conditioning_prob:
                with torch.no_grad():                       # the estimate the network is conditioned on: a jump from t + 1 to 0 (:1016-1035)
                    t_sc = (t_int + 1) / self.T
                    z_sc, _ = self.compute_noised_representation(xh, bi, mask, inflate_batch_array(self.gamma(t_sc), x))
                    self_cond = self.sample_p_zs_given_zt(s=torch.zeros_like(t_sc), t=t_sc, z=z_sc, batch_index=bi, node_mask=mask,
                                                          context=getattr(batch, "props_context", None), fix_noise=fix_self_conditioning_noise,
                                                          self_condition=True).detach()
            _, net_out = self.dynamics_network(batch, z_t, t[bi], xh_self_cond=self_cond)
            error_t = self.sum_node_features_except_batch((eps_t - net_out) ** 2, bi, B)
            if l2:
                delta_log_px, neg_log_constants = torch.zeros_like(delta_log_px), torch.zeros_like(neg_log_constants)
                SNR_weight = torch.ones_like(error_t)
            else:
                SNR_weight = (self.SNR(gamma_s - gamma_t) - 1).squeeze(-1)
            log_px, log_ph = self.log_pxh_given_z0_without_constants(h=h, z_0=z_t, eps=eps_t, net_out=net_out, gamma_0=gamma_t, batch_index=bi,
                                                                     node_mask=mask, device=dev)
            loss_0_x, loss_0_h = -log_px * t_is_zero, -log_ph * t_is_zero
            error_t = error_t * (1 - t_is_zero)
        else:
            # L_0 from its own draw at t = 0 (second evaluation of the network, :1107-1128)
            t_zeros = torch.zeros_like(s)
            gamma_0 = inflate_batch_array(self.gamma(t_zeros), x)
            z_0, eps_0 = self.compute_noised_representation(xh, bi, mask, gamma_0, eps=None if noise is None else noise[1].to(dev))

            def two_evaluations():          # no host sync between them; ONE check of the range guard after both (one sync per batch)
                a = self.dynamics_network(batch, z_t, t[bi], xh_self_cond=None, **self._deferred())[1]
                b = self.dynamics_network(batch, z_0, t_zeros[bi], xh_self_cond=None, **self._deferred())[1]
                self._final_range_check()
                return a, b
            try:
                net_out, net_out_0 = two_evaluations()
            except F16RangeError:           # an activation left the f16 images: the handle now runs fp32 MFMA, same inputs again
                try:
                    net_out, net_out_0 = two_evaluations()
                finally:                    # ... and returns to its default mode: one overflowing batch must not slow every later call
                    self.dynamics_network.set_mfma_mode(1)
            error_t = self.sum_node_features_except_batch((eps_t - net_out) ** 2, bi, B)
            SNR_weight = (self.SNR(gamma_s - gamma_t) - 1).squeeze(-1)
            log_px, log_ph = self.log_pxh_given_z0_without_constants(h=h, z_0=z_0, eps=eps_0, net_out=net_out_0, gamma_0=gamma_0, batch_index=bi,
                                                                     node_mask=mask, device=dev)
            loss_0_x, loss_0_h = -log_px, -log_ph
        terms = (delta_log_px, error_t, SNR_weight, loss_0_x, loss_0_h, neg_log_constants, kl_prior, self.log_pN(num_nodes), t_int.squeeze(-1))
        if not return_loss_info:
            return terms
        cnt = torch.zeros(B, dtype=x.dtype, device=dev).index_add_(0, bi, torch.ones_like(bi, dtype=x.dtype)).clamp(min=1)
        gmean = lambda v: (torch.zeros(B, dtype=x.dtype, device=dev).index_add_(0, bi, v) / cnt).mean()
        no = net_out.detach()
        info = {"eps_hat_x": gmean(no[:, : self.num_x_dims].abs().mean(-1)), "eps_hat_h": gmean(no[:, self.num_x_dims:].abs().mean(-1))}
        return (*terms, info)

    # ---- one step with the reference's signature (:1204-1278) --------------------------------------
    @torch.inference_mode()
    def sample_p_zs_given_zt(self, s, t, z, batch_index, node_mask, batch=None, context=None, fix_noise: bool = False,
                             generate_x_only: bool = False, self_condition: bool = False, xh_self_cond=None,
                             noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        gamma_s, gamma_t = self.gamma(s), self.gamma(t)
        sigma2_t_given_s, sigma_t_given_s, alpha_t_given_s = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, z)
        sigma_s = self.sigma(gamma_s, target_tensor=z)
        sigma_t = self.sigma(gamma_t, target_tensor=z)
        if batch is None:
            batch = _Batch(batch=batch_index, mask=node_mask, props_context=context)
        # deferred range guard: the reference's loop calls this T times; call k looks at the flag word of call k-1 (no host sync), and
        # sample_p_xh_given_z0 -- the call that ends every one of the reference's loops -- checks the last one before it returns samples
        _, eps_t = self.dynamics_network(batch, z, t[batch_index], x_self_cond=xh_self_cond, xh_self_cond=xh_self_cond, **self._deferred())
        mu = z / alpha_t_given_s[batch_index] - (sigma2_t_given_s[batch_index] / alpha_t_given_s[batch_index] / sigma_t[batch_index]) * eps_t
        sigma = sigma_t_given_s * sigma_s / sigma_t
        if noise is not None:  # raw standard-normal draws [N,3+F] (x-part gets CoM-projected like the reference's sampler)
            B = int(s.shape[0])
            nx = _segment_mean_sub(noise[:, : self.num_x_dims] * node_mask.float().unsqueeze(-1), batch_index, B, node_mask)
            noise = torch.cat([nx, noise[:, self.num_x_dims:] * node_mask.float().unsqueeze(-1)], dim=-1)
        zs = self.sample_normal(mu, sigma, batch_index, node_mask, fix_noise=fix_noise, generate_x_only=generate_x_only, eps=noise, generator=generator)
        zs_x = _segment_mean_sub(zs[:, : self.num_x_dims], batch_index, int(s.shape[0]), node_mask)
        return zs_x if generate_x_only else torch.cat([zs_x, zs[:, self.num_x_dims:]], dim=-1)

    def unnormalize(self, x, node_mask, h_cat=None, h_int=None, generate_x_only: bool = False):
        """(:735-759)"""
        nv, nb = cfg_get(self.diffusion_cfg, "norm_values"), cfg_get(self.diffusion_cfg, "norm_biases")
        x = x * nv[0]
        if generate_x_only:
            return x, None, None
        m = node_mask.float().unsqueeze(-1)
        h_cat = (h_cat * nv[1] + nb[1]) * m
        h_int = h_int * nv[2] + nb[2]
        if self.include_charges:
            h_int = h_int * m
        return x, h_cat, h_int

    @torch.inference_mode()
    def sample_p_xh_given_z0(self, z_0, batch_index, node_mask, batch_size: int, batch=None, context=None, fix_noise: bool = False,
                             generate_x_only: bool = False, xh_self_cond=None, noise: Optional[torch.Tensor] = None,
                             generator: Optional[torch.Generator] = None):
        """x, h ~ p(x, h | z0) with the reference's signature (:839-907): the call that ends each of the reference's sampling loops.  One
        network evaluation at t = 0 plus O(N) torch algebra; extension: ``noise`` = the raw standard-normal draw [N, 3 + F].  Before the
        samples are returned the deferred range guard of this evaluation AND of the sample_p_zs_given_zt calls in front of it is checked
        (one host sync); F16RangeError means the trajectory has to be re-run (the handle has been switched to fp32 MFMA)."""
        t_zeros = torch.zeros(size=(batch_size, 1), device=batch_index.device)
        gamma_0 = self.gamma(t_zeros)
        sigma_x = self.SNR(-0.5 * gamma_0)
        if batch is None:
            batch = _Batch(batch=batch_index, mask=node_mask, props_context=context)
        _, net_out = self.dynamics_network(batch, z_0, t_zeros[batch_index], x_self_cond=xh_self_cond, xh_self_cond=xh_self_cond, **self._deferred())
        mu_x = self.compute_x_pred(z_0, net_out, gamma_0, batch_index)
        if noise is not None:
            m = node_mask.float().unsqueeze(-1)
            noise = torch.cat([_segment_mean_sub(noise[:, : self.num_x_dims] * m, batch_index, batch_size, node_mask), noise[:, self.num_x_dims:] * m], dim=-1)
        xh = self.sample_normal(mu=mu_x, sigma=sigma_x, batch_index=batch_index, node_mask=node_mask, fix_noise=fix_noise, generate_x_only=generate_x_only, eps=noise,
                                generator=generator)
        x = xh[:, : self.num_x_dims]
        if generate_x_only:              # positions only (:894-897): no node features to decode
            x, _, _ = self.unnormalize(x, node_mask, generate_x_only=True)
            self._final_range_check()
            return x, {}
        h_cat = xh[:, self.num_x_dims:-1] if self.include_charges else xh[:, self.num_x_dims:]
        h_int = xh[:, -1:] if self.include_charges else torch.zeros(0, device=x.device)
        x, h_cat, h_int = self.unnormalize(x, node_mask, h_cat=h_cat, h_int=h_int)
        h_cat = F.one_hot(torch.argmax(h_cat, dim=-1), self.num_atom_types) * node_mask.long().unsqueeze(-1)
        h_int = torch.round(h_int).long() * node_mask.long().unsqueeze(-1)
        self._final_range_check()
        return x, {"integer": h_int, "categorical": h_cat}

    def _deferred(self) -> Dict[str, Any]:
        """Keyword that selects the deferred range guard for one module-level call -- only for this package's GCPNetDynamics (a foreign
        dynamics network gets no extra keyword)."""
        return {"_range_check": "deferred"} if hasattr(self.dynamics_network, "check_deferred_flags") else {}

    def _final_range_check(self) -> None:
        """Deferred range guard of the module-level calls (GCPNetDynamics.check_f16_range): the flag word of the LAST network evaluation is
        looked at here, with one host sync, before a driver hands results to its caller.  Raises F16RangeError (handle switched to fp32)."""
        chk = getattr(self.dynamics_network, "check_deferred_flags", None)
        if chk is not None:
            chk(wait=True)

    # ---- production loop (:1282-1412) ---------------------------------------------------------------
    def _native(self, device):
        dyn = self.dynamics_network
        if not hasattr(dyn, "_ensure_handle"):
            raise RuntimeError("mol_gen_sample needs the bio-diffusion_amd GCPNetDynamics (HIP) as dynamics_network")
        dyn._ensure_handle(torch.device(device))
        dyn.sync_weights()
        lib, h = dyn._lib, dyn._handle
        if self._gamma_uploaded is not h:
            g = self.gamma.gamma.detach().to("cpu", torch.float32).contiguous()
            _native.check(lib, h, lib.gcdm_set_gamma(h, C.c_void_p(g.data_ptr()), g.numel()), "gcdm_set_gamma")
            self._gamma_uploaded = h
        return dyn, lib, h

    def _mol_gen_sample_modules(self, num_samples, num_nodes, device, return_frames, num_timesteps, node_mask, context, fix_noise,
                                fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, generate_x_only: bool = False,
                                seed: int = 1234, init_xh: Optional[torch.Tensor] = None):
        """mol_gen_sample (:1282-1412) step by step through the reference-signature methods of this class -- torch algebra on the device around
        one network evaluation per step on whichever HIP path the configuration / mask selects.  Serves masked nodes inside the loop and the
        configurations the fused sampling kernels are not built for; ~10x slower per step than the fused loop.  ``noise_fn(k)``: raw draw k;
        without it the draws come from a device torch.Generator seeded with ``seed`` (reproducible per seed, in the reference's randn order).
        An activation beyond the f16 range of the split-precision kernels (F16RangeError of the deferred guard; the handle is then in fp32
        MFMA) re-runs the loop from z_T on the same noise, and the handle returns to its default mode afterwards."""
        dyn = self.dynamics_network
        try:
            return self._mol_gen_sample_modules_once(num_samples, num_nodes, device, return_frames, num_timesteps, node_mask, context, fix_noise,
                                                     fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, generate_x_only, seed, init_xh)
        except F16RangeError:
            log.warning("An activation left the f16 range of the split-precision kernels; re-running the sampling loop with fp32 MFMA.")
            try:
                out = self._mol_gen_sample_modules_once(num_samples, num_nodes, device, return_frames, num_timesteps, node_mask, context, fix_noise,
                                                        fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, generate_x_only, seed, init_xh)
            finally:
                if getattr(dyn, "_handle", None) is not None:
                    dyn.set_mfma_mode(1)
            self.last_flags = _native.FLAG_F16_RANGE          # reported: this sample was computed with fp32 MFMA
            return out

    def _mol_gen_sample_modules_once(self, num_samples, num_nodes, device, return_frames, num_timesteps, node_mask, context, fix_noise,
                                     fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, generate_x_only, seed, init_xh=None):
        num_timesteps = self.T if num_timesteps is None else num_timesteps
        assert 0 < return_frames <= num_timesteps, "Number of frames cannot be greater than number of timesteps."
        assert num_timesteps % return_frames == 0, "Number of frames must be evenly divisible by number of timesteps."
        num_nodes = torch.as_tensor(num_nodes)
        bi = num_nodes_to_batch_index(num_samples, num_nodes.to(device), device=device)
        node_mask = torch.ones_like(bi).bool() if node_mask is None else node_mask.to(device)
        if context is not None:
            context = context.to(device)[bi] * node_mask.float().unsqueeze(-1)
        t_norm = self.T if norm_with_original_timesteps else num_timesteps
        k = [0]
        gen = None
        if noise_fn is None:
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))

        def draw():
            if noise_fn is None:
                return None
            k[0] += 1
            return noise_fn(k[0] - 1).to(device, torch.float32)

        m = node_mask.float().unsqueeze(-1)
        raw = None
        if init_xh is not None:
            # optimisation loop (mol_gen_optimize :1451-1464): z = normalize(samples), no initial draw; the reference's assert_mean_zero_with_mask
            xin = init_xh.to(device, torch.float32)
            nv, nb = cfg_get(self.diffusion_cfg, "norm_values"), cfg_get(self.diffusion_cfg, "norm_biases")
            z = torch.cat((xin[:, : self.num_x_dims] / nv[0], (xin[:, self.num_x_dims:] - nb[1]) / nv[1] * m), dim=-1)
            self.assert_mean_zero_with_mask(z[:, : self.num_x_dims], node_mask)
        else:
            raw = draw()
        if init_xh is not None:
            pass
        elif raw is None:
            z = self.sample_combined_position_feature_noise(torch.zeros_like(bi) if fix_noise else bi, node_mask, generate_x_only=generate_x_only,
                                                            generator=gen, num_graphs=1 if fix_noise else num_samples)
        else:
            z = torch.cat((_segment_mean_sub(raw[:, : self.num_x_dims] * m, bi, num_samples, node_mask), raw[:, self.num_x_dims:] * m), dim=-1)
        self_cond_on = bool(cfg_get(self.diffusion_cfg, "self_condition", False))
        self_cond = None
        out = torch.zeros((return_frames,) + tuple(z.shape), device=device)
        for s in reversed(range(num_timesteps)):
            s_arr = torch.full((num_samples, 1), s / t_norm, device=device)
            t_arr = torch.full((num_samples, 1), (s + 1) / t_norm, device=device)
            z = self.sample_p_zs_given_zt(s=s_arr, t=t_arr, z=z, batch_index=bi, node_mask=node_mask, context=context, fix_noise=fix_noise,
                                          generate_x_only=generate_x_only, xh_self_cond=self_cond, noise=draw(), generator=gen)
            if step_callback is not None:
                step_callback(s, z)
            if (s * return_frames) % num_timesteps == 0:
                out[(s * return_frames) // num_timesteps] = self.unnormalize_z(z, node_mask, generate_x_only=generate_x_only)
            if self_cond_on:
                self_cond = self.sample_p_zs_given_zt(s=torch.zeros_like(s_arr), t=s_arr, z=z, batch_index=bi, node_mask=node_mask, context=context,
                                                      fix_noise=fix_self_conditioning_noise, self_condition=True, noise=draw(), generator=gen)
        x, h = self.sample_p_xh_given_z0(z_0=z, batch_index=bi, node_mask=node_mask, batch_size=num_samples, context=context,
                                         fix_noise=fix_self_conditioning_noise if self_cond_on else fix_noise, generate_x_only=generate_x_only,
                                         xh_self_cond=self_cond, noise=draw(), generator=gen)
        if return_frames == 1:
            cog = torch.zeros(num_samples, self.num_x_dims, device=device).index_add_(0, bi, x).abs().max().item()
            if cog > 5e-2:
                x = _segment_mean_sub(x, bi, num_samples, node_mask)
        out[0] = x if generate_x_only else torch.cat([x, h["categorical"].to(x.dtype)] + ([h["integer"].to(x.dtype)] if self.include_charges else []), dim=-1)
        self.last_flags = 0
        return out.squeeze(0), bi, node_mask

    def unnormalize_z(self, z, node_mask, generate_x_only: bool = False):
        """(:761-793)"""
        nx, nt = self.num_x_dims, self.num_atom_types
        if generate_x_only:
            return self.unnormalize(z[:, :nx], node_mask, generate_x_only=True)[0]
        x, h_cat, h_int = self.unnormalize(z[:, :nx], node_mask, h_cat=z[:, nx:nx + nt], h_int=z[:, nx + nt:])
        return torch.cat([x, h_cat] + ([h_int] if self.include_charges else []), dim=-1)

    @torch.inference_mode()
    def mol_gen_sample(self, num_samples: int, num_nodes: torch.Tensor, device: Union[torch.device, str], return_frames: int = 1,
                       num_timesteps: Optional[int] = None, node_mask: Optional[torch.Tensor] = None,
                       context: Optional[torch.Tensor] = None, fix_noise: bool = False, generate_x_only: bool = False,
                       fix_self_conditioning_noise: bool = False, norm_with_original_timesteps: bool = False,
                       noise_fn: Optional[Callable[[int], torch.Tensor]] = None, seed: int = 1234,
                       step_callback: Optional[Callable[[int, torch.Tensor], None]] = None, _retry_fp32: bool = False,
                       _init_xh: Optional[torch.Tensor] = None, _t_norm: Optional[int] = None, lanes: int = 1
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Draw samples.  ``noise_fn(k)`` (optional) returns the k-th raw standard-normal draw [N,3+F] on ``device``
        (k = 0 for z_T, then one per step, then one for the final decode: the reference's randn call order,
        SURVEY A.5); without it noise comes from on-device Philox(seed)."""
        if generate_x_only:
            # position-only diffusion (:1292, 1325-1327, 1349, 1385, 1407-1408): z = z_x, every draw is the centred x-noise alone, the result is the
            # [N, 3] positions.  It needs a dynamics network built WITHOUT node features (dataloader_cfg.num_atom_types = 0, include_charges = False:
            # xh is [N, 3] then, as in the reference) -- the general loop on the module path.
            if getattr(self.dynamics_network, "num_atom_types", 0) + int(getattr(self.dynamics_network, "include_charges", False)) > 0:
                raise ValueError("generate_x_only needs a dynamics network built without node features (num_atom_types = 0, include_charges = False); "
                                 "the reference fails on the feature width of this one too (gcpnet.py:1093-1110)")
            return self._mol_gen_sample_modules(num_samples, num_nodes, device, return_frames, num_timesteps, node_mask, context, fix_noise,
                                                fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, generate_x_only=True, seed=seed)
        masked = node_mask is not None and not bool(node_mask.all())
        if masked or getattr(self.dynamics_network, "fused_unsupported", None) is not None or getattr(self.dynamics_network, "path", "auto") == "modules":
            # general loop: masked nodes inside the loop, or a configuration the fused sampling kernels are not built for
            if _t_norm is not None:
                raise NotImplementedError("an explicit time normalisation runs on the fused path only")
            return self._mol_gen_sample_modules(num_samples, num_nodes, torch.device(device), return_frames, num_timesteps, node_mask, context,
                                                fix_noise, fix_self_conditioning_noise, norm_with_original_timesteps, noise_fn, step_callback, seed=seed,
                                                init_xh=_init_xh)
        self_cond_on = bool(getattr(self.dynamics_network, "self_condition", False))
        if fix_noise or self_cond_on:
            lanes = 1                  # fix_noise: the noise is centred over the whole flat batch; self-conditioning: not sliced (yet)
        if self_cond_on and fix_self_conditioning_noise != fix_noise:
            raise NotImplementedError("fix_self_conditioning_noise must equal fix_noise")
        num_timesteps = self.T if num_timesteps is None else num_timesteps
        assert 0 < return_frames <= num_timesteps, "Number of frames cannot be greater than number of timesteps."
        assert num_timesteps % return_frames == 0, "Number of frames must be evenly divisible by number of timesteps."
        # time normalisation of the loop (:1333-1341): s / T_norm with T_norm = self.T if norm_with_original_timesteps else num_timesteps
        t_norm = _t_norm if _t_norm is not None else (self.T if norm_with_original_timesteps else num_timesteps)
        if num_timesteps > t_norm:
            raise ValueError("num_timesteps exceeds the normalising number of timesteps")
        if lanes > 1 and not _retry_fp32 and len(num_nodes) >= 2 * lanes and not (
                noise_fn is not None or return_frames != 1 or _init_xh is not None or step_callback is not None):
            # slices of the flat batch on several handles / streams: plain sampling with on-device noise; anything else runs on one handle
            return self._mol_gen_sample_lanes(num_samples, num_nodes, device, num_timesteps, t_norm, context, seed, lanes)
        device = torch.device(device)
        dyn, lib, h = self._native(device)
        num_nodes = torch.as_tensor(num_nodes)
        batch_index = num_nodes_to_batch_index(num_samples, num_nodes.to(device), device=device)
        node_mask = torch.ones_like(batch_index).bool()
        dyn.plan(num_nodes.cpu())
        N, D = int(batch_index.shape[0]), self.num_x_dims + self.num_node_scalar_features
        ctx_ptr = None
        context_in = context
        if context is not None:
            context = context.to(device, torch.float32)[batch_index].contiguous()
            ctx_ptr = C.c_void_p(context.data_ptr())
        elif dyn.condition_on_context:
            raise ValueError("context required by a context-conditioned model")
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        flags = torch.zeros(1, dtype=torch.int32, device=device)
        fptr = C.c_void_p(flags.data_ptr())
        z = torch.empty((N, D), dtype=torch.float32, device=device)
        _native.check(lib, h, lib.gcdm_set_option(h, b"fix_noise", int(bool(fix_noise))), "gcdm_set_option")
        frames = torch.zeros((return_frames, N, D), dtype=torch.float32, device=device)     # frame 0 = the final sample (:1404-1410)
        out = frames[0]
        k = 0

        def nptr():
            nonlocal k
            if noise_fn is None:
                k += 1
                return None, None
            nz = noise_fn(k).to(device, torch.float32).contiguous()
            k += 1
            return nz, C.c_void_p(nz.data_ptr())

        if _init_xh is None:
            keep, p = nptr()
            _native.check(lib, h, lib.gcdm_sample_init(h, C.c_void_p(z.data_ptr()), p, C.c_uint64(seed), stream), "gcdm_sample_init")
        else:                                # optimisation loop: z = normalize(samples) (:1451-1464), no initial draw
            xin = _init_xh.to(device, torch.float32).contiguous()
            if xin.shape != (N, D):
                raise ValueError(f"samples have shape {tuple(xin.shape)}, expected {(N, D)}")
            _native.check(lib, h, lib.gcdm_encode_samples(h, C.c_void_p(xin.data_ptr()), C.c_void_p(z.data_ptr()), fptr, stream),
                          "gcdm_encode_samples")
        self_cond = torch.zeros_like(z) if self_cond_on else None     # the estimate fed back into the next step (:1363-1375)
        guard = _RangeCheckpoints(device) if (dyn.mfma_mode == 1 and not _retry_fp32) else None
        fell_back = False

        def restart(point):
            """Resume from a clean snapshot with fp32 MFMA (the f16-range flag was raised after it)."""
            nonlocal k, fell_back
            st0, (z0, *rest) = point
            z.copy_(z0)
            if self_cond_on:
                self_cond.copy_(rest[0])
            k = st0["k"]
            _RangeCheckpoints.restore_flags(flags, st0)
            if not fell_back:
                log.warning("An activation left the f16 range of the split-precision kernels; resuming from step %d with fp32 MFMA.", st0["s"])
                dyn.set_mfma_mode(0)
                fell_back = True
                self.last_range_resume_step = st0["s"]
            return st0["s"]

        try:
            s = num_timesteps - 1
            first = {"s": s, "k": k}
            if guard is not None:
                guard.good = (first, [z.clone()] + ([self_cond.clone()] if self_cond_on else []))
            while True:
                while s >= 0:
                    if guard is not None and not fell_back and (num_timesteps - 1 - s) % RANGE_CHECK_EVERY == 0 and s != num_timesteps - 1:
                        point = guard.snapshot({"s": s, "k": k}, [z] + ([self_cond] if self_cond_on else []), flags)
                        if point is not None:
                            s = restart(point)
                            continue
                    keep, p = nptr()
                    if self_cond_on:
                        keep2, p2 = nptr()
                        st = lib.gcdm_sample_step_sc(h, C.c_void_p(z.data_ptr()), C.c_void_p(self_cond.data_ptr()), int(s != num_timesteps - 1), ctx_ptr, s, t_norm,
                                                     p, p2, C.c_uint64(seed), fptr, stream)
                    else:
                        st = lib.gcdm_sample_step(h, C.c_void_p(z.data_ptr()), ctx_ptr, s, t_norm, p, C.c_uint64(seed), fptr, stream)
                    _native.check(lib, h, st, "gcdm_sample_step")
                    if return_frames > 1 and (s * return_frames) % num_timesteps == 0:             # save frame (:1354-1361)
                        fr = frames[(s * return_frames) // num_timesteps]
                        _native.check(lib, h, lib.gcdm_unnormalize_z(h, C.c_void_p(z.data_ptr()), C.c_void_p(fr.data_ptr()), stream), "gcdm_unnormalize_z")
                    if step_callback is not None:
                        step_callback(s, z)              # (fires again for the steps a resumed run repeats)
                    s -= 1
                keep, p = nptr()
                _native.check(lib, h, lib.gcdm_set_option(h, b"cog_fix", 1 if return_frames == 1 else 0), "gcdm_set_option")   # :1389
                if self_cond_on:
                    st = lib.gcdm_sample_final_sc(h, C.c_void_p(z.data_ptr()), C.c_void_p(self_cond.data_ptr()) if num_timesteps > 0 else None, ctx_ptr, p,
                                                  C.c_uint64(seed), C.c_void_p(out.data_ptr()), fptr, stream)
                else:
                    st = lib.gcdm_sample_final(h, C.c_void_p(z.data_ptr()), ctx_ptr, p, C.c_uint64(seed), C.c_void_p(out.data_ptr()), fptr, stream)
                lib.gcdm_set_option(h, b"cog_fix", 1)
                _native.check(lib, h, st, "gcdm_sample_final")
                fl = int(flags.item())   # the one host sync of a clean run
                point = guard.resolve(fl) if (guard is not None and not fell_back) else None
                if point is None:
                    break
                s = restart(point)       # an overflow in the last interval (or in the decode): repeat it in fp32
        finally:
            lib.gcdm_set_option(h, b"fix_noise", 0)
            if fell_back:
                dyn.set_mfma_mode(1)
        if fl & _native.FLAG_F16_RANGE:
            raise RuntimeError("f16 range flag raised in fp32 mode (internal error)")
        if fell_back:
            fl |= _native.FLAG_F16_RANGE          # reported in last_flags: part of this sample was computed with fp32 MFMA
        self.last_range_rewinds = 0 if guard is None else guard.rewinds
        if guard is not None and (guard.tail_flag or (fl & _native.FLAG_TAIL)):
            dyn.disable_fused_layer("mol_gen_sample")      # (the affected interval was repeated with two launches per layer by the range rewind)
        if not fell_back:
            self.last_range_resume_step = None
        if fl & _native.FLAG_MEAN_NOT_ZERO:
            raise AssertionError("Mean is not zero: the supplied samples are not centred (assert_mean_zero_with_mask, relative error >= 1e-2)")
        if fl & _native.FLAG_NAN_VEL:
            log.warning("Detected NaN in `vel` -> GCPNet `vel` output was reset to zero for at least one time step.")
        if fl & _native.FLAG_COG_DRIFT:
            log.warning("CoG drift above 5e-2. Projected the positions down.")
        self.last_flags = fl
        return (out if return_frames == 1 else frames), batch_index, node_mask

    @torch.inference_mode()
    def inpaint(self, molecule: Dict[str, Any], node_mask_fixed: torch.Tensor, num_resamplings: int = 1, jump_length: int = 1,
                return_frames: int = 1, num_timesteps: Optional[int] = None, context: Optional[torch.Tensor] = None,
                generate_x_only: bool = False, noise_fn: Optional[Callable[[int], torch.Tensor]] = None, seed: int = 1234,
                _retry_fp32: bool = False) -> torch.Tensor:
        """Draw samples while keeping parts of the given molecules fixed (RePaint; variational_diffusion.py:1582-1789).
        ``molecule``: dict with "x" [N,3], "one_hot" [N,F], "charges" [N,1] (if the model has charges), "num_nodes" [B] (and optionally
        "batch_index", which must be the contiguous one); ``node_mask_fixed`` [N] bool.  Returns [N,3+F], or [return_frames,N,3+F].
        The reference method raises on every call (:1650, :1177); this is that method with the two tokens repaired (include/gcdm_hip.h,
        DESIGN.md 7), pinned by tests/golden/inpaint_small_qm9.npz.  ``noise_fn(k)``: the k-th raw draw in the reference's order -- z_T, then
        per step [known part, model step, self-conditioning estimate if any], one per jump back, and the final decode."""
        num_timesteps = self.T if num_timesteps is None else num_timesteps
        assert 0 < return_frames <= num_timesteps, "Number of frames cannot be greater than number of timesteps."
        assert num_timesteps % return_frames == 0, "Number of frames must be evenly divisible by number of timesteps."
        assert jump_length == 1 or return_frames == 1, "Chain visualization is only implemented for `jump_length=1`"
        if (generate_x_only or getattr(self.dynamics_network, "fused_unsupported", None) is not None
                or getattr(self.dynamics_network, "path", "auto") == "modules"):
            # position-only diffusion (a dynamics network without node features) and configurations off the fused kernels: the general loop
            return self._inpaint_modules(molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context,
                                         generate_x_only, noise_fn, seed)
        num_nodes = torch.as_tensor(molecule["num_nodes"])
        device = torch.device(molecule["x"].device)
        dyn, lib, h = self._native(device)
        self_cond_on = bool(getattr(dyn, "self_condition", False))
        batch_index = num_nodes_to_batch_index(len(num_nodes), num_nodes.to(device), device=device)
        if "batch_index" in molecule and not torch.equal(molecule["batch_index"].to(device), batch_index):
            raise ValueError("molecule['batch_index'] must be the contiguous index implied by molecule['num_nodes']")
        dyn.plan(num_nodes.cpu())
        N, D = int(batch_index.shape[0]), self.num_x_dims + self.num_node_scalar_features
        parts = [molecule["x"], molecule["one_hot"]] + ([molecule["charges"]] if self.include_charges else [])
        xh0 = torch.cat([p.to(device, torch.float32) for p in parts], dim=-1).contiguous()
        fixed = node_mask_fixed.to(device).bool().contiguous()
        if xh0.shape != (N, D) or fixed.shape != (N,):
            raise ValueError(f"molecule has shape {tuple(xh0.shape)} / mask {tuple(fixed.shape)}, expected {(N, D)} / {(N,)}")
        ctx_ptr, context_in = None, context
        if context is not None:
            context = context.to(device, torch.float32)[batch_index].contiguous()
            ctx_ptr = C.c_void_p(context.data_ptr())
        elif dyn.condition_on_context:
            raise ValueError("context required by a context-conditioned model")
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        flags = torch.zeros(1, dtype=torch.int32, device=device)
        fptr, sd = C.c_void_p(flags.data_ptr()), C.c_uint64(seed)
        ptr = lambda t_: None if t_ is None else C.c_void_p(t_.data_ptr())       # noqa: E731
        z = torch.empty((N, D), dtype=torch.float32, device=device)
        frames = torch.zeros((return_frames, N, D), dtype=torch.float32, device=device)
        self_cond = torch.zeros_like(z) if self_cond_on else None
        k = 0
        held: List[Optional[torch.Tensor]] = []

        def draw():
            """Tape tensor (kept alive until the stream has consumed it) or None = Philox draw number k."""
            nonlocal k
            nz = None if noise_fn is None else noise_fn(k).to(device, torch.float32).contiguous()
            k += 1
            held.append(nz)
            return ptr(nz)

        _native.check(lib, h, lib.gcdm_inpaint_center(h, ptr(xh0), ptr(fixed), ptr(xh0), stream), "gcdm_inpaint_center")
        _native.check(lib, h, lib.gcdm_sample_init(h, ptr(z), draw(), sd, stream), "gcdm_sample_init")
        schedule = repaint_schedule(num_resamplings, jump_length, num_timesteps)
        s = num_timesteps - 1
        first = True
        for i, num_denoise_steps in enumerate(schedule):
            for j in range(num_denoise_steps):
                base = k
                p_known, p_unknown = draw(), draw()
                p_sc = draw() if self_cond_on else None
                st = lib.gcdm_inpaint_step(h, ptr(z), ptr(xh0), ptr(fixed), ptr(self_cond), int(not first), ctx_ptr, s, num_timesteps, p_known, p_unknown,
                                           p_sc, sd, base, fptr, stream)
                _native.check(lib, h, st, "gcdm_inpaint_step")
                first = False
                # frame at the end of a resample cycle (:1707-1715)
                if return_frames > 1 and (num_denoise_steps > jump_length or i == len(schedule) - 1) and (s * return_frames) % num_timesteps == 0:
                    fr = frames[(s * return_frames) // num_timesteps]
                    _native.check(lib, h, lib.gcdm_unnormalize_z(h, ptr(z), ptr(fr), stream), "gcdm_unnormalize_z")
                if j == num_denoise_steps - 1 and i < len(schedule) - 1:       # go back `jump_length` steps (:1717-1737)
                    t = s + jump_length
                    dk = k
                    _native.check(lib, h, lib.gcdm_inpaint_jump(h, ptr(z), s, t, num_timesteps, draw(), sd, dk, stream), "gcdm_inpaint_jump")
                    s = t
                s -= 1
        out = frames[0]
        _native.check(lib, h, lib.gcdm_set_option(h, b"cog_fix", 1 if return_frames == 1 else 0), "gcdm_set_option")   # :1767
        if self_cond_on:
            st = lib.gcdm_sample_final_sc(h, ptr(z), ptr(self_cond) if not first else None, ctx_ptr, draw(), sd, ptr(out), fptr, stream)
        else:
            st = lib.gcdm_sample_final(h, ptr(z), ctx_ptr, draw(), sd, ptr(out), fptr, stream)
        lib.gcdm_set_option(h, b"cog_fix", 1)
        _native.check(lib, h, st, "gcdm_sample_final")
        fl = int(flags.item())                                                  # the one host sync of the run
        held.clear()
        if fl & _native.FLAG_F16_RANGE:
            if _retry_fp32:
                raise RuntimeError("f16 range flag raised in fp32 mode (internal error)")
            log.warning("An activation left the f16 range of the split-precision kernels; re-running the inpainting with fp32 MFMA.")
            dyn.set_mfma_mode(0)
            try:
                return self.inpaint(molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context_in,
                                    generate_x_only, noise_fn=noise_fn, seed=seed, _retry_fp32=True)
            finally:
                dyn.set_mfma_mode(1)
        if fl & _native.FLAG_NAN_VEL:
            log.warning("Detected NaN in `vel` -> GCPNet `vel` output was reset to zero for at least one time step.")
        if fl & _native.FLAG_COG_DRIFT:
            log.warning("CoG drift above 5e-2. Projected the positions down.")
        self.last_flags = fl
        return out if return_frames == 1 else frames

    def sample_p_zt_given_zs(self, zs, batch_index, node_mask, gamma_t, gamma_s, generate_x_only: bool = False, noise: Optional[torch.Tensor] = None,
                             generator: Optional[torch.Generator] = None):
        """The forward jump z_s -> z_t of RePaint (variational_diffusion.py:1163-1201) WITH the reference's crashing token repaired: `:1177` indexes
        a [B, 1] factor with the [N] node mask; every sibling gathers per-molecule factors with `[batch_index]`, and so does this.  ``gamma_*``:
        [B, 1] (the reference inflates them to [B, 1] as well)."""
        _, sigma_t_given_s, alpha_t_given_s = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, zs)
        B = int(gamma_t.shape[0])
        if noise is None:
            eps = self.sample_combined_position_feature_noise(batch_index, node_mask, generate_x_only=generate_x_only, generator=generator, num_graphs=B)
        else:
            m = node_mask.float().unsqueeze(-1)
            eps = torch.cat([_segment_mean_sub(noise[:, : self.num_x_dims] * m, batch_index, B, node_mask), noise[:, self.num_x_dims:] * m], dim=-1)
        zt = alpha_t_given_s[batch_index] * zs + sigma_t_given_s[batch_index] * eps
        zx = _segment_mean_sub(zt[:, : self.num_x_dims], batch_index, B, node_mask)
        return zx if generate_x_only else torch.cat([zx, zt[:, self.num_x_dims:]], dim=-1)

    def _inpaint_modules(self, molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context, generate_x_only,
                         noise_fn, seed):
        """_inpaint_modules_once with the range recovery of _mol_gen_sample_modules: an activation beyond the f16 range of the split-precision kernels
        (F16RangeError of the deferred guard; the handle is then in fp32 MFMA) re-runs the loop from the start on the same noise (the draws are
        indexed / seeded inside the loop), the handle returns to its default mode, and ``last_flags`` reports FLAG_F16_RANGE."""
        dyn = self.dynamics_network
        try:
            return self._inpaint_modules_once(molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context, generate_x_only,
                                              noise_fn, seed)
        except F16RangeError:
            log.warning("An activation left the f16 range of the split-precision kernels; re-running the inpainting loop with fp32 MFMA.")
            try:
                out = self._inpaint_modules_once(molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context, generate_x_only,
                                                 noise_fn, seed)
            finally:
                if getattr(dyn, "_handle", None) is not None:
                    dyn.set_mfma_mode(1)
            self.last_flags |= _native.FLAG_F16_RANGE         # reported: this result was computed with fp32 MFMA
            return out

    def _inpaint_modules_once(self, molecule, node_mask_fixed, num_resamplings, jump_length, return_frames, num_timesteps, context, generate_x_only,
                              noise_fn, seed):
        """inpaint (:1582-1789, the two repairs of the fused method's docstring) step by step through the reference-signature methods of this class:
        torch algebra on the device around one network evaluation per step.  Serves ``generate_x_only`` (position-only diffusion: the molecule is
        its positions, z = z_x, [N, 3] out -- the dynamics network must be built without node features, as for mol_gen_sample) and configurations
        the fused kernels are not built for.  ``noise_fn(k)``: raw draws in the reference's order; otherwise a device generator seeded with ``seed``."""
        device = torch.device(molecule["x"].device)
        nx = self.num_x_dims
        if generate_x_only and getattr(self.dynamics_network, "num_atom_types", 0) + int(getattr(self.dynamics_network, "include_charges", False)) > 0:
            raise ValueError("generate_x_only needs a dynamics network built without node features (num_atom_types = 0, include_charges = False); "
                             "the reference fails on the feature width of this one too (gcpnet.py:1093-1110)")
        num_nodes = torch.as_tensor(molecule["num_nodes"])
        B = len(num_nodes)
        bi = num_nodes_to_batch_index(B, num_nodes.to(device), device=device)
        if "batch_index" in molecule and not torch.equal(molecule["batch_index"].to(device), bi):
            raise ValueError("molecule['batch_index'] must be the contiguous index implied by molecule['num_nodes']")
        ones = torch.ones_like(bi).bool()
        fixed = node_mask_fixed.to(device).bool()
        if context is not None:
            context = context.to(device)[bi]
        if generate_x_only:
            xh0 = molecule["x"].to(device, torch.float32).clone()
        else:
            parts = [molecule["x"], molecule["one_hot"]] + ([molecule["charges"]] if self.include_charges else [])
            xh0 = torch.cat([p_.to(device, torch.float32) for p_ in parts], dim=-1)

        def fixed_mean(x):                                   # scatter(x[fixed], batch_index[fixed], reduce="mean"), one row per molecule (0 where none is fixed)
            f = fixed.float().unsqueeze(-1)
            sums = torch.zeros((B, x.shape[1]), device=device).index_add_(0, bi, x * f)
            cnt = torch.zeros(B, device=device).index_add_(0, bi, fixed.float()).clamp(min=1)
            return sums / cnt[:, None]

        xh0[:, :nx] = xh0[:, :nx] - fixed_mean(xh0[:, :nx])[bi]                    # :1625-1633
        k = [0]
        gen = None
        if noise_fn is None:
            gen = torch.Generator(device=device)
            gen.manual_seed(int(seed))

        def draw():
            if noise_fn is None:
                return None
            k[0] += 1
            return noise_fn(k[0] - 1).to(device, torch.float32)

        raw = draw()
        if raw is None:
            z = self.sample_combined_position_feature_noise(bi, ones, generate_x_only=generate_x_only, generator=gen, num_graphs=B)
        else:
            z = torch.cat((_segment_mean_sub(raw[:, :nx], bi, B, ones), raw[:, nx:]), dim=-1)
        out = torch.zeros((return_frames,) + tuple(z.shape), device=device)
        schedule = repaint_schedule(num_resamplings, jump_length, num_timesteps)
        self_cond_on = bool(cfg_get(self.diffusion_cfg, "self_condition", False))
        self_cond = None
        s = num_timesteps - 1
        fm = fixed.float().unsqueeze(-1)
        for i, num_denoise_steps in enumerate(schedule):
            for j in range(num_denoise_steps):
                s_arr = torch.full((B, 1), s / num_timesteps, device=device)
                t_arr = torch.full((B, 1), (s + 1) / num_timesteps, device=device)
                gamma_s = self.gamma(s_arr)
                rk = draw()
                if rk is None and gen is not None:
                    rk = torch.randn(xh0.shape, device=device, generator=gen)
                z_known, _ = self.compute_noised_representation(xh0, bi, ones, gamma_s, generate_x_only=generate_x_only, eps=rk)
                z_unknown = self.sample_p_zs_given_zt(s=s_arr, t=t_arr, z=z, batch_index=bi, node_mask=ones, context=context,
                                                      generate_x_only=generate_x_only, xh_self_cond=self_cond, noise=draw(), generator=gen)
                if self_cond_on:
                    self_cond = self.sample_p_zs_given_zt(s=torch.zeros_like(s_arr), t=s_arr, z=z_unknown, batch_index=bi, node_mask=ones, context=context,
                                                          generate_x_only=generate_x_only, self_condition=True, noise=draw(), generator=gen)
                shift = fixed_mean(z_unknown[:, :nx]) - fixed_mean(z_known[:, :nx])       # :1680-1697
                z_known = torch.cat((z_known[:, :nx] + shift[bi], z_known[:, nx:]), dim=-1)
                z = z_known * fm + z_unknown * (1 - fm)
                self.assert_mean_zero_with_mask(z[:, :nx], ones)
                if (num_denoise_steps > jump_length or i == len(schedule) - 1) and (s * return_frames) % num_timesteps == 0:
                    out[(s * return_frames) // num_timesteps] = self.unnormalize_z(z, ones, generate_x_only=generate_x_only)
                if j == num_denoise_steps - 1 and i < len(schedule) - 1:                  # go back `jump_length` steps (:1717-1737)
                    t = s + jump_length
                    gamma_t = self.gamma(torch.full((B, 1), t / num_timesteps, device=device))
                    z = self.sample_p_zt_given_zs(z, bi, ones, gamma_t, gamma_s, generate_x_only=generate_x_only, noise=draw(), generator=gen)
                    s = t
                s -= 1
        x, h = self.sample_p_xh_given_z0(z_0=z, batch_index=bi, node_mask=ones, batch_size=B, context=context, generate_x_only=generate_x_only,
                                         xh_self_cond=self_cond, noise=draw(), generator=gen)
        self.assert_mean_zero_with_mask(x, ones)
        if return_frames == 1:
            cog = torch.zeros(B, nx, device=device).index_add_(0, bi, x).abs().max().item()
            if cog > 5e-2:
                x = _segment_mean_sub(x, bi, B, ones)
        if generate_x_only:
            out[0] = x
        else:
            out[0] = torch.cat([x, h["categorical"].to(x.dtype)] + ([h["integer"].to(x.dtype)] if self.include_charges else []), dim=-1)
        # the device check word of the network evaluations of this loop (NaN in vel, ...; the range bit was handled by the deferred guard above)
        rf = getattr(self.dynamics_network, "read_flags", None)
        self.last_flags = (int(rf()) & ~_native.FLAG_F16_RANGE) if rf is not None and getattr(self.dynamics_network, "_handle", None) is not None else 0
        return out.squeeze(0)

    # ---- several independent batches in flight (evaluation driver) -------------------------------------------------------------
    class _Lane:
        """One extra library handle + stream: its own packed weights (26 MB) and workspace, so that the launches of different
        batches are independent and the GPU can fill the CUs a 100-molecule batch leaves idle."""

        def __init__(self, ddpm: "EquivariantVariationalDiffusion", device: torch.device):
            dyn = ddpm.dynamics_network
            self.lib = _native.load()
            self.h = C.c_void_p()
            idx = device.index if device.index is not None else torch.cuda.current_device()
            cfg = dyn._native_config(idx)
            _native.check(self.lib, self.h, self.lib.gcdm_create(C.byref(cfg), C.byref(self.h)), "gcdm_create")
            for key, val in dyn.state_dict().items():
                w = val.detach().to("cpu", torch.float32).contiguous()
                _native.check(self.lib, self.h, self.lib.gcdm_set_weight(self.h, key.encode(), C.c_void_p(w.data_ptr()), w.numel()), "gcdm_set_weight")
            _native.check(self.lib, self.h, self.lib.gcdm_finalize_weights(self.h), "gcdm_finalize_weights")
            g = ddpm.gamma.gamma.detach().to("cpu", torch.float32).contiguous()
            _native.check(self.lib, self.h, self.lib.gcdm_set_gamma(self.h, C.c_void_p(g.data_ptr()), g.numel()), "gcdm_set_gamma")
            self.lib.gcdm_set_option(self.h, b"mfma_mode", dyn.mfma_mode)
            # lane handles run CONCURRENTLY with other handles (slices of one batch, batches in flight): two launches per layer there -- the fused layer launch
            # (option "fuse_node") packs a single handle's tiles better (-3 % per step), but beside another launch its node role is gated and loses (+3 %)
            # (GCDM_LANE_FUSE=1: A/B hook.  Stream priorities for the slices -- lane 0 high, lane 1 normal, so that one slice's launch would be dispatched whole before
            #  the other's -- were measured too: no effect on the dispatch interleave, 7.00 vs 7.00 ms per step un-fused, 7.21 vs 7.22 fused; profiles/r06_ab_log.txt)
            self.lib.gcdm_set_option(self.h, b"fuse_node", int(os.environ.get("GCDM_LANE_FUSE", "0")))
            self.stream = torch.cuda.Stream(device)
            self.key = ddpm._lane_key(device)

        def fresh(self, ddpm: "EquivariantVariationalDiffusion", device: torch.device) -> bool:
            """Still a copy of the primary handle's weights / schedule / device?"""
            return self.h is not None and self.key == ddpm._lane_key(device)

        def close(self):
            if self.h:
                self.lib.gcdm_destroy(self.h)
                self.h = None

    @torch.inference_mode()
    def mol_gen_sample_concurrent(self, num_nodes_list: List[torch.Tensor], device: Union[torch.device, str],
                                  num_timesteps: Optional[int] = None, contexts: Optional[List[Optional[torch.Tensor]]] = None,
                                  seeds: Optional[List[int]] = None) -> List[Tuple[torch.Tensor, torch.Tensor, torch.Tensor]]:
        """`mol_gen_sample` for several independent batches at once: batch b runs on its own handle and HIP stream, the step
        launches of all batches are interleaved on the host.  Results are those of `mol_gen_sample(..., seed=seeds[b])` called one
        after the other (bit-identical); small batches (the evaluation driver's 100 molecules) no longer leave CUs idle."""
        device = torch.device(device)
        K = len(num_nodes_list)
        T = self.T if num_timesteps is None else num_timesteps
        contexts = contexts if contexts is not None else [None] * K
        seeds = seeds if seeds is not None else [1234 + b for b in range(K)]
        self._native(device)                                   # validates the dynamics network, uploads the primary handle
        lanes = self._get_lanes(K, device)
        D = self.num_x_dims + self.num_node_scalar_features
        work = []
        for b in range(K):
            ln = lanes[b]
            nn_ = torch.as_tensor(num_nodes_list[b], dtype=torch.int32, device="cpu").contiguous()
            _native.check(ln.lib, ln.h, ln.lib.gcdm_plan_batch(ln.h, len(nn_), C.c_void_p(nn_.data_ptr())), "gcdm_plan_batch")
            bi = num_nodes_to_batch_index(len(nn_), nn_.to(device), device=device)
            N = int(bi.shape[0])
            ctx = contexts[b]
            if ctx is not None:
                ctx = ctx.to(device, torch.float32)[bi].contiguous()
            elif self.dynamics_network.condition_on_context:
                raise ValueError("context required by a context-conditioned model")
            work.append(dict(lane=ln, bi=bi, z=torch.empty((N, D), dtype=torch.float32, device=device), out=torch.empty((N, D), dtype=torch.float32, device=device),
                             flags=torch.zeros(1, dtype=torch.int32, device=device), ctx=ctx, seed=C.c_uint64(seeds[b])))
        torch.cuda.synchronize(device)
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        for w in work:
            ln = w["lane"]
            _native.check(ln.lib, ln.h, ln.lib.gcdm_sample_init(ln.h, ptr(w["z"]), None, w["seed"], C.c_void_p(ln.stream.cuda_stream)), "gcdm_sample_init")
        for s in reversed(range(T)):
            for w in work:
                ln = w["lane"]
                st = ln.lib.gcdm_sample_step(ln.h, ptr(w["z"]), ptr(w["ctx"]), s, T, None, w["seed"], ptr(w["flags"]), C.c_void_p(ln.stream.cuda_stream))
                _native.check(ln.lib, ln.h, st, "gcdm_sample_step")
        for w in work:
            ln = w["lane"]
            st = ln.lib.gcdm_sample_final(ln.h, ptr(w["z"]), ptr(w["ctx"]), None, w["seed"], ptr(w["out"]), ptr(w["flags"]), C.c_void_p(ln.stream.cuda_stream))
            _native.check(ln.lib, ln.h, st, "gcdm_sample_final")
        torch.cuda.synchronize(device)
        results = []
        for b, w in enumerate(work):
            fl = int(w["flags"].item())
            if fl & _native.FLAG_F16_RANGE:                    # rare: redo this batch on the primary handle (which falls back to fp32 MFMA)
                log.warning("An activation left the f16 range in a concurrent batch; re-running it with fp32 MFMA.")
                results.append(self.mol_gen_sample(len(num_nodes_list[b]), num_nodes_list[b], device, num_timesteps=T, context=contexts[b], seed=seeds[b]))
                continue
            if fl & _native.FLAG_NAN_VEL:
                log.warning("Detected NaN in `vel` -> GCPNet `vel` output was reset to zero for at least one time step.")
            results.append((w["out"], w["bi"], torch.ones_like(w["bi"]).bool()))
        return results

    def _lane_key(self, device: torch.device):
        dyn = self.dynamics_network
        idx = device.index if device.index is not None else torch.cuda.current_device()
        g = self.gamma.gamma
        return (dyn._params_fingerprint(), idx, g.data_ptr(), g._version)

    def _get_lanes(self, K: int, device: torch.device):
        """K extra handles that mirror the primary one.  A lane is a COPY of the weights: after load_state_dict / an EMA swap / fine-tuning /
        `.to(other device)` the stale ones are rebuilt (the primary handle re-uploads through sync_weights)."""
        lanes = [ln for ln in (getattr(self, "_lanes", None) or [])]
        for i, ln in enumerate(lanes):
            if not ln.fresh(self, device):
                ln.close()
                lanes[i] = self._Lane(self, device)
        while len(lanes) < K:
            lanes.append(self._Lane(self, device))
        for ln in lanes:
            ln.lib.gcdm_set_option(ln.h, b"mfma_mode", self.dynamics_network.mfma_mode)
        self._lanes = lanes
        return lanes

    def release_lanes(self):
        for ln in getattr(self, "_lanes", None) or []:
            ln.close()
        self._lanes = []

    def __del__(self):
        try:
            self.release_lanes()
        except Exception:
            pass

    class _SlicedBatch:
        """One flat batch sampled as K contiguous slices of molecules, each on its own handle and HIP stream (same semantics and the
        same Philox noise as the single-handle run: a slice's boundary nodes read their flat neighbours from the adjacent slice,
        options "flat_prev" / "flat_next" / "node_base").  The latent is double-buffered (`gcdm_sample_step_to`) and the slices join
        once per step, so no slice ever reads a row its neighbour is writing.  Fills the round-quantisation tails of the big
        configurations (+5-6 % at 1024 QM9 / 256 GEOM molecules on MI355X)."""

        def __init__(self, ddpm: "EquivariantVariationalDiffusion", num_nodes, device: torch.device, context: Optional[torch.Tensor], seed: int, K: int):
            self.ddpm, self.device, self.K = ddpm, device, K
            dyn, lib, h0 = ddpm._native(device)
            self.dyn = dyn
            nn_ = torch.as_tensor(num_nodes, dtype=torch.int32, device="cpu")
            Bm = len(nn_)
            cuts = slice_cuts(nn_, K)
            self.cuts = cuts
            self.node_off = torch.cat((torch.zeros(1, dtype=torch.long), nn_.long().cumsum(0))).tolist()
            lanes = ddpm._get_lanes(K, device)
            self.batch_index = num_nodes_to_batch_index(Bm, nn_.to(device), device=device)
            N, D = int(self.batch_index.shape[0]), ddpm.num_x_dims + ddpm.num_node_scalar_features
            self.ctx = None
            if context is not None:
                self.ctx = context.to(device, torch.float32)[self.batch_index].contiguous()
            elif dyn.condition_on_context:
                raise ValueError("context required by a context-conditioned model")
            self.bufs = [torch.empty((N, D), dtype=torch.float32, device=device) for _ in range(2)]
            self.out = torch.empty((N, D), dtype=torch.float32, device=device)
            self.flags = torch.zeros(K, dtype=torch.int32, device=device)
            self.sd = C.c_uint64(seed)
            self.sl = []
            for k in range(K):
                ln = lanes[k]
                part = nn_[cuts[k]:cuts[k + 1]].contiguous()
                _native.check(ln.lib, ln.h, ln.lib.gcdm_plan_batch(ln.h, len(part), C.c_void_p(part.data_ptr())), "gcdm_plan_batch")
                n0 = self.node_off[cuts[k]]
                for name, val in ((b"flat_prev", int(k > 0)), (b"flat_next", int(k < K - 1)), (b"node_base", n0), (b"mfma_mode", dyn.mfma_mode)):
                    _native.check(ln.lib, ln.h, ln.lib.gcdm_set_option(ln.h, name, val), "gcdm_set_option")
                self.sl.append(dict(lane=ln, n0=n0, stream=C.c_void_p(ln.stream.cuda_stream), ev=torch.cuda.Event(),
                                    fl=C.c_void_p(self.flags.data_ptr() + 4 * k)))
            self.cur = 0

        @staticmethod
        def _row(t_, n0):
            return C.c_void_p(t_.data_ptr() + 4 * n0 * t_.shape[1])

        def _cptr(self, n0):
            return None if self.ctx is None else self._row(self.ctx, n0)

        def _join(self):
            for a_ in self.sl:
                for b_ in self.sl:
                    if a_ is not b_:
                        a_["lane"].stream.wait_event(b_["ev"])

        def init(self):
            start = torch.cuda.Event()
            start.record(torch.cuda.current_stream(self.device))
            for w in self.sl:
                ln = w["lane"]
                ln.stream.wait_event(start)
                _native.check(ln.lib, ln.h, ln.lib.gcdm_sample_init(ln.h, self._row(self.bufs[0], w["n0"]), None, self.sd, w["stream"]), "gcdm_sample_init")
                w["ev"].record(ln.stream)
            self.cur = 0

        def step(self, s: int, t_norm: int):
            self._join()
            cur, nxt = self.cur, 1 - self.cur
            for w in self.sl:
                ln = w["lane"]
                st = ln.lib.gcdm_sample_step_to(ln.h, self._row(self.bufs[cur], w["n0"]), self._row(self.bufs[nxt], w["n0"]), self._cptr(w["n0"]), s, t_norm,
                                                None, self.sd, w["fl"], w["stream"])
                _native.check(ln.lib, ln.h, st, "gcdm_sample_step_to")
                w["ev"].record(ln.stream)
            self.cur = nxt

        def final(self):
            self._join()
            for w in self.sl:
                ln = w["lane"]
                st = ln.lib.gcdm_sample_final(ln.h, self._row(self.bufs[self.cur], w["n0"]), self._cptr(w["n0"]), None, self.sd, self._row(self.out, w["n0"]),
                                              w["fl"], w["stream"])
                _native.check(ln.lib, ln.h, st, "gcdm_sample_final")
                w["ev"].record(ln.stream)
            self.wait()

        def wait(self):
            """The caller's current stream waits for every slice (no host sync)."""
            cs = torch.cuda.current_stream(self.device)
            for w in self.sl:
                cs.wait_event(w["ev"])

        def close(self):
            for w in self.sl:                                   # lanes go back to whole-batch behaviour
                for name in (b"flat_prev", b"flat_next", b"node_base"):
                    w["lane"].lib.gcdm_set_option(w["lane"].h, name, 0)

        def recentre_undrifted(self, drift: List[bool]):
            """The reference re-projects the WHOLE batch when any molecule drifted (:1389-1402): slices that saw no drift follow."""
            for k, w in enumerate(self.sl):
                if not drift[k]:
                    n0, n1 = w["n0"], self.node_off[self.cuts[k + 1]]
                    bi = self.batch_index[n0:n1] - self.batch_index[n0]
                    cnt = torch.bincount(bi).clamp(min=1).to(torch.float32)[:, None]
                    mean = torch.zeros((int(bi.max()) + 1, 3), device=self.device).index_add_(0, bi, self.out[n0:n1, :3]) / cnt
                    self.out[n0:n1, :3] -= mean[bi]

    @torch.inference_mode()
    def _mol_gen_sample_lanes(self, num_samples, num_nodes, device, num_timesteps, t_norm, context, seed, K):
        device = torch.device(device)
        sb = self._SlicedBatch(self, num_nodes, device, context, seed, K)
        guard = _RangeCheckpoints(device) if sb.dyn.mfma_mode == 1 else None
        fell_back = False
        cs = torch.cuda.current_stream(device)

        def fence():                         # the slices continue only after what the caller's stream has just done with their buffers
            ev = torch.cuda.Event()
            ev.record(cs)
            for w in sb.sl:
                w["lane"].stream.wait_event(ev)

        def restart(point):
            nonlocal fell_back
            st0, (z0,) = point
            sb.wait()
            sb.bufs[sb.cur].copy_(z0)
            _RangeCheckpoints.restore_flags(sb.flags, st0)
            if not fell_back:
                log.warning("An activation left the f16 range of the split-precision kernels; resuming from step %d with fp32 MFMA.", st0["s"])
                for w in sb.sl:
                    w["lane"].lib.gcdm_set_option(w["lane"].h, b"mfma_mode", 0)
                fell_back = True
                self.last_range_resume_step = st0["s"]
            fence()
            return st0["s"]

        try:
            sb.init()
            s = num_timesteps - 1
            if guard is not None:
                sb.wait()
                guard.good = ({"s": s}, [sb.bufs[sb.cur].clone()])
                fence()
            while True:
                while s >= 0:
                    if guard is not None and not fell_back and (num_timesteps - 1 - s) % RANGE_CHECK_EVERY == 0 and s != num_timesteps - 1:
                        sb.wait()
                        point = guard.snapshot({"s": s}, [sb.bufs[sb.cur]], sb.flags)
                        fence()
                        if point is not None:
                            s = restart(point)
                            continue
                    sb.step(s, t_norm)
                    s -= 1
                sb.final()
                fl_all = sb.flags.cpu().tolist()                 # the one host sync of a clean run
                fl = 0
                for v in fl_all:
                    fl |= int(v)
                point = guard.resolve(fl) if (guard is not None and not fell_back) else None
                if point is None:
                    break
                s = restart(point)
        finally:
            if fell_back:
                for w in sb.sl:
                    w["lane"].lib.gcdm_set_option(w["lane"].h, b"mfma_mode", 1)
            sb.close()
        if fl & _native.FLAG_F16_RANGE:
            raise RuntimeError("f16 range flag raised in fp32 mode (internal error)")
        if fell_back:
            fl |= _native.FLAG_F16_RANGE              # reported in last_flags: part of this sample was computed with fp32 MFMA
        self.last_range_rewinds = 0 if guard is None else guard.rewinds
        if not fell_back:
            self.last_range_resume_step = None
        drift = [bool(int(v) & _native.FLAG_COG_DRIFT) for v in fl_all]
        if any(drift) and not all(drift):
            sb.recentre_undrifted(drift)
        if fl & _native.FLAG_NAN_VEL:
            log.warning("Detected NaN in `vel` -> GCPNet `vel` output was reset to zero for at least one time step.")
        if fl & _native.FLAG_COG_DRIFT:
            log.warning("CoG drift above 5e-2. Projected the positions down.")
        self.last_flags = fl
        return sb.out, sb.batch_index, torch.ones_like(sb.batch_index).bool()

    def get_repaint_schedule(self, resamplings: int, jump_length: int, num_timesteps: int) -> List[int]:
        """variational_diffusion.py:1548-1578."""
        return repaint_schedule(resamplings, jump_length, num_timesteps)

    @torch.inference_mode()
    def mol_gen_optimize(self, samples: List[Tuple[torch.Tensor, torch.Tensor]], num_nodes: torch.Tensor, device: Union[torch.device, str],
                         return_frames: int = 1, num_timesteps: Optional[int] = None, node_mask: Optional[torch.Tensor] = None,
                         context: Optional[torch.Tensor] = None, generate_x_only: bool = False, norm_with_original_timesteps: bool = False,
                         noise_fn: Optional[Callable[[int], torch.Tensor]] = None, seed: int = 1234
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Optimise existing samples with the generative model (variational_diffusion.py:1416-1546): the samples
        ``[(x [n,3], one_hot [n,F]), ...]`` are normalised and taken as z at t = num_timesteps / T_norm, denoised for ``num_timesteps``
        steps and decoded.  As in the reference z carries no charge column (``"integer": torch.tensor([])``, :1457), so the model must
        have ``include_charges=False`` (the property-conditional QM9 models).  ``noise_fn(k)``: k = 0 is the first step's draw.
        ``return_frames > 1``: [frames, N, 3 + F] -- frame (s * return_frames) // T after the step to s, frame 0 the decoded sample (:1490-1497,
        1540-1546).  Configurations off the fused path (and ``dynamics_network.path = "modules"``) run the general loop."""
        if generate_x_only:
            # nothing to mirror: the reference's own call raises -- normalize(..., generate_x_only=True) does `h.float()` on the {"categorical", "integer"}
            # dict mol_gen_optimize hands it (variational_diffusion.py:716 via :1454-1459: AttributeError)
            raise NotImplementedError("mol_gen_optimize: generate_x_only raises in the reference too (variational_diffusion.py:716 via :1454: `h.float()` on a dict)")
        if self.include_charges:
            raise NotImplementedError("mol_gen_optimize builds z without the charge column (reference :1457): include_charges must be False")
        if len(samples) != len(num_nodes):
            raise ValueError("one (x, h) pair per molecule")
        xh = torch.cat([torch.cat((x.to(torch.float32), h.to(torch.float32)), dim=-1) for x, h in samples], dim=0)
        return self.mol_gen_sample(num_samples=len(samples), num_nodes=num_nodes, device=device, return_frames=return_frames, num_timesteps=num_timesteps,
                                   node_mask=node_mask, context=context, norm_with_original_timesteps=norm_with_original_timesteps,
                                   noise_fn=noise_fn, seed=seed, _init_xh=xh)
