"""bio-diffusion_amd -- MI355X-native implementation of the GCDM denoising inner loop.

Drop-in for ONE path of BioinfoMachineLearning/bio-diffusion: the GCPNet dynamics network evaluated at
every DDPM step of molecule sampling (reference: src/models/components/gcpnet.py:933-1232 and the sampler
in src/models/components/variational_diffusion.py:1204-1412).  Host code is Python on PyTorch-ROCm
(device memory + streams only); the compute is hand-written HIP for gfx950 behind the C ABIs declared in
``include/gcdm_hip.h`` (``libgcdm_hip.so``: the fused sampling path) and ``include/gcdm_ops.h`` (``libgcdm_ops.so``: the module-level
operators, forward and backward -- plug point 3, every configuration, training).  There is no CPU fallback: without the library or without a
GPU the forward raises.

The directory name contains a hyphen; import it with ``importlib.import_module("bio-diffusion_amd")`` or
through the alias module ``bio_diffusion_amd`` at the repository root.
"""
from .config import AttrDict, default_cfgs, load_cfg_tree, dataset_info          # noqa: F401
from .gcpnet import GCP2, GCPNetDynamics, F16RangeError                                        # noqa: F401
from .gcp_modules import GCP, GCPEmbedding, GCPMessagePassing, GCPInteractions, GCPLayerNorm, GCPDropout, get_GCP_with_custom_cfg  # noqa: F401
from . import ops                                                                # noqa: F401
from .variational_diffusion import EquivariantVariationalDiffusion, PredefinedNoiseSchedule, NumNodesDistribution  # noqa: F401
from .mol_gen_ddpm import QM9MoleculeGenerationDDPM, GEOMMoleculeGenerationDDPM, sample_sweep_conditionally  # noqa: F401
from . import _native, stability, xyz, sdf                                      # noqa: F401
from .sdf import write_sdf_file, build_molecules, bond_order_matrices, Molecule  # noqa: F401
from .xyz import save_xyz_file, write_xyz_file                                  # noqa: F401
from .stability import check_molecular_stability, check_molecular_stability_batch, get_bond_length_arrays, CategoricalDistribution  # noqa: F401

__all__ = [
    "AttrDict", "default_cfgs", "load_cfg_tree", "dataset_info", "GCP", "GCP2", "GCPEmbedding", "GCPMessagePassing", "GCPInteractions", "GCPLayerNorm",
    "GCPDropout", "get_GCP_with_custom_cfg", "ops", "GCPNetDynamics",
    "EquivariantVariationalDiffusion", "PredefinedNoiseSchedule", "NumNodesDistribution",
    "QM9MoleculeGenerationDDPM", "GEOMMoleculeGenerationDDPM",
    "check_molecular_stability", "check_molecular_stability_batch", "get_bond_length_arrays", "CategoricalDistribution",
    "save_xyz_file", "write_xyz_file", "write_sdf_file", "build_molecules", "bond_order_matrices", "Molecule", "F16RangeError",
]
