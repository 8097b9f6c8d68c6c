"""ctypes binding of libgcdm_hip.so (C ABI: include/gcdm_hip.h).  This file IS the binding a maintainer of the
reference would add; it contains no compute."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# GCDM_HIP_LIB: another build of the same sources (kernel variants for A/B and hazard runs: tools/build_variants.sh, tests/test_hazards_gpu.py); the
# default is the in-tree library __graft_entry__.build() produces
DEFAULT_LIB_PATH = os.path.join(_HERE, "libgcdm_hip.so")
LIB_PATH = os.environ.get("GCDM_HIP_LIB") or DEFAULT_LIB_PATH
SOURCES = [os.path.join(_HERE, "csrc", "gcdm_api.hip")]
HEADERS = ([os.path.join(_HERE, "csrc", f) for f in sorted(os.listdir(os.path.join(_HERE, "csrc"))) if f.endswith(".h") and f != "gcdm_ops.hip.h"]
           + [os.path.join(os.path.dirname(_HERE), "include", "gcdm_hip.h")])
# the module-level operators (forward + backward) live in their own small library (include/gcdm_ops.h)
OPS_LIB_PATH = os.path.join(_HERE, "libgcdm_ops.so")
OPS_SOURCES = [os.path.join(_HERE, "csrc", "gcdm_ops.hip")]
OPS_HEADERS = [os.path.join(_HERE, "csrc", "gcdm_ops.hip.h"), os.path.join(os.path.dirname(_HERE), "include", "gcdm_ops.h")]
ABI_VERSION = 2

FLAG_NAN_VEL, FLAG_MEAN_NOT_ZERO, FLAG_COG_DRIFT, FLAG_F16_RANGE = 1, 2, 4, 8
FLAG_TAIL = 16          # fused layer launch: placement check / bounded wait failed (raised together with FLAG_F16_RANGE: the fp32 re-run repairs the result)


class GcdmConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("num_atom_types", C.c_int32), ("include_charges", C.c_int32),
        ("num_context", C.c_int32), ("condition_on_time", C.c_int32), ("num_layers", C.c_int32),
        ("h_hidden_dim", C.c_int32), ("chi_hidden_dim", C.c_int32), ("e_hidden_dim", C.c_int32),
        ("xi_hidden_dim", C.c_int32), ("bottleneck", C.c_int32), ("num_timesteps", C.c_int32),
        ("node_positions_weight", C.c_float), ("norm_values", C.c_float * 3), ("norm_biases", C.c_float * 3),
        ("device", C.c_int32), ("self_condition", C.c_int32),
    ]


STABILITY_MAX_TYPES = 16


class GcdmBondTables(C.Structure):
    _fields_ = [
        ("num_types", C.c_int32), ("limit_bonds_to_one", C.c_int32),
        ("thr1", C.c_float * (STABILITY_MAX_TYPES * STABILITY_MAX_TYPES)),
        ("thr2", C.c_float * (STABILITY_MAX_TYPES * STABILITY_MAX_TYPES)),
        ("thr3", C.c_float * (STABILITY_MAX_TYPES * STABILITY_MAX_TYPES)),
        ("allowed_mask", C.c_uint32 * STABILITY_MAX_TYPES),
    ]


EXPORTS = [
    "gcdm_create", "gcdm_destroy", "gcdm_last_error", "gcdm_set_weight", "gcdm_finalize_weights", "gcdm_set_gamma",
    "gcdm_plan_batch", "gcdm_forward", "gcdm_sample_step", "gcdm_sample_final", "gcdm_sample_init", "gcdm_debug_read",
    "gcdm_debug_set_layer_limit", "gcdm_num_nodes", "gcdm_num_edges", "gcdm_forward_flops_executed",
    "gcdm_profile_enable", "gcdm_profile_edge_kernel_ms", "gcdm_profile_node_kernel_ms", "gcdm_set_option", "gcdm_get_option", "gcdm_check_stability", "gcdm_encode_samples", "gcdm_unnormalize_z", "gcdm_sample_step_to", "gcdm_forward_sc", "gcdm_sample_step_sc", "gcdm_sample_final_sc",
    "gcdm_inpaint_center", "gcdm_inpaint_step", "gcdm_inpaint_jump", "gcdm_timestep_index", "gcdm_bond_orders", "gcdm_plan_batch_masked",
]

_lib: Optional[C.CDLL] = None


def build(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU).  Rebuilds when a source is newer than the .so."""
    deps = SOURCES + HEADERS
    if not force and os.path.exists(DEFAULT_LIB_PATH) and all(os.path.getmtime(DEFAULT_LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return DEFAULT_LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = f"{DEFAULT_LIB_PATH}.{os.getpid()}.tmp"          # concurrent builders (one per rank) never see a half-written library
    # packed-fp32 VALU ops (v_pk_fma_f32 & co.) are switched off: with them the 32-edge split-precision kernel is not bit-reproducible
    # from run to run (first wrong values: the pre-phase FMAs the SLP vectoriser had packed, fed by per-lane VMEM loads), and
    # they buy nothing here -- round 5 wrote the packable steps of the VALU phases on float2 by hand (csrc/gcdm_edge_x3.hip.h, "packed fp32"):
    # 13 % fewer VALU instructions, +1.4 % tile cycles (+7.7 % with packed ops between the MFMAs); profiles/r05_packed_fp32_ab.md.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops",
           "-o", tmp] + SOURCES
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, DEFAULT_LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return DEFAULT_LIB_PATH


def build_ops(force: bool = False) -> str:
    deps = OPS_SOURCES + OPS_HEADERS
    if not force and os.path.exists(OPS_LIB_PATH) and all(os.path.getmtime(OPS_LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return OPS_LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = f"{OPS_LIB_PATH}.{os.getpid()}.tmp"
    try:
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", tmp] + OPS_SOURCES, check=True)
        os.replace(tmp, OPS_LIB_PATH)
    finally:
        if os.path.exists(tmp):
            os.remove(tmp)
    return OPS_LIB_PATH


P, I32, I64 = C.c_void_p, C.c_int32, C.c_int64
OPS_SIGNATURES = {
    "gcdm_op_gemm": [P, I64, I64, P, I64, I64, P, P, I64, I32, I64, I32, P],
    "gcdm_op_reduce_slices": [P, P, I64, I32, P],
    "gcdm_op_colsum": [P, P, I64, I32, P],
    "gcdm_op_colsum_slices": [P, P, I64, I32, I32, P],
    "gcdm_op_act": [I32, P, P, I64, P],
    "gcdm_op_act_bwd": [I32, P, P, P, I64, P],
    "gcdm_op_norm3": [P, P, I64, I32, I32, P],
    "gcdm_op_norm3_bwd": [P, P, P, P, I64, I32, I32, P],
    "gcdm_op_scalarize": [P, P, P, I64, I32, P],
    "gcdm_op_scalarize_bwd": [P, P, P, I64, I32, P],
    "gcdm_op_vectorize": [P, P, P, I64, I32, P],
    "gcdm_op_vectorize_bwd": [P, P, P, I64, I32, P],
    "gcdm_op_rowscale": [P, P, P, I64, I32, P],
    "gcdm_op_rowscale_bwd": [P, P, P, P, P, I64, I32, P],
    "gcdm_op_rowptr": [P, I64, I64, P, P, P],
    "gcdm_op_gather": [P, P, P, I64, I32, P],
    "gcdm_op_segment_sum": [P, P, P, I64, I32, I32, P],
    "gcdm_op_segment_bwd": [P, P, P, P, I64, I32, I32, P],
    "gcdm_op_scatter_add": [P, P, P, I64, I32, P],
    "gcdm_op_localize": [P, P, P, P, I64, I32, P],
    "gcdm_op_edge_features": [P, P, P, P, P, I64, P],
    "gcdm_op_orientations": [P, P, I64, P],
    "gcdm_op_centralize": [P, P, P, P, I64, I32, P],
    "gcdm_op_fc_edges": [P, P, I32, P, P, I64, P],
}
OPS_EXPORTS = list(OPS_SIGNATURES)
_ops_lib: Optional[C.CDLL] = None


def load_ops() -> C.CDLL:
    """Loads libgcdm_ops.so (module-level operators); raises if it has not been built -- there is no eager / CPU fallback."""
    global _ops_lib
    if _ops_lib is not None:
        return _ops_lib
    if not os.path.exists(OPS_LIB_PATH):
        raise RuntimeError(f"{OPS_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950)")
    lib = C.CDLL(OPS_LIB_PATH)
    for name, sig in OPS_SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = sig
        fn.restype = C.c_int
    _ops_lib = lib
    return lib


def load() -> C.CDLL:
    """Loads the library; raises (loudly) if it has not been built -- there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if os.environ.get("GCDM_HIP_LIB"):
        # a variant build (A/B, hazard runs): build() neither checks nor rebuilds it, so a library compiled from older sources would be loaded as it is
        import warnings
        warnings.warn(f"bio-diffusion_amd: loading the kernel library from GCDM_HIP_LIB={LIB_PATH} instead of the in-tree build "
                      f"({DEFAULT_LIB_PATH}); it is NOT checked against the sources of this tree", RuntimeWarning)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950).  bio-diffusion_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.gcdm_create.argtypes = [C.POINTER(GcdmConfig), C.POINTER(H)]
    lib.gcdm_destroy.argtypes = [H]
    lib.gcdm_last_error.argtypes = [H]
    lib.gcdm_last_error.restype = C.c_char_p
    lib.gcdm_set_weight.argtypes = [H, C.c_char_p, C.c_void_p, C.c_int64]
    lib.gcdm_finalize_weights.argtypes = [H]
    lib.gcdm_set_gamma.argtypes = [H, C.c_void_p, C.c_int64]
    lib.gcdm_plan_batch.argtypes = [H, C.c_int32, C.c_void_p]
    lib.gcdm_plan_batch_masked.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
    lib.gcdm_forward.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_forward_sc.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_sample_step.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.gcdm_sample_step_to.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
    lib.gcdm_sample_step_sc.argtypes = [H, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64,
                                        C.c_void_p, C.c_void_p]
    lib.gcdm_sample_final_sc.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_sample_final.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_sample_init.argtypes = [H, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.gcdm_encode_samples.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_unnormalize_z.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_inpaint_center.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.gcdm_inpaint_step.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.gcdm_inpaint_jump.argtypes = [H, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.gcdm_timestep_index.argtypes = [C.c_float, C.c_int32]
    lib.gcdm_timestep_index.restype = C.c_int32
    lib.gcdm_debug_read.argtypes = [H, C.c_char_p, C.c_void_p, C.c_int64]
    lib.gcdm_debug_read.restype = C.c_int64
    lib.gcdm_debug_set_layer_limit.argtypes = [H, C.c_int32]
    lib.gcdm_num_nodes.argtypes = [H]
    lib.gcdm_num_nodes.restype = C.c_int64
    lib.gcdm_num_edges.argtypes = [H]
    lib.gcdm_num_edges.restype = C.c_int64
    lib.gcdm_set_option.argtypes = [H, C.c_char_p, C.c_int32]
    lib.gcdm_get_option.argtypes = [H, C.c_char_p]
    lib.gcdm_profile_enable.argtypes = [H, C.c_int32]
    lib.gcdm_profile_edge_kernel_ms.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.gcdm_profile_node_kernel_ms.argtypes = [H, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    lib.gcdm_forward_flops_executed.argtypes = [H]
    lib.gcdm_forward_flops_executed.restype = C.c_double
    lib.gcdm_check_stability.argtypes = [C.POINTER(GcdmBondTables), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p]
    lib.gcdm_bond_orders.argtypes = [C.POINTER(GcdmBondTables), C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
    for name in EXPORTS:
        if getattr(lib, name).restype is C.c_int:
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def tensor_version(t) -> int:
    """In-place modification counter of a tensor, -1 for inference tensors (which do not track one; they are created inside
    torch.inference_mode() and are not modified in place by this package)."""
    try:
        return t._version
    except RuntimeError:
        return -1


class NativeError(RuntimeError):
    pass


def check(lib, handle, status, what: str):
    if status < 0:
        msg = lib.gcdm_last_error(handle)
        raise NativeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
    return status
