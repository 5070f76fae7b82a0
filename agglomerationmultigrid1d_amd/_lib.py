"""ctypes binding of libaggmg_hip.so -- exactly the symbols declared in include/aggmg_hip.h.

There is no CPU fallback: if the shared library is missing or no HIP device is usable, the
product path raises (ImportError / AggmgError).  Build with `python __graft_entry__.py build`
(or `make -C agglomerationmultigrid1d_amd/csrc`)."""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# AGGMG_HIP_LIB points at an alternative build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("AGGMG_HIP_LIB") or os.path.join(_HERE, "libaggmg_hip.so")

AGGMG_OK = 0
ERR_ARGUMENT, ERR_DIMENSION, ERR_SINGULAR, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4, -5
OP_STIFFNESS, OP_TRANSFER = 0, 1
OPT_SYMMETRIC_PACKING = 1
OPT_COARSE_CHUNK_LOG2 = 2
OPT_DETECT_CHAIN = 3
OPT_PAIR_LEVELS = 4
OPT_MG_CHECKPOINT = 5
PROFILE_NTAGS = 256
KIND_FUSED_DOWN, KIND_FUSED_UP, KIND_SMOOTH, KIND_RESIDUAL, KIND_RESTRICT, KIND_PROLONG, \
    KIND_JACOBI, KIND_BLOCK_APPLY, KIND_COARSE, KIND_FUSED_MID = range(10)
KIND_NAMES = ["fused_down", "fused_up", "smooth", "residual", "restrict", "prolong", "jacobi",
              "block_apply", "coarse", "fused_mid"]
COARSE_HOST_BANDED, COARSE_DEVICE_CR, COARSE_AUTO, COARSE_EXTERNAL = 0, 1, 2, 3
# aggmg_hier_set_restriction modes (include/aggmg_hip.h)
RESTRICT_EXPLICIT, RESTRICT_PRECONDITIONED = 0, 1
RESTRICT_PRECONDITIONED_MAX_ELEMS = 1 << 21
# aggmg_hier_level_kind
LEVEL_GENERIC, LEVEL_FUSED_BTD, LEVEL_FUSED_CHAIN, LEVEL_COARSEST = 0, 1, 2, 3
RCCL_ID_BYTES = 128
DIST_X0_GHOSTS_VALID, DIST_OVERLAP_NEXT, DIST_GRAPH = 1, 2, 4
ALLGATHER_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p)
# aggmg_sendrecv_fn(user, nops, peer[], is_send[], dev_ptr[], count[], hip_stream)
SENDRECV_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_void_p), POINTER(c_int64),
                               c_void_p)
LEVEL_KIND_NAMES = ["generic", "fused_btd", "fused_chain", "coarsest"]


class AggmgError(RuntimeError):
    """Base of the error classes of the C ABI (status < 0)."""
    status = None


class ArgumentError(AggmgError, ValueError):
    """Julia ArgumentError (src/mesh_heirarchy.jl:33-39,142-148)"""
    status = ERR_ARGUMENT


class DimensionMismatch(AggmgError, ValueError):
    """Julia DimensionMismatch (src/block_diagonal.jl:138,167-169,300-302)"""
    status = ERR_DIMENSION


class SingularException(AggmgError, ArithmeticError):
    """LinearAlgebra.SingularException from la.lu (src/smoother.jl:160)"""
    status = ERR_SINGULAR


class HipError(AggmgError):
    status = ERR_HIP


class UnsupportedError(AggmgError):
    """Julia ErrorException (`error(...)` paths)"""
    status = ERR_UNSUPPORTED


_ERR = {c.status: c for c in (ArgumentError, DimensionMismatch, SingularException, HipError,
                              UnsupportedError)}

# name -> (restype, argtypes); every symbol of include/aggmg_hip.h
_P = c_void_p
_PD = POINTER(c_double)
SYMBOLS = {
    "aggmg_create": (c_int, [c_int, POINTER(_P)]),
    "aggmg_destroy": (c_int, [_P]),
    "aggmg_last_error": (c_char_p, [_P]),
    "aggmg_set_stream": (c_int, [_P, _P]),
    "aggmg_reset_stream": (c_int, [_P]),
    "aggmg_synchronize": (c_int, [_P]),
    "aggmg_set_option": (c_int, [_P, c_int, c_int]),
    "aggmg_dev_alloc": (c_int, [_P, c_int64, POINTER(_P)]),
    "aggmg_dev_free": (c_int, [_P, _P]),
    "aggmg_memcpy_h2d": (c_int, [_P, _P, _P, c_int64]),
    "aggmg_memcpy_d2h": (c_int, [_P, _P, _P, c_int64]),
    "aggmg_csc_upload": (c_int, [_P, c_int64, c_int64, POINTER(c_int64), POINTER(c_int64), _PD,
                                 c_int, c_int, POINTER(_P)]),
    "aggmg_op_free": (c_int, [_P, _P]),
    "aggmg_op_shape": (c_int, [_P, _P, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    "aggmg_op_download": (c_int, [_P, _P, c_int, POINTER(c_int32), POINTER(c_int32), _PD]),
    "aggmg_op_release_host": (c_int, [_P, _P]),
    "aggmg_bd_sp_apply": (c_int, [_P, _P, _P, c_int, POINTER(_P)]),
    "aggmg_sp_matmul": (c_int, [_P, _P, _P, c_int, POINTER(_P)]),
    "aggmg_sp_sub": (c_int, [_P, _P, _P, c_int, POINTER(_P)]),
    "aggmg_op_transpose": (c_int, [_P, _P, c_int, POINTER(_P)]),
    "aggmg_op_download_csc": (c_int, [_P, _P, POINTER(c_int32), POINTER(c_int32), _PD]),
    "aggmg_smoother_download_blocks": (c_int, [_P, _P, _PD]),
    "aggmg_debug_scan_counts": (c_int, [_P, POINTER(c_int32), c_int64, POINTER(c_int64)]),
    "aggmg_debug_stream_copy": (c_int, [_P, _P, _P, c_int64, c_int, c_int, POINTER(c_double)]),
    "aggmg_blockjacobi_setup": (c_int, [_P, _P, c_int64, c_int64, POINTER(c_int64), c_int, c_int,
                                        POINTER(_P)]),
    "aggmg_blockdiag_setup": (c_int, [_P, c_int64, c_int64, _PD, c_int, POINTER(_P)]),
    "aggmg_jacobi_setup": (c_int, [_P, _P, POINTER(_P)]),
    "aggmg_jacobi_setup_elements": (c_int, [_P, _P, c_int64, c_int64, POINTER(c_int64), c_int, POINTER(_P)]),
    "aggmg_smoother_free": (c_int, [_P, _P]),
    "aggmg_smoother_apply": (c_int, [_P, _P, _PD, c_int64, c_int64, c_double, _PD]),
    "aggmg_smoother_is_structured": (c_int, [_P, _P, POINTER(c_int)]),
    "aggmg_smooth": (c_int, [_P, _P, _P, _PD, _PD, c_double, c_int]),
    "aggmg_residual": (c_int, [_P, _P, _PD, _PD, _PD]),
    "aggmg_restrict": (c_int, [_P, _P, _PD, _PD]),
    "aggmg_prolong_add": (c_int, [_P, _P, _PD, _PD]),
    "aggmg_smooth_dev": (c_int, [_P, _P, _P, _P, _P, c_double, c_int, _P]),
    "aggmg_residual_dev": (c_int, [_P, _P, _P, _P, _P]),
    "aggmg_restrict_dev": (c_int, [_P, _P, _P, _P]),
    "aggmg_prolong_add_dev": (c_int, [_P, _P, _P, _P]),
    "aggmg_hier_create": (c_int, [_P, c_int, POINTER(_P), POINTER(_P), POINTER(_P), c_int,
                                  POINTER(_P)]),
    "aggmg_hier_free": (c_int, [_P, _P]),
    "aggmg_vcycle": (c_int, [_P, _P, _PD, _PD, c_int, c_int, c_double, _PD]),
    "aggmg_vcycle_dev": (c_int, [_P, _P, _P, _P, c_int, c_int, c_double, _P]),
    "aggmg_vcycles_dev": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_double, _P]),
    "aggmg_hier_set_restriction": (c_int, [_P, _P, c_int]),
    "aggmg_hier_get_restriction": (c_int, [_P, _P, POINTER(c_int)]),
    "aggmg_vcycle_down_dev": (c_int, [_P, _P, _P, _P, c_int, c_double]),
    "aggmg_vcycle_up_dev": (c_int, [_P, _P, _P, c_int, c_double, _P]),
    "aggmg_vcycle_up_coarse_dev": (c_int, [_P, _P, _P, c_int, c_double, c_int, c_int64, c_int64]),
    "aggmg_vcycle_up_split_dev": (c_int, [_P, _P, _P, c_int, c_double, _P, c_int64, c_int64, c_int]),
    "aggmg_hier_coarse_buffers": (c_int, [_P, _P, POINTER(_P), POINTER(_P), POINTER(c_int64)]),
    "aggmg_coarse_plan": (c_int, [_P, _P, POINTER(c_int), POINTER(c_int64), POINTER(c_int), POINTER(c_int64)]),
    "aggmg_coarse_chunk_forward_dev": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P]),
    "aggmg_coarse_boundary_solve_dev": (c_int, [_P, _P, _P, _P, _P]),
    "aggmg_coarse_chunk_backward_dev": (c_int, [_P, _P, _P, c_int64, c_int64, _P, _P]),
    "aggmg_hier_level_kind": (c_int, [_P, _P, c_int, POINTER(c_int)]),
    "aggmg_host_register": (c_int, [_P, _P, c_int64]),
    "aggmg_host_unregister": (c_int, [_P, _P]),
    "aggmg_host_alloc": (c_int, [_P, c_int64, POINTER(_P)]),
    "aggmg_host_free": (c_int, [_P, _P]),
    "aggmg_hier_level_paired": (c_int, [_P, _P, c_int, c_int, POINTER(c_int)]),
    "aggmg_hier_launch_bytes": (c_int, [_P, _P, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "aggmg_smoother_launch_bytes": (c_int, [_P, _P, _P, c_int, POINTER(c_int64), POINTER(c_int64)]),
    "aggmg_hier_coarse_info": (c_int, [_P, _P, POINTER(c_int), POINTER(c_int), POINTER(c_double)]),
    "aggmg_hier_coarse_probe": (c_int, [_P, _P, POINTER(c_double)]),
    "aggmg_hier_coarse_tail": (c_int, [_P, _P, POINTER(c_int), POINTER(c_int64)]),
    "aggmg_hier_last_coarse_ms": (c_int, [_P, _P, POINTER(c_double)]),
    "aggmg_copy_segments_dev": (c_int, [_P, c_int, POINTER(_P), POINTER(_P), POINTER(c_int64), POINTER(c_int64),
                                        POINTER(c_int64), POINTER(c_int64)]),
    "aggmg_dist_create": (c_int, [_P, _P, _P, c_int, c_int, c_int, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64),
                                  POINTER(c_int64), POINTER(c_int64), POINTER(c_int32), POINTER(c_int32), POINTER(_P)]),
    "aggmg_dist_free": (c_int, [_P, _P]),
    "aggmg_rccl_unique_id": (c_int, [_P, _P, c_int]),
    "aggmg_rccl_available": (c_int, [c_char_p, c_int]),
    "aggmg_dist_init_rccl": (c_int, [_P, _P, _P, c_int, POINTER(c_int)]),
    "aggmg_dist_set_allgather": (c_int, [_P, _P, _P, _P]),
    "aggmg_dist_set_loopback": (c_int, [_P, _P]),
    "aggmg_dist_set_sendrecv": (c_int, [_P, _P, _P]),
    "aggmg_dist_set_neighbor_layout": (c_int, [_P, _P, c_int] + [c_int, POINTER(c_int64), POINTER(c_int64)] * 4),
    "aggmg_dist_set_exchange_layout": (c_int, [_P, _P, c_int, c_int64, c_int] + [POINTER(c_int64)] * 3 + [c_int] +
                                       [POINTER(c_int64)] * 3 + [c_int] + [POINTER(c_int64)] * 3),
    "aggmg_dist_allgather_dev": (c_int, [_P, _P, _P, _P, c_int64]),
    "aggmg_dist_exchange_ghosts_dev": (c_int, [_P, _P, _P, c_int]),
    "aggmg_dist_vcycle_dev": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_double, c_int]),
    "aggmg_dist_set_coarse_overlap": (c_int, [_P, _P, c_int]),
    "aggmg_dist_info": (c_int, [_P, _P, POINTER(c_int64), POINTER(c_int), POINTER(c_int)]),
    "aggmg_dist_graph_info": (c_int, [_P, _P, POINTER(c_int64), POINTER(c_int), POINTER(c_int)]),
    "aggmg_dot_dev": (c_int, [_P, _P, _P, c_int64, POINTER(c_double)]),
    "aggmg_norm2_dev": (c_int, [_P, _P, c_int64, POINTER(c_double)]),
    "aggmg_residual_norm_dev": (c_int, [_P, _P, _P, _P, POINTER(c_double)]),
    "aggmg_multigrid_dev": (c_int, [_P, _P, _P, _P, c_int, c_double, c_int, c_int, c_int, c_double, _P,
                                    _PD, POINTER(c_int), POINTER(c_int), _P, _PD]),
    "aggmg_smoother_solve_dev": (c_int, [_P, _P, _P, _P, _P, c_int, c_double, c_double, c_int, _P,
                                         _PD, POINTER(c_int), POINTER(c_int), _P, _PD]),
    "aggmg_pcg_dev": (c_int, [_P, _P, _P, _P, c_int, c_double, c_int, c_int, c_double, _PD, POINTER(c_int)]),
    "aggmg_profile_enable": (c_int, [_P, c_int]),
    "aggmg_profile_collect": (c_int, [_P, _PD, POINTER(c_int64)]),
    "aggmg_version": (c_char_p, []),
}

_lib = None


def load():
    """dlopen the in-tree library and attach prototypes.  Raises ImportError when it is absent:
    the product has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` from the repo root. "
            "agglomerationmultigrid1d_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, ctx_handle=None):
    if status == AGGMG_OK:
        return
    msg = load().aggmg_last_error(ctx_handle)
    msg = msg.decode() if msg else ""
    raise _ERR.get(status, AggmgError)(f"[aggmg status {status}] {msg}")
