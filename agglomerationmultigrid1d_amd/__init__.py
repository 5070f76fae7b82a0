"""agglomerationmultigrid1d_amd -- MI355X-native V-cycle hot path of AgglomerationMultigrid1D.

Host mirror of the reference's operator interface (api.py) over the C ABI of libaggmg_hip.so
(include/aggmg_hip.h, csrc/).  Importing the package does not touch the GPU; the first
Context() does, and fails loudly when the HIP library or device is missing."""
from ._lib import (AggmgError, ArgumentError, DimensionMismatch, HipError, SingularException,
                   UnsupportedError, LIB_PATH, SYMBOLS)
from .api import (AbstractSmoother, AdditiveSchwarzSmoother, BlockDiagonal, BlockDiagonalLU, BlockGaussSeidel,
                  BlockJacobi, Context,
                  DeviceOperator,
                  DeviceVector, HybridSchwarzSmoother, JacobiSmoother, MeshHierarchy,
                  apply_smoother, cg_smoother, default_context, dg_smoother,
                  dot, iterative_smoother_solve, ldiv, multigrid, multigrid_dev, multigrid_v_cycle, norm2, pcg,
                  prolong_add, residual, restrict, smooth, smoother_launch_bytes, smoother_solve_dev)

from . import interpolation
from .interpolation import (aggdg_aggdg_interpolation, aggdg_cg_interpolation, aggdg_dg_interpolation,
                            aggdg_dg_interpolation2, cg_cg_interpolation, dg_cg_interpolation, dg_dg_interpolation)

__all__ = [n for n in dir() if not n.startswith("_")]
