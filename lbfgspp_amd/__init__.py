"""lbfgspp_amd -- MI355X-native L-BFGS / L-BFGS-B hot path behind the yixuan/LBFGSpp solver API.

csrc/            HIP kernels (gfx950) + C ABI (include/lbfgsx.h, include/lbfgsx_solver.h)
solver.py        Python mirror of LBFGSSolver / LBFGSBSolver / LBFGSParam over that C ABI
"""
from ._lib import (F32, F64, LS_BACKTRACKING, LS_BRACKETING, LS_MORE_THUENTE, LS_NOCEDAL_WRIGHT,
                   NativeLibraryMissing, load)
from .solver import (DiagQuadratic, ExtendedRosenbrock, LBFGSBParam, LBFGSBSolver, LBFGSParam, LBFGSSolver,
                     TraceBuffer)

__all__ = ["F32", "F64", "LS_BACKTRACKING", "LS_BRACKETING", "LS_MORE_THUENTE", "LS_NOCEDAL_WRIGHT",
           "NativeLibraryMissing", "load", "DiagQuadratic", "ExtendedRosenbrock", "LBFGSBParam", "LBFGSParam",
           "LBFGSSolver", "LBFGSBSolver", "TraceBuffer"]
