"""MI355X-native Gauss-Newton normal-equation assembly for SLAM pose solvers.

The product is two in-tree shared libraries: ``csrc/libnos_hip.so`` (C ABI of include/nos.h +
hand-written gfx950 kernels) and ``csrc/libnos_host.so`` (C++ drop-in solver classes with the
reference's Solve()/Options surface).  This package only binds them; importing it does not need
a GPU, using it does — there is no CPU fallback.
"""
from . import _lib
from .api import Context, NdtDataset, NdtIndexedDataset, ReprojDataset, make_loss
from .solvers import (MahalanobisDistanceMinimizerHip, MahalanobisDistanceMinimizerHip3DOF, Options, Pose,
                      ReprojectionErrorMinimizerHip)

__all__ = [
    "Context", "NdtDataset", "NdtIndexedDataset", "ReprojDataset", "make_loss", "Options", "Pose",
    "MahalanobisDistanceMinimizerHip", "MahalanobisDistanceMinimizerHip3DOF",
    "ReprojectionErrorMinimizerHip",
]
