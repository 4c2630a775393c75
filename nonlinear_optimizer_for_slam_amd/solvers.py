"""Python view of the C++ drop-in solver classes (csrc/host/nos_hip_solvers.hpp).

Same names and argument meaning as the reference's C++ API — Options, SetLossFunction, Solve —
so tests read like the reference's own test drivers.  All work happens in libnos_host.so
(C++ LM loop, Eigen-style 6x6 LDLT on the host) and libnos_hip.so (HIP kernels).
"""
import ctypes

import numpy as np

from . import _lib
from .api import make_loss
from .synth import host_lib


class Options:
    """nonlinear_optimizer::Options (NO/options.h:15-28); analytic solvers read only these."""

    def __init__(self, max_iterations=40, gradient_tolerance=1e-6, parameter_tolerance=1e-6):
        self.max_iterations = max_iterations
        self.gradient_tolerance = gradient_tolerance
        self.parameter_tolerance = parameter_tolerance


class Pose:
    """Isometry: R [3,3], t [3]."""

    def __init__(self, R=None, t=None):
        self.R = np.eye(3) if R is None else np.array(R, dtype=np.float64).reshape(3, 3)
        self.t = np.zeros(3) if t is None else np.array(t, dtype=np.float64).reshape(3)

    def inverse(self):
        return Pose(self.R.T, -self.R.T @ self.t)


class SolveReport:
    def __init__(self, arr):
        self.iterations = int(arr[0])
        self.printed_cost = float(arr[1])
        self.last_cost = float(arr[2])
        self.final_lambda = float(arr[3])
        self.status = int(arr[4])


def _planes_arg(planes, count):
    planes = np.ascontiguousarray(planes, dtype=np.float64)
    if planes.ndim != 2 or planes.shape[0] != count:
        raise ValueError("planes must be [%d, n]" % count)
    arr = (_lib.c_double_p * count)(*[planes[k].ctypes.data_as(_lib.c_double_p) for k in range(count)])
    return arr, planes


class _HipSolverBase:
    def __init__(self, device_ids=(0,), dtype="f64", print_cost_line=False, device_loop=True, simd_class=False,
                 simd_class_threads=1):
        """device_loop: run the whole LM loop device-resident (nos_*_solve, the default on single-device contexts);
        False = the host loop around nos_*_accumulate.  simd_class: the semantics of the reference's fp32 classes
        (HipOptions::simd_class: fp32, tail drop to T*floor(floor(N/8)/T)*8 with T = simd_class_threads, float lambda /
        previous_cost for NDT, depth > 0 weight mask and float 1/fx for reprojection)."""
        self._loss = None
        self.simd_class = bool(simd_class)
        self.simd_class_threads = max(1, int(simd_class_threads))
        self.device_loop = device_loop
        self.device_ids = tuple(device_ids)
        self.dtype = dtype
        self.print_cost_line = print_cost_line
        self.report = None

    def _flags(self):
        return (int(bool(self.print_cost_line)) | (0 if self.device_loop else 2) | (4 if self.simd_class else 0)
                | ((self.simd_class_threads & 0xff) << 8))

    def SetLossFunction(self, loss):
        """loss: None | ("exponential", c1, c2) | ("huber", threshold)."""
        self._loss = loss

    def SetMultiThreadExecutor(self, executor):
        """Accepted for API compatibility; the GPU grid replaces the thread pool."""


class MahalanobisDistanceMinimizerHip(_HipSolverBase):
    """6-DoF NDT scan-to-map pose (↔ MahalanobisDistanceMinimizerAnalytic[SIMD])."""
    _dof = 6

    def Solve(self, options, planes, pose, repeat_solves=1):
        arr, keep = _planes_arg(planes, 15)
        l = make_loss(self._loss)
        t = np.ascontiguousarray(pose.t, dtype=np.float64).copy()
        R = np.ascontiguousarray(pose.R, dtype=np.float64).reshape(-1).copy()
        rep = np.zeros(5)
        ids = (ctypes.c_int * len(self.device_ids))(*self.device_ids)
        ok = host_lib().nos_host_ndt_solve(
            ctypes.c_int(self._dof), ctypes.c_size_t(keep.shape[1]), arr, ctypes.c_int(l.kind),
            ctypes.c_double(l.a), ctypes.c_double(l.b), ctypes.c_int(options.max_iterations),
            ctypes.c_double(options.gradient_tolerance), ctypes.c_double(options.parameter_tolerance),
            ctypes.c_int(_lib.NOS_F32 if self.dtype == "f32" else _lib.NOS_F64), ids, len(self.device_ids),
            ctypes.c_int(self._flags()), ctypes.c_int(repeat_solves),
            t.ctypes.data_as(_lib.c_double_p), R.ctypes.data_as(_lib.c_double_p),
            rep.ctypes.data_as(_lib.c_double_p))
        self.report = SolveReport(rep)
        if ok:
            pose.t = t
            pose.R = R.reshape(3, 3)
        return bool(ok)


    def SolveDataset(self, options, dataset, pose):
        """Solve on a device-resident NdtDataset (e.g. the matcher's output); additive API."""
        l = make_loss(self._loss)
        t = np.ascontiguousarray(pose.t, dtype=np.float64).copy()
        R = np.ascontiguousarray(pose.R, dtype=np.float64).reshape(-1).copy()
        rep = np.zeros(5)
        ok = host_lib().nos_host_ndt_solve_dataset(
            ctypes.c_int(self._dof), dataset._h, ctypes.c_int(l.kind), ctypes.c_double(l.a), ctypes.c_double(l.b),
            ctypes.c_int(options.max_iterations), ctypes.c_double(options.gradient_tolerance),
            ctypes.c_double(options.parameter_tolerance), ctypes.c_int(self._flags()),
            t.ctypes.data_as(_lib.c_double_p), R.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))
        self.report = SolveReport(rep)
        if ok:
            pose.t = t
            pose.R = R.reshape(3, 3)
        return bool(ok)


class MahalanobisDistanceMinimizerHip3DOF(MahalanobisDistanceMinimizerHip):
    """Planar (x, y, yaw) NDT pose (↔ MahalanobisDistanceMinimizerAnalytic3DOF[SIMD])."""
    _dof = 3


class ReprojectionErrorMinimizerHip(_HipSolverBase):
    """6-DoF pose from 3D↔2D correspondences (↔ ReprojectionErrorMinimizerAnalytic[SIMD])."""

    def Solve(self, options, planes, camera_intrinsics, pose):
        """camera_intrinsics: (fx, fy, cx, cy)."""
        arr, keep = _planes_arg(planes, 5)
        l = make_loss(self._loss)
        intr = np.ascontiguousarray(camera_intrinsics, dtype=np.float64).reshape(4)
        t = np.ascontiguousarray(pose.t, dtype=np.float64).copy()
        R = np.ascontiguousarray(pose.R, dtype=np.float64).reshape(-1).copy()
        rep = np.zeros(5)
        ids = (ctypes.c_int * len(self.device_ids))(*self.device_ids)
        ok = host_lib().nos_host_reproj_solve(
            ctypes.c_size_t(keep.shape[1]), arr, intr.ctypes.data_as(_lib.c_double_p), ctypes.c_int(l.kind),
            ctypes.c_double(l.a), ctypes.c_double(l.b), ctypes.c_int(options.max_iterations),
            ctypes.c_double(options.gradient_tolerance), ctypes.c_double(options.parameter_tolerance),
            ctypes.c_int(_lib.NOS_F32 if self.dtype == "f32" else _lib.NOS_F64), ids, len(self.device_ids),
            ctypes.c_int(self._flags()), t.ctypes.data_as(_lib.c_double_p),
            R.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))
        self.report = SolveReport(rep)
        if ok:
            pose.t = t
            pose.R = R.reshape(3, 3)
        return bool(ok)


def describe_loss(loss):
    """Round-trip a loss through the C++ LossFunction objects and the descriptor recovery
    (DescribeLossFunction); CPU only."""
    l = make_loss(loss)
    k = ctypes.c_int()
    a = ctypes.c_double()
    b = ctypes.c_double()
    ok = host_lib().nos_host_describe_loss(ctypes.c_int(l.kind), ctypes.c_double(l.a), ctypes.c_double(l.b),
                                           ctypes.byref(k), ctypes.byref(a), ctypes.byref(b))
    return bool(ok), k.value, a.value, b.value
