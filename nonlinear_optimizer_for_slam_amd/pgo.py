"""Pose-graph optimisation handles over the C ABI (nos_pgo_*, include/nos.h) and a Python view of the
C++ drop-in class PoseGraphOptimizerHip (csrc/host/nos_pgo_solver.hpp)."""
import ctypes
import sys

import numpy as np

from . import _lib
from ._lib import c_double_p, check


def _dp(a):
    return a.ctypes.data_as(c_double_p)


class PoseGraph:
    """Device-resident pose graph (nos_pose_graph).  poses [n,7] = px py pz qw qx qy qz, meas [m,7] likewise."""

    def __init__(self, ctx, poses, ref, qry, meas, switch_init=None, switch_free=None, fixed=None):
        self._ctx = ctx
        self._lib = ctx._lib
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
        meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 7)
        ref = np.ascontiguousarray(ref, dtype=np.int32)
        qry = np.ascontiguousarray(qry, dtype=np.int32)
        self.n_poses, self.n_edges = poses.shape[0], ref.size
        ip = ctypes.POINTER(ctypes.c_int32)
        sw = None if switch_init is None else np.ascontiguousarray(switch_init, dtype=np.float64)
        swf = None if switch_free is None else np.ascontiguousarray(switch_free, dtype=np.uint8).tobytes()
        fx = None if fixed is None else np.ascontiguousarray(fixed, dtype=np.uint8).tobytes()
        h = ctypes.c_void_p()
        check(self._lib.nos_pgo_create(ctx.handle, self.n_poses, _dp(poses), self.n_edges, ref.ctypes.data_as(ip),
                                       qry.ctypes.data_as(ip), _dp(meas), None if sw is None else _dp(sw), swf, fx,
                                       ctypes.byref(h)), "nos_pgo_create")
        self._h = h
        ctx._adopt(self)

    @property
    def n_unknowns(self):
        return int(self._lib.nos_pgo_num_unknowns(self._h))

    def linearize(self):
        """→ (cost, |gradient|)."""
        c, g = ctypes.c_double(), ctypes.c_double()
        check(self._lib.nos_pgo_linearize(self._h, ctypes.byref(c), ctypes.byref(g)), "nos_pgo_linearize")
        return c.value, g.value

    def solve(self, lam, max_iterations=500, rel_tolerance=1e-10):
        """→ (pcg iterations, relative residual, |step|)."""
        it, res, sn = ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        check(self._lib.nos_pgo_solve(self._h, ctypes.c_double(lam), max_iterations, ctypes.c_double(rel_tolerance),
                                      ctypes.byref(it), ctypes.byref(res), ctypes.byref(sn)), "nos_pgo_solve")
        return it.value, res.value, sn.value

    def retract(self):
        check(self._lib.nos_pgo_retract(self._h), "nos_pgo_retract")

    def state(self):
        poses = np.zeros((self.n_poses, 7))
        sw = np.zeros(max(self.n_edges, 1))
        check(self._lib.nos_pgo_get_state(self._h, _dp(poses), _dp(sw)), "nos_pgo_get_state")
        return poses, sw[:self.n_edges]

    def vector(self, which):
        """which: "gradient" | "step" | "hdiag"."""
        idx = {"gradient": 0, "step": 1, "hdiag": 2}[which]
        out = np.zeros(21 * self.n_poses if idx == 2 else self.n_unknowns)
        check(self._lib.nos_pgo_get_vector(self._h, idx, _dp(out)), "nos_pgo_get_vector")
        return out

    def matvec(self, lam, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros_like(x)
        check(self._lib.nos_pgo_matvec(self._h, ctypes.c_double(lam), _dp(x), _dp(y)), "nos_pgo_matvec")
        return y

    def layout_info(self):
        """What the sweeps touch (nos_pgo_layout_info): poses, constraints, entries / block size / blocks / halo poses of
        the block-local product (0 when the owner-computes product runs), aggregates and PCR levels of the coarse level."""
        info = (ctypes.c_ulonglong * 8)()
        check(self._lib.nos_pgo_layout_info(self._h, info), "nos_pgo_layout_info")
        keys = ("poses", "constraints", "entries", "block_poses", "blocks", "halo_poses", "aggregates", "pcr_levels")
        return dict(zip(keys, (int(v) for v in info)))

    def time_sweep(self, which, lam=1e-3, repeats=20):
        """ms per device-resident sweep (nos_pgo_time_sweep): which = "matvec" (the product of a PCG iteration) or
        "linearize" (the linearisation kernels, without the host's scalar readbacks)."""
        ms = ctypes.c_double()
        check(self._lib.nos_pgo_time_sweep(self._h, {"matvec": 0, "linearize": 1}[which], ctypes.c_double(lam), int(repeats),
                                           ctypes.byref(ms)), "nos_pgo_time_sweep")
        return ms.value

    def optimize(self, max_iterations=40, gradient_tolerance=1e-6, parameter_tolerance=1e-6, pcg_iterations=500,
                 pcg_tolerance=1e-10):
        """The reference's LM loop shape (always step; lambda x2 / x0.6 on the cost, clamp [1e-6, 1e-2]) driven
        from Python — the C++ class PoseGraphOptimizerHip runs the same loop natively."""
        lam, prev = 1e-3, np.finfo(np.float64).max
        hist = []
        it = 0
        for it in range(max_iterations):
            cost, gnorm = self.linearize()
            pcg_it, res, step = self.solve(lam, pcg_iterations, pcg_tolerance)
            self.retract()
            hist.append((cost, gnorm, step, pcg_it, res))
            if step < parameter_tolerance or gnorm < gradient_tolerance:
                break
            lam = min(max(lam * (2.0 if cost > prev else 0.6), 1e-6), 1e-2)
            prev = cost
        return it, hist

    def close(self):
        if self._h:
            self._lib.nos_pgo_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class PoseGraphOptimizerHip:
    """Python view of the C++ drop-in class (csrc/host/nos_pgo_solver.hpp) with the reference's surface:
    SetPose(index, pose7) / SetConstraint(ref, qry, rel_pose7, is_loop) / SetPoseConstant(index) / Solve(options).
    pose7 = px py pz qw qx qy qz.  After a successful Solve the registered pose arrays hold the optimum."""

    def __init__(self, pcg_max_iterations=2000, pcg_tolerance=1e-10):
        self._poses = {}
        self._constraints = []
        self._fixed = []
        self.pcg_max_iterations = pcg_max_iterations
        self.pcg_tolerance = pcg_tolerance
        self.report = None
        self.switch_parameters = None

    def SetPose(self, pose_index, pose7):
        self._poses[int(pose_index)] = pose7  # kept by reference, overwritten on success

    def SetConstraint(self, reference_pose_index, query_pose_index, relative_pose7, is_loop=False):
        self._constraints.append((int(reference_pose_index), int(query_pose_index),
                                  np.asarray(relative_pose7, dtype=np.float64).reshape(7), bool(is_loop)))

    def SetPoseConstant(self, pose_index):
        self._fixed.append(int(pose_index))

    def Solve(self, options):
        from .synth import host_lib
        idx = np.array(sorted(self._poses), dtype=np.int32)
        poses = np.ascontiguousarray(np.stack([np.asarray(self._poses[i], dtype=np.float64) for i in idx]))
        m = len(self._constraints)
        ref = np.array([c[0] for c in self._constraints], dtype=np.int32)
        qry = np.array([c[1] for c in self._constraints], dtype=np.int32)
        meas = np.ascontiguousarray(np.stack([c[2] for c in self._constraints])) if m else np.zeros((1, 7))
        loop = np.array([c[3] for c in self._constraints], dtype=np.uint8).tobytes() if m else b"\0"
        fixed = np.array(self._fixed, dtype=np.int32)
        sw = np.ones(max(m, 1))
        rep = np.zeros(6)
        ip = ctypes.POINTER(ctypes.c_int)
        ok = host_lib().nos_host_pgo_solve(
            ctypes.c_size_t(idx.size), idx.ctypes.data_as(ip), _dp(poses), ctypes.c_size_t(m), ref.ctypes.data_as(ip),
            qry.ctypes.data_as(ip), _dp(meas), loop, ctypes.c_size_t(fixed.size), fixed.ctypes.data_as(ip),
            ctypes.c_int(options.max_iterations), ctypes.c_double(options.gradient_tolerance),
            ctypes.c_double(options.parameter_tolerance), ctypes.c_int(self.pcg_max_iterations),
            ctypes.c_double(self.pcg_tolerance), _dp(sw), _dp(rep))
        self.report = {"iterations": int(rep[0]), "initial_cost": rep[1], "final_cost": rep[2],
                       "final_gradient_norm": rep[3], "total_pcg_iterations": int(rep[4]), "status": int(rep[5])}
        if ok:
            for k, i in enumerate(idx):
                self._poses[int(i)][:] = poses[k]
            self.switch_parameters = sw[:m].copy()
        return bool(ok)
