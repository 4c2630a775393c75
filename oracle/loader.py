"""ctypes access to the CPU oracle — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module (see oracle/nos_oracle.h).  It never touches the GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "build")

c_double_p = ctypes.POINTER(ctypes.c_double)


class OracleLoss(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("a", ctypes.c_double), ("b", ctypes.c_double)]


class OracleOptions(ctypes.Structure):
    _fields_ = [
        ("max_iterations", ctypes.c_int),
        ("gradient_tolerance", ctypes.c_double),
        ("parameter_tolerance", ctypes.c_double),
        ("linear_solver", ctypes.c_int),
    ]


class OracleReport(ctypes.Structure):
    _fields_ = [
        ("iterations", ctypes.c_int),
        ("printed_cost", ctypes.c_double),
        ("last_cost", ctypes.c_double),
        ("final_lambda", ctypes.c_double),
    ]


def build(force=False):
    """Compile the oracle libraries with the committed Makefile (gcc only)."""
    need = force or not all(
        os.path.exists(os.path.join(_BUILD, f))
        for f in ("libnos_oracle.so", "libnos_oracle_avx.so", "libnos_scene_oracle.so", "libnos_pgo_oracle.so")
    )
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))


_lib = None
_avx = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(os.path.join(_BUILD, "libnos_oracle.so"))
        _lib.oracle_ndt6_accumulate.restype = None
        _lib.oracle_ndt3_accumulate.restype = None
        _lib.oracle_reproj_accumulate.restype = None
    return _lib


def avx():
    global _avx
    if _avx is None:
        build()
        _avx = ctypes.CDLL(os.path.join(_BUILD, "libnos_oracle_avx.so"))
        _avx.oracle_avx_ndt6_accumulate.restype = ctypes.c_int
    return _avx


def _planes_arg(planes, count):
    """planes: array [count, n] float64 C-contiguous → (ctypes array of row pointers, keepalive)."""
    planes = np.ascontiguousarray(planes, dtype=np.float64)
    assert planes.ndim == 2 and planes.shape[0] == count, planes.shape
    arr = (c_double_p * count)()
    for k in range(count):
        arr[k] = planes[k].ctypes.data_as(c_double_p)
    return arr, planes


def _vec(x, n):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    assert a.size == n, (a.size, n)
    return a


def make_loss(loss):
    """loss: None | ("exponential", c1, c2) | ("huber", th) | ("none",)."""
    if loss is None or loss[0] == "none":
        return OracleLoss(0, 0.0, 0.0)
    if loss[0] == "exponential":
        return OracleLoss(1, float(loss[1]), float(loss[2]))
    if loss[0] == "huber":
        return OracleLoss(2, float(loss[1]), 0.0)
    raise ValueError(loss)


def ndt6_accumulate(planes, R, t, loss=None):
    arr, keep = _planes_arg(planes, 15)
    R = _vec(R, 9)
    t = _vec(t, 3)
    out = np.zeros(28)
    l = make_loss(loss)
    lib().oracle_ndt6_accumulate(
        ctypes.c_size_t(keep.shape[1]), arr, R.ctypes.data_as(c_double_p),
        t.ctypes.data_as(c_double_p), ctypes.byref(l), out.ctypes.data_as(c_double_p))
    return out


def ndt6_accumulate_f32lanes(planes, R, t, loss=None, drop_tail=False):
    arr, keep = _planes_arg(planes, 15)
    R = _vec(R, 9)
    t = _vec(t, 3)
    out = np.zeros(28)
    l = make_loss(loss)
    lib().oracle_ndt6_accumulate_f32lanes(
        ctypes.c_size_t(keep.shape[1]), arr, R.ctypes.data_as(c_double_p),
        t.ctypes.data_as(c_double_p), ctypes.byref(l), ctypes.c_int(int(drop_tail)),
        out.ctypes.data_as(c_double_p))
    return out


def ndt3_accumulate(planes, R2, t2, loss=None):
    arr, keep = _planes_arg(planes, 15)
    R2 = _vec(R2, 4)
    t2 = _vec(t2, 2)
    out = np.zeros(10)
    l = make_loss(loss)
    lib().oracle_ndt3_accumulate(
        ctypes.c_size_t(keep.shape[1]), arr, R2.ctypes.data_as(c_double_p),
        t2.ctypes.data_as(c_double_p), ctypes.byref(l), out.ctypes.data_as(c_double_p))
    return out


def reproj_accumulate(planes, R, t, intr, loss=None, min_depth=0.03):
    arr, keep = _planes_arg(planes, 5)
    R = _vec(R, 9)
    t = _vec(t, 3)
    intr = _vec(intr, 4)
    out = np.zeros(28)
    l = make_loss(loss)
    lib().oracle_reproj_accumulate(
        ctypes.c_size_t(keep.shape[1]), arr, R.ctypes.data_as(c_double_p),
        t.ctypes.data_as(c_double_p), intr.ctypes.data_as(c_double_p), ctypes.byref(l),
        ctypes.c_double(min_depth), out.ctypes.data_as(c_double_p))
    return out


def _options(max_iterations=40, gradient_tolerance=1e-6, parameter_tolerance=1e-6,
             linear_solver=0):
    return OracleOptions(int(max_iterations), float(gradient_tolerance),
                         float(parameter_tolerance), int(linear_solver))


def _solve_result(t, R, rep):
    return {
        "t": t.copy(), "R": R.reshape(3, 3).copy(), "iterations": rep.iterations,
        "printed_cost": rep.printed_cost, "last_cost": rep.last_cost,
        "final_lambda": rep.final_lambda,
    }


def ndt6_solve(planes, t0, R0, loss=None, **opt):
    arr, keep = _planes_arg(planes, 15)
    t = _vec(t0, 3).copy()
    R = _vec(R0, 9).copy()
    o = _options(**opt)
    l = make_loss(loss)
    rep = OracleReport()
    lib().oracle_ndt6_solve(
        ctypes.c_size_t(keep.shape[1]), arr, ctypes.byref(o), ctypes.byref(l),
        t.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p), ctypes.byref(rep))
    return _solve_result(t, R, rep)


def ndt3_solve(planes, t0, R0, loss=None, **opt):
    arr, keep = _planes_arg(planes, 15)
    t = _vec(t0, 3).copy()
    R = _vec(R0, 9).copy()
    o = _options(**opt)
    l = make_loss(loss)
    rep = OracleReport()
    lib().oracle_ndt3_solve(
        ctypes.c_size_t(keep.shape[1]), arr, ctypes.byref(o), ctypes.byref(l),
        t.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p), ctypes.byref(rep))
    return _solve_result(t, R, rep)


def reproj_solve(planes, intr, t0, R0, loss=None, min_depth=0.03, **opt):
    arr, keep = _planes_arg(planes, 5)
    intr = _vec(intr, 4)
    t = _vec(t0, 3).copy()
    R = _vec(R0, 9).copy()
    o = _options(**opt)
    l = make_loss(loss)
    rep = OracleReport()
    lib().oracle_reproj_solve(
        ctypes.c_size_t(keep.shape[1]), arr, intr.ctypes.data_as(c_double_p), ctypes.byref(o),
        ctypes.byref(l), ctypes.c_double(min_depth), t.ctypes.data_as(c_double_p),
        R.ctypes.data_as(c_double_p), ctypes.byref(rep))
    return _solve_result(t, R, rep)


def lm_step6(out28, lam, linear_solver=0):
    out28 = _vec(out28, 28)
    step = np.zeros(6)
    lib().oracle_lm_step6(out28.ctypes.data_as(c_double_p), ctypes.c_double(lam),
                          ctypes.c_int(linear_solver), step.ctypes.data_as(c_double_p))
    return step


def lm_step3(out10, lam):
    out10 = _vec(out10, 10)
    step = np.zeros(3)
    lib().oracle_lm_step3(out10.ctypes.data_as(c_double_p), ctypes.c_double(lam),
                          step.ctypes.data_as(c_double_p))
    return step


def quat_from_matrix(R):
    R = _vec(R, 9)
    q = np.zeros(4)
    lib().oracle_quat_from_matrix(R.ctypes.data_as(c_double_p), q.ctypes.data_as(c_double_p))
    return q


def quat_to_matrix(q):
    q = _vec(q, 4)
    R = np.zeros(9)
    lib().oracle_quat_to_matrix(q.ctypes.data_as(c_double_p), R.ctypes.data_as(c_double_p))
    return R.reshape(3, 3)


def _avx_call(fn_name, planes_f32, count, n_out, vecs, loss, threads):
    p = np.ascontiguousarray(planes_f32, dtype=np.float32)
    assert p.ndim == 2 and p.shape[0] == count
    fp = ctypes.POINTER(ctypes.c_float)
    arr = (fp * count)()
    for k in range(count):
        arr[k] = p[k].ctypes.data_as(fp)
    out = np.zeros(n_out)
    l = make_loss(loss)
    fn = getattr(avx(), fn_name)
    fn.restype = ctypes.c_int
    args = [ctypes.c_size_t(p.shape[1]), arr] + [v.ctypes.data_as(c_double_p) for v in vecs]
    rc = fn(*args, ctypes.byref(l), ctypes.c_int(threads), out.ctypes.data_as(c_double_p))
    if rc != 0:
        raise RuntimeError("%s failed: %d" % (fn_name, rc))
    return out


def avx_ndt3_accumulate(planes_f32, R2, t2, loss=None, threads=1):
    """AVX2/FMA fp32 restatement of MDM/..._analytic_3dof_simd.cc:85-158 (floor(n/8)*8 items)."""
    return _avx_call("oracle_avx_ndt3_accumulate", planes_f32, 15, 10, [_vec(R2, 4), _vec(t2, 2)], loss, threads)


def avx_reproj_accumulate(planes_f32, R, t, intr, loss=None, threads=1):
    """AVX2/FMA fp32 restatement of REM/..._analytic_simd.cc:55-138; intr = {inv_fx, inv_fy, cx, cy}."""
    return _avx_call("oracle_avx_reproj_accumulate", planes_f32, 5, 28, [_vec(R, 9), _vec(t, 3), _vec(intr, 4)], loss,
                     threads)


def avx_ndt6_accumulate_f64(planes, R, t, loss=None, threads=1):
    """4-lane fp64 AVX2/FMA restatement of SolveDouble's inner loop (MDM/..._analytic_simd_various.cc:42-134): the
    same-precision CPU baseline of the fp64 headline.  planes: [15, n] float64; floor(n/4)*4 items are used."""
    p = np.ascontiguousarray(planes, dtype=np.float64)
    assert p.ndim == 2 and p.shape[0] == 15
    arr = (c_double_p * 15)()
    for k in range(15):
        arr[k] = p[k].ctypes.data_as(c_double_p)
    R = _vec(R, 9)
    t = _vec(t, 3)
    out = np.zeros(28)
    l = make_loss(loss)
    fn = avx().oracle_avx_ndt6_accumulate_f64
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_size_t(p.shape[1]), arr, R.ctypes.data_as(c_double_p), t.ctypes.data_as(c_double_p),
            ctypes.byref(l), ctypes.c_int(threads), out.ctypes.data_as(c_double_p))
    if rc != 0:
        raise RuntimeError("oracle_avx_ndt6_accumulate_f64 failed: %d" % rc)
    return out


def avx_ndt6_accumulate(planes_f32, R, t, loss=None, threads=1):
    """AVX2/FMA fp32 baseline.  planes_f32: [15, n] float32, n multiple of 8 is processed
    (floor(n/8)*8 like the reference)."""
    p = np.ascontiguousarray(planes_f32, dtype=np.float32)
    assert p.ndim == 2 and p.shape[0] == 15
    fp = ctypes.POINTER(ctypes.c_float)
    arr = (fp * 15)()
    for k in range(15):
        arr[k] = p[k].ctypes.data_as(fp)
    R = _vec(R, 9)
    t = _vec(t, 3)
    out = np.zeros(28)
    l = make_loss(loss)
    rc = avx().oracle_avx_ndt6_accumulate(
        ctypes.c_size_t(p.shape[1]), arr, R.ctypes.data_as(c_double_p),
        t.ctypes.data_as(c_double_p), ctypes.byref(l), ctypes.c_int(threads),
        out.ctypes.data_as(c_double_p))
    if rc != 0:
        raise RuntimeError("oracle_avx_ndt6_accumulate failed: %d" % rc)
    return out


def pack_records_f32(records, stride, field_offsets):
    """The AoS → SoA pack loop of the reference's SIMD classes (MDM/..._analytic_simd.cc:19-28) restated: records =
    n x stride bytes, 15 doubles per record at field_offsets → [15, n] float32 planes.  Single-threaded, as there."""
    rec = np.ascontiguousarray(records).view(np.uint8).reshape(-1)
    n = rec.size // stride
    planes = np.empty((15, n), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    arr = (fp * 15)(*[planes[k].ctypes.data_as(fp) for k in range(15)])
    offs = (ctypes.c_size_t * 15)(*[int(o) for o in field_offsets])
    fn = avx().oracle_pack_records_f32
    fn.restype = ctypes.c_int
    rc = fn(ctypes.c_size_t(n), rec.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(stride), offs, arr)
    if rc != 0:
        raise RuntimeError("oracle_pack_records_f32 failed: %d" % rc)
    return planes


# ---- pose-graph linearisation in C (pgo_oracle.c; the twin of oracle_pgo.Graph.linearize)
_pgo = None


def pgo():
    global _pgo
    if _pgo is None:
        build()
        _pgo = ctypes.CDLL(os.path.join(_BUILD, "libnos_pgo_oracle.so"))
        dp, ip, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_ubyte)
        _pgo.oracle_pgo_linearize.argtypes = [ctypes.c_size_t, dp, ctypes.c_size_t, ip, ip, dp, dp, bp, bp, dp, dp, dp, dp, dp]
        _pgo.oracle_pgo_linearize.restype = ctypes.c_int
    return _pgo


def pgo_linearize(poses, ref, qry, meas, sw=None, sw_free=None, fixed=None):
    """→ (hdiag [n, 21] upper triangles row-major, grad [n, 6], hs [m], gs [m], cost)."""
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 7)
    meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 7)
    ref = np.ascontiguousarray(ref, dtype=np.int32)
    qry = np.ascontiguousarray(qry, dtype=np.int32)
    n, m = poses.shape[0], ref.size
    sw = np.ones(m) if sw is None else np.ascontiguousarray(sw, dtype=np.float64)
    dp, ip, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_ubyte)

    def flags(x):
        if x is None:
            return None, None
        a = np.ascontiguousarray(np.asarray(x, dtype=bool).astype(np.uint8))
        return a, a.ctypes.data_as(bp)
    swf, swf_p = flags(sw_free)
    fx, fx_p = flags(fixed)
    hdiag, grad, hs, gs = np.zeros((n, 21)), np.zeros((n, 6)), np.zeros(m), np.zeros(m)
    cost = ctypes.c_double()
    rc = pgo().oracle_pgo_linearize(n, poses.ctypes.data_as(dp), m, ref.ctypes.data_as(ip), qry.ctypes.data_as(ip),
                                    meas.ctypes.data_as(dp), sw.ctypes.data_as(dp), swf_p, fx_p, hdiag.ctypes.data_as(dp),
                                    grad.ctypes.data_as(dp), hs.ctypes.data_as(dp), gs.ctypes.data_as(dp), ctypes.byref(cost))
    if rc != 0:
        raise ValueError("oracle_pgo_linearize: pose index out of range")
    return hdiag, grad, hs, gs, cost.value
