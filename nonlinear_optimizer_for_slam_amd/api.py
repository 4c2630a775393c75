"""Thin Python handles over the C ABI (include/nos.h): Context, NdtDataset, ReprojDataset.

numpy arrays go in and out; torch is optional (import it BEFORE creating the first Context if it is going to be
used in the same process: torch bundles its own HIP runtime with the system runtime's soname, and the first one
loaded serves the whole process) and only used for device-resident planes
(`from_device_planes`) and for device-resident results (`*_async`), i.e. as plumbing for
device memory, streams and torch.distributed — all arithmetic happens in libnos_hip.so.
"""
import ctypes
import sys
import weakref

import numpy as np

from . import _lib
from ._lib import (NOS_F32, NOS_F64, NOS_LOSS_EXPONENTIAL, NOS_LOSS_HUBER, NOS_LOSS_NONE,
                   NosLoss, c_double_p, c_void_pp, check, hip_lib)

_DTYPES = {"f64": NOS_F64, "f32": NOS_F32, NOS_F64: NOS_F64, NOS_F32: NOS_F32}


def make_loss(loss):
    """None | ("none",) | ("exponential", c1, c2) | ("huber", threshold) → NosLoss.

    Mirrors the constructors of the reference's loss_function.h (ExponentialLossFunction(c1, c2),
    HuberLossFunction(threshold)); argument validation happens inside the C ABI.
    """
    if loss is None or loss[0] == "none":
        return NosLoss(NOS_LOSS_NONE, 0, 0.0, 0.0)
    if loss[0] == "exponential":
        return NosLoss(NOS_LOSS_EXPONENTIAL, 0, float(loss[1]), float(loss[2]))
    if loss[0] == "huber":
        return NosLoss(NOS_LOSS_HUBER, 0, float(loss[1]), 0.0)
    raise ValueError("unknown loss %r" % (loss,))


def _dvec(x, n):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    if a.size != n:
        raise ValueError("expected %d doubles, got %d" % (n, a.size))
    return a


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def runtime_info():
    """Where the HIP runtime and librccl mapped into this process come from (nos_runtime_info) → dict."""
    import json
    buf = ctypes.create_string_buffer(4096)
    check(hip_lib().nos_runtime_info(buf, 4096), "nos_runtime_info")
    return json.loads(buf.value.decode())


def new_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, all ranks pass it to Context.comm_init)."""
    buf = ctypes.create_string_buffer(128)
    check(hip_lib().nos_comm_get_unique_id(buf), "nos_comm_get_unique_id")
    return buf.raw


def shm_unlink(name):
    check(hip_lib().nos_comm_shm_unlink(name.encode()), "nos_comm_shm_unlink")


class Context:
    """nos_ctx: one HIP stream + workspace per listed device (include/nos.h)."""

    def __init__(self, device_ids=(0,)):
        self._lib = hip_lib()
        ids = (ctypes.c_int * len(device_ids))(*device_ids)
        h = ctypes.c_void_p()
        check(self._lib.nos_ctx_create(ids, len(device_ids), ctypes.byref(h)), "nos_ctx_create")
        self._h = h
        self.device_ids = tuple(device_ids)
        self._children = weakref.WeakSet()  # datasets / maps / scans living on this context

    def _adopt(self, child):
        self._children.add(child)

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr, shard=0):
        check(self._lib.nos_ctx_set_stream(self._h, shard, ctypes.c_void_p(stream_ptr)), "nos_ctx_set_stream")

    def use_torch_stream(self, shard=0):
        import torch
        self.set_stream(torch.cuda.current_stream().cuda_stream, shard)

    def set_launch(self, blocks_per_cu=0, variant=0):
        check(self._lib.nos_ctx_set_launch(self._h, blocks_per_cu, variant), "nos_ctx_set_launch")

    def set_option(self, key, value):
        """Experiment knob of the context (include/nos.h, "experiment knobs"): e.g. set_option("lm_cluster", 0)."""
        check(self._lib.nos_ctx_set_option(self._h, key.encode(), int(value)), "nos_ctx_set_option")

    def get_option(self, key):
        v = ctypes.c_int()
        check(self._lib.nos_ctx_get_option(self._h, key.encode(), ctypes.byref(v)), "nos_ctx_get_option")
        return v.value

    def options(self, **kv):
        """with ctx.options(lm_cluster=0, lm_single=0): ... — set for the block, restored afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self.get_option(k) for k in kv}
            try:
                for k, v in kv.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return scope()

    @property
    def comm_rccl_count(self):
        """Ranks RCCL reports for this context's communicator (ncclCommCount); 0 without an RCCL communicator."""
        v = ctypes.c_int()
        check(self._lib.nos_ctx_comm_rccl_count(self._h, ctypes.byref(v)), "nos_ctx_comm_rccl_count")
        return v.value

    def comm_init(self, n_ranks, rank, unique_id):
        """Collective: join the RCCL communicator identified by the 128-byte unique_id."""
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.nos_ctx_comm_init(self._h, n_ranks, rank, buf), "nos_ctx_comm_init")

    def comm_init_from_torch(self, group=None):
        """Bootstrap the native RCCL communicator through an initialised torch.distributed group:
        rank 0 creates the unique id, broadcasts it, every rank joins."""
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        payload = [new_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(payload, src=0, group=group)
        self.comm_init(world, rank, payload[0])

    def comm_init_shm(self, n_ranks, rank, name, device_memory=False):
        """Collective (ranks of one node): join the mailbox communicator `name` ('/...'); the sums are then exchanged
        inside the launch.  device_memory=False: slots in the POSIX shared-memory segment (nos_ctx_comm_init_shm);
        True: slots in fine-grained device memory of every rank, shared through HIP IPC handles — peers write into each
        other's buffers device to device (nos_ctx_comm_init_shm_device)."""
        if device_memory:
            check(self._lib.nos_ctx_comm_init_shm_device(self._h, n_ranks, rank, name.encode()), "nos_ctx_comm_init_shm_device")
        else:
            check(self._lib.nos_ctx_comm_init_shm(self._h, n_ranks, rank, name.encode()), "nos_ctx_comm_init_shm")

    def comm_init_shm_from_torch(self, group=None, device_memory=False):
        """Bootstrap the mailbox communicator through an initialised torch.distributed group: rank 0 picks a fresh
        name, every rank attaches, rank 0 unlinks the name once all are in (the mapping stays alive)."""
        import torch.distributed as dist
        import uuid
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        payload = ["/nos_%s" % uuid.uuid4().hex if rank == 0 else None]
        dist.broadcast_object_list(payload, src=0, group=group)
        try:
            self.comm_init_shm(world, rank, payload[0], device_memory=device_memory)
        finally:
            dist.barrier(group=group)
            if rank == 0:
                shm_unlink(payload[0])

    def comm_destroy(self):
        check(self._lib.nos_ctx_comm_destroy(self._h), "nos_ctx_comm_destroy")

    @property
    def comm_size(self):
        return int(self._lib.nos_ctx_comm_size(self._h))

    def comm_allreduce(self, values):
        v = np.ascontiguousarray(values, dtype=np.float64).copy()
        check(self._lib.nos_ctx_comm_allreduce(self._h, _dp(v), v.size), "nos_ctx_comm_allreduce")
        return v

    def last_kernel(self, shard=0):
        """Demangled symbol of the hot-path kernel launched last on this context (nos_ctx_last_kernel)."""
        buf = ctypes.create_string_buffer(1024)
        check(self._lib.nos_ctx_last_kernel(self._h, shard, buf, len(buf)), "nos_ctx_last_kernel")
        return buf.value.decode()

    def profile_begin(self, max_launches=4096, sample_every=1):
        check(self._lib.nos_ctx_profile_begin(self._h, max_launches, sample_every), "nos_ctx_profile_begin")

    def profile_end(self):
        """→ (n_launches, mean_ms, min_ms, max_ms) of the assemble kernels launched since profile_begin."""
        n = ctypes.c_int()
        mean = ctypes.c_double()
        lo = ctypes.c_double()
        hi = ctypes.c_double()
        check(self._lib.nos_ctx_profile_end(self._h, ctypes.byref(n), ctypes.byref(mean), ctypes.byref(lo),
                                            ctypes.byref(hi)), "nos_ctx_profile_end")
        return n.value, mean.value, lo.value, hi.value

    def synchronize(self):
        check(self._lib.nos_ctx_synchronize(self._h), "nos_ctx_synchronize")

    def close(self):
        if self._h:
            for child in list(self._children):  # device objects must not outlive their context
                child.close()
            self._lib.nos_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def _lm_call(fn, where, pre_args, post_args, R, t, max_iterations, gradient_tolerance, parameter_tolerance,
             launches_in_flight):
    """Shared body of the device-resident solve methods: returns (R, t, report dict)."""
    from ._lib import NosLmOptions, NosLmReport
    hist = np.full(max(int(max_iterations), 1), np.nan)
    opt = NosLmOptions(int(max_iterations), int(launches_in_flight), float(gradient_tolerance),
                       float(parameter_tolerance), _dp(hist))
    rep = NosLmReport()
    check(fn(*pre_args, _dp(R), _dp(t), *post_args, ctypes.byref(opt), ctypes.byref(rep)), where)
    executed = int(np.count_nonzero(~np.isnan(hist[:max(int(max_iterations), 0)])))
    return R, t, {"iterations": rep.iterations, "ok": bool(rep.ok), "launches": rep.launches, "fallback": bool(rep.fallback),
                  "printed_cost": rep.printed_cost, "last_cost": rep.last_cost, "final_lambda": rep.final_lambda,
                  "cost_history": hist[:executed].copy()}


class _Dataset:
    _n_planes = 0
    _create = _create_dev = _create_rec = None

    def __init__(self, ctx, handle):
        self._ctx = ctx
        self._lib = ctx._lib
        self._h = handle
        ctx._adopt(self)

    @classmethod
    def from_planes(cls, ctx, planes, dtype="f64"):
        """planes: [n_planes, n] float64 host array (plane order of include/nos.h)."""
        planes = np.ascontiguousarray(planes, dtype=np.float64)
        if planes.ndim != 2 or planes.shape[0] != cls._n_planes:
            raise ValueError("planes must be [%d, n]" % cls._n_planes)
        arr = (c_double_p * cls._n_planes)(*[planes[k].ctypes.data_as(c_double_p) for k in range(cls._n_planes)])
        h = ctypes.c_void_p()
        fn = getattr(ctx._lib, cls._create)
        check(fn(ctx.handle, planes.shape[1], arr, _DTYPES[dtype], ctypes.byref(h)), cls._create)
        return cls(ctx, h)

    @classmethod
    def from_device_planes(cls, ctx, planes, dtype="f64"):
        """planes: torch CUDA tensor [n_planes, n], float64 or float32, contiguous."""
        import torch
        if not planes.is_cuda or planes.dim() != 2 or planes.shape[0] != cls._n_planes or not planes.is_contiguous():
            raise ValueError("planes must be a contiguous CUDA tensor [%d, n]" % cls._n_planes)
        src = NOS_F64 if planes.dtype == torch.float64 else NOS_F32
        if planes.dtype not in (torch.float64, torch.float32):
            raise ValueError("planes must be float64 or float32")
        n = planes.shape[1]
        step = planes.element_size() * n
        arr = (ctypes.c_void_p * cls._n_planes)(*[planes.data_ptr() + k * step for k in range(cls._n_planes)])
        h = ctypes.c_void_p()
        torch.cuda.current_stream().synchronize()
        fn = getattr(ctx._lib, cls._create_dev)
        check(fn(ctx.handle, n, arr, src, _DTYPES[dtype], ctypes.byref(h)), cls._create_dev)
        return cls(ctx, h)

    @classmethod
    def from_records(cls, ctx, records, stride_bytes, field_offsets, dtype="f64"):
        """records: host bytes-like / uint8 array of n*stride_bytes; field_offsets: byte offsets
        of the n_planes doubles inside a record (AoS ingestion, include/nos.h)."""
        rec = np.ascontiguousarray(np.frombuffer(records, dtype=np.uint8) if not isinstance(records, np.ndarray)
                                   else records.view(np.uint8).reshape(-1))
        if rec.size % stride_bytes != 0:
            raise ValueError("records size is not a multiple of stride_bytes")
        n = rec.size // stride_bytes
        offs = (ctypes.c_size_t * cls._n_planes)(*[int(o) for o in field_offsets])
        h = ctypes.c_void_p()
        fn = getattr(ctx._lib, cls._create_rec)
        check(fn(ctx.handle, n, rec.ctypes.data_as(ctypes.c_void_p), stride_bytes, offs, _DTYPES[dtype],
                 ctypes.byref(h)), cls._create_rec)
        return cls(ctx, h)

    def __len__(self):
        return int(self._lib.nos_dataset_size(self._h))

    @property
    def stream_bytes(self):
        return int(self._lib.nos_dataset_stream_bytes(self._h))

    def set_simd_class(self, on=True):
        """Semantics of the reference's fp32 ("SIMD") solver classes for this dataset (nos_dataset_set_simd_class): NDT —
        float lambda / previous_cost in the LM loop; reprojection — depth > 0 mask on the weight only.  The tail drop
        (first floor(N/8)*8 correspondences) is the caller's: create the dataset from that prefix."""
        check(self._lib.nos_dataset_set_simd_class(self._h, int(bool(on))), "nos_dataset_set_simd_class")
        return self

    @property
    def dtype(self):
        return "f32" if self._lib.nos_dataset_dtype(self._h) == NOS_F32 else "f64"

    def close(self):
        if self._h:
            self._lib.nos_dataset_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class NdtDataset(_Dataset):
    """Device-resident NDT correspondences (point, mean, sqrt-information)."""
    _n_planes = 15
    _create = "nos_ndt_dataset_create"
    _create_dev = "nos_ndt_dataset_create_from_device"
    _create_rec = "nos_ndt_dataset_create_from_records"

    def drop_last_matches(self, n_drop):
        """Clear the last n_drop non-empty records (slot order) of a matcher-written dataset: the tail drop of the
        reference's solver classes, e.g. n_matches % 4 for the scalar 3-DoF class (nos_dataset_drop_last_matches)."""
        check(self._lib.nos_dataset_drop_last_matches(self._h, int(n_drop)), "nos_dataset_drop_last_matches")

    def accumulate6(self, R, t, loss=None):
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        out = np.zeros(28)
        l = make_loss(loss)
        check(self._lib.nos_ndt6_accumulate(self._h, _dp(R), _dp(t), ctypes.byref(l), _dp(out)), "nos_ndt6_accumulate")
        return out

    def solve6(self, R, t, loss=None, max_iterations=100, gradient_tolerance=1e-6, parameter_tolerance=1e-6,
               launches_in_flight=0):
        """Whole LM loop on the device (nos_ndt6_solve).  Returns (R[9], t[3], report)."""
        R = _dvec(R, 9).copy()
        t = _dvec(t, 3).copy()
        l = make_loss(loss)
        return _lm_call(self._lib.nos_ndt6_solve, "nos_ndt6_solve", (self._h,), (ctypes.byref(l),), R, t,
                        max_iterations, gradient_tolerance, parameter_tolerance, launches_in_flight)

    def solve3(self, R2, t2, loss=None, max_iterations=100, gradient_tolerance=1e-6, parameter_tolerance=1e-6,
               launches_in_flight=0):
        """Planar LM loop on the device (nos_ndt3_solve).  Returns (R2[4], t2[2], report)."""
        R2 = _dvec(R2, 4).copy()
        t2 = _dvec(t2, 2).copy()
        l = make_loss(loss)
        return _lm_call(self._lib.nos_ndt3_solve, "nos_ndt3_solve", (self._h,), (ctypes.byref(l),), R2, t2,
                        max_iterations, gradient_tolerance, parameter_tolerance, launches_in_flight)

    def accumulate6_async(self, R, t, loss, out_tensor):
        """Enqueue on the context stream; result lands in the CUDA float64 tensor out_tensor[28]."""
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        l = make_loss(loss)
        check(self._lib.nos_ndt6_accumulate_async(self._h, _dp(R), _dp(t), ctypes.byref(l),
                                                  ctypes.c_void_p(out_tensor.data_ptr())), "nos_ndt6_accumulate_async")

    def accumulate3(self, R2, t2, loss=None):
        R2 = _dvec(R2, 4)
        t2 = _dvec(t2, 2)
        out = np.zeros(10)
        l = make_loss(loss)
        check(self._lib.nos_ndt3_accumulate(self._h, _dp(R2), _dp(t2), ctypes.byref(l), _dp(out)), "nos_ndt3_accumulate")
        return out

    def accumulate3_async(self, R2, t2, loss, out_tensor):
        R2 = _dvec(R2, 4)
        t2 = _dvec(t2, 2)
        l = make_loss(loss)
        check(self._lib.nos_ndt3_accumulate_async(self._h, _dp(R2), _dp(t2), ctypes.byref(l),
                                                  ctypes.c_void_p(out_tensor.data_ptr())), "nos_ndt3_accumulate_async")

    def time_kernel6(self, R, t, loss=None, repeats=20):
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        l = make_loss(loss)
        k = ctypes.c_double()
        tot = ctypes.c_double()
        check(self._lib.nos_ndt6_time_kernel(self._h, _dp(R), _dp(t), ctypes.byref(l), repeats, ctypes.byref(k),
                                             ctypes.byref(tot)), "nos_ndt6_time_kernel")
        return k.value, tot.value

    def time_kernel3(self, R2, t2, loss=None, repeats=20):
        R2 = _dvec(R2, 4)
        t2 = _dvec(t2, 2)
        l = make_loss(loss)
        k = ctypes.c_double()
        tot = ctypes.c_double()
        check(self._lib.nos_ndt3_time_kernel(self._h, _dp(R2), _dp(t2), ctypes.byref(l), repeats, ctypes.byref(k),
                                             ctypes.byref(tot)), "nos_ndt3_time_kernel")
        return k.value, tot.value


class NdtIndexedDataset(NdtDataset):
    """Voxel-indexed NDT correspondences: points [3,n] + voxel ids [K,n] (-1 = none) + a voxel table
    (means [V,3], sqrt-informations [V,3,3]).  Same accumulate6 / accumulate3 interface and the same sums as
    the flat NdtDataset, at 24 B + 4 B·K per point instead of 120 B per correspondence (include/nos.h)."""
    _n_planes = 3  # what nos_dataset_download returns for this kind: the (voxel-sorted) point planes

    @classmethod
    def from_arrays(cls, ctx, points, index, means, sqrt_infos, dtype="f64", sort_by_voxel=True):
        points = np.ascontiguousarray(points, dtype=np.float64)
        index = np.ascontiguousarray(np.atleast_2d(index), dtype=np.int32)
        means = np.ascontiguousarray(means, dtype=np.float64).reshape(-1, 3)
        S = np.ascontiguousarray(sqrt_infos, dtype=np.float64).reshape(-1, 9)
        if points.ndim != 2 or points.shape[0] != 3 or index.shape[1] != points.shape[1]:
            raise ValueError("points must be [3, n] and index [K, n]")
        K, n = index.shape
        pp = (c_double_p * 3)(*[points[k].ctypes.data_as(c_double_p) for k in range(3)])
        ip_t = ctypes.POINTER(ctypes.c_int32)
        ip = (ip_t * K)(*[index[k].ctypes.data_as(ip_t) for k in range(K)])
        h = ctypes.c_void_p()
        check(ctx._lib.nos_ndt_indexed_dataset_create(ctx.handle, n, pp, K, ip, means.shape[0], _dp(means), _dp(S),
                                                      _DTYPES[dtype], int(bool(sort_by_voxel)), ctypes.byref(h)),
              "nos_ndt_indexed_dataset_create")
        return cls(ctx, h)


class ReprojDataset(_Dataset):
    """Device-resident 3D↔2D correspondences (X, Y, Z, u, v)."""
    _n_planes = 5
    _create = "nos_reproj_dataset_create"
    _create_dev = "nos_reproj_dataset_create_from_device"
    _create_rec = "nos_reproj_dataset_create_from_records"

    def accumulate(self, R, t, intr, loss=None, min_depth=0.03):
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        intr = _dvec(intr, 4)
        out = np.zeros(28)
        l = make_loss(loss)
        check(self._lib.nos_reproj_accumulate(self._h, _dp(R), _dp(t), _dp(intr), ctypes.byref(l),
                                              ctypes.c_double(min_depth), _dp(out)), "nos_reproj_accumulate")
        return out

    def solve(self, R, t, intr, loss=None, min_depth=0.03, max_iterations=100, gradient_tolerance=1e-6,
              parameter_tolerance=1e-6, launches_in_flight=0):
        """Whole LM loop on the device (nos_reproj_solve).  Returns (R[9], t[3], report)."""
        R = _dvec(R, 9).copy()
        t = _dvec(t, 3).copy()
        intr = _dvec(intr, 4)
        l = make_loss(loss)
        return _lm_call(self._lib.nos_reproj_solve, "nos_reproj_solve", (self._h,),
                        (_dp(intr), ctypes.byref(l), ctypes.c_double(min_depth)), R, t,
                        max_iterations, gradient_tolerance, parameter_tolerance, launches_in_flight)

    def accumulate_async(self, R, t, intr, loss, out_tensor, min_depth=0.03):
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        intr = _dvec(intr, 4)
        l = make_loss(loss)
        check(self._lib.nos_reproj_accumulate_async(self._h, _dp(R), _dp(t), _dp(intr), ctypes.byref(l),
                                                    ctypes.c_double(min_depth),
                                                    ctypes.c_void_p(out_tensor.data_ptr())),
              "nos_reproj_accumulate_async")

    def time_kernel(self, R, t, intr, loss=None, min_depth=0.03, repeats=20):
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        intr = _dvec(intr, 4)
        l = make_loss(loss)
        k = ctypes.c_double()
        tot = ctypes.c_double()
        check(self._lib.nos_reproj_time_kernel(self._h, _dp(R), _dp(t), _dp(intr), ctypes.byref(l),
                                               ctypes.c_double(min_depth), repeats, ctypes.byref(k),
                                               ctypes.byref(tot)), "nos_reproj_time_kernel")
        return k.value, tot.value


class NdtMap:
    """Device-resident NDT voxel map for matching (nos_ndt_map): means [V,3], sqrt-information
    [V,3,3] row-major, optional validity mask; search_radius_sq is the FLANN `radius` of the
    reference's MatchPointCloud (squared distance, 1.0 there)."""

    def __init__(self, ctx, means, sqrt_infos, valid=None, search_radius_sq=1.0):
        self._ctx = ctx
        self._lib = ctx._lib
        means = np.ascontiguousarray(means, dtype=np.float64).reshape(-1, 3)
        S = np.ascontiguousarray(sqrt_infos, dtype=np.float64).reshape(-1, 9)
        if means.shape[0] != S.shape[0]:
            raise ValueError("means and sqrt_infos disagree on the voxel count")
        vbuf = None
        if valid is not None:
            vbuf = np.ascontiguousarray(valid, dtype=np.uint8).tobytes()
        h = ctypes.c_void_p()
        check(self._lib.nos_ndt_map_create(ctx.handle, means.shape[0], _dp(means), _dp(S), vbuf,
                                           ctypes.c_double(search_radius_sq), ctypes.byref(h)), "nos_ndt_map_create")
        self._h = h
        ctx._adopt(self)

    @classmethod
    def build(cls, ctx, points, voxel_resolution=1.0, search_radius_sq=1.0, proper_sqrt_information=True,
              reference_exact=False, return_stats=True):
        """Construct the map from raw points [n,3] on the GPU (nos_ndt_map_build, the reference's
        UpdateNdtMap).  → (NdtMap, stats dict with means, sqrt_infos, valid, counts, cells).

        proper_sqrt_information=True (default) stores D^-1/2 V^T, the true square root of the inverse
        covariance; False reproduces the harness formula D^-1/2 V, which is only meaningful when V happens
        to be symmetric (include/nos.h).

        reference_exact=True (NOS_MAP_REFERENCE_EXACT): the harness formula with the reference binary's rounding —
        sequential per-voxel accumulation, Eigen's SelfAdjointEigenSolver restated, the reference build's fused
        multiply-adds; voxels in first-seen order; the stats dict also carries eigvals / eigvecs.

        return_stats=False: out_stats = NULL — the voxel statistics never leave the device (the matcher's tables are
        built there); → (NdtMap, None)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        h = ctypes.c_void_p()
        hs = ctypes.c_void_p()
        lib = ctx._lib
        flags = 2 if reference_exact else int(bool(proper_sqrt_information))
        check(lib.nos_ndt_map_build(ctx.handle, pts.shape[0], _dp(pts), ctypes.c_double(voxel_resolution),
                                    ctypes.c_double(search_radius_sq), flags,
                                    ctypes.byref(h), ctypes.byref(hs) if return_stats else None),
              "nos_ndt_map_build")
        if not return_stats:
            self = cls.__new__(cls)
            self._ctx = ctx
            self._lib = lib
            self._h = h
            ctx._adopt(self)
            return self, None
        V = int(lib.nos_map_stats_size(hs))
        means = np.zeros((V, 3))
        S = np.zeros((V, 9))
        valid = np.zeros(V, dtype=np.uint8)
        counts = np.zeros(V, dtype=np.uint32)
        cells = np.zeros((V, 3), dtype=np.int64)
        check(lib.nos_map_stats_get(hs, _dp(means), _dp(S), valid.ctypes.data_as(ctypes.c_char_p),
                                    counts.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                                    cells.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))), "nos_map_stats_get")
        stats = {"means": means, "sqrt_infos": S.reshape(V, 3, 3), "valid": valid.astype(bool),
                 "counts": counts, "cells": cells}
        if reference_exact:
            evals = np.zeros((V, 3))
            evecs = np.zeros((V, 9))
            check(lib.nos_map_stats_get_eigen(hs, _dp(evals), _dp(evecs)), "nos_map_stats_get_eigen")
            stats["eigvals"] = evals
            stats["eigvecs"] = evecs.reshape(V, 3, 3)
        lib.nos_map_stats_destroy(hs)
        self = cls.__new__(cls)
        self._ctx = ctx
        self._lib = lib
        self._h = h
        ctx._adopt(self)
        return self, stats

    def __len__(self):
        return int(self._lib.nos_ndt_map_size(self._h))

    def match(self, scan, R, t, max_neighbors=2, dtype="f64"):
        """→ (NdtDataset with 2 slots per scan point, number of real matches)."""
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        h = ctypes.c_void_p()
        n = ctypes.c_size_t()
        check(self._lib.nos_ndt_match(self._h, scan._h, _dp(R), _dp(t), max_neighbors, _DTYPES[dtype],
                                      ctypes.byref(h), ctypes.byref(n)), "nos_ndt_match")
        return NdtDataset(self._ctx, h), int(n.value)

    def match_indexed(self, scan, R, t, max_neighbors=2, dtype="f64", sort_by_voxel=True):
        """→ (NdtIndexedDataset with max_neighbors voxel slots per scan point, number of real matches)."""
        R = _dvec(R, 9)
        t = _dvec(t, 3)
        h = ctypes.c_void_p()
        n = ctypes.c_size_t()
        check(self._lib.nos_ndt_match_indexed(self._h, scan._h, _dp(R), _dp(t), max_neighbors, _DTYPES[dtype],
                                              int(bool(sort_by_voxel)), ctypes.byref(h), ctypes.byref(n)),
              "nos_ndt_match_indexed")
        return NdtIndexedDataset(self._ctx, h), int(n.value)

    def close(self):
        if self._h:
            self._lib.nos_ndt_map_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


class Scan:
    """Device-resident scan points in the sensor's local frame (nos_scan): points [n,3]."""

    def __init__(self, ctx, points, sort_cell=None):
        """sort_cell: if set, reorder the points by grid cell of that edge (nos_scan_sort_by_cell) — faster matching
        for scans whose points do not arrive in spatial order; `order` then maps positions to input indices."""
        self._ctx = ctx
        self._lib = ctx._lib
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        h = ctypes.c_void_p()
        check(self._lib.nos_scan_create(ctx.handle, pts.shape[0], _dp(pts), ctypes.byref(h)), "nos_scan_create")
        self._h = h
        ctx._adopt(self)
        if sort_cell is not None:
            self.sort_by_cell(sort_cell)

    def sort_by_cell(self, cell_edge):
        check(self._lib.nos_scan_sort_by_cell(self._h, ctypes.c_double(cell_edge)), "nos_scan_sort_by_cell")

    @property
    def order(self):
        """order[j] = index (in the array given to the constructor) of the point stored at position j."""
        out = np.zeros(len(self), dtype=np.uint32)
        check(self._lib.nos_scan_order(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))), "nos_scan_order")
        return out

    def __len__(self):
        return int(self._lib.nos_scan_size(self._h))

    def close(self):
        if self._h:
            self._lib.nos_scan_destroy(self._h)
            self._h = None

    def __del__(self):
        # at interpreter shutdown the HIP runtime may already be gone: leave the handle to the OS
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass


def download(dataset):
    """Dataset → host planes [n_planes, n] (diagnostics / tests)."""
    n = len(dataset)
    planes = np.zeros((dataset._n_planes, n))
    arr = (c_double_p * dataset._n_planes)(*[planes[k].ctypes.data_as(c_double_p) for k in range(dataset._n_planes)])
    check(dataset._lib.nos_dataset_download(dataset._h, arr), "nos_dataset_download")
    return planes
