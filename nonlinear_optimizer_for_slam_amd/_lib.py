"""ctypes binding of libnos_hip.so (the C ABI in include/nos.h).

The library is the product; this module only loads it.  If the shared object is missing or a
HIP device is not usable, every call raises — there is no CPU fallback anywhere in the package.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# NOS_HIP_LIB: an alternative build of the same library (e.g. the -DNOS_LM_TIMING probe build of tools/); default in-tree
LIB_HIP = os.environ.get("NOS_HIP_LIB") or os.path.join(CSRC, "libnos_hip.so")
LIB_HOST = os.path.join(CSRC, "libnos_host.so")

c_double_p = ctypes.POINTER(ctypes.c_double)
c_void_pp = ctypes.POINTER(ctypes.c_void_p)

NOS_OK = 0
NOS_F64 = 0
NOS_F32 = 1
NOS_LOSS_NONE = 0
NOS_LOSS_EXPONENTIAL = 1
NOS_LOSS_HUBER = 2

# every symbol include/nos.h declares; tests check the library exports all of them
C_ABI_SYMBOLS = (
    "nos_ctx_create", "nos_ctx_destroy", "nos_ctx_num_devices", "nos_ctx_set_stream",
    "nos_ctx_synchronize", "nos_comm_get_unique_id", "nos_ctx_comm_init", "nos_ctx_comm_size", "nos_ctx_comm_init_shm", "nos_ctx_comm_init_shm_device", "nos_comm_shm_unlink", "nos_ctx_comm_destroy",
    "nos_ctx_comm_allreduce", "nos_ndt_dataset_create", "nos_reproj_dataset_create",
    "nos_ndt_dataset_create_from_device", "nos_reproj_dataset_create_from_device",
    "nos_ndt_dataset_create_from_records", "nos_reproj_dataset_create_from_records",
    "nos_dataset_download", "nos_ndt_map_create", "nos_ndt_map_destroy", "nos_ndt_map_size", "nos_scan_create",
    "nos_scan_destroy", "nos_scan_size", "nos_scan_sort_by_cell", "nos_scan_order", "nos_ndt_match", "nos_ndt_indexed_dataset_create", "nos_ndt_match_indexed", "nos_ndt_map_build", "nos_map_stats_size",
    "nos_map_stats_get", "nos_map_stats_get_eigen", "nos_map_stats_destroy", "nos_dataset_drop_last_matches", "nos_pgo_create", "nos_pgo_destroy", "nos_pgo_num_unknowns",
    "nos_pgo_linearize", "nos_pgo_solve", "nos_pgo_retract", "nos_pgo_get_state", "nos_pgo_get_vector",
    "nos_pgo_matvec", "nos_pgo_time_sweep", "nos_pgo_layout_info", "nos_debug_lm_step", "nos_dataset_destroy", "nos_dataset_size", "nos_dataset_dtype", "nos_dataset_stream_bytes",
    "nos_dataset_set_simd_class",
    "nos_ndt6_accumulate", "nos_ndt3_accumulate", "nos_reproj_accumulate",
    "nos_ndt6_accumulate_async", "nos_ndt3_accumulate_async", "nos_reproj_accumulate_async",
    "nos_ndt6_solve", "nos_ndt3_solve", "nos_reproj_solve",
    "nos_ctx_set_launch", "nos_ctx_set_option", "nos_ctx_get_option", "nos_runtime_info", "nos_ctx_comm_rccl_count",
    "nos_ctx_last_kernel", "nos_ctx_profile_begin", "nos_ctx_profile_end", "nos_ndt6_time_kernel", "nos_reproj_time_kernel",
    "nos_ndt3_time_kernel", "nos_status_string", "nos_last_error", "nos_version",
)


class NosLoss(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("a", ctypes.c_double), ("b", ctypes.c_double)]


class NosLmOptions(ctypes.Structure):
    _fields_ = [("max_iterations", ctypes.c_int32), ("launches_in_flight", ctypes.c_int32),
                ("gradient_tolerance", ctypes.c_double), ("parameter_tolerance", ctypes.c_double),
                ("cost_history", ctypes.POINTER(ctypes.c_double))]


class NosLmReport(ctypes.Structure):
    _fields_ = [("iterations", ctypes.c_int32), ("ok", ctypes.c_int32), ("launches", ctypes.c_int32),
                ("fallback", ctypes.c_int32), ("printed_cost", ctypes.c_double), ("last_cost", ctypes.c_double),
                ("final_lambda", ctypes.c_double)]


class NosError(RuntimeError):
    def __init__(self, status, where, detail):
        super().__init__("%s failed: status %d (%s)" % (where, status, detail))
        self.status = status


_hip = None


def hip_lib():
    """Load libnos_hip.so; raises if it has not been built (no fallback)."""
    global _hip
    if _hip is None:
        if not os.path.exists(LIB_HIP):
            raise ImportError(
                "libnos_hip.so is missing at %s — build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950); this package has no CPU fallback." % LIB_HIP)
        lib = ctypes.CDLL(LIB_HIP)
        _declare(lib)
        _hip = lib
    return _hip


def _declare(lib):
    vp = ctypes.c_void_p
    sz = ctypes.c_size_t
    i = ctypes.c_int
    dp = c_double_p
    lp = ctypes.POINTER(NosLoss)
    lib.nos_ctx_create.argtypes = [ctypes.POINTER(ctypes.c_int), i, c_void_pp]
    lib.nos_ctx_destroy.argtypes = [vp]
    lib.nos_ctx_num_devices.argtypes = [vp]
    lib.nos_ctx_set_stream.argtypes = [vp, i, vp]
    lib.nos_ctx_synchronize.argtypes = [vp]
    lib.nos_ctx_set_launch.argtypes = [vp, i, i]
    lib.nos_ctx_set_option.argtypes = [vp, ctypes.c_char_p, i]
    lib.nos_ctx_get_option.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(i)]
    lib.nos_runtime_info.argtypes = [ctypes.c_char_p, sz]
    lib.nos_ctx_comm_rccl_count.argtypes = [vp, ctypes.POINTER(i)]
    lib.nos_comm_get_unique_id.argtypes = [ctypes.c_char_p]
    lib.nos_ctx_comm_init.argtypes = [vp, i, i, ctypes.c_char_p]
    lib.nos_ctx_comm_init_shm.argtypes = [vp, i, i, ctypes.c_char_p]
    if hasattr(lib, "nos_ctx_comm_init_shm_device"):
        lib.nos_ctx_comm_init_shm_device.argtypes = [vp, i, i, ctypes.c_char_p]
    lib.nos_comm_shm_unlink.argtypes = [ctypes.c_char_p]
    lib.nos_ctx_comm_destroy.argtypes = [vp]
    lib.nos_ctx_comm_size.argtypes = [vp]
    lib.nos_ctx_comm_allreduce.argtypes = [vp, dp, i]
    if hasattr(lib, "nos_ctx_last_kernel"):  # absent from older builds loaded through NOS_HIP_LIB (tools/ab_resident.sh)
        lib.nos_ctx_last_kernel.argtypes = [vp, i, ctypes.c_char_p, sz]
    lib.nos_ctx_profile_begin.argtypes = [vp, i, i]
    lib.nos_ctx_profile_end.argtypes = [vp, ctypes.POINTER(i), dp, dp, dp]
    lib.nos_ndt_dataset_create.argtypes = [vp, sz, ctypes.POINTER(dp), i, c_void_pp]
    lib.nos_reproj_dataset_create.argtypes = [vp, sz, ctypes.POINTER(dp), i, c_void_pp]
    lib.nos_ndt_dataset_create_from_device.argtypes = [vp, sz, c_void_pp, i, i, c_void_pp]
    lib.nos_reproj_dataset_create_from_device.argtypes = [vp, sz, c_void_pp, i, i, c_void_pp]
    lib.nos_ndt_dataset_create_from_records.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), i, c_void_pp]
    lib.nos_reproj_dataset_create_from_records.argtypes = [vp, sz, vp, sz, ctypes.POINTER(sz), i, c_void_pp]
    lib.nos_dataset_download.argtypes = [vp, ctypes.POINTER(dp)]
    lib.nos_ndt_map_create.argtypes = [vp, sz, dp, dp, ctypes.c_char_p, ctypes.c_double, c_void_pp]
    lib.nos_ndt_map_destroy.argtypes = [vp]
    lib.nos_ndt_map_size.argtypes = [vp]
    lib.nos_ndt_map_size.restype = sz
    lib.nos_scan_create.argtypes = [vp, sz, dp, c_void_pp]
    lib.nos_scan_destroy.argtypes = [vp]
    lib.nos_scan_size.argtypes = [vp]
    lib.nos_scan_size.restype = sz
    lib.nos_scan_sort_by_cell.argtypes = [vp, ctypes.c_double]
    lib.nos_scan_order.argtypes = [vp, ctypes.POINTER(ctypes.c_uint32)]
    lib.nos_ndt_match.argtypes = [vp, vp, dp, dp, i, i, c_void_pp, ctypes.POINTER(sz)]
    lib.nos_ndt_indexed_dataset_create.argtypes = [vp, sz, ctypes.POINTER(dp), i, ctypes.POINTER(ctypes.POINTER(ctypes.c_int32)),
                                                   sz, dp, dp, i, i, c_void_pp]
    lib.nos_ndt_match_indexed.argtypes = [vp, vp, dp, dp, i, i, i, c_void_pp, ctypes.POINTER(sz)]
    lib.nos_ndt_map_build.argtypes = [vp, sz, dp, ctypes.c_double, ctypes.c_double, i, c_void_pp, c_void_pp]
    lib.nos_map_stats_size.argtypes = [vp]
    lib.nos_map_stats_size.restype = sz
    lib.nos_map_stats_get.argtypes = [vp, dp, dp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32),
                                      ctypes.POINTER(ctypes.c_int64)]
    if hasattr(lib, "nos_map_stats_get_eigen"):
        lib.nos_map_stats_get_eigen.argtypes = [vp, dp, dp]
    lib.nos_map_stats_destroy.argtypes = [vp]
    if hasattr(lib, "nos_dataset_drop_last_matches"):
        lib.nos_dataset_drop_last_matches.argtypes = [vp, sz]
    ip = ctypes.POINTER(ctypes.c_int32)
    lib.nos_pgo_create.argtypes = [vp, sz, dp, sz, ip, ip, dp, dp, ctypes.c_char_p, ctypes.c_char_p, c_void_pp]
    lib.nos_pgo_destroy.argtypes = [vp]
    lib.nos_pgo_num_unknowns.argtypes = [vp]
    lib.nos_pgo_num_unknowns.restype = sz
    lib.nos_pgo_linearize.argtypes = [vp, dp, dp]
    lib.nos_pgo_solve.argtypes = [vp, ctypes.c_double, i, ctypes.c_double, ctypes.POINTER(i), dp, dp]
    lib.nos_pgo_retract.argtypes = [vp]
    lib.nos_pgo_get_state.argtypes = [vp, dp, dp]
    lib.nos_pgo_get_vector.argtypes = [vp, i, dp]
    lib.nos_pgo_matvec.argtypes = [vp, ctypes.c_double, dp, dp]
    if hasattr(lib, "nos_pgo_layout_info"):
        lib.nos_pgo_layout_info.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong)]
    if hasattr(lib, "nos_debug_lm_step"):
        lib.nos_debug_lm_step.argtypes = [vp, i, dp, dp, dp]
    if hasattr(lib, "nos_pgo_time_sweep"):  # absent from older builds loaded through NOS_HIP_LIB
        lib.nos_pgo_time_sweep.argtypes = [vp, i, ctypes.c_double, i, dp]
    lib.nos_dataset_destroy.argtypes = [vp]
    lib.nos_dataset_set_simd_class.argtypes = [vp, ctypes.c_int]
    lib.nos_dataset_set_simd_class.restype = ctypes.c_int
    lib.nos_dataset_size.argtypes = [vp]
    lib.nos_dataset_size.restype = sz
    lib.nos_dataset_dtype.argtypes = [vp]
    lib.nos_dataset_stream_bytes.argtypes = [vp]
    lib.nos_dataset_stream_bytes.restype = sz
    lib.nos_ndt6_accumulate.argtypes = [vp, dp, dp, lp, dp]
    lib.nos_ndt3_accumulate.argtypes = [vp, dp, dp, lp, dp]
    lib.nos_reproj_accumulate.argtypes = [vp, dp, dp, dp, lp, ctypes.c_double, dp]
    lib.nos_ndt6_accumulate_async.argtypes = [vp, dp, dp, lp, vp]
    lib.nos_ndt3_accumulate_async.argtypes = [vp, dp, dp, lp, vp]
    lib.nos_reproj_accumulate_async.argtypes = [vp, dp, dp, dp, lp, ctypes.c_double, vp]
    lmo, lmr = ctypes.POINTER(NosLmOptions), ctypes.POINTER(NosLmReport)
    lib.nos_ndt6_solve.argtypes = [vp, dp, dp, lp, lmo, lmr]
    lib.nos_ndt3_solve.argtypes = [vp, dp, dp, lp, lmo, lmr]
    lib.nos_reproj_solve.argtypes = [vp, dp, dp, dp, lp, ctypes.c_double, lmo, lmr]
    lib.nos_ndt6_time_kernel.argtypes = [vp, dp, dp, lp, i, dp, dp]
    lib.nos_ndt3_time_kernel.argtypes = [vp, dp, dp, lp, i, dp, dp]
    lib.nos_reproj_time_kernel.argtypes = [vp, dp, dp, dp, lp, ctypes.c_double, i, dp, dp]
    lib.nos_status_string.argtypes = [i]
    lib.nos_status_string.restype = ctypes.c_char_p
    lib.nos_last_error.restype = ctypes.c_char_p
    lib.nos_version.restype = ctypes.c_char_p


def check(status, where):
    if status != NOS_OK:
        lib = hip_lib()
        detail = "%s: %s" % (lib.nos_status_string(status).decode(), lib.nos_last_error().decode())
        raise NosError(status, where, detail)
