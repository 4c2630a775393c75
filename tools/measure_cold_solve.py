"""Wall time of the drop-in Solve() as the reference's user calls it (std::vector<Correspondence> in, pose out)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import _lib, synth
host = synth.host_lib()
for n, v in ((2_900, 96), (100_000, 5_000), (1_000_000, 50_000), (10_000_000, 200_000)):
    planes = synth.ndt_planes(n, v)
    arr = (_lib.c_double_p * 15)(*[planes[k].ctypes.data_as(_lib.c_double_p) for k in range(15)])
    out = np.zeros(5)
    ok = host.nos_host_ndt_cold_solve_timing(ctypes.c_size_t(n), arr, ctypes.c_int(1), ctypes.c_double(1.0), ctypes.c_double(1.0),
                                             ctypes.c_int(100), ctypes.c_int(_lib.NOS_F64), ctypes.c_int(5 if n >= 1e6 else 20),
                                             out.ctypes.data_as(_lib.c_double_p))
    print("n=%9d: Solve() min %.3f ms mean %.3f ms | Prepare %.3f ms + SolvePrepared %.3f ms | %d LM iterations | ok=%d"
          % (n, out[0], out[1], out[2], out[3], int(out[4]), ok), flush=True)
