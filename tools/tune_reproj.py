"""Launch-geometry sweep for the reprojection kernel at configs[2] size (2 M correspondences, Huber)."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nonlinear_optimizer_for_slam_amd import Context, ReprojDataset, _lib, solvers, synth  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
planes = synth.reproj_planes(n)
loss = ("huber", synth.REPROJ_HUBER_THRESHOLD)
host = synth.host_lib()
l = solvers.make_loss(loss)
intr = np.array(synth.REPROJ_INTR4)
for dtype in ("f64", "f32"):
    ctx = Context((0,))
    ds = ReprojDataset.from_planes(ctx, planes, dtype)
    for variant in range(7):
        for bpc in (0, 1, 2, 3, 4):
            ctx.set_launch(bpc, variant)
            try:
                k, tot = ds.time_kernel(np.eye(3), np.zeros(3), synth.REPROJ_INTR4, loss, repeats=200)
            except Exception as exc:
                print("ERR", dtype, variant, bpc, exc); continue
            pt, pR, rep = np.zeros(3), np.eye(3).reshape(-1).copy(), np.zeros(5)
            def run(kk):
                host.nos_host_reproj_iterate(ds._h, intr.ctypes.data_as(_lib.c_double_p), ctypes.byref(l), ctypes.c_double(0.03), ctypes.c_int(kk), pt.ctypes.data_as(_lib.c_double_p), pR.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))
            run(20); t0 = time.perf_counter(); run(300); it = (time.perf_counter() - t0) / 300
            print("%s variant=%d bpc=%d kernel %.2f us fused %.2f us LM iteration %.2f us" % (dtype, variant, bpc, 1e3 * k, 1e3 * tot, 1e6 * it), flush=True)
    ds.close(); ctx.close()
