"""Per-kernel durations and inter-kernel gaps of the assemble kernel in four launch patterns, from one rocprofv3
kernel trace:   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_loop -- python3 tools/loop_probe.py run
                python tools/loop_probe.py report gpurun_out/prof_loop
Phases (K launches each, in this order): A unfused back-to-back, B fused back-to-back (device result only),
C host LM loop, D device-resident LM loop, E device-resident loop with the stand-alone step kernel."""
import ctypes
import glob
import os
import sys

sys.path.insert(0, os.getcwd())
K = 60
N = int(os.environ.get("NOS_PROBE_POINTS", "10000000"))


def run():
    import numpy as np
    from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, _lib, solvers, synth
    loss_t = ("exponential", 1.0, 1.0)
    ctx = Context((0,))
    planes = synth.ndt_planes(N, max(1, N // 50))
    ds = NdtDataset.from_planes(ctx, planes, os.environ.get("NOS_PROBE_DTYPE", "f64"))
    R0, t0 = np.eye(3), np.zeros(3)
    ds.solve6(R0, t0, loss_t, max_iterations=5, gradient_tolerance=0.0, parameter_tolerance=0.0)  # warm (5 launches)
    ds.time_kernel6(R0, t0, loss_t, repeats=K)  # 2 warm-ups + K unfused + K fused
    host = synth.host_lib()
    loss = solvers.make_loss(loss_t)
    t = np.zeros(3)
    R = np.eye(3).reshape(-1).copy()
    rep = np.zeros(5)
    host.nos_host_ndt6_iterate(ds._h, ctypes.byref(loss), ctypes.c_int(K), t.ctypes.data_as(_lib.c_double_p),
                               R.ctypes.data_as(_lib.c_double_p), rep.ctypes.data_as(_lib.c_double_p))
    ds.solve6(R0, t0, loss_t, max_iterations=K, gradient_tolerance=0.0, parameter_tolerance=0.0)
    os.environ["NOS_LM_FUSED"] = "0"
    ds.solve6(R0, t0, loss_t, max_iterations=K, gradient_tolerance=0.0, parameter_tolerance=0.0)
    ctx.synchronize()


def report(d):
    import csv
    path = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
    rows = [r for r in csv.DictReader(open(path))]
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    asm = [(s, e) for s, e, n in ks if "assemble_kernel" in n]
    print("trace:", path, "assemble launches:", len(asm))
    phases = [("warm device loop", 5), ("time_kernel warm-up", 2), ("A unfused back-to-back", K),
              ("B fused back-to-back", K), ("C host loop", K), ("D device loop", K), ("E device loop, step kernel", K)]
    i = 0
    for name, cnt in phases:
        seg = asm[i:i + cnt]
        i += cnt
        if len(seg) < 2:
            continue
        dur = [e - s for s, e in seg]
        gap = [seg[j + 1][0] - seg[j][1] for j in range(len(seg) - 1)]
        period = [seg[j + 1][0] - seg[j][0] for j in range(len(seg) - 1)]
        med = lambda v: sorted(v)[len(v) // 2]
        print("%-30s n=%3d  duration med %.2f min %.2f us | gap med %.2f us | period med %.2f us"
              % (name, len(seg), med(dur) / 1e3, min(dur) / 1e3, med(gap) / 1e3, med(period) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        report(sys.argv[2])
