"""Is this box one of the slow ones, and what do its clocks / power do under the headline load?  Runs the 10 M fp64 6-DoF
LM loop (ndt6) and the 3-DoF one (ndt3: same bytes, a third of the arithmetic) for a few seconds each while rocm-smi is
sampled in the background."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, synth
ctx = Context((0,))
planes = synth.ndt_planes(10_000_000, 200_000)
ds = NdtDataset.from_planes(ctx, planes, "f64")
EXP = ("exponential", 1.0, 1.0)
samples, stop = [], False


def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True,
                                 text=True, timeout=5).stdout.strip().splitlines()
            samples.append((time.perf_counter(), out[-1] if out else ""))
            if len(samples) == 1 and len(out) > 1:
                print("rocm-smi header:", out[0], flush=True)
        except Exception as exc:  # noqa: BLE001
            samples.append((time.perf_counter(), "rocm-smi failed: %r" % (exc,)))
            return
        time.sleep(0.3)


th = threading.Thread(target=sampler, daemon=True)
th.start()
for name, fn in (("idle", None),
                 ("ndt6", lambda: ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0)),
                 ("ndt3", lambda: ds.solve3(np.eye(2), np.zeros(2), EXP, max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0)),
                 ("ndt6", lambda: ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=200, gradient_tolerance=0.0, parameter_tolerance=0.0))):
    t_begin = time.perf_counter()
    if fn is None:
        time.sleep(1.5)
        print("[%s] t=%.1f-%.1f" % (name, t_begin, time.perf_counter()), flush=True)
        continue
    per = []
    while time.perf_counter() - t_begin < 4.0:
        t0 = time.perf_counter(); fn(); per.append(1e3 * (time.perf_counter() - t0) / 200)
    print("[%s] t=%.1f-%.1f  ms/iteration: first %.5f  min %.5f  median %.5f  last %.5f" % (
        name, t_begin, time.perf_counter(), per[0], min(per), float(np.median(per)), per[-1]), flush=True)
stop = True
th.join(timeout=6)
for t, line in samples:
    print("  t=%.1f  %s" % (t, line))
