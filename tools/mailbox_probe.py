import os, sys, uuid
sys.path.insert(0, os.getcwd())
import numpy as np
from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, api, synth
ctx = Context((0,))
name = "/nos_probe_%s" % uuid.uuid4().hex
ctx.comm_init_shm(1, 0, name)
api.shm_unlink(name)
planes = synth.ndt_planes(100_000, 2000)
ds = NdtDataset.from_planes(ctx, planes, "f64")
ds.solve6(np.eye(3), np.zeros(3), ("exponential", 1.0, 1.0), max_iterations=6, gradient_tolerance=0.0, parameter_tolerance=0.0)
ds.close()
