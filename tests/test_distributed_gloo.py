"""world_size-2 gloo test of the sharded path (CPU): each rank assembles its contiguous shard
(the oracle stands in for the GPU kernel), the 28 scalars are all-reduced, and both ranks run the
C++ LM loop; the result must equal the single-process solve over all correspondences."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonlinear_optimizer_for_slam_amd import distributed, solvers, synth
    from oracle import loader
    planes = synth.ndt_planes(n, 500)
    b, e = distributed.shard_range(n, rank, world)
    local = np.ascontiguousarray(planes[:, b:e])
    loss = ("exponential", 1.0, 1.0)
    asm = distributed.ShardedAssembler(lambda R, t: torch.from_numpy(loader.ndt6_accumulate(local, R, t, loss)))
    pose = solvers.Pose()
    rep = distributed.solve_ndt6(asm, solvers.Options(), pose)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), t=pose.t, R=pose.R, it=rep.iterations,
             cost=rep.printed_cost, calls=asm.calls)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_solve_matches_single_process(tmp_path, oracle):
    from nonlinear_optimizer_for_slam_amd import synth
    from tests import helpers
    n = 10_001  # odd: ragged shards
    mp.spawn(_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank sees the same sums → identical trajectories, bit for bit
    assert np.array_equal(r0["t"], r1["t"]) and np.array_equal(r0["R"], r1["R"])
    assert int(r0["it"]) == int(r1["it"])
    planes = synth.ndt_planes(n, 500)
    want = oracle.ndt6_solve(planes, np.zeros(3), np.eye(3), loss=("exponential", 1.0, 1.0), linear_solver=1)
    assert int(r0["it"]) == want["iterations"]
    dt, dq = helpers.pose_delta(r0["R"], r0["t"], want["R"], want["t"])
    assert dt < 1e-9 and dq < 1e-9


def _gpu_worker(rank, world, port, n, out_dir):
    """Two processes share cuda:0: each assembles its shard with the HIP kernels, the 28 scalars are summed over
    gloo, every rank runs the C++ LM loop."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, distributed, solvers, synth
    planes = synth.ndt_planes(n, 500)
    b, e = distributed.shard_range(n, rank, world)
    ctx = Context((0,))
    ds = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, b:e]), "f64")
    loss = ("exponential", 1.0, 1.0)
    asm = distributed.ShardedAssembler(lambda R, t: torch.from_numpy(ds.accumulate6(R, t, loss)))
    pose = solvers.Pose()
    rep = distributed.solve_ndt6(asm, solvers.Options(), pose)
    np.savez(os.path.join(out_dir, "gpu_rank%d.npz" % rank), t=pose.t, R=pose.R, it=rep.iterations)
    ds.close()
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_two_processes_sharing_one_gpu_match_the_single_process_solve(tmp_path):
    from nonlinear_optimizer_for_slam_amd import solvers, synth
    from tests import helpers
    n = 60_001
    mp.spawn(_gpu_worker, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "gpu_rank0.npz")
    r1 = np.load(tmp_path / "gpu_rank1.npz")
    assert np.array_equal(r0["t"], r1["t"]) and np.array_equal(r0["R"], r1["R"])
    single = solvers.MahalanobisDistanceMinimizerHip()
    single.SetLossFunction(("exponential", 1.0, 1.0))
    pose = solvers.Pose()
    assert single.Solve(solvers.Options(), synth.ndt_planes(n, 500), pose)
    assert int(r0["it"]) == single.report.iterations
    dt, dq = helpers.pose_delta(r0["R"], r0["t"], pose.R, pose.t)
    assert dt < 1e-9 and dq < 1e-9
