"""Soak of the in-launch mailbox exchange: W processes on GPU 0, solves and accumulates in lock-step, bit-repeatable."""
import os, sys, time, uuid
sys.path.insert(0, os.getcwd())
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, name, rounds, out_dir):
    from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, distributed, synth
    EXP = ("exponential", 1.0, 1.0)
    ctx = Context((0,))
    ctx.comm_init_shm(world, rank, name)
    sets = {}
    for n in (5_000, 150_000, 1_200_000):
        planes = synth.ndt_planes(n, max(10, n // 40), seed=n)
        lo, hi = distributed.shard_range(n, rank, world)
        sets[n] = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, lo:hi]), "f64")
    first = {}
    for r in range(rounds):   # a fixed count on every rank: the exchange is collective
        for n, ds in sets.items():
            R, t, rep = ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=20)
            key = (R.tobytes(), t.tobytes(), rep["iterations"])
            first.setdefault(n, key)
            assert key == first[n], ("mismatch", rank, n, r)
            out = ds.accumulate6(np.eye(3), [0.01, 0.02, 0.03], EXP).tobytes()
            first.setdefault((n, "acc"), out)
            assert out == first[(n, "acc")], ("mismatch acc", rank, n, r)
    with open(os.path.join(out_dir, "soak_rank%d.txt" % rank), "w") as f:
        f.write(repr(first[5_000][2]))
    ctx.close()


if __name__ == "__main__":
    world = min(5, int(sys.argv[1])) if len(sys.argv) > 1 else 3  # a GPU box allows at most 6 processes on its card
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    name = "/nos_soak_%s" % uuid.uuid4().hex
    os.makedirs("gpurun_out", exist_ok=True)
    t0 = time.time()
    try:
        mp.spawn(worker, args=(world, name, rounds, "gpurun_out"), nprocs=world, join=True)
    finally:
        from nonlinear_optimizer_for_slam_amd import api
        api.shm_unlink(name)
    print("mailbox soak ok: %d ranks x %d rounds x 3 sizes in %.1f s" % (world, rounds, time.time() - t0))
