"""Soak of the in-launch mailbox exchange: W processes on GPU 0, solves and accumulates in lock-step, bit-repeatable.

usage: python tools/soak_mailbox.py [ranks = 3] [rounds = 2000] [form = host | device | one_launch]
  host        slots in host shared memory, one launch per iteration (round 1's protocol)
  device      slots in fine-grained device memory (HIP IPC), one launch per iteration
  one_launch  the same buffers, the exchange as the third stage of the one-launch loop's all-reduce (round 4); every
              rank's grid is capped at 256 / ranks workgroups so that all of them are resident together; reports how many
              solves fell back to one launch per iteration"""
import os, sys, time, uuid
sys.path.insert(0, os.getcwd())
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, name, rounds, out_dir, form="host"):
    from nonlinear_optimizer_for_slam_amd import Context, NdtDataset, distributed, synth
    EXP = ("exponential", 1.0, 1.0)
    ctx = Context((0,))
    if form == "host":
        ctx.comm_init_shm(world, rank, name)
    else:
        ctx.comm_init_shm(world, rank, name, device_memory=True)
        ctx.set_option("lm_cluster_max_blocks", max(1, 256 // world))
        if form == "device":
            ctx.set_option("lm_cluster", 0)
    fallbacks = 0
    sets = {}
    for n in (5_000, 150_000, 1_200_000):
        planes = synth.ndt_planes(n, max(10, n // 40), seed=n)
        lo, hi = distributed.shard_range(n, rank, world)
        sets[n] = NdtDataset.from_planes(ctx, np.ascontiguousarray(planes[:, lo:hi]), "f64")
    first = {}
    for r in range(rounds):   # a fixed count on every rank: the exchange is collective
        if rank == 0 and r % 2000 == 0 and r:
            print("  round %d, fallbacks on rank 0 so far: %d" % (r, fallbacks), flush=True)
        for n, ds in sets.items():
            R, t, rep = ds.solve6(np.eye(3), np.zeros(3), EXP, max_iterations=20)
            fallbacks += int(rep.get("fallback", 0))
            key = (R.tobytes(), t.tobytes(), rep["iterations"])
            first.setdefault(n, key)
            assert key == first[n], ("mismatch", rank, n, r)
            out = ds.accumulate6(np.eye(3), [0.01, 0.02, 0.03], EXP).tobytes()
            first.setdefault((n, "acc"), out)
            assert out == first[(n, "acc")], ("mismatch acc", rank, n, r)
    with open(os.path.join(out_dir, "soak_rank%d.txt" % rank), "w") as f:
        f.write("%r fallbacks %d" % (first[5_000][2], fallbacks))
    ctx.close()


if __name__ == "__main__":
    world = min(5, int(sys.argv[1])) if len(sys.argv) > 1 else 3  # a GPU box allows at most 6 processes on its card
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    form = sys.argv[3] if len(sys.argv) > 3 else "host"
    name = "/nos_soak_%s" % uuid.uuid4().hex
    os.makedirs("gpurun_out", exist_ok=True)
    t0 = time.time()
    try:
        mp.spawn(worker, args=(world, name, rounds, "gpurun_out", form), nprocs=world, join=True)
    finally:
        from nonlinear_optimizer_for_slam_amd import api
        api.shm_unlink(name)
    fb = [open(os.path.join("gpurun_out", "soak_rank%d.txt" % r)).read().split("fallbacks")[-1].strip() for r in range(world)]
    print("mailbox soak ok (%s): %d ranks x %d rounds x 3 sizes in %.1f s; solves that fell back to one launch per iteration, per rank: %s"
          % (form, world, rounds, time.time() - t0, fb))
