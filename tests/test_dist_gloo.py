"""N > 1 path on CPU: world_size 2 over gloo. Covers bench.py's sharding plumbing (trust4_amd/dist.py):
rank-seeded shards are disjoint streams, the timed interval is max-reduced, totals are sum-reduced, and
each rank's shard is processed independently (checked here with the CPU oracle standing in as the checker
of the per-rank results; the GPU engine itself is exercised by the -m gpu suite)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import t4libs
    import trust4_amd.dist as t4dist
    dist = t4dist.init("gloo")
    assert dist is not None and dist.get_world_size() == world
    arr = t4libs.Synth(200, t4dist.shard_seed(1, rank)).next_reads(50)
    o = t4libs.Oracle(9, t4libs.REF_FA, 17)
    exp, hp, tot = o.annotate_batch(arr, arr.shape[1], arr.shape[0])
    checksum = float(np.frombuffer(arr.tobytes(), dtype=np.uint8).astype(np.int64).sum())
    dist.barrier()
    mx = t4dist.max_over_ranks(dist, float(rank + 1), "cpu")
    sm = t4dist.sum_over_ranks(dist, float(tot), "cpu")
    q.put((rank, checksum, int(tot), mx, sm, int((exp["seqIdx"] != -1).sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding():
    import t4libs
    t4libs.build_checkers()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, c0, t0, mx0, sm0, a0), (r1, c1, t1, mx1, sm1, a1) = res
    assert (r0, r1) == (0, 1)
    assert c0 != c1                      # different shards
    assert mx0 == mx1 == 2.0             # MAX over ranks
    assert sm0 == sm1 == float(t0 + t1)  # whole-job aggregate
    assert a0 > 0 and a1 > 0
