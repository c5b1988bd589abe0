"""CPU, world_size 2 over gloo: the N>1 path of bench.py -- stream sharding,
the max-over-ranks timing reduction and the optional ordered gather of finished
frames to rank 0 -- with the frames produced by the CPU oracle standing in for
the per-rank HIP render (this test checks the distributed plumbing, not pixels)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path[:0] = [os.path.join(ROOT, "flu-plugins-oss_amd"), os.path.join(ROOT, "tests")]
    import oracle_lib as O
    import streams
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes", "k06_animation.npz"))
        W, H, n = int(z["width"]), int(z["height"]), int(z["n_states"])
        mine = streams.streams_of_rank(world, rank, world)
        assert mine == [rank]
        sched = streams.state_schedule(n, rank, step=3, batch=2, world=world)
        frames = [O.render(W, H, z[f"regions{i}"], z[f"prims{i}"], z["atlas"]) for i in sched]
        got = streams.gather_frames(torch.from_numpy(frames[0].view(np.uint8).reshape(-1).copy()))
        t = streams.max_over_ranks(0.5 + rank, torch.device("cpu"))
        if rank == 0:
            other = streams.state_schedule(n, 1, step=3, batch=2, world=world)[0]
            ok = (np.array_equal(got[0].numpy().view(np.uint32).reshape(H, W), z["frames"][sched[0]])
                  and np.array_equal(got[1].numpy().view(np.uint32).reshape(H, W), z["frames"][other]))
            q.put((ok, t, sched[0] != other))
        else:
            assert got is None
            q.put((True, t, True))
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[0] for r in res)
    assert all(abs(r[1] - 1.5) < 1e-9 for r in res)     # MAX over ranks
    assert all(r[2] for r in res)


def test_stream_partition_covers_all_streams_once():
    sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
    import streams
    for n, w in [(8, 8), (8, 3), (5, 2), (1, 4), (16, 8)]:
        got = sum((streams.streams_of_rank(n, r, w) for r in range(w)), [])
        assert got == list(range(n))
