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
        # the overlapped gather of bench.py --gather: 5 blocks through 2 slots
        g = streams.FrameGatherer(64, torch.device("cpu"), slots=2)
        for k in range(5):
            blk = torch.full((64,), 16 * k + rank, dtype=torch.uint8)
            dstbufs = g.submit(blk)
            if k >= 3:          # the last two are still resident after flush
                keep = keep + [dstbufs] if k > 3 else [dstbufs]
        g.flush()
        if rank == 0:
            for j, bufs in enumerate(keep):
                for r in range(world):
                    assert bool((bufs[r] == 16 * (3 + j) + r).all())
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


def test_bench_gpus_n_starts_n_ranks_itself(tmp_path):
    """`python bench.py --gpus N` with no launcher around it must run N ranks (the
    driver's SCALE run) -- checked on the launcher alone: the children are this
    test's stand-in script, which reports the environment each rank was given."""
    import importlib.util
    import json
    import subprocess
    child = tmp_path / "child.py"
    child.write_text(
        "import json, os, sys\n"
        "json.dump({k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE',"
        " 'MASTER_ADDR', 'MASTER_PORT')}, open(sys.argv[1] + os.environ['RANK'], 'w'))\n"
        "sys.exit(3 if os.environ['RANK'] == '2' and sys.argv[2] == 'fail' else 0)\n")
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for mode, want_rc in (("ok", 0), ("fail", 3)):
        code = ("import sys; sys.argv = ['bench.py', %r, %r]; import importlib.util as u;"
                "s = u.spec_from_file_location('b', %r); b = u.module_from_spec(s);"
                "s.loader.exec_module(b); b.__file__ = %r; sys.exit(b.self_launch(4))"
                % (str(tmp_path / "rank"), mode, os.path.join(ROOT, "bench.py"), str(child)))
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")}
        r = subprocess.run([sys.executable, "-c", code], env=env, timeout=120)
        assert r.returncode == want_rc
        if mode == "fail":      # (the launcher ends the other ranks when one dies)
            continue
        ranks = [json.load(open(str(tmp_path / "rank") + str(i))) for i in range(4)]
        assert [x["RANK"] for x in ranks] == ["0", "1", "2", "3"]
        assert all(x["WORLD_SIZE"] == "4" and x["MASTER_ADDR"] == "127.0.0.1" for x in ranks)
        assert len({x["MASTER_PORT"] for x in ranks}) == 1
    # and a launcher that disagrees with --gpus is refused, not silently run as 1 rank
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr


def test_end_to_end_streams_leg_meets_and_adds_up(tmp_path):
    """bench.py --gpus N runs the ELEMENT on every GPU at once before it touches the GPU
    itself (e2e_streams_leg: one child pipeline per rank with FLUTL_DEVICE = its GPU; the
    ranks meet through files).  Two ranks here, the pipeline run replaced by a stand-in that
    reports the environment it was given: the ranks wait for each other's warm-up, each
    measures its own stream of config 3 (seed 20 + rank) on its own device, and rank 0's
    block is all frames over the common span."""
    import json
    import subprocess
    fake = tmp_path / "fake"
    fake.mkdir()
    (fake / "e2e_timing.py").write_text(
        "import os, time\n"
        "def measure(ttml, W, H, extra_env=None, **kw):\n"
        "    t0 = time.monotonic() * 1e6\n"
        "    time.sleep(0.2)\n"
        "    return {'buffers': 101, 'frames_per_s': 500.0, 'record_us_per_buffer': 1.0,\n"
        "            'device_us_per_buffer': 1.0, 't_first_us': t0, 't_end_us': t0 + 2e5,\n"
        "            'seen': [os.path.basename(ttml), extra_env.get('FLUTL_DEVICE')]}\n")
    code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r); import e2e_timing;"
            "assert 'fake' in e2e_timing.__file__; import importlib.util as u;"
            "s = u.spec_from_file_location('b', %r); b = u.module_from_spec(s); s.loader.exec_module(b);"
            "import gstenv; gstenv.available = lambda d: True;"
            "rank = int(sys.argv[1]); meet = b.e2e_streams_leg(rank, rank, 2, 3840, 2160);"
            "print(json.dumps(b.e2e_streams_collect(meet, 2)) if rank == 0 else '{}')"
            % (os.path.join(ROOT, "tools"), str(fake), os.path.join(ROOT, "bench.py")))
    env = dict(os.environ, MASTER_PORT="45%03d" % (os.getpid() % 1000), PYTHONPATH=str(fake))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    block = json.loads(outs[0][0].strip().split("\n")[-1])
    assert block["streams"] == 2 and block["errors"] == []
    assert block["per_stream"] == [500.0, 500.0]
    # 2 x 100 frames over a span of 0.2 s and whatever the two timed runs were apart
    assert 0.2 <= block["span_s"] < 0.5
    assert abs(block["frames_per_s_total"] - 200 / block["span_s"]) < 1.0       # (span_s is rounded)
