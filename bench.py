#!/usr/bin/env python3
"""bench.py -- fluttml-hip headline benchmark (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of input: `--batch` (default
64) overlay frames, each a DISTINCT scene state, composited by ONE launch of the
tile kernel from display lists + glyph atlas already resident in HBM into a
ring of device frames larger than the 256 MiB Infinity Cache.  At N>1 every
rank renders its own independent stream (weak scaling, no data-path
collective); `--gather` adds the RCCL gather of finished frames to rank 0 that
SURVEY §8(e) describes for the "single ordered output" case.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

CONFIGS = {
    # name: (W, H, regions, font_px, outline_px, lines)  -- BASELINE.json configs
    "cfg1_1080p_4regions": (1920, 1080, 4, 54, 0, 2),
    "cfg2_2160p_8regions_outline": (3840, 2160, 8, 96, 3, 2),
    "cfg4_4320p_16regions": (7680, 4320, 16, 160, 3, 2),
}


def load_states(cfg, n_states, stream_seed):
    """Returns list of (regions, prims) and one shared atlas for a stream."""
    import scene_pack
    pack = scene_pack.find_pack(cfg.split("_")[0], stream_seed)
    if os.environ.get("FLUTL_BENCH_VARIANT"):
        pack = None               # diagnosis variants use the synthetic generator
    if pack is not None:
        return pack.width, pack.height, pack.states(n_states), pack.atlas, pack.source
    import scenes
    if stream_seed is None:
        stream_seed = 20
    W, H, nreg, font, outl, lines = CONFIGS[cfg]
    R, P, A = scenes.subtitle_like_scene(stream_seed, W, H, n_regions=nreg,
                                         font_px=font, outline_px=outl, lines=lines)
    variant = os.environ.get("FLUTL_BENCH_VARIANT", "full")   # diagnosis only
    if variant != "full":
        keep = {"empty": lambda p: np.zeros(len(p), bool),
                "bg": lambda p: p["op"] != 2,
                "nooutline": lambda p: ~((p["op"] == 2) & (p["rule"] == 1)),
                "noglyph": lambda p: ~((p["op"] == 2) & (p["rule"] == 0))}[variant](P)
        newidx = np.cumsum(keep) - keep
        R = R.copy()
        for r in R:
            sel = keep[r["first"]:r["first"] + r["count"]]
            r["first"] = newidx[r["first"]] if r["count"] else 0
            r["count"] = int(sel.sum())
        P = P[keep]
        if variant == "empty":
            R = R[:0]
        source = "synthetic-generator:" + variant
    rng = np.random.default_rng(stream_seed)
    states = []
    for s in range(n_states):
        # distinct scene states the way <set> animates them: colours change
        Ps = P.copy()
        glyph = (Ps["op"] == 2) & (Ps["rule"] == 0)
        Ps["color"][glyph] = 0xff000000 | int(rng.integers(0x404040, 0xffffff))
        states.append((R, Ps))
    return W, H, states, A, ("synthetic-generator" if variant == "full"
                             else "synthetic-generator:" + variant)


def profiled_traffic(cfg, batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this very
    command (profiles/<round>_<cfg>_b<batch>_summary.json, tools/profile_gpu.sh):
    WRITE_SIZE (exact for 16-byte stores) + FETCH_SIZE (raw: uncalibrated for the
    4-byte mask reads).  None if that configuration has not been profiled."""
    import glob
    pat = os.path.join(ROOT, "profiles", f"r*_{cfg.split('_')[0]}_b{batch}_summary.json")
    files = sorted(glob.glob(pat))
    if not files:
        return None
    t = json.load(open(files[-1])).get("hbm_traffic_bytes_per_launch")
    return int(t["write"] + t["fetch_raw"]) if t else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="cfg2_2160p_8regions_outline",
                    choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=64,
                    help="frames (distinct scene states) per launch = per step")
    ap.add_argument("--states", type=int, default=64, help="distinct scene states")
    ap.add_argument("--ring", type=int, default=0, help="output frames in the ring")
    ap.add_argument("--row-pad", type=int, default=0,
                    help="experiment: bytes of padding after each row of the device frames")
    ap.add_argument("--gather", action="store_true",
                    help="RCCL gather of every finished frame to rank 0")
    ap.add_argument("--over-video", action="store_true",
                    help="fused mode: composite each overlay OVER a resident video frame "
                         "(flutl_launch_batch_over); adds 4*W*H read bytes per frame")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import pyflutl as F
    import scenes
    import streams

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP path)")
    # FLUTL_BENCH_REHEARSAL=1: walk the N>1 control flow on a box with fewer GPUs
    # than ranks (all ranks share the visible devices, gloo instead of RCCL)
    rehearsal = bool(os.environ.get("FLUTL_BENCH_REHEARSAL"))
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1

    # one independent stream per rank (weak scaling); no data-path collective
    my_streams = streams.streams_of_rank(world, rank, world)
    # N = 1 is BASELINE config 2 (the canonical cfg2 document); N > 1 is config 3:
    # cfg2 with seeds 20.., one independent stream per GPU
    stream_seed = 20 + my_streams[0] if world > 1 else None
    W, H, states, atlas, source = load_states(args.config, args.states, stream_seed)
    n_states = len(states)
    B = args.batch
    stride = W * 4 + args.row_pad
    frame_bytes = stride * H
    ring = args.ring or max(2 * B, (640 << 20) // frame_bytes + 1)
    ring = (ring + B - 1) // B * B

    # ---- make everything resident in HBM -----------------------------------
    for (R, P) in states:
        assert F.validate(W, H, R, P, atlas.size) == 0
    alg_bytes = [scenes.algorithmic_bytes(W, H, R, P) for (R, P) in states]
    regs, prims, r_first, r_count = [], [], [], []
    for (R, P) in states:
        R = R.copy()
        R["first"] += sum(len(p) for p in prims)
        r_first.append(sum(len(r) for r in regs))
        r_count.append(len(R))
        regs.append(R)
        prims.append(P)
    R_all, P_all = np.concatenate(regs), np.concatenate(prims)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
    R_dev, P_dev, A_dev = dev(R_all), dev(P_all), dev(atlas)
    out = torch.empty(ring * frame_bytes, dtype=torch.uint8, device="cuda")
    video = None
    if args.over_video:      # opaque synthetic video, one frame per ring slot
        video = torch.randint(0, 256, (ring * frame_bytes,), dtype=torch.uint8, device="cuda")
        video[3::4] = 255
    # batch descriptors: step s renders states [s*B, s*B+B) mod n_states into
    # ring slots [s*B, ...) mod ring
    n_desc_steps = ((n_states * ring) // np.gcd(n_states, ring) // B
                    if (B > 1 or args.over_video) else 0)
    if B > 1 or args.over_video:
        n_desc_steps = min(max(n_desc_steps, 1), 4096)
        descs = np.zeros(n_desc_steps * B, F.FRAME_DTYPE)
        for s in range(n_desc_steps):
            for j, st in enumerate(streams.state_schedule(n_states, rank, s, B, world)):
                descs[s * B + j] = (r_first[st], r_count[st],
                                    ((s * B + j) % ring) * frame_bytes)
        D_dev = dev(descs)
    stream = torch.cuda.current_stream().cuda_stream

    def step(s):
        if B == 1 and video is None:
            st = streams.state_schedule(n_states, rank, s, 1, world)[0]
            slot = s % ring
            F.launch_frame(stream, W, H, out.data_ptr() + slot * frame_bytes,
                           R_dev.data_ptr() + r_first[st] * 16, r_count[st],
                           P_dev, A_dev)
            if args.gather and world > 1:
                streams.gather_frames(out[slot * frame_bytes:(slot + 1) * frame_bytes])
        else:
            ds = s % n_desc_steps
            if video is not None:
                F.launch_batch_over(stream, W, H, out, video,
                                    D_dev.data_ptr() + ds * B * 16, B, R_dev, P_dev, A_dev)
            else:
                F.launch_batch(stream, W, H, out, D_dev.data_ptr() + ds * B * 16, B,
                               R_dev, P_dev, A_dev, stride=stride)
            if args.gather and world > 1:
                for j in range(B):
                    slot = (ds * B + j) % ring
                    streams.gather_frames(out[slot * frame_bytes:(slot + 1) * frame_bytes])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for s in range(args.warmup):
        step(s)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        step(args.warmup + s)
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = streams.max_over_ranks(elapsed, torch.device("cpu" if rehearsal else "cuda"))

    # ---- per-launch kernel duration (HIP events on the launch stream) ------
    kern_ms = None
    if rank == 0:
        n_ev = min(args.steps, 400)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(n_ev)]
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(evs):
            a.record()
            saved = args.gather
            args.gather = False
            step(args.warmup + i)
            args.gather = saved
            b.record()
        torch.cuda.synchronize()
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    if world > 1:
        dist.barrier()

    if rank == 0:
        frames = args.steps * B * world
        fps = frames / elapsed
        mean_alg = float(np.mean(alg_bytes)) + (frame_bytes if args.over_video else 0)
        achieved = mean_alg * B / (kern_ms * 1e-3) / 1e9
        res = {
            "metric": "4K overlay frames/s" if H == 2160 else "overlay frames/s",
            "value": round(fps, 2),
            "unit": "frames/s",
            "mpix_per_s": round(fps * W * H / 1e6, 1),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": args.config, "width": W, "height": H,
                       "frames_per_step": B, "distinct_states": n_states,
                       "scene_source": source, "output_ring_frames": ring,
                       "streams": world, "gather_to_rank0": bool(args.gather),
                       "fused_over_video": bool(args.over_video),
                       "parallelism": f"{world} independent stream(s), one per GPU"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": None,
                         "kernel": "flutl_composite_kernel",
                         "kernel_ms": round(kern_ms, 5),
                         "algorithmic_bytes_per_launch": int(mean_alg * B)},
        }
        if not args.over_video:
            res["roofline"]["traffic"] = profiled_traffic(args.config, B)
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "oracle"))
            import cpu_baseline
            res["cpu_baseline"] = cpu_baseline.measure(args.config, W, H, states, atlas)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
