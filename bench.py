#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X-native LK-mode hot path on BASELINE.json's config #2
("KITTI-00 full seq, FAST+LK, 1xMI355X") using the synthetic KITTI-like sequence S0 (1241x376;
there is no KITTI data offline).

One "step" = one svo_track_batch call = B consecutive stereo pairs (B+1 frames resident in HBM
before the timed region) through pyramid -> FAST -> 4-call circular LK -> compaction ->
triangulation -> RANSAC-EPnP+LM -> gates -> pose chain.  Every consecutive frame pair of the
reference is independent, so the full sequence is B-pair batches back to back.

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); rank r tracks its own
sequence (seed 100 + r), no data-path collective; rank 0 gathers the poses only (16 doubles per
pair) once per step.  scaling = "weak".

Prints ONE JSON line on rank 0 (contract in the task statement) including `roofline` for the LK
kernel (HIP-event timed on the launch stream, algorithmic bytes per SURVEY.md 8d) and
`cpu_baseline` (the CPU oracle timed on this host on a bounded sample of the same frames).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

W, H, PITCH = 1241, 376, 1280
LK_BYTES_PER_POINT_CALL = 4 * (24 * 24 + 22 * 22) + 17      # 4257 B
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="stereo pairs per step (per GPU)")
    ap.add_argument("--cpu-pairs", type=int, default=24, help="pairs of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--no-timing-marks", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the pose stage in stream order")
    ap.add_argument("--mode", choices=["lk", "orb"], default="lk",
                    help="lk = BASELINE config #2 (FAST+LK, the quoted metric); orb = config #3 (ORB extractor + "
                         "descriptor match path, the reference's shipped default track_mode)")
    ap.add_argument("--shard", choices=["sequences", "pairs"], default="sequences",
                    help="N > 1: one independent sequence per GPU (default, BASELINE config #5), or ONE sequence cut "
                         "into contiguous chunks of frame pairs with a one-frame halo, relative motions gathered "
                         "and chained on rank 0 (SURVEY.md 8e granularity 2)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1: nccl (= RCCL, the real path) or gloo (rehearsal of the "
                         "multi-rank control flow with several ranks sharing one GPU: collectives on CPU copies)")
    ap.add_argument("--frames-cache", default="", help="torch file to load/save the rendered S0 frames "
                    "(keeps profiler traces free of the renderer's torch kernels)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    gloo = args.dist_backend == "gloo"
    if gloo:                                       # rehearsal: every rank on the same card
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    mg = importlib.import_module(entry.PKG_NAME + ".multigpu")

    B = args.batch
    F = B + 1
    dev = torch.device("cuda", local_rank)
    # ---- synthetic S0 frames, resident in HBM before the timed region -----------------------
    by_pairs = args.shard == "pairs"
    seed = mg.sequence_seed(0, 1) if by_pairs else mg.sequence_seed(rank, world)
    first_frame = mg.shard_pairs(world * B + 1, world, rank)[0] if by_pairs else 0
    seq = synth.StereoSequence(width=W, height=H, n_frames=first_frame + F, seed=seed, device=dev)
    cache = args.frames_cache if world == 1 else ""
    if cache and os.path.exists(cache):
        blob = torch.load(cache)
        assert blob["L"].shape == (F, H, PITCH) and blob["seed"] == seed, "stale frames cache"
        L, R = blob["L"].to(dev), blob["R"].to(dev)
    else:
        L = torch.zeros((F, H, PITCH), dtype=torch.uint8, device=dev)
        R = torch.zeros((F, H, PITCH), dtype=torch.uint8, device=dev)
        for f in range(F):
            l, r = seq.render(first_frame + f)
            L[f, :, :W] = l
            R[f, :, :W] = r
        if cache:
            torch.save({"L": L.cpu(), "R": R.cpu(), "seed": seed}, cache)
    Lv, Rv = L[:, :, :W], R[:, :, :W]
    P1, P2 = seq.proj()
    mode_kw = {}
    if args.mode == "orb":       # config/default.yaml:75,87-93: ORB_stereof2f_pnp, minmove 0.05, maxmove 10
        mode_kw = dict(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    ctx = pkg.Context(W, H, device=local_rank, max_batch=B, P1=P1, P2=P2, **mode_kw)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)            # launches, events and the RCCL gather share one stream
    ctx.set_overlap(not args.no_overlap)          # pose stage of step k runs beside the front end of step k+1
    # two result buffers: step k's pose stage (side stream) fills one while the records of step k-1
    # are gathered from the other -- the gather never waits for the pose stage it overlaps
    res_buf = [torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)]
    pose_off = pkg.STEP_DTYPE.fields["pose"][1]
    trel_off, ok_off = pkg.STEP_DTYPE.fields["T_rel_inv"][1], pkg.STEP_DTYPE.fields["ok"][1]
    state = {"k": 0}

    def collect(res):
        """The only inter-GPU traffic: per pair 16 doubles (poses) or 17 (relative motion + ok) to rank 0."""
        host = (lambda t: t.cpu()) if gloo else (lambda t: t)
        if by_pairs:       # chunks of ONE sequence: gather, then chain on rank 0
            g = mg.gather_relative(host(mg.field_view(res, trel_off, B, 16)), host(mg.int_field(res, ok_off, B)), rank, world, dst=0)
            if rank == 0:
                ctx.chain_relative(g[0].to(dev), g[1].to(dev))
        elif world > 1:
            mg.gather_poses(host(mg.poses_view(res, pose_off, B)), rank, world, dst=0)

    def step():
        k = state["k"]
        ctx.track_batch(Lv, Rv, results=res_buf[k & 1])
        # svo_track_batch(k) has already ordered the context's stream after the pose stage of batch
        # k-1 (it reuses that stage's buffers), so batch k-1's records are complete here
        if k > 0 and (world > 1 or by_pairs):
            collect(res_buf[(k - 1) & 1])
        state["k"] = k + 1

    def drain():
        """Records of the last step: wait for its pose stage, then collect them."""
        if state["k"] > 0 and (world > 1 or by_pairs):
            ctx.wait_results()
            collect(res_buf[(state["k"] - 1) & 1])
        state["k"] = 0

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if not args.no_timing_marks:
        ctx.enable_timing(True)
        ctx.get_timing()                          # clear the log
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()                                       # every step's records are collected inside the timed region
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    elapsed = mg.max_over_ranks(elapsed, torch.device("cpu") if gloo else dev, world)

    ctx.sync()
    stage_ms = dict(ctx.get_timing()) if not args.no_timing_marks else {}
    ctx.enable_timing(False)
    results = res_buf[(args.steps - 1) & 1]
    res = np.frombuffer(results.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    n_ok = int(res["ok"].sum())
    pts_total = int(res["n_prev_kps"].sum())

    if rank == 0:
        pairs = world * B * args.steps
        value = pairs / elapsed
        out = {
            "metric": "stereo frames/sec on KITTI-00 1241x376; LK-kernel achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "stereo pairs/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": ("S0 synthetic KITTI-like stereo sequence 1241x376 (config #2 stand-in), "
                                    "FAST+LK track_mode LK_stereof2f_pnp, batched frame pairs, frames resident in HBM")
                       if args.mode == "lk" else
                       ("S0 synthetic KITTI-like stereo sequence 1241x376 (config #3 stand-in), ORB extractor + "
                        "descriptor match path track_mode ORB_stereof2f_pnp, batched frame pairs, frames resident in HBM"),
                       "pairs_per_step_per_gpu": B, "mean_keypoints_per_pair": round(pts_total / B, 1),
                       "pairs_ok_last_step": n_ok, "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                       "parallelism": ((f"one sequence in {world} chunks of frame pairs (1-frame halo), RCCL gather of "
                                        f"relative motions, prefix product on rank 0") if by_pairs else
                                       (f"sequence-per-GPU x{world}, RCCL gather of poses only" if world > 1 else "1 GPU"))},
        }
        lk_ms = stage_ms.get("lk")
        if lk_ms:
            alg_bytes = pts_total * 4 * LK_BYTES_PER_POINT_CALL       # 4 fused calls per launch
            achieved = alg_bytes / (lk_ms * 1e-3) / 1e9
            # HBM bytes per launch from the rocprofv3 PMC passes of this command (profiles/), if present
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_lk_traffic.json")
            if os.path.exists(tpath) and B == 256:
                traffic = json.load(open(tpath)).get("traffic_bytes")
            out["roofline"] = {"bound": "hbm", "kernel": "lk_kernel (4-call circular chain, one launch per step)",
                               "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "launch_ms": round(lk_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}
        elif stage_ms.get("orb_cellfast"):
            # ORB mode: the largest kernel group is the per-cell FAST over the 8-level pyramid
            # (3.09 W H bytes read per image; the candidate records are negligible)
            cf_ms = stage_ms["orb_cellfast"]
            alg_bytes = int(3.09 * W * H * 2 * (B + 1))
            achieved = alg_bytes / (cf_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "orb_cellfast_kernel (8 launches per step, one per pyramid level)",
                               "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                               "launch_ms": round(cf_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}
        else:
            out["roofline"] = None
        # ---- cpu_baseline: the oracle (CPU restatement of the reference path) on a bounded sample
        if args.cpu_pairs > 0 and world == 1 and args.mode == "lk":
            O = entry.load_oracle()
            O.build()
            n = min(args.cpu_pairs, B)
            fl = Lv[:n + 1].cpu().numpy()
            fr = Rv[:n + 1].cpu().numpy()
            prm = O.make_params(P1, P2)
            kps = O.fast(fl[0])
            pose = np.eye(4)
            c0 = time.perf_counter()
            for t in range(1, n + 1):
                _, kps, pose = O.lk_track_step(prm, fl[t - 1], fr[t - 1], fl[t], fr[t], kps, pose, threads=1)
            c1 = time.perf_counter()
            out["cpu_baseline"] = {"value": round(n / (c1 - c0), 3), "unit": "stereo pairs/s", "cores": 1,
                                   "kind": "port",
                                   "sample": f"first {n} pairs of the same S0 frames, oracle/ (CPU restatement of the "
                                             f"reference OpenCV path), 1 thread, {os.cpu_count()} host cpus visible"}
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
