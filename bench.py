#!/usr/bin/env python3
"""bench.py -- stereo pairs/s of the MI355X-native hot path on BASELINE.json's config #2
("KITTI-00 full seq, FAST+LK, 1xMI355X") using the synthetic KITTI-like sequence S0 (1241x376;
there is no KITTI data offline).

One "step" = one svo_track_batch call = B consecutive stereo pairs through pyramid -> FAST -> 4-call
circular LK -> compaction -> triangulation -> RANSAC-EPnP + LM -> gates -> pose chain.  Every
consecutive frame pair of the reference is independent, so a sequence is B-pair
batches back to back; step k tracks chunk k of the rendered sequence (DISTINCT frames every step; at
most --chunks chunks are rendered and cycled).

What the one JSON line holds (rank 0):
  value            whole-job pairs/s with the frames RESIDENT IN HBM when the timed region starts (the
                   task contract's definition of `value`)
  m1               SURVEY.md 8(d) M1: the same steps with the frames in PAGE-LOCKED HOST memory, the
                   host-to-device copies INSIDE the timed region (double-buffered on a copy stream),
                   from the first svo_track_* call to the last record on the host       (N = 1 only)
  online           config #2 "single-stream": svo_add_frame, one pair per call, host frames (N = 1 only)
  roofline         lk_kernel: algorithmic HBM bytes per launch / launch time (HIP events on the launch
                   stream) against the 8 TB/s roof, the PMC-measured HBM traffic, and the VALU issue
                   roof the kernel actually sits under (measured instruction issue rates:
                   profiles/r02_valu_roof.txt); the profile-derived fields are stamped with the hash
                   of the kernel source they were measured on and nulled when it differs
  cpu_baseline     the CPU oracle on this host: 1 thread and all usable cores, bounded samples

N > 1: one process per GPU (torch.distributed, backend nccl == RCCL); no data-path collective.
  --shard sequences  (default) rank r tracks its own sequence; poses only gathered to rank 0.  weak.
  --shard pairs      ONE sequence cut into per-rank chunks of frame pairs (one-frame halo); relative
                     motions gathered, prefix product on rank 0.
  --config5          BASELINE config #5 as stated: the 8 KITTI sequence lengths (271..4661 frames) dealt
                     to the ranks; every rank runs ceil((len-1)/B) steps per sequence (so --steps is
                     ignored), per-rank busy time and the imbalance are reported.
"""
import argparse
import hashlib
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
N_SIMD, CLOCK_GHZ = 1024, 2.4                               # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md peak clock


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="stereo pairs per step (per GPU)")
    ap.add_argument("--chunks", type=int, default=4, help="distinct B-pair chunks of the sequence rendered (steps cycle through them)")
    ap.add_argument("--cpu-pairs", type=int, default=100, help="pairs of the 1-thread cpu_baseline sample: BASELINE config #1 says "
                    "\"first 100 pairs\" (0 = skip cpu_baseline)")
    ap.add_argument("--e2e-frames", type=int, default=1025, help="frames of the end-to-end leg (run_kitti_stereo from PGM and PNG "
                    "files on disk, process start included; 0 = skip)")
    ap.add_argument("--e2e-full-frames", type=int, default=4541, help="frames of the full-sequence end-to-end leg (KITTI-00's length; "
                    "0 = skip; rendered, written to PGM and PNG and removed again: about a minute of set-up)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the m1 / online legs (value, roofline, cpu_baseline only)")
    ap.add_argument("--no-self-check", action="store_true", help="skip the oracle comparison of the last step's pairs")
    ap.add_argument("--self-check-pairs", type=int, default=256, help="pairs of the LAST timed step compared with the oracle in the main "
                    "measurement and the lk_accum_sse2 leg (256 = every pair of a default step; the orb and hd legs check an eighth, at least 32)")
    ap.add_argument("--self-check-sabotage", action="store_true",
                    help="(test of the test) run the main self-check's oracle in the OTHER accumulation order: it must fail and "
                         "bench.py must exit with 3")
    ap.add_argument("--no-legs", action="store_true", help="skip the lk_accum_sse2 / orb (config #3) / hd (config #4) legs")
    ap.add_argument("--hd-batch", type=int, default=128, help="pairs per step of the hd leg (1920x1080, exactly 2000 corners)")
    ap.add_argument("--no-timing-marks", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="run the pose stage in stream order")
    ap.add_argument("--mode", choices=["lk", "orb"], default="lk",
                    help="lk = BASELINE config #2 (FAST+LK, the quoted metric); orb = config #3 (ORB extractor + "
                         "descriptor match path, the reference's shipped default track_mode)")
    ap.add_argument("--lk-accum", choices=["exact", "sse2", "simd128", "sse2_legacy"], default="exact",
                    help="order of the float sums inside the LK tracker (svo_config.lk_accum): exact = the canonical integer sums; "
                         "sse2 / simd128 / sse2_legacy = float accumulation in a lane order of upstream's x86 SIMD code as restated in "
                         "oracle/lk.c (modes 2 / 4 / 3; DESIGN.md section 2 C11), bit-identical to that oracle mode")
    ap.add_argument("--shard", choices=["sequences", "pairs"], default="sequences")
    ap.add_argument("--config5", action="store_true", help="KITTI 00-07 sequence lengths dealt to the ranks (see the docstring)")
    ap.add_argument("--scaling-table", action="store_true",
                    help="run N = 1, 2, 4, ... up to --gpus one after the other (same flags, secondary legs off) and print the curve: "
                         "one JSON line per N as the runs finish, then one summary line {\"scaling_table\": [...]} -- the 1/2/4/8-GPU "
                         "batch scaling curve of north_star in one command, e.g. `python bench.py --gpus 8 --scaling-table [--config5 | --shard pairs]`")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for N > 1: nccl (= RCCL, the real path) or gloo (rehearsal of the "
                         "multi-rank control flow with several ranks sharing one GPU: collectives on CPU copies)")
    ap.add_argument("--dist-single", action="store_true",
                    help="with --gpus 1: initialise the process group anyway (a communicator of ONE rank) and send every gather / "
                         "barrier / all-reduce of the N > 1 path through it -- the rehearsal of the RCCL calls a one-GPU box allows")
    ap.add_argument("--frames-cache", default="", help="torch file to load/save the rendered S0 frames "
                    "(keeps profiler traces free of the renderer's torch kernels)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.config5 and args.shard == "pairs":
        # the ranks of --config5 run DIFFERENT step counts; --shard pairs has a collective in every step
        ap.error("--config5 deals whole sequences to the ranks: it cannot be combined with --shard pairs")
    return args


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, one child
    process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the child's environment, exactly what
    `python -m torch.distributed.run --nproc-per-node N` would have set).  This parent never touches
    the GPU (no torch import, no HIP call) and never execs: it waits, stops the other ranks as soon as
    one of them fails (by exact PID) and returns the first non-zero exit code.  Rank 0's one JSON line
    goes straight to the inherited stdout."""
    import subprocess
    import tempfile
    # rendezvous through a FILE, not a TCP port picked here: a port found free now can be taken by another process before
    # rank 0's store binds it (the ranks would then hang in init_process_group until the timeout)
    rdv = tempfile.NamedTemporaryFile(prefix="svo_bench_rdv_", delete=False)
    rdv.close()
    os.unlink(rdv.name)                             # torch creates it; a stale file from another run must not exist
    n = args.gpus
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   SVO_BENCH_RDV_FILE=rdv.name, SVO_BENCH_SELF_LAUNCHED="1")
        env.pop("MASTER_PORT", None)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc, live, t_fail = 0, set(range(n)), None
    while live:
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc, t_fail = code, time.monotonic()
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in live:
                    procs[q].terminate()
        if live:
            time.sleep(0.05)
            if t_fail is not None and time.monotonic() - t_fail > 20:      # a rank that ignored SIGTERM
                for q in live:
                    procs[q].kill()
    try:
        os.unlink(rdv.name)
    except OSError:
        pass
    return rc


def kernel_source_hash(sse2=False):
    """sha256 over the sources lk_kernel (sse2: lk_sse2_kernel) is built from: profile-derived numbers are only valid for them."""
    h = hashlib.sha256()
    for f in ("lk.hip", "lk_common.h", "svo_device.h", "svo_kernels.h") + (("lk_sse2.hip",) if sse2 else ()):
        with open(os.path.join(entry.PKG_DIR, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def orb_source_hash():
    """sha256 over the sources of the ORB kernels (tools/gpu/orb_pmc_json.py writes the same into its profile)."""
    h = hashlib.sha256()
    for f in ("orb.hip", "orb_pattern.h", "svo_device.h", "svo_kernels.h"):
        with open(os.path.join(entry.PKG_DIR, "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def newest_profile(name, src_hash=None):
    """profiles/rNN_<name>_pmc.json of the highest round (whose source hash matches, when one is given)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{name}_pmc.json")), reverse=True):
        prof = json.load(open(path))
        if src_hash is None or prof.get("source_sha256_16") == src_hash:
            prof["_file"] = os.path.relpath(path, ROOT)
            return prof
    return None


def roofline_lk(stage_ms, pts_total, B, sse2=False, profile=None):
    """The LK launch of a mean step against the HBM roof (4257 algorithmic bytes per point per call,
    4 fused calls per launch) and against the VALU issue roof it actually sits under; the profile-derived fields
    (PMC traffic, instruction count) are valid only for the kernel source they were measured on."""
    lk_ms = stage_ms["lk"]
    src_hash = kernel_source_hash(sse2)
    alg_bytes = pts_total * 4 * LK_BYTES_PER_POINT_CALL
    achieved = alg_bytes / (lk_ms * 1e-3) / 1e9
    traffic, valu = None, None
    # profile-derived fields belong to ONE workload: the 256-pair S0 step (profiles/rNN_lk[_sse2]_pmc.json) or, for the hd leg
    # (profile = "hd_lk"), the 128-pair 1920x1080 step on 2000 corners (tools/gpu/prof_hd.sh)
    prof = newest_profile(profile, src_hash) if profile else (newest_profile("lk_sse2" if sse2 else "lk", src_hash) if B == 256 else None)
    if prof:
        traffic = prof.get("traffic_bytes")
        n_valu = prof.get("valu_wave_instructions")
        peak = prof.get("valu_peak_wave_instr_per_cycle_per_simd")
        # shader clock of the profiled launch: GRBM_GUI_ACTIVE counts the cycles of all 8 XCDs over the launch
        clk = prof.get("clock_ghz_measured") or CLOCK_GHZ
        if n_valu and peak:
            rate = n_valu / (lk_ms * 1e-3 * N_SIMD * clk * 1e9)
            valu = {"achieved": round(rate, 4), "peak": peak, "unit": "wave-instructions/cycle/SIMD",
                    "frac": round(rate / peak, 4), "clock_ghz": clk,
                    "clock_source": ("GRBM_GUI_ACTIVE / 8 XCDs / launch time of the profiled run" if prof.get("clock_ghz_measured")
                                     else "MI355X_MICROARCH.md peak clock (no GRBM pass in the profile)"),
                    "profile": prof["_file"],
                    "note": "peak = measured issue rate of v_dot2 / v_perm / v_pk_* / DPP / v_cndmask (4 cycles per "
                            "wave-instruction); profiles/r02_valu_roof.txt"}
    return {"bound": "hbm", "kernel": ("lk_sse2_kernel (lk_accum = sse2: float sums in an x86 OpenCV's lane order; " if sse2 else "lk_kernel (") +
                                      "4-call circular chain, one launch per step)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "frac_measured_traffic": (round(traffic / (lk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None),
            "launch_ms": round(lk_ms, 4), "algorithmic_bytes_per_launch": alg_bytes,
            "limiting_resource": "valu-issue", "valu": valu, "kernel_source_sha256_16": src_hash}


def roofline_orb(stage_ms, B, w, h):
    """ORB mode: the largest kernel group is the per-cell FAST over the 8-level pyramid (3.09 W H bytes
    read per image; the candidate records are negligible)."""
    cf_ms = stage_ms["orb_cellfast"]
    alg_bytes = int(3.09 * w * h * 2 * (B + 1))
    achieved = alg_bytes / (cf_ms * 1e-3) / 1e9
    oprof = newest_profile("orb", orb_source_hash()) if B == 256 else None      # only a profile of THESE kernel sources
    return {"bound": "hbm", "kernel": "orb_cellfast_kernel (a step's launches together: one per run of pyramid levels of one LDS-occupancy class)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": (oprof or {}).get("cellfast_traffic_bytes"), "profile": (oprof or {}).get("_file"),
            "launch_ms": round(cf_ms, 4), "algorithmic_bytes_per_launch": alg_bytes}


def run_leg(pkg, torch, dev, L, R, width, height, B, steps, warmup, ctx_kw):
    """`steps` svo_track_batch calls of B pairs each on a FRESH context -- overlap on, records device-resident in two
    buffers, the rendered chunks cycled -- timed like the main measurement (frames resident in HBM).  Returns the
    context (its buffers still hold the last step: svo_get_batch_tracks), the elapsed seconds, the stage times, the last
    step's records and the index of the first frame of the last step's chunk."""
    NC = (L.shape[0] - 1) // B
    ctx = pkg.Context(width, height, device=dev.index, max_batch=B, **ctx_kw)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.set_overlap(True)
    bufs = [torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)]

    def one(k):
        c = k % NC
        ctx.track_batch(L[c * B:c * B + B + 1, :, :width], R[c * B:c * B + B + 1, :, :width], results=bufs[k & 1])

    for k in range(warmup):
        one(k)
    ctx.sync()
    torch.cuda.synchronize()
    ctx.enable_timing(True)
    ctx.get_timing()
    t0 = time.perf_counter()
    for k in range(steps):
        one(k)
    ctx.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    stage_ms = dict(ctx.get_timing())
    ctx.enable_timing(False)
    recs = np.frombuffer(bufs[(steps - 1) & 1].cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    return ctx, el, stage_ms, recs, ((steps - 1) % NC) * B


def self_check(pkg, O, ctx, recs, L, R, width, f0, P1, P2, mode="lk", sse2=False, keep=0, n_check=256, seed=20261004):
    """After the timing: `n_check` pairs of the LAST step (all of them when n_check >= the step) against the CPU oracle on the same frames -- fail_stage,
    keypoint / track / inlier counts, RANSAC and LM iteration numbers, every matched track and the inlier mask byte for
    byte, the relative motion to 1e-9.  A mismatch makes bench.py exit non-zero: a throughput number for wrong results
    is not a number."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(seed)
    pairs = sorted(int(x) for x in rng.choice(len(recs), size=min(n_check, len(recs)), replace=False))
    need = sorted({f for p in pairs for f in (p, p + 1)})
    fl = {f: L[f0 + f, :, :width].cpu().numpy() for f in need}
    fr = {f: R[f0 + f, :, :width].cpu().numpy() for f in need}
    K = np.asarray(P1, np.float64).reshape(3, 4)[:, :3].copy()

    def one_lk(p):
        prm = O.make_params(P1, P2)
        kps = O.fast(fl[p])
        n_all = len(kps)
        if keep and n_all > keep:
            kps = kps[np.sort(np.argsort(-kps["response"], kind="stable")[:keep])]
        r, cur, _ = O.lk_track_step(prm, fl[p], fr[p], fl[p + 1], fr[p + 1], kps, np.eye(4), want_tracks=True, threads=1)
        r["n_cur_kps"] = min(len(cur), keep) if keep else len(cur)
        tr = r["tracks"]
        pnp = O.pnp_ransac(O.triangulate(P1, P2, tr[0], tr[1]), tr[3], K) if r["n_tracked"] >= 5 else None
        return r, [tr[0], tr[1], tr[2], tr[3]], pnp

    def one_orb(p):
        prm = O.make_params(P1, P2, min_t2=0.05 ** 2, max_t2=10.0 ** 2)
        (kL, dL), (kR, dR), (k2, d2) = (O.orb_extract(im)[:2] for im in (fl[p], fr[p], fl[p + 1]))
        r, _ = O.orb_track_step(prm, kL, dL, kR, dR, k2, d2, np.eye(4))
        t2l, t1l, t1r = O.orb_robust_match(kL, dL, kR, dR, k2, d2)
        pnp = O.pnp_ransac(O.triangulate(P1, P2, t1l, t1r), t2l, K) if len(t1l) >= 5 else None
        return r, [t1l, t1r, None, t2l], pnp

    # sse2: False / True (oracle mode 2) or the oracle's accumulation mode itself (4 = the SIMD128 order, 3 = the legacy SSE2 block)
    old = O.set_lk_accum(O.LK_ACCUM_EXACT if not sse2 else (O.LK_ACCUM_FLOAT_SSE if sse2 is True else int(sse2)))
    try:
        with ThreadPoolExecutor(max_workers=min(len(pairs), usable_cores())) as ex:
            refs = list(ex.map(one_orb if mode == "orb" else one_lk, pairs))
    finally:
        O.set_lk_accum(old)
    bad = []
    for p, (r, tr, pnp) in zip(pairs, refs):
        g = recs[p]
        for k in ("ok", "fail_stage", "n_prev_kps", "n_cur_kps", "n_tracked", "n_inliers"):
            if int(g[k]) != int(r[k]):
                bad.append(f"pair {p}: {k} {int(g[k])} != {int(r[k])}")
        got = ctx.batch_tracks(p)
        for k, name in enumerate(("t1_left", "t1_right", "t2_right", "t2_left")):
            if tr[k] is not None and got[k].tobytes() != np.ascontiguousarray(tr[k]).tobytes():
                bad.append(f"pair {p}: tracks {name} differ")
        if pnp is not None:
            for k in ("ransac_iters", "lm_iters"):
                if int(g[k]) != int(pnp[k]):
                    bad.append(f"pair {p}: {k} {int(g[k])} != {int(pnp[k])}")
            if got[4].tobytes() != pnp["mask"].tobytes():
                bad.append(f"pair {p}: RANSAC inlier mask differs")
        if r["ok"]:
            e = float(np.linalg.norm(g["T_rel_inv"].reshape(4, 4) - r["T_rel_inv"]) / np.linalg.norm(r["T_rel_inv"]))
            if not e <= 1e-9:
                bad.append(f"pair {p}: T_rel_inv off by {e:.2e}")
    return {"pairs": len(pairs), "pairs_of_step": len(recs), "ok": not bad, "compared": "fail_stage, counts, ransac_iters, lm_iters, tracks + inlier mask "
            "(bytes), T_rel_inv (1e-9) vs oracle/ on the same frames" + ("" if not sse2 else ", oracle in SSE2 accumulation order" if sse2 is True else ", oracle in accumulation mode %d" % int(sse2)),
            "which_pairs_of_last_step": ("all" if len(pairs) == len(recs) else pairs), **({"mismatches": bad[:12]} if bad else {})}


def cpu_orb(O, L, R, width, P1, P2, n1, n_all):
    """config #3 on the CPU: ORBextractor on the left and right image of every frame, then the matcher + pose step;
    1 thread frame by frame (the reference's order), then every image / every pair over the usable cores."""
    from concurrent.futures import ThreadPoolExecutor
    prm = O.make_params(P1, P2, min_t2=0.05 ** 2, max_t2=10.0 ** 2)
    cores = usable_cores()

    def orb_run(n, workers):
        fl = L[:n + 1, :, :width].cpu().numpy()
        fr = R[:n + 1, :, :width].cpu().numpy()
        c0 = time.perf_counter()
        if workers == 1:
            prev, pose = None, np.eye(4)
            for t in range(n + 1):
                cur = (O.orb_extract(fl[t])[:2], O.orb_extract(fr[t])[:2])
                if prev is not None:
                    _, pose = O.orb_track_step(prm, *prev[0], *prev[1], *cur[0], pose)
                prev = cur
        else:
            with ThreadPoolExecutor(max_workers=workers) as ex:
                feats = list(ex.map(lambda im: O.orb_extract(im)[:2], [im for t in range(n + 1) for im in (fl[t], fr[t])]))
                list(ex.map(lambda t: O.orb_track_step(prm, *feats[2 * t - 2], *feats[2 * t - 1], *feats[2 * t], np.eye(4)),
                            range(1, n + 1)))
        return n / (time.perf_counter() - c0)

    v1 = orb_run(n1, 1)
    vp = orb_run(n_all, cores)
    return {"value": round(v1, 3), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
            "sample": f"first {n1} pairs of the same S0 frames, oracle/ (CPU restatement of ORBextractor + the reference's matcher "
                      f"and pose step, not the reference binary), 1 thread, frame by frame",
            "all_cores": {"value": round(vp, 3), "cores": cores,
                          "sample": f"first {n_all} pairs: all {2 * (n_all + 1)} extractions, then all pair steps, one 1-thread oracle "
                                    f"call per worker thread, {cores} workers ({os.cpu_count()} host cpus visible)"}}


def stream_leg(pkg, stream_mod, L, R, width, height, P1, P2, mode_kw, depths=(1, 2, 4, 8, 16, 32), target=4000.0):
    """The pipelined stream behind Step_ros (FrameStream == System::StreamPush / StreamPoll): frames arrive one at a
    time in host memory, as fast as the stream accepts them; micro-batches of k pairs, at most two in flight.  Per k:
    sustained pairs/s from the first push to the last pose on the host, and the latency of a pose = its arrival minus
    the push of its frame (under saturation: the queueing a k-deep pipeline adds)."""
    res = {"definition": "frames pushed one by one from host memory (each copied into page-locked memory) as fast as the stream accepts them "
                         "(closed loop), median of three passes; micro-batches of k "
                         "pairs through svo_upload_frames + svo_track_uploaded_async(continue_chain), two in flight, records polled; "
                         "latency = pose on the host - push of its frame; poses are byte-identical to the per-frame loop (tests/test_gpu_stream.py)",
           "depths": {}}
    best = None
    for k in depths:
        n = int(min(L.shape[0], max(96, 40 * k)))
        fl = [L[f, :, :width].cpu().numpy() for f in range(n)]
        fr = [R[f, :, :width].cpu().numpy() for f in range(n)]
        ctx = pkg.Context(width, height, max_batch=k, P1=P1, P2=P2, **mode_kw)
        ctx.set_overlap(True)
        fs = stream_mod.FrameStream(ctx, k)
        passes = []
        for rep in range(4):                                   # the first pass warms the context up; the median of three counts
            fs.restart()
            t_push, t_done, ok = [], [], 0
            t0 = time.perf_counter()
            for f in range(n):
                t_push.append(time.perf_counter())
                for chunk in fs.push(fl[f], fr[f]):
                    now = time.perf_counter()
                    t_done.extend([now] * len(chunk))
                    ok += int(chunk["ok"].sum())
            for chunk in fs.flush():
                now = time.perf_counter()
                t_done.extend([now] * len(chunk))
                ok += int(chunk["ok"].sum())
            el = time.perf_counter() - t0
            if rep:
                passes.append((el, t_push, t_done, ok))
        el, t_push, t_done, ok = sorted(passes, key=lambda p: p[0])[1]
        fs.close()
        ctx.close()
        lat = np.array(t_done) - np.array(t_push[1:len(t_done) + 1])
        rate = (n - 1) / el
        res["depths"][str(k)] = {"pairs_per_s": round(rate, 1), "frames": n, "pairs_ok": ok,
                                 "latency_ms_median": round(1e3 * float(np.median(lat)), 3),
                                 "latency_ms_p90": round(1e3 * float(np.percentile(lat, 90)), 3)}
        if best is None and rate >= target:
            best = k
    res["target_pairs_per_s"] = target
    res["smallest_depth_sustaining_target"] = best
    if best is not None:
        res["latency_ms_median_at_that_depth"] = res["depths"][str(best)]["latency_ms_median"]
    return res


def pose_latency_probe(pkg, L, R, width, height, B, ctx_kw, reps=3):
    """When do the poses of batch k exist once batch k + 1 has been launched?  Two B-pair batches from page-locked host
    frames through svo_track_uploaded_async, overlap on: batch k's pose stage (side stream) runs beside batch k + 1's front
    end, whose LK grid fills the chip.  Reported: the LK launch of a batch (HIP events, a synchronous batch) and, per
    repetition, how long after the SECOND launch the first batch's records were ready (svo_results_ready, polled) and how
    long the second batch's took.  A pose stage starved of wave slots behind the next LK grid shows as first ~ second
    (round 5, float-order modes: finalize_chain_kernel waited up to a whole LK launch); the reference's Step_ros returns
    when the pose exists (src/System.cpp:60-74)."""
    pitch = (width + 255) // 256 * 256
    ctx = pkg.Context(width, height, max_batch=B, **ctx_kw)
    ctx.set_overlap(True)
    hl, hr = ctx.host_frames(B + 1, pitch), ctx.host_frames(B + 1, pitch)
    hl[:, :, :width] = L[:B + 1, :, :width].cpu().numpy()
    hr[:, :, :width] = R[:B + 1, :, :width].cpu().numpy()
    for buf in (0, 1):
        ctx.upload_frames(buf, hl, hr)
        ctx.wait_upload(buf)
    ctx.track_uploaded(0, B + 1)                                   # warm-up
    ctx.enable_timing(True)
    ctx.get_timing()
    ctx.track_uploaded(0, B + 1)
    lk_ms = float(dict(ctx.get_timing()).get("lk", 0.0))
    ctx.enable_timing(False)
    first, second = [], []
    for _ in range(reps):
        ctx.sync()
        ctx.track_uploaded_async(0, B + 1)
        ctx.track_uploaded_async(1, B + 1)
        t0 = time.perf_counter()
        while ctx.results_ready() == 0:
            pass
        first.append(1e3 * (time.perf_counter() - t0))
        ctx.collect_results(B)
        while ctx.results_ready() == 0:
            pass
        second.append(1e3 * (time.perf_counter() - t0))
        ctx.collect_results(B)
    ctx.host_free(hl)
    ctx.host_free(hr)
    ctx.close()
    return {"pairs_per_batch": B, "lk_launch_ms": round(lk_ms, 3),
            "first_batch_ready_ms_after_second_launch": [round(v, 3) for v in first],
            "second_batch_ready_ms_after_its_launch": [round(v, 3) for v in second],
            "definition": "two svo_track_uploaded_async batches back to back, overlap on; records polled with svo_results_ready"}


def e2e_leg(args, n, frame_source, P1, width, png_level=3):
    """run_kitti_stereo (the reference's CLI, batched runner) on a KITTI-layout directory of S0 frames, once from PGM and once
    from PNG files (the reference's input format): pairs/s from process start to exit.  `frame_source(t0, t1)` yields the frames
    [t0, t1) as two uint8 numpy arrays (n, h, width): the bench's resident chunks for the 1025-frame leg, the renderer for the
    4541-frame one (BASELINE config #2 as worded: the whole sequence, single stream -- src/System.cpp:31-43, 75-104)."""
    import re
    import shutil
    import subprocess
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    host = os.path.join(entry.PKG_DIR, "host")
    exe = os.path.join(host, "run_kitti_stereo")
    if not os.path.exists(exe):
        return {"error": "run_kitti_stereo is not built (run __graft_entry__.build())"}
    # both formats of n stereo frames at once: 2 x n x (0.47 MB raw + ~0.45 MB PNG); /dev/shm when it has the room
    need = int(2 * n * 376 * width * 2.2)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > need + (1 << 30) else None
    root = tempfile.mkdtemp(prefix="svo_e2e_", dir=shm)
    res = {"frames": n, "runner": f"run_kitti_stereo, batch_size {args.batch}, decode_threads = usable cores", "definition":
           "pairs_per_s = (frames - 1) / wall time of the whole process (start-up, HIP context, file read + decode, H2D, tracking, "
           "pose file); loop_pairs_per_s = the same pairs / the runner's own clock from its first decode to its last pose row; "
           "startup_ms = the runner's phase log of the best run (LZB_VIO_TIMING: the HIP runtime's own start -- hipGetDeviceCount, "
           "hipSetDevice + first stream -- is outside the library's reach); files in " + ("/dev/shm" if shm else "the temp dir") +
           f", PNG zlib level {png_level}"}
    try:
        mode = "ORB_stereof2f_pnp" if args.mode == "orb" else "LK_stereof2f_pnp"
        for fmt in ("pgm", "png"):
            d = os.path.join(root, fmt)
            for cam in (0, 1):
                os.makedirs(os.path.join(d, f"image_{cam}"))
            with open(os.path.join(d, "cfg.yaml"), "w") as f:
                f.write("%YAML:1.0\n" + f"dataset_path: {d}\n" +
                        "".join(f"camera_{c}.{k}: {v}\n" for c in "lr" for k, v in (("fx", P1[0]), ("fy", P1[5]), ("cx", P1[2]), ("cy", P1[6]))) +
                        "t_lr0: -0.537\nt_lr1: 0.0\nt_lr2: 0.0\n" + "".join(f"R_lr{i}: {1.0 if i % 4 == 0 else 0.0}\n" for i in range(9)) +
                        "num_features: 500\nnum_features_init: 20\ninit_landmarks: 5\nfeature_match_error: 3\nnum_features_tracking: 5\n"
                        "num_features_tracking_bad: 10\nnum_features_needed_for_keyframe: 60\n" + f"track_mode: {mode}\n" +
                        "inlier_rate: 0.01\niterationsCount: 500\nreprojectionError: 0.5\nconfidence: 0.99\ndisplay_scale: 1\ndisplay_x: 400\n"
                        "display_y: 200\nminmove: 0.05\nmaxmove: 10\nfMinThFAST: 7\nfIniThFAST: 20\nnLevels: 8\nfScaleFactor: 1.2\nnFeatures: 2000\n" +
                        f"batch_size: {args.batch}\n")

        def write(job):
            t, cam, img = job
            with open(os.path.join(root, "pgm", f"image_{cam}", f"{t:06d}.pgm"), "wb") as f:
                f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
                f.write(img.tobytes())
            Image.fromarray(img).save(os.path.join(root, "png", f"image_{cam}", f"{t:06d}.png"), compress_level=png_level)

        t_w = time.perf_counter()
        with ThreadPoolExecutor(max_workers=usable_cores()) as pool:        # the PNG encoder releases the GIL
            for t0 in range(0, n, 128):
                fl, fr = frame_source(t0, min(n, t0 + 128))
                list(pool.map(write, [(t0 + i, cam, fs[i]) for i in range(fl.shape[0]) for cam, fs in ((0, fl), (1, fr))]))
        res["files_written_s"] = round(time.perf_counter() - t_w, 1)
        for fmt in ("pgm", "png"):
            d = os.path.join(root, fmt)
            best, loop, clean, phases = None, None, None, None
            for it in range(4):                                  # the later runs have the files in the page cache for sure
                # runs 0..2: the documented fast exit (files closed, device synchronised, teardown left to the OS); run 3: the
                # runner's default, orderly teardown -- timed beside it so that the line shows what the opt-in buys
                env = {k: v for k, v in os.environ.items() if k != "LZB_VIO_FAST_EXIT"}
                env["LZB_VIO_TIMING"] = "1"
                env["SVO_TIMING"] = "1"
                if it < 3:
                    env["LZB_VIO_FAST_EXIT"] = "1"
                t0 = time.perf_counter()
                r = subprocess.run([exe, os.path.join(d, "cfg.yaml"), os.path.join(d, "poses.txt")], capture_output=True, env=env)
                el = time.perf_counter() - t0
                if it == 3:
                    clean = el if r.returncode == 0 else None
                    continue
                if r.returncode != 0:
                    return {"error": r.stderr.decode()[-400:]}
                if best is None or el < best:
                    best = el
                    err = r.stderr.decode()
                    m = re.search(r"batched loop: (\d+) pairs in ([0-9.]+) s", err)
                    loop = (int(m.group(1)), float(m.group(2))) if m else None
                    phases = {"hip_runtime_start": sum(float(x) for x in re.findall(r"\[svo_create\]\s+([0-9.]+) ms  hipGetDeviceCount", err)),
                              "hipSetDevice_and_first_stream": sum(float(x) for x in re.findall(r"\[svo_create\]\s+([0-9.]+) ms  hipSetDevice", err)),
                              "library (code objects, one device allocation, events, page-locked scratch)":
                                  sum(float(x) for x in re.findall(r"\[svo_create\]\s+([0-9.]+) ms  (?:plan|one device|events|stream sync)", err))}
                    tm = {name: float(sec) for sec, name in re.findall(r"\[TIMING\]\s+([0-9.]+) s  (.+)", err)}
                    if "svo_create done" in tm:
                        phases["process_start_to_context_ready_and_chunk_0_decoded"] = round(1e3 * tm.get("page-locked buffers allocated", tm["svo_create done"]), 1)
                    if "loop done (last pose row written)" in tm:
                        phases["process_start_to_last_pose_row"] = round(1e3 * tm["loop done (last pose row written)"], 1)
                    phases = {k: round(v, 1) for k, v in phases.items()}
            rows = sum(1 for _ in open(os.path.join(d, "poses.txt")))
            res[fmt] = {"pairs_per_s": round((n - 1) / best, 1), "seconds": round(best, 3), "pose_rows": rows,
                        "exit": "LZB_VIO_FAST_EXIT=1 (outputs closed + svo_sync, then _exit)",
                        "seconds_orderly_teardown": round(clean, 3) if clean else None,
                        # the runner's own clock around its loop: first decode to last pose row (process start, context
                        # creation and buffer allocation -- most of a 1025-frame run's wall time -- excluded)
                        "loop_pairs_per_s": round(loop[0] / loop[1], 1) if loop else None,
                        "loop_seconds": round(loop[1], 4) if loop else None, "startup_ms": phases}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return res


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 64))


def scaling_table(args):
    """`--scaling-table`: the parent (which never touches the GPU) runs `bench.py --gpus N` for N = 1, 2, 4, ... <= --gpus,
    each exactly as the driver would start it (self-launched ranks), and collects rank 0's line of every run."""
    import subprocess
    flags, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a == "--scaling-table":
            continue
        if a == "--gpus":
            skip = True
            continue
        if a.startswith("--gpus="):
            continue
        flags.append(a)
    for extra in ("--no-secondary", "--no-legs"):
        if extra not in flags:
            flags.append(extra)
    if "--cpu-pairs" not in flags:
        flags += ["--cpu-pairs", "0"]
    rows, rc = [], 0
    n = 1
    while n <= args.gpus:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
        t0 = time.perf_counter()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(n)] + flags, capture_output=True, env=env)
        wall = time.perf_counter() - t0
        line = next((ln for ln in reversed(r.stdout.decode().splitlines()) if ln.startswith("{")), None)
        if r.returncode != 0 or line is None:
            print(f"bench.py --scaling-table: the {n}-GPU run failed ({r.returncode}): {r.stderr.decode()[-600:]}", file=sys.stderr, flush=True)
            rows.append({"n_gpus": n, "error": r.returncode})
            rc = rc or r.returncode or 1
        else:
            print(line, flush=True)
            o = json.loads(line)
            rows.append({"n_gpus": n, "value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "scaling": o["scaling"],
                         "ranks": o.get("ranks"), "driver_wall_s": round(wall, 2), "imbalance_max_over_mean": (o.get("config5") or {}).get("imbalance_max_over_mean")})
        n *= 2
    base = next((r["value"] for r in rows if r.get("n_gpus") == 1 and "value" in r), None)
    for r in rows:
        if base and "value" in r:
            r["speedup_vs_1"] = round(r["value"] / base, 3)
    print(json.dumps({"scaling_table": rows, "flags": flags}), flush=True)
    for r in rows:
        print(f"  N={r['n_gpus']}: " + (f"{r['value']:.0f} {r['unit']}, x{r.get('speedup_vs_1')}" if "value" in r else "FAILED"), file=sys.stderr)
    return rc


def main():
    args = parse()
    if args.scaling_table and "RANK" not in os.environ:
        sys.exit(scaling_table(args))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))                # before anything in this process touches the GPU
    import torch
    import torch.distributed as dist

    exit_code = 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    gloo = args.dist_backend == "gloo"
    # --gpus is the contract: never report a run as N GPUs that was not N ranks on N cards
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus N` "
                         f"(it launches the ranks itself) or under torch.distributed.run with --nproc-per-node N")
    n_dev = torch.cuda.device_count()               # counting devices does not initialise the GPU
    if n_dev < 1 or (not gloo and n_dev < world):
        raise SystemExit(f"bench.py: --gpus {args.gpus} over {args.dist_backend} needs {world if not gloo else 1} GPU(s), this node "
                         f"has {n_dev} (one rank per GPU; --dist-backend gloo rehearses the control flow with ranks sharing a card)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if gloo:                                       # rehearsal: every rank on the same card
        local_rank = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or args.dist_single
    if dist_on:
        # under torch.distributed.run the launcher's MASTER_ADDR / MASTER_PORT store is used; the ranks bench.py starts
        # itself (and a --dist-single run) meet through a file: no port to lose a race for
        init_kw = {}
        rdv_file = os.environ.get("SVO_BENCH_RDV_FILE")
        if rdv_file is None and "MASTER_PORT" not in os.environ:
            import tempfile
            fd, rdv_file = tempfile.mkstemp(prefix="svo_bench_rdv_")
            os.close(fd)
            os.unlink(rdv_file)
        if rdv_file is not None:
            init_kw["init_method"] = "file://" + rdv_file
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo", rank=rank, world_size=world, **init_kw)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank), **init_kw)

    pkg = entry.load_package()
    import importlib
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    mg = importlib.import_module(entry.PKG_NAME + ".multigpu")

    B = args.batch
    dev = torch.device("cuda", local_rank)
    coll_dev = torch.device("cpu") if gloo else dev
    by_pairs = args.shard == "pairs"
    # ---- how many steps this rank runs, on which chunks, with how many pairs each ---------------------
    steps = args.steps
    my_seqs = None
    plan = [(None, B, 0)] * steps                  # (sequence, pairs in this step, first pair's index in the sequence)
    if args.config5:
        deal = mg.deal_sequences(mg.KITTI_LENGTHS, world)
        my_seqs = deal[rank]
        plan = []
        for s_ in my_seqs:                         # a sequence = full batches + one ragged last batch
            n_pairs = mg.KITTI_LENGTHS[s_] - 1
            plan += [(s_, min(B, n_pairs - o), o) for o in range(0, n_pairs, B)]
        steps = len(plan)
        assert steps == mg.steps_for(mg.KITTI_LENGTHS, my_seqs, B)
    NC = max(1, min(args.chunks, max(steps, 1)))
    F = NC * B + 1
    # ---- synthetic S0 frames: NC chunks of B pairs (+ the halo frame), resident in HBM ----------
    seed = mg.sequence_seed(0, 1) if by_pairs else mg.sequence_seed(rank, world)
    first_frame = mg.shard_pairs(world * NC * B + 1, world, rank)[0] if by_pairs else 0
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
    P1, P2 = seq.proj()
    mode_kw = {}
    if args.mode == "orb":       # config/default.yaml:75,87-93: ORB_stereof2f_pnp, minmove 0.05, maxmove 10
        mode_kw = dict(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    if args.lk_accum != "exact":
        mode_kw["lk_accum"] = {"sse2": pkg.LK_ACCUM_SSE2, "simd128": pkg.LK_ACCUM_SIMD128, "sse2_legacy": pkg.LK_ACCUM_SSE2_LEGACY}[args.lk_accum]
    ctx = pkg.Context(W, H, device=local_rank, max_batch=B, P1=P1, P2=P2, **mode_kw)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)            # launches, events and the RCCL gather share one stream
    ctx.set_overlap(not args.no_overlap)          # pose stage of step k runs beside the front end of step k+1
    # two result buffers: step k's pose stage (side stream) fills one while the records of step k-1
    # are gathered from the other -- the gather never waits for the pose stage it overlaps
    res_buf = [torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)]
    pose_off = pkg.STEP_DTYPE.fields["pose"][1]
    trel_off, ok_off = pkg.STEP_DTYPE.fields["T_rel_inv"][1], pkg.STEP_DTYPE.fields["ok"][1]
    state = {"k": 0, "ctx": None}
    # config #5: a sequence runs at ITS frame size with ITS rig (KITTI 03: 1242x375, 04-07: 1226x370; a context and a set
    # of rendered chunks per size).  `sized[rig]` = (L, R, width, context); the 1241x376 entry is the main one.
    sized = {id(mg.KITTI_RIG_A): (L, R, W, ctx)}
    if args.config5:
        for s_ in my_seqs:
            rig = mg.KITTI_RIGS[s_]
            if id(rig) in sized:
                continue
            seq_r = synth.StereoSequence(n_frames=F, seed=seed + 1000 + rig["width"], device=dev, **rig)
            Lr = torch.zeros((F, rig["height"], PITCH), dtype=torch.uint8, device=dev)
            Rr = torch.zeros_like(Lr)
            for f in range(F):
                l, r = seq_r.render(f)
                Lr[f, :, :rig["width"]] = l
                Rr[f, :, :rig["width"]] = r
            P1r, P2r = seq_r.proj()
            cr = pkg.Context(rig["width"], rig["height"], device=local_rank, max_batch=B, P1=P1r, P2=P2r, **mode_kw)
            cr.set_stream(stream.cuda_stream)
            cr.set_overlap(not args.no_overlap)
            sized[id(rig)] = (Lr, Rr, rig["width"], cr)
    # config #5: every sequence's relative motions + ok flags stay on the device until the sequence is
    # chained (svo_chain_relative) and its FULL pose list goes to rank 0 in the one ragged gather
    acc = ({s_: torch.zeros((mg.KITTI_LENGTHS[s_] - 1, 17), dtype=torch.float64, device=dev) for s_ in my_seqs}
           if args.config5 else {})
    gathered = {}

    def chunk(k, n_pairs, s_):
        """(left frames, right frames, context) of step k: chunk k % NC of the frames rendered at the sequence's size."""
        Ls, Rs, ws, cs = sized[id(mg.KITTI_RIGS[s_] if s_ is not None else mg.KITTI_RIG_A)]
        c = k % NC
        return Ls[c * B:c * B + n_pairs + 1, :, :ws], Rs[c * B:c * B + n_pairs + 1, :, :ws], cs

    def collect(res, item):
        """The only inter-GPU traffic: per pair 16 doubles (poses) or 17 (relative motion + ok) to rank 0."""
        host = (lambda t: t.cpu()) if gloo else (lambda t: t)
        s_, n, off = item
        if by_pairs:       # chunks of ONE sequence: gather, then chain on rank 0
            g = mg.gather_relative(host(mg.field_view(res, trel_off, B, 16)), host(mg.int_field(res, ok_off, B)), rank, world, dst=0)
            if rank == 0:
                ctx.chain_relative(g[0].to(dev), g[1].to(dev))
        elif args.config5:  # no collective per step (the ranks run different step counts): keep the records
            acc[s_][off:off + n, :16] = mg.field_view(res, trel_off, n, 16)
            acc[s_][off:off + n, 16] = mg.int_field(res, ok_off, n).to(torch.float64)
        elif dist_on:
            mg.gather_poses(host(mg.poses_view(res, pose_off, B)), rank, world, dst=0)

    def step(items):
        k = state["k"]
        Lk, Rk, ck = chunk(k, items[k % len(items)][1], items[k % len(items)][0])
        if state["ctx"] is not None and state["ctx"] is not ck:
            state["ctx"].wait_results()            # another frame size = another context: its pose stage is not ordered
        state["ctx"] = ck                          # before the next launch by svo_track_batch itself
        ck.track_batch(Lk, Rk, results=res_buf[k & 1])
        # svo_track_batch(k) has already ordered the context's stream after the pose stage of batch
        # k-1 (it reuses that stage's buffers), so batch k-1's records are complete here
        if k > 0 and (dist_on or by_pairs or args.config5):
            collect(res_buf[(k - 1) & 1], items[(k - 1) % len(items)])
        state["k"] = k + 1

    def drain(items):
        """Records of the last step: wait for its pose stage, then collect them; config #5: chain every
        sequence on the device and send the full pose lists to rank 0 (one ragged gather)."""
        k = state["k"]
        if k > 0 and (dist_on or by_pairs or args.config5):
            (state["ctx"] or ctx).wait_results()
            collect(res_buf[(k - 1) & 1], items[(k - 1) % len(items)])
        state["ctx"] = None
        if args.config5:
            lists = [ctx.chain_relative(acc[s_][:, :16].contiguous(), acc[s_][:, 16].to(torch.int32)) for s_ in my_seqs]
            mine = torch.cat(lists, 0) if lists else torch.zeros((0, 16), dtype=torch.float64, device=dev)
            if gloo:
                ctx.sync()
                mine = mine.cpu()
            got = mg.gather_ragged(mine, rank, world, dst=0)
            if rank == 0:
                gathered.clear()
                for r_, d in enumerate(mg.deal_sequences(mg.KITTI_LENGTHS, world)):
                    o = 0
                    for s_ in d:
                        gathered[s_] = got[r_][o:o + mg.KITTI_LENGTHS[s_] - 1]
                        o += mg.KITTI_LENGTHS[s_] - 1
                    assert o == got[r_].shape[0], "ragged pose gather: a rank's message has the wrong length"
        state["k"] = 0

    wplan = plan[:max(1, min(args.warmup, len(plan)))] if plan else plan
    for _ in range(args.warmup if plan else 0):
        step(wplan)
    if plan or args.config5:
        drain(plan if args.config5 else wplan)     # config #5: every rank takes part in the warm-up gather too
    torch.cuda.synchronize()
    if not args.no_timing_marks:
        ctx.enable_timing(True)
        ctx.get_timing()                          # clear the log
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(plan)
    if plan or args.config5:
        drain(plan)                               # every step's records are collected inside the timed region
    torch.cuda.synchronize()
    busy = time.perf_counter() - t0               # this rank's own busy time (config #5: ranks differ)
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = mg.max_over_ranks(elapsed, coll_dev, world)

    ctx.sync()
    stage_ms = dict(ctx.get_timing()) if not args.no_timing_marks else {}
    ctx.enable_timing(False)
    last = res_buf[(steps - 1) & 1] if steps > 0 else res_buf[0]
    res = np.frombuffer(last.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    n_last = plan[-1][1] if plan else 0            # pairs of the last step (config #5: ragged)
    res = res[:n_last]
    n_ok = int(res["ok"].sum())
    mean_kps = float(res["n_prev_kps"].sum()) / max(n_last, 1)
    pts_total = int(round(mean_kps * (sum(it[1] for it in plan) / max(steps, 1))))   # points of a MEAN step: the stage times are means too

    # ---- config #5: per-rank busy time (the full pose lists were gathered inside the timed region) ----
    per_rank = None
    if args.config5:
        mine = torch.tensor([[float(rank), float(steps), busy, float(sum(mg.KITTI_LENGTHS[s_] for s_ in my_seqs))]],
                            dtype=torch.float64, device=coll_dev)
        rows = mg.gather_ragged(mine, rank, world, dst=0)
        if rank == 0:
            tab = torch.cat(rows, 0).cpu().numpy()
            per_rank = [{"rank": int(r[0]), "sequences": mg.deal_sequences(mg.KITTI_LENGTHS, world)[int(r[0])],
                         "frames": int(r[3]), "steps": int(r[1]), "busy_s": round(float(r[2]), 4)} for r in tab]
            assert sorted(gathered) == list(range(len(mg.KITTI_LENGTHS)))
            assert all(gathered[s_].shape == (mg.KITTI_LENGTHS[s_] - 1, 16) for s_ in gathered)

    out = None
    if rank == 0:
        if args.config5:
            pairs = sum(n - 1 for n in mg.KITTI_LENGTHS)       # the last batch of a sequence is ragged: exact pair count
        else:
            pairs = world * B * steps
        value = pairs / elapsed
        wl = ("S0 synthetic KITTI-like stereo sequence 1241x376 (config #2 stand-in), FAST+LK track_mode LK_stereof2f_pnp"
              if args.mode == "lk" else
              "S0 synthetic KITTI-like stereo sequence 1241x376 (config #3 stand-in), ORB extractor + descriptor match path "
              "track_mode ORB_stereof2f_pnp")
        out = {
            "metric": "stereo frames/sec on KITTI-00 1241x376; LK-kernel achieved HBM GB/s vs peak",
            "value": round(value, 2), "unit": "stereo pairs/s", "n_gpus": world, "ranks": world,
            "dist_backend": (args.dist_backend + (" (RCCL)" if not gloo else " (rehearsal: ranks share a card)") +
                             (", one-rank communicator (--dist-single)" if world == 1 else "")) if dist_on else None,
            "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / max(steps, 1), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl + f", batched frame pairs ({B} per step, {NC} distinct chunks cycled), frames resident in HBM "
                                      "for `value`; `m1` = host-resident frames, H2D included",
                       "lk_accum": (None if args.mode != "lk" else
                                    "exact (canonical C0: the five LK sums as exact integers, order-free; NOT an x86 OpenCV's float order -- "
                                    "see the lk_accum_* legs and value_x86_order)" if args.lk_accum == "exact" else
                                    args.lk_accum + " (float sums in a restated x86 OpenCV lane order, DESIGN.md section 2)"),
                       "pairs_per_step_per_gpu": B, "mean_keypoints_per_pair": round(mean_kps, 1),
                       "pairs_ok_last_step": n_ok, "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
                       "parallelism": ((f"one sequence in {world} chunks of frame pairs (1-frame halo), RCCL gather of "
                                        f"relative motions, prefix product on rank 0") if by_pairs else
                                       (f"sequence-per-GPU x{world}, RCCL gather of poses only" if world > 1 else "1 GPU"))},
        }
        if args.config5:
            busys = [p["busy_s"] for p in per_rank]
            out["config"]["config5"] = {"kitti_lengths": list(mg.KITTI_LENGTHS), "per_rank": per_rank,
                                        "imbalance_max_over_mean": round(max(busys) / (sum(busys) / len(busys)), 3),
                                        "poses_gathered": {"sequences": len(gathered), "pairs": int(sum(g.shape[0] for g in gathered.values())),
                                                           "bytes": int(sum(g.numel() * 8 for g in gathered.values()))},
                                        "note": "--steps ignored: every rank runs ceil((len-1)/B) steps per sequence, the last one ragged; "
                                                "each sequence's relative motions are chained on its GPU and its FULL pose list goes to rank 0 "
                                                "in one ragged gather inside the timed region; every sequence runs at ITS KITTI frame size and rig "
                                                "(00-02 1241x376, 03 1242x375, 04-07 1226x370: a context per size), on the rank's rendered chunks "
                                                "of that size, cycled (sequence LENGTHS and SIZES are modelled, not their content)",
                                        "frame_sizes": [f"{r['width']}x{r['height']}" for r in mg.KITTI_RIGS]}
        out["roofline"] = (roofline_lk(stage_ms, pts_total, B, sse2=args.lk_accum != "exact",
                                       profile=("lk_" + args.lk_accum if args.lk_accum in ("simd128", "sse2_legacy") else None))   # (counters exist for exact / sse2)
                           if stage_ms.get("lk") else
                           roofline_orb(stage_ms, B, W, H) if stage_ms.get("orb_cellfast") else None)
        if world == 1 and not args.config5 and not args.no_self_check and steps > 0:
            O = entry.load_oracle()
            O.build()
            # the context still holds the last step's tracks and masks (B pairs, overlap on, chunks cycled)
            out["self_check"] = self_check(pkg, O, ctx, res, L, R, W, ((steps - 1) % NC) * B, P1, P2, mode=args.mode,
                                           sse2=({"exact": False, "sse2": True, "simd128": 4, "sse2_legacy": 3}[args.lk_accum] if not args.self_check_sabotage
                                                 else args.lk_accum == "exact"),
                                           n_check=args.self_check_pairs if args.mode == "lk" else max(32, args.self_check_pairs // 8))

    # ---- secondary legs (N = 1): M1 with H2D inside the timed region, and the online path ---------
    if world == 1 and not args.no_secondary and not args.config5:
        hostL = [ctx.host_frames(B + 1, PITCH) for _ in range(NC)]
        hostR = [ctx.host_frames(B + 1, PITCH) for _ in range(NC)]
        for c in range(NC):
            hostL[c][:] = L[c * B:c * B + B + 1].cpu().numpy()
            hostR[c][:] = R[c * B:c * B + B + 1].cpu().numpy()

        def m1_run(n_steps):
            """upload(0) track(0) upload(1) | track(k) upload(k+1) collect(k-1) ...: two batches outstanding, so chunk k+1's
            H2D and (overlap mode) chunk k's pose stage run beside chunk k+1's front end; every step's records
            reach the host; the pose chain continues across the chunks on the device."""
            ctx.upload_frames(0, hostL[0], hostR[0])
            ctx.track_uploaded_async(0, B + 1)
            if n_steps > 1:
                ctx.upload_frames(1, hostL[1 % NC], hostR[1 % NC])
            recs = None
            for k in range(1, n_steps):
                ctx.track_uploaded_async(k & 1, B + 1, continue_chain=True)
                # chunk k+1's copy is queued before chunk k-1's records are waited for (the copy itself waits ON THE
                # DEVICE until chunk k-1's kernels have read the buffer): the pose stage of chunk k-1 runs beside
                # chunk k's kernels and can finish late, and the next copy must not queue up behind that wait
                if k + 1 < n_steps:
                    c = (k + 1) % NC
                    ctx.upload_frames((k + 1) & 1, hostL[c], hostR[c])
                recs = ctx.collect_results(B)                       # records of step k - 1 on the host
            recs = ctx.collect_results(B)                           # ... and of the last step
            return recs

        m1_run(min(2, args.warmup + 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        recs = m1_run(steps)
        m1_el = time.perf_counter() - t0
        out["m1"] = {"value": round(B * steps / m1_el, 2), "unit": "stereo pairs/s", "ms_per_step": round(1e3 * m1_el / steps, 4),
                     "definition": "SURVEY.md 8(d) M1: page-locked host frames, H2D inside the timed region (double-buffered "
                                   "svo_upload_frames beside the previous batch's kernels), first svo_track_* call to last record on the host",
                     "h2d_bytes_per_step": int(2 * (B + 1) * H * PITCH), "pairs_ok_last_step": int(recs["ok"].sum())}
        for c in range(NC):
            ctx.host_free(hostL[c])
            ctx.host_free(hostR[c])
        # online: one pair per svo_add_frame call, frames in (pageable) host memory
        n_on = min(64, F)
        octx = pkg.Context(W, H, device=local_rank, max_batch=1, P1=P1, P2=P2, **mode_kw)
        fl = [L[f, :, :W].cpu().numpy().copy() for f in range(n_on)]
        fr = [R[f, :, :W].cpu().numpy().copy() for f in range(n_on)]
        lat, ok_on = [], 0
        for f in range(n_on):
            t0 = time.perf_counter()
            rc, _ = octx.add_frame(fl[f], fr[f])
            lat.append(time.perf_counter() - t0)
            ok_on += int(rc == 0)
        octx.close()
        lat = np.array(lat[8:]) * 1e3
        out["online"] = {"ms_per_pair_median": round(float(np.median(lat)), 3), "ms_per_pair_p90": round(float(np.percentile(lat, 90)), 3),
                         "pairs_per_s": round(1e3 / float(np.mean(lat)), 1), "pairs": len(lat), "pairs_ok": ok_on - 8,
                         "definition": "config #2 single-stream: svo_add_frame per stereo pair, frames in host memory, result on the host"}
        # ---- the pipelined stream behind Step_ros: k frames of latency for throughput -------------------------------
        if not args.no_legs:
            out["stream"] = stream_leg(pkg, importlib.import_module(entry.PKG_NAME + ".stream"), L, R, W, H, P1, P2,
                                       {k: v for k, v in mode_kw.items()})
            if args.mode == "lk" and args.lk_accum == "exact":
                # the same stream with the LK sums in an x86 float order (round 5: the pose stage's last launch could starve behind
                # lk_sse2_kernel's single-wave workgroups; the stream's latency would have shown it)
                out["stream_sse2"] = stream_leg(pkg, importlib.import_module(entry.PKG_NAME + ".stream"), L, R, W, H, P1, P2,
                                                dict(mode_kw, lk_accum=pkg.LK_ACCUM_SSE2), depths=(2, 8, 32))
        # ---- legs on their own contexts (N = 1): the other BASELINE configs and the x86-order LK mode ----------------
        if not args.no_legs:
            O = None
            if not args.no_self_check or args.cpu_pairs > 0:
                O = entry.load_oracle()
                O.build()
            # as many timed steps as the main measurement (at most 20) after two warm-up steps: six steps after one (rounds 3-5) still
            # carried the pipeline's ramp -- the ORB leg read 3 % below the same workload run as its own process
            leg_steps, leg_warm = max(2, min(steps, 20)), 2

            def finish(name, lctx, el, st_ms, recs, f0, n_steps, Bl, extra, check_kw, frames=(L, R, W), proj=(P1, P2)):
                """The leg's entry of the JSON line; its last step is checked against the oracle before the context goes."""
                leg = {"value": round(Bl * n_steps / el, 2), "unit": "stereo pairs/s", "ms_per_step": round(1e3 * el / n_steps, 4),
                       "steps": n_steps, "pairs_per_step": Bl, "pairs_ok_last_step": int(recs["ok"].sum()),
                       "mean_keypoints_per_pair": round(float(recs["n_prev_kps"].mean()), 1),
                       "stage_ms_per_step": {k: round(v, 4) for k, v in st_ms.items()}, **extra}
                if O is not None and not args.no_self_check:
                    leg["self_check"] = self_check(pkg, O, lctx, recs, frames[0], frames[1], frames[2], f0, proj[0], proj[1], **check_kw)
                if leg.get("cpu_baseline"):
                    leg["vs_cpu_baseline_1_thread"] = round(leg["value"] / leg["cpu_baseline"]["value"], 1)
                lctx.close()
                out[name] = leg

            if args.mode == "lk" and args.lk_accum == "exact":
                # (1) the LK tracker in an x86 OpenCV's accumulation order: what bit-identity with the reference CPU path costs
                lctx, el, st_ms, recs, f0 = run_leg(pkg, torch, dev, L, R, W, H, B, leg_steps, leg_warm,
                                                    dict(P1=P1, P2=P2, lk_accum=pkg.LK_ACCUM_SSE2))
                finish("lk_accum_sse2", lctx, el, st_ms, recs, f0, leg_steps, B,
                       {"definition": "the main workload with svo_config.lk_accum = SVO_LK_ACCUM_SSE2 (lk_sse2_kernel: float sums in the "
                                      "lane order of upstream's CV_SSE2 block, bit-identical to oracle/lk.c mode 2)",
                        "roofline": roofline_lk(st_ms, int(round(float(recs["n_prev_kps"].mean()) * B)), B, sse2=True) if st_ms.get("lk") else None,
                        "lk_ms_per_step": round(st_ms.get("lk", 0.0), 4),
                        "lk_ms_per_step_exact": round(stage_ms.get("lk", 0.0), 4) if stage_ms.get("lk") else None,
                        "pose_latency": pose_latency_probe(pkg, L, R, W, H, B, dict(P1=P1, P2=P2, lk_accum=pkg.LK_ACCUM_SSE2))},
                       dict(mode="lk", sse2=True, n_check=args.self_check_pairs))
                # (1b) the other two restated x86 orders, briefly (which one an OpenCV 3 build runs depends on
                # its version): the same workload, 64 pairs of the last step checked against their oracle modes (4 / 3)
                for oname, oacc, omode in (("simd128", pkg.LK_ACCUM_SIMD128, 4), ("sse2_legacy", pkg.LK_ACCUM_SSE2_LEGACY, 3)):
                    lctx, el, st_ms, recs, f0 = run_leg(pkg, torch, dev, L, R, W, H, B, max(2, leg_steps // 2), 1,
                                                        dict(P1=P1, P2=P2, lk_accum=oacc))
                    finish("lk_accum_" + oname, lctx, el, st_ms, recs, f0, max(2, leg_steps // 2), B,
                           {"definition": "the main workload with svo_config.lk_accum = " + oname + " (lk_sse2_kernel, bit-identical to oracle/lk.c mode %d)" % omode,
                            "roofline": (roofline_lk(st_ms, int(round(float(recs["n_prev_kps"].mean()) * B)), B, sse2=True, profile="lk_" + oname)
                                         if st_ms.get("lk") else None),
                            "lk_ms_per_step": round(st_ms.get("lk", 0.0), 4)},
                           dict(mode="lk", sse2=omode, n_check=min(64, args.self_check_pairs)))
            if args.mode == "lk":
                # (2) BASELINE config #3: the ORB extractor + descriptor-match path on the same frames
                lctx, el, st_ms, recs, f0 = run_leg(pkg, torch, dev, L, R, W, H, B, leg_steps, leg_warm,
                                                    dict(P1=P1, P2=P2, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2))
                extra = {"definition": "BASELINE config #3: track_mode ORB_stereof2f_pnp (nFeatures 2000, 8 levels, 20 / 7) on the same S0 frames",
                         "roofline": roofline_orb(st_ms, B, W, H) if st_ms.get("orb_cellfast") else None}
                if O is not None and args.cpu_pairs > 0:
                    n1 = max(4, min(args.cpu_pairs // 4, B))
                    extra["cpu_baseline"] = cpu_orb(O, L, R, W, P1, P2, n1, min(max(4 * n1, 2 * usable_cores()), B))
                finish("orb", lctx, el, st_ms, recs, f0, leg_steps, B, extra, dict(mode="orb", n_check=max(32, args.self_check_pairs // 8)))
            if args.mode == "lk" and args.hd_batch > 0:
                # (3) BASELINE config #4: 1920x1080, EXACTLY the 2000 highest-response FAST corners per frame
                Bh, Wh, Hh, Ph = args.hd_batch, 1920, 1080, 1920
                seqh = synth.StereoSequence(width=Wh, height=Hh, n_frames=2 * Bh + 1, seed=1, device=dev)
                Lh = torch.zeros((2 * Bh + 1, Hh, Ph), dtype=torch.uint8, device=dev)
                Rh = torch.zeros_like(Lh)
                for f in range(2 * Bh + 1):
                    Lh[f], Rh[f] = seqh.render(f)
                P1h, P2h = seqh.proj()
                lctx, el, st_ms, recs, f0 = run_leg(pkg, torch, dev, Lh, Rh, Wh, Hh, Bh, leg_steps, leg_warm,
                                                    dict(P1=P1h, P2=P2h, max_keypoints=1 << 16, fast_keep_strongest=2000))
                pts = int(round(float(recs["n_prev_kps"].mean()) * Bh))
                extra = {"definition": "BASELINE config #4: synthetic 1920x1080 stereo stream, fast_keep_strongest = 2000 (the 2000 "
                                       "highest-response FAST(20) corners of every frame, ties by raster order), FAST+LK",
                         "roofline": roofline_lk(st_ms, pts, Bh, profile="hd_lk" if Bh == 128 else None) if st_ms.get("lk") else None}
                if O is not None and args.cpu_pairs > 0:
                    n1 = max(2, min(args.cpu_pairs // 12, Bh))
                    prm = O.make_params(P1h, P2h)
                    flh = Lh[:n1 + 1].cpu().numpy()
                    frh = Rh[:n1 + 1].cpu().numpy()
                    c0 = time.perf_counter()
                    for t in range(1, n1 + 1):
                        kp = O.fast(flh[t - 1])
                        kp = kp[np.sort(np.argsort(-kp["response"], kind="stable")[:2000])]
                        O.lk_track_step(prm, flh[t - 1], frh[t - 1], flh[t], frh[t], kp, np.eye(4), threads=1)
                    v1 = n1 / (time.perf_counter() - c0)
                    extra["cpu_baseline"] = {"value": round(v1, 3), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
                                             "sample": f"first {n1} pairs of the same 1920x1080 frames, oracle/ (FAST, selection of the 2000 "
                                                       f"strongest, LK step), 1 thread"}
                finish("hd", lctx, el, st_ms, recs, f0, leg_steps, Bh, extra, dict(mode="lk", keep=2000, n_check=max(32, args.self_check_pairs // 8)),
                       frames=(Lh, Rh, Wh), proj=(P1h, P2h))
                del Lh, Rh

    if rank == 0:
        # ---- cpu_baseline: the oracle (CPU restatement of the reference path) on bounded samples
        if args.cpu_pairs > 0 and world == 1 and args.mode == "lk":
            O = entry.load_oracle()
            O.build()
            prm = O.make_params(P1, P2)
            cores = usable_cores()

            def cpu_run(n, threads):
                fl = L[:n + 1, :, :W].cpu().numpy()
                fr = R[:n + 1, :, :W].cpu().numpy()
                kps = O.fast(fl[0])
                pose = np.eye(4)
                c0 = time.perf_counter()
                for t in range(1, n + 1):
                    _, kps, pose = O.lk_track_step(prm, fl[t - 1], fr[t - 1], fl[t], fr[t], kps, pose, threads=threads)
                return n / (time.perf_counter() - c0)

            def cpu_run_pairs(n, workers):
                """SURVEY 8(d) (ii), batch mode: the frame pairs are independent, one 1-thread oracle step per worker
                (ctypes releases the GIL); every worker runs its pair's FAST itself, the pose product is not timed."""
                from concurrent.futures import ThreadPoolExecutor
                fl = L[:n + 1, :, :W].cpu().numpy()
                fr = R[:n + 1, :, :W].cpu().numpy()

                def one(t):
                    kp = O.fast(fl[t - 1])
                    O.lk_track_step(prm, fl[t - 1], fr[t - 1], fl[t], fr[t], kp, np.eye(4), threads=1)
                c0 = time.perf_counter()
                with ThreadPoolExecutor(max_workers=workers) as ex:
                    list(ex.map(one, range(1, n + 1)))
                return n / (time.perf_counter() - c0)

            n1 = min(args.cpu_pairs, B)
            v1 = cpu_run(n1, 1)
            nn = min(4 * args.cpu_pairs, B)
            vn = cpu_run(nn, cores)
            np_pairs = min(max(4 * args.cpu_pairs, 2 * cores), B)
            vp = cpu_run_pairs(np_pairs, cores)
            out["cpu_baseline"] = {"value": round(v1, 3), "unit": "stereo pairs/s", "cores": 1, "kind": "port",
                                   "sample": f"first {n1} pairs of the same S0 frames, oracle/ (CPU restatement of the "
                                             f"reference OpenCV path, not the reference binary), 1 thread",
                                   "all_cores": {"value": round(vp, 3), "cores": cores,
                                                 "sample": f"first {np_pairs} pairs, one 1-thread oracle step per worker thread "
                                                           f"(frame pairs are independent), {cores} workers "
                                                           f"({os.cpu_count()} host cpus visible)",
                                                 "points_parallel_only": {"value": round(vn, 3),
                                                                          "sample": f"first {nn} pairs in sequence, OpenMP over the LK points, {cores} threads"}}}
            out["vs_cpu_baseline_1_thread"] = round(out["value"] / v1, 1)
            out["vs_cpu_baseline_all_cores"] = round(out["value"] / vp, 1)
        elif args.cpu_pairs > 0 and world == 1 and args.mode == "orb":
            O = entry.load_oracle()
            O.build()
            n1 = min(args.cpu_pairs, B)
            out["cpu_baseline"] = cpu_orb(O, L, R, W, P1, P2, n1, min(max(4 * args.cpu_pairs, 2 * usable_cores()), B))
            out["vs_cpu_baseline_1_thread"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
            out["vs_cpu_baseline_all_cores"] = round(out["value"] / out["cpu_baseline"]["all_cores"]["value"], 1)
        else:
            out["cpu_baseline"] = None

        # ---- e2e: the drop-in binary from image FILES (decode + H2D + tracking + pose file, process start included)
        if world == 1 and args.e2e_frames >= 3 and not args.config5 and not args.no_secondary:
            n_e2e = min(args.e2e_frames, L.shape[0])
            out["e2e"] = e2e_leg(args, n_e2e, lambda t0, t1: (L[t0:t1, :, :W].cpu().numpy(), R[t0:t1, :, :W].cpu().numpy()), P1, W)
            if args.e2e_full_frames > n_e2e:
                # BASELINE config #2 as worded: the FULL sequence (KITTI-00: 4541 frames), single stream, from image files; frames
                # beyond the resident chunks are rendered here, written out and dropped (the PNG encode runs on a thread pool
                # beside the renderer)
                seq_full = synth.StereoSequence(width=W, height=H, n_frames=args.e2e_full_frames, seed=seed, device=dev)

                def rendered(t0, t1):
                    fl, fr = seq_full.render_range(t0, t1)
                    return fl.cpu().numpy(), fr.cpu().numpy()

                try:                                         # set-up heavy (9 082 files): a full disk must not cost the whole line
                    out["e2e_full"] = e2e_leg(args, args.e2e_full_frames, rendered, P1, W)
                except Exception as exc:                     # noqa: BLE001
                    out["e2e_full"] = {"error": f"{type(exc).__name__}: {exc}"}
                if "definition" in out["e2e_full"]:
                    out["e2e_full"]["definition"] = ("BASELINE config #2 as worded (full sequence: KITTI-00's 4541 frames, single stream, image files): " +
                                                     out["e2e_full"]["definition"])
        # the same workload in the float orders an x86 OpenCV 3 can run (whichever the reference's build has): the slowest of the three
        x86 = {k: out[k]["value"] for k in ("lk_accum_sse2", "lk_accum_simd128", "lk_accum_sse2_legacy") if isinstance(out.get(k), dict)}
        if x86:
            worst = min(x86, key=x86.get)
            out["value_x86_order"] = {"value": x86[worst], "unit": "stereo pairs/s", "leg": worst, "all": x86,
                                      "definition": "`value`'s workload with the LK sums in float, in the slowest of the three restated x86 lane orders "
                                                    "(bit-identical to oracle/lk.c modes 2 / 4 / 3); `value` itself is lk_accum = exact"}
        print(json.dumps(out), flush=True)
        checks = [out.get("self_check")] + [out[k].get("self_check") for k in ("lk_accum_sse2", "orb", "hd") if isinstance(out.get(k), dict)]
        if any(c is not None and not c["ok"] for c in checks):
            print("bench.py: self_check FAILED -- the timed results differ from the oracle's: " +
                  json.dumps([c.get("mismatches") for c in checks if c and not c["ok"]]), file=sys.stderr, flush=True)
            exit_code = 3
    if dist_on:
        dist.destroy_process_group()
    for _, _, _, cs in sized.values():
        if cs is not ctx:
            cs.close()
    ctx.close()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
