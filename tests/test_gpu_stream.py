"""The pipelined stream behind Step_ros (-m gpu): micro-batches of k frames, results k frames late, poses byte for byte
the per-frame loop's.  Python twin (stream.FrameStream on the C-ABI) for k = 1, 2, 3, 5 and a partial last batch, both
modes; the C++ System::StreamPush / StreamPoll through `run_kitti_stereo` with the YAML key stream_depth.
Reference entry point: src/System.cpp:60-74 (Step_ros: one externally supplied frame per call)."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import conftest

pytestmark = pytest.mark.gpu
HOST = os.path.join(conftest.ROOT, "stereo-visual-odometry_amd", "host")


@pytest.fixture(scope="module")
def frames11(synth):
    seq = synth.StereoSequence(width=416, height=128, n_frames=11, seed=5)
    return seq, [tuple(x.numpy() for x in seq.render(t)) for t in range(11)]


def _online(pkg, seq, frames, **kw):
    P1, P2 = seq.proj()
    c = pkg.Context(frames[0][0].shape[1], frames[0][0].shape[0], device=0, P1=P1, P2=P2, **kw)
    recs = []
    c.add_frame(*frames[0])
    for fr in frames[1:]:
        rc, g = c.add_frame(*fr)
        recs.append(g)
    c.close()
    return recs


@pytest.mark.parametrize("mode", ["lk", "orb"])
@pytest.mark.parametrize("depth", [1, 2, 3, 5])
def test_frame_stream_equals_the_per_frame_loop(pkg, frames11, depth, mode):
    stream = importlib.import_module(conftest.entry.PKG_NAME + ".stream")
    seq, frames = frames11
    kw = dict(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=100.0, orb_nlevels=3, orb_nfeatures=400) if mode == "orb" else {}
    want = _online(pkg, seq, frames, **kw)
    P1, P2 = seq.proj()
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=depth, **kw)
    c.set_overlap(True)
    fs = stream.FrameStream(c, depth)
    got, arrived_at = [], []
    for t, fr in enumerate(frames):
        for chunk in fs.push(*fr):
            got.extend(chunk)
            arrived_at.extend([t] * len(chunk))
    for chunk in fs.flush():
        got.extend(chunk)
    fs.close()
    c.close()
    assert len(got) == len(want) == 10
    for p, (g, w) in enumerate(zip(got, want)):
        for k in ("ok", "fail_stage", "n_prev_kps", "n_cur_kps", "n_tracked", "n_inliers", "ransac_iters", "lm_iters"):
            assert int(g[k]) == int(w[k]), (p, k)
        assert g["T_rel_inv"].tobytes() == w["T_rel_inv"].tobytes() and g["pose"].tobytes() == w["pose"].tobytes(), p
    # pair p rides micro-batch p // depth, launched with the push of frame (p // depth + 1) * depth; its records are
    # there at the latest when the micro-batch after the next one is launched (two in flight)
    for p, t in enumerate(arrived_at):
        assert (p // depth + 1) * depth <= t <= (p // depth + 3) * depth, (p, t)


def test_frame_stream_with_failing_pairs_and_polling(pkg, frames11):
    """A blank frame in the stream: the two pairs that touch it fail (few keypoints / few tracks), the pose chain skips
    them exactly as the per-frame loop does; svo_results_ready says 0 with nothing in flight and the pair count once a
    micro-batch is complete."""
    import time
    stream = importlib.import_module(conftest.entry.PKG_NAME + ".stream")
    seq, frames = frames11
    frames = list(frames[:9])
    blank = np.full_like(frames[0][0], 128)
    frames[4] = (blank, blank)
    want = _online(pkg, seq, frames)
    assert [int(w["ok"]) for w in want] == [1, 1, 1, 0, 0, 1, 1, 1]
    P1, P2 = seq.proj()
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=3)
    c.set_overlap(True)
    assert c.results_ready() == 0
    fs = stream.FrameStream(c, 3)
    got = []
    for fr in frames[:4]:
        for chunk in fs.push(*fr):
            got.extend(chunk)
    for _ in range(2000):                                     # the first micro-batch (3 pairs) is in flight: wait for it by polling
        if c.results_ready():
            break
        time.sleep(0.001)
    assert c.results_ready() == 3
    for fr in frames[4:]:
        for chunk in fs.push(*fr):
            got.extend(chunk)
    for chunk in fs.flush():
        got.extend(chunk)
    assert c.results_ready() == 0
    fs.close()
    c.close()
    assert len(got) == len(want) == 8
    for g, w in zip(got, want):
        assert int(g["ok"]) == int(w["ok"]) and int(g["fail_stage"]) == int(w["fail_stage"])
        assert g["pose"].tobytes() == w["pose"].tobytes() and g["T_rel_inv"].tobytes() == w["T_rel_inv"].tobytes()


def test_run_kitti_stereo_stream_depth(pkg, frames11, tmp_path):
    """YAML `stream_depth: k`: System::Run feeds Step_ros, which queues the frames (StreamPush) and writes the poses as
    they complete -- the pose file equals the per-frame loop's byte for byte, in both modes."""
    from test_host_api import _write_pgm, _write_yaml
    pkg.build_library()
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    seq, frames = frames11
    for cam in (0, 1):
        os.makedirs(tmp_path / f"image_{cam}")
    for t, (L, R) in enumerate(frames):
        _write_pgm(tmp_path / "image_0" / f"{t:06d}.pgm", L)
        _write_pgm(tmp_path / "image_1" / f"{t:06d}.pgm", R)
    exe = os.path.join(HOST, "run_kitti_stereo")
    for mode in ("LK_stereof2f_pnp", "ORB_stereof2f_pnp"):
        _write_yaml(tmp_path / "base.yaml", str(tmp_path), fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy, mode=mode)
        txt = open(tmp_path / "base.yaml", encoding="utf-8").read()
        outs = {}
        for name, extra in (("loop", ""), ("k1", "stream_depth: 1\n"), ("k4", "stream_depth: 4\n"), ("k16", "stream_depth: 16\n")):
            open(tmp_path / f"{name}.yaml", "w", encoding="utf-8").write(txt + extra)
            out = tmp_path / f"{name}.{mode}.txt"
            r = subprocess.run([exe, str(tmp_path / f"{name}.yaml"), str(out)], capture_output=True, timeout=300)
            assert r.returncode == 0, r.stderr.decode()[-2000:]
            outs[name] = open(out, "rb").read()
        assert len(outs["loop"].splitlines()) == 11
        assert outs["k1"] == outs["loop"] and outs["k4"] == outs["loop"] and outs["k16"] == outs["loop"], mode


def test_frame_stream_in_place_producer(pkg, frames11):
    """next_slot() / commit(): a producer that writes the frame where it is uploaded from -- same records as push(); and the
    halo frame of a micro-batch is carried on the device (never uploaded again): the slot-0 rows of the later micro-batches
    are poisoned on the host to prove nothing reads them."""
    stream = importlib.import_module(conftest.entry.PKG_NAME + ".stream")
    seq, frames = frames11
    P1, P2 = seq.proj()
    want = _online(pkg, seq, frames)
    c = pkg.Context(416, 128, device=0, P1=P1, P2=P2, max_batch=3)
    fs = stream.FrameStream(c, 3)
    got = []
    for t, (L, R) in enumerate(frames):
        l, r = fs.next_slot()
        l[:], r[:] = L, R
        if fs.n == 1 and fs.chunk > 0:                    # first new frame of a later micro-batch: slot 0 is the carried halo
            fs.pin[fs.buf][0][0] = 0x5A
            fs.pin[fs.buf][1][0] = 0xA5
        for chunk in fs.commit():
            got.extend(chunk)
    for chunk in fs.flush():
        got.extend(chunk)
    fs.close()
    c.close()
    assert len(got) == len(want) == 10
    for p, (g, w) in enumerate(zip(got, want)):
        assert int(g["ok"]) == int(w["ok"]) == 1 and int(g["n_tracked"]) == int(w["n_tracked"]), p
        assert g["T_rel_inv"].tobytes() == w["T_rel_inv"].tobytes() and g["pose"].tobytes() == w["pose"].tobytes(), p


@pytest.mark.parametrize("accum", ["sse2", "exact"])
def test_poses_of_a_batch_do_not_wait_for_the_next_lk_launch(pkg, synth, accum):
    """Step_ros returns when the pose exists (src/System.cpp:60-74); in the pipelined stream the pose stage of batch k runs on
    the side stream beside batch k + 1's front end.  Round 5: in the float-order LK modes finalize_chain_kernel (then one
    256-thread workgroup) found no four free wave slots behind lk_sse2_kernel's single-wave workgroups until that grid
    drained, and batch k's poses arrived a whole LK launch late (profiles/r06_sse2_timeline_before.csv).  As one wave it takes
    the first slot that frees: after the second launch, the first batch's records must be there long before the second
    batch's -- the two arrivals at least 60 % of an LK launch apart (starved, they were a pose stage apart).  256 pairs a
    batch, as the batched runner and the bench use: the next batch's pyramid + FAST kernels (1.2 ms) then cover the hypothesis
    kernels of the pose stage, and the finalize launch is the one that lands beside the LK grid."""
    import sys
    import torch
    sys.path.insert(0, conftest.ROOT)
    import bench
    B = 256
    dev = torch.device("cuda", 0)
    seq = synth.StereoSequence(width=1241, height=376, n_frames=B + 1, seed=20200710, device=dev)
    fr = [seq.render(f) for f in range(B + 1)]
    L = torch.stack([f[0] for f in fr])
    R = torch.stack([f[1] for f in fr])
    P1, P2 = seq.proj()
    kw = dict(P1=P1, P2=P2, lk_accum=pkg.LK_ACCUM_SSE2 if accum == "sse2" else 0)
    r = bench.pose_latency_probe(pkg, L, R, 1241, 376, B, kw, reps=3)
    assert r["lk_launch_ms"] > 1.0, r
    gaps = [b - a for a, b in zip(r["first_batch_ready_ms_after_second_launch"], r["second_batch_ready_ms_after_its_launch"])]
    assert sorted(gaps)[1] > 0.6 * r["lk_launch_ms"], r          # all but at most one repetition (host jitter)
    del L, R
