"""KITTI's OTHER frame sizes through the whole step (-m gpu).  The reference reads whatever is on disk
(src/System.cpp:75-104): sequence 03 is 1242x375 and 04-07 are 1226x370, each with its own rectified calibration
 -- odd heights change every pyramid level size, the LK border handling and the ORB cell grid.
Eight consecutive pairs per size, LK mode (svo_track_batch and the online path) and ORB mode, at the same bars as the
1241x376 sequence tests; and run_kitti_stereo on three sequences of three DIFFERENT sizes at once."""
import os
import subprocess

import numpy as np
import pytest

import conftest
from test_gpu_parity_sequence import HOST, _check_batch, _check_online, _check_orb_sequence, _oracle_lk_sequence, _write_pgm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def mg():
    import importlib
    return importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")


def _render_rig(synth, tc, rig, n, seed):
    seq = synth.StereoSequence(n_frames=n, seed=seed, device=tc.device("cuda", 0), **rig)
    return seq, [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(n)]


@pytest.mark.parametrize("which", ["B", "C"])
def test_lk_eight_pairs_at_kitti_03_and_04_sizes(pkg, oracle, tc, synth, mg, which):
    rig = {"B": mg.KITTI_RIG_B, "C": mg.KITTI_RIG_C}[which]
    seq, frames = _render_rig(synth, tc, rig, 9, 300 + ord(which))
    assert frames[0][0].shape == (rig["height"], rig["width"])
    ref = _oracle_lk_sequence(oracle, seq, frames)
    assert all(r["ok"] for r, _, _, _ in ref) and min(r["n_tracked"] for r, _, _, _ in ref) > 1000
    _check_batch(pkg, tc, seq, frames, ref)
    _check_online(pkg, seq, frames, ref)


def test_lk_sse2_mode_at_kitti_04_size(pkg, oracle, tc, synth, mg):
    """The x86-order LK mode on a frame size whose pyramid levels are all odd-sized (1226x370 -> 613x185 -> 307x93 -> 154x47)."""
    seq, frames = _render_rig(synth, tc, mg.KITTI_RIG_C, 4, 371)
    old = oracle.set_lk_accum(oracle.LK_ACCUM_FLOAT_SSE)
    try:
        ref = _oracle_lk_sequence(oracle, seq, frames)
    finally:
        oracle.set_lk_accum(old)
    _check_batch(pkg, tc, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2)
    _check_online(pkg, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2)


@pytest.mark.parametrize("which", ["B", "C"])
def test_orb_eight_pairs_at_kitti_03_and_04_sizes(pkg, oracle, tc, synth, mg, which):
    rig = {"B": mg.KITTI_RIG_B, "C": mg.KITTI_RIG_C}[which]
    seq, frames = _render_rig(synth, tc, rig, 9, 300 + ord(which))
    _check_orb_sequence(pkg, oracle, tc, seq, frames, min_ok=8)


def test_run_kitti_stereo_three_sequences_of_three_sizes(pkg, synth, tc, mg, tmp_path):
    """`run_kitti_stereo a.yaml b.yaml c.yaml` with 1241x376, 1242x375 and 1226x370 sequences (their own YAML rigs,
    LK / ORB / batched LK): a context per size, pose files byte-identical to the single-sequence runs."""
    from test_host_api import _write_yaml
    pkg.build_library()
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    specs = [("k00", mg.KITTI_RIG_A, 7, "LK_stereof2f_pnp", ""), ("k03", mg.KITTI_RIG_B, 5, "ORB_stereof2f_pnp", ""),
             ("k04", mg.KITTI_RIG_C, 9, "LK_stereof2f_pnp", "batch_size: 4\ndecode_threads: 2\n")]
    yamls = []
    for name, rig, n, mode, extra in specs:
        seq, frames = _render_rig(synth, tc, rig, n, 40 + n)
        d = tmp_path / name
        for cam in (0, 1):
            os.makedirs(d / f"image_{cam}")
        for t, (L, R) in enumerate(frames):
            _write_pgm(d / "image_0" / f"{t:06d}.pgm", L)
            _write_pgm(d / "image_1" / f"{t:06d}.pgm", R)
        y = tmp_path / f"{name}.yaml"
        _write_yaml(y, str(d), fx=rig["fx"], fy=rig["fy"], cx=rig["cx"], cy=rig["cy"], mode=mode, baseline=rig["baseline"])
        with open(y, "a", encoding="utf-8") as f:
            f.write(extra)
        yamls.append(str(y))
    exe = os.path.join(HOST, "run_kitti_stereo")
    single = []
    for y in yamls:
        r = subprocess.run([exe, y, y + ".single.txt"], capture_output=True, timeout=300)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        single.append(open(y + ".single.txt", "rb").read())
    os.makedirs(tmp_path / "out")
    r = subprocess.run([exe] + yamls + ["--poses-dir", str(tmp_path / "out")], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    for (name, _, n, _, _), want in zip(specs, single):
        got = open(tmp_path / "out" / f"{name}.yaml.poses.txt", "rb").read()
        assert got == want and len(got.splitlines()) == n
        poses = np.loadtxt(tmp_path / "out" / f"{name}.yaml.poses.txt").reshape(-1, 3, 4)
        assert np.linalg.norm(poses[-1][:, 3]) > 0.5 * (n - 1) * 0.8          # it did move forward ~1 m per frame
