"""bench.py's one-GPU line (-m gpu): the legs the driver's default run carries (lk_accum_sse2 / _simd128 / _sse2_legacy, orb = BASELINE config #3,
hd = config #4 on exactly 2000 corners) and the oracle self-check, on a small batch; and the self-check's teeth: with the
oracle deliberately run in the other accumulation order bench.py must exit with 3."""
import json
import os
import subprocess
import sys

import pytest

import conftest

pytestmark = pytest.mark.gpu


def _bench(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(conftest.ROOT, "bench.py")] + list(flags), capture_output=True,
                          timeout=timeout, env=env)


def test_default_line_carries_the_legs_and_a_green_self_check():
    r = _bench("--steps", "3", "--warmup", "1", "--batch", "12", "--chunks", "2", "--cpu-pairs", "12", "--e2e-frames", "0",
               "--hd-batch", "3")
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0 and out["cpu_baseline"]["value"] > 0
    assert out["self_check"]["ok"] and out["self_check"]["pairs"] == 12 and out["self_check"]["which_pairs_of_last_step"] == "all"
    for leg, stage in (("lk_accum_sse2", "lk"), ("lk_accum_simd128", "lk"), ("lk_accum_sse2_legacy", "lk"), ("orb", "orb_cellfast"), ("hd", "lk")):
        d = out[leg]
        assert d["value"] > 0 and d["self_check"]["ok"] and d["stage_ms_per_step"][stage] > 0, leg
        assert d["self_check"]["pairs"] == d["self_check"]["pairs_of_step"], leg          # small batches are checked whole
    assert out["lk_accum_sse2"]["lk_ms_per_step"] > out["lk_accum_sse2"]["lk_ms_per_step_exact"] > 0
    assert out["config"]["lk_accum"].startswith("exact") and out["value_x86_order"]["leg"] in ("lk_accum_sse2", "lk_accum_simd128", "lk_accum_sse2_legacy")
    assert out["value_x86_order"]["value"] == min(out[k]["value"] for k in ("lk_accum_sse2", "lk_accum_simd128", "lk_accum_sse2_legacy"))
    assert len(out["lk_accum_sse2"]["pose_latency"]["first_batch_ready_ms_after_second_launch"]) == 3
    for leg in ("lk_accum_simd128", "lk_accum_sse2_legacy"):
        assert out[leg]["roofline"]["frac"] > 0, leg
    assert out["orb"]["roofline"]["frac"] > 0 and out["orb"]["cpu_baseline"]["value"] > 0
    assert out["hd"]["mean_keypoints_per_pair"] == 2000.0 and out["hd"]["roofline"]["frac"] > 0
    assert out["hd"]["cpu_baseline"]["cores"] == 1
    st = out["stream"]
    assert sorted(int(k) for k in st["depths"]) == [1, 2, 4, 8, 16, 32]
    assert all(d["pairs_per_s"] > 0 and d["pairs_ok"] >= d["frames"] - 2 and d["latency_ms_median"] > 0 for d in st["depths"].values())
    assert st["depths"]["8"]["pairs_per_s"] > st["depths"]["1"]["pairs_per_s"]          # depth buys throughput
    assert sorted(int(k) for k in out["stream_sse2"]["depths"]) == [2, 8, 32]
    assert all(d["pairs_ok"] >= d["frames"] - 2 and d["latency_ms_median"] > 0 for d in out["stream_sse2"]["depths"].values())


def test_self_check_mismatch_exits_non_zero():
    r = _bench("--steps", "1", "--warmup", "0", "--batch", "10", "--chunks", "1", "--cpu-pairs", "0", "--e2e-frames", "0",
               "--no-legs", "--no-secondary", "--self-check-sabotage")
    assert r.returncode == 3, (r.returncode, r.stderr.decode()[-2000:])
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert out["self_check"]["ok"] is False and out["self_check"]["mismatches"]
    assert "self_check FAILED" in r.stderr.decode()
