"""N > 1 path on CPU: world-size-2 gloo run of the sharding / pose-gather / max-over-ranks helpers
bench.py uses with RCCL on the GPUs (no GPU here, so the tracked poses are stand-ins)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import conftest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    entry = conftest.entry
    entry.load_package()
    import importlib
    mg = importlib.import_module(entry.PKG_NAME + ".multigpu")
    b = importlib.import_module(entry.PKG_NAME + ".binding")
    n = 6
    # packed svo_step_result records whose pose field encodes (rank, pair)
    rec = np.zeros(n, dtype=b.STEP_DTYPE)
    for p in range(n):
        rec["pose"][p] = np.arange(16) + 100 * rank + 1000 * p
    raw = torch.from_numpy(np.frombuffer(rec.tobytes(), dtype=np.uint8).reshape(n, b.STEP_DTYPE.itemsize).copy())
    poses = mg.poses_view(raw, b.STEP_DTYPE.fields["pose"][1], n)
    got = mg.gather_poses(poses, rank, world, dst=0)
    tmax = mg.max_over_ranks(1.0 + rank, torch.device("cpu"), world)
    mine = mg.shard_sequences(len(mg.KITTI_LENGTHS), world, rank)
    ok = tmax == float(world)
    if rank == 0:
        ok = ok and len(got) == world
        for r in range(world):
            exp = np.stack([np.arange(16) + 100 * r + 1000 * p for p in range(n)]).astype(np.float64)
            ok = ok and np.array_equal(got[r].numpy(), exp)
    else:
        ok = ok and got is None
    # frame-pair-granular sharding of one sequence: chunk c -> rank c, relative motions gathered and
    # chained on rank 0 == the serial prefix product over the whole sequence
    n_frames = 12
    rng = np.random.default_rng(5)
    Tall = np.tile(np.eye(4), (n_frames - 1, 1, 1))
    for i in range(n_frames - 1):
        a = rng.normal(size=3) * 0.05
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
        Tall[i, :3, :3] = np.eye(3) + K + K @ K / 2
        Tall[i, :3, 3] = rng.normal(size=3)
    okall = np.ones(n_frames - 1, np.int32)
    okall[[3, 7]] = 0
    first, nf = mg.shard_pairs(n_frames, world, rank)
    Tm = torch.from_numpy(Tall[first:first + nf - 1].reshape(-1, 16).copy())
    okm = torch.from_numpy(okall[first:first + nf - 1].copy())
    g = mg.gather_relative(Tm, okm, rank, world, dst=0)
    if rank == 0:
        chained = mg.chain_relative(g[0], g[1]).numpy()
        P, ref = np.eye(4), []
        for i in range(n_frames - 1):
            if okall[i]:
                P = P @ Tall[i]
            ref.append(P.copy())
        ok = ok and chained.shape == (n_frames - 1, 4, 4) and np.allclose(chained, np.array(ref), rtol=0, atol=1e-12)   # BLAS-dependent last ulps
        ok = ok and np.array_equal(g[1].numpy(), okall)
    else:
        ok = ok and g is None
    # BASELINE config #5: unequal sequence lengths dealt to the ranks, ragged result gather
    deal = mg.deal_sequences(mg.KITTI_LENGTHS, world)
    n_mine = mg.steps_for(mg.KITTI_LENGTHS, deal[rank], 256)
    ragged = torch.arange(n_mine * 16, dtype=torch.float64).view(n_mine, 16) + 1e6 * rank
    gr = mg.gather_ragged(ragged, rank, world, dst=0)
    if rank == 0:
        ok = ok and [g.shape[0] for g in gr] == [mg.steps_for(mg.KITTI_LENGTHS, d, 256) for d in deal]
        ok = ok and all(float(gr[r][0, 0]) == 1e6 * r and float(gr[r][-1, -1]) == gr[r].shape[0] * 16 - 1 + 1e6 * r for r in range(world))
    else:
        ok = ok and gr is None
    ok = ok and mg.shard_pairs(10, 3, 0) == (0, 4) and mg.shard_pairs(10, 3, 1) == (3, 4) and mg.shard_pairs(10, 3, 2) == (6, 4)
    ok = ok and mg.shard_pairs(3, 4, 3) == (2, 0)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {mg.sequence_seed(rank, world)} {','.join(map(str, mine))}")
    dist.barrier()
    dist.destroy_process_group()


def test_pose_gather_and_sharding_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = open(tmp_path / "rank0.txt").read().split()
    r1 = open(tmp_path / "rank1.txt").read().split()
    assert r0[0] == "1" and r1[0] == "1"
    assert (r0[1], r1[1]) == ("100", "101")                      # one synthetic sequence per rank
    assert r0[2] == "0,2,4,6" and r1[2] == "1,3,5,7"             # KITTI 00-07 dealt round-robin


def _worker_world8(rank, world, port, out_dir):
    """BASELINE config #5 at its REAL world size: one KITTI sequence per rank, every rank's full pose list (270 ... 4660
    rows of 16 doubles) to rank 0 in one ragged gather, the slowest rank's time by all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    conftest.entry.load_package()
    import importlib
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")
    deal = mg.deal_sequences(mg.KITTI_LENGTHS, world)
    (s_,) = deal[rank]                                            # exactly one sequence per rank
    n = mg.KITTI_LENGTHS[s_] - 1
    mine = (torch.arange(n * 16, dtype=torch.float64).view(n, 16) + 1e7 * s_)     # row p of sequence s_: 1e7 s_ + 16 p + column
    got = mg.gather_ragged(mine, rank, world, dst=0)
    tmax = mg.max_over_ranks(0.001 * n, torch.device("cpu"), world)               # "busy time" ~ sequence length
    ok = abs(tmax - 4.660) < 1e-12                                                # sequence 02 sets the wall time
    ok = ok and mg.steps_for(mg.KITTI_LENGTHS, deal[rank], 256) == (n + 255) // 256
    if rank == 0:
        ok = ok and [g.shape[0] for g in got] == [mg.KITTI_LENGTHS[d[0]] - 1 for d in deal]
        ok = ok and sorted(g.shape[0] for g in got) == [270, 800, 1100, 1100, 1100, 2760, 4540, 4660]
        for r, g in enumerate(got):
            sq = deal[r][0]
            ok = ok and float(g[0, 0]) == 1e7 * sq and float(g[-1, -1]) == 1e7 * sq + g.shape[0] * 16 - 1
    else:
        ok = ok and got is None
    rig = mg.KITTI_RIGS[s_]
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as f:
        f.write(f"{int(ok)} {s_} {rig['width']}x{rig['height']}")
    dist.barrier()
    dist.destroy_process_group()


def test_config5_control_flow_at_world_size_8(tmp_path):
    """The 8-rank control flow has run once before a node shows up (CPU, gloo): the deal is one sequence per rank,
    longest first (rank 0 = sequence 02), eight DIFFERENT message lengths through the ragged gather, the max
    all-reduce; every rank knows its sequence's frame size.  (Eight ranks SHARING the one GPU of a test box are
    not possible: the box allows six processes on the card.)"""
    world = 8
    mp.spawn(_worker_world8, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    rows = [open(tmp_path / f"rank{r}.txt").read().split() for r in range(world)]
    assert all(r[0] == "1" for r in rows)
    assert [int(r[1]) for r in rows] == [2, 0, 5, 1, 6, 7, 3, 4]                   # longest first, ties by index
    assert [r[2] for r in rows] == ["1241x376", "1241x376", "1226x370", "1241x376", "1226x370", "1226x370", "1242x375", "1226x370"]


def test_deal_sequences_balances_kitti_lengths(pkg):
    """Longest-first greedy deal of the KITTI 00-07 lengths: every sequence exactly once; with 8 ranks it
    is one sequence per rank (the imbalance BASELINE config #5 has by construction: the wall time is
    sequence 02's), with fewer ranks the loads even out."""
    import importlib
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")
    Ls = mg.KITTI_LENGTHS
    for world in (1, 2, 3, 4, 8, 12):
        deal = mg.deal_sequences(Ls, world)
        assert len(deal) == world and sorted(s for d in deal for s in d) == list(range(8))
        loads = [sum(Ls[s] - 1 for s in d) for d in deal]
        if world == 8:
            assert all(len(d) == 1 for d in deal) and max(loads) == 4660
        if world <= 4:
            assert max(loads) <= 1.25 * (sum(loads) / world)         # LPT: within 25 % of perfect balance here
    assert mg.deal_sequences(Ls, 2) == [[2, 1, 6, 7, 4], [0, 5, 3]]
    assert mg.steps_for(Ls, [0], 256) == 18 and mg.steps_for(Ls, [4], 256) == 2 and mg.steps_for(Ls, [], 256) == 0
    x = torch.arange(6.0).view(3, 2)
    assert mg.gather_ragged(x, 0, 1)[0] is x


def test_single_rank_is_identity(pkg):
    import importlib
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")
    p = torch.arange(32, dtype=torch.float64).view(2, 16)
    assert mg.gather_poses(p, 0, 1)[0] is p
    assert mg.max_over_ranks(2.5, torch.device("cpu"), 1) == 2.5
    assert mg.sequence_seed(0, 1) == 20200710
    assert mg.shard_sequences(8, 1, 0) == list(range(8))


import json
import subprocess
import sys

import pytest


@pytest.mark.gpu
def test_bench_config5_two_ranks_rehearsal_on_one_gpu():
    """bench.py --config5 (the KITTI 00-07 lengths dealt to the ranks, unequal step counts, per-rank busy
    time, ragged gather) with two ranks sharing the test box's GPU over gloo; a large batch keeps it short."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(conftest.ROOT, "bench.py"), "--gpus", "2", "--warmup", "1",
           "--batch", "24", "--chunks", "1", "--cpu-pairs", "0", "--dist-backend", "gloo", "--config5", "--no-timing-marks"]
    r = subprocess.run(cmd, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    c5 = out["config"]["config5"]
    assert [p["sequences"] for p in c5["per_rank"]] == [[2, 1, 6, 7, 4], [0, 5, 3]]
    assert [p["steps"] for p in c5["per_rank"]] == [(4660 + 23) // 24 + 3 * ((1100 + 23) // 24) + (270 + 23) // 24,
                                                     (4540 + 23) // 24 + (2760 + 23) // 24 + (800 + 23) // 24]
    assert out["value"] > 0 and c5["imbalance_max_over_mean"] >= 1.0 and all(p["busy_s"] > 0 for p in c5["per_rank"])
    # every sequence's FULL pose list reached rank 0 (16 doubles per pair), not a stand-in
    n_pairs = sum(n - 1 for n in (4541, 1101, 4661, 801, 271, 2761, 1101, 1101))
    assert c5["poses_gathered"] == {"sequences": 8, "pairs": n_pairs, "bytes": n_pairs * 128}


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["sequences", "pairs"])
def test_bench_two_ranks_rehearsal_on_one_gpu(shard):
    """bench.py's N > 1 control flow (double-buffered collection of the previous step's records,
    final drain, barrier, max over ranks, one JSON line from rank 0) with two ranks sharing the one
    GPU of the test box; the collectives run over gloo instead of RCCL (--dist-backend gloo)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(conftest.ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--batch", "6", "--cpu-pairs", "0", "--dist-backend", "gloo", "--shard", shard]
    r = subprocess.run(cmd, capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["pairs_ok_last_step"] == 6 and out["roofline"]["frac"] > 0


def _bench(*flags, env=None, timeout=600):
    cmd = [sys.executable, os.path.join(conftest.ROOT, "bench.py")] + list(flags)
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run(cmd, capture_output=True, timeout=timeout, env=e)


@pytest.mark.gpu
def test_bench_gpus_flag_launches_the_ranks_itself():
    """Plain `python bench.py --gpus 2` (no torch.distributed.run around it, the way the driver calls
    `--gpus 1`): bench.py starts the two ranks itself and rank 0 reports n_gpus == 2.  gloo backend,
    the ranks share the test box's one GPU."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "6", "--cpu-pairs", "0", "--dist-backend", "gloo")
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks"] == 2 and out["steps"] == 2 and out["value"] > 0
    assert out["dist_backend"].startswith("gloo")


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [(), ("--shard", "pairs")])
def test_bench_scaling_table_in_one_command(extra):
    """`python bench.py --gpus 2 --scaling-table`: the N = 1 and N = 2 runs one after the other (each with self-launched
    ranks; gloo, the ranks share the one card), a line per N and the summary line with the curve -- the command that prints
    the 1/2/4/8-GPU batch scaling curve on a node that has the cards."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _bench("--gpus", "2", "--scaling-table", "--steps", "2", "--warmup", "1", "--batch", "6", "--dist-backend", "gloo", *extra)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [json.loads(ln) for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert [ln.get("n_gpus") for ln in lines[:2]] == [1, 2] and "scaling_table" in lines[-1]
    tab = lines[-1]["scaling_table"]
    assert [t["n_gpus"] for t in tab] == [1, 2] and all(t["value"] > 0 for t in tab) and tab[0]["speedup_vs_1"] == 1.0


@pytest.mark.gpu
def test_bench_refuses_more_nccl_ranks_than_gpus():
    """`--gpus N` over nccl (RCCL) on a node with fewer than N cards exits non-zero with a message: a run is
    never reported as N GPUs unless N ranks ran on N cards."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    n = torch.cuda.device_count() + 1
    r = _bench("--gpus", str(n), "--steps", "1", "--warmup", "0", "--batch", "6", "--cpu-pairs", "0", timeout=300)
    assert r.returncode != 0
    assert f"needs {n} GPU(s)" in r.stderr.decode() and not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [["--steps", "3"], ["--config5", "--chunks", "1"], ["--steps", "3", "--shard", "pairs"]])
def test_bench_rccl_calls_on_a_one_rank_communicator(extra):
    """The `nccl` (= RCCL) branch on the one GPU a test box has: `--dist-single` initialises a process group of ONE
    rank and every collective of the N > 1 path (barrier, gather of poses / relative motions, the ragged gather of
    config #5's full pose lists, the max all-reduce) goes through RCCL on device tensors.  Not a scaling measurement --
    it is the only execution of that branch this pipeline allows, and it catches API misuse the gloo runs cannot."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    r = _bench("--gpus", "1", "--dist-single", "--warmup", "1", "--batch", "64", "--cpu-pairs", "0", "--e2e-frames", "0",
               "--no-secondary", *extra, timeout=600)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = json.loads([ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and "nccl (RCCL), one-rank communicator" in out["dist_backend"] and out["value"] > 0
    if "--config5" in extra:
        n_pairs = sum(n - 1 for n in (4541, 1101, 4661, 801, 271, 2761, 1101, 1101))
        assert out["config"]["config5"]["poses_gathered"] == {"sequences": 8, "pairs": n_pairs, "bytes": n_pairs * 128}


def test_bench_gpus_flag_is_checked_without_a_gpu():
    """CPU box: the launcher still starts N ranks, every rank finds no card and exits non-zero, the parent
    returns that code (and prints no JSON line); a WORLD_SIZE that contradicts --gpus is refused too;
    --config5 with --shard pairs (unequal step counts around a per-step collective) is an argument error."""
    r = _bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-pairs", "0", "--dist-backend", "gloo", timeout=300)
    if r.returncode == 0:
        pytest.skip("this box has a GPU: covered by the gpu tests")
    err = r.stderr.decode()
    assert "this node has 0" in err and "stopping the other ranks" in err or "exited with" in err
    assert not [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    r = _bench("--gpus", "2", "--steps", "1", env={"RANK": "0", "WORLD_SIZE": "1"}, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE is 1" in r.stderr.decode()
    r = _bench("--gpus", "2", "--config5", "--shard", "pairs", timeout=60)
    assert r.returncode == 2 and "cannot be combined" in r.stderr.decode()


def test_shard_pairs_partitions_every_pair_exactly_once():
    """Property (hypothesis): for any sequence length and rank count the chunks tile the frame pairs
    in order, neighbouring chunks share exactly one (halo) frame, and sizes differ by at most one."""
    import importlib
    from hypothesis import given, settings, strategies as st
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")

    @settings(max_examples=300, deadline=None)
    @given(st.integers(min_value=2, max_value=5000), st.integers(min_value=1, max_value=16))
    def check(n_frames, world):
        nxt, sizes = 0, []
        for rank in range(world):
            first, nf = mg.shard_pairs(n_frames, world, rank)
            if nf == 0:
                continue
            assert first == nxt and nf >= 2            # starts on the previous chunk's last frame
            nxt = first + nf - 1
            sizes.append(nf - 1)
        assert nxt == n_frames - 1 and sum(sizes) == n_frames - 1
        assert max(sizes) - min(sizes) <= 1 or len(sizes) < world
    check()

    @settings(max_examples=100, deadline=None)
    @given(st.integers(min_value=1, max_value=40), st.integers(min_value=0, max_value=2 ** 31 - 1))
    def chain(n, seed):
        rng = np.random.default_rng(seed)
        T = np.tile(np.eye(4), (n, 1, 1))
        T[:, :3, 3] = rng.normal(size=(n, 3))
        ok = rng.integers(0, 2, n).astype(np.int32)
        got = mg.chain_relative(torch.from_numpy(T.reshape(n, 16).copy()), torch.from_numpy(ok)).numpy()
        acc = np.cumsum(T[:, :3, 3] * ok[:, None], 0)          # pure translations: the product is a running sum
        assert np.allclose(got[:, :3, 3], acc, atol=1e-12) and np.allclose(got[:, :3, :3], np.eye(3))
    chain()
