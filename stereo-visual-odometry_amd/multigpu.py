"""Sequence-per-GPU sharding helpers: one process per GPU over torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).  The hot path has NO data-path collective:
independent stereo sequences are dealt to ranks, every rank tracks its own sequences, and only the
poses (16 doubles per frame pair) are gathered to rank 0.  The collective is latency-bound
(<= 0.6 MB per rank for the longest KITTI sequence), so nothing here is tuned for xGMI bandwidth."""
import torch
import torch.distributed as dist

# KITTI odometry sequence lengths 00..07 (public benchmark numbers; BASELINE config #5)
KITTI_LENGTHS = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101)

# Frame size and rectified calibration of the same sequences (public KITTI odometry calib.txt values; SURVEY.md 8(d)
# config #5: "real 03 is 1242x375 and 04-07 1226x370 with different intrinsics; one YAML per sequence").  The reference
# reads whatever size is on disk (src/System.cpp:75-104) and takes the rig from its YAML (src/parameter.cpp:18-45):
# 00-02 are the values config/default.yaml ships.
KITTI_RIG_A = dict(width=1241, height=376, fx=718.856, fy=718.856, cx=607.193, cy=185.216, baseline=0.537)       # 00-02
KITTI_RIG_B = dict(width=1242, height=375, fx=721.5377, fy=721.5377, cx=609.5593, cy=172.854, baseline=0.53715)  # 03
KITTI_RIG_C = dict(width=1226, height=370, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104, baseline=0.53715)  # 04-12
KITTI_RIGS = (KITTI_RIG_A, KITTI_RIG_A, KITTI_RIG_A, KITTI_RIG_B, KITTI_RIG_C, KITTI_RIG_C, KITTI_RIG_C, KITTI_RIG_C)


def sequence_seed(rank, world):
    """Seed of the synthetic sequence a rank renders (single GPU: S0's seed)."""
    return 20200710 if world == 1 else 100 + rank


def shard_sequences(n_sequences, world, rank):
    """Round-robin deal of sequence indices to ranks: sequence s -> rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


def deal_sequences(lengths, world):
    """Longest-processing-time-first deal of sequences to ranks (BASELINE config #5: 8 KITTI sequences of
    271..4661 frames on 8 GPUs are imbalanced by construction; with fewer ranks the greedy deal
    evens the load out).  Returns a list of `world` lists of sequence indices; deterministic (ties:
    lower sequence index first, lower rank first)."""
    order = sorted(range(len(lengths)), key=lambda s: (-lengths[s], s))
    load = [0] * world
    mine = [[] for _ in range(world)]
    for s in order:
        r = min(range(world), key=lambda q: (load[q], q))
        mine[r].append(s)
        load[r] += max(lengths[s] - 1, 0)
    return mine


def steps_for(lengths, seqs, batch):
    """Batched steps a rank needs for its sequences: ceil((len - 1) / batch) launches per sequence."""
    return sum((max(lengths[s] - 1, 0) + batch - 1) // batch for s in seqs)


def _solo(world):
    """One rank and no process group: the collectives are identities.  (One rank WITH a group -- `bench.py
    --dist-single`, the one-GPU rehearsal of the RCCL calls -- goes through torch.distributed like any other size.)"""
    return world == 1 and not (dist.is_available() and dist.is_initialized())


def gather_ragged(x, rank, world, dst=0):
    """Gathers per-rank tensors of DIFFERENT leading length (n_r, k) to dst: lengths are exchanged
    first, messages are padded to the longest, the padding is cut off again on dst.  Returns the
    list of world tensors on dst, None elsewhere."""
    if _solo(world):
        return [x]
    n = torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    nmax = max(int(v.item()) for v in lens)
    msg = torch.zeros((nmax,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    msg[:x.shape[0]] = x
    buf = [torch.empty_like(msg) for _ in range(world)] if rank == dst else None
    dist.gather(msg, buf, dst=dst)
    if rank != dst:
        return None
    return [buf[r][:int(lens[r].item())] for r in range(world)]


def shard_pairs(n_frames, world, rank):
    """Frame-pair-granular sharding of ONE sequence (SURVEY.md 8e granularity 2 / 8f rank 1): the
    n_frames - 1 consecutive pairs are cut into `world` contiguous chunks, chunk c -> rank c, with
    a one-frame halo (a chunk needs the frame before its first pair).  Returns (first_frame,
    n_chunk_frames): the rank tracks frames first_frame .. first_frame + n_chunk_frames - 1."""
    n_pairs = n_frames - 1
    base, extra = divmod(n_pairs, world)
    first = rank * base + min(rank, extra)
    mine = base + (1 if rank < extra else 0)
    return first, (mine + 1 if mine > 0 else 0)


def chain_relative(T_rel_inv, ok, pose0=None):
    """Prefix product of per-pair inverse relative motions, skipping failed pairs -- the serial part
    of Tracking's `frame_pose_ = frame_pose_ * T.inv()` (reference src/tracking.cpp:318) done once
    for chunks that were tracked independently.  T_rel_inv: (n, 16) or (n, 4, 4) float64, ok: (n,)
    integer flags.  Returns (n, 4, 4): the pose after each pair (same association order as the
    reference: left to right)."""
    T = T_rel_inv.reshape(-1, 4, 4).to(torch.float64)
    P = torch.eye(4, dtype=torch.float64, device=T.device) if pose0 is None else pose0.reshape(4, 4).to(T)
    out = torch.empty_like(T)
    for i in range(T.shape[0]):
        if int(ok[i]):
            P = P @ T[i]
        out[i] = P
    return out


def gather_relative(T_rel_inv, ok, rank, world, dst=0):
    """Gathers every rank's (n, 16) relative motions + (n,) ok flags to dst as ONE (n, 17) float64
    message per rank (chunks padded to the longest), returns (T (N, 16), ok (N,)) in sequence order
    on dst, None elsewhere.  `lengths` need not match across ranks."""
    n = torch.tensor([T_rel_inv.shape[0]], dtype=torch.int64, device=T_rel_inv.device)
    if _solo(world):
        return T_rel_inv.reshape(-1, 16), ok
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    nmax = int(max(int(x.item()) for x in lens))
    msg = torch.zeros((nmax, 17), dtype=torch.float64, device=T_rel_inv.device)
    msg[:T_rel_inv.shape[0], :16] = T_rel_inv.reshape(-1, 16)
    msg[:T_rel_inv.shape[0], 16] = ok.to(torch.float64)
    buf = [torch.empty_like(msg) for _ in range(world)] if rank == dst else None
    dist.gather(msg, buf, dst=dst)
    if rank != dst:
        return None
    parts = [buf[r][:int(lens[r].item())] for r in range(world)]
    allm = torch.cat(parts, 0)
    return allm[:, :16].contiguous(), allm[:, 16].to(torch.int64)


def gather_poses(poses, rank, world, dst=0):
    """poses: (n, 16) float64 tensor of this rank.  Returns the list of all ranks' tensors on dst,
    None elsewhere.  One gather per call; with world == 1 it is the identity."""
    if _solo(world):
        return [poses]
    buf = [torch.empty_like(poses) for _ in range(world)] if rank == dst else None
    dist.gather(poses, buf, dst=dst)
    return buf


def max_over_ranks(seconds, device, world):
    """The slowest rank's wall time (the bench contract: MAX over ranks)."""
    if _solo(world):
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def field_view(results_u8, offset, n, n_doubles):
    """(n, n_doubles) float64 copy of a double field of n packed svo_step_result records."""
    return results_u8[:n, offset:offset + 8 * n_doubles].contiguous().view(torch.float64).view(n, n_doubles)


def int_field(results_u8, offset, n):
    """(n,) int32 copy of an int field of n packed svo_step_result records."""
    return results_u8[:n, offset:offset + 4].contiguous().view(torch.int32).view(n)


def poses_view(results_u8, pose_offset, n):
    """(n, 16) float64 view-copy of the `pose` field of n packed svo_step_result records."""
    return results_u8[:n, pose_offset:pose_offset + 128].contiguous().view(torch.float64).view(n, 16)
