"""Sequence-per-GPU sharding helpers: one process per GPU over torch.distributed
(backend "nccl" == RCCL on ROCm; "gloo" in the CPU tests).  The hot path has NO data-path collective:
independent stereo sequences are dealt to ranks, every rank tracks its own sequences, and only the
poses (16 doubles per frame pair) are gathered to rank 0.  The collective is latency-bound
(<= 0.6 MB per rank for the longest KITTI sequence), so nothing here is tuned for xGMI bandwidth."""
import torch
import torch.distributed as dist

# KITTI odometry sequence lengths 00..07 (public benchmark numbers; BASELINE config #5)
KITTI_LENGTHS = (4541, 1101, 4661, 801, 271, 2761, 1101, 1101)


def sequence_seed(rank, world):
    """Seed of the synthetic sequence a rank renders (single GPU: S0's seed)."""
    return 20200710 if world == 1 else 100 + rank


def shard_sequences(n_sequences, world, rank):
    """Round-robin deal of sequence indices to ranks: sequence s -> rank s % world."""
    return [s for s in range(n_sequences) if s % world == rank]


def gather_poses(poses, rank, world, dst=0):
    """poses: (n, 16) float64 tensor of this rank.  Returns the list of all ranks' tensors on dst,
    None elsewhere.  One gather per call; with world == 1 it is the identity."""
    if world == 1:
        return [poses]
    buf = [torch.empty_like(poses) for _ in range(world)] if rank == dst else None
    dist.gather(poses, buf, dst=dst)
    return buf


def max_over_ranks(seconds, device, world):
    """The slowest rank's wall time (the bench contract: MAX over ranks)."""
    if world == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def poses_view(results_u8, pose_offset, n):
    """(n, 16) float64 view-copy of the `pose` field of n packed svo_step_result records."""
    return results_u8[:n, pose_offset:pose_offset + 128].contiguous().view(torch.float64).view(n, 16)
