"""Data-parallel sharding of the mean-teacher step: one process per GPU, RCCL over xGMI.

The reference is single-process, so this is new capability; what it has to preserve
is the reference's batch contract.  main.py builds every batch as [weak | unlabeled | strong] with
sizes [B/4, B/2, B/4] (main.py:238-247) and addresses the labelled parts with positional slices
``weak_mask = slice(B/4)``, ``strong_mask = slice(3B/4, B)``.  Each rank therefore takes its share
OF EACH STREAM (not a contiguous chunk of the concatenated batch), so that

* the same positional masks stay valid on the local batch, and
* every mean-reduced loss term is a mean over equally many elements on every rank, hence the
  average of the per-rank gradients equals the gradient of the global-batch loss
  (BatchNorm statistics are per rank, as in standard DDP).

The only collective is ONE sum all-reduce of the flat fp32 gradient buffer per step (857 KB),
issued as two buckets - the GRU/heads tail (ready first) and the conv head - so the first overlaps
the conv-block backward.  ``torch.distributed`` backend "nccl" is RCCL on ROCm; everything here is
backend-agnostic and is covered on CPU with world_size-2 gloo tests.
"""
import torch
import torch.distributed as dist


def local_batch_sizes(batch_sizes, world):
    """Per-rank stream sizes for global stream sizes (main.py:240 ``batch_sizes``)."""
    for b in batch_sizes:
        if b % world != 0:
            raise ValueError(f"every stream size must be divisible by world size {world}: {list(batch_sizes)}")
    return [b // world for b in batch_sizes]


def local_masks(batch_sizes, world):
    """(weak_mask, strong_mask) valid on a rank's local batch (main.py:241,247)."""
    lb = local_batch_sizes(batch_sizes, world)
    total = sum(lb)
    weak = slice(lb[0])
    strong = slice(total - lb[-1], total) if len(lb) == 3 else None
    return weak, strong


def shard_indices(batch_indices, batch_sizes, rank, world):
    """Rank's share of ONE global batch of dataset indices laid out stream after stream
    (what MultiStreamBatchSampler yields, DataLoad.py:562-571)."""
    lb = local_batch_sizes(batch_sizes, world)
    out, off = [], 0
    for b, l in zip(batch_sizes, lb):
        out.extend(batch_indices[off + rank * l: off + (rank + 1) * l])
        off += b
    return out


def shard_batch(tensors, batch_sizes, rank, world):
    """Rank's share of already-collated global tensors (each [B_global, ...])."""
    idx = shard_indices(list(range(sum(batch_sizes))), batch_sizes, rank, world)
    idx_t = torch.as_tensor(idx)
    return [t[idx_t.to(t.device)] for t in tensors]


def grad_buckets(layout):
    """Two contiguous buckets of the flat gradient buffer in the order backward produces them:
    (tail = GRU + heads, head = conv blocks).  ``layout``: [(start, end, shape)] per parameter in
    named_parameters() order; the first 18 tensors are the cnn (models/CRNN.py:12-31)."""
    cnn_end = layout[17][1]
    total = layout[-1][1]
    return (cnn_end, total), (0, cnn_end)


def allreduce_bucket(flat, lo, hi, group=None, async_op=False, force=False):
    """Sum all-reduce of flat[lo:hi] in place; returns the work handle when async_op.  A one-rank group is a no-op
    unless ``force`` (path test of the collective on a single-GPU box)."""
    if group is None and not dist.is_initialized():
        return None
    if dist.get_world_size(group) == 1 and not force:
        return None
    return dist.all_reduce(flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def broadcast_parameters(flat_tensors, group=None, src=0):
    """Make every rank start from rank ``src``'s parameters / optimiser state."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in flat_tensors:
        dist.broadcast(t, src=src, group=group)


_M64 = 2 ** 64 - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def fold_rank_seed(seed, rank):
    """Philox base seed of data-parallel rank ``rank``.  Rank 0 keeps the user's seed (a one-process run and rank 0 of a
    data-parallel run draw the same stream); every other rank gets a HASH of (seed, rank).  The per-step keys are
    base + k * 0x9E37...15 + 1 (student k = 2 * step, teacher k = 2 * step + 1, csrc/optim.hip), so the rank must not
    enter as a multiple of that same constant: ``seed + rank * 0x9E37...15`` made rank 1's student masks equal rank 0's
    teacher masks of the same step and rank 2's stream rank 0's shifted by one step."""
    seed &= _M64
    if rank == 0:
        return seed
    return _splitmix64(seed ^ _splitmix64((rank + 0xD1B54A32D192ED03) & _M64))
