"""Batch sharding across GPUs (one process per GPU) and the path's only exchange.

Images are independent: image i of a global batch lives on rank i // per_rank and never
leaves that GPU's HBM.  The only collective is the all-reduce of the batch-wide content min/max boost
(2 floats; RCCL when the tensors are on GPU, gloo in the CPU tests).  The reference has no counterpart
for this statistic (it writes constants, ultrahdr.cpp:250-257)."""
import torch


def shard_range(n_images, rank, world):
    """contiguous [lo, hi) slice of a global batch owned by `rank`; sizes differ by at most one"""
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def image_seed(global_index, base_seed=1234):
    """LCG seed of global image `global_index` (seed = 1234 + image_index)"""
    return base_seed + global_index


def reduce_content_minmax(per_image_minmax, dist=None, scratch=None, async_op=False):
    """per_image_minmax: float32 tensor [2*n] = (min_0, max_0, min_1, ...) of this rank's images.
    Returns a 2-element tensor (global min, global max).  One all-reduce(MIN) on (min, -max).
    With async_op=True returns (scratch, work): the collective overlaps whatever the caller enqueues next;
    finish with finish_content_minmax(scratch, work)."""
    mm = per_image_minmax.view(-1, 2)
    red = scratch if scratch is not None else torch.empty(2, dtype=torch.float32, device=per_image_minmax.device)
    if mm.shape[0] == 0:
        red[0] = float("inf")
        red[1] = float("inf")
    else:
        red[0] = mm[:, 0].min()
        red[1] = -mm[:, 1].max()
    work = None
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        work = dist.all_reduce(red, op=dist.ReduceOp.MIN, async_op=async_op)
    if async_op:
        return red, work
    return finish_content_minmax(red, None)


def finish_content_minmax(red, work):
    if work is not None:
        work.wait()
    out = red.clone()
    out[1] = -out[1]
    return out
