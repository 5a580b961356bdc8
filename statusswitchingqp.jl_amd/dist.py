"""Sharding of a batch of independent QPs over ranks (one process per GPU) and the final gather.

The path partitions by problem: rank r owns the contiguous block
[r*ceil(P/G), (r+1)*ceil(P/G)) and solves it with no data-path collective; the only exchange is one
all-gather of (z, S, status) at the end (RCCL over xGMI when the backend is "nccl", gloo on CPU in tests).
"""
import torch
import torch.distributed as dist


def shard_range(nprob, world, rank):
    """Contiguous block of problem ids owned by `rank` (possibly empty for trailing ranks)."""
    per = -(-nprob // world)
    lo = min(nprob, rank * per)
    hi = min(nprob, lo + per)
    return lo, hi


def gather_results(z, S, status, nprob_total=None):
    """All-gather of the per-rank results.  Every rank passes tensors of the SAME leading size (pad the last
    shard if needed); returns (z, S, status) of the whole batch, trimmed to nprob_total rows."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = (z, S, status)
    else:
        world = dist.get_world_size()
        outs = []
        for t in (z, S, status):
            t = t.contiguous()
            full = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(full, t)
            outs.append(full)
        out = tuple(outs)
    if nprob_total is not None:
        out = tuple(t[:nprob_total] for t in out)
    return out
