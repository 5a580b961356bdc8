"""Sharding of a batch of independent QPs over ranks (one process per GPU) and the final gather.

The path partitions by problem: rank r owns the contiguous block
[r*ceil(P/G), (r+1)*ceil(P/G)) and solves it with no data-path collective; the only exchange is ONE
all-gather of the packed results (z, S, status) at the end (RCCL over xGMI when the backend is "nccl", gloo on CPU in
tests).  The three result arrays of a DeviceBatch are views of one byte buffer (`DeviceBatch.out`), so the send side
needs no packing copy, and the receive buffer is allocated once (`PackedGather`) -- nothing is allocated per step.
"""
import torch
import torch.distributed as dist


def shard_range(nprob, world, rank):
    """Contiguous block of problem ids owned by `rank` (possibly empty for trailing ranks)."""
    per = -(-nprob // world)
    lo = min(nprob, rank * per)
    hi = min(nprob, lo + per)
    return lo, hi


def packed_layout(P, N, J):
    """byte offsets of (z, S, status) inside one rank's packed result buffer, and its size (all 8-byte aligned)"""
    oz = 0
    oS = oz + P * N * 8
    ost = oS + -(-(P * (N + J) * 4) // 8) * 8
    return oz, oS, ost, ost + P * 8


def packed_views(buf, P, N, J):
    """(z (P,N) f64, S (P,N+J) i32, status (P,) i64) as views of a packed byte buffer of packed_layout's size"""
    oz, oS, ost, total = packed_layout(P, N, J)
    z = buf[oz:oz + P * N * 8].view(torch.float64).view(P, N)
    S = buf[oS:oS + P * (N + J) * 4].view(torch.int32).view(P, N + J)
    st = buf[ost:ost + P * 8].view(torch.int64)
    return z, S, st


class PackedGather:
    """One pre-allocated receive buffer and ONE collective per step: all_gather_into_tensor of every rank's packed
    (z, S, status).  `gather(send)` returns nothing; `results()` gives views (world*P rows) of the receive buffer."""

    def __init__(self, P, N, J, device, world=None):
        self.P, self.N, self.J = P, N, J
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bytes = packed_layout(P, N, J)[3]
        self.recv = torch.empty(self.world * self.bytes, dtype=torch.uint8, device=device)
        self.calls = 0

    def gather(self, send):
        assert send.numel() == self.bytes and send.dtype == torch.uint8
        if self.world == 1:
            self.recv.copy_(send)
        else:
            dist.all_gather_into_tensor(self.recv, send)
        self.calls += 1

    def results(self):
        """(z, S, status) of the whole batch: rank r's block at rows [r*P, (r+1)*P) (copies, for checking)"""
        parts = [packed_views(self.recv[r * self.bytes:(r + 1) * self.bytes], self.P, self.N, self.J)
                 for r in range(self.world)]
        return tuple(torch.cat([p[k] for p in parts]) for k in range(3))


def gather_results(z, S, status, nprob_total=None):
    """All-gather of per-rank results held in three separate tensors (every rank passes tensors of the SAME leading
    size; pad the last shard if needed): packs them into one buffer and issues ONE collective.  Returns (z, S, status)
    of the whole batch, trimmed to nprob_total rows.  Convenience form (allocates); steady-state callers keep a
    PackedGather and a DeviceBatch, whose outputs are already packed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        out = (z, S, status)
    else:
        P, N = z.shape
        J = S.shape[1] - N
        pg = PackedGather(P, N, J, z.device)
        send = torch.empty(pg.bytes, dtype=torch.uint8, device=z.device)
        vz, vS, vst = packed_views(send, P, N, J)
        vz.copy_(z)
        vS.copy_(S)
        vst.copy_(status)
        pg.gather(send)
        out = pg.results()
    if nprob_total is not None:
        out = tuple(t[:nprob_total] for t in out)
    return out
