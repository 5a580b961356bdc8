"""-m "not gpu": the multi-rank path (shard by problem, one all-gather at the end) on CPU with gloo, world_size 2."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, nprob, N, J, out):
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = pkg.dist.shard_range(nprob, world, rank)
    per = -(-nprob // world)
    # stand-in for the per-rank solve: results that encode the global problem id
    ids = torch.arange(lo, lo + per)
    z = ids[:, None].double() + torch.arange(N)[None, :].double() / 1000
    S = (ids[:, None] % 5).int().repeat(1, N + J)
    status = ids.long() + 1
    gz, gS, gst = pkg.dist.gather_results(z, S, status, nprob_total=nprob)
    if rank == 0:
        np.savez(out, z=gz.numpy(), S=gS.numpy(), status=gst.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nprob", [8, 7])
def test_shard_and_gather_world2(tmp_path, nprob):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    N, J = 6, 2
    out = str(tmp_path / "g.npz")
    mp.spawn(_worker, args=(2, port, nprob, N, J, out), nprocs=2, join=True)
    r = np.load(out)
    assert r["status"].tolist() == list(range(1, nprob + 1))
    assert np.allclose(r["z"][:, 0], np.arange(nprob)) and r["S"].shape == (nprob, N + J)


def test_shard_range_partitions_everything():
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    for nprob in (0, 1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = pkg.dist.shard_range(nprob, world, r)
                assert 0 <= lo <= hi <= nprob
                got += list(range(lo, hi))
            assert got == list(range(nprob))
