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


def _worker_real(rank, world, port, nprob, out):
    """every rank solves ITS shard (the CPU oracle stands in for the GPU solve: no GPU in this test) and the results
    are gathered exactly as bench.py does after DeviceBatch.solve()"""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = pkg.GenConfig(24, 1, 2, 48, 1e-3, 0.2, 1.05, 0.1)
    per = -(-nprob // world)
    lo, hi = pkg.dist.shard_range(nprob, world, rank)
    prob = pkg.generate_batch(cfg, per, pkg.BASE_SEED + rank * per)     # rank r owns seeds [r*per, (r+1)*per)
    x0, S0, st = pkg.phase1_batch(prob)
    z, S, status, _, _ = orc.solveQP_warm_batch(*[prob[k] for k in "VAGqbgdu"], S0, x0, nthreads=1)
    gz, gS, gst = pkg.dist.gather_results(torch.from_numpy(z), torch.from_numpy(S), torch.from_numpy(status),
                                          nprob_total=nprob)
    if rank == 0:
        np.savez(out, z=gz.numpy(), S=gS.numpy(), status=gst.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_solve_and_gather_equals_unsharded(tmp_path):
    import socket
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from oracle import oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    nprob = 7
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker_real, args=(2, port, nprob, out), nprocs=2, join=True)
    r = np.load(out)
    cfg = pkg.GenConfig(24, 1, 2, 48, 1e-3, 0.2, 1.05, 0.1)
    prob = pkg.generate_batch(cfg, 8, pkg.BASE_SEED)                     # the same seeds, one process
    x0, S0, st = pkg.phase1_batch(prob)
    z, S, status, _, _ = orc.solveQP_warm_batch(*[prob[k] for k in "VAGqbgdu"], S0, x0, nthreads=1)
    assert np.array_equal(r["status"], status[:nprob]) and np.array_equal(r["S"], S[:nprob])
    assert np.array_equal(r["z"], z[:nprob])


def _worker_packed(rank, world, port, steps, out):
    """the steady-state form bench.py uses: outputs packed in one buffer, ONE pre-allocated collective per step"""
    import sys
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    pkg = ge.load_package()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P, N, J = 5, 6, 3
    total = pkg.dist.packed_layout(P, N, J)[3]
    send = torch.zeros(total, dtype=torch.uint8)
    z, S, st = pkg.dist.packed_views(send, P, N, J)
    pg = pkg.dist.PackedGather(P, N, J, "cpu")
    recv_ptr = pg.recv.data_ptr()
    ncalls = [0]
    real = dist.all_gather_into_tensor

    def counting(*a, **k):
        ncalls[0] += 1
        return real(*a, **k)
    dist.all_gather_into_tensor = counting
    for step in range(steps):
        z.copy_(torch.arange(P * N, dtype=torch.float64).view(P, N) + 1000.0 * rank + 0.5 * step)
        S.copy_(torch.full((P, N + J), rank * 10 + step, dtype=torch.int32))
        st.copy_(torch.arange(P, dtype=torch.int64) + 100 * rank + step)
        pg.gather(send)
    dist.all_gather_into_tensor = real
    assert ncalls[0] == steps and pg.calls == steps and pg.recv.data_ptr() == recv_ptr   # one collective per step, same buffer
    gz, gS, gst = pg.results()
    if rank == 0:
        np.savez(out, z=gz.numpy(), S=gS.numpy(), status=gst.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_packed_gather_one_collective_per_step(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "p.npz")
    steps = 3
    mp.spawn(_worker_packed, args=(2, port, steps, out), nprocs=2, join=True)
    r = np.load(out)
    P, N, J = 5, 6, 3
    assert r["z"].shape == (2 * P, N) and r["S"].shape == (2 * P, N + J) and r["status"].shape == (2 * P,)
    for rank in range(2):
        blk = slice(rank * P, (rank + 1) * P)
        assert np.array_equal(r["z"][blk], np.arange(P * N).reshape(P, N) + 1000.0 * rank + 0.5 * (steps - 1))
        assert (r["S"][blk] == rank * 10 + steps - 1).all()
        assert np.array_equal(r["status"][blk], np.arange(P) + 100 * rank + steps - 1)
