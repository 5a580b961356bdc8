import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    # always the incremental make (a no-op when up to date): the tests never validate a stale binary
    ge.build()
    p = ge.load_package()
    # ... and say so: the binary names the sources it was built from (csrc/Makefile: HASHED)
    import bench
    ver = p._capi.lib().ssqp_version().decode()
    assert ver.endswith("src=" + bench.kernel_source_hash()), "libssqp_hip.so is stale: %s" % ver
    return p


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


def colmajor(block, rows):
    """(N, rows) column-major block -> (rows, N) matrix."""
    return np.ascontiguousarray(block).reshape(-1, rows).T if rows else np.zeros((0, block.shape[0]))


def oracle_batch(orc, prob, S0, x0, settings=None, nthreads=0):
    return orc.solveQP_warm_batch(prob["V"], prob["A"], prob["G"], prob["q"], prob["b"], prob["g"], prob["d"],
                                  prob["u"], S0, x0, settings=settings, nthreads=nthreads)


def assert_parity(res_z, res_S, res_status, zo, So, sto, rtol=1e-10):
    """status vectors bit-exact, iteration counts identical, z within rtol relative (inf-norm per problem)."""
    assert np.array_equal(res_status, sto), (res_status[:8], sto[:8])
    assert np.array_equal(res_S, So), "S differs at %s" % (np.argwhere(res_S != So)[:5],)
    scale = np.maximum(np.abs(zo).max(axis=-1), 1e-300)
    rel = (np.abs(res_z - zo).max(axis=-1) / scale).max()
    assert rel < rtol, rel
    return rel
