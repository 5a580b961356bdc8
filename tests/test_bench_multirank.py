"""bench.py's own multi-rank control flow: `--gpus N` starts the ranks itself; a launcher/argument mismatch is an
error.  The GPU test runs two ranks on ONE card (rehearsal mode: gloo instead of RCCL, both ranks on device 0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["SSQP_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--nprob", "64", "--no-cpu", "--skip-dense", "--streams", "2"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["all_converged"] and d["value"] > 0 and d["pipeline"]["batches_distinct"]
    # ONE collective per timed step, issued inside the timed region, and nothing allocated there (pre-allocated
    # receive buffers; the send side is the batch's packed output buffer)
    assert d["collective"]["gathers"] == 2 and d["collective"]["per_step"] == 1
    # (the rehearsal runs on gloo, which stages device tensors through temporaries of its own inside the call)
    assert (d["collective"]["cuda_allocations_in_timed_region"] ==
            d["collective"]["of_which_inside_the_backend_collective_call"]), d["collective"]
