#!/bin/bash
# GPU box: parity suite, then the cfg4 and cfg3 bench lines (value, kernel ms, converged)
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -x > gpurun_out/qb_tests.log 2>&1; rc=$?; echo tests_exit=$rc; tail -2 gpurun_out/qb_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc   # a failed or faulted GPU step: nothing else runs on this box
for cfg in cfg4 cfg4 cfg4 cfg3; do
  timeout -k 10 200 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu --skip-dense --streams 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', round(d['value'],1), round(d['roofline']['kernel_ms'],3), d['all_converged'])"
done
