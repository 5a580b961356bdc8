#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats + separate PMC passes (HBM traffic) of bench.py's workload,
# once for the default formulation and once for the dense one (SSQP_DENSE_GAMMA=1).
# usage: tools/rocprof_bench.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in default dense; do
  if [ $mode = dense ]; then export SSQP_DENSE_GAMMA=1; else export SSQP_DENSE_GAMMA=0; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --skip-dense --streams 1 "$@" > $OUT/bench_trace_$mode.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --skip-dense --streams 1 "$@" > $OUT/bench_fetch_$mode.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$mode -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --skip-dense --streams 1 "$@" > $OUT/bench_write_$mode.log 2>&1 || exit 1
done
find $OUT -name "*kernel_stats.csv"
