#!/bin/bash
# GPU box: a few PMC passes on bench.py's workload (each pass separate; counters only, no tracing domains)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu --skip-dense --streams 1 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ssqp_solve" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print("%-28s n=%d mean=%.4g" % (k,len(v),sum(v)/len(v)))
PY
