#!/bin/bash
# GPU box: PMC passes (each pass separate; counters only, no tracing domains beyond --kernel-trace) over
#   python3 tools/debug_parity.py <args...>      (one warm launch + three timed ones per pass)
# usage: tools/pmc_probe.sh <tag> "<debug_parity args>" <counter group> [<counter group> ...]
set -o pipefail
TAG=$1; ARGS=$2; shift; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/debug_parity.py $ARGS > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn=r["Kernel_Name"]
        if "ssqp_solve" in kn or "ssqp_wave" in kn:
            acc[kn.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    print("==", kn)
    for k,v in sorted(d.items()):
        print("  %-28s n=%d mean=%.4g" % (k,len(v),sum(v)/len(v)))
PY
