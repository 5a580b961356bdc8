#!/bin/bash
# GPU box: PMC passes over the Phase-1 kernel alone (python3 tools/run_phase1.py <cfg> <nprob> 3): issue / wait / instruction
# mix and the instruction cache.   usage: tools/pmc_phase1.sh <tag> <cfg> <nprob>
set -o pipefail
TAG=$1; CFG=$2; NP=$3
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_p1_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $GRAFT_REPO_ROOT/tools/run_phase1.py $CFG $NP 3 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn=r["Kernel_Name"]
        if "phase1" in kn:
            acc[kn.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kn,d in acc.items():
    print("==", kn)
    for k,v in sorted(d.items()):
        print("  %-28s n=%d mean=%.5g" % (k,len(v),sum(v)/len(v)))
PY
