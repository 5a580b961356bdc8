#!/bin/bash
# GPU box: the round's rocprofv3 evidence for bench.py's workload (cfg4, 1024 QPs, serial launches):
#   kernel-trace --stats for the default and the dense formulation, then SEPARATE --pmc passes (counters only):
#   FETCH_SIZE, WRITE_SIZE (HBM bytes), two SQ groups (issue / wait / instruction mix).
# usage: tools/profile_round.sh <tag>        -> gpurun_out/prof_<tag>/ (+ pmc_counters.json, ready for profiles/)
set -o pipefail
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --skip-dense --streams 1 --mode serial"
run() { name=$1; shift; timeout -k 10 400 "$@" > $OUT/$name.log 2>&1 || { echo "FAILED $name"; tail -5 $OUT/$name.log; exit 1; }; }
run trace_default rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- $B --steps 5 --warmup 1
run trace_dense   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_dense   -- $B --steps 3 --warmup 1 --dense
run trace_lanes   rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lanes   -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --skip-dense --streams 3 --mode lanes --steps 20 --warmup 3
run fetch_default rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_default -- $B --steps 1 --warmup 0
run write_default rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_default -- $B --steps 1 --warmup 0
run fetch_dense   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_dense -- $B --steps 1 --warmup 0 --dense
run write_dense   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_dense -- $B --steps 1 --warmup 0 --dense
run sq1_default   rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1_default -- $B --steps 1 --warmup 0
run sq2_default   rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_default -- $B --steps 1 --warmup 0
run sq3_default   rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq3_default -- $B --steps 1 --warmup 0
# the eight-per-CU build of the timed region (launch lanes); counter collection serialises the launches
L="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --skip-dense --streams 3 --mode lanes --steps 3 --warmup 0"
run fetch_lanes   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_lanes -- $L
run write_lanes   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_lanes -- $L
run sq1_lanes     rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1_lanes -- $L
# cfg3: the big-factor build of the wavefront kernel (kernel trace + HBM bytes + issue counters)
C3="python3 $GRAFT_REPO_ROOT/bench.py --config cfg3 --no-cpu --skip-dense --streams 1 --mode serial"
run trace_cfg3    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_cfg3 -- $C3 --steps 5 --warmup 1
run fetch_cfg3    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_cfg3 -- $C3 --steps 1 --warmup 0
run write_cfg3    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_cfg3 -- $C3 --steps 1 --warmup 0
run sq1_cfg3      rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_sq1_cfg3 -- $C3 --steps 1 --warmup 0
# ... its instruction cache (a 190 k-instruction kernel against a 64 KiB cache) and vector-memory instruction mix
run sq3_cfg3      rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_HITS SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sq3_cfg3 -- $C3 --steps 1 --warmup 0
# the large-K (blocked LDL', f64 MFMA) path of the workgroup kernel: K -> 320 rows, 256 QPs (no BASELINE config reaches it)
K320="python3 $GRAFT_REPO_ROOT/tools/debug_parity.py gen:320,1,4,640,0.5,0.0,1.0,0.0 256 wave_kernel=0"
run mfma_k320     rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_mfma_k320 -- $K320
python3 $GRAFT_REPO_ROOT/tools/pmc_to_json.py $TAG
find $OUT -name "*kernel_stats.csv"
