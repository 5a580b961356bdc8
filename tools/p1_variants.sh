#!/bin/bash
# GPU box, development aid: runs the diagnostic Phase-1 libraries libssqp_hip_x<name>.so (built by hand with different -D
# switches) on cfg5 -- bit-identity against the host stage, time, phase profile.   usage: tools/p1_variants.sh out_dir name...
out=$1; shift
mkdir -p $out
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/statusswitchingqp.jl_amd/libssqp_hip_x$v.so
  echo "=== $v" >> $out/variants.log
  SSQP_HIP_LIB=$lib timeout -k 10 120 python tools/dbg_phase1.py cfg5 1 0 2>&1 | grep phase1_wave >> $out/variants.log || exit 1
  SSQP_PROF_LIB=$lib timeout -k 10 120 python tools/phase1_phase_profile.py cfg5 1 2>&1 | grep -E "cycles per QP|inv\(lu|Y = |xb = |many-rows" >> $out/variants.log || exit 1
done
