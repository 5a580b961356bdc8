#!/bin/bash
# GPU box: the part of a round's evidence that is not rocprofv3 (tools/profile_round.sh) nor one bench line per config
# (tools/bench_configs.sh): in-kernel phase profiles (diagnostic build, shipped as libssqp_hip_profx.so), the Phase-1
# counters, the shared-V sweep and the host-buffer entry points.   usage: tools/evidence_round.sh <tag>
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/evid_$TAG
mkdir -p $OUT
PROF=$GRAFT_REPO_ROOT/statusswitchingqp.jl_amd/libssqp_hip_profx.so
F='^make\|amdgpu.ids'
{
  echo "### Phase-1 kernels, cfg4 x 1024 (tools/phase1_phase_profile.py, diagnostic build): workgroup kernel, then the one-wavefront-per-QP kernel"
  SSQP_PROF_LIB=$PROF timeout -k 10 300 python tools/phase1_phase_profile.py cfg4 1024 2>&1 | grep -v "$F"
  echo; echo "### Phase-1 workgroup kernel (many-rows build), cfg5 x 1"
  SSQP_PROF_LIB=$PROF timeout -k 10 300 python tools/phase1_phase_profile.py cfg5 1 2>&1 | grep -v "$F" | head -16
  echo; echo "### wavefront kernel, four-per-CU build, cfg4 x 1024 (tools/wave_phase_profile.py)"
  SSQP_PROF_LIB=$PROF timeout -k 10 300 python tools/wave_phase_profile.py cfg4 1024 2>&1 | grep -v "$F"
  echo; echo "### big-factor build from the first pass, cfg3 x 1024 (wave_kernel=2)"
  SSQP_PROF_LIB=$PROF timeout -k 10 300 python tools/wave_phase_profile.py cfg3 1024 wave_kernel=2 2>&1 | grep -v "$F"
} > $OUT/phase_profiles.txt 2>&1 || exit 1
tools/pmc_phase1.sh $TAG cfg4 1024 > $OUT/pmc_phase1_wave.txt 2>&1 || { tail -5 $OUT/pmc_phase1_wave.txt; exit 1; }
timeout -k 10 300 python tools/shared_v_bench.py 1024 2>&1 | grep -v "$F" > $OUT/shared_v.txt || exit 1
timeout -k 10 300 python tools/pcie_inclusive.py 2>&1 | grep -v "$F" > $OUT/host_buffer_entries.txt || exit 1
for c in "cfg4 1024" "cfg2 8" "cfg3 1024" "cfg1 1024"; do timeout -k 10 120 python tools/dbg_full.py $c 2>&1 | grep -v "$F"; done > $OUT/single_launch.txt || exit 1
for c in "cfg4 1024" "cfg3 1024" "cfg1 256" "cfg2 8" "cfg5 1"; do timeout -k 10 200 python tools/dbg_phase1.py $c 2>&1 | grep -v "$F"; done > $OUT/phase1_kernels.txt || exit 1
tail -4 $OUT/phase_profiles.txt; cat $OUT/single_launch.txt; cat $OUT/phase1_kernels.txt; tail -3 $OUT/shared_v.txt
