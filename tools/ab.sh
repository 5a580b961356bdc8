#!/bin/bash
# GPU box: same-box A/B of two builds of the library (build/ab/libA.so, libB.so), alternating
for rep in 1 2; do
  for v in A B; do
    SSQP_HIP_LIB=$GRAFT_REPO_ROOT/build/ab/lib$v.so python tools/debug_parity.py ${1:-cfg4} ${2:-1024} 2>&1 | grep -v amdgpu | sed "s/^/$v: /"
  done
done
