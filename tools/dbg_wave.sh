#!/bin/bash
# first checks of the wavefront kernel on the GPU box (small cases first; every step under its own timeout)
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/dbg_wave.log
: > $L
run() { timeout -k 10 150 python tools/debug_parity.py "$@" >> $L 2>&1 || { echo "FAILED/TIMEOUT: $*" >> $L; return 1; }; }
run cfg1 8 && run gen:64,1,4,128,1e-3,0.12,1.05,0.1 16 && run gen:96,3,5,200,1e-3,0.1,1.02,0.1 8 && \
run gen:160,1,5,320,1e-3,0.06,0.98,0.1 16 && run cfg4 16 && run cfg4 64 100 && run cfg3 8 && run cfg4 64 0 wave_kernel=0
tail -40 $L
