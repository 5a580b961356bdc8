#!/bin/bash
# Register / spill / scratch figures of every solve kernel as the compiler reports them in the code-object metadata
# (same flags as the Makefile; --cuda-device-only -S keeps the AMDGPU assembly with its .amdhsa metadata).
cd "$(dirname "$0")/../statusswitchingqp.jl_amd/csrc"
for spec in ssqp_wave.hip:0 ssqp_wave.hip:1 ssqp_kernels.hip: ssqp_phase1.hip:; do
  f=${spec%%:*}; v=${spec##*:}
  extra=""; [ $f = ssqp_phase1.hip ] && extra="-ffp-contract=off"
  [ -n "$v" ] && extra="-DSSQP_WAVE_VARIANT=$v"
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../include -I. -mllvm -sink-insts-to-avoid-spills=1 $extra -S --cuda-device-only $f -o /tmp/notes_$f$v.s 2>/dev/null &
done
wait
for spec in ssqp_wave.hip:0 ssqp_wave.hip:1 ssqp_kernels.hip: ssqp_phase1.hip:; do
  f=${spec%%:*}; v=${spec##*:}
  python3 - /tmp/notes_$f$v.s $f <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
md=txt[txt.index('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    blk='.agpr_count:'+blk
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if 'genV' in name or 'prep' in name: continue
    print('%-16s %-58s vgpr %s (agpr %s) sgpr %s  vgpr_spill %s  sgpr_spill %s  scratch %s B  code %d instructions' % (
        sys.argv[2], name[:58], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
        g('private_segment_fixed_size'), 0))
PY
done
