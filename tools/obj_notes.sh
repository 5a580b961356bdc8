#!/bin/bash
# Register / spill / scratch figures of the kernels in a compiled object (no recompilation): the device code object is
# cut out of the host object's fat binary and its metadata notes are read.   usage: tools/obj_notes.sh <file.o>...
LLVM=/opt/rocm/lib/llvm/bin
for o in "$@"; do
  tmp=$(mktemp -d)
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $o $tmp/fb.bin
  $LLVM/clang-offload-bundler --unbundle --type=o --input=$tmp/fb.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co
  $LLVM/llvm-readelf --notes $tmp/dev.co > $tmp/notes.txt
  ninstr=$($LLVM/llvm-objdump -d $tmp/dev.co | grep -c "^\s*[a-z_0-9]* .*//")
  python3 - $tmp/notes.txt "$(basename $o)" $ninstr <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
for blk in re.split(r'\n\s*- \.agpr_count:', txt)[1:]:
    blk='.agpr_count:'+blk
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if 'genV' in name or 'prep' in name: continue
    print('%-18s %-50s vgpr %s (agpr %s) sgpr %s  vgpr_spill %s  sgpr_spill %s  scratch %s B  lds %s B' % (
        sys.argv[2], name[:50], g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'),
        g('private_segment_fixed_size'), g('group_segment_fixed_size')))
print('%-18s %s instructions in the code object' % (sys.argv[2], sys.argv[3]))
PY
  rm -rf $tmp
done
