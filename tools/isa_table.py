#!/usr/bin/env python3
"""Emitted-code table of kernels in a `hipcc -S --cuda-device-only` listing: registers, spills, vector instructions, of which
multiplications (v_mad_u64_u32), 64-bit additions (v_lshl_add_u64: QAcc folds AND 64-bit address arithmetic - the per-line
breakdown `addr64` counts only those with an SGPR operand, i.e. uniform base + lane offset address formation), loads by kind.
STATIC counts (loops are counted once).   usage: tools/isa_table.py listing.s kernel-name-part [...]"""
import re,sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_isa_properties as t
def table(path, parts):
    asm=open(path).read()
    ks=t._kernels(asm)
    for part in parts:
        for name in t._find(ks,part):
            body=ks[name][0]
            ops={}
            for line in body.split("\n"):
                m=re.match(r"\s*(v_[a-z0-9_]+|s_load\w+|global_load\w+|buffer_load\w+|ds_read\w+|scratch_\w+)",line)
                if m: ops[m.group(1)]=ops.get(m.group(1),0)+1
            valu=sum(v for k,v in ops.items() if k.startswith("v_"))
            addr64=len(re.findall(r"v_lshl_add_u64 v\[\d+:\d+\], (?:s\[\d+:\d+\], \d+, v\[\d+:\d+\]|v\[\d+:\d+\], \d+, s\[\d+:\d+\])", body))
            sg=ks[name][1]['sgpr_spill']
            short=re.sub(r"^_ZN3lmn\d+","",name)[:40]
            print("%-42s vgpr %3d spill %d  VALU %5d  mad_u64 %4d  add64 %3d (addr %3d)  sgpr_spill %2d  readfirstlane %3d  global_load %3d  buffer_load %3d  ds_read %3d  s_load %3d"%(short,ks[name][1]['vgpr'],ks[name][1]['vgpr_spill'],valu,ops.get('v_mad_u64_u32',0),ops.get('v_lshl_add_u64',0),addr64,sg,ops.get('v_readfirstlane_b32',0),sum(v for k,v in ops.items() if k.startswith('global_load')),sum(v for k,v in ops.items() if k.startswith('buffer_load')),sum(v for k,v in ops.items() if k.startswith('ds_read')),sum(v for k,v in ops.items() if k.startswith('s_load'))))
if __name__=="__main__":
    table(sys.argv[1], sys.argv[2:])
