#!/usr/bin/env python3
"""per-kernel register / spill / LDS figures of a gfx950 assembly file (hipcc -save-temps=obj):
   tools/regs.py file.s [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for blk in s.split('  - .agpr_count:')[1:]:
    def g(key):
        m = re.search(r'\.%s:\s+(\S+)' % key, blk)
        return m.group(1) if m else '?'
    name = g('name')
    if flt not in name:
        continue
    d = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    d = d.replace('ipa::', '').split('(')[0]
    print('%-110s vgpr %s sgpr %s sgpr_spill %s vgpr_spill %s lds %s scratch %s' % (
        d[:110], g('vgpr_count'), g('sgpr_count'), g('sgpr_spill_count'), g('vgpr_spill_count'),
        g('group_segment_fixed_size'), g('private_segment_fixed_size')))
