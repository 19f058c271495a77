#!/usr/bin/env python3
"""instruction mix of the hand-scheduled steps of a wave_stencil kernel (wave_pipe.hpp):
   tools/pipe_isa.py file.s <mangled-name substring> [step index]
The kernel's text is cut at the operand-less waits that open a step (s_waitcnt vmcnt(9|8) pairs)
and the hot path of one step (the branch over the border-aware sampler not taken) is counted."""
import collections
import re
import sys

s = open(sys.argv[1]).read().split('\n')
flt = sys.argv[2]
want = int(sys.argv[3]) if len(sys.argv) > 3 else 2
start = [i for i, l in enumerate(s) if l.startswith('_Z') and flt in l and l.rstrip().endswith(':')
         or (l.startswith('_Z') and flt in l and ':' in l.split(';')[0])]
start = start[0]
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
k = s[start:end]
# step starts: first of each (vmcnt(8), vmcnt(9)) wait pair in either order
opens = []
i = 0
while i < len(k):
    if re.search(r's_waitcnt vmcnt\((8|9|10|11)\)\s*$', k[i]):
        opens.append(i)
        # skip the partner wait
        j = i + 1
        while j < len(k) and j < i + 40 and not re.search(r's_waitcnt vmcnt\((8|9|10|11)\)\s*$', k[j]):
            j += 1
        i = j + 1
    else:
        i += 1
print('kernel lines', len(k), 'steps found', len(opens))
a, b = opens[want], opens[want + 1]
raw = k[a:b]
bi = None
for i, l in enumerate(raw):
    if 's_cbranch_vccz' in l and any('ds_write' in raw[j] for j in range(max(0, i - 4), i)):
        bi = i
        tgt = l.split()[-1]
        break
hot = raw
if bi is not None:
    ti = [i for i, l in enumerate(raw) if l.startswith(tgt + ':')][0]
    hot = raw[:bi + 1] + raw[ti:]
c = collections.Counter()
for l in hot:
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.'):
        continue
    c[l.split()[0]] += 1
n = sum(c.values())
print('step %d: hot instructions %d  VALU %d  SALU %d  VMEM %d  LDS %d' % (
    want, n, sum(v for o, v in c.items() if o.startswith('v_')),
    sum(v for o, v in c.items() if o.startswith('s_')),
    sum(v for o, v in c.items() if o.startswith(('buffer_', 'global_'))),
    sum(v for o, v in c.items() if o.startswith('ds_'))))
print(', '.join('%s %d' % kv for kv in c.most_common(45)))
