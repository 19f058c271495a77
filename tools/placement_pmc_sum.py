"""per-counter means of the headline kernel's dispatches behind marker 1 (best candidate) and
behind marker 2 (worst): python tools/placement_pmc_sum.py <dir with counter_collection.csv ...>"""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        rows = list(csv.DictReader(open(f)))
        disp = collections.OrderedDict()
        for r in rows:
            disp.setdefault(int(r['Dispatch_Id']), (r['Kernel_Name'], {}))[1].setdefault(r['Counter_Name'], 0.0)
            disp[int(r['Dispatch_Id'])][1][r['Counter_Name']] += float(r['Counter_Value'])
        phase, acc = 0, {1: collections.defaultdict(list), 2: collections.defaultdict(list)}
        for i in sorted(disp):
            name, c = disp[i]
            if 'copyBuffer' in name or 'rocclr' in name:
                phase += 1
            elif 'wave_stencil_kernel' in name and phase in (1, 2):
                for k, v in c.items():
                    acc[phase][k].append(v)
        for k in sorted(acc[1]):
            a, b = acc[1][k], acc[2].get(k, [])
            if a and b:
                ma, mb = sum(a) / len(a), sum(b) / len(b)
                print('%-44s best %14.0f  worst %14.0f  worst/best %.3f  (n %d / %d)' % (k, ma, mb, mb / ma if ma else 0, len(a), len(b)))
