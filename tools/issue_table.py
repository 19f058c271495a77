"""Per-kernel instruction / LDS issue figures from the rocprofv3 counter passes of tools/r06_pmc.sh:

    python3 tools/issue_table.py gpurun_out/r06 <tag> [--json profiles/r06_issue.json]

For every case (c4, fused, lz4q, cubicq, c3lin, c3cubic) the dominant kernel's counters per launch and the derived
figures the bench line quotes instead of a guessed roofline label:
  valu_per_unit       vector instructions per wave and unit of work (a 64-sample step of the tile kernels, a row
                      step of the strip kernels)
  frac_valu_issue     SQ_ACTIVE_INST_VALU (quad-cycles the SIMDs spend issuing vector instructions) / (SIMDs x the
                      launch's quad-cycles)
  frac_lds_busy       SQ_LDS_IDX_ACTIVE (LDS-array cycles) / (CUs x the launch's cycles)
  frac_lds_conflict   SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
The launch's cycles: GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)."""
import collections
import csv
import glob
import json
import os
import sys

CUS, SIMDS = 256, 1024
UNITS = {   # units of work per launch: (frames, what)
    'c4': (64, 'row'), 'fused': (64, 'row'), 'lz4q': (16, 'px64'), 'cubicq': (16, 'px64'),
    'c3lin': (16, 'row'), 'c3cubic': (16, 'px64'),
    'lz16q': (16, 'px64'), 'lz8q': (16, 'px64'), 'remaplin': (16, 'px64'), 'remaplz4': (16, 'px64'),
    'c5': (16, 'row'), 'conv11': (16, 'row'),
}
H, W = 2160, 3840


def counters(d):
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            per[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: {c: sum(v.values()) / len(v) for c, v in cs.items()} for k, cs in per.items()}


def main():
    root, tag = sys.argv[1], sys.argv[2]
    out = {}
    for case, (frames, what) in UNITS.items():
        if not os.path.isdir(os.path.join(root, 'pmc_%s_%s' % (tag, case))):
            continue
        per = counters(os.path.join(root, 'pmc_%s_%s' % (tag, case)))
        per = {k: v for k, v in per.items() if 'build_' not in k and 'rocclr' not in k and 'store_coords' not in k}
        if not per:
            continue
        dom = max(per, key=lambda k: per[k].get('GRBM_GUI_ACTIVE', 0))
        c = per[dom]
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        if what == 'px64':
            units = frames * H * W / 64.0
        else:   # row steps of a wave: 16 strips per row, strips of 144 / 72 rows + K - 1 halo rows: ~2 % over H
            units = frames * 16 * H * 1.03
        e = {'kernel': dom[:120], 'unit': 'a wave\'s 64 samples' if what == 'px64' else 'a wave\'s row step (approx.)',
             'cycles_per_launch': round(cyc),
             'valu_per_unit': round(c['SQ_INSTS_VALU'] / units, 1),
             'salu_per_unit': round(c['SQ_INSTS_SALU'] / units, 1),
             'lds_instr_per_unit': round(c['SQ_INSTS_LDS'] / units, 1),
             'vmem_rd_per_unit': round(c['SQ_INSTS_VMEM_RD'] / units, 1),
             'frac_valu_issue': round(c['SQ_ACTIVE_INST_VALU'] * 4 / SIMDS / cyc, 3),
             'frac_lds_busy': round(c.get('SQ_LDS_IDX_ACTIVE', 0) / CUS / cyc, 3),
             'frac_lds_conflict': round(c.get('SQ_LDS_BANK_CONFLICT', 0) / max(1.0, c.get('SQ_LDS_IDX_ACTIVE', 0)), 3),
             'frac_wait_inst': round(c['SQ_WAIT_INST_ANY'] / max(1.0, c['SQ_WAVE_CYCLES']), 3),
             'insts_valu_per_launch': int(c['SQ_INSTS_VALU'])}
        out[case] = e
    print('%-8s %-62s %9s %7s %7s %7s %7s %10s %9s %9s' % ('case', 'kernel', 'cycles', 'VALU/u', 'SALU/u', 'LDS/u', 'VMEM/u',
                                                          'VALU issue', 'LDS busy', 'conflict'))
    for case, e in out.items():
        print('%-8s %-62s %9d %7.1f %7.1f %7.1f %7.1f %10.3f %9.3f %9.3f' % (
            case, e['kernel'][10:72], e['cycles_per_launch'], e['valu_per_unit'], e['salu_per_unit'],
            e['lds_instr_per_unit'], e['vmem_rd_per_unit'], e['frac_valu_issue'], e['frac_lds_busy'],
            e['frac_lds_conflict']))
    if '--json' in sys.argv:
        p = sys.argv[sys.argv.index('--json') + 1]
        json.dump(out, open(p, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
