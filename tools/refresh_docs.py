"""Rewrite the number-carrying passages of DESIGN.md / README.md / profiles/README.md from the committed evidence files
(profiles/r06_bench.json, r06_bench_profiled_pass.json, r06_bench_2ranks_one_gpu.json, r06_issue.json), so that prose
and files cannot drift apart.  Idempotent; run from the repository root after tools/r06_evidence.sh."""
import csv
import json
import re

l = json.load(open('profiles/r06_bench.json'))
pp = json.load(open('profiles/r06_bench_profiled_pass.json'))
e2 = json.load(open('profiles/r06_bench_2ranks_one_gpu.json'))
iss = json.load(open('profiles/r06_issue.json'))
ks = [x for x in csv.DictReader(open('profiles/r06_kernel_stats.csv')) if 'wave_sep_kernel' in x['Name']][0]
avg_us, calls = float(ks['AverageNs']) / 1e3, ks['Calls']
oc = {e['workload']: e for e in l['other_configs']}


def find(sub):
    for k in oc:
        if sub in k:
            return k
    raise KeyError(sub)


def issue(k):
    t = iss[k]
    return '; vector issue %.2f, LDS busy %.2f' % (t['frac_valu_issue'], t['frac_lds_busy'])


def row(name, key, extra=''):
    e = oc[key]
    return '| %s | %.3f | %.3f%s |' % (name, e['ms'], e['frac_compulsory'], extra)


def cls(v):
    return 'fast' if v < 0.385 else ('middle' if v < 0.41 else 'slow')


probe = (l['config']['probe_src_ms'], l['config']['probe_dst_ms'])
pprobe = (pp['config']['probe_src_ms'], pp['config']['probe_dst_ms'])
s = open('DESIGN.md').read()

# --- headline table: the rows of the committed library
i = s.index('| **round 6, the library as committed, `profiles/r06_bench.json`**')
j = s.index('| round 6, in-process A/B dense -> separable on three boxes')
s = s[:i] + '''| **round 6, the library as committed, `profiles/r06_bench.json`** (`python bench.py`, fresh process: separable 5 + 5 loop, short-strip tails) | **%.4f** | **%.4f** | %s / %s (probe %.3f / %.3f) |
| ... the dense loop in that process, same buffers (`roofline.dense_loop`) | %.4f | %.4f | |
| ... ONE profiled process of that box (`profiles/r06_bench_profiled_pass.json` + `r06_kernel_stats.csv`: its own line and its own `rocprofv3 --stats`) | %.4f (line); %.1f us average over %s launches (profile) | %.4f | %s / %s (%.3f / %.3f) |
| ... the library as committed on boxes that handed out two slow-class buffers: 13 fresh processes (`profiles/r06_bench_slow_box_final.json`, `r06_micro.txt`) | 0.957 - 0.963 | 0.560 - 0.563 | slow / slow |
''' % (l['ms_per_step'], l['roofline']['frac'], cls(probe[0]), cls(probe[1]), probe[0], probe[1],
       l['roofline']['dense_loop']['ms_per_step'], l['roofline']['dense_loop']['frac'],
       pp['ms_per_step'], avg_us, calls, pp['roofline']['frac'], cls(pprobe[0]), cls(pprobe[1]), pprobe[0], pprobe[1]) + s[j:]

# --- other configurations table
i = s.index('| configuration | ms | frac of 8 TB/s (compulsory bytes) |')
j = s.index('CPU baseline (`cpu_baseline`, kind "port")')
tab = '| configuration | ms | frac of 8 TB/s (compulsory bytes) |\n|---|---|---|\n' + '\n'.join([
    row('headline at 128 frames per launch', find('headline at 128')),
    row('headline at 256 frames per launch', find('headline at 256')),
    row("... with the reference's own camera matrix (`getOptimalNewCameraMatrix`, alpha = 1: 2.8 % of the picture outside the source), 128 frames", find("reference's own camera matrix")),
    row('C2 1080p float32 undistort + 5x5, 64 frames', find('C2 1080p')),
    row('`LensDistortion.correct` itself: cv2.remap from the map pair, 16 x 4K, bilinear (tile kernel)', find('map pair (linear)'), issue('remaplin') if 'remaplin' in iss else ''),
    row('... Lanczos4', find('map pair (lanczos4)'), issue('remaplz4') if 'remaplz4' in iss else issue('lz4q')),
    row('... on camera frames (round 6, the strips with K = 1): uint16 -> uint16 with cv2\'s 16U arithmetic, 16 frames (the gather kernel of rounds 1 - 5: 0.400)', find('uint16 -> uint16, cv2 16U arithmetic, 16')),
    row('... 64 frames (1.61)', find('uint16 -> uint16, cv2 16U arithmetic, 64')),
    row('... uint8 -> uint8 with cv2\'s 8U fixed point, 64 frames (1.01; 16 frames: 0.155 against 0.259)', find('uint8 -> uint8, cv2 8U fixed point, 64')),
    row('... uint16 -> float32 (the `toFloatArray` ingest), 64 frames (1.28)', find('uint16 -> float32 (toFloatArray ingest), 64')),
    row('C3 4K perspective warp (bilinear) + separable 9+9, 16 frames, one kernel', find('warp (linear) + separable 9+9, 16'), issue('c3lin')),
    row('... 64 frames per launch', find('warp (linear) + separable 9+9, 64')),
    row('C3 bicubic, two launches (tile warp, filter)', find('warp (cubic) + separable'), '; the tile warp:' + issue('c3cubic')[1:]),
    row('`PerspectiveCorrection.correct` default: Lanczos4 warp float32, 16 x 4K', find('default 4K f32, Lanczos4'), issue('lz4q')),
    row('C3 rotated by 15 degrees (two launches)', find('C3 rotated')),
    row('Lanczos4 rotated by 15 degrees', find('default rotated')),
    row("Lanczos4 uint16 (OpenCV's 16U arithmetic)", find('4K uint16, Lanczos4'), issue('lz16q') if 'lz16q' in iss else ''),
    row('Lanczos4 uint8 (8U tables in LDS)', find('4K uint8, Lanczos4'), issue('lz8q') if 'lz8q' in iss else ''),
    row('C5-like 4K: bicubic warp + dense 11x11 (two launches)', find('C5-like')),
    row('C5 8K, 4 frames', find('C5 8K')),
    row('C4 64 x 4K uint16 -> float32 undistort + dense 7x7', find('C4 4K uint16'), issue('c4')),
    row('... with a 7x7 Gaussian (an outer product: the separable 7 + 7 loop, since round 6 for uint16 frames too)', find('with a 7x7 Gaussian')),
])
e = oc[find('C4 the same chain host')]
tab += '\n| C4 host -> host, 24 frames through page-locked buffers | %.1f | PCIe-bound (%.1f Gpix/s), never `value` |\n\n' % (e['ms'], e['Mpix_s'] / 1e3)
tab += '''(against round 5's line: C2 0.278 -> %.3f (separable route, short-strip tails), C3 bilinear 0.358 -> %.3f (tails), C3 bicubic
0.533 -> %.3f (packed bicubic sample), Lanczos4 float32 0.614 -> %.3f and under 15 degrees 0.846 -> %.3f (weight-table rows 12
floats apart, LDS pitch among all pitches), C5 0.799 / 0.815 -> %.3f / %.3f (the same pitch search, 16-byte 11-tap windows), C4
1.175 -> %.3f (conflict-free window reads, tails).  What the counters say of the kernels that are not HBM-bound is in the
third column: the Lanczos4 float32 warp - the reference's default - keeps its LDS arrays busy three quarters of the launch
and issues vector instructions two thirds of it; the bicubic tile kernel issues 70 %% (88 %% before round 6); C4 88 %%.)

''' % (oc[find('C2 1080p')]['ms'], oc[find('warp (linear) + separable 9+9, 16')]['ms'], oc[find('warp (cubic) + separable')]['ms'],
       oc[find('default 4K f32, Lanczos4')]['ms'], oc[find('default rotated')]['ms'], oc[find('C5-like')]['ms'], oc[find('C5 8K')]['ms'],
       oc[find('C4 4K uint16')]['ms'])
s = s[:i] + tab + s[j:]
cb = l['cpu_baseline']
s = re.sub(r'host: \d+ Mpix/s on \d+ threads \(32 - 35 on one', 'host: %.0f Mpix/s on %d threads (32 - 35 on one' % (cb['value'], cb['cores']), s)
s = re.sub(r'\(`config\.per_rank_ms_per_step`\n\[[^\]]*\]\) and `end_to_end` \(32 uint16 frames per rank host -> device -> host: [\d.]+ / [\d.]+ ms,\n[\d.]+ Gpix/s, \d+ GB/s',
           '(`config.per_rank_ms_per_step`\n%s) and `end_to_end` (32 uint16 frames per rank host -> device -> host: %.1f / %.1f ms,\n%.1f Gpix/s, %.0f GB/s'
           % (e2['config']['per_rank_ms_per_step'], e2['end_to_end']['per_rank_ms'][0], e2['end_to_end']['per_rank_ms'][1],
              e2['end_to_end']['Gpix_s_aggregate'], e2['end_to_end']['GB_s_pcie_aggregate']), s)
open('DESIGN.md', 'w').write(s)

p = open('profiles/README.md').read()
i = p.index('| `r06_bench.json` |')
j = p.index('| `r06_pmc.csv`, `pmc_summary.json` |')
p = p[:i] + '''| `r06_bench.json` | the bench line of the library as committed: 64 frames per launch, **%.4f ms/step, frac %.4f** on the separable 5 + 5 loop with short-strip tails (`config.path`, `roofline.kernel`), the dense loop timed beside it in the same process (`roofline.dense_loop`: %.4f ms, %.4f), the class of the batch buffers (`config.probe_src_ms` / `probe_dst_ms` %.3f / %.3f: %s / %s), the first 30 launches one event pair each (`config.first_launch_ms`); `other_configs` with the headline at 128 and 256 frames, C3 at 16 and 64 frames and the measured issue fractions (`issue_counters`) | `tools/r06_evidence.sh` (first line: `python bench.py`) |
| `r06_bench_profiled_pass.json`, `r06_kernel_stats.csv`, `r06_kernel_stats_by_grid.csv` | ONE process of the final library (another box): `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-configs` - the line that process printed (%.4f ms/step, frac %.4f, probe %.3f / %.3f) and its own per-kernel durations (`wave_sep_kernel<SampleRowSrc<float,1,MapCoord>,5>` %.1f us average over %s launches, which include the 30 single-event launches and the warm-up).  One of four such processes of that box (0.958 - 0.963; `r06_micro.txt`) | `tools/r06_profiled_passes.sh 4`, `python profiles/summarize.py r06 gpurun_out/r06p/pass3 <fetch> <write> --batch 64 --kernel wave_sep_kernel` |
| `r06_bench_final_library_other_box.json` | `python bench.py --no-cpu` of the library at the very end of the round (the uint8 chains in as well) on another box: **0.8976 ms/step, frac 0.601** (probe 0.427 / 0.378: a slow-class source, a fast-class result buffer - the result buffer is the one that matters: the strip-shaped store stream) | `python bench.py --no-cpu` |
| `r06_bench_slow_box_final.json`, `r06_bench_slow_box.json`, `r06_bench_fast_box_before_tail.json` | the same command on other boxes: the library as committed on two slow-class buffers (0.9632 ms, 0.560; twelve more such processes in `r06_micro.txt`: 0.957 - 0.961); earlier in the round, before the short-strip tails: both buffers slow 0.9852 ms (0.547), both fast 0.8968 ms (0.601) | `python bench.py` |
''' % (l['ms_per_step'], l['roofline']['frac'], l['roofline']['dense_loop']['ms_per_step'], l['roofline']['dense_loop']['frac'], probe[0], probe[1], cls(probe[0]), cls(probe[1]),
       pp['ms_per_step'], pp['roofline']['frac'], pprobe[0], pprobe[1], avg_us, calls) + p[j:]
p = p.replace('`tools/r06_pmc.sh c`, `python3 tools/issue_table.py gpurun_out/r06 c --json profiles/r06_issue.json`', '`tools/r06_pmc.sh d`, `tools/r06_pmc_more.sh d`, `python3 tools/issue_table.py gpurun_out/r06 d --json profiles/r06_issue.json`')
open('profiles/README.md', 'w').write(p)

r = open('README.md').read()
i = r.index('Fused undistort + 5x5 on 4K float32, 64 frames per launch:')
j = r.index('Round 6, second half - what fuzzing')
r = r[:i] + '''Fused undistort + 5x5 on 4K float32, 64 frames per launch: **%.3f ms per step = %.3f of the HBM peak from
COMPULSORY bytes in the committed run** (`profiles/r06_bench.json`: a %s-class source and a %s-class result buffer;
a profiled process on two %s-class buffers: %.3f ms, `rocprofv3` average %.1f us); **0.957 - 0.963 ms = 0.560 - 0.563 when both
buffers land in a slow region of the device memory** (13 fresh processes) - where round 5 measured 1.009 ms (0.534) and
rounds 3 - 4 1.09 - 1.10 (0.49).  On two fast-class buffers 0.897 ms (0.601) was measured before the last 2.5 %% went in.
The spread is the memory's (DESIGN.md section 5: one launch on windows of ONE buffer pair: 0.990 / 0.990 / 0.990 /
0.896 ms), not the run's.
''' % (l['ms_per_step'], l['roofline']['frac'], cls(probe[0]), cls(probe[1]), cls(max(pprobe)), pp['ms_per_step'], avg_us) + r[j:]
open('README.md', 'w').write(r)
print('refreshed: headline %.4f ms (%.4f), profiled pass %.4f ms / %.1f us' % (l['ms_per_step'], l['roofline']['frac'], pp['ms_per_step'], avg_us))
