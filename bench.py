#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of lens-undistort + 5x5 filter on 4K float32 frames.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (LensDistortion-style undistort, bilinear,
BORDER_CONSTANT, then a 5x5 Gaussian given as an explicit kernel, 'reflect'
border) over one batch of synthetic 4K (3840x2160) float32 frames (64 by
default = the per-GPU batch of BASELINE.json's sharded configuration: 2.1 GB in,
2.1 GB out) that are already resident in HBM.  The batch is far larger than
the 256 MiB Infinity Cache, so source and destination really stream from/to
HBM.

For N > 1 the driver launches one rank per GPU (torch.distributed.run); frames
are independent, every rank processes its own batch (weak scaling), no
data-path collective; torch.distributed (gloo) is used only for the barrier
and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline`
(dominant kernel; COMPULSORY HBM bytes of the launch / HIP-event time, the
map pair counted once per launch), `other_configs` (BASELINE configurations
C2..C5 kernel-only, C4 also PCIe-inclusive; N=1 only) and `cpu_baseline` (the
oracle C restatement timed on this box's host cores, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H4K, W4K = 2160, 3840
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_PEAK_FLOPS = 157.3e12  # fp32 vector peak (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK = 256 * 64 * 2.4e9   # vector lane-instructions per second, every SIMD issuing
LDS_PEAK_BPS = 150e12       # aggregate ds_read_b64/b128 rate, every CU streaming (same guide)


def synth_frames(n, h, w, seed0=0):
    """SURVEY §8(d) content: smooth pattern + noise in [0,1]; cheap per-frame variation"""
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    base = 0.5 + 0.25 * np.sin(2 * np.pi * x / 97) + 0.25 * np.cos(2 * np.pi * y / 61)
    out = np.empty((n, h, w), np.float32)
    # 16 seeded frames; frame i >= 16 is frame i % 16 rolled down by 37 rows per reuse, so every
    # frame of a large batch is distinct without drawing 8 Mpx of normals for each
    for i in range(min(n, 16)):
        rng = np.random.default_rng(seed0 + i)
        out[i] = np.clip(base + 0.05 * rng.standard_normal((h, w), dtype=np.float32), 0, 1)
    for i in range(16, n):
        out[i] = np.roll(out[i % 16], 37 * (i // 16), axis=0)
    return out


def camera(h, w):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    return K, dist


def gauss5():
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    return np.outer(g, g)


def usable_cores():
    """cores this process may actually use: affinity mask capped by the cgroup CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota not in ('max', '-1'):
                n = min(n, max(1, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def cpu_baseline(h, w, K, dist, k5, budget_s=15.0):
    """oracle (C restatement, 'port') on the host cores; bounded sample.  The thread counts
    1, usable cores and a few in between are timed; the fastest is reported with ITS count."""
    from oracle import oracle as orc
    orc.build()
    frame = synth_frames(1, h, w, 1000)[0]
    mx, my = orc.build_undistort_map(K, dist, K, h, w)
    avail = max(1, min(usable_cores(), orc.max_threads()))
    counts = sorted({1, avail} | {c for c in (4, 8, 16, 32) if c < avail})
    res, total = {}, 0
    for threads in counts:
        orc.set_threads(threads)
        orc.remap_conv2d(frame, mx, my, k5)  # warm
        n, t0 = 0, time.perf_counter()
        while True:
            orc.remap_conv2d(frame, mx, my, k5)
            n += 1
            el = time.perf_counter() - t0
            if el > budget_s / len(counts) or n >= 50:
                break
        res[threads] = n * h * w / el / 1e6
        total += n
    orc.set_threads(1)
    best = max(res, key=res.get)
    return {'value': round(res[best], 2), 'unit': 'Mpix/s', 'cores': best, 'kind': 'port',
            'sample': '%d frame(s) of %dx%d float32 in total, map-based undistort + 5x5, '
                      'oracle/oracle.c with OpenMP; Mpix/s by thread count: %s'
                      % (total, w, h, ', '.join('%d: %.1f' % (c, res[c]) for c in counts))}


def pmc_traffic(variant, batch, h, w):
    """HBM-side bytes per launch of the dominant kernel, from the rocprofv3 --pmc passes of THIS
    round on this same command (profiles/pmc_summary.json, written by profiles/summarize.py:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE).  Counters cannot be read from inside the
    run, so this is a recorded measurement: (bytes, source) or (None, reason)."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')) as f:
            s = json.load(f)
        e = s.get(variant)
        if e and e.get('batch') == batch and e.get('height') == h and e.get('width') == w:
            return e['traffic_bytes_per_launch'], ('recorded: profiles/pmc_summary.json (%s; '
                                                   'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, '
                                                   'separate passes of this command)'
                                                   % e.get('round', 'r02'))
        return None, 'no PMC pass recorded for variant=%s batch=%d %dx%d' % (variant, batch, w, h)
    except (OSError, ValueError, KeyError):
        return None, 'profiles/pmc_summary.json missing'


def timed(ctx, fn, steps, warmup):
    """ms per call over `steps` calls, HIP events on the context's stream"""
    for _ in range(warmup):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / steps


def timed_settled(ctx, fn, steps, warmup, settle_s=0.2):
    """`timed` after `settle_s` seconds of the same launches: every configuration below builds
    its frames on the host first, the GPU idles meanwhile and its clocks take ~100 ms of load to
    come back - three warm-up launches of a 1 ms kernel are inside that ramp"""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle_s:
        for _ in range(8):
            fn()
        ctx.synchronize()
    return timed(ctx, fn, steps, warmup)


def other_configs(ctx, ia, ops, budget_launches=60):
    """BASELINE.json configurations C2..C5, kernel-only on device-resident data (C4 also
    PCIe-inclusive), each with its COMPULSORY HBM bytes: every input element read once, every
    output written once, maps counted once per launch (frames of a launch share them)."""
    from imgprocessor_amd.utils import getPerspectiveTransform
    out = []
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)

    try:
        with open(os.path.join(ROOT, 'profiles', 'r06_issue.json')) as f:
            issue_tab = json.load(f)
    except (OSError, ValueError):
        issue_tab = {}

    def entry(name, frames, h, w, ms, comp_bytes, launches, note=None, bound='hbm', work=None, issue=None):
        """comp_bytes: the WORKLOAD's compulsory HBM bytes (inputs once, outputs once) whatever the
        number of launches - an intermediate image through the workspace is not compulsory.
        bound: the roofline that binds the dominant kernel; for 'valu' / 'lds' `work` is the
        launch's flops / LDS bytes and the fraction is taken against that unit's peak."""
        e = {'workload': name, 'frames': frames, 'ms': round(ms, 4),
             'Mpix_s': round(frames * h * w / ms / 1e3, 1),
             'compulsory_bytes': int(comp_bytes),
             'frac_compulsory': round(comp_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             'launches': launches, 'bound': bound}
        if bound == 'valu' and work:
            e['flops'] = int(work)
            e['frac_valu'] = round(work / (ms * 1e-3) / VALU_PEAK_FLOPS, 4)
        if bound == 'lds' and work:
            e['lds_bytes'] = int(work)
            e['frac_lds'] = round(work / (ms * 1e-3) / LDS_PEAK_BPS, 4)
        if bound == 'valu_issue' and work:
            # integer work: vector instructions x lanes against what the SIMDs can issue
            # (256 CUs x 64 lanes per clock at the guide's 2.4 GHz)
            e['vector_lane_instructions'] = int(work)
            e['frac_valu_issue'] = round(work / (ms * 1e-3) / VALU_ISSUE_PEAK, 4)
        if issue and issue in issue_tab:
            # what bounds a kernel that is not HBM-bound, MEASURED: the rocprofv3 counter passes of this round on
            # this configuration (tools/r06_pmc.sh -> tools/issue_table.py -> profiles/r06_issue.json): the
            # fraction of the launch the SIMDs spend issuing vector instructions, the fraction the LDS arrays
            # are busy, bank conflicts as a share of that
            t = issue_tab[issue]
            e['issue_counters'] = {k: t[k] for k in ('kernel', 'valu_per_unit', 'unit', 'frac_valu_issue',
                                                     'frac_lds_busy', 'frac_lds_conflict') if k in t}
            e['issue_counters']['source'] = 'recorded: profiles/r06_issue.json (rocprofv3 --pmc SQ_* passes, round 6)'
            fv, fl = t.get('frac_valu_issue', 0), t.get('frac_lds_busy', 0)
            e['bound'] = ('lds + valu_issue' if min(fv, fl) > 0.55 else ('lds' if fl > fv else 'valu_issue'))
        if note:
            e['note'] = note
        out.append(e)

    def placed(shape, dtype, launch=None):
        """result buffer of a configuration: a plain Context.empty - blocks of 256 MiB and more
        are placed by the context's block pool itself (device.py::_alloc_placed), like every
        array a caller of the library allocates"""
        return ctx.empty(shape, dtype)

    # C2: 1080p float32, radial undistort (maps) + 5x5 Gaussian, 1 GPU
    h, w, B = 1080, 1920, 64
    K, dcoef = camera(h, w)
    src = ctx.to_device(synth_frames(B, h, w, 200))
    dmx, dmy = ops.build_undistort_map(K, dcoef, K, h, w, ctx=ctx, device=True)
    dst = placed((B, h, w), np.float32, lambda d: ops.remap_conv2d(src, dmx, dmy, k5, out=d))
    ms = timed_settled(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst), budget_launches, 5)
    entry('C2 1080p f32, LensDistortion undistort (maps) + 5x5 Gaussian, %d frames/launch' % B,
          B, h, w, ms, (8 * B + 8) * h * w, 1)
    del src, dst, dmx, dmy

    # LensDistortion.correct itself: cv2.remap from the map pair, no filter behind it (SURVEY section
    # 8, rows a1 - a3) - bilinear as the reference calls it, and Lanczos4
    h, w, B = H4K, W4K, 16
    K, dcoef = camera(h, w)
    src = ctx.to_device(synth_frames(B, h, w, 250))
    dmx, dmy = ops.build_undistort_map(K, dcoef, K, h, w, ctx=ctx, device=True)
    dst = ctx.empty((B, h, w), np.float32)
    for interp in ('linear', 'lanczos4'):
        ms = timed_settled(ctx, lambda: ops.remap(src, dmx, dmy, interp, out=dst), budget_launches, 5)
        entry('LensDistortion.correct 4K f32, cv2.remap from the map pair (%s), %d frames/launch' % (interp, B),
              B, h, w, ms, (8 * B + 8) * h * w, 1,
              'tile kernel with the map pair as coordinate source (csrc/tile_warp.hpp)',
              **({'bound': 'lds', 'work': 40 * 8 * B * h * w, 'issue': 'lz4q'} if interp == 'lanczos4' else {}))
    del src, dst
    # ... on CAMERA frames (round 6): cv2.remap keeps the element type - uint16 with cv2's 16U arithmetic (what the
    # wrapper asks for: 'linear_cv_q5'), uint8 with its 8U fixed point - and the float32 ingest of uint16 frames
    # (transformations.toFloatArray): the marching strips of the chains without a filter (knob strip_remap)
    f16 = np.round(synth_frames(B, h, w, 260) * 4095).astype(np.uint16)
    for name, arr, interp, odt, bpp in (('uint16 -> uint16, cv2 16U arithmetic', f16, 'linear_cv_q5', np.uint16, 4),
                                        ('uint8 -> uint8, cv2 8U fixed point', (f16 >> 4).astype(np.uint8), 'linear', np.uint8, 2),
                                        ('uint16 -> float32 (toFloatArray ingest)', f16, 'linear', np.float32, 6)):
        for nb in (B, 64):
            dsrc = ctx.to_device(arr if nb == B else np.concatenate([arr] * (nb // B)))
            dd = ctx.empty((nb, h, w), odt)
            ms = timed_settled(ctx, lambda: ops.remap(dsrc, dmx, dmy, interp, out_dtype=odt, out=dd), budget_launches // (1 if nb == B else 2), 3)
            entry('LensDistortion.correct 4K %s, %d frames/launch' % (name, nb), nb, h, w, ms, (bpp * nb + 8) * h * w, 1,
                  'wave_sep_kernel with K = 1 on the shared-record loop (csrc/wave_pipe.hpp CV16 / fused_sep_c.hip); the gather '
                  'kernels these calls took in rounds 1 - 5: profiles/r06_micro.txt')
            del dsrc, dd
    del dmx, dmy

    # C3: 4K float32, perspective remap (homography in the kernel: no maps) + separable 9+9
    h, w, B = H4K, W4K, 16
    quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
    rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
    g9 = ops.gaussian_kernel1d(1.0)
    src = ctx.to_device(synth_frames(B, h, w, 300))
    dst = placed((B, h, w), np.float32,
                 lambda d: ops.warp_perspective_sepconv2d(src, Hm, (h, w), g9, g9, 'linear', out=d))
    for interp in ('linear', 'cubic'):
        ms = timed_settled(ctx, lambda: ops.warp_perspective_sepconv2d(src, Hm, (h, w), g9, g9, interp,
                                                               out=dst), budget_launches, 5)
        two = interp != 'linear'
        entry('C3 4K f32, PerspectiveCorrection warp (%s) + separable 9+9, %d frames/launch'
              % (interp, B), B, h, w, ms, 8 * B * h * w, 2 if two else 1,
              'two launches through the workspace: tile warp (vector-issue-bound, see issue_counters), then the '
              'filter (16 B/px of traffic for an 8 B/px workload)' if two else None,
              issue='c3cubic' if two else 'c3lin')

    # C3 bilinear at the headline's batch: 64 frames per launch (the coordinate table of the launch - the
    # homography evaluated once, csrc/stored_coords.hpp - is shared by 16 frame groups instead of 4)
    B64 = 64
    s64 = ctx.to_device(synth_frames(B64, h, w, 300))
    d64 = ctx.empty((B64, h, w), np.float32)
    ms = timed_settled(ctx, lambda: ops.warp_perspective_sepconv2d(s64, Hm, (h, w), g9, g9, 'linear', out=d64),
                       budget_launches // 2, 3)
    entry('C3 4K f32, PerspectiveCorrection warp (linear) + separable 9+9, %d frames/launch' % B64, B64, h, w, ms,
          8 * B64 * h * w, 1, issue='c3lin')
    del s64, d64

    # PerspectiveCorrection.correct as the reference calls it (cv2.warpPerspective with
    # INTER_LANCZOS4, camera/PerspectiveCorrection.py:401-405): float32 frames and the camera's
    # uint8 frames (OpenCV's 8U short-weight arithmetic, bit-exact against the oracle)
    ms = timed_settled(ctx, lambda: ops.warp_perspective(src, Hm, (h, w), 'lanczos4', out=dst),
               budget_launches, 5)
    entry('PerspectiveCorrection default 4K f32, Lanczos4 warp, %d frames/launch' % B, B, h, w, ms,
          8 * B * h * w, 1, 'tile kernel (csrc/tile_warp.hpp): the source box of a 64 x 32 output tile '
          'in LDS, rows in interleaved pairs; 40 8-byte LDS reads per sample (5 row pairs x 8 columns)',
          bound='lds', work=40 * 8 * B * h * w, issue='lz4q')

    # the same three chains with the picture rotated by 15 degrees: the row-walking kernels pay per
    # cache line a wave's gather touches (profiles/r04_micro.txt: up to 3x at 45 degrees); the tile
    # kernel's time hardly depends on the angle
    def rot15(h, w):
        a = np.deg2rad(15.0)
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                      [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy],
                      [0, 0, 1.0]])
        return np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]]) @ R
    Hr = rot15(h, w)
    ms = timed_settled(ctx, lambda: ops.warp_perspective_sepconv2d(src, Hr, (h, w), g9, g9, 'linear', out=dst),
                       budget_launches, 5)
    entry('C3 rotated by 15 degrees: 4K f32, warp (linear) + separable 9+9, %d frames/launch' % B, B, h, w,
          ms, 8 * B * h * w, 2, 'tile warp into the workspace, then the filter (16 B/px of traffic for '
          'an 8 B/px workload); the one fused kernel takes 0.72 ms at this angle')
    ms = timed_settled(ctx, lambda: ops.warp_perspective(src, Hr, (h, w), 'lanczos4', out=dst), budget_launches, 5)
    entry('PerspectiveCorrection default rotated by 15 degrees: 4K f32, Lanczos4 warp, %d frames/launch' % B,
          B, h, w, ms, 8 * B * h * w, 1, 'tile kernel; ring + gather kernels: 1.7 ms at this angle '
          '(issue_counters: the unrotated launch of the same kernel)',
          bound='lds', work=40 * 8 * B * h * w, issue='lz4q')
    u16 = ctx.to_device(np.round(synth_frames(B, h, w, 320) * 65535).astype(np.uint16))
    d16 = ctx.empty((B, h, w), np.uint16)
    ms = timed_settled(ctx, lambda: ops.warp_perspective(u16, Hm, (h, w), 'lanczos4', out=d16),
                       budget_launches // 2, 3)
    entry('PerspectiveCorrection default 4K uint16, Lanczos4 warp, %d frames/launch' % B, B, h, w, ms,
          4 * B * h * w, 1, "OpenCV's 16U arithmetic (float32 table weights, every product and sum rounded, "
          'no fma), bit-exact against the oracle; tile kernel with the box clipped to the source: per '
          'sample 40 8-byte LDS reads and ~200 vector instructions (80 packed multiplies, 40 packed adds '
          'for the 64 taps and their weights; counted by rocprofv3: SQ_INSTS_VALU per wave and sample)',
          bound='valu_issue', work=200 * B * h * w)
    del u16, d16
    u8 = ctx.to_device(np.round(synth_frames(B, h, w, 310) * 255).astype(np.uint8))
    d8 = ctx.empty((B, h, w), np.uint8)
    ms = timed_settled(ctx, lambda: ops.warp_perspective(u8, Hm, (h, w), 'lanczos4', out=d8),
               budget_launches // 2, 3)
    entry('PerspectiveCorrection default 4K uint8, Lanczos4 warp, %d frames/launch' % B, B, h, w,
          ms, 2 * B * h * w, 1, "OpenCV's 8U fixed-point weight table resident in LDS (integer-exact); "
          'bound by integer vector work, an HBM fraction is the wrong roofline: per sample 32 '
          'v_dot2_i32_i16 (the 64 taps), 32 v_perm_b32 (bytes out of the aligned tap dwords), ~40 for '
          'the homography, the 1/32-px fractions, the table and tap addresses and the rounding = ~104 '
          'vector instructions per lane (counted in the source, not from the ISA)',
          bound='valu_issue', work=104 * B * h * w)
    del u8, d8

    # C5: bicubic (a=-0.5) warp under rotation + perspective, dense 11x11 - on 4K frames here
    # and on 8K frames below
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()

    def rot_persp(h, w):
        a = np.deg2rad(7.0)
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                      [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy],
                      [0, 0, 1.0]])
        P = np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]])
        return P @ R
    ms = timed_settled(ctx, lambda: ops.warp_perspective_conv2d(src, rot_persp(h, w), (h, w), k11,
                                                        'cubic', out=dst), budget_launches, 5)
    entry('C5-like 4K f32, bicubic warp + dense 11x11, %d frames/launch' % B, B, h, w, ms,
          8 * B * h * w, 2, 'two launches through the workspace; the 11x11 filter is fma-bound',
          bound='valu', work=(2 * 121 + 2 * 16) * B * h * w)
    del src, dst

    h, w, B = 4320, 7680, 4
    src = ctx.to_device(synth_frames(B, h, w, 500))
    dst = ctx.empty((B, h, w), np.float32)
    ms = timed_settled(ctx, lambda: ops.warp_perspective_conv2d(src, rot_persp(h, w), (h, w), k11,
                                                        'cubic', out=dst), budget_launches // 2, 3)
    entry('C5 8K f32, bicubic warp + dense 11x11, %d frames/launch' % B, B, h, w, ms,
          8 * B * h * w, 2, 'two launches through the workspace; the 11x11 filter is fma-bound',
          bound='valu', work=(2 * 121 + 2 * 16) * B * h * w)
    del src, dst

    # C4: 4K uint16 -> float32 frames, undistort (maps) + dense 7x7; 64 frames per GPU
    h, w, B = H4K, W4K, 64
    K, dcoef = camera(h, w)
    k7 = np.random.default_rng(123).random((7, 7))
    k7 /= k7.sum()
    f16 = np.round(synth_frames(16, h, w, 400) * 4095).astype(np.uint16)
    u16 = ctx.to_device(np.concatenate([np.roll(f16, 29 * i, axis=1) for i in range(B // 16)]))
    dmx, dmy = ops.build_undistort_map(K, dcoef, K, h, w, ctx=ctx, device=True)
    dst = placed((B, h, w), np.float32, lambda d: ops.remap_conv2d(u16, dmx, dmy, k7, out=d))
    ms = timed_settled(ctx, lambda: ops.remap_conv2d(u16, dmx, dmy, k7, out=dst), budget_launches // 2, 3)
    entry('C4 4K uint16 -> float32, undistort (maps) + dense 7x7, %d frames/launch (kernel only)'
          % B, B, h, w, ms, (6 * B + 8) * h * w, 1,
          'streams 3.2 GB while its SIMDs issue vector instructions ~80 % of the launch (98 packed fmas for the 49 '
          'taps x 4 pixels + ~120 for sampling, unpacking and coefficient restores per row step): priced against '
          'HBM, bound by vector issue', issue='c4')
    # ... with a 7x7 GAUSSIAN in place of the random kernel (how the reference obtains its smoothing kernels:
    # scipy.ndimage.gaussian_filter): an outer product - the library takes its separable 7 + 7 loop, since round 6 for
    # uint16 frames too (knob sep_u16)
    g7 = np.exp(-0.5 * np.arange(-3, 4) ** 2)
    g7 /= g7.sum()
    ms = timed_settled(ctx, lambda: ops.remap_conv2d(u16, dmx, dmy, np.outer(g7, g7), out=dst), budget_launches // 2, 3)
    entry('... with a 7x7 Gaussian (outer product: the separable 7 + 7 loop)', B, h, w, ms, (6 * B + 8) * h * w, 1)
    del u16, dst
    # the same chain host -> host through page-locked buffers (PCIe-inclusive; never `value`)
    try:
        from imgprocessor_amd.sharding import FramePipeline
        N = 24
        pipe = FramePipeline(ctx.device_id, 3)
        maps = {id(c): ops.build_undistort_map(K, dcoef, K, h, w, ctx=c, device=True)
                for c in pipe.contexts}
        fin = pipe.pinned_empty((N, h, w), np.uint16)
        fout = pipe.pinned_empty((N, h, w), np.float32)
        fin[...] = f16[:1]

        def fn(c, d, o):
            mx, my = maps[id(c)]
            ops.remap_conv2d(d, mx, my, k7, out=o)
        pipe.run(fin[:3], fout[:3], fn)
        t0 = time.perf_counter()
        pipe.run(fin, fout, fn)
        ms_host = (time.perf_counter() - t0) * 1e3
        e = {'workload': 'C4 the same chain host -> host (page-locked buffers, 3 overlapped '
                         'workers on one GPU), PCIe-inclusive', 'frames': N,
             'ms': round(ms_host, 3), 'Mpix_s': round(N * h * w / ms_host / 1e3, 1),
             'pcie_bytes': int(6 * N * h * w), 'launches': N}
        out.append(e)
        del fin, fout
    except Exception as ex:  # noqa: BLE001 - a report line, not the measurement
        out.append({'workload': 'C4 host -> host', 'error': repr(ex)})
    return out


def collectives(world, rank):
    """(barrier, max_over_ranks, gather_over_ranks) - the ONLY cross-rank traffic of this benchmark: timing
    plumbing over gloo (frames are independent: no data-path collective); at world = 1 plain functions"""
    if world <= 1:
        return (lambda: None), (lambda v: v), (lambda v: [v])
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)

    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def gather_over_ranks(v):
        mine = torch.tensor([v], dtype=torch.float64)
        every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(every, mine)
        return [float(t[0]) for t in every]
    return dist.barrier, max_over_ranks, gather_over_ranks


def e2e_plan(world, frames_per_rank, h, w):
    """the end-to-end leg at N > 1 (ingest, not HBM, limits the sharded batch): every rank
    streams its OWN block of uint16 frames host -> device -> host (float32 results) - BASELINE's C4 chain -
    through page-locked buffers; bytes over PCIe per rank and in total"""
    per_rank = (2 + 4) * frames_per_rank * h * w
    return {'frames_per_rank': frames_per_rank, 'frames_total': world * frames_per_rank,
            'pcie_bytes_per_rank': per_rank, 'pcie_bytes': world * per_rank,
            'workload': 'C4 chain host -> host: %d x %dx%d uint16 frames per rank in, undistort (maps) + dense 7x7, '
                        'float32 frames out; FramePipeline (3 overlapped workers per GPU, page-locked buffers, '
                        'threads pinned to the GPU\'s NUMA node)' % (frames_per_rank, w, h)}


def end_to_end_c4(ia, ops, device, rank, h, w, n_frames, barrier):
    """milliseconds this rank takes to stream n_frames uint16 frames host -> device -> host through the C4
    chain; barrier before and after (the caller takes the MAX over ranks).  Every rank reaches BOTH barriers
    whatever happens to it - a rank whose setup or run raised reports (0, False, reason) instead of leaving
    the others waiting."""
    from imgprocessor_amd.sharding import FramePipeline, numa_cpus_of_device
    pipe = fin = fout = fn = cpus = None
    err = None
    try:
        K, dcoef = camera(h, w)
        k7 = np.random.default_rng(123).random((7, 7))
        k7 /= k7.sum()
        cpus = numa_cpus_of_device(device)
        pipe = FramePipeline(device, 3, cpus=cpus)
        maps = {id(c): ops.build_undistort_map(K, dcoef, K, h, w, ctx=c, device=True) for c in pipe.contexts}
        fin = pipe.pinned_empty((n_frames, h, w), np.uint16)
        fout = pipe.pinned_empty((n_frames, h, w), np.float32)
        one = np.round(synth_frames(1, h, w, 400 + rank)[0] * 4095).astype(np.uint16)
        for i in range(n_frames):
            fin[i] = np.roll(one, 29 * i, axis=1)

        def fn(c, d, o):
            mx, my = maps[id(c)]
            ops.remap_conv2d(d, mx, my, k7, out=o)
        pipe.run(fin[:3], fout[:3], fn)   # buffers, maps and code objects in place
    except Exception as ex:  # noqa: BLE001 - reported in the line; the barriers below must still be reached
        err = repr(ex)
    barrier()
    ms = 0.0
    if err is None:
        try:
            t0 = time.perf_counter()
            pipe.run(fin, fout, fn)
            ms = (time.perf_counter() - t0) * 1e3
        except Exception as ex:  # noqa: BLE001
            err = repr(ex)
    barrier()
    ok = err is None and bool(np.isfinite(fout[n_frames - 1]).all() and float(fout[n_frames - 1].max()) > 0)
    del fin, fout
    return ms, ok, ((sorted(cpus)[:1] + sorted(cpus)[-1:]) if cpus else None) if err is None else err


def self_launch_cmd(n_gpus, argv, port):
    """the command a bare `python bench.py --gpus N` (N > 1, no launcher around it) runs as a child:
    the driver's own launch line, one rank per GPU, rendezvous on 127.0.0.1"""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
            '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: ~0.3 s of timed device work
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    # 64 frames per launch = the per-GPU batch of BASELINE.json's sharded configuration (512
    # frames over 8 GPUs); 4.2 GB of the 288 GB in + out.  --batch 128 is reported as an extra
    # line of other_configs.
    ap.add_argument('--batch', type=int, default=64, help='4K frames per step per GPU')
    ap.add_argument('--variant', default='fused_map',
                    choices=['fused_map', 'fused_analytic', 'two_kernel', 'two_kernel_analytic'])
    ap.add_argument('--height', type=int, default=H4K)
    ap.add_argument('--width', type=int, default=W4K)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--placements', type=int, default=0,
                    help="candidate allocations per large block of the context pool (0 = the library's "
                         'default: IMGPROC_HIP_PLACE, 1 = every allocation as it comes; 2 = the better of two)')
    ap.add_argument('--no-settle', action='store_true',
                    help='skip the untimed clock-settling launches of the setup phase')
    ap.add_argument('--e2e-frames', type=int, default=64,
                    help='N > 1: uint16 frames every rank streams host -> device -> host in the end-to-end leg '
                         '(0 skips it)')
    ap.add_argument('--no-configs', action='store_true',
                    help='skip the other_configs leg (BASELINE configurations C2..C5)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # --gpus N without a launcher: this process becomes the launcher.  It has not touched the
        # GPU (no HIP call, not even a device count), starts the N ranks as CHILD processes through
        # torch.distributed.run - one rank per GPU, rank r on device r modulo the devices the box
        # has - and ends with their exit code; rank 0's JSON line passes through.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        sys.exit(subprocess.call(self_launch_cmd(args.gpus, sys.argv[1:], port)))

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and rank == 0:
        print('bench.py: --gpus %d but WORLD_SIZE=%d: the line below is a %d-rank measurement'
              % (args.gpus, world, world), file=sys.stderr)
    dist_on = world > 1
    barrier, max_over_ranks, gather_over_ranks = collectives(world, rank)
    if dist_on:
        import torch.distributed as dist

    import imgprocessor_amd as ia
    from imgprocessor_amd import ops
    ndev = ia.device_count()
    ctx = ia.Context(local_rank % max(ndev, 1))
    if args.placements > 0:
        ctx._place_n = min(2, args.placements)
    h, w, B = args.height, args.width, args.batch
    K, dcoef = camera(h, w)
    k5 = gauss5()

    # per-rank batch: distinct frames, resident in HBM before the timed region
    frames = synth_frames(B, h, w, seed0=rank * B)
    d_src = ctx.empty((B, h, w), np.float32)
    d_dst = ctx.empty((B, h, w), np.float32)
    d_tmp = ctx.empty((B, h, w), np.float32) if args.variant.startswith('two') else None
    # WHERE in the device memory a 2 GB batch buffer lies moves this streaming kernel by up to 10 %
    # on this part - a property of physical regions of the HBM that no caller can choose
    # (profiles/r05_micro.txt).  The buffers above are taken as the driver
    # hands them out (the library's default since round 5; --placements 2 / IMGPROC_HIP_PLACE=2:
    # the better of two allocations, logged below).  What class this run drew is reported, not
    # chosen: the pool's strip-shaped probe (a plain 3x3 filter from one half of the block into
    # the other) on both buffers before the frames go in - ~0.377 ms: the fast class of a 2 GiB
    # block, ~0.39: the middle one, ~0.425: the slow one.
    probe_ms = {'source': round(ctx._probe_block(d_src.ptr, d_src.nbytes), 4),
                'result': round(ctx._probe_block(d_dst.ptr, d_dst.nbytes), 4)}
    d_src.set(frames)
    dmx, dmy = ops.build_undistort_map(K, dcoef, K, h, w, ctx=ctx, device=True)
    ctx.synchronize()
    placement = {'by': 'none: every buffer as the driver hands it out (library default)' if ctx._place_n <= 1 else
                       'Context block pool, opt-in: the better of two allocations per block >= 256 MiB by a '
                       'strip-shaped 3x3 probe, at most one extra block held',
                 'candidates': ctx._place_n, 'blocks': list(ctx.placement_log),
                 'class_probe_ms': probe_ms,
                 'class_probe_note': '3x3 filter from one half of the block into the other; 2 GiB blocks: ~0.377 fast, '
                                     '~0.39 middle, ~0.425 slow class (the headline launch on pairs of them: '
                                     '0.95 / 1.04 / 1.09 ms)'}

    px = B * h * w
    # bytes per launch.  `compulsory`: what must cross the HBM interface - source and result
    # once per frame, the map pair ONCE per launch (the frames of a launch share it through
    # L2 / MALL; the PMC traffic confirms it).  `literal`: SURVEY section 8(d)'s per-pixel figure
    # that charges the maps to every frame - kept as a second, clearly named number.
    if args.variant == 'fused_map':
        def step():
            ops.remap_conv2d(d_src, dmx, dmy, k5, out=d_dst)
        literal_px, compulsory, launches = 16, (8 * B + 8) * h * w, 1
        kname = 'wave_stencil_kernel<SampleRowSrc<float,linear,MapCoord>,5>'
    elif args.variant == 'fused_analytic':
        def step():
            ops.undistort_conv2d(d_src, K, dcoef, K, k5, out=d_dst)
        literal_px, compulsory, launches = 8, 8 * B * h * w, 1
        kname = 'wave_stencil_kernel<SampleRowSrc<float,linear,UndistortCoord>,5>'
    elif args.variant == 'two_kernel':
        def step():
            ops.remap(d_src, dmx, dmy, out=d_tmp)
            ops.conv2d(d_tmp, k5, out=d_dst)
        literal_px, compulsory, launches = 24, (16 * B + 8) * h * w, 2
        kname = 'remap_kernel<float,float,linear,MapCoord> + wave_stencil_kernel<LoadRowSrc,5>'
    else:
        def step():
            ops.undistort(d_src, K, dcoef, K, out=d_tmp)
            ops.conv2d(d_tmp, k5, out=d_dst)
        literal_px, compulsory, launches = 16, 16 * B * h * w, 2
        kname = 'remap_kernel<float,float,linear,UndistortCoord> + wave_stencil_kernel<LoadRowSrc,5>'

    # which loop the library takes for this call: the bench's 5x5 is outer(g, g) - an exact outer product goes
    # to the separable 5 + 5 loop (knob rank1_sep, on by default; csrc/fused.hip::rank1_chain)
    routed0 = ctx.get_tuning('rank1_routed')
    # the very first launches of this process, one event pair each: the clock ramp of a GPU that idled while
    # the host built the frames, as a driver-side record sees it (config.first_launch_ms)
    n_first = 30
    fev = [ctx.event() for _ in range(n_first + 1)]
    fev[0].record()
    for i in range(n_first):
        step()
        fev[i + 1].record()
    ctx.synchronize()
    first_ms = [round(fev[i].elapsed_ms(fev[i + 1]), 4) for i in range(n_first)]
    del fev
    sep_route = ctx.get_tuning('rank1_routed') > routed0
    if sep_route:
        kname_dense = kname
        kname = kname.replace('wave_stencil_kernel', 'wave_sep_kernel').replace(',5>', ',5+5>')

    # What a run WITHOUT any preparation sees (clocks still ramping up from idle): the first
    # min(steps, 20) steps after the driver's warm-up count, reported as no_settle_*.
    ns_steps = min(args.steps, 20)
    ns_ms = timed(ctx, step, ns_steps, args.warmup)

    # Setup, not measurement: let the GPU clocks settle (reported as config.clock_settle_launches;
    # --no-settle skips it and the no_settle_* numbers above are then the whole story).
    settle = 0 if args.no_settle else max(20, 4800 // B)
    for _ in range(settle):
        step()
    ctx.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.synchronize()
    barrier()
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        step()
    e1.record()
    ctx.synchronize()
    el = time.perf_counter() - t0
    barrier()
    per_rank_ms = gather_over_ranks(el / args.steps * 1e3)
    el = max_over_ranks(el)
    ev_ms = e0.elapsed_ms(e1)  # HIP events on the stream the kernels ran on

    # the dense 5x5 loop on the same buffers, same process (knob rank1_sep = 0): what the call cost before
    # round 6 routed outer products to the separable loop - reported beside the line, never as `value`
    dense_ms = None
    if sep_route and rank == 0:
        old_knob = ctx.set_tuning(rank1_sep=0)
        try:
            dense_ms = timed(ctx, step, max(20, min(args.steps, 60)), 5)
        finally:
            ctx.set_tuning(**old_knob)

    # SURVEY section 8(d) asks for the MEDIAN of >= 20 iterations: a second, untimed-for-`value`
    # pass with one event pair per step (the contract's `value` / `ms_per_step` stay the
    # whole-region figures above)
    n_med = max(20, min(args.steps, 100))
    evs = [ctx.event() for _ in range(n_med + 1)]
    evs[0].record()
    for i in range(n_med):
        step()
        evs[i + 1].record()
    ctx.synchronize()
    per_step = sorted(evs[i].elapsed_ms(evs[i + 1]) for i in range(n_med))
    med_ms = per_step[n_med // 2]
    del evs

    # N > 1: the end-to-end half (every rank: its own uint16 frames host -> device -> host).  The kernel-only
    # line above scales trivially; this is the leg SURVEY section 8(e) names as the real limiter.
    e2e = None
    if dist_on and args.e2e_frames > 0 and (h, w) == (H4K, W4K):
        ms_e2e, ok_e2e, cpu_span = end_to_end_c4(ia, ops, ctx.device_id, rank, h, w, args.e2e_frames, barrier)
        failed = 0.0 if ok_e2e else 1.0
        e2e_ms = gather_over_ranks(ms_e2e)
        e2e_failed = max_over_ranks(failed)
        plan = e2e_plan(world, args.e2e_frames, h, w)
        worst = max(e2e_ms)
        e2e = dict(plan, per_rank_ms=[round(v, 2) for v in e2e_ms],
                   Gpix_s_aggregate=round(plan['frames_total'] * h * w / worst / 1e6, 2) if worst > 0 else None,
                   GB_s_pcie_aggregate=round(plan['pcie_bytes'] / worst / 1e6, 2) if worst > 0 else None,
                   numa_cpus_rank0=cpu_span, ok=not e2e_failed,
                   note='barrier, every rank streams its block, barrier; MAX over ranks; PCIe-inclusive, never `value`')

    if rank == 0:
        value = world * px * args.steps / el / 1e6
        step_s = ev_ms * 1e-3 / args.steps
        ach = compulsory / step_s / 1e9
        traffic, traffic_src = pmc_traffic(args.variant, B, h, w)
        # the result the timed launches left in d_dst, against the oracle (first and last frame of
        # the batch; the oracle is the checker here, never the thing measured)
        checked = None
        if not args.no_cpu:
            from oracle import oracle as orc
            orc.build()
            orc.set_threads(max(1, min(usable_cores(), orc.max_threads())))
            mxh, myh = dmx.get(), dmy.get()
            worst, frames_checked = 0.0, sorted({0, B - 1})
            for i in frames_checked:
                got = d_dst.frame(i).get()
                want = orc.conv2d(orc.remap(frames[i], mxh, myh, orc.LINEAR, orc.CONSTANT, 0.0),
                                  k5, 'reflect')
                floor = 1e-3 * float(np.max(np.abs(want)))
                worst = max(worst, float(np.max(np.abs(got - want) / np.maximum(np.abs(want), floor))))
            orc.set_threads(1)
            checked = {'max_rel_err': float('%.3g' % worst), 'frames': frames_checked,
                       'tolerance': 1e-5,
                       'note': 'result of the timed launches vs oracle/oracle.c (map-based '
                               'bilinear remap + 5x5, double accumulation); relative to '
                               'max(|ref|, 1e-3 max|ref|)'}
        # what a plain device copy of the same buffers reaches in this run (read + write): the
        # practical ceiling of a streaming kernel on this box, next to the 8 TB/s of the guide
        # (after the oracle check: it overwrites the result batch)
        copy_ms = timed(ctx, lambda: d_dst.copy_from(d_src), 20, 3)
        copy_gbs = 2.0 * d_src.nbytes / (copy_ms * 1e-3) / 1e9
        # for reference: the same launches on a pair of buffers taken as the driver hands them out
        # (placement off), allocated now - what a caller that bypasses the pool's choice would see
        if args.variant.startswith('fused') and ctx._place_n > 1:
            keep_n, ctx._place_n = ctx._place_n, 1
            try:
                s_raw = ctx.empty((B, h, w), np.float32)
                d_raw = ctx.empty((B, h, w), np.float32)
                s_raw.copy_from(d_src)
                if args.variant == 'fused_map':
                    raw_ms = timed(ctx, lambda: ops.remap_conv2d(s_raw, dmx, dmy, k5, out=d_raw), 20, 5)
                else:
                    raw_ms = timed(ctx, lambda: ops.undistort_conv2d(s_raw, K, dcoef, K, k5, out=d_raw), 20, 5)
                placement['unplaced_pair_ms_per_step'] = round(raw_ms, 4)
                del s_raw, d_raw
                ctx.trim()
            except MemoryError:
                pass
            finally:
                ctx._place_n = keep_n
        line = {
            'metric': 'Mpix/s undistort+5x5 filter, 4K f32',
            'value': round(value, 1), 'unit': 'Mpix/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(el / args.steps * 1e3, 4),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%dx%d float32, LensDistortion undistort (bilinear, constant '
                                   'border) + 5x5 Gaussian (reflect), %d frames/step/GPU, '
                                   'variant=%s' % (w, h, B, args.variant),
                       'frames_per_step_per_gpu': B, 'variant': args.variant,
                       'path': ('separable 5 + 5 loop: the 5x5 kernel is an exact outer product (library default, '
                                'knob rank1_sep)' if sep_route else 'dense 5x5 loop'),
                       'probe_src_ms': probe_ms['source'], 'probe_dst_ms': probe_ms['result'],
                       'first_launch_ms': first_ms,
                       'per_rank_ms_per_step': [round(v, 4) for v in per_rank_ms],
                       'clock_settle_launches': settle, 'buffer_placement': placement,
                       'no_settle_ms_per_step': round(ns_ms, 4),
                       'no_settle_value': round(world * px / ns_ms / 1e3, 1),
                       'no_settle_steps': ns_steps,
                       'sharding': 'independent frames, %d rank(s), no collective' % world},
            'roofline': {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                         'traffic': traffic, 'traffic_source': traffic_src,
                         'checked_vs_oracle': checked,
                         'median_step_ms': round(med_ms, 4),
                         'median_frac': round(compulsory / (med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         'median_over_steps': n_med,
                         'bytes_model': 'compulsory: source + result per frame, map pair once '
                                        'per launch',
                         'compulsory_bytes_per_launch': compulsory // launches
                         if launches == 1 else None,
                         'compulsory_bytes_per_step': compulsory,
                         'limiter': 'the gather + strip-store stream of the marching-strip shape (a gather COPY in that '
                                    'shape - same strips, same order, same stores, no arithmetic - runs 0.80 - 0.97 ms '
                                    'by the box, tools/sector_micro.hip); the filter is worth 2 - 3 % of the launch '
                                    '(dense 25 taps -> separable 5 + 5: profiles/r06_micro.txt); HBM is the roofline the '
                                    'compulsory bytes are priced against, not what saturates',
                         'kernel': kname, 'launches_per_step': launches,
                         'dense_loop': ({'kernel': kname_dense, 'ms_per_step': round(dense_ms, 4),
                                         'frac': round(compulsory / (dense_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                         'note': 'the same call with rank1_sep = 0, same buffers, same process'}
                                        if dense_ms else None),
                         'avg_step_ms_hip_events': round(ev_ms / args.steps, 4),
                         'literal_survey_8d': {
                             'bytes_per_px': literal_px,
                             'achieved': round(literal_px * px / step_s / 1e9, 1),
                             'frac': round(literal_px * px / step_s / 1e9 / HBM_PEAK_GBS, 4),
                             'note': 'charges the map pair to every frame; NOT the roofline '
                                     'fraction'},
                         'no_settle_frac': round(compulsory / (ns_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         'device_copy': {
                             'achieved': round(copy_gbs, 1), 'unit': 'GB/s',
                             'kernel_vs_copy': round(ach / copy_gbs, 4),
                             'note': 'hipMemcpyAsync device-to-device of the source batch into '
                                     'the result batch in this run, read + write bytes; context '
                                     'for `peak`, not a substitute for it'}},
        }
        if e2e is not None:
            line['end_to_end'] = e2e
        if world == 1 and not args.no_configs and (h, w) == (H4K, W4K):
            del d_src, d_dst, d_tmp
            extra = []
            if args.variant == 'fused_map' and B != 128:
                B2 = 128
                s2 = ctx.to_device(synth_frames(B2, h, w, seed0=7))
                o2 = ctx.empty((B2, h, w), np.float32)
                ms2 = timed_settled(ctx, lambda: ops.remap_conv2d(s2, dmx, dmy, k5, out=o2), 40, 5)
                c2 = (8 * B2 + 8) * h * w
                extra.append({'workload': 'headline at %d frames/launch' % B2, 'frames': B2,
                              'ms': round(ms2, 4), 'Mpix_s': round(B2 * h * w / ms2 / 1e3, 1),
                              'compulsory_bytes': c2,
                              'frac_compulsory': round(c2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              'launches': 1})
                # the same chain with the camera matrix LensDistortion.py:350-353 really asks for -
                # cv2.getOptimalNewCameraMatrix(alpha = 1): every source pixel kept, 2.8 % of the
                # output outside the source along the rim (BASELINE's configuration says newK = K,
                # where no footprint touches the border)
                from imgprocessor_amd.utils.geometry import getOptimalNewCameraMatrix
                nK = getOptimalNewCameraMatrix(K, dcoef, (w, h), 1.0)[0]
                ax, ay = ops.build_undistort_map(K, dcoef, nK, h, w, ctx=ctx, device=True)
                ms3 = timed_settled(ctx, lambda: ops.remap_conv2d(s2, ax, ay, k5, out=o2), 40, 5)
                extra.append({'workload': "headline with the reference's own camera matrix "
                                          '(getOptimalNewCameraMatrix, alpha = 1), %d frames/launch' % B2,
                              'frames': B2, 'ms': round(ms3, 4), 'Mpix_s': round(B2 * h * w / ms3 / 1e3, 1),
                              'compulsory_bytes': c2,
                              'frac_compulsory': round(c2 / (ms3 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                              'launches': 1,
                              'note': 'footprints on the source border blended from the taps in registers '
                                      '(wave_pipe.hpp::border_blend); through sample() this ran 45 % '
                                      'over the line above'})
                del s2, o2, ax, ay
                # ... and at 256 frames per launch (8.5 GB in, 8.5 GB out of the 288 GB): the per-frame cost of this
                # launch shape falls with the frames per launch until ~192 (tools/batch_sweep.py: 15.4 us per frame
                # at 64, 14.8 at 128, 14.1 at 192 - 256 on a slow-class box); `value` stays on BASELINE's per-GPU
                # batch of 64
                try:
                    B3 = 256
                    s3 = ctx.empty((B3, h, w), np.float32)
                    part = ctx.to_device(synth_frames(64, h, w, seed0=11))
                    for i0 in range(0, B3, 64):
                        ctx._check(ctx._lib.ipa_memcpy_d2d(ctx.handle, ctypes.c_void_p(s3.ptr.value + i0 * h * w * 4),
                                                          part.ptr, part.nbytes), 'memcpy_d2d')
                    del part
                    o3 = ctx.empty((B3, h, w), np.float32)
                    ms4 = timed_settled(ctx, lambda: ops.remap_conv2d(s3, dmx, dmy, k5, out=o3), 20, 3)
                    c3 = (8 * B3 + 8) * h * w
                    extra.append({'workload': 'headline at %d frames/launch' % B3, 'frames': B3,
                                  'ms': round(ms4, 4), 'Mpix_s': round(B3 * h * w / ms4 / 1e3, 1),
                                  'compulsory_bytes': c3,
                                  'frac_compulsory': round(c3 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                  'launches': 1})
                    del s3, o3
                    ctx.trim()
                except MemoryError:
                    pass
            line['other_configs'] = extra + other_configs(ctx, ia, ops)
        if world == 1 and not args.no_cpu:
            line['cpu_baseline'] = cpu_baseline(h, w, K, dcoef, k5)
        # a result that is not the oracle's is not a measurement: say so in the line and fail
        bad = checked is not None and not (checked['max_rel_err'] <= checked['tolerance'])
        if bad:
            line['invalid'] = ('result of the timed launches differs from the oracle: max relative '
                               'error %.3g > %g' % (checked['max_rel_err'], checked['tolerance']))
        print(json.dumps(line))
        if bad:
            if dist_on:
                dist.destroy_process_group()
            sys.exit(1)
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
