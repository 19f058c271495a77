#!/usr/bin/env python3
"""Headline benchmark: Mpix/s of lens-undistort + 5x5 filter on 4K float32 frames.

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path (LensDistortion-style undistort, bilinear,
BORDER_CONSTANT, then a 5x5 Gaussian given as an explicit kernel, 'reflect'
border) over one batch of synthetic 4K (3840x2160) float32 frames (128 by
default: 4.2 GB in, 4.2 GB out) that are already resident in HBM.  The batch
is far larger than the 256 MiB Infinity Cache, so source and destination
really stream from/to HBM.

For N > 1 the driver launches one rank per GPU (torch.distributed.run); frames
are independent, every rank processes its own batch (weak scaling), no
data-path collective; torch.distributed (gloo) is used only for the barrier
and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline`
(dominant kernel, algorithmic bytes / HIP-event time) and `cpu_baseline`
(the oracle C restatement timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H4K, W4K = 2160, 3840
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


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
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC
    passes of this same command (profiles/pmc_summary.json, written by
    profiles/summarize.py: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE).  None if that
    profile does not cover this variant/shape: counters cannot be read from inside the run."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_summary.json')) as f:
            s = json.load(f)
        e = s.get(variant)
        if e and e.get('batch') == batch and e.get('height') == h and e.get('width') == w:
            return e['traffic_bytes_per_launch']
    except (OSError, ValueError, KeyError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: ~0.6 s of timed device work.  Very short runs under-report by ~10 %: the GPU
    # clocks are still ramping up from idle during the first ~100 ms of load (16-frame launches,
    # 20 steps: 0.445 ms/step, 200: 0.407, 2000: 0.403 on the same box)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    # 128 frames = 4.2 GB in + 4.2 GB out of the 288 GB: launches of 16 / 32-64 / 128 frames reach
    # 0.67 / 0.70 / 0.72-0.77 of the HBM peak (the per-launch ramp and drain
    # and the map rows are shared by more frames)
    ap.add_argument('--batch', type=int, default=128, help='4K frames per step per GPU')
    ap.add_argument('--variant', default='fused_map',
                    choices=['fused_map', 'fused_analytic', 'two_kernel', 'two_kernel_analytic'])
    ap.add_argument('--height', type=int, default=H4K)
    ap.add_argument('--width', type=int, default=W4K)
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-settle', action='store_true',
                    help='skip the untimed clock-settling launches of the setup phase')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    dist_on = world > 1
    if dist_on:
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)

        def barrier():
            dist.barrier()

        def max_over_ranks(v):
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t[0])
    else:
        def barrier():
            pass

        def max_over_ranks(v):
            return v

    import imgprocessor_amd as ia
    from imgprocessor_amd import ops
    ndev = ia.device_count()
    ctx = ia.Context(local_rank % max(ndev, 1))
    h, w, B = args.height, args.width, args.batch
    K, dcoef = camera(h, w)
    k5 = gauss5()

    # per-rank batch: distinct frames, resident in HBM before the timed region
    frames = synth_frames(B, h, w, seed0=rank * B)
    d_src = ctx.to_device(frames)
    d_dst = ctx.empty((B, h, w), np.float32)
    d_tmp = ctx.empty((B, h, w), np.float32) if args.variant.startswith('two') else None
    dmx, dmy = ops.build_undistort_map(K, dcoef, K, h, w, ctx=ctx, device=True)
    ctx.synchronize()

    px = B * h * w
    if args.variant == 'fused_map':
        def step():
            ops.remap_conv2d(d_src, dmx, dmy, k5, out=d_dst)
        bytes_per_px, launches, kname = 16, 1, 'wave_stencil_kernel<SampleRowSrc<float,linear,MapCoord>,5>'
    elif args.variant == 'fused_analytic':
        def step():
            ops.undistort_conv2d(d_src, K, dcoef, K, k5, out=d_dst)
        bytes_per_px, launches, kname = 8, 1, 'wave_stencil_kernel<SampleRowSrc<float,linear,UndistortCoord>,5>'
    elif args.variant == 'two_kernel':
        def step():
            ops.remap(d_src, dmx, dmy, out=d_tmp)
            ops.conv2d(d_tmp, k5, out=d_dst)
        bytes_per_px, launches, kname = 24, 2, 'remap_kernel<float,float,linear,MapCoord> + wave_stencil_kernel<LoadRowSrc,5>'
    else:
        def step():
            ops.undistort(d_src, K, dcoef, K, out=d_tmp)
            ops.conv2d(d_tmp, k5, out=d_dst)
        bytes_per_px, launches, kname = 16, 2, 'remap_kernel<float,float,linear,UndistortCoord> + wave_stencil_kernel<LoadRowSrc,5>'

    # Setup, not measurement: let the GPU clocks settle.  From idle they ramp up over the first
    # ~100 ms of load; a short run (e.g. --steps 20 --warmup 3) would otherwise time the ramp
    # (0.445 vs 0.403 ms/step).  Reported as config.clock_settle_launches.
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
    el = max_over_ranks(el)
    ev_ms = e0.elapsed_ms(e1)  # HIP events on the stream the kernels ran on

    if rank == 0:
        value = world * px * args.steps / el / 1e6
        ach = bytes_per_px * px * args.steps / (ev_ms * 1e-3) / 1e9
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
                       'clock_settle_launches': settle,
                       'sharding': 'independent frames, %d rank(s), no collective' % world},
            'roofline': {'bound': 'hbm', 'achieved': round(ach, 1), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4),
                         'traffic': pmc_traffic(args.variant, B, h, w),
                         'algorithmic_bytes_per_launch': bytes_per_px * px // launches
                         if launches == 1 else None,
                         'kernel': kname, 'algorithmic_bytes_per_px': bytes_per_px,
                         'launches_per_step': launches,
                         'avg_step_ms_hip_events': round(ev_ms / args.steps, 4)},
        }
        if world == 1 and not args.no_cpu:
            line['cpu_baseline'] = cpu_baseline(h, w, K, dcoef, k5)
        print(json.dumps(line))
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
