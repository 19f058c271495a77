#!/usr/bin/env python3
"""Memory-safety probe (GPU box): every array a kernel gets ENDS EXACTLY WHERE ITS OWN ALLOCATION ENDS (a block of a
whole number of 2 MiB, the array in its tail), so a kernel that reads or writes past an array - a map row past the
frame, a vector load over a ragged row end, a store of a masked-off lane - takes a page fault instead of quietly
touching a neighbour.  (Source frames are read through range-checked buffer descriptors; maps, coordinate tables,
kernels' outputs, masks and filter inputs are raw pointers.)  Round 6: the hand-scheduled loops' slow path read up to
three map rows past a strip that ends with the frame - found only because a test's map happened to end with its slab.

A fault aborts the process, so the cases run in child processes:
    python tools/guard_probe.py            # driver: runs every case, restarts after a fault, lists what faulted
    python tools/guard_probe.py --from N   # child: cases N, N + 1, ... in this process, "CASE i name" before each
Ragged sizes on purpose (rows that are no whole vectors, heights that are no whole strips or blocks)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
SLAB = 2 << 20


def build_cases():
    import imgprocessor_amd as ia
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import DeviceArray
    ctx = ia.default_context(0)
    keep = []

    def tail(arr=None, shape=None, dtype=None):
        """a DeviceArray whose last byte is the last byte of a fresh allocation of whole 2 MiB slabs"""
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            shape, dtype = arr.shape, arr.dtype
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        block = (nbytes + SLAB - 1) // SLAB * SLAB
        p = ctx._alloc_raw(block)
        keep.append(p)
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.dtype, v.nbytes = ctx, tuple(shape), np.dtype(dtype), nbytes
        v.ptr = C.c_void_p(p.value + block - nbytes)
        v._owner = False
        if arr is not None:
            v.set(arr)
        return v

    rng = np.random.default_rng(7)
    cases = []

    def frames(n, h, w, dt):
        a = rng.random((n, h, w))
        return a.astype(np.float32) if dt == np.float32 else np.round(a * (255 if dt == np.uint8 else 4095)).astype(dt)

    def maps(h, w, kind):
        y, x = np.mgrid[0:h, 0:w].astype(np.float64)
        if kind == 'shift':
            return (x - 6.3).astype(np.float32), (y - 4.6).astype(np.float32)
        a = np.deg2rad(4.0)
        cx, cy = w / 2.0, h / 2.0
        return ((np.cos(a) * (x - cx) - np.sin(a) * (y - cy)) * 1.03 + cx).astype(np.float32), \
               ((np.sin(a) * (x - cx) + np.cos(a) * (y - cy)) * 1.03 + cy).astype(np.float32)

    M = np.array([[1.01, 0.01, -5.3], [-0.008, 0.99, 3.1], [1e-5, -2e-5, 1.0]])
    shapes = [(97, 333), (61, 257), (130, 512), (33, 1030), (146, 1999), (290, 3844)]
    for (h, w) in shapes:
        Kc = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
        dist = np.array([-0.15, 0.03, 1e-3, -5e-4, 0.0])
        for n in ((1, 4, 7, 8) if w < 1500 else (4, 16, 20)):   # (16 / 20 frames: chunked frame groups, short-strip tails)
            for dt in (np.float32, np.uint16, np.uint8):
                for border in ('constant', 'replicate', 'wrap'):
                    for kind in ('shift', 'rot'):
                        mx, my = maps(h, w, kind)
                        interps = ('linear', 'cubic', 'lanczos4', 'nearest', 'linear_cv_q5') if (n, border, kind) in (
                            (4, 'constant', 'rot'), (1, 'replicate', 'shift')) else ('linear',)
                        for interp in interps:
                            tag = '%s %dx%d n=%d %s %s %s' % (np.dtype(dt).name, h, w, n, interp, border, kind)

                            def remap_case(h=h, w=w, n=n, dt=dt, border=border, interp=interp, mx=mx, my=my, odt=None):
                                d = tail(frames(n, h, w, dt))
                                out = tail(shape=(n, h, w), dtype=odt or dt)
                                ops.remap(d, tail(mx), tail(my), interp, border, 3.0, out_dtype=odt, out=out)
                                ctx.synchronize()
                            cases.append(('remap ' + tag, remap_case))
                            if dt != np.float32:
                                cases.append(('remap -> f32 ' + tag, lambda f=remap_case: f(odt=np.float32)))
                            if kind == 'shift' and dt != np.uint8:
                                def warp_case(h=h, w=w, n=n, dt=dt, border=border, interp=interp):
                                    d = tail(frames(n, h, w, dt))
                                    ops.warp_perspective(d, M, (h, w), interp, border, 3.0, out_dtype=np.float32,
                                                         out=tail(shape=(n, h, w), dtype=np.float32))
                                    ops.undistort(d, Kc, dist, Kc, interp, border, 3.0, out_dtype=np.float32,
                                                  out=tail(shape=(n, h, w), dtype=np.float32))
                                    ctx.synchronize()
                                cases.append(('warp + undistort -> f32 ' + tag, warp_case))
                            if dt != np.uint8 and interp in ('linear', 'cubic', 'linear_cv_q5'):
                                for K in ((3, 5, 7, 9, 11) if (border, kind) == ('constant', 'shift') else (5,)):
                                    def chain_case(h=h, w=w, n=n, dt=dt, border=border, interp=interp, mx=mx, my=my, K=K):
                                        d = tail(frames(n, h, w, dt))
                                        k = rng.random((K, K))
                                        g = rng.random(K) + 0.1
                                        dmx, dmy = tail(mx), tail(my)
                                        ops.remap_conv2d(d, dmx, dmy, k, interp, border, 3.0, 'reflect',
                                                         out=tail(shape=(n, h, w), dtype=np.float32))
                                        ops.remap_conv2d(d, dmx, dmy, np.outer(g, g), interp, border, 3.0, 'constant',
                                                         out=tail(shape=(n, h, w), dtype=np.float32))
                                        if K <= 9:
                                            ops.remap_sepconv2d(d, dmx, dmy, g, g, interp, border, 3.0, 'wrap',
                                                                out=tail(shape=(n, h, w), dtype=np.float32))
                                            ops.warp_perspective_sepconv2d(d, M, (h, w), g, g, interp, border, 3.0, 'nearest',
                                                                           out=tail(shape=(n, h, w), dtype=np.float32))
                                        ops.warp_perspective_conv2d(d, M, (h, w), k, interp, border, 3.0, 'mirror',
                                                                    out=tail(shape=(n, h, w), dtype=np.float32))
                                        ops.undistort_conv2d(d, Kc, dist, Kc, k, interp, border, 3.0, 'reflect',
                                                             out=tail(shape=(n, h, w), dtype=np.float32))
                                        ctx.synchronize()
                                    cases.append(('chains K=%d ' % K + tag, chain_case))
        # the plain filters and stencils on the same ragged shapes
        for n in (1, 3, 4):
            for K in (3, 5, 7, 9, 11, 13):
                for mode in ('reflect', 'constant', 'wrap'):
                    def conv_case(h=h, w=w, n=n, K=K, mode=mode):
                        d = tail(frames(n, h, w, np.float32))
                        g = rng.random(K) + 0.1
                        ops.conv2d(d, rng.random((K, K)), mode, out=tail(shape=(n, h, w), dtype=np.float32))
                        ops.conv2d(d, np.outer(g, g), mode, out=tail(shape=(n, h, w), dtype=np.float32))
                        ops.sepconv2d(d, g, g, mode, out=tail(shape=(n, h, w), dtype=np.float32))
                        ctx.synchronize()
                    cases.append(('filters %dx%d n=%d K=%d %s' % (h, w, n, K, mode), conv_case))

        def stencil_case(h=h, w=w):
            from imgprocessor_amd import filters, interpolate
            img = rng.random((h, w))
            mask = rng.random((h, w)) < 0.2
            for dt in (np.float32, np.float64):
                d = tail(img.astype(dt))
                ops.local_std(d, tail(img.astype(dt)), (5, 3))
                ops.median_threshold(d, 0.1, size=3)
                ops.median_threshold(d, 0.1, size=6)
                ops.nan_max(d, 5)
                ops.masked_mean(tail(img.astype(dt)), tail(mask.astype(np.uint8)), 4, fn='mean')
                ops.masked_mean(tail(img.astype(dt)), tail(mask.astype(np.uint8)), 3)
                ops.resize(d, (h // 2 + 3, w // 3 + 5), 'area')
                ops.resize(d, (h + 7, w + 9), 'linear')
                ops.resize(d, (h + 7, w // 2), 'lanczos4')
                ops.gaussian_filter(d, 1.3)
                holes = img.astype(dt).copy()
                holes[mask] = np.nan
                if dt == np.float64:
                    yy, xx = np.mgrid[-4:5, -4:5]
                    wts = 1.0 / np.maximum(np.hypot(yy, xx), 0.5) ** 2
                    ops.idw_fill(tail(holes), tail(mask.astype(np.uint8)), 4, wts)
            ctx.synchronize()
        cases.append(('stencils %dx%d' % (h, w), stencil_case))
    return cases, ctx, keep


def child(start):
    cases, ctx, keep = build_cases()
    for i in range(start, len(cases)):
        name, fn = cases[i]
        print('CASE %d %s' % (i, name), flush=True)
        try:
            fn()
        except (NotImplementedError, TypeError, ValueError, AttributeError) as ex:
            print('   skipped: %s: %s' % (type(ex).__name__, str(ex)[:100]), flush=True)
        for p in keep:
            ctx._lib.ipa_free(ctx.handle, p)
        del keep[:]
    print('END %d' % len(cases), flush=True)


def driver():
    selftest()
    start, faults, total = 0, [], None
    while True:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--from', str(start)], capture_output=True, text=True)
        last = None
        for line in r.stdout.splitlines():
            if line.startswith('CASE '):
                last = line
            elif line.startswith('END '):
                total = int(line.split()[1])
                last = None
            elif line.startswith('   skipped'):
                print(last, line)
        if last is None:
            break
        i = int(last.split()[1])
        msg = [l for l in r.stderr.splitlines() if 'fault' in l.lower() or 'error' in l.lower()][:2]
        faults.append(last + '   ' + ' | '.join(msg))
        print('FAULT ' + faults[-1], flush=True)
        start = i + 1
    print('done: %s cases, %d faulted' % (total, len(faults)))
    return 1 if faults else 0


def selftest():
    """does the guard bite?  a map view that hangs 4 bytes over the end of its allocation must fault"""
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, ctypes as C; import imgprocessor_amd as ia; "
            "from imgprocessor_amd import ops; from imgprocessor_amd.device import DeviceArray; ctx = ia.default_context(0); "
            "h, w = 97, 333; y, x = np.mgrid[0:h, 0:w].astype(np.float32); p = ctx._alloc_raw(2 << 20); "
            "v = DeviceArray.__new__(DeviceArray); v.ctx, v.shape, v.dtype, v.nbytes = ctx, x.shape, x.dtype, x.nbytes; "
            "v.ptr = C.c_void_p(p.value + (2 << 20) - x.nbytes + 4); v._owner = False; "
            "ops.remap(ctx.to_device(np.zeros((1, h, w), np.float32)), v, ctx.to_device(y)).get(); print('SURVIVED')" % R)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    bites = 'SURVIVED' not in r.stdout
    print('guard self-test: a 4-byte over-read %s' % ('faults (the probe can see over-reads)' if bites else 'SURVIVED - the probe is blind here'))
    return bites


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--selftest':
        sys.exit(0 if selftest() else 1)
    if len(sys.argv) > 2 and sys.argv[1] == '--from':
        child(int(sys.argv[2]))
    else:
        sys.exit(driver())
