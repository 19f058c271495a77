"""A/B timing of BUILDS of the library (make VARIANT=x DEFS=...) inside ONE process: the package
is imported once per build under its own name (each with its own library handle, context and
stream), the calls are alternated round by round, so the builds share the box's clock state.
The 4K headline (map-based undistort + 5x5), the plain 5x5 filter and a device copy; the fused
result of every build is compared bit for bit with the first one.  GPU box only.

    python tools/ab_libs.py [--batch 64] [--rounds 3] [--what fused5,conv5] default nopipe ...
    (a name may carry knobs: wpb8:strip_h=48)
"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_build(alias, libname):
    path = os.path.join(ROOT, 'imgprocessor_amd')
    os.environ['IMGPROC_HIP_LIB'] = os.path.join(
        path, 'libimgproc_hip.so' if libname == 'default' else 'libimgproc_hip_%s.so' % libname)
    spec = importlib.util.spec_from_file_location(alias, os.path.join(path, '__init__.py'),
                                                  submodule_search_locations=[path])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    importlib.import_module(alias + '._lib').lib()  # bind the library now
    return mod


def timeit(ctx, fn, n, warm):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    ctx.synchronize()
    return e0.elapsed_ms(e1) / n


def main():
    args = sys.argv[1:]
    batch, rounds, names, what = 64, 3, [], ['fused5', 'conv5', 'copy']
    while args:
        a = args.pop(0)
        if a == '--batch':
            batch = int(args.pop(0))
        elif a == '--rounds':
            rounds = int(args.pop(0))
        elif a == '--what':
            what = args.pop(0).split(',')
        else:
            names.append(a)
    names = names or ['default']
    h, w = 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    g7 = np.exp(-0.5 * (np.arange(-3, 4) / 1.5) ** 2)
    k7 = np.outer(g7, g7) / g7.sum() ** 2
    quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
    rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    A, rhs = [], []
    for (x, y), (u, v) in zip(rect, quad):   # rectangle (destination) -> quad (source)
        A += [[x, y, 1, 0, 0, 0, -u * x, -u * y], [0, 0, 0, x, y, 1, -v * x, -v * y]]
        rhs += [u, v]
    Hq = np.append(np.linalg.solve(np.array(A), np.array(rhs)), 1.0).reshape(3, 3)
    g9 = np.exp(-0.5 * np.arange(-4, 5) ** 2)
    g9 /= g9.sum()
    k7r = np.random.default_rng(123).random((7, 7))
    k7r /= k7r.sum()
    k9 = np.random.default_rng(99).random((9, 9))
    k9 /= k9.sum()
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()
    def rot(deg):
        a = np.deg2rad(deg)
        cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
        R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                      [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy], [0, 0, 1.0]])
        return np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]]) @ R
    rng = np.random.default_rng(0)
    one = rng.random((16, h, w), dtype=np.float32)
    host = np.concatenate([one] * (batch // 16)) if batch >= 16 else one[:batch]
    builds = []
    libs = {}
    for i, spec in enumerate(names):
        lib, _, kv = spec.partition(':')
        knobs = {k: int(v) for k, v in (x.split('=') for x in kv.split(',') if x)}
        if lib not in libs:
            libs[lib] = load_build('ia_build_%d' % i, lib)
        mod = libs[lib]
        ops = importlib.import_module(mod.__name__ + '.ops')
        ctx = mod.default_context(0)
        if 'src' not in mod.__dict__:
            if not builds:
                mod.src = ctx.to_device(host)
                mod.dst = ctx.empty((batch, h, w), np.float32)
                mod.dmx, mod.dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
                mod.u16 = ctx.to_device(np.round(host * 4095).astype(np.uint16)) if ('c4' in what or 'lz16' in what or 'lz16r' in what) else None
                mod.d16 = ctx.empty((batch, h, w), np.uint16) if mod.u16 is not None else None
            else:
                # the SAME device buffers for every build (one process, one device: a pointer of
                # the first build's allocator is valid in the others) - where a buffer lands in
                # physical memory moves a streaming kernel by several per cent
                first = builds[0][1]
                dev = importlib.import_module(mod.__name__ + '.device')

                def view(a):
                    v = dev.DeviceArray.__new__(dev.DeviceArray)
                    v.ctx, v.shape, v.dtype, v.nbytes = ctx, a.shape, a.dtype, a.nbytes
                    v.ptr = a.ptr
                    v._owner = False
                    v._base = a
                    return v
                mod.src, mod.dst = view(first.src), view(first.dst)
                mod.dmx, mod.dmy = view(first.dmx), view(first.dmy)
                mod.u16 = view(first.u16) if first.u16 is not None else None
                mod.d16 = view(first.d16) if first.u16 is not None else None
        calls = {
            'fused5': lambda o=ops, m=mod: o.remap_conv2d(m.src, m.dmx, m.dmy, k5, out=m.dst),
            'fused7': lambda o=ops, m=mod: o.remap_conv2d(m.src, m.dmx, m.dmy, k7, out=m.dst),
            'c4': lambda o=ops, m=mod: o.remap_conv2d(m.u16, m.dmx, m.dmy, k7r, out=m.dst),
            'conv5': lambda o=ops, m=mod: o.conv2d(m.src, k5, out=m.dst),
            'cubic': lambda o=ops, m=mod: o.warp_perspective(m.src, Hq, (h, w), 'cubic', out=m.dst),
            'c3cubic': lambda o=ops, m=mod: o.warp_perspective_sepconv2d(m.src, Hq, (h, w), g9, g9, 'cubic', out=m.dst),
            'c3lin': lambda o=ops, m=mod: o.warp_perspective_sepconv2d(m.src, Hq, (h, w), g9, g9, 'linear', out=m.dst),
            'sepmap9': lambda o=ops, m=mod: o.remap_sepconv2d(m.src, m.dmx, m.dmy, g9, g9, out=m.dst),
            'sepmap5': lambda o=ops, m=mod: o.remap_sepconv2d(m.src, m.dmx, m.dmy, g9[2:7] / g9[2:7].sum(), g9[2:7] / g9[2:7].sum(), out=m.dst),
            'sep9': lambda o=ops, m=mod: o.sepconv2d(m.src, g9, g9, out=m.dst),
            'fused3': lambda o=ops, m=mod: o.remap_conv2d(m.src, m.dmx, m.dmy, k5[1:4, 1:4] / k5[1:4, 1:4].sum(), out=m.dst),
            'lz4': lambda o=ops, m=mod: o.warp_perspective(m.src, Hq, (h, w), 'lanczos4', out=m.dst),
            'conv9': lambda o=ops, m=mod: o.conv2d(m.src, k9, out=m.dst),
            'conv11': lambda o=ops, m=mod: o.conv2d(m.src, k11, out=m.dst),
            'fused9': lambda o=ops, m=mod: o.remap_conv2d(m.src, m.dmx, m.dmy, k9, out=m.dst),
            'fused11': lambda o=ops, m=mod: o.remap_conv2d(m.src, m.dmx, m.dmy, k11, out=m.dst),
            'remap': lambda o=ops, m=mod: o.remap(m.src, m.dmx, m.dmy, out=m.dst),
            'remapcubic': lambda o=ops, m=mod: o.remap(m.src, m.dmx, m.dmy, 'cubic', out=m.dst),
            'remaplz4': lambda o=ops, m=mod: o.remap(m.src, m.dmx, m.dmy, 'lanczos4', out=m.dst),
            # perspective warps on the tile kernel under a rotation of 15 / 45 degrees
            'lin15': lambda o=ops, m=mod: o.warp_perspective(m.src, rot(15), (h, w), 'linear', out=m.dst),
            'lin45': lambda o=ops, m=mod: o.warp_perspective(m.src, rot(45), (h, w), 'linear', out=m.dst),
            'cub15': lambda o=ops, m=mod: o.warp_perspective(m.src, rot(15), (h, w), 'cubic', out=m.dst),
            'lz15': lambda o=ops, m=mod: o.warp_perspective(m.src, rot(15), (h, w), 'lanczos4', out=m.dst),
            'lin0': lambda o=ops, m=mod: o.warp_perspective(m.src, rot(0), (h, w), 'linear', out=m.dst),
            # uint16 frames, OpenCV's 16U Lanczos4 (PerspectiveCorrection's default on the camera's frames)
            'lz16': lambda o=ops, m=mod: o.warp_perspective(m.u16, Hq, (h, w), 'lanczos4', out=m.d16),
            'lz16r': lambda o=ops, m=mod: o.warp_perspective(m.u16, rot(15), (h, w), 'lanczos4', out=m.d16),
            'copy': lambda m=mod: m.dst.copy_from(m.src),
        }
        builds.append((spec, mod, ctx, knobs, calls))
    n = max(10, 1600 // batch)
    res = {(spec, c): [] for spec, *_ in builds for c in what}
    for r in range(rounds):
        for spec, mod, ctx, knobs, calls in builds:
            old = ctx.set_tuning(**knobs) if knobs else {}
            for c in what:
                res[(spec, c)].append(timeit(ctx, calls[c], n, n // 3))
            if old:
                ctx.set_tuning(**old)
    ref = None
    print('batch %d x 4K float32, ms per launch, %d rounds alternated in one process' % (batch, rounds))
    for c in what:
        for spec, *_ in builds:
            v = res[(spec, c)]
            print('%-8s %-24s %s   min %.4f' % (c, spec, '  '.join('%.4f' % x for x in v), min(v)))
    for spec, mod, ctx, knobs, calls in builds:
        old = ctx.set_tuning(**knobs) if knobs else {}
        calls[what[0]]()
        got = mod.dst.frame(batch - 1).get()
        if old:
            ctx.set_tuning(**old)
        if ref is None:
            ref = got
        same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
        print('%-24s %s result %s' % (spec, what[0], 'identical bits' if same else
                                          'DIFFERS from %s (max |d| %.3g)' % (names[0], np.abs(got - ref).max())))


if __name__ == '__main__':
    main()
