"""A/B timing of builds of the library (make VARIANT=x DEFS=...): the 4K headline (map-based
undistort + 5x5), the plain 5x5 filter and the standalone remap, each build in its own process,
alternated.  GPU box only.

    python tools/ab_libs.py [--batch 64] [--rounds 3] default nopipe noreuse ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(batch):
    import numpy as np
    sys.path.insert(0, ROOT)
    import imgprocessor_amd as ia
    from imgprocessor_amd import ops

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

    ctx = ia.default_context(0)
    h, w = 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    k3 = np.ones((3, 3)) / 9.0
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    rng = np.random.default_rng(0)
    one = rng.random((16, h, w), dtype=np.float32)
    src = ctx.to_device(np.concatenate([one] * (batch // 16)) if batch >= 16 else one[:batch])
    dst = ctx.empty((batch, h, w), np.float32)
    n = max(10, 3200 // batch)
    out = {}
    out['fused5'] = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k5, out=dst), n, n // 2)
    out['fused3'] = timeit(ctx, lambda: ops.remap_conv2d(src, dmx, dmy, k3, out=dst), n // 2, 5)
    out['conv5'] = timeit(ctx, lambda: ops.conv2d(src, k5, out=dst), n // 2, 5)
    out['conv3'] = timeit(ctx, lambda: ops.conv2d(src, k3, out=dst), n // 2, 5)
    out['copy'] = timeit(ctx, lambda: dst.copy_from(src), n // 2, 5)
    # a checksum of the fused result: builds must agree bit for bit
    ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    got = dst.frame(batch - 1).get()
    out['sum'] = float(np.float64(got.astype(np.float64).sum()))
    out['crc'] = int(np.bitwise_xor.reduce(got.view(np.uint32).ravel()))
    print('AB ' + json.dumps(out))


def main():
    args = sys.argv[1:]
    if args and args[0] == '--child':
        return child(int(args[1]))
    batch, rounds, names = 64, 3, []
    while args:
        a = args.pop(0)
        if a == '--batch':
            batch = int(args.pop(0))
        elif a == '--rounds':
            rounds = int(args.pop(0))
        else:
            names.append(a)
    names = names or ['default']
    res = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            env = dict(os.environ)
            if n != 'default':
                env['IMGPROC_HIP_LIB'] = os.path.join(ROOT, 'imgprocessor_amd',
                                                      'libimgproc_hip_%s.so' % n)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(batch)],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith('AB ')]
            if not line:
                print(n, 'FAILED', p.stdout[-2000:], p.stderr[-2000:])
                continue
            res[n].append(json.loads(line[0][3:]))
    print('batch %d x 4K float32, ms per launch (rounds alternated)' % batch)
    for n in names:
        for key in ('fused5', 'fused3', 'conv5', 'conv3', 'copy'):
            print('%-10s %-7s %s' % (n, key, '  '.join('%.4f' % r[key] for r in res[n])))
        print('%-10s checksum %s' % (n, sorted({(r['sum'], r['crc']) for r in res[n]})))


if __name__ == '__main__':
    main()
