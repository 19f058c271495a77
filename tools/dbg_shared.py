"""debug: shared-record loop (batches of 4 frames) against the per-frame path (frames_wg = 0) and
the oracle, for float32 / uint16 frames, K = 5 / 7, maps / homography; prints where bits differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia
from imgprocessor_amd import ops

ctx = ia.default_context(0)
h, w, n = int(os.environ.get('H', 540)), int(os.environ.get('W', 1920)), 4
K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
rng = np.random.default_rng(0)
f32 = rng.random((n, h, w), dtype=np.float32)
u16 = np.round(f32 * 4095).astype(np.uint16)
M = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
for name, src in (('f32', f32), ('u16', u16)):
    d = ctx.to_device(src)
    for ksz in (5, 7):
        k = rng.random((ksz, ksz)); k /= k.sum()
        for what, fn in (('maps', lambda: ops.remap_conv2d(d, dmx, dmy, k)),
                         ('homography', lambda: ops.warp_perspective_conv2d(d, M, (h, w), k))):
            if what == 'homography' and name == 'u16':
                continue
            res = []
            for rep in range(3):
                old = ctx.set_tuning(frames_wg=1)
                a = fn().get()
                ctx.set_tuning(frames_wg=0)
                b = fn().get()
                ctx.set_tuning(**old)
                bad = a.view(np.uint32) != b.view(np.uint32)
                idx = np.argwhere(bad)
                res.append(int(bad.sum()))
                if bad.any() and rep == 0:
                    rows = np.bincount(idx[:, 1], minlength=h)
                    cols = np.bincount(idx[:, 2] // 64, minlength=(w + 63) // 64)
                    print('  %s K=%d %s: %d differ; frames %s; rows %s; 64-px column groups %s; first %s %r vs %r' % (
                        name, ksz, what, bad.sum(), sorted(set(idx[:, 0].tolist())),
                        [(int(r), int(c)) for r, c in enumerate(rows) if c][:12],
                        [(int(r), int(c)) for r, c in enumerate(cols) if c][:12], idx[0].tolist(),
                        a[tuple(idx[0])], b[tuple(idx[0])]))
            print('%s K=%d %-10s mismatching values in 3 runs: %s' % (name, ksz, what, res))
