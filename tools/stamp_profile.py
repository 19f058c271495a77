"""Where do the waves of the headline kernel spend their cycles?  A build with -DIPA_DEBUG_STAMPS=1
(make VARIANT=st DEFS=-DIPA_DEBUG_STAMPS=1 ONLY=fused_k5) sums s_memtime differences per phase of
a step over every interior strip and leaves them in the strip's first output row; this script
runs one 64 x 4K launch (after warm-up launches) and adds them up.  Measurement only: the
results of that build are not the filter's.  GPU box only.

    IMGPROC_HIP_LIB=imgprocessor_amd/libimgproc_hip_st.so python tools/stamp_profile.py [knob=v ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def main():
    knobs = {k: int(v) for k, v in (a.split('=') for a in sys.argv[1:])}
    ctx = ia.default_context(0)
    ctx._place_n = 1
    ctx.set_tuning(**knobs)
    B, h, w = 64, 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    one = np.random.default_rng(0).random((16, h, w), dtype=np.float32)
    src = ctx.to_device(np.concatenate([one] * (B // 16)))
    dst = ctx.empty((B, h, w), np.float32)
    for _ in range(60):
        ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(20):
        ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    e1.record()
    ctx.synchronize()
    ms = e0.elapsed_ms(e1) / 20
    halo = ctx.get_tuning('halo_shared')
    step, x0 = (256, 0) if halo else (248, -4)
    sh = 144
    tot = np.zeros(8)
    n = 0
    for f in (0, 17, 40, 63):
        fr = dst.frame(f).get()
        for syi in range(1, h // sh - 1):
            for sxi in range(1, 14):
                xs = sxi * step + x0
                v = fr[syi * sh, xs + 4: xs + 12].astype(np.float64)
                if v[7] > 0:
                    tot += v
                    n += 1
    names = ['barrier', 'wait for the gathers', 'blend + LDS row (+ publish)', 'footprints + gather issue',
             'filter (LDS windows + fma)', 'store issue']
    print('knobs %s: %.4f ms per launch; %d strips read, %.1f steps each' % (knobs, ms, n, tot[7] / n))
    for i, nm in enumerate(names):
        print('  %-30s %8.1f cycles per step  %5.1f %%' % (nm, tot[i] / tot[7], 100 * tot[i] / tot[6]))
    print('  %-30s %8.1f cycles per step' % ('sum', tot[6] / tot[7]))


if __name__ == '__main__':
    main()
