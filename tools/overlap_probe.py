"""Do a texture-addresser-bound remap and an fma-bound 11x11 filter overlap when they run on two
streams at once?  Two contexts (= two streams) of one process, the launches enqueued alternately
by one host thread; wall time of both against each alone.  GPU box only.
    python tools/overlap_probe.py [frames] [h] [w]"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from imgprocessor_amd.device import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
h = int(sys.argv[2]) if len(sys.argv) > 2 else 4320
w = int(sys.argv[3]) if len(sys.argv) > 3 else 7680
a, b = Context(0), Context(0)
rng = np.random.default_rng(0)
img = rng.random((n, h, w), dtype=np.float32)
ang = np.deg2rad(7.0)
M = np.array([[np.cos(ang), -np.sin(ang), 300.0], [np.sin(ang), np.cos(ang), -200.0],
              [2e-6, -1e-6, 1.0]])
k11 = rng.random((11, 11))
k11 /= k11.sum()
sa, da = a.to_device(img), a.empty((n, h, w), np.float32)
sb, db = b.to_device(img), b.empty((n, h, w), np.float32)


def remap():
    ops.warp_perspective(sa, M, (h, w), 'cubic', out=da)


def conv():
    ops.conv2d(sb, k11, out=db)


def wall(fns, reps=10):
    for f in fns:
        f()
    a.synchronize(); b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fns:
            f()
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(2):
    tr, tc, tb = wall([remap]), wall([conv]), wall([remap, conv])
    print('%d x %dx%d: remap %.3f ms, 11x11 %.3f ms, both streams at once %.3f ms (sum %.3f)'
          % (n, h, w, tr, tc, tb, tr + tc))
