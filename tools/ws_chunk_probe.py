"""Probe: the two-launch warp -> filter chains (C3 bicubic, rotated C3, C5) with the batch cut
into chunks of frames, so that the chunk of the workspace between the launches (33 MB per 4K frame)
stays within the 256 MB memory-side cache.  Host-side chunking through the public API."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import imgprocessor_amd as ia
from imgprocessor_amd import ops
from imgprocessor_amd.utils import getPerspectiveTransform

ctx = ia.default_context(0)
h, w, B = 2160, 3840, 16
quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
Hm = np.linalg.inv(getPerspectiveTransform(quad, rect))
g9 = ops.gaussian_kernel1d(1.0)
src = ctx.to_device(np.random.default_rng(0).random((B, h, w), dtype=np.float32))
dst = ctx.empty((B, h, w), np.float32)
ref = ctx.empty((B, h, w), np.float32)


def frames(arr, f, n):
    """view of frames f .. f + n - 1 (no copy, not owning)"""
    import ctypes as C
    from imgprocessor_amd.device import DeviceArray
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype = arr.ctx, (n,) + arr.shape[1:], arr.dtype
    v.nbytes = arr.nbytes // arr.shape[0] * n
    v.ptr = C.c_void_p(arr.ptr.value + f * (arr.nbytes // arr.shape[0]))
    v._owner = False
    v._base = arr
    return v


views = {c: [(frames(src, f, c), frames(dst, f, c)) for f in range(0, B, c)] for c in (16, 8, 4, 2, 1)}


def run(chunk, interp, M):
    for s, d in views[chunk]:
        ops.warp_perspective_sepconv2d(s, M, (h, w), g9, g9, interp, out=d)


def t(fn, n=20):
    for _ in range(5): fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event(); e0.record()
    for _ in range(n): fn()
    e1.record(); ctx.synchronize()
    return e0.elapsed_ms(e1) / n


a = np.deg2rad(15.0)
cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
R = np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
              [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy], [0, 0, 1.0]])
Hr = np.array([[1, 0, 0], [0, 1, 0], [2e-6, 1e-6, 1.0]]) @ R
for _ in range(50): run(16, 'cubic', Hm)
for name, interp, M in (('C3 bicubic', 'cubic', Hm), ('C3 rotated 15 linear', 'linear', Hr), ('C3 rotated 15 cubic', 'cubic', Hr)):
    res = {}
    for rnd in range(2):
        for chunk in (16, 8, 4, 2, 1):
            res.setdefault(chunk, []).append(t(lambda: run(chunk, interp, M)))
    print(name + ': ' + '  '.join('chunk %d: %.4f' % (c, min(v)) for c, v in res.items()), flush=True)
