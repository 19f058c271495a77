"""helpers of the GPU tests: synthetic frame batches, map pairs, kernels, bit comparison"""
import numpy as np

from .conftest import synth


def frames(n, h, w, dtype=np.float32):
    out = np.stack([synth((h, w), 100 + i) for i in range(n)])
    if dtype == np.uint16:
        return np.round(out * 4095).astype(np.uint16)
    return out.astype(dtype)


def radial_maps(h, w, k1=-0.12, shift=0.0):
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([k1, 0.03, 1e-3, -5e-4, 0.0])
    from imgprocessor_amd import ops
    mx, my = ops.build_undistort_map(K, dist, K, h, w)
    return (mx + np.float32(shift)).astype(np.float32), my.astype(np.float32), K, dist


def rot_maps(h, w, deg):
    a = np.deg2rad(deg)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    mx = np.cos(a) * (x - cx) - np.sin(a) * (y - cy) + cx
    my = np.sin(a) * (x - cx) + np.cos(a) * (y - cy) + cy
    return mx.astype(np.float32), my.astype(np.float32)


def kern(K, seed=5):
    k = np.random.default_rng(seed).random((K, K))
    return k / k.sum()


def same_bits(a, b, what):
    assert a.shape == b.shape
    bad = a.view(np.uint32) != b.view(np.uint32)
    bad &= ~(np.isnan(a) & np.isnan(b))
    assert not bad.any(), '%s: %d of %d values differ, first at %s (%r vs %r)' % (
        what, bad.sum(), a.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])
