"""GPU: dense K x K kernels that are an outer product ky (x) kx take the separable K + K loops
(knob rank1_sep; the reference obtains its Gaussians separably - scipy.ndimage.gaussian_filter,
filters/standardDeviation.py:23, filters/fastFilter.py:42 - and the bench's 5x5 is outer(g, g)).
What is checked: the route is taken where it is claimed (read-only counter "rank1_routed"), the
routed result is the separable entry point's bit for bit and the oracle's double-precision dense
sum within 1e-5, kernels that are NOT an outer product stay on the dense loop, and the knob turns
it off.  Also here: the read-back of the frame-group chunk a launch chose ("group_chunk_used").
"""
import numpy as np
import pytest

from .conftest import assert_close
from .gpu_helpers import frames, radial_maps, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def gauss(k, sigma=1.0):
    g = np.exp(-0.5 * ((np.arange(k) - k // 2) / sigma) ** 2)
    return g / g.sum()


def routed(ctx):
    return ctx.get_tuning('rank1_routed')


@pytest.mark.parametrize('K', [3, 5, 7, 9])
def test_rank1_chain_is_the_separable_chain(ia, oracle, K):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 140, 610, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    ky, kx = gauss(K), gauss(K, 0.8)            # an asymmetric outer product
    k = np.outer(ky, kx)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    before = routed(ctx)
    got = ops.remap_conv2d(d_src, dmx, dmy, k).get()
    assert routed(ctx) == before + 1, 'an exact outer product must take the separable chain'
    sep = ops.remap_sepconv2d(d_src, dmx, dmy, ky, kx).get()
    # the factors the library recovers are ky, kx up to a common scale: float32(taps) may differ in
    # the last bit from the caller's own, so the comparison with the separable entry point is 2 ulp
    assert_close(got, sep, 3e-7, 3e-7 * np.abs(sep).max(), 'routed vs remap_sepconv2d')
    for f in range(n):
        want = oracle.conv2d(oracle.remap(src[f], mx, my), k)
        assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(), 'K=%d frame %d vs oracle' % (K, f))
    # knob off: the dense loop, same answer within float32 summation order
    old = ctx.set_tuning(rank1_sep=0)
    try:
        before = routed(ctx)
        dense = ops.remap_conv2d(d_src, dmx, dmy, k).get()
        assert routed(ctx) == before
    finally:
        ctx.set_tuning(**old)
    assert_close(got, dense, 2e-6, 2e-6 * np.abs(dense).max(), 'separable vs dense loop')


def test_not_rank1_stays_dense(ia, oracle):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 120, 530
    src = frames(4, h, w)
    mx, my, _, _ = radial_maps(h, w)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    g = gauss(5)
    almost = np.outer(g, g)
    almost[1, 3] *= 1.0 + 1e-9                   # nine orders above the double rounding of outer()
    rnd = np.random.default_rng(5).random((5, 5))
    for name, k in (('perturbed outer product', almost), ('random', rnd), ('zeros', np.zeros((5, 5)))):
        before = routed(ctx)
        got = ops.remap_conv2d(d_src, dmx, dmy, k).get()
        assert routed(ctx) == before, name
        want = oracle.conv2d(oracle.remap(src[0], mx, my), k)
        assert_close(got[0], want, 1e-5, 1e-5 * max(np.abs(want).max(), 1e-30), name)
    # bicubic taps: the separable chain would take two launches - dense
    before = routed(ctx)
    ops.remap_conv2d(d_src, dmx, dmy, np.outer(g, g), 'cubic').get()
    assert routed(ctx) == before


@pytest.mark.parametrize('n', [1, 3, 4, 8])
@pytest.mark.parametrize('K', [3, 5, 7, 9])
def test_uint16_frames_take_the_one_kernel_separable_chain(ia, oracle, n, K):
    """round 6: map-based bilinear remap -> separable K + K filter on uint16 frames (camera frames as
    transformations.toFloatArray ingests them) runs in ONE kernel as float32 frames always did (knob sep_u16), and an
    outer-product K x K kernel then takes that route too; both against the oracle, the two-launch form beside it"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 150, 610
    src = frames(n, h, w, np.uint16)
    mx, my, Kc, dist = radial_maps(h, w, shift=3.3)
    mx = mx - np.float32(20.0)              # a rim of border pixels on the left: the filter sees the constant border
    ky, kx = gauss(K), gauss(K, 0.8)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    got = ops.remap_sepconv2d(d, dmx, dmy, ky, kx, 'linear', 'constant', 100.0).get()
    before = routed(ctx)
    dense_call = ops.remap_conv2d(d, dmx, dmy, np.outer(ky, kx), 'linear', 'constant', 100.0).get()
    assert routed(ctx) == before + 1, 'uint16 frames + maps + outer product: the separable chain'
    old = ctx.set_tuning(sep_u16=0)
    try:
        two = ops.remap_sepconv2d(d, dmx, dmy, ky, kx, 'linear', 'constant', 100.0).get()
        before = routed(ctx)
        dense = ops.remap_conv2d(d, dmx, dmy, np.outer(ky, kx), 'linear', 'constant', 100.0).get()
        # (9 x 9 on uint16 frames is remap -> workspace -> plain filter, and the plain 9 x 9 filter has its own route)
        assert routed(ctx) == before + (1 if K == 9 else 0), 'knob off: the dense loop'
    finally:
        ctx.set_tuning(**old)
    for f in range(n):
        mid = oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 100.0, out_dtype=np.float32)
        want = oracle.sepconv2d(mid, ky, kx)
        for name, a in (('one kernel', got), ('routed dense call', dense_call), ('two launches', two), ('dense loop', dense)):
            assert_close(a[f], want, 1e-5, 1e-5 * 4095, '%s, %d+%d taps, frame %d of %d' % (name, K, K, f, n))
    # a homography as the coordinate source: the same kernel family (C3's chain on camera frames)
    Mh = np.array([[1.01, 0.004, -3.0], [-0.003, 0.99, 2.0], [1e-5, -2e-5, 1.0]])
    got_h = ops.warp_perspective_sepconv2d(d, Mh, (h, w), ky, kx, 'linear', 'constant', 100.0).get()
    before = routed(ctx)
    got_hd = ops.warp_perspective_conv2d(d, Mh, (h, w), np.outer(ky, kx), 'linear', 'constant', 100.0).get()
    assert routed(ctx) == before + 1
    old = ctx.set_tuning(sep_u16=0)
    try:
        two_h = ops.warp_perspective_sepconv2d(d, Mh, (h, w), ky, kx, 'linear', 'constant', 100.0).get()
    finally:
        ctx.set_tuning(**old)
    for f in (0, n - 1):
        mid = oracle.warp_perspective(src[f], Mh, (h, w), oracle.LINEAR, oracle.CONSTANT, 100.0, out_dtype=np.float32)
        want = oracle.sepconv2d(mid, ky, kx)
        for name, a in (('one kernel', got_h), ('routed dense call', got_hd), ('two launches', two_h)):
            assert_close(a[f], want, 1e-5, 1e-5 * 4095, 'homography, %s, %d+%d taps, frame %d of %d' % (name, K, K, f, n))
    # the lens model by value: its cached map takes the same route
    before = routed(ctx)
    got_u = ops.undistort_conv2d(d, Kc, dist, Kc, np.outer(ky, kx)).get()
    assert routed(ctx) >= before + 1
    umx, umy = oracle.build_undistort_map(Kc, dist, Kc, h, w)
    want = oracle.conv2d(oracle.remap(src[n - 1], umx, umy, out_dtype=np.float32), np.outer(ky, kx))
    assert_close(got_u[n - 1], want, 1e-5, 1e-5 * 4095, 'undistort + outer product, uint16 frames')


def test_rank1_other_coordinate_sources(ia, oracle):
    """the lens model and the homography as coordinate sources take the same route"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 130, 520
    src = frames(4, h, w)
    K = np.array([[w * 1.0, 0, (w - 1) / 2.0], [0, w * 1.0, (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    M = np.array([[1.01, 0.02, -3.0], [-0.015, 0.99, 2.0], [1e-5, -2e-5, 1.0]])
    g = gauss(5)
    k = np.outer(g, g)
    d_src = ctx.to_device(src)
    before = routed(ctx)
    got_u = ops.undistort_conv2d(d_src, K, dist, K, k).get()
    got_h = ops.warp_perspective_conv2d(d_src, M, (h, w), k).get()
    assert routed(ctx) == before + 2
    mx, my = oracle.build_undistort_map(K, dist, K, h, w)
    want = oracle.conv2d(oracle.remap(src[1], mx, my), k)
    assert_close(got_u[1], want, 1e-5, 1e-5 * np.abs(want).max(), 'undistort + outer product')
    want = oracle.conv2d(oracle.warp_perspective(src[2], M, (h, w)), k)
    assert_close(got_h[2], want, 1e-5, 1e-5 * np.abs(want).max(), 'warp + outer product')


def test_plain_filter_rank1(ia, oracle):
    """the plain filter: 9 x 9 outer products go to the separable loop (the dense loops hold their rows
    in registers at stream rate up to 7 x 7 and stay), masks and constant borders with a value stay dense"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    img = frames(3, 150, 600)
    d = ctx.to_device(img)
    for K, expect in ((3, 0), (5, 0), (7, 0), (9, 1)):
        k = np.outer(gauss(K), gauss(K, 1.3))
        before = routed(ctx)
        got = ops.conv2d(d, k).get()
        assert routed(ctx) == before + expect, K
        want = oracle.conv2d(img[1], k)
        assert_close(got[1], want, 1e-5, 1e-5 * np.abs(want).max(), 'plain K=%d' % K)
    k9 = np.outer(gauss(9), gauss(9))
    before = routed(ctx)
    got = ops.conv2d(d, k9, mode='constant', cval=0.3).get()       # does not factor: dense
    assert routed(ctx) == before
    want = oracle.conv2d(img[0], k9, 'constant', 0.3)
    assert_close(got[0], want, 1e-5, 1e-5 * np.abs(want).max(), 'constant border with a value')
    got = ops.conv2d(d, k9, mode='constant', cval=0.0).get()       # factors
    assert routed(ctx) == before + 1
    want = oracle.conv2d(img[0], k9, 'constant', 0.0)
    assert_close(got[0], want, 1e-5, 1e-5 * np.abs(want).max(), 'constant border 0')
    for mode in ('reflect', 'wrap', 'nearest', 'mirror'):
        got = ops.conv2d(d, k9, mode=mode).get()
        want = oracle.conv2d(img[2], k9, mode)
        assert_close(got[2], want, 1e-5, 1e-5 * np.abs(want).max(), 'mode %s' % mode)


def test_group_chunk_read_back(ia):
    """frame groups are walked a chunk at a time; a chunk that does not divide the group count goes to the
    nearest divisor (it fell back to 0 silently until round 5) and the choice can be read back"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 96, 520
    mx, my, _, _ = radial_maps(h, w)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    k = np.random.default_rng(1).random((5, 5))
    # (36 frames = 9 groups, a quarter = 2: divisors 1 and 3 are equally near - the smaller)
    for n, want in ((8, 0), (16, 1), (20, 1), (36, 1), (40, 2), (64, 4)):
        src = frames(n, h, w)
        got = ops.remap_conv2d(ctx.to_device(src), dmx, dmy, k).get()
        assert ctx.get_tuning('group_chunk_used') == want, (n, ctx.get_tuning('group_chunk_used'))
        old = ctx.set_tuning(group_chunk=0)
        try:
            flat = ops.remap_conv2d(ctx.to_device(src), dmx, dmy, k).get()
            assert ctx.get_tuning('group_chunk_used') == 0
        finally:
            ctx.set_tuning(**old)
        same_bits(got, flat, '%d frames: chunked order vs all groups together' % n)


@pytest.mark.parametrize('n', [8, 16, 20, 32, 64])
@pytest.mark.parametrize('shape', [(301, 517), (700, 530), (1080, 300)])
def test_short_strips_at_the_end_of_every_xcd_share(ia, n, shape):
    """knob tail_rows (round 6): chunked batches end every XCD's share of the launch on short strips - a
    non-uniform strip geometry (segments of tall strips followed by short ones, WaveParams::seg_count) that
    must tile the frame exactly as the uniform one does: identical bits for every short height, frame count
    (chunks of 1 .. 4 groups, group counts that do not divide) and ragged size, dense and separable loops"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    dense = np.random.default_rng(2).random((5, 5))
    g = gauss(5)
    used = set()
    for sh in (0, 40, 96):
        for kern, what in ((dense, 'dense 5x5'), (np.outer(g, g), 'separable 5 + 5')):
            ref = None
            for tail in (0, -1, 8, 24, 50, 1000):
                old = ctx.set_tuning(strip_h=sh, tail_rows=tail)
                try:
                    got = ops.remap_conv2d(d_src, dmx, dmy, kern).get()
                    used.add(ctx.get_tuning('tail_rows_used'))
                finally:
                    ctx.set_tuning(**old)
                if ref is None:
                    ref = got
                else:
                    same_bits(got, ref, '%s, %d frames, strip_h %d, tail_rows %d' % (what, n, sh, tail))
    # (8 frames = 2 groups: no chunks; 20 frames = 5 chunks of one group, which do not map onto the 8 XCDs' block
    # ranges: uniform strips; the others must have taken the geometry at least once)
    assert (used == {0}) == (n in (8, 20)), used


@pytest.mark.parametrize('n', [1, 3, 4, 8])
@pytest.mark.parametrize('K', [3, 5, 7, 9])
def test_uint8_frames_take_the_one_kernel_separable_chain(ia, oracle, n, K):
    """... and 8-bit camera frames (round 6): map-based bilinear remap -> separable K + K filter in one kernel, the
    outer-product K x K route with it; against the oracle, with the two-launch form (knob sep_u16 = 0) beside it"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 150, 610
    src = frames(n, h, w, np.uint8)
    mx, my, Kc, dist = radial_maps(h, w, shift=3.3)
    mx = mx - np.float32(20.0)
    ky, kx = gauss(K), gauss(K, 0.8)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    got = ops.remap_sepconv2d(d, dmx, dmy, ky, kx, 'linear', 'constant', 100.0).get()
    before = routed(ctx)
    dense_call = ops.remap_conv2d(d, dmx, dmy, np.outer(ky, kx), 'linear', 'constant', 100.0).get()
    assert routed(ctx) == before + 1, 'uint8 frames + maps + outer product: the separable chain'
    old = ctx.set_tuning(sep_u16=0)
    try:
        two = ops.remap_sepconv2d(d, dmx, dmy, ky, kx, 'linear', 'constant', 100.0).get()
    finally:
        ctx.set_tuning(**old)
    same_bits(got, two, 'one kernel against the two launches')
    for f in range(n):
        mid = oracle.remap(src[f], mx, my, oracle.LINEAR, oracle.CONSTANT, 100.0, out_dtype=np.float32)
        want = oracle.sepconv2d(mid, ky, kx)
        for name, a in (('one kernel', got), ('routed dense call', dense_call)):
            assert_close(a[f], want, 1e-5, 1e-5 * 255, '%s, %d+%d taps, frame %d of %d' % (name, K, K, f, n))
