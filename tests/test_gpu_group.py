"""GPU: the ring kernel (csrc/ring_stencil.hpp: planned clean strips, bilinear taps from an LDS
ring of source rows, the other strips left to the per-frame kernel through a skip mask) and the
experimental frame-group kernel (csrc/group_stencil.hpp: one workgroup per strip of 4 frames,
shared footprint records) against the per-frame marching kernel - the same arithmetic, so the results must agree BIT FOR BIT - and against the
oracle.  Geometries are chosen to exercise every branch: ring rows, rows that do not fit the
ring (rotation), footprints on the source border, constant / reflect filter borders, rim
strips, ragged sizes, batches that do not fill the last group.
"""
import numpy as np
import pytest

from .conftest import assert_close, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


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


def run_three(ia, src, mx, my, k, **kw):
    """per-frame kernel, group kernel with gathers only, group kernel with the ring"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    if not ctx.get_tuning('experimental'):
        # default build (no `make EXPERIMENTAL=1`): the shelved batch kernels are not in the
        # library; the callers' oracle comparisons still run on the per-frame kernel
        ref = ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()
        return ref, ref, ref
    old = ctx.set_tuning(group=0, ring=0, group_min=1, group_ring=0, ring_min=1)
    try:
        ref = ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()
        # clean strips on the ring kernel, the rest on the per-frame kernel (skip mask)
        ctx.set_tuning(ring=1)
        split = ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()
        same_bits(split, ref, 'ring kernel + per-frame kernel')
        ctx.set_tuning(ring=0, group=1)
        gat = ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()
        ctx.set_tuning(group_ring=1)
        ring = ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()
    finally:
        ctx.set_tuning(**old)
    return ref, gat, ring


def same_bits(a, b, what):
    assert a.shape == b.shape
    bad = a.view(np.uint32) != b.view(np.uint32)
    bad &= ~(np.isnan(a) & np.isnan(b))
    assert not bad.any(), '%s: %d of %d values differ, first at %s (%r vs %r)' % (
        what, bad.sum(), a.size, np.argwhere(bad)[0], a[bad][0], b[bad][0])


@pytest.mark.parametrize('n', [1, 3, 4, 6])
@pytest.mark.parametrize('shape', [(96, 300), (131, 517), (200, 1030)])
def test_group_matches_per_frame_radial(ia, oracle, n, shape):
    h, w = shape
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5)
    ref, gat, ring = run_three(ia, src, mx, my, k)
    same_bits(gat, ref, 'gather mode')
    same_bits(ring, ref, 'ring mode')
    want = oracle.conv2d(oracle.remap(src[n - 1], mx, my), k)
    assert_close(ring[n - 1], want, 1e-5, 1e-5 * np.abs(want).max(), 'vs oracle')


@pytest.mark.parametrize('case', ['rot3', 'rot20', 'rot90', 'rot180', 'shift_out', 'pincushion',
                                  'flipx', 'zoom_out', 'zoom_in'])
def test_group_geometries(ia, oracle, case):
    h, w, n = 150, 700, 5
    src = frames(n, h, w)
    if case.startswith('rot'):
        mx, my = rot_maps(h, w, float(case[3:]))
    elif case == 'shift_out':
        mx, my, _, _ = radial_maps(h, w, shift=-40.5)
    elif case == 'pincushion':
        mx, my, _, _ = radial_maps(h, w, k1=0.25)
    elif case == 'flipx':
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        mx, my = (w - 1 - x + 0.25).astype(np.float32), (y + 0.5).astype(np.float32)
    elif case == 'zoom_out':
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        mx, my = (x * 1.7 - 100).astype(np.float32), (y * 1.7 - 30).astype(np.float32)
    else:
        y, x = np.mgrid[0:h, 0:w].astype(np.float32)
        mx, my = (x * 0.31 + 7.3).astype(np.float32), (y * 0.31 + 3.1).astype(np.float32)
    k = kern(5, 9)
    for kw in ({}, {'conv_mode': 'constant'}, {'border_mode': 'reflect', 'conv_mode': 'wrap'}):
        ref, gat, ring = run_three(ia, src, mx, my, k, **kw)
        same_bits(gat, ref, '%s gather %r' % (case, kw))
        same_bits(ring, ref, '%s ring %r' % (case, kw))
        # every border combination of the per-frame kernel against the oracle, two frames
        for f in (0, n - 1):
            want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR,
                                              oracle._mode(kw.get('border_mode', 'constant')), 0.0),
                                 k, kw.get('conv_mode', 'reflect'))
            assert_close(ref[f], want, 1e-5, 1e-5 * np.abs(want).max(),
                         '%s %r frame %d vs oracle' % (case, kw, f))


def test_group_nan_and_far_coordinates(ia):
    h, w, n = 80, 600, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    mx = mx.copy(); my = my.copy()
    mx[10, 50:60] = np.nan
    my[20, 300:310] = np.inf
    mx[30, 400:420] = 3e7
    mx[40:44, :] = -5.0
    ref, gat, ring = run_three(ia, src, mx, my, kern(5), border_value=0.25)
    same_bits(gat, ref, 'gather')
    same_bits(ring, ref, 'ring')
    from oracle import oracle as orc
    orc.build()
    for f in range(n):
        want = orc.conv2d(orc.remap(src[f], mx, my, orc.LINEAR, orc.CONSTANT, 0.25), kern(5))
        assert_close(ref[f], want, 1e-5, 1e-5 * np.nanmax(np.abs(want)), 'nan / far coordinates vs oracle')


def test_group_q5(ia):
    from imgprocessor_amd import ops
    h, w, n = 120, 520, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    ref, gat, ring = run_three(ia, src, mx, my, kern(5), interpolation='linear_cv_q5')
    same_bits(gat, ref, 'gather q5')
    same_bits(ring, ref, 'ring q5')


def test_group_analytic_sources(ia, oracle):
    """undistort (lens model in the kernel) and homography coordinate sources"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 140, 900, 6
    src = frames(n, h, w)
    _, _, K, dist = radial_maps(h, w)
    k = kern(5, 3)
    d_src = ctx.to_device(src)
    M = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
    if not ctx.get_tuning('experimental'):
        pytest.skip('shelved round-2 kernels: build with make EXPERIMENTAL=1')
    old = ctx.set_tuning(group=0, ring=0, ring_min=1)
    try:
        ref_u = ops.undistort_conv2d(d_src, K, dist, K, k).get()
        ref_h = ops.warp_perspective_conv2d(d_src, M, (h, w), k).get()
        ctx.set_tuning(ring=1)
        same_bits(ops.undistort_conv2d(d_src, K, dist, K, k).get(), ref_u, 'undistort, ring kernel')
        same_bits(ops.warp_perspective_conv2d(d_src, M, (h, w), k).get(), ref_h,
                  'homography, ring kernel')
        ctx.set_tuning(ring=0)
        for ring in (0, 1):
            ctx.set_tuning(group=1, group_min=1, group_ring=ring)
            same_bits(ops.undistort_conv2d(d_src, K, dist, K, k).get(), ref_u,
                      'undistort ring=%d' % ring)
            same_bits(ops.warp_perspective_conv2d(d_src, M, (h, w), k).get(), ref_h,
                      'homography ring=%d' % ring)
    finally:
        ctx.set_tuning(**old)


def test_group_4k_strip_geometry(ia, oracle):
    """4K frames: tall strips, many chunks per strip, all interior strips on the fast path"""
    h, w, n = 2160, 3840, 5
    base = synth((h, w), 7)
    src = np.stack([np.roll(base, 11 * i, axis=0) for i in range(n)])
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5, 1)
    ref, gat, ring = run_three(ia, src, mx, my, k)
    same_bits(gat, ref, '4K gather')
    same_bits(ring, ref, '4K ring')
    sub = slice(1000, 1200)
    want = oracle.conv2d(oracle.remap(src[4], mx, my), k)
    assert_close(ring[4][sub], want[sub], 1e-5, 1e-5 * np.abs(want).max(), '4K vs oracle')


@pytest.mark.parametrize('n', [4, 8, 12])
@pytest.mark.parametrize('K', [3, 5, 7, 9, 11])
def test_frames_of_a_strip_in_one_workgroup(ia, K, n):
    """WaveParams::frames_wg (the waves of a workgroup = consecutive frames of one strip, map-based
    fused kernels): the bits of the strip-per-wave order, dense and separable chains"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    for (h, w) in ((200, 1030), (131, 517)):
        src = frames(n, h, w)
        mx, my, _, _ = radial_maps(h, w)
        d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
        g = kern(K)[K // 2].copy()
        g /= g.sum()
        res = []
        for knob in (0, 1):
            old = ctx.set_tuning(frames_wg=knob)
            try:
                res.append((ops.remap_conv2d(d_src, dmx, dmy, kern(K)).get(),
                            ops.remap_sepconv2d(d_src, dmx, dmy, g, g).get() if K <= 9 else None))
            finally:
                ctx.set_tuning(**old)
        same_bits(res[1][0], res[0][0], 'frames_wg dense K=%d n=%d %r' % (K, n, (h, w)))
        if K <= 9:
            same_bits(res[1][1], res[0][1], 'frames_wg separable K=%d n=%d %r' % (K, n, (h, w)))


@pytest.mark.parametrize('n', [2, 3, 5, 8])
@pytest.mark.parametrize('K', [3, 5])
@pytest.mark.parametrize('knob', [1, 2])
def test_frame_pair_kernel_matches_per_frame(ia, K, n, knob):
    """one wave per strip of a frame PAIR (csrc/wave_pair.hpp, pair=1) and a sampler wave + a filter
    wave per strip (csrc/wave_split.hpp, pair=2): the bits of the per-frame kernel"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    if not ctx.get_tuning('experimental'):
        pytest.skip('shelved round-2 kernels: build with make EXPERIMENTAL=1')
    for (h, w), q5 in (((200, 1030), False), ((131, 517), True), ((330, 780), False)):
        src = frames(n, h, w)
        mx, my, _, _ = radial_maps(h, w)
        mx = mx.copy()
        mx[h // 2, 40:60] = np.nan
        d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
        interp = 'linear_cv_q5' if q5 else 'linear'
        for kw in ({}, {'conv_mode': 'constant'}, {'border_mode': 'reflect', 'conv_mode': 'wrap'}):
            old = ctx.set_tuning(pair=0, ring=0, group=0)
            try:
                ref = ops.remap_conv2d(d_src, dmx, dmy, kern(K), interp, **kw).get()
                ctx.set_tuning(pair=knob)
                got = ops.remap_conv2d(d_src, dmx, dmy, kern(K), interp, **kw).get()
            finally:
                ctx.set_tuning(**old)
            same_bits(got, ref, 'pair=%d kernel K=%d n=%d %r %r' % (knob, K, n, (h, w), kw))


def test_batches_that_are_no_multiple_of_the_workgroup_frames(ia):
    """n frames with n % 4 != 0: the shared-record loop runs the first n - n % 4 frames and then
    the last 4 (up to three of them a second time) - every frame must have the bits of the
    single-frame call, for the dense and the separable chain, float32 and uint16 frames"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    rng = np.random.default_rng(17)
    h, w = 150, 700
    mx, my = radial_maps(h, w)[:2]
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    k5 = rng.random((5, 5))
    k5 /= k5.sum()
    g = np.exp(-0.5 * np.arange(-4, 5) ** 2)
    g /= g.sum()
    for n in (5, 7, 9, 10, 15):
        f32 = rng.random((n, h, w), dtype=np.float32)
        u16 = (f32 * 4095).astype(np.uint16)
        for src in (f32, u16):
            d = ctx.to_device(src)
            got = ops.remap_conv2d(d, dmx, dmy, k5).get()
            gots = ops.remap_sepconv2d(d, dmx, dmy, g, g).get() if src.dtype == np.float32 else None
            for f in range(n):
                one = ctx.to_device(src[f])
                ref = ops.remap_conv2d(one, dmx, dmy, k5).get()
                assert np.array_equal(got[f].view(np.uint32), ref.view(np.uint32)), (n, f, src.dtype)
                if gots is not None:
                    refs = ops.remap_sepconv2d(one, dmx, dmy, g, g).get()
                    assert np.array_equal(gots[f].view(np.uint32), refs.view(np.uint32)), (n, f, 'sep')
