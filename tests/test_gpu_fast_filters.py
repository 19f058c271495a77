"""GPU: filters/fastFilter.py and filters/fastMean.py - the strided window statistics against the
reference's own output (tests/golden/fast_filter.npz), cv2.resize (restated from OpenCV's
published algorithm, cv2-unpinned) against the oracle's restatement bit for bit and against the
identities the algorithm must satisfy."""
import numpy as np
import pytest

from .conftest import load_golden, assert_close, load_cv_golden
from .test_oracle_golden import fast_filter_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)  # raises without a gfx950 device: no fallback
    return imgprocessor_amd


def test_fast_filter_statistics_golden(ia):
    from imgprocessor_amd.filters import fastFilter
    g = load_golden('fast_filter.npz')
    cases = fast_filter_cases(g)
    assert len(cases) == 21
    for key, img, ksize, every, fn, smooth in cases:
        got = fastFilter(img, ksize, every, resize=False, fn=fn, smoothksize=smooth)
        assert got.dtype == np.float64
        assert_close(got, g[key], 1e-12, 0, key)


def test_fast_filter_statistics_vs_oracle(ia, oracle):
    rng = np.random.default_rng(2)
    for (h, w, k, every) in ((200, 310, 40, 2), (97, 64, 7, 1), (64, 300, 33, 11), (33, 35, 3, 1)):
        for dt in (np.float64, np.float32):
            a = (rng.random((h, w)) * 50).astype(dt)
            a[rng.random((h, w)) < 0.1] = np.nan
            a[:h // 3, :w // 4] = np.nan
            a[h // 2] = a[h // 2, 0]          # ties: many equal values in a window
            for fn in ('median', 'nanmedian', 'mean', 'nanmean'):
                src = a if fn.startswith('nan') else np.nan_to_num(a, nan=1.0)
                got = ia.ops.fast_filter_stat(src, k, every, fn)
                n0, n1 = -(-h // every), -(-w // every)
                want = np.empty((n0, n1))
                oracle._chk(oracle.lib().orc_fast_filter_stat(
                    oracle._p(np.ascontiguousarray(src)), oracle._dt(src), oracle.C.c_long(h),
                    oracle.C.c_long(w), oracle.C.c_long(k), oracle.C.c_long(every),
                    oracle.C.c_int(oracle._FF_FN[fn]), oracle._p(want)), 'stat')
                # medians are order statistics: exact; means differ in summation order
                tol = 0 if 'median' in fn else 1e-12
                assert_close(got, want, tol, 0, '%s %s' % ((h, w, k, every), fn))
    # a plain median window that holds a NaN is NaN (np.median)
    b = np.ones((20, 20))
    b[10, 10] = np.nan
    out = ia.ops.fast_filter_stat(b, 3, 1, 'median')
    assert np.isnan(out[10, 10]) and np.isnan(out[8, 12]) and out[0, 0] == 1.0
    with pytest.raises(NotImplementedError):     # 200 x 200 window samples do not fit a wave's LDS
        ia.ops.fast_filter_stat(np.zeros((300, 300)), 100, 1, 'median')


def test_resize_vs_oracle_bit_for_bit(ia, oracle):
    rng = np.random.default_rng(4)
    interps = (('linear', oracle.RESIZE_LINEAR), ('cubic', oracle.RESIZE_CUBIC),
               ('lanczos4', oracle.RESIZE_LANCZOS4))
    shapes = (((37, 53), (120, 171)), ((11, 17), (120, 171)), ((60, 80), (60, 80)),
              ((50, 70), (31, 44)), ((9, 300), (40, 77)), ((3, 2), (17, 9)), ((1, 1), (5, 6)))
    for dt in (np.float32, np.float64):
        for (s, d) in shapes:
            a = rng.standard_normal(s).astype(dt)
            for name, oid in interps:
                got = ia.ops.resize(a, d, name)
                want = oracle.resize(a, d, oid)
                assert got.dtype == dt and got.shape == d
                assert np.array_equal(got, want), (dt.__name__, s, d, name)
        for (s, d) in (((36, 52), (18, 26)), ((36, 52), (12, 13)), ((37, 53), (10, 17)),
                       ((120, 171), (12, 17)), ((40, 40), (40, 40)), ((35, 50), (17, 25)),
                       ((2160, 384), (216, 38))):
            a = rng.random(s).astype(dt)
            got = ia.ops.resize(a, d, 'area')
            assert np.array_equal(got, oracle.resize(a, d, oracle.RESIZE_AREA)), (dt.__name__, s, d)
    # identities of the published algorithm, against numpy (independent of the oracle)
    a = rng.random((37, 53)).astype(np.float32)
    for name in ('linear', 'cubic', 'lanczos4', 'area'):
        assert np.array_equal(ia.ops.resize(a, a.shape, name), a), name
    b = rng.random((36, 52))
    assert_close(ia.ops.resize(b, (12, 13), 'area'), b.reshape(12, 3, 13, 4).mean(axis=(1, 3)), 1e-6)
    ramp = np.tile(np.arange(20, dtype=np.float32), (6, 1))
    up = ia.ops.resize(ramp, (12, 40), 'linear')
    assert_close(up[3, 1:-1], ((np.arange(40) + 0.5) / 2 - 0.5)[1:-1], 1e-6)
    assert up[3, 0] == 0 and up[3, -1] == 19
    with pytest.raises(NotImplementedError):
        ia.ops.resize(a, (80, 120), 'area')
    with pytest.raises(NotImplementedError):
        ia.ops.resize(ia.default_context(0).to_device(np.zeros((8, 8), np.uint8)), (4, 4))


def test_fast_filter_and_fast_mean_end_to_end(ia, oracle):
    from imgprocessor_amd.filters import fastFilter, fastMean
    rng = np.random.default_rng(5)
    img = rng.random((240, 342)) * 100
    img[100:104] = np.nan
    img[:, 110:113] = np.nan
    for kw in (dict(ksize=30), dict(ksize=40, every=2, fn='nanmedian'),
               dict(ksize=40, every=2, fn='nanmean'), dict(ksize=20, every=5, fn='nanmean', smoothksize=1),
               dict(ksize=12, every=4, fn='nanmedian', interpolation=1)):
        src = img if kw.get('fn', 'median').startswith('nan') else np.nan_to_num(img, nan=50.0)
        okw = dict(kw)
        got = fastFilter(src, **kw)
        want = oracle.fastFilter(src, **okw)
        assert got.shape == src.shape and got.dtype == np.float64
        assert_close(got, want, 1e-11, 1e-11, 'fastFilter %s' % kw)
    # the reference's vignetting use (camera/flatField/vignettingFromRandomSteps.py:287-294):
    # f = max(shape) / 9, every = f / 3.5
    s0, s1 = img.shape
    f = int(max(s0, s1) / 9)
    ff = fastFilter(np.nan_to_num(img, nan=50.0), f, int(f / 3.5))
    assert ff.shape == img.shape and np.isfinite(ff).all()
    for dt in (np.float32, np.float64):
        a = np.nan_to_num(img, nan=50.0).astype(dt)
        for fac in (10, 20, 7.3):
            got = fastMean(a, fac)
            assert got.dtype == dt
            assert np.array_equal(got, oracle.fastMean(a, fac)), (dt.__name__, fac)
    # integer images keep their dtype like cv2.resize - BOTH times: the INTER_AREA result is an integer
    # image (rounded half to even, saturated) before INTER_LINEAR enlarges it, and so is the result
    for dt in (np.uint8, np.uint16):
        ui = (rng.random((100, 70)) * np.iinfo(dt).max).astype(dt)
        mx = np.iinfo(dt).max
        small = oracle.resize(ui.astype(np.float32), (5, 4), oracle.RESIZE_AREA)
        small = np.clip(np.rint(small), 0, mx).astype(np.float32)
        want = np.clip(np.rint(oracle.resize(small, (100, 70), oracle.RESIZE_LINEAR)), 0, mx).astype(dt)
        got = fastMean(ui, 20)
        assert got.dtype == dt and np.array_equal(got, want)
        ci = ui.copy()
        assert fastMean(ci, 20, inplace=True) is ci and np.array_equal(ci, want)
    # an exact 2 x 2 reduction with INTER_LINEAR is the area average (OpenCV's rule)
    from imgprocessor_amd import ops
    sq = rng.random((64, 96)).astype(np.float32)
    assert np.array_equal(ops.resize(sq, (32, 48), 'linear'), ops.resize(sq, (32, 48), 'area'))
    assert np.array_equal(ops.resize(sq, (32, 48), 'linear'), oracle.resize(sq, (32, 48), oracle.RESIZE_LINEAR))
    with pytest.raises(ValueError):
        ops.resize(sq, (32, 48), 0)
    c = a.copy()
    assert fastMean(c, 10, inplace=True) is c and np.array_equal(c, oracle.fastMean(a, 10))
    # device arrays stay on the device
    ctx = ia.default_context(0)
    d = ctx.to_device(a.astype(np.float32))
    r = ia.ops.resize(d, (24, 34), 'area')
    assert np.array_equal(r.get(), oracle.resize(a.astype(np.float32), (24, 34), oracle.RESIZE_AREA))
    # ... through fastMean and fastFilter too (same bits as from host arrays)
    a32 = a.astype(np.float32)
    m = fastMean(d, 10)
    assert isinstance(m, ia.DeviceArray) and np.array_equal(m.get(), oracle.fastMean(a32, 10))
    d2 = ctx.to_device(a32)
    assert fastMean(d2, 7.3, inplace=True) is d2 and np.array_equal(d2.get(), oracle.fastMean(a32, 7.3))
    src = np.nan_to_num(img, nan=50.0)
    for kw in (dict(ksize=30), dict(ksize=12, every=4, fn='mean', interpolation=1),
               dict(ksize=20, every=5, fn='median', smoothksize=1), dict(ksize=30, resize=False)):
        got = fastFilter(ctx.to_device(src), **kw)
        host = fastFilter(src, **kw)
        if kw.get('smoothksize') or not kw.get('resize', True):
            assert isinstance(got, np.ndarray)      # the small grid comes back to the host
        else:
            assert isinstance(got, ia.DeviceArray)
            got = got.get()
        assert got.dtype == np.float64 and np.array_equal(got, host), kw


def test_resize_vs_second_restatement(ia):
    """the GPU against the numpy restatement's fixture (cv_resize.npz), independent of oracle.c"""
    from .test_oracle_golden import cv_resize_cases
    g, pin = load_cv_golden('cv_resize.npz')
    print(pin)
    n = 0
    for key, src, dsize, kind, _ in cv_resize_cases(g):
        got = ia.ops.resize(src, dsize, kind)
        if kind == 'lanczos4':
            assert_close(got, g[key], 0, 2e-6, key)
        else:
            assert np.array_equal(got, g[key]), key
        n += 1
    assert n == 34


def test_resize_tables_kept_between_calls(ia, oracle):
    """the coefficient tables of the last resize stay on the device for the next call of the same
    shape; other shapes, interpolations and other uploads in between must not be served stale
    tables"""
    ctx = ia.default_context(0)
    rng = np.random.default_rng(12)
    a = rng.random((90, 131)).astype(np.float32)
    b = rng.random((90, 131)).astype(np.float32)
    da, db = ctx.to_device(a), ctx.to_device(b)
    o = oracle
    seq = ((da, a, (200, 300), 'lanczos4', o.RESIZE_LANCZOS4), (db, b, (200, 300), 'lanczos4', o.RESIZE_LANCZOS4),
           (da, a, (200, 300), 'cubic', o.RESIZE_CUBIC), (da, a, (200, 301), 'cubic', o.RESIZE_CUBIC),
           (db, b, (200, 300), 'cubic', o.RESIZE_CUBIC), (da, a, (40, 57), 'area', o.RESIZE_AREA),
           (db, b, (200, 300), 'cubic', o.RESIZE_CUBIC), (db, b, (200, 300), 'cubic', o.RESIZE_CUBIC),
           (da, a, (30, 131), 'area', o.RESIZE_AREA), (da, a, (200, 300), 'linear', o.RESIZE_LINEAR),
           (db, b, (200, 300), 'linear', o.RESIZE_LINEAR))
    for i, (d, h, dsz, name, oi) in enumerate(seq):
        got = ia.ops.resize(d, dsz, name).get()
        assert np.array_equal(got, o.resize(h, dsz, oi), equal_nan=True), (i, dsz, name)
        if i == 6:   # another user of the table buffer in between
            src8 = (rng.random((64, 80)) * 255).astype(np.uint8)
            yy, xx = np.mgrid[0:64, 0:80].astype(np.float32)
            r = ia.ops.remap(src8, xx * 0.9 + 1.3, yy * 0.9 + 0.7, 'lanczos4')
            assert np.array_equal(r, o.remap(src8, xx * 0.9 + 1.3, yy * 0.9 + 0.7, o.LANCZOS4, o.CONSTANT, 0.0))
    # the top-left part of an array through its pitch (fastFilter's cropped grid)
    g = rng.random((20, 30))
    want = o.resize(np.ascontiguousarray(g[:19, :29]), (60, 90), o.RESIZE_LANCZOS4)
    got = ia.ops.resize(ctx.to_device(g), (60, 90), 'lanczos4', src_shape=(19, 29))
    assert np.array_equal(got.get(), want, equal_nan=True)
    assert np.array_equal(ia.ops.resize(g, (60, 90), 'lanczos4', src_shape=(19, 29)), want, equal_nan=True)
    with pytest.raises(ValueError):
        ia.ops.resize(ctx.to_device(g), (60, 90), 'linear', src_shape=(21, 30))
