"""GPU: the three further interpolate/ fills (interp_more.hip) and the flat-field warps of
vignettingFromRandomSteps against the reference's own output (tests/golden/interp_more.npz,
generated from the reference source through the numba shim) and against the oracle at sizes
and masks the fixtures do not reach."""
import numpy as np
import pytest

from .conftest import load_golden, assert_close
from .test_oracle_golden import interp_more_cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)  # raises without a gfx950 device: no fallback
    return imgprocessor_amd


def test_interp_more_golden(ia):
    from imgprocessor_amd import interpolate
    g = load_golden('interp_more.npz')
    cases = interp_more_cases(g)
    assert len(cases) == 20
    for key, run in cases:
        want = g[key]
        got = run(interpolate)
        assert got.dtype == want.dtype and got.shape == want.shape, key
        # float32 grids: the interpreted source accumulates `python float * np.float32` in
        # float32, numba (the reference as it really runs) and this build in float64
        tol = 1e-12 if want.dtype == np.float64 else (2e-6 if key.startswith('c32') else 2.5e-7)
        assert_close(got, want, tol, 1e-14 if tol < 1e-9 else 0, key)


def test_unstructured_idw_vs_oracle(ia, oracle):
    from imgprocessor_amd.interpolate import interpolate2dUnstructuredIDW
    rng = np.random.default_rng(5)
    for (h, w, n, power) in ((100, 200, 30, 2), (67, 65, 1, 2), (33, 130, 257, 1.5), (5, 3, 4, 3)):
        x = rng.integers(0, h, n).astype(float)
        y = rng.integers(0, w, n).astype(float)
        x[::3] += rng.random(x[::3].size)   # a third of the points off the pixel grid
        v = rng.standard_normal(n)
        for dt, tol in ((np.float64, 1e-12), (np.float32, 2.5e-7)):
            grid = np.zeros((h, w), dt)
            got = interpolate2dUnstructuredIDW(x, y, v, grid, power)
            assert got is grid                # in place, returns the grid
            want = oracle.interpolate2dUnstructuredIDW(x, y, v, np.zeros((h, w), dt), power)
            # power 2 is sum-for-sum the reference's arithmetic
            assert_close(got, want, tol if power != 2 or dt == np.float32 else 1e-15, 0,
                         'unstructured %s' % ((h, w, n, power),))
    # duplicate points on one pixel: the first one wins (the reference breaks out of its loop)
    grid = np.zeros((8, 8))
    interpolate2dUnstructuredIDW([2, 2, 5], [3, 3, 1], [7.0, 9.0, 1.0], grid)
    assert grid[2, 3] == 7.0 and grid[5, 1] == 1.0
    with pytest.raises(ValueError):
        interpolate2dUnstructuredIDW([], [], [], grid)
    with pytest.raises(TypeError):
        interpolate2dUnstructuredIDW([1], [1], [1], np.zeros((4, 4), np.int32))


def test_unstructured_idw_device_array(ia, oracle):
    ctx = ia.default_context(0)
    rng = np.random.default_rng(6)
    h, w, n = 1000, 2000, 30   # the reference's own demo size (:44-56)
    x, y, v = rng.integers(0, h, n), rng.integers(0, w, n), rng.integers(0, 10, n)
    d = ctx.empty((h, w), np.float32)
    ia.ops.unstructured_idw(x, y, v, d, 2)
    want = oracle.interpolate2dUnstructuredIDW(x, y, v, np.zeros((h, w), np.float32), 2)
    assert_close(d.get(), want, 2.5e-7, 0, 'device grid')


def test_circular_idw_vs_oracle(ia, oracle):
    from imgprocessor_amd.interpolate import interpolateCircular2dStructuredIDW
    rng = np.random.default_rng(7)
    for (h, w, k, power, fr, fphi) in ((70, 70, 4, 2, 1, 0.2), (65, 130, 15, 2, 1, 1),
                                       (9, 9, 3, 3, 0.5, 2), (130, 131, 6, 1, 1, 0.3)):
        grid = rng.random((h, w))
        mask = rng.random((h, w)) < 0.3       # masked pixels right up to every edge
        cx, cy = h // 2 + 1, w // 2 + 1
        got = interpolateCircular2dStructuredIDW(grid.copy(), mask, k, power, fr, fphi, cx, cy)
        want = oracle.interpolateCircular2dStructuredIDW(grid.copy(), mask, k, power, fr, fphi,
                                                         cx, cy)
        assert_close(got, want, 1e-11, 1e-14, 'circular %s' % ((h, w, k, power),))
        # columns >= shape[0] stay as they were (gy = grid.shape[0] in the source)
        assert np.array_equal(got[:, h:], grid[:, h:])
        g32 = grid.astype(np.float32)
        got = interpolateCircular2dStructuredIDW(g32.copy(), mask, k, power, fr, fphi, cx, cy)
        want = oracle.interpolateCircular2dStructuredIDW(g32.copy(), mask, k, power, fr, fphi,
                                                         cx, cy)
        assert_close(got, want, 2.5e-7, 0, 'circular32')
    with pytest.raises(ValueError):   # fewer columns than rows: out of bounds in the reference
        interpolateCircular2dStructuredIDW(np.zeros((8, 6)), np.zeros((8, 6), bool))
    # a fully masked neighbourhood leaves the pixel untouched (`if sumWi:`)
    grid = np.full((20, 20), 3.0)
    mask = np.zeros((20, 20), bool)
    mask[5:15, 5:15] = True
    out = interpolateCircular2dStructuredIDW(grid.copy(), mask, 2, 2, 1, 1, 10, 10)
    assert out[10, 10] == 3.0 and out[8, 8] == 3.0 and out[5, 5] != 3.0


def test_cross_avg_vs_oracle(ia, oracle):
    from imgprocessor_amd.interpolate import interpolate2dStructuredCrossAvg
    rng = np.random.default_rng(9)
    for (h, w, k, power, dens) in ((200, 200, 20, 2, 0.0), (90, 150, 5, 1, 0.3), (150, 90, 7, 2, 0.3),
                                   (33, 70, 3, 3, 0.6), (64, 64, 0, 2, 0.2)):
        grid = rng.random((h, w)) + np.linspace(5, 10, w)[None, :]
        mask = rng.random((h, w)) < dens      # masked pixels right up to every edge
        mask[h // 4:3 * h // 4, w // 4:3 * w // 4] = True   # the reference's demo hole (:127-129)
        mask[0, :] = False
        mask[:, 0] = False
        mask[h // 2, :w // 4] = True          # a row whose left end is masked: the stale slot
        mask[h // 2 + 3, :2] = True
        for dt, tol in ((np.float64, 1e-12), (np.float32, 2.5e-7)):
            g = grid.astype(dt)
            got = interpolate2dStructuredCrossAvg(g.copy(), mask, k, power)
            want = oracle.interpolate2dStructuredCrossAvg(g.copy(), mask, k, power)
            assert got.dtype == dt
            assert_close(got, want, tol, 0, 'cross %s %s' % ((h, w, k, power), dt.__name__))
    # nothing unmasked along the pixel's row and column: the empty blend writes 0
    grid = np.ones((6, 6))
    mask = np.zeros((6, 6), bool)
    mask[2, :] = True
    mask[:, 3] = True
    got = interpolate2dStructuredCrossAvg(grid.copy(), mask, 2, 2)
    want = oracle.interpolate2dStructuredCrossAvg(grid.copy(), mask, 2, 2)
    assert_close(got, want, 1e-12, 0, 'cross isolated')
    assert got[2, 3] == 0.0
    # masked pixels in column 0 before any pixel has filled slot 2: the slot is left out
    mask = np.zeros((6, 6), bool)
    mask[1, 0] = mask[1, 1] = True
    got = interpolate2dStructuredCrossAvg(grid.copy() * 2, mask, 2, 2)
    want = oracle.interpolate2dStructuredCrossAvg(grid.copy() * 2, mask, 2, 2)
    assert_close(got, want, 1e-12, 0, 'cross first')


def test_cross_avg_device_arrays(ia, oracle):
    ctx = ia.default_context(0)
    rng = np.random.default_rng(10)
    h, w = 300, 500
    grid = rng.random((h, w)).astype(np.float32)
    mask = np.zeros((h, w), bool)
    mask[50:150, 100:300] = True
    d, dm = ctx.to_device(grid), ctx.to_device(mask.astype(np.uint8))
    ia.ops.cross_avg_fill(d, dm, 20, 2)
    want = oracle.interpolate2dStructuredCrossAvg(grid.copy(), mask, 20, 2)
    assert_close(d.get(), want, 2.5e-7, 0, 'device cross')
    ia.ops.circular_idw_fill(d, dm, 3, 2, 1, 0.5, 150, 250)   # nothing unmasked nearby inside
    assert np.isfinite(d.get()).all()


def test_vignetting_random_steps_warps(ia, oracle):
    """camera/flatField/vignettingFromRandomSteps.py:245,260,309: plain cv2.warpPerspective
    calls (INTER_LINEAR at 1/32 px, constant border 0) on float64 images, one of them with the
    background set to NaN and nan_to_num afterwards"""
    from imgprocessor_amd.camera.flatField.vignettingFromRandomSteps import (
        warpFlatField, warpRatioToFlatField, fitToObject)
    rng = np.random.default_rng(11)
    H, W = 120, 160
    ff = np.clip(1 - 0.5 * ((np.mgrid[0:H, 0:W][0] - 60) ** 2 +
                            (np.mgrid[0:H, 0:W][1] - 80) ** 2) / 1e4, 0, 1)   # float64
    a = np.deg2rad(3.0)
    Hm = np.array([[np.cos(a), -np.sin(a), 6.0], [np.sin(a), np.cos(a), -4.0], [1e-4, -5e-5, 1.0]])
    Hinv = np.linalg.inv(Hm)
    q5 = oracle.LINEAR | oracle.Q5
    # :245 - cv2 inverts the matrix it is given: destination -> source is inv(Hinv) = Hm
    got = warpFlatField(ff, Hinv, (100, 140))
    want = oracle.warp_perspective(ff, Hm, (100, 140), q5, oracle.CONSTANT, 0.0)
    assert got.dtype == np.float64 and got.shape == (100, 140)
    assert_close(got, want, 1e-12, 1e-14, 'warpFlatField')
    # :309
    img = rng.random((H, W)).astype(np.float32)
    got = fitToObject(img, Hinv, (90, 130))
    want = oracle.warp_perspective(img, Hm, (90, 130), q5, oracle.CONSTANT, 0.0)
    assert got.dtype == np.float32
    assert_close(got, want, 1e-5, 1e-5, 'fitToObject')
    # :255-262 - NaN background: a bilinear footprint that touches a NaN is NaN (also with
    # weight 0), NaN becomes 0 afterwards
    fit = rng.random((100, 140)) + 0.5
    obj = rng.random((100, 140)) + 0.5
    obj[10, 10] = 0.0                       # division by zero -> inf, as in the reference
    fmask = np.zeros((100, 140), bool)
    fmask[:12, :] = True
    fmask[40:60, 50:90] = True
    got = warpRatioToFlatField(fit, obj, fmask, Hm, (H, W))
    with np.errstate(divide='ignore', invalid='ignore'):
        div = fit / obj
    div[fmask] = np.nan
    want = np.nan_to_num(oracle.warp_perspective(div, Hinv, (H, W), q5, oracle.CONSTANT, 0.0))
    assert_close(got, want, 1e-12, 1e-14, 'warpRatioToFlatField')
    assert (got == 0).any() and (got != 0).any()
    # integer-position footprints next to a NaN: weight-0 taps still poison the pixel
    src = np.ones((8, 8))
    src[3, 4] = np.nan
    out = ia.ops.warp_perspective(src, np.eye(3), (8, 8), 'linear_cv_q5')
    ref = oracle.warp_perspective(src, np.eye(3), (8, 8), q5, oracle.CONSTANT, 0.0)
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    assert np.isnan(out[3, 4]) and np.isnan(out[3, 3]) and np.isnan(out[2, 4])


def test_point_spread_idw_golden_and_oracle(ia, oracle):
    """interpolate2dStructuredPointSpreadIDW: against the reference's own output (point_spread.npz)
    and, on larger grids with big holes, masked rims and more rows than 16 waves, against the oracle;
    the sums run over the lanes of a wave instead of raster order: float64 within a few ulps, the
    values a sweep leaves feed the next"""
    from imgprocessor_amd.interpolate import interpolate2dStructuredPointSpreadIDW as psidw
    g = load_golden('point_spread.npz')
    for name in ('sq', 'wide'):
        grid, mask = g['ps_grid_' + name], g['ps_mask_' + name]
        for kern, power in ((5, 2), (3, 1), (8, 3)):
            got = psidw(grid, mask, kern, power)
            assert_close(got, g['ps_%s_k%d_p%d' % (name, kern, power)], 1e-11, 0,
                         'golden %s k%d p%d' % (name, kern, power))
        assert mask.any()
    got = psidw(g['ps_grid_edge'], g['ps_mask_edge'], 4, 2)
    assert_close(got, g['ps_edge_k4_p2'], 1e-11, 0, 'golden edge')
    got = psidw(g['ps_grid_sq'].astype(np.float32), g['ps_mask_sq'], 5, 2)
    assert got.dtype == np.float32
    assert_close(got, g['ps32_sq_k5_p2'], 2e-6, 0, 'golden float32')
    gr, m = g['ps_grid_sq'].copy(), g['ps_mask_sq'].copy()
    r = psidw(gr, m, 5, 2, copy=False)
    assert r is gr and not m.any()
    assert_close(gr, g['ps_sq_k5_p2'], 1e-11, 0, 'in place')
    rng = np.random.default_rng(4)
    for (h, w, k, power, dens) in ((150, 210, 15, 2, 0.2), (96, 96, 4, 1, 0.6), (70, 300, 70, 3, 0.1),
                                   (40, 64, 2, 2.5, 0.9)):
        grid = rng.random((h, w)) + np.linspace(5, 10, w)[None, :]
        mask = rng.random((h, w)) < dens
        mask[h // 4:3 * h // 4, w // 4:3 * w // 4] = True
        mask[0, 0] = False
        mask[:, -1] |= rng.random(h) < 0.5     # rows that end masked / unmasked: the -1 quirk
        mask[-1, :] |= rng.random(w) < 0.5
        for dt, tol in ((np.float64, 1e-10), (np.float32, 3e-6)):
            got = psidw(grid.astype(dt), mask, k, power)
            want = oracle.interpolate2dStructuredPointSpreadIDW(grid.astype(dt), mask, k, power)
            assert_close(got, want, tol, 0, 'point spread %s %s' % ((h, w, k, power), dt.__name__))
    # device arrays, limited sweeps: the mask that is left matches the oracle's
    ctx = ia.default_context(0)
    grid = rng.random((120, 160))
    mask = np.zeros((120, 160), bool)
    mask[30:90, 40:120] = True
    dg, dm = ctx.to_device(grid), ctx.to_device(mask.view(np.uint8))
    out = psidw(dg, dm, 6, 2, maxIter=3, copy=False)
    wg, wm = grid.copy(), mask.copy()
    oracle.interpolate2dStructuredPointSpreadIDW(wg, wm, 6, 2, maxIter=3, copy=False)
    assert out is dg and np.array_equal(dm.get().astype(bool), wm) and wm.any()
    assert_close(dg.get(), wg, 1e-10, 0, 'three sweeps')
    # nothing masked / everything masked: nothing happens
    assert np.array_equal(psidw(grid, np.zeros_like(mask), 5, 2), grid)
    assert np.array_equal(psidw(grid, np.ones_like(mask), 5, 2), grid)
