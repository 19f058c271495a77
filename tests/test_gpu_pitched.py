"""GPU: the C ABI with row pitches larger than the width and frame strides larger than a frame
(a caller that passes regions of larger device buffers), for every kernel family behind the
remap / remap + filter entry points - plain, ring, frame-pair - against the same call on
contiguous copies: identical bits."""
import ctypes as C

import numpy as np
import pytest

from .gpu_helpers import frames, kern, radial_maps, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def _embed(a, pad_x, pad_rows, fill):
    """(n, h, w) -> buffer (n, h + pad_rows, w + pad_x) holding a in its top-left corner"""
    n, h, w = a.shape
    big = np.full((n, h + pad_rows, w + pad_x), fill, a.dtype)
    big[:, :h, :w] = a
    return big


@pytest.mark.parametrize('odd', [0, 1])
@pytest.mark.parametrize('interp', ['linear', 'cubic', 'lanczos4'])
@pytest.mark.parametrize('tune', [dict(ring_remap=0), dict(ring_remap=2), dict(tile_warp=2)])
def test_remap_with_pitches(ia, interp, tune, odd):
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import dtype_id
    ctx = ia.default_context(0)
    n, h, w = 3, 150, 520
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    old = ctx.set_tuning(ring_min=1, **tune)
    try:
        ref_knobs = ctx.set_tuning(tile_warp=0, ring_remap=0)   # the reference: the plain gather kernel
        want = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), interp).get()
        ctx.set_tuning(**{k: v for k, v in ref_knobs.items()})
        # pitches (elements): rows 16-byte aligned, or (odd) at every 4-byte alignment
        sp, dp, mp = w + 24 + odd, w + 8 + 3 * odd, w + 12 + odd
        sbig = ctx.to_device(_embed(src, sp - w, 5, 7.0))
        mbx = ctx.to_device(_embed(mx[None], mp - w, 0, -1e9)[0])
        mby = ctx.to_device(_embed(my[None], mp - w, 0, -1e9)[0])
        dbig = ctx.to_device(np.full((n, h + 3, dp), -5.0, np.float32))
        ctx._check(ctx._lib.ipa_remap_dev(
            ctx.handle, sbig.ptr, dtype_id(np.float32), h, w, sp, mbx.ptr, mby.ptr, mp, dbig.ptr,
            dtype_id(np.float32), h, w, dp, n, (h + 5) * sp, (h + 3) * dp, ops.interp_id(interp),
            ops.border_id('constant'), 0.0), 'remap')
        got = dbig.get()
    finally:
        ctx.set_tuning(**old)
    same_bits(np.ascontiguousarray(got[:, :h, :w]), want, 'pitched remap ' + interp)
    assert (got[:, h:, :] == -5.0).all() and (got[:, :, w:] == -5.0).all(), 'wrote outside'


@pytest.mark.parametrize('odd', [0, 1])
@pytest.mark.parametrize('interp', ['linear', 'linear_cv_q5', 'cubic_cv', 'lanczos4'])
def test_uint8_remap_with_pitches(ia, interp, odd):
    """the uint8 kernels (fixed-point bilinear, bicubic / Lanczos4 tables in LDS) on regions of
    larger byte buffers: row pitches at any byte alignment, frame strides larger than a frame"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import dtype_id
    ctx = ia.default_context(0)
    n, h, w = 3, 150, 777
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    mx, my, _, _ = radial_maps(h, w)
    want = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), interp, 'reflect').get()
    sp, dp, mp = w + 23 + odd, w + 8 + 3 * odd, w + 12 + odd
    sbig = ctx.to_device(_embed(src, sp - w, 5, 7))
    mbx = ctx.to_device(_embed(mx[None], mp - w, 0, -1e9)[0])
    mby = ctx.to_device(_embed(my[None], mp - w, 0, -1e9)[0])
    dbig = ctx.to_device(np.full((n, h + 3, dp), 99, np.uint8))
    ctx._check(ctx._lib.ipa_remap_dev(
        ctx.handle, sbig.ptr, dtype_id(np.uint8), h, w, sp, mbx.ptr, mby.ptr, mp, dbig.ptr,
        dtype_id(np.uint8), h, w, dp, n, (h + 5) * sp, (h + 3) * dp, ops.interp_id(interp),
        ops.border_id('reflect'), 0.0), 'remap')
    got = dbig.get()
    assert np.array_equal(got[:, :h, :w], want), 'pitched uint8 remap ' + interp
    assert (got[:, h:, :] == 99).all() and (got[:, :, w:] == 99).all(), 'wrote outside'


@pytest.mark.parametrize('odd', [0, 1])
@pytest.mark.parametrize('K', [5, 9])
def test_remap_conv_with_pitches(ia, K, odd):
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import dtype_id
    ctx = ia.default_context(0)
    n, h, w = 4, 140, 780
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    k = np.ascontiguousarray(kern(K), dtype=np.float64)
    old = ctx.set_tuning(ring_min=1)
    try:
        want = ops.remap_conv2d(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), k).get()
        sp, dp, mp = w + 20 + odd, w + 4 + odd, w + 16 + 3 * odd
        sbig = ctx.to_device(_embed(src, sp - w, 2, 3.0))
        mbx = ctx.to_device(_embed(mx[None], mp - w, 0, -1e9)[0])
        mby = ctx.to_device(_embed(my[None], mp - w, 0, -1e9)[0])
        dbig = ctx.to_device(np.full((n, h + 1, dp), -5.0, np.float32))
        cb = ops.border_id('reflect')
        ctx._check(ctx._lib.ipa_remap_conv2d_dev(
            ctx.handle, sbig.ptr, dtype_id(np.float32), h, w, sp, mbx.ptr, mby.ptr, mp,
            k.ctypes.data_as(C.POINTER(C.c_double)), K, K, dbig.ptr, dtype_id(np.float32), h, w, dp,
            n, (h + 2) * sp, (h + 1) * dp, ops.interp_id('linear'), ops.border_id('constant'), 0.0,
            cb, cb), 'remap_conv2d')
        got = dbig.get()
    finally:
        ctx.set_tuning(**old)
    same_bits(np.ascontiguousarray(got[:, :h, :w]), want, 'pitched remap_conv K=%d' % K)
    assert (got[:, h:, :] == -5.0).all() and (got[:, :, w:] == -5.0).all(), 'wrote outside'
