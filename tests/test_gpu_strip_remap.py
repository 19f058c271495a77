"""GPU: standalone bilinear remaps of uint16 frames INTO float32 (camera frames as transformations.toFloatArray ingests them,
transformations.py:78-87; LensDistortion.correct / PerspectiveCorrection.correct on them, camera/LensDistortion.py:316-330,
camera/PerspectiveCorrection.py:374-406) run on the marching strips of the fused chains with no filter (round 6, knob
strip_remap; csrc/remap.hip::strip_remap_takes, wave_sep_kernel with K = 1): the bits of the gather kernels they replace
(knob off), the oracle's values, the route taken where it is claimed (read-only counter "strip_remaps") and nowhere else."""
import ctypes as C

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


def taken(ctx):
    return ctx.get_tuning('strip_remaps')


M_UP = np.array([[1.01, 0.004, -3.0], [-0.003, 0.99, 2.0], [1e-5, -2e-5, 1.0]])


def rot(deg, h, w):
    a = np.deg2rad(deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    return np.array([[np.cos(a), -np.sin(a), cx - np.cos(a) * cx + np.sin(a) * cy],
                     [np.sin(a), np.cos(a), cy - np.sin(a) * cx - np.cos(a) * cy], [0, 0, 1.0]])


@pytest.mark.parametrize('shape', [(150, 610), (97, 333), (301, 1030)])
@pytest.mark.parametrize('n', [4, 8, 3, 7])
@pytest.mark.parametrize('interp', ['linear', 'linear_cv_q5'])
def test_strip_remap_is_the_gather_remap(ia, oracle, shape, n, interp):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w, np.uint16)
    mx, my, Kc, dist = radial_maps(h, w, shift=2.3)
    mx = mx - np.float32(15.0)       # a rim outside the source
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    oi = oracle.LINEAR | (oracle.Q5 if interp.endswith('q5') else 0)
    for border in ('constant', 'replicate', 'reflect', 'wrap', 'reflect101'):
        calls = {
            # (counts that are no multiple of 4: from 7 frames on - whole workgroups + the last four frames again)
            'maps': (lambda: ops.remap(d, dmx, dmy, interp, border, 100.0, out_dtype=np.float32), n % 4 == 0 or n >= 7),
            'lens model': (lambda: ops.undistort(d, Kc, dist, Kc, interp, border, 100.0, out_dtype=np.float32), True),
            'homography': (lambda: ops.warp_perspective(d, M_UP, (h, w), interp, border, 100.0, out_dtype=np.float32), n % 4 == 0 or n >= 7),
        }
        for name, (fn, expect) in calls.items():
            before = taken(ctx)
            got = fn().get()
            assert taken(ctx) == before + (1 if expect else 0), (name, n, 'route')
            old = ctx.set_tuning(strip_remap=0)
            try:
                before = taken(ctx)
                ref = fn().get()
                assert taken(ctx) == before
            finally:
                ctx.set_tuning(**old)
            assert got.dtype == np.float32
            same_bits(got, ref, '%s, %s, %s, %d frames of %d x %d' % (name, interp, border, n, h, w))
        if border in ('constant', 'reflect'):
            f = n - 1
            bo = {'constant': oracle.CONSTANT, 'reflect': oracle.REFLECT}[border]
            want = oracle.remap(src[f], mx, my, oi, bo, 100.0, out_dtype=np.float32)
            got = ops.remap(d, dmx, dmy, interp, border, 100.0, out_dtype=np.float32).get()
            assert_close(got[f], want, 1e-5, 1e-5 * 4095, 'maps vs oracle, %s' % border)


def test_what_stays_with_the_other_kernels(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 120, 520
    u16 = ctx.to_device(frames(4, h, w, np.uint16))
    f32 = ctx.to_device(frames(4, h, w))
    mx, my, _, _ = radial_maps(h, w)
    dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
    before = taken(ctx)
    ops.remap(f32, dmx, dmy)                                           # float32 frames: the tile kernel
    ops.remap(u16, dmx, dmy)                                           # uint16 -> uint16
    ops.remap(u16, dmx, dmy, 'cubic', out_dtype=np.float32)            # bicubic taps
    ops.remap(u16, dmx, dmy, 'nearest', out_dtype=np.float32)
    ops.warp_perspective(u16, rot(12.0, h, w), (h, w), 'linear', out_dtype=np.float32)   # a picture that turns
    assert taken(ctx) == before
    ops.warp_perspective(u16, rot(0.2, h, w), (h, w), 'linear', out_dtype=np.float32)
    assert taken(ctx) == before + 1


def test_strip_remap_with_pitches(ia):
    """the C ABI with pitches and frame strides larger than the frames (regions of larger buffers)"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import dtype_id
    ctx = ia.default_context(0)
    n, h, w = 4, 130, 500
    src = frames(n, h, w, np.uint16)
    mx, my, _, _ = radial_maps(h, w)
    want = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), out_dtype=np.float32).get()
    sp, dp, mp = w + 26, w + 8, w + 12
    sbig = np.full((n, h + 5, sp), 7, np.uint16)
    sbig[:, :h, :w] = src
    mbx, mby = np.full((h, mp), -1e9, np.float32), np.full((h, mp), -1e9, np.float32)
    mbx[:, :w], mby[:, :w] = mx, my
    dbig = ctx.to_device(np.full((n, h + 3, dp), -5.0, np.float32))
    dsb, dmbx, dmby = ctx.to_device(sbig), ctx.to_device(mbx), ctx.to_device(mby)
    before = taken(ctx)
    ctx._check(ctx._lib.ipa_remap_dev(
        ctx.handle, dsb.ptr, dtype_id(np.uint16), h, w, sp, dmbx.ptr, dmby.ptr, mp, dbig.ptr,
        dtype_id(np.float32), h, w, dp, n, (h + 5) * sp, (h + 3) * dp, ops.interp_id('linear'),
        ops.border_id('constant'), C.c_double(0.0)), 'remap')
    assert taken(ctx) == before + 1
    got = dbig.get()
    same_bits(np.ascontiguousarray(got[:, :h, :w]), want, 'pitched strip remap')
    assert (got[:, h:, :] == -5.0).all() and (got[:, :, w:] == -5.0).all(), 'wrote outside'


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16])
@pytest.mark.parametrize('interp', ['linear', 'linear_cv_q5', 'cubic_cv_q5', 'lanczos4', 'nearest'])
def test_integer_batches_undistort_through_the_cached_map(ia, oracle, dtype, interp):
    """round 6: the lens model BY VALUE on batches of integer frames goes through the context's cached map (the model's
    float32 coordinates are the same for every frame; bit for bit what the per-pixel evaluation gives) instead of being
    evaluated per pixel and frame: 64 x 4K uint16 1.77 -> 1.31 ms.  Same bits as the analytic kernels (knob lens_cache = 0)."""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 141, 530, 5
    src = frames(n, h, w, dtype)
    _, _, Kc, dist = radial_maps(h, w)
    newK = Kc.copy()
    newK[0, 0] *= 0.93
    newK[1, 1] *= 0.93          # a rim outside the source
    d = ctx.to_device(src)
    for odt in (None, np.float32):
        got = ops.undistort(d, Kc, dist, newK, interp, 'constant', 9.0, out_dtype=odt).get()
        old = ctx.set_tuning(lens_cache=0, strip_remap=0)
        try:
            ref = ops.undistort(d, Kc, dist, newK, interp, 'constant', 9.0, out_dtype=odt).get()
        finally:
            ctx.set_tuning(**old)
        if got.dtype == np.float32:
            same_bits(got, ref, '%s %s -> float32' % (np.dtype(dtype).name, interp))
        else:
            assert got.dtype == dtype and np.array_equal(got, ref), (dtype, interp)
    oi = {'linear': oracle.LINEAR, 'linear_cv_q5': oracle.LINEAR | oracle.Q5, 'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5,
          'lanczos4': oracle.LANCZOS4, 'nearest': oracle.NEAREST}[interp]
    mx, my = oracle.build_undistort_map(Kc, dist, newK, h, w)
    got = ops.undistort(d, Kc, dist, newK, interp, 'constant', 9.0).get()
    assert np.array_equal(got[n - 1], oracle.remap(src[n - 1], mx, my, oi, oracle.CONSTANT, 9.0))


@pytest.mark.parametrize('shape', [(150, 612), (97, 336), (301, 1032), (97, 333)])
@pytest.mark.parametrize('n', [4, 8, 12, 6, 7, 9])
def test_uint16_into_uint16_with_cv2_arithmetic(ia, oracle, shape, n):
    """round 6: cv2.remap on uint16 frames RETURNS uint16 - what LensDistortion.correct gives for camera frames
    (camera/LensDistortion.py:323-326; the wrapper asks for cv2's 16U arithmetic at 1/32-px coordinates,
    'linear_cv_q5').  Batches of a multiple of 4 frames with rows of whole vectors run on the marching strips with that
    arithmetic (wave_pipe.hpp CV16): the gather kernel's integers and the oracle's, bit for bit."""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w, np.uint16)
    src[:, 0, 0] = src[:, -1, -1] = 65535            # the corners, and saturation
    mx, my, Kc, dist = radial_maps(h, w, shift=2.3)
    mx = mx - np.float32(15.0)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    expect = (n % 4 == 0 or n >= 7) and w % 4 == 0
    for border in ('constant', 'replicate', 'reflect', 'wrap', 'reflect101'):
        for name, fn in (('maps', lambda: ops.remap(d, dmx, dmy, 'linear_cv_q5', border, 17.6)),
                         ('lens model', lambda: ops.undistort(d, Kc, dist, Kc, 'linear_cv_q5', border, 17.6))):
            before = taken(ctx)
            got = fn().get()
            assert taken(ctx) == before + (1 if expect else 0), (name, shape, n)
            old = ctx.set_tuning(strip_remap=0)
            try:
                ref = fn().get()
            finally:
                ctx.set_tuning(**old)
            assert got.dtype == np.uint16 and np.array_equal(got, ref), (name, border, shape, n)
        want = oracle.remap(src[n - 1], mx, my, oracle.LINEAR | oracle.Q5, {'constant': oracle.CONSTANT, 'replicate': oracle.REPLICATE,
                            'reflect': oracle.REFLECT, 'wrap': oracle.WRAP, 'reflect101': oracle.REFLECT101}[border], 17.6)
        got = ops.remap(d, dmx, dmy, 'linear_cv_q5', border, 17.6).get()
        assert np.array_equal(got[n - 1], want), (border, shape, n)
    # exact coordinates ('linear' on integer frames: double arithmetic) is not this route
    before = taken(ctx)
    ops.remap(d, dmx, dmy, 'linear', 'constant', 0.0)
    assert taken(ctx) == before


def test_lens_distortion_correct_on_a_uint16_batch(ia, oracle):
    """the reference's call itself: LensDistortion.correct on (n, h, w) camera frames"""
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    ctx = ia.default_context(0)
    h, w, n = 120, 640, 8
    ld = LensDistortion()
    ld.setCameraParams(float(w), float(w), (w - 1) / 2.0, (h - 1) / 2.0, -0.12, 0.03, 0.0, 1e-3, -5e-4)
    src = frames(n, h, w, np.uint16)
    before = taken(ctx)
    got = ld.correct(ctx.to_device(src), keepSize=True)
    got = got.get() if hasattr(got, 'get') else got
    assert got.dtype == np.uint16 and got.shape == (n, h, w)
    old = ctx.set_tuning(strip_remap=0)
    try:
        ref = ld.correct(ctx.to_device(src), keepSize=True)
        ref = ref.get() if hasattr(ref, 'get') else ref
    finally:
        ctx.set_tuning(**old)
    assert np.array_equal(got, ref)
    assert taken(ctx) > before


@pytest.mark.parametrize('shape', [(150, 608), (97, 336), (301, 1040), (97, 340), (97, 342)])
@pytest.mark.parametrize('n', [4, 8, 12, 6, 7, 9])
def test_uint8_into_uint8_with_cv2_fixed_point(ia, oracle, shape, n):
    """... and 8-bit camera frames: cv2.remap's 8U bilinear (15-bit fixed-point weights from the 1/32-px fractions, rounded
    shift - every bilinear remap of uint8 frames is that arithmetic, with or without the q5 flag) on the same strips: one
    16-bit load per tap row, a lane's four results as one dword.  Integers of the gather kernel and of the oracle."""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w, np.uint8)
    src[:, 0, 0] = src[:, -1, -1] = 255
    mx, my, Kc, dist = radial_maps(h, w, shift=2.3)
    mx = mx - np.float32(15.0)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    expect = (n % 4 == 0 or n >= 7) and w % 4 == 0          # (rows of whole dwords)
    for interp in ('linear', 'linear_cv_q5'):
        for border in ('constant', 'replicate', 'reflect', 'wrap', 'reflect101'):
            for name, fn in (('maps', lambda: ops.remap(d, dmx, dmy, interp, border, 17.6)),
                             ('lens model', lambda: ops.undistort(d, Kc, dist, Kc, interp, border, 17.6))):
                before = taken(ctx)
                got = fn().get()
                assert taken(ctx) == before + (1 if expect else 0), (name, shape, n, interp)
                old = ctx.set_tuning(strip_remap=0)
                try:
                    ref = fn().get()
                finally:
                    ctx.set_tuning(**old)
                assert got.dtype == np.uint8 and np.array_equal(got, ref), (name, interp, border, shape, n)
    want = oracle.remap(src[n - 1], mx, my, oracle.LINEAR | oracle.Q5, oracle.CONSTANT, 17.6)
    got = ops.remap(d, dmx, dmy, 'linear', 'constant', 17.6).get()
    assert np.array_equal(got[n - 1], want), (shape, n)


@pytest.mark.parametrize('dtype', [np.uint16, np.uint8])
@pytest.mark.parametrize('pads', [(24, 8, 12), (25, 9, 13), (32, 16, 16), (26, 10, 12)])
def test_integer_strip_remap_with_pitches(ia, dtype, pads):
    """the integer forms through the C ABI with pitches and frame strides larger than the frames: rows of whole dwords take
    the strips, others the gather kernel - the same integers either way, nothing written outside the frames"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.device import dtype_id
    ctx = ia.default_context(0)
    n, h, w = 8, 130, 512
    src = frames(n, h, w, dtype)
    mx, my, _, _ = radial_maps(h, w)
    interp = 'linear_cv_q5'
    old = ctx.set_tuning(strip_remap=0)
    try:
        want = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), interp, 'constant', 5.0).get()
    finally:
        ctx.set_tuning(**old)
    sp, dp, mp = w + pads[0], w + pads[1], w + pads[2]
    sbig = np.full((n, h + 5, sp), 7, dtype)
    sbig[:, :h, :w] = src
    mbx, mby = np.full((h, mp), -1e9, np.float32), np.full((h, mp), -1e9, np.float32)
    mbx[:, :w], mby[:, :w] = mx, my
    dbig = ctx.to_device(np.full((n, h + 3, dp), 201, dtype))
    dsb, dmbx, dmby = ctx.to_device(sbig), ctx.to_device(mbx), ctx.to_device(mby)
    before = taken(ctx)
    ctx._check(ctx._lib.ipa_remap_dev(
        ctx.handle, dsb.ptr, dtype_id(dtype), h, w, sp, dmbx.ptr, dmby.ptr, mp, dbig.ptr,
        dtype_id(dtype), h, w, dp, n, (h + 5) * sp, (h + 3) * dp, ops.interp_id(interp),
        ops.border_id('constant'), C.c_double(5.0)), 'remap')
    got = dbig.get()
    assert np.array_equal(got[:, :h, :w], want), (dtype, pads)
    assert (got[:, h:, :] == 201).all() and (got[:, :, w:] == 201).all(), 'wrote outside'
    # rows and frames of the result that start on a dword take the strips (uint16 rows then start on any 4 bytes: the
    # two-dword stores need no more), the others the gather kernel
    row_bytes = dp * np.dtype(dtype).itemsize
    assert taken(ctx) - before == (1 if row_bytes % 4 == 0 and ((h + 3) * row_bytes) % 4 == 0 else 0), (dtype, pads)


@pytest.mark.parametrize('shape', [(150, 610), (97, 333), (301, 1030)])
@pytest.mark.parametrize('n', [4, 8, 3, 7])
def test_uint8_frames_into_float32(ia, oracle, shape, n):
    """8-bit camera frames through transformations.toFloatArray: uint8 -> float32 bilinear remaps with the map pair (and
    the lens model by value through its cached map) on the strips - bits of the gather kernel, values of the oracle"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = shape
    src = frames(n, h, w, np.uint8)
    src[:, 0, 0] = src[:, -1, -1] = 255
    mx, my, Kc, dist = radial_maps(h, w, shift=2.3)
    mx = mx - np.float32(15.0)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    for interp in ('linear', 'linear_cv_q5'):
        for border in ('constant', 'replicate', 'reflect', 'wrap', 'reflect101'):
            for name, fn, expect in (('maps', lambda: ops.remap(d, dmx, dmy, interp, border, 9.0, out_dtype=np.float32),
                                      n % 4 == 0 or n >= 7),
                                     ('lens model', lambda: ops.undistort(d, Kc, dist, Kc, interp, border, 9.0, out_dtype=np.float32),
                                      n >= 4 and (n % 4 == 0 or n >= 7))):
                before = taken(ctx)
                got = fn().get()
                assert taken(ctx) == before + (1 if expect else 0), (name, n, shape)
                old = ctx.set_tuning(strip_remap=0)
                try:
                    ref = fn().get()
                finally:
                    ctx.set_tuning(**old)
                same_bits(got, ref, '%s, %s, %s, %d uint8 frames of %d x %d' % (name, interp, border, n, h, w))
    want = oracle.remap(src[n - 1], mx, my, oracle.LINEAR, oracle.CONSTANT, 9.0, out_dtype=np.float32)
    got = ops.remap(d, dmx, dmy, 'linear', 'constant', 9.0, out_dtype=np.float32).get()
    assert_close(got[n - 1], want, 1e-5, 1e-5 * 255, 'uint8 -> float32 vs oracle')


@pytest.mark.parametrize('dtype', [np.uint16, np.uint8])
def test_full_size_camera_batch_every_frame(ia, oracle, dtype):
    """16 x 4K camera frames through LensDistortion.correct's remap at full size, every frame against the oracle's integers
    (the size other_configs of bench.py times: 'LensDistortion.correct 4K uint16 -> uint16 ...')"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 2160, 3840, 16
    Kc = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    newK = Kc.copy()
    newK[0, 0] *= 0.97
    newK[1, 1] *= 0.97              # a rim of border pixels, all four source corners inside the picture
    mx, my = oracle.build_undistort_map(Kc, dist, newK, h, w)
    rng = np.random.default_rng(11)
    top = 65535 if dtype == np.uint16 else 255
    src = rng.integers(0, top + 1, (n, h, w), dtype=dtype)
    before = taken(ctx)
    got = ops.remap(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), 'linear_cv_q5', 'constant', 7.0).get()
    assert taken(ctx) == before + 1
    for f in range(n):
        want = oracle.remap(src[f], mx, my, oracle.LINEAR | oracle.Q5, oracle.CONSTANT, 7.0)
        assert np.array_equal(got[f], want), (np.dtype(dtype).name, f, int(np.abs(got[f].astype(np.int64) - want).max()))
