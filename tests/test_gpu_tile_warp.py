"""GPU: perspective warps on the tile kernel (csrc/tile_warp.hpp: the source box of a 64 x 32 output
tile in LDS) - PerspectiveCorrection's cv2.warpPerspective (camera/PerspectiveCorrection.py:377-378,
401-405) under rotations, where the row-walking kernels pay per cache line.  Against the oracle and,
bit for bit, against the gather kernel over rotations, perspective, zooms, every interpolation and
border mode, ragged sizes, batches that do not divide by the frames of a workgroup, coordinates
outside the source and not finite.
"""
import numpy as np
import pytest

from .conftest import assert_close
from .gpu_helpers import frames, same_bits

pytestmark = pytest.mark.gpu

INTERPS = ('linear', 'cubic', 'cubic_cv', 'linear_cv_q5', 'cubic_cv_q5', 'lanczos4')
BORDERS = ('constant', 'replicate', 'reflect', 'wrap', 'reflect101')


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def rot_persp(h, w, deg, persp=(1e-4, 5e-5), shift=(0.0, 0.0), zoom=1.0):
    a = np.deg2rad(deg)
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    R = np.array([[zoom * np.cos(a), -zoom * np.sin(a), cx - zoom * (np.cos(a) * cx - np.sin(a) * cy) + shift[0]],
                  [zoom * np.sin(a), zoom * np.cos(a), cy - zoom * (np.sin(a) * cx + np.cos(a) * cy) + shift[1]],
                  [0, 0, 1.0]])
    P = np.array([[1, 0, 0], [0, 1, 0], [persp[0], persp[1], 1.0]])
    return P @ R


def both(ia, src, M, shape, interp, border='constant', cval=0.25):
    """the same warp on the gather / ring kernels (tile_warp = 0) and on the tile kernel (2)"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    d = ctx.to_device(src)
    out = []
    try:
        for tw in (0, 2):
            ctx.set_tuning(tile_warp=tw)
            out.append(ops.warp_perspective(d, M, shape, interp, border, border_value=cval).get())
    finally:
        ctx.set_tuning(tile_warp=1)
    return out


@pytest.mark.parametrize('deg', [0, 3, 17, 45, 90, 133, 180, 271])
@pytest.mark.parametrize('interp', INTERPS)
def test_tile_kernel_has_the_gather_kernels_bits(ia, deg, interp):
    h, w, n = 301, 517, 3
    src = frames(n, h, w)
    M = rot_persp(h, w, deg, shift=(40.0, -25.0) if deg == 17 else (0.0, 0.0))
    for border in BORDERS:
        for shape in ((h, w), (h + 13, w - 7)):
            ref, got = both(ia, src, M, shape, interp, border)
            same_bits(got, ref, '%s %s %d deg -> %s' % (interp, border, deg, shape))


@pytest.mark.parametrize('interp', ['linear', 'cubic', 'lanczos4'])
def test_tile_kernel_against_the_oracle(ia, oracle, interp):
    h, w, n = 150, 260, 2
    src = frames(n, h, w)
    oi = {'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS, 'lanczos4': oracle.LANCZOS4}[interp]
    for deg, zoom in ((28.0, 1.0), (-61.0, 0.8), (5.0, 1.3)):
        M = rot_persp(h, w, deg, zoom=zoom)
        _, got = both(ia, src, M, (h, w), interp, 'constant', 0.5)
        for f in range(n):
            want = oracle.warp_perspective(src[f], M, (h, w), oi, oracle.CONSTANT, 0.5)
            assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(), '%s %g deg frame %d' % (interp, deg, f))


@pytest.mark.parametrize('n', [1, 5, 8, 9, 19])
def test_batches_that_do_not_divide_by_the_frames_of_a_workgroup(ia, n):
    h, w = 130, 200
    src = frames(n, h, w)
    M = rot_persp(h, w, 23.0)
    for interp in ('linear', 'cubic', 'lanczos4'):
        ref, got = both(ia, src, M, (h, w), interp)
        same_bits(got, ref, '%s, %d frames' % (interp, n))


def test_sizes_around_the_tile(ia):
    for (h, w) in ((1, 1), (2, 3), (31, 63), (32, 64), (33, 65), (64, 128), (70, 9), (9, 700)):
        src = frames(2, h, w)
        for deg in (0.0, 37.0):
            M = rot_persp(h, w, deg, persp=(1e-3 / max(w, 8), 0.0))
            for interp in ('linear', 'cubic_cv_q5', 'lanczos4'):
                for border in ('constant', 'reflect'):
                    ref, got = both(ia, src, M, (h, w), interp, border)
                    same_bits(got, ref, '%dx%d %s %s %g deg' % (h, w, interp, border, deg))


def test_coordinates_outside_the_source_and_not_finite(ia):
    h, w = 120, 180
    src = frames(2, h, w)
    cases = {
        'far outside': np.array([[1.0, 0, 5000.0], [0, 1.0, -3000.0], [0, 0, 1.0]]),
        'half outside': np.array([[1.0, 0.2, -90.0], [-0.2, 1.0, 60.0], [0, 0, 1.0]]),
        'horizon in the picture': np.array([[1.0, 0, 0], [0, 1.0, 0], [0.02, 0.0, -1.0]]),
        'huge zoom out': np.array([[40.0, 0, 0], [0, 40.0, 0], [0, 0, 1.0]]),
        # boxes of 120-135 columns: around the 128 the kernel's fill covers (found by tools/fuzz_paths.py)
        'zoom out 1.85': np.array([[1.85, 0.03, -70.0], [-0.03, 1.85, -40.0], [0, 0, 1.0]]),
        'zoom out 2': np.array([[1.99, -0.067, -80.0], [0.069, 1.999, -50.0], [0, 0, 1.0]]),
        'degenerate': np.array([[1.0, 0, 0], [0, 1.0, 0], [0, 0, 0.0]]),
    }
    for name, M in cases.items():
        for interp in ('linear', 'cubic', 'lanczos4'):
            for border in ('constant', 'replicate', 'wrap'):
                ref, got = both(ia, src, M, (h, w), interp, border)
                same_bits(got, ref, '%s %s %s' % (name, interp, border))


def test_default_policy_gives_the_same_bits(ia):
    """tile_warp = 1 (where it pays) takes the tile kernel for some of these and not for others"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 240, 400, 8
    src = frames(n, h, w)
    d = ctx.to_device(src)
    for deg in (0.5, 9.0, 60.0):
        M = rot_persp(h, w, deg, persp=(2e-5, 1e-5))
        for interp in ('linear', 'cubic', 'lanczos4'):
            ref, _ = both(ia, src, M, (h, w), interp)
            got = ops.warp_perspective(d, M, (h, w), interp, 'constant', border_value=0.25).get()
            same_bits(got, ref, 'policy %s %g deg' % (interp, deg))


def test_repeated_calls_with_changing_homographies(ia):
    """the host-side box of the last homography is cached in the context: a new matrix, size or
    interpolation must not reuse it"""
    h, w = 160, 230
    src = frames(4, h, w)
    seq = [(rot_persp(h, w, 12.0), (h, w), 'lanczos4'), (rot_persp(h, w, 12.0), (h, w), 'cubic'),
           (rot_persp(h, w, 70.0), (h, w), 'cubic'), (rot_persp(h, w, 70.0), (h - 20, w + 11), 'cubic'),
           (rot_persp(h, w, 12.0), (h, w), 'lanczos4')]
    for M, shape, interp in seq:
        ref, got = both(ia, src, M, shape, interp)
        same_bits(got, ref, 'sequence %s %s' % (interp, shape))


def test_rotated_fused_chains_in_two_launches_have_the_same_bits(ia):
    """warp + filter of a batch the tile kernel takes, under a rotation: tile warp into the
    workspace, then the filter (fused.hip: rotated_warp_in_two_launches) - the bits of the one
    fused kernel"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    n, h, w = 8, 2160, 3840
    rng = np.random.default_rng(3)
    d = ctx.to_device(rng.random((n, h, w), dtype=np.float32))
    M = rot_persp(h, w, 5.0, persp=(2e-6, 1e-6))
    g9 = ops.gaussian_kernel1d(1.0)
    k5 = np.outer(g9[2:7], g9[2:7])
    k5 /= k5.sum()
    out = {}
    try:
        for tw in (0, 1):
            ctx.set_tuning(tile_warp=tw)
            out[tw] = (ops.warp_perspective_sepconv2d(d, M, (h, w), g9, g9).get(),
                       ops.warp_perspective_conv2d(d, M, (h, w), k5).get())
    finally:
        ctx.set_tuning(tile_warp=1)
    same_bits(out[1][0], out[0][0], 'warp + separable 9+9')
    same_bits(out[1][1], out[0][1], 'warp + dense 5x5')


@pytest.mark.parametrize('interp', ['cubic_cv_q5', 'lanczos4'])
def test_uint16_frames_in_opencvs_16u_arithmetic(ia, oracle, interp):
    """the camera's uint16 frames (PerspectiveCorrection.correct on CV_16U images): the tile kernel
    with the box clipped to the source, border footprints tap by tap - the gather kernel's and
    the oracle's values exactly"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    rng = np.random.default_rng(11)
    oi = {'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5, 'lanczos4': oracle.LANCZOS4}[interp]
    for (h, w, n) in ((150, 260, 2), (33, 200, 5), (301, 517, 3)):
        src = rng.integers(0, 65536, (n, h, w)).astype(np.uint16)
        d = ctx.to_device(src)
        for deg, zoom in ((0.0, 1.0), (17.0, 1.0), (-61.0, 0.8), (90.0, 1.2), (133.0, 1.0)):
            M = rot_persp(h, w, deg, zoom=zoom)
            for border, ob in (('constant', oracle.CONSTANT), ('replicate', oracle.REPLICATE),
                               ('reflect', oracle.REFLECT), ('wrap', oracle.WRAP)):
                out = []
                try:
                    for tw in (0, 2):
                        ctx.set_tuning(tile_warp=tw)
                        out.append(ops.warp_perspective(d, M, (h + 5, w - 3), interp, border, border_value=1000).get())
                finally:
                    ctx.set_tuning(tile_warp=1)
                assert np.array_equal(out[0], out[1]), '%s %s %g deg: tile kernel differs from the gather kernel' % (
                    interp, border, deg)
                if (h, w) == (150, 260) and border in ('constant', 'replicate'):
                    want = oracle.warp_perspective(src[0], M, (h + 5, w - 3), oi, ob, 1000)
                    assert np.array_equal(out[1][0], want), '%s %s %g deg vs oracle' % (interp, border, deg)


@pytest.mark.parametrize('odd', [0, 1])
@pytest.mark.parametrize('case', [('linear', np.float32), ('cubic', np.float32), ('lanczos4', np.float32),
                                  ('cubic_cv_q5', np.uint16), ('lanczos4', np.uint16)])
def test_tile_kernel_with_pitches_and_frame_strides(ia, case, odd):
    """the C ABI on regions of larger device buffers (row pitch > width, frame stride > frame):
    the bits of the same warp on contiguous arrays, nothing written outside the region"""
    import ctypes as C
    from imgprocessor_amd import ops
    from imgprocessor_amd import _lib as L
    from imgprocessor_amd.device import dtype_id
    interp, dt = case
    ctx = ia.default_context(0)
    n, h, w, dh, dw = 5, 150, 333, 170, 301
    rng = np.random.default_rng(5)
    src = (rng.integers(0, 65536, (n, h, w)).astype(np.uint16) if dt == np.uint16
           else rng.random((n, h, w), dtype=np.float32))
    M = rot_persp(h, w, 21.0, zoom=0.9)
    fill = dt(7)
    old = ctx.set_tuning(tile_warp=2)
    try:
        want = ops.warp_perspective(ctx.to_device(src), M, (dh, dw), interp, 'reflect').get()
        sp, dp = w + 24 + odd, dw + 8 + 3 * odd
        sbig = np.full((n, h + 5, sp), fill, dt)
        sbig[:, :h, :w] = src
        d_s = ctx.to_device(sbig)
        d_d = ctx.to_device(np.full((n, dh + 3, dp), dt(9), dt))
        Mv = L.dbl(np.ravel(M), 9)
        ctx._check(ctx._lib.ipa_warp_perspective_dev(
            ctx.handle, d_s.ptr, dtype_id(dt), h, w, sp, Mv, d_d.ptr, dtype_id(dt), dh, dw, dp, n,
            (h + 5) * sp, (dh + 3) * dp, ops.interp_id(interp), ops.border_id('reflect'), 0.0), 'warp')
        got = d_d.get()
    finally:
        ctx.set_tuning(**old)
    assert np.array_equal(np.ascontiguousarray(got[:, :dh, :dw]).view(np.uint8), want.view(np.uint8)), \
        'pitched tile warp %s %s' % (interp, np.dtype(dt).name)
    assert (got[:, dh:, :] == dt(9)).all() and (got[:, :, dw:] == dt(9)).all(), 'wrote outside the region'


@pytest.mark.parametrize('case', [('linear', np.float32), ('cubic_cv', np.float32), ('lanczos4', np.float32),
                                  ('cubic_cv_q5', np.uint16), ('lanczos4', np.uint16)])
def test_small_tiles_where_the_picture_shrinks(ia, case):
    """homographies that shrink the picture (or parts of it: the far side of a trapezoid): the source
    box of a 64 x 32 tile exceeds the kernel's 128 columns / 40 KB and the launch goes to 32 x 32 or
    32 x 16 tiles (two output rows per wave); beyond that, back to the gather kernel - same bits"""
    from imgprocessor_amd import ops
    interp, dt = case
    ctx = ia.default_context(0)
    n, h, w = 3, 420, 640
    rng = np.random.default_rng(8)
    src = (rng.integers(0, 65536, (n, h, w)).astype(np.uint16) if dt == np.uint16
           else rng.random((n, h, w), dtype=np.float32))
    d = ctx.to_device(src)
    trapezoid = np.array([[1.0, 0.35, -70.0], [0.02, 1.9, -150.0], [0.0, 0.0022, 1.0]])   # far side 2x denser
    Ms = [rot_persp(h, w, 12.0, zoom=z) for z in (1.5, 2.2, 3.4, 4.6)] + [trapezoid]
    for M in Ms:
        for border in ('constant', 'reflect'):
            out = []
            try:
                for tw in (0, 2):
                    ctx.set_tuning(tile_warp=tw)
                    out.append(ops.warp_perspective(d, M, (h - 20, w + 30), interp, border, border_value=3).get())
            finally:
                ctx.set_tuning(tile_warp=1)
            assert np.array_equal(out[0].view(np.uint8), out[1].view(np.uint8)), \
                '%s %s %s' % (interp, border, np.round(M, 3).tolist())


@pytest.mark.parametrize('interp', INTERPS)
def test_map_remaps_on_the_tile_kernel(ia, oracle, interp):
    """cv2.remap's map pair as the coordinate source (camera/LensDistortion.py:323-326): the box of a
    tile is the span of its own footprints, clamped to a fixed reserve of LDS - smooth lens maps,
    zooms, rotations, a map that jumps about (most of it tap by tap then), NaN / far-away entries;
    the gather kernel's bits, and the oracle"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    rng = np.random.default_rng(2)
    oi = {'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS, 'cubic_cv': oracle.CUBIC_CV,
          'linear_cv_q5': oracle.LINEAR | oracle.Q5, 'cubic_cv_q5': oracle.CUBIC_CV | oracle.Q5,
          'lanczos4': oracle.LANCZOS4}[interp]
    for (h, w, n) in ((301, 517, 3), (33, 200, 5), (150, 520, 1)):
        src = rng.random((n, h, w), dtype=np.float32)
        d = ctx.to_device(src)
        yy, xx = np.mgrid[0:h + 9, 0:w - 5].astype(np.float32)
        r2 = (xx - w / 2) ** 2 + (yy - h / 2) ** 2
        mapsets = {'radial': ((xx - w / 2) * (1 + 1e-6 * r2) + w / 2, (yy - h / 2) * (1 + 1e-6 * r2) + h / 2),
                   'zoom out': ((xx - w / 2) * 1.07 + w / 2, (yy - h / 2) * 1.07 + h / 2),
                   'rotated': (np.cos(.35) * (xx - w / 2) - np.sin(.35) * (yy - h / 2) + w / 2,
                               np.sin(.35) * (xx - w / 2) + np.cos(.35) * (yy - h / 2) + h / 2),
                   'wild': (xx + 30 * np.sin(yy / 7), yy + 25 * np.cos(xx / 5))}
        for mname, (mx, my) in mapsets.items():
            mx, my = mx.astype(np.float32), my.astype(np.float32)
            if mname == 'wild':
                mx[3, 4] = np.nan
                my[5, 6] = 1e9
            dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
            for border, ob in (('constant', oracle.CONSTANT), ('replicate', oracle.REPLICATE),
                               ('reflect', oracle.REFLECT), ('wrap', oracle.WRAP)):
                out = []
                try:
                    for tw in (0, 2):
                        ctx.set_tuning(tile_warp=tw)
                        out.append(ops.remap(d, dmx, dmy, interp, border, 0.25).get())
                finally:
                    ctx.set_tuning(tile_warp=1)
                same_bits(out[1], out[0], '%s %s %s %dx%d' % (mname, interp, border, h, w))
                if (h, w) == (33, 200) and border in ('constant', 'reflect') and mname != 'wild':
                    want = oracle.remap(src[0], mx, my, oi, ob, 0.25)
                    assert_close(out[1][0], want, 1e-5, 1e-5, '%s %s %s vs oracle' % (mname, interp, border))


def test_maps_that_need_more_than_the_reserve(ia):
    """a map pair that shrinks the picture 2.5 times: its tiles' boxes exceed the kernel's LDS
    reserve, the pixels outside go tap by tap and are counted; the default policy hands the pair to
    the gather kernel from the second call on - the same bits on every call, on either kernel"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    n, h, w = 8, 1100, 1930          # 17 Mpx: the default policy takes the tile kernel for Lanczos4
    rng = np.random.default_rng(4)
    d = ctx.to_device(rng.random((n, h, w), dtype=np.float32))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    dmx = ctx.to_device(((xx - w / 2) * 2.5 + w / 2).astype(np.float32))
    dmy = ctx.to_device(((yy - h / 2) * 2.5 + h / 2).astype(np.float32))
    for interp in ('lanczos4', 'cubic'):
        try:
            ctx.set_tuning(tile_warp=0)
            ref = ops.remap(d, dmx, dmy, interp, 'reflect').get()
            ctx.set_tuning(tile_warp=2)
            same_bits(ops.remap(d, dmx, dmy, interp, 'reflect').get(), ref, interp + ' forced')
        finally:
            ctx.set_tuning(tile_warp=1)
        for call in range(3):
            same_bits(ops.remap(d, dmx, dmy, interp, 'reflect').get(), ref, '%s default, call %d' % (interp, call))
            ctx.synchronize()


def test_more_homographies_in_turn_than_the_plan_table_holds(ia):
    """round 5: the host-side boxes of the last FOUR homographies are kept (least recently used
    replaced); six matrices used in turn - every one evicted and rebuilt - keep the gather kernel's
    bits on every call, on the default policy too"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 170, 260
    src = frames(8, h, w)
    d = ctx.to_device(src)
    mats = [rot_persp(h, w, deg, zoom=z) for deg, z in ((3, 1.0), (21, 0.9), (44, 1.1), (77, 1.0), (130, 0.8), (200, 1.2))]
    refs = [both(ia, src, M, (h, w), 'lanczos4')[0] for M in mats]
    for turn in range(3):
        for M, ref in zip(mats, refs):
            same_bits(ops.warp_perspective(d, M, (h, w), 'lanczos4', 'constant', border_value=0.25).get(), ref,
                      'turn %d' % turn)


def test_more_map_pairs_in_turn_than_the_hint_table_holds(ia):
    """round 5: the slow-pixel hints of map remaps on the tile kernel live in a table of four
    (map pair, geometry) entries, each with its own device / page-locked word and the event behind
    its read-back.  Five map pairs in turn - among them one that shrinks the picture 2.5 times,
    whose hint sends it to the gather kernel - give the gather kernel's bits on every call"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    n, h, w = 8, 700, 1210
    rng = np.random.default_rng(9)
    d = ctx.to_device(rng.random((n, h, w), dtype=np.float32))
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    pairs = []
    for k, zoom in enumerate((1.0, 2.5, 0.8, 1.3, 1.05)):
        mx = ((xx - w / 2) * zoom + w / 2 + 3 * np.sin(yy / 37 + k)).astype(np.float32)
        my = ((yy - h / 2) * zoom + h / 2 + 2 * np.cos(xx / 53 + k)).astype(np.float32)
        pairs.append((ctx.to_device(mx), ctx.to_device(my)))
    refs = []
    try:
        ctx.set_tuning(tile_warp=0)
        for dmx, dmy in pairs:
            refs.append(ops.remap(d, dmx, dmy, 'lanczos4', 'reflect').get())
    finally:
        ctx.set_tuning(tile_warp=1)
    for turn in range(4):
        for (dmx, dmy), ref in zip(pairs, refs):
            same_bits(ops.remap(d, dmx, dmy, 'lanczos4', 'reflect').get(), ref, 'turn %d' % turn)
        ctx.synchronize()
