"""GPU: perspective warp + separable filter in ONE launch on the tile skeleton (csrc/tile_chain.hpp) -
PerspectiveCorrection.correct (camera/PerspectiveCorrection.py:401-405) followed by
scipy.ndimage.gaussian_filter (filters/fastFilter.py:42; the chain of
camera/lens/estimateSystematicErrorLensCorrection.py:199-207).  Bit for bit against the two launches
through the workspace (knob tile_chain = 0, the default: tile / gather warp, then the separable filter) over
rotations, perspective, zooms, tap counts, both interpolations, every border mode of the warp and of
the filter, ragged sizes, short pictures, batches that do not divide by the frames of a workgroup and
coordinates outside the source; and against the oracle.
"""
import os
import sys

import numpy as np
import pytest

# an EXPERIMENT's test, not part of tests/: the chain kernel is not in the product library since round 6.  Run it on a
# GPU box against the experiment build (make -C imgprocessor_amd/csrc VARIANT=chain TILE_CHAIN=1):
#   IMGPROC_HIP_LIB=$PWD/imgprocessor_amd/libimgproc_hip_chain.so python -m pytest tools/tile_chain/test_tile_chain.py -m gpu -q
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.conftest import assert_close, oracle  # noqa: E402,F401  (the oracle fixture)
from tests.gpu_helpers import frames, same_bits  # noqa: E402
from tests.test_gpu_tile_warp import rot_persp  # noqa: E402

pytestmark = pytest.mark.gpu

BORDERS = ('constant', 'replicate', 'reflect', 'wrap', 'reflect101')


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def taps(K, seed=3):
    r = np.random.default_rng(seed + K)
    ky, kx = r.random(K) + 0.1, r.random(K) + 0.1
    return ky / ky.sum(), kx / kx.sum()


def both(ia, src, M, shape, interp, K, border='constant', cval=0.25, conv='reflect', expect_chain=True, **knobs):
    """the chain as two launches (tile_chain = 0) and as one (1)"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    d = ctx.to_device(src)
    ky, kx = taps(K)
    out = []
    try:
        for tc in (0, 1):
            ctx.set_tuning(tile_chain=2 * tc, **knobs)   # 2: every chain the kernel covers (1: only the two-launch ones)
            before = ctx.get_tuning('chain_launches')
            out.append(ops.warp_perspective_sepconv2d(d, M, shape, ky, kx, interp, border, cval, conv).get())
            took = ctx.get_tuning('chain_launches') - before
            assert expect_chain is None or took == (1 if tc and expect_chain else 0), 'tile_chain = %d: %d launches of the chain kernel' % (tc, took)
    finally:
        ctx.set_tuning(tile_chain=0, chain_steps=0, chain_frames=0)
    return out


@pytest.mark.parametrize('deg', [0, 3, 17, 45, 90, 133, 180, 271])
@pytest.mark.parametrize('interp', ['linear', 'cubic', 'cubic_cv', 'linear_cv_q5', 'cubic_cv_q5'])
def test_one_launch_has_the_bits_of_the_two(ia, deg, interp):
    h, w, n = 301, 517, 3
    src = frames(n, h, w)
    M = rot_persp(h, w, deg, shift=(40.0, -25.0) if deg == 17 else (0.0, 0.0))
    for K in (9, 5):
        for shape in ((h, w), (h + 13, w - 7)):
            ref, got = both(ia, src, M, shape, interp, K)
            same_bits(got, ref, '%s %d taps %d deg -> %s' % (interp, K, deg, shape))


@pytest.mark.parametrize('K', [3, 5, 7, 9])
@pytest.mark.parametrize('interp', ['linear', 'cubic'])
def test_every_border_mode_of_the_warp_and_of_the_filter(ia, K, interp):
    h, w, n = 150, 203, 2
    src = frames(n, h, w)
    M = rot_persp(h, w, 11.0, zoom=1.15, shift=(12.0, 7.0))   # parts of the output see nothing of the source
    for border in BORDERS:
        for conv in BORDERS:
            ref, got = both(ia, src, M, (h + 5, w + 9), interp, K, border, 0.5, conv)
            same_bits(got, ref, '%s %d taps, warp %s, filter %s' % (interp, K, border, conv))


@pytest.mark.parametrize('shape', [(1, 1), (2, 300), (7, 64), (24, 65), (31, 130), (33, 63), (120, 64),
                                   (121, 128), (129, 190), (250, 7)])
def test_short_and_narrow_pictures(ia, shape):
    h, w, n = 97, 140, 2
    src = frames(n, h, w)
    M = rot_persp(h, w, 6.0)
    for interp, K in (('linear', 9), ('cubic', 9), ('cubic', 3)):
        for conv in ('reflect', 'constant', 'wrap'):
            ref, got = both(ia, src, M, shape, interp, K, 'constant', 0.0, conv,
                            expect_chain=None if min(shape) < 4 else True)
            same_bits(got, ref, '%s %d taps -> %s, filter %s' % (interp, K, shape, conv))


@pytest.mark.parametrize('n', [1, 5, 8, 9, 19])
def test_batches_that_do_not_divide_by_the_frames_of_a_workgroup(ia, n):
    h, w = 130, 200
    src = frames(n, h, w)
    M = rot_persp(h, w, 21.0)
    for fw in (0, 1, 3, 8):
        ref, got = both(ia, src, M, (h, w), 'cubic', 9, chain_frames=fw)
        same_bits(got, ref, '%d frames, %d per workgroup' % (n, fw))


@pytest.mark.parametrize('steps', [1, 2, 3, 4, 7, 16])
def test_any_number_of_steps_per_workgroup(ia, steps):
    h, w, n = 333, 260, 3
    src = frames(n, h, w)
    M = rot_persp(h, w, -8.0)
    for interp, K in (('linear', 7), ('cubic', 9)):
        ref, got = both(ia, src, M, (h, w), interp, K, chain_steps=steps)
        same_bits(got, ref, '%s %d taps, %d steps' % (interp, K, steps))


def test_coordinates_outside_the_source_and_not_finite(ia):
    h, w, n = 120, 160, 2
    src = frames(n, h, w)
    # the plane's horizon crosses the output (the kernel declines: the two launches take it), a warp
    # that looks past the source on every side, a zoom that shrinks the picture fourfold
    cases = [np.array([[1, 0, 0], [0, 1, 0], [1e-2, 0, -0.5]]),
             rot_persp(h, w, 30.0, zoom=2.5),
             rot_persp(h, w, 5.0, zoom=4.0),
             rot_persp(h, w, 0.0, zoom=0.3)]
    for i, M in enumerate(cases):
        for interp in ('linear', 'cubic'):
            for border in ('constant', 'reflect'):
                ref, got = both(ia, src, M, (h, w), interp, 9, border, np.nan if border == 'constant' else 0.0,
                                expect_chain=None if i else False)
                same_bits(got, ref, '%s %s' % (interp, border))


@pytest.mark.parametrize('interp', ['linear', 'cubic'])
def test_one_launch_against_the_oracle(ia, oracle, interp):
    from imgprocessor_amd import ops
    h, w, n = 150, 260, 2
    src = frames(n, h, w)
    oi = {'linear': oracle.LINEAR, 'cubic': oracle.CUBIC_KEYS}[interp]
    g = ops.gaussian_kernel1d(1.0)
    ctx = ia.default_context(0)
    d = ctx.to_device(src)
    for deg, zoom in ((28.0, 1.0), (-61.0, 0.8), (5.0, 1.3)):
        M = rot_persp(h, w, deg, zoom=zoom)
        try:
            ctx.set_tuning(tile_chain=2)
            got = ops.warp_perspective_sepconv2d(d, M, (h, w), g, g, interp, 'constant', 0.5).get()
        finally:
            ctx.set_tuning(tile_chain=0)
        for f in range(n):
            warped = oracle.warp_perspective(src[f], M, (h, w), oi, oracle.CONSTANT, 0.5)
            want = oracle.sepconv2d(warped, g, g, 'reflect')
            assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(), '%s %g deg frame %d' % (interp, deg, f))


def test_a_4k_batch_in_one_launch(ia):
    """8 x 4K bicubic + 9 + 9: the one launch has the bits of the two; by default the library takes
    the two (the one launch measured slower: profiles/r05_micro.txt)"""
    from imgprocessor_amd import ops
    h, w, n = 2160, 3840, 8
    src = frames(1, h, w)
    src = np.concatenate([src + np.float32(0.01 * i) for i in range(n)])
    ctx = ia.default_context(0)
    d = ctx.to_device(src)
    M = rot_persp(h, w, 2.0, persp=(2e-6, 1e-6))
    g = ops.gaussian_kernel1d(1.0)
    assert ctx.get_tuning('tile_chain') == 0
    out = []
    try:
        for tc in (0, 1):
            ctx.set_tuning(tile_chain=tc)
            before = ctx.get_tuning('chain_launches')
            out.append(ops.warp_perspective_sepconv2d(d, M, (h, w), g, g, 'cubic').get())
            assert ctx.get_tuning('chain_launches') - before == tc
    finally:
        ctx.set_tuning(tile_chain=0)
    same_bits(out[1], out[0], '8 x 4K bicubic + 9 + 9')
