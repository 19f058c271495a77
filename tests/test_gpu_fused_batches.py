"""GPU: the fused remap -> K x K kernel on batches, over geometries chosen to exercise every branch
of the strip loops - footprints on the source border, rotations, flips, zooms, constant / reflect /
wrap filter borders, rim strips, ragged sizes, NaN and far-away coordinates, 1/32-px coordinates,
the lens model and the homography as coordinate sources - against the oracle, and the dispatch
orders against each other bit for bit.
"""
import numpy as np
import pytest

from .conftest import assert_close, synth
from .gpu_helpers import frames, kern, radial_maps, rot_maps, same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


def run_fused(ia, src, mx, my, k, **kw):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    return ops.remap_conv2d(d_src, dmx, dmy, k, **kw).get()


@pytest.mark.parametrize('n', [1, 3, 4, 6])
@pytest.mark.parametrize('shape', [(96, 300), (131, 517), (200, 1030)])
def test_fused_batches_radial(ia, oracle, n, shape):
    h, w = shape
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5)
    got = run_fused(ia, src, mx, my, k)
    for f in range(n):
        want = oracle.conv2d(oracle.remap(src[f], mx, my), k)
        assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(), 'frame %d vs oracle' % f)


@pytest.mark.parametrize('case', ['rot3', 'rot20', 'rot90', 'rot180', 'shift_out', 'pincushion',
                                  'flipx', 'zoom_out', 'zoom_in'])
def test_fused_geometries(ia, oracle, case):
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
        got = run_fused(ia, src, mx, my, k, **kw)
        # every border combination against the oracle, two frames
        for f in (0, n - 1):
            want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR,
                                              oracle._mode(kw.get('border_mode', 'constant')), 0.0),
                                 k, kw.get('conv_mode', 'reflect'))
            assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(),
                         '%s %r frame %d vs oracle' % (case, kw, f))


def test_fused_nan_and_far_coordinates(ia):
    h, w, n = 80, 600, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    mx = mx.copy(); my = my.copy()
    mx[10, 50:60] = np.nan
    my[20, 300:310] = np.inf
    mx[30, 400:420] = 3e7
    mx[40:44, :] = -5.0
    got = run_fused(ia, src, mx, my, kern(5), border_value=0.25)
    from oracle import oracle as orc
    orc.build()
    for f in range(n):
        want = orc.conv2d(orc.remap(src[f], mx, my, orc.LINEAR, orc.CONSTANT, 0.25), kern(5))
        assert_close(got[f], want, 1e-5, 1e-5 * np.nanmax(np.abs(want)), 'nan / far coordinates vs oracle')


def test_fused_q5(ia, oracle):
    """cv2's 1/32-px coordinates: the fused chain = the standalone remap followed by the filter"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 120, 520, 4
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5)
    got = run_fused(ia, src, mx, my, k, interpolation='linear_cv_q5')
    d_src, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    two = ops.conv2d(ops.remap(d_src, dmx, dmy, 'linear_cv_q5'), k).get()
    assert_close(got, two, 1e-6, 1e-6 * np.abs(two).max(), 'q5 fused vs two launches')


def test_fused_analytic_sources(ia, oracle):
    """undistort (lens model) and homography coordinate sources against the oracle"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w, n = 140, 900, 6
    src = frames(n, h, w)
    mx, my, K, dist = radial_maps(h, w)
    k = kern(5, 3)
    d_src = ctx.to_device(src)
    M = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
    got_u = ops.undistort_conv2d(d_src, K, dist, K, k).get()
    got_h = ops.warp_perspective_conv2d(d_src, M, (h, w), k).get()
    for f in (0, n - 1):
        want = oracle.conv2d(oracle.remap(src[f], mx, my), k)
        assert_close(got_u[f], want, 1e-5, 1e-5 * np.abs(want).max(), 'lens model frame %d' % f)
        want = oracle.conv2d(oracle.warp_perspective(src[f], M, (h, w)), k)
        assert_close(got_h[f], want, 1e-5, 1e-5 * np.abs(want).max(), 'homography frame %d' % f)


def test_fused_4k_strip_geometry(ia, oracle):
    """4K frames: tall strips, all interior strips on the fast path"""
    h, w, n = 2160, 3840, 5
    base = synth((h, w), 7)
    src = np.stack([np.roll(base, 11 * i, axis=0) for i in range(n)])
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5, 1)
    got = run_fused(ia, src, mx, my, k)
    sub = slice(1000, 1200)
    want = oracle.conv2d(oracle.remap(src[4], mx, my), k)
    assert_close(got[4][sub], want[sub], 1e-5, 1e-5 * np.abs(want).max(), '4K vs oracle')

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


@pytest.mark.parametrize('K', [3, 5, 7])
def test_compiler_scheduled_fallback_has_the_same_bits(ia, K):
    """knob pipe = 0: no strip takes the hand-scheduled loops of csrc/wave_pipe.hpp (inline-asm
    loads, hand-counted vmcnt waits); the compiler-scheduled chunked loops must give the same bits
    for the plain filter, the fused dense chain (float32 and uint16 frames), the separable filter
    and the fused separable chain"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    for (h, w), n in (((200, 1030), 8), ((131, 517), 4), ((300, 780), 3)):
        f32 = frames(n, h, w)
        u16 = frames(n, h, w, np.uint16)
        mx, my, _, _ = radial_maps(h, w)
        d32, d16 = ctx.to_device(f32), ctx.to_device(u16)
        dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
        k = kern(K)
        g = k[K // 2].copy()
        g /= g.sum()
        res = []
        for knob in (1, 0):
            old = ctx.set_tuning(pipe=knob)
            try:
                res.append([ops.conv2d(d32, k).get(), ops.remap_conv2d(d32, dmx, dmy, k).get(),
                            ops.remap_conv2d(d16, dmx, dmy, k).get(), ops.sepconv2d(d32, g, g).get(),
                            ops.remap_sepconv2d(d32, dmx, dmy, g, g).get(),
                            ops.remap_conv2d(d32, dmx, dmy, k, conv_mode='constant').get()])
            finally:
                ctx.set_tuning(**old)
        names = ['conv2d', 'remap_conv2d f32', 'remap_conv2d u16', 'sepconv2d', 'remap_sepconv2d',
                 'remap_conv2d constant border']
        for a, b, nm in zip(res[0], res[1], names):
            same_bits(a, b, 'pipe=1 vs pipe=0: %s K=%d %r n=%d' % (nm, K, (h, w), n))


def test_sharded_runner_two_contexts_on_one_device(ia):
    """sharding.ShardedRunner (one context + host thread per device, contiguous blocks of frames,
    no collective) driven with devices=[0, 0]: the two halves of a batch computed by two contexts
    on the one GPU of the test box equal the single-launch result bit for bit"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.sharding import ShardedRunner, frame_block
    ctx = ia.default_context(0)
    n, h, w = 8, 270, 1000
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    k = kern(5, 11)
    want = ops.remap_conv2d(ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my), k).get()
    out = np.empty_like(want)
    runner = ShardedRunner(devices=[0, 0])
    assert len(runner.contexts) == 2 and runner.contexts[0] is not runner.contexts[1]

    def work(c, start, stop):
        d = ops.remap_conv2d(c.to_device(src[start:stop]), c.to_device(mx), c.to_device(my), k)
        out[start:stop] = d.get()
        return (start, stop)

    blocks = runner.run(n, work)
    assert blocks == [frame_block(n, 2, 0), frame_block(n, 2, 1)] == [(0, 4), (4, 8)]
    same_bits(out, want, 'ShardedRunner(devices=[0, 0])')
    # ragged: 7 frames over 2 contexts, and more contexts than frames
    out7 = np.empty_like(want[:7])

    def work7(c, start, stop):
        out7[start:stop] = ops.remap_conv2d(c.to_device(src[start:stop]), c.to_device(mx),
                                            c.to_device(my), k).get()
        return stop - start

    assert runner.run(7, work7) == [4, 3]
    same_bits(out7, want[:7], 'ShardedRunner, 7 frames')
    assert ShardedRunner(devices=[0, 0, 0]).run(2, lambda c, a, b: (a, b)) == [(0, 1), (1, 2), None]


@pytest.mark.parametrize('n', [4, 8, 7, 12])
def test_homography_coordinates_stored_once_per_batch(ia, n):
    """fused bilinear chains of a batch through a homography: the double coordinates are evaluated
    once per (matrix, geometry) into the plan buffer and the record producers of the shared loop
    read them as a table (csrc/stored_coords.hpp) - the bits of the per-pixel evaluation
    (knob stored_coords = 0), for the dense and the separable chain, a second call on the cached
    table, another matrix in between"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    for (h, w) in ((200, 1030), (333, 517)):
        src = ctx.to_device(frames(n, h, w))
        M1 = np.array([[0.98, 0.03, 4.0], [-0.02, 1.01, 2.5], [1e-5, -2e-5, 1.0]])
        M2 = np.array([[1.05, -0.04, -7.0], [0.03, 0.97, 3.5], [-2e-5, 1e-5, 1.0]])
        k = kern(5, 2)
        g = np.exp(-0.5 * np.arange(-4, 5) ** 2)
        g /= g.sum()
        calls = {
            'dense M1': lambda: ops.warp_perspective_conv2d(src, M1, (h, w), k),
            'sep M1': lambda: ops.warp_perspective_sepconv2d(src, M1, (h, w), g, g),
            'dense M2, smaller result': lambda: ops.warp_perspective_conv2d(src, M2, (h - 11, w - 6), k),
            'sep M2, reflect border': lambda: ops.warp_perspective_sepconv2d(
                src, M2, (h, w), g, g, 'linear', 'reflect'),
        }
        old = ctx.set_tuning(stored_coords=0)
        try:
            ref = {nm: fn().get() for nm, fn in calls.items()}
        finally:
            ctx.set_tuning(**old)
        for rep in range(2):
            for nm, fn in calls.items():
                same_bits(fn().get(), ref[nm], 'stored coordinates: %s n=%d %r call %d' % (nm, n, (h, w), rep))


@pytest.mark.parametrize('dtype', [np.float32, np.uint16])
@pytest.mark.parametrize('n', [1, 3, 4, 8])
def test_footprints_on_the_source_rim_in_every_row(ia, oracle, dtype, n):
    """maps that leave the source along whole edges (what getOptimalNewCameraMatrix(alpha = 1) gives
    every undistorted picture, camera/LensDistortion.py:350-357): constant border - the hand-
    scheduled loops blend those footprints from the taps they hold (wave_pipe.hpp::border_blend),
    uint16 frames send the corner where the packed tap dword starts before the frame through
    sample() -, and the other border modes; against the oracle"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 97, 530
    src = frames(n, h, w, dtype)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    cases = {
        'half a pixel out at the top left': (xx - 0.5, yy - 0.5),
        'out at the bottom right': (xx + 0.6, yy + 0.4),
        'zoomed out 3 %': ((xx - w / 2) * 1.03 + w / 2, (yy - h / 2) * 1.03 + h / 2),
        'far out on the left': (xx - 40.25, yy + 0.1),
    }
    for name, (mx, my) in cases.items():
        mx, my = mx.astype(np.float32), my.astype(np.float32)
        dmx, dmy = ctx.to_device(mx), ctx.to_device(my)
        for K in (3, 5, 7):
            k = kern(K)
            for border, ob in (('constant', oracle.CONSTANT), ('replicate', oracle.REPLICATE)):
                got = ops.remap_conv2d(ctx.to_device(src), dmx, dmy, k, 'linear', border, 0.25).get()
                for f in range(n):
                    want = oracle.conv2d(oracle.remap(src[f], mx, my, oracle.LINEAR, ob, 0.25,
                                                      out_dtype=np.float32), k)
                    assert_close(got[f], want, 1e-5, 1e-5 * np.abs(want).max(),
                                 '%s, %dx%d, %s, frame %d of %d %s' % (name, K, K, border, f, n, np.dtype(dtype).name))


@pytest.mark.parametrize('n', [8, 16, 24, 32])
def test_frame_groups_walked_a_chunk_at_a_time_have_the_same_bits(ia, n):
    """round 5: a fused batch walks its frame groups a quarter at a time (wave_stencil.hpp::wave_grid,
    knob group_chunk; the tile kernel's groups the same way) - an order of the workgroups, nothing
    else: every chunk size gives the bits of all groups together (group_chunk = 0), for the dense
    and the separable filter, map remaps on the tile kernel and homography warps"""
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    h, w = 170, 610
    src = frames(n, h, w)
    mx, my, _, _ = radial_maps(h, w)
    d, dmx, dmy = ctx.to_device(src), ctx.to_device(mx), ctx.to_device(my)
    k5, g = kern(5), ops.gaussian_kernel1d(1.0)
    a = np.deg2rad(11.0)
    M = np.array([[np.cos(a), -np.sin(a), 40.0], [np.sin(a), np.cos(a), -30.0], [2e-5, 1e-5, 1.0]])
    calls = {
        'remap + 5x5': lambda: ops.remap_conv2d(d, dmx, dmy, k5).get(),
        'remap + separable 9+9': lambda: ops.remap_sepconv2d(d, dmx, dmy, g, g).get(),
        'map remap, Lanczos4': lambda: ops.remap(d, dmx, dmy, 'lanczos4').get(),
        'warp, bicubic': lambda: ops.warp_perspective(d, M, (h, w), 'cubic').get(),
    }
    old = ctx.set_tuning(group_chunk=0, tile_warp=2)
    try:
        ref = {name: fn() for name, fn in calls.items()}
        for gc in (-1, 1, 2, 3, 4):
            ctx.set_tuning(group_chunk=gc)
            for name, fn in calls.items():
                same_bits(fn(), ref[name], '%s, %d frames, group_chunk %d' % (name, n, gc))
    finally:
        ctx.set_tuning(**old)
