"""GPU: the round-2 additions to the drop-in surface (SURVEY section 8 a4 / f1 callers and the
advisor's findings): PerspectiveCorrection.distort(quad) and uncorrect's border,
shiftImage, the flat-field rescale step, undistortPoints as written in the reference, the
empty-roi crop, argument validation of device maps - and BASELINE configuration C4 at its full
per-GPU batch (64 frames, EVERY frame compared with the oracle).
"""
import numpy as np
import pytest

from .conftest import assert_close, load_golden, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ia():
    import imgprocessor_amd
    imgprocessor_amd.default_context(0)
    return imgprocessor_amd


@pytest.fixture(scope='module')
def orc(oracle):
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    oracle.set_threads(max(1, min(n, oracle.max_threads(), 32)))
    yield oracle
    oracle.set_threads(1)


def close32(got, want, what='', scale=None):
    want = np.asarray(want, dtype=np.float64)
    s = np.nanmax(np.abs(want)) if scale is None else scale
    assert_close(got, want, 1e-5, 1e-5 * s, what)


def test_perspective_distort_quad_and_uncorrect_border(ia, oracle, capsys):
    """reference: camera/PerspectiveCorrection.py:193-270 (distort), :374-378 (uncorrect)"""
    from imgprocessor_amd.camera.PerspectiveCorrection import PerspectiveCorrection
    from imgprocessor_amd.utils.geometry import sortCorners, getPerspectiveTransform
    g = load_golden('warp_skimage.npz')
    img = g['img']
    quad = np.array([(8, 2), (120, 6), (122, 90), (5, 93)], float)
    pc = PerspectiveCorrection(img.shape, new_size=(96, 128))
    pc.setReference(quad)
    target = np.array([(30, 20), (100, 26), (96, 80), (26, 74)], float)
    out = pc.distort(img, quad=target.copy())
    assert out.shape == img.shape and out.dtype == np.float64
    # the same chain through the oracle
    H = pc_h = getPerspectiveTransform(sortCorners(quad).astype(np.float32),
                                       np.float32([[0, 0], [128, 0], [128, 96], [0, 96]]))
    corr = oracle.warp_perspective(img, np.linalg.inv(H), (96, 128), oracle.LANCZOS4)
    wq = sortCorners(target.copy())
    wq -= wq.min(axis=0)
    objP = np.array([[0, 0], [128, 0], [128, 96], [0, 96]], dtype=np.float32)
    H2 = getPerspectiveTransform(wq.astype(np.float32), objP)
    w = wq[:, 0].max() - wq[:, 0].min()
    h = wq[:, 1].max() - wq[:, 1].min()
    dist = oracle.warp_perspective(corr.astype(np.float32), H2, (int(h), int(w)),
                                   oracle.CUBIC_CV | oracle.Q5)
    bg = np.zeros(img.shape)
    ref = (int(bg.shape[0] / 2 - dist.shape[0] / 2), int(bg.shape[1] / 2 - dist.shape[1] / 2))
    bg[ref[0]:dist.shape[0] + ref[0], ref[1]:dist.shape[1] + ref[1]] = dist
    close32(out, bg, 'distort(quad)', scale=1.0)
    assert np.array_equal(pc.quad, wq + (ref[1], ref[0]))
    assert pc._homography is None and pc_h is not None
    with pytest.raises(NotImplementedError):
        pc.distort(img, rotX=10)
    capsys.readouterr()
    # uncorrect ignores cv2_opts (the reference passes only flags): constant 0 border
    pcr = PerspectiveCorrection(img.shape, new_size=(96, 128),
                                cv2_opts={'borderMode': 1, 'borderValue': 9})
    pcr.setReference(quad)
    warped = pcr.correct(img)
    close32(warped, oracle.warp_perspective(img, np.linalg.inv(pcr.homography), (96, 128),
                                            oracle.LANCZOS4, oracle.REPLICATE), 'cv2_opts border',
            scale=1.0)
    un = pcr.uncorrect(warped)
    close32(un, oracle.warp_perspective(warped, pcr.homography, warped.shape,
                                        oracle.CUBIC_CV | oracle.Q5, oracle.CONSTANT, 0.0),
            'uncorrect border', scale=1.0)


def test_shift_image_and_flatfield_rescale(ia, oracle):
    """simulate/navierStokes.py:52-62 and camera/flatField/vignettingFromDiscreteSteps.py:312-314"""
    from imgprocessor_amd.simulate import shiftImage
    from imgprocessor_amd.camera.flatField import rescaleToGrid
    img = synth((90, 140), 3, np.float64)
    yy, xx = np.mgrid[0:90, 0:140].astype(np.float64)
    u = 2.5 * np.sin(yy / 17.0)
    v = -1.5 * np.cos(xx / 23.0)
    got = shiftImage(u, v, 0.7, img)
    assert got.dtype == np.float32 and got.shape == img.shape
    sx = (xx + u * 0.7).astype(np.float32)
    sy = (yy + v * 0.7).astype(np.float32)
    close32(got, oracle.remap(img.astype(np.float32), sx, sy, oracle.LANCZOS4), 'shiftImage',
            scale=1.0)
    close32(shiftImage(u, v, 0.7, img, 'linear'),
            oracle.remap(img.astype(np.float32), sx, sy, oracle.LINEAR), 'shiftImage linear',
            scale=1.0)
    d = ia.default_context().to_device(img.astype(np.float32))
    close32(shiftImage(u, v, 0.7, d).get(), got, 'device input', scale=1.0)
    # flat-field rescale: coarse grid -> image resolution, Lanczos4 with a reflecting border
    ff = synth((12, 16), 5, np.float64)
    gy, gx = np.mgrid[0:12:0.125, 0:16:0.125]
    gy -= 0.4
    gx -= 0.4
    got = rescaleToGrid(ff, gx, gy)
    assert got.dtype == np.float64 and got.shape == gx.shape
    want = oracle.remap(ff, gx.astype(np.float32), gy.astype(np.float32), oracle.LANCZOS4,
                        oracle.REFLECT)
    assert_close(got, want, 1e-12, 1e-12, 'rescaleToGrid')


def test_lens_distortion_quirks(ia):
    """undistortPoints as written (camera/LensDistortion.py:308-311), empty roi crop (:327-329)"""
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    h, w = 120, 160
    ld = LensDistortion()
    ld.setCameraParams(150., 150., 79.5, 59.5, -0.2, 0.05, 0.0, 1e-3, -1e-3)
    img = synth((h, w), 1)
    ld.correct(img, keepSize=True)
    pts = np.array([(20., 30.), (100., 90.), (60., 60.)])
    full = ld.undistortPoints(pts, keepSize=True)
    crop = ld.undistortPoints(pts, keepSize=False)
    assert full.shape == crop.shape == (1, 3, 2)
    xx, yy = ld.roi[:2]
    shifted = pts.copy()
    shifted[0] -= xx          # both coordinates of point 0 by the x offset
    shifted[1] -= yy          # both coordinates of point 1 by the y offset
    assert_close(crop, ld.undistortPoints(shifted, keepSize=True), 0, 1e-6, 'as written')
    if xx or yy:
        assert not np.allclose(crop, full)
    with pytest.raises(IndexError):
        ld.undistortPoints(pts[:1], keepSize=False)
    # an optimal camera matrix without a valid rectangle -> empty crop like dst[y:y+0, x:x+0]
    ld.roi = (5, 7, 0, 0)
    e = ld.correct(img, keepSize=False)
    assert e.shape == (0, 0) and e.dtype == img.dtype
    rgb = np.stack([img, img], axis=2)
    assert ld.correct(rgb, keepSize=False).shape == (0, 0, 2)
    # device in, device out - the empty crop too
    d = ia.default_context(0).to_device(np.stack([img, img]))
    ed = ld.correct(d, keepSize=False)
    assert isinstance(ed, ia.device.DeviceArray) and ed.shape == (2, 0, 0) and ed.get().size == 0


def test_device_argument_validation(ia):
    from imgprocessor_amd import ops
    ctx = ia.default_context(0)
    img = ctx.to_device(synth((40, 50), 0))
    yy, xx = np.mgrid[0:40, 0:50].astype(np.float32)
    dmx, dmy = ctx.to_device(xx), ctx.to_device(yy)
    k = np.ones((3, 3)) / 9
    with pytest.raises(TypeError):
        ops.remap(img, ctx.to_device(xx.astype(np.float64)), dmy)
    with pytest.raises(ValueError):
        ops.remap(img, dmx, ctx.to_device(yy[:30]))
    with pytest.raises(TypeError):
        ops.remap_conv2d(img, ctx.to_device(xx.astype(np.float64)),
                         ctx.to_device(yy.astype(np.float64)), k)
    with pytest.raises(ValueError):
        ops.remap_sepconv2d(img, dmx, ctx.to_device(yy[:, :20].copy()), [1.0], [1.0])
    with pytest.raises(ValueError):
        ops.conv2d(img, k, mask=ctx.to_device(np.ones((40, 49), np.uint8)))
    with pytest.raises(ValueError):
        ops.conv2d(img, k, mask=ctx.to_device(np.ones((40, 50), np.float32)))
    assert ops.conv2d(img, k, mask=ctx.to_device(np.ones((40, 50), np.uint8))).shape == (40, 50)
    # tuning knobs live in the context, not in the environment
    old = ctx.set_tuning(strip_h=16)
    assert ctx.get_tuning('strip_h') == 16
    ctx.set_tuning(**old)
    with pytest.raises(ValueError):
        ctx.set_tuning(no_such_knob=1)


def test_c4_full_batch_every_frame(ia, orc):
    """BASELINE C4 at its per-GPU batch: 64 x 4K uint16 -> float32, undistort + dense 7x7 in ONE
    launch (72-row strips), every frame against the oracle"""
    h, w, n = 2160, 3840, 64
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    d = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    k7 = np.random.default_rng(123).random((7, 7))
    k7 /= k7.sum()
    base = [np.round(synth((h, w), s, np.float64) * 4095).astype(np.uint16) for s in range(4)]
    frames = np.stack([np.roll(base[i % 4], (17 * (i // 4), 31 * (i // 4)), axis=(0, 1))
                       for i in range(n)])
    ctx = ia.default_context(0)
    dmx, dmy = ia.ops.build_undistort_map(K, d, K, h, w, device=True)
    mx, my = dmx.get(), dmy.get()
    got = ia.ops.remap_conv2d(ctx.to_device(frames), dmx, dmy, k7).get()
    assert got.dtype == np.float32 and got.shape == (n, h, w)
    worst = 0.0
    for i in range(n):
        want = orc.conv2d(orc.remap(frames[i], mx, my, out_dtype=np.float32), k7)
        close32(got[i], want, 'C4 frame %d' % i)
        worst = max(worst, float(np.max(np.abs(got[i] - want) / np.maximum(np.abs(want), 1.0))))
    print('C4 64 frames: max relative error %.3g' % worst)


def test_headline_64x4k_every_frame(ia, orc):
    """the launch bench.py times - 64 x 4K float32, map-based undistort (bilinear, constant 0) +
    5x5 Gaussian ('reflect'), default knobs, ONE launch - every frame against the oracle"""
    import bench
    h, w, n = bench.H4K, bench.W4K, 64
    K, d = bench.camera(h, w)
    k5 = bench.gauss5()
    frames = bench.synth_frames(n, h, w)
    ctx = ia.default_context(0)
    assert ctx.get_tuning('strip_h') == 0 and ctx.get_tuning('frames_wg') == 1
    dmx, dmy = ia.ops.build_undistort_map(K, d, K, h, w, device=True)
    mx, my = dmx.get(), dmy.get()
    got = ia.ops.remap_conv2d(ctx.to_device(frames), dmx, dmy, k5, 'linear', 'constant', 0.0,
                              'reflect').get()
    assert got.dtype == np.float32 and got.shape == (n, h, w)
    worst = 0.0
    for i in range(n):
        want = orc.conv2d(orc.remap(frames[i], mx, my, orc.LINEAR, orc.CONSTANT, 0.0), k5,
                          'reflect')
        close32(got[i], want, 'headline frame %d' % i)
        worst = max(worst, float(np.max(np.abs(got[i] - want)) / np.max(np.abs(want))))
    print('headline 64 x 4K: max |err| / max |ref| = %.3g' % worst)
    assert worst < 1e-5


def test_colour_frames_without_a_host_transpose(ia, capsys):
    """(H, W, C) images as the reference hands them to cv2 (camera/PerspectiveCorrection.py:401-405 -
    its demo warps a colour PNG, :858-900 -, camera/LensDistortion.py:323-326): the channels are
    split into planes and put back ON THE DEVICE (ipa_deinterleave_dev / ipa_interleave_dev);
    the result is bit for bit the per-plane result"""
    from imgprocessor_amd import ops
    from imgprocessor_amd.camera.PerspectiveCorrection import PerspectiveCorrection
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    rng = np.random.default_rng(5)
    h, w = 203, 317
    # layout copies alone, every dtype, odd sizes
    ctx = ia.default_context(0)
    for dt, c in ((np.uint8, 4), (np.uint16, 3), (np.float32, 2), (np.float64, 1)):
        a = (rng.random((h, w, c)) * 200).astype(dt)
        planes = ops.to_planes(a)
        assert planes.shape == (c, h, w)
        assert np.array_equal(planes.get(), np.moveaxis(a, 2, 0))
        assert np.array_equal(ops.from_planes(planes).get(), a)
    # PerspectiveCorrection.correct on a 4-channel uint8 frame (the class default: Lanczos4)
    rgba = (rng.random((h, w, 4)) * 255).astype(np.uint8)
    quad = np.array([(12, 9), (300, 14), (305, 190), (8, 185)], float)
    pc = PerspectiveCorrection((h, w), new_size=(180, 260))
    pc.setReference(quad)
    got = pc.correct(rgba)
    assert got.shape == (180, 260, 4) and got.dtype == np.uint8 and got.flags.c_contiguous
    for k in range(4):
        assert np.array_equal(got[:, :, k], pc.correct(np.ascontiguousarray(rgba[:, :, k])))
    # LensDistortion.correct on a 3-channel float32 frame
    rgb = rng.random((h, w, 3)).astype(np.float32)
    ld = LensDistortion()
    ld.setCameraParams(300., 300., (w - 1) / 2, (h - 1) / 2, -0.1, 0.02, 0.0, 1e-3, -5e-4)
    und = ld.correct(rgb, keepSize=True)
    assert und.shape == rgb.shape and und.dtype == np.float32
    for k in range(3):
        assert np.array_equal(und[:, :, k], ld.correct(np.ascontiguousarray(rgb[:, :, k]), keepSize=True))
    capsys.readouterr()


def test_empty_placed_keeps_the_fastest_candidate(ia):
    """Context.empty_placed: candidates are distinct allocations, the one the probe likes best
    is returned, the others go back to the driver"""
    ctx = ia.default_context(0)
    seen = []

    def probe(a):
        seen.append(a.ptr.value)
        return {0: 3.0, 1: 1.0, 2: 2.0}[len(seen) - 1]
    src = ctx.to_device(np.arange(12, dtype=np.float32).reshape(3, 4))
    a, times = ctx.empty_placed((3, 4), np.float32, probe, candidates=3,
                                fill=lambda c: c.copy_from(src))
    assert times == [3.0, 1.0, 2.0] and len(set(seen)) == 3
    assert a.ptr.value == seen[1] and a.shape == (3, 4)
    assert np.array_equal(a.get(), src.get())


def test_large_blocks_are_placed_by_the_pool():
    """device.py::_alloc_placed (opt-in, IMGPROC_HIP_PLACE=2): a block of 256 MiB and more that the
    pool cannot serve is the better of TWO allocations by a strip-shaped probe - never more than one
    extra block held - and logged; the pool hands the same block out again; smaller blocks and the
    default context (placement off) take the first allocation"""
    import imgprocessor_amd as ia
    ctx = ia.Context(0)
    try:
        assert ctx._place_n == 1 and ctx.placement_log == []      # off by default
        free0 = ctx.mem_info()[0]
        plain = ctx.empty((72, 1024, 1024), np.float32)
        assert ctx.placement_log == []
        del plain
        ctx.trim()
        ctx._place_n = 2
        small = ctx.empty((1024, 1024), np.float32)
        assert ctx.placement_log == []
        big = ctx.empty((72, 1024, 1024), np.float32)          # 288 MiB
        assert len(ctx.placement_log) == 1
        e = ctx.placement_log[0]
        assert e['nbytes'] == big.nbytes and 1 <= len(e['ms']) <= 2 and 0 <= e['kept'] < len(e['ms'])
        assert e['ms'][e['kept']] == min(e['ms']) and all(t > 0 for t in e['ms'])
        # the candidate that was not kept went back to the driver
        assert free0 - ctx.mem_info()[0] < 2 * big.nbytes
        ptr = big.ptr.value
        big.set(np.ones(big.shape, np.float32))                 # the block is usable
        assert float(big.frame(71).get()[5, 5]) == 1.0
        del big
        again = ctx.empty((72, 1024, 1024), np.float32)          # from the pool: the same block, no probe
        assert again.ptr.value == ptr and len(ctx.placement_log) == 1
        # a probe that raises leaves no candidate behind
        free1 = ctx.mem_info()[0]
        real = ctx._probe_block

        def boom(p, n):
            raise RuntimeError('probe failed')
        ctx._probe_block = boom
        with pytest.raises(RuntimeError):
            ctx.empty((80, 1024, 1024), np.float32)
        ctx._probe_block = real
        ctx.synchronize()
        assert abs(free1 - ctx.mem_info()[0]) < (64 << 20)
        ctx._place_n = 1
        other = ctx.empty((80, 1024, 1024), np.float32)
        assert len(ctx.placement_log) == 1
        del small, again, other
    finally:
        ctx.close()
