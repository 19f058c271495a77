"""CPU-only: the C-ABI library loads and exports every symbol the header declares, the
host-side logic (geometry helpers, weight tables, dtype rules, argument checks, sharding)
behaves like the reference, and the product never touches the oracle.
No compute call is made here: there is no GPU in this container.
"""
import ctypes as C
import io
import contextlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from .conftest import ROOT, load_golden, assert_close, load_cv_golden

HEADER = os.path.join(ROOT, 'include', 'imgproc_hip.h')


def _declared_symbols():
    txt = open(HEADER).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(ipa_[a-z0-9_]+)\s*\(', txt)))


def test_abi_exports_every_declared_symbol():
    from imgprocessor_amd import _lib
    lib = _lib.lib()  # raises ImportError if the extension is missing: no fallback
    names = _declared_symbols()
    assert len(names) >= 38
    for n in names:
        assert hasattr(lib, n), 'libimgproc_hip.so does not export %s' % n
    bound = set(_lib.PROTOTYPES) | set(_lib._CHARP)
    assert set(names) == bound, 'ctypes prototypes out of sync with the header: %s' % (
        set(names) ^ bound)
    assert lib.ipa_version() == 100
    assert lib.ipa_status_string(0) == b'ok'
    assert b'unsupported' in lib.ipa_status_string(-2)


def test_no_device_fails_loudly():
    """without a gfx950 device the context refuses to exist: no CPU fallback anywhere"""
    import imgprocessor_amd as ia
    if ia.device_count() > 0:
        pytest.skip('a GPU is visible here')
    h = C.c_void_p()
    from imgprocessor_amd import _lib
    rc = _lib.lib().ipa_ctx_create(0, C.byref(h))
    assert rc == _lib.ERR_NO_DEVICE and not h.value
    with pytest.raises(_lib.ImgProcHipError):
        ia.Context(0)
    with pytest.raises(_lib.ImgProcHipError):
        ia.ops.conv2d(np.zeros((8, 8), np.float32), np.ones((3, 3)))
    # NULL context is a bad argument, never a crash
    assert _lib.lib().ipa_ctx_synchronize(None) == _lib.ERR_BAD_ARG


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'imgprocessor_amd')
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith(('.py', '.hip', '.hpp', '.cpp', '.h')) or fn == 'Makefile':
                txt = open(os.path.join(dp, fn), errors='ignore').read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, re.M), fn
                assert 'liboracle' not in txt and 'oracle.c' not in txt, fn
    out = subprocess.check_output(['ldd', os.path.join(pkg, 'libimgproc_hip.so')]).decode()
    assert 'oracle' not in out and 'torch' not in out


# ---------------------------------------------------------------- geometry --
def test_geometry_helpers(oracle):
    from imgprocessor_amd.utils import (genericCameraMatrix, sortCorners, getPerspectiveTransform,
                                        getOptimalNewCameraMatrix, perspectiveTransform)
    K = genericCameraMatrix((480, 640))
    assert K.dtype == np.float32 and K[0, 2] == 320 and K[1, 2] == 240
    assert_close(K[0, 0], 320 / np.tan(np.deg2rad(30)), 1e-6)
    quad = np.array([(8, 2), (198, 6), (198, 410), (9, 411)], float)  # PerspectiveCorrection.py:866-869
    for perm in ([0, 1, 2, 3], [2, 0, 3, 1], [3, 2, 1, 0]):
        assert np.array_equal(sortCorners(quad[perm]), quad)  # TL, TR, BR, BL
    dst = np.array([[0, 0], [200, 0], [200, 400], [0, 400]], float)
    H = getPerspectiveTransform(quad, dst)
    assert_close(H, oracle.get_perspective_transform(quad, dst), 1e-9, 1e-12)
    assert_close(perspectiveTransform(quad, H), dst, 0, 1e-9)
    # zero distortion: the optimal matrix maps the full frame onto itself
    K = np.array([[500., 0, 319.5], [0, 500., 239.5], [0, 0, 1]])
    nK, roi = getOptimalNewCameraMatrix(K, np.zeros(5), (640, 480), 1)
    assert tuple(roi) == (0, 0, 639, 479)
    assert np.allclose(nK, K, rtol=0, atol=1e-9)  # the 4.x definition is the identity here
    # barrel distortion: alpha=1 keeps every source pixel -> smaller focal length, inner roi
    nK, roi = getOptimalNewCameraMatrix(K, [-0.2, 0.05, 0, 0, 0], (640, 480), 1)
    assert nK[0, 0] < 500 and 0 < roi[2] < 640 and 0 < roi[3] < 480


def test_weight_tables_match_reference(oracle):
    from imgprocessor_amd.interpolate.interpolate2dStructuredIDW import idw_weights
    from imgprocessor_amd.interpolate.interpolate2dStructuredFastIDW import growPositions
    from imgprocessor_amd import ops
    g = load_golden('idw.npz')
    pos, dist = growPositions(4)
    assert np.array_equal(pos, g['grow4_pos']) and np.array_equal(dist, g['grow4_dist'])
    for (k, p, fx, fy) in ((3, 2, 1, 1), (5, 1, 2, 0.5), (15, 3, 1, 1)):
        assert np.array_equal(idw_weights(k, p, fx, fy), oracle.idw_weights(k, p, fx, fy))
    for s in (0.5, 1.0, 1.25, 2.0, 3.7):
        assert_close(ops.gaussian_kernel1d(s), oracle.gaussian_kernel1d(s), 1e-15)
    assert ops.gaussian_kernel1d(0.5).size == 5 and ops.gaussian_kernel1d(1.0).size == 9


def test_var_y_kernel_tables_match_scipy():
    """the vectorised per-row tables of varYSizeGaussianFilter == one gaussian_filter(delta) per
    row, as the reference builds them (filters/varYSizeGaussianFilter.py:40-46)"""
    from scipy.ndimage import gaussian_filter
    from imgprocessor_amd.filters.varYSizeGaussianFilter import _row_kernels
    for (mn, mx, stdx, s0) in ((0, 4, 1, 40), (1, 3, 2, 57), (0, 4, 0, 33), (3, 3, 0, 10),
                               (0, 0.3, 0.2, 9), (2, 9, 3, 64)):
        stdys = np.linspace(mn, mx, s0)
        kx = int(stdx * 2.5)
        kx += 1 - kx % 2
        ky = int(mx * 2.5)
        ky += 1 - ky % 2
        inp = np.zeros((ky, kx))
        inp[ky // 2, kx // 2] = 1
        want = np.stack([gaussian_filter(inp, (sy, stdx)) for sy in stdys])
        assert_close(_row_kernels(stdys, stdx, ky, kx), want, 1e-13, 1e-15)


def test_names_and_argument_checks():
    from imgprocessor_amd import ops, _lib
    from imgprocessor_amd.filters import maskedConvolve, extendArrayForConvolution
    assert ops.interp_id('linear') == 1 and ops.interp_id('lanczos4') == 4
    assert ops.interp_id('linear_cv_q5') == (1 | 0x100) and ops.interp_id('cubic') == 5
    assert ops.border_id('reflect') == 2 and ops.border_id('mirror') == 4
    assert ops.border_id('grid-wrap') == 3 and ops.border_id('nearest') == 1
    with pytest.raises(ValueError):
        ops.interp_id('bogus')
    with pytest.raises(ValueError):
        ops.border_id('bogus')
    a = np.zeros((10, 12))
    m = np.ones((10, 12), bool)
    with pytest.raises(ValueError):  # non-square kernels are out of bounds in the reference
        maskedConvolve(a, np.ones((3, 5)), m)
    with pytest.raises(Exception):   # modey='wrap' raises in the reference as well
        with contextlib.redirect_stdout(io.StringIO()):
            maskedConvolve(a, np.ones((3, 3)), m, mode='wrap')
    with pytest.raises(Exception):
        extendArrayForConvolution(a, (3, 3), modey='wrap')
    assert _lib.dbl([1, 2, 3], 3)[2] == 3.0
    with pytest.raises(ValueError):
        _lib.dbl([1, 2], 3)


def test_dtype_rules():
    from imgprocessor_amd.transformations import toFloatArray, toUIntArray
    assert toFloatArray(np.zeros(3, np.uint8)).dtype == np.float32
    assert toFloatArray(np.zeros(3, np.uint16)).dtype == np.float32
    assert toFloatArray(np.zeros(3, np.uint32)).dtype == np.float64
    assert toFloatArray(np.zeros(3, np.float64)).dtype == np.float64
    a = np.array([-3.7, 0.2, 1.9, 254.6, 300.0])
    u = toUIntArray(a, dtype=np.uint8)
    assert u.dtype == np.uint8 and u.tolist() == [0, 0, 1, 254, 255]  # clip, then truncate


def test_lens_distortion_host_side(tmp_path):
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    ld = LensDistortion()
    with pytest.raises(RuntimeError):
        ld.coeffs
    ld.setCameraParams(800., 810., 319.5, 239.5, -0.1, 0.02, 0.003, 1e-3, -2e-3)
    # argument order k1,k2,k3,p1,p2 vs stored [k1,k2,p1,p2,k3] (LensDistortion.py:360-380)
    assert ld.coeffs['distortionCoeffs'].tolist() == [[-0.1, 0.02, 1e-3, -2e-3, 0.003]]
    assert ld.getCameraParams() == (800., 810., 319.5, 239.5, -0.1, 0.02, 0.003, 1e-3, -2e-3)
    ld.coeffs = LensDistortion.makeCoeffs(ld.coeffs['cameraMatrix'], ld.coeffs['distortionCoeffs'],
                                          (480, 640))
    fn = ld.writeToFile(str(tmp_path / 'cal'))
    assert fn.endswith('.npz')
    ld2 = LensDistortion()
    c = ld2.readFromFile(fn)
    assert np.array_equal(c['cameraMatrix'], ld.coeffs['cameraMatrix']) and c['shape'] == (480, 640)
    with pytest.raises(NotImplementedError):
        ld.calibrate()
    pts = ld.undistortPoints([(319.5, 239.5), (10., 20.)], keepSize=True)
    assert pts.shape == (1, 2, 2)


def test_perspective_correction_host_side():
    from imgprocessor_amd.camera.PerspectiveCorrection import PerspectiveCorrection
    quad = np.array([(8, 2), (198, 6), (198, 410), (9, 411)], float)
    pc = PerspectiveCorrection((420, 210), new_size=(400, 200), border=5)
    pc.setReference(quad)
    H = pc.homography
    got = pc.correctPoints(quad)[0]
    assert_close(got, [[5, 5], [195, 5], [195, 395], [5, 395]], 0, 1e-3)
    assert H.shape == (3, 3)
    with pytest.raises(NotImplementedError):
        PerspectiveCorrection((10, 10), do_correctIntensity=True)
    pc2 = PerspectiveCorrection((10, 10))
    pc2.setReference(quad)
    with pytest.raises(NotImplementedError):  # new_size must be explicit
        pc2.homography


def test_camera_calibration_host_side(tmp_path):
    """container, date lookup and .cal pickle round trip (CameraCalibration.py:60-322, 583-604)"""
    from imgprocessor_amd.camera.CameraCalibration import CameraCalibration, _History
    from imgprocessor_amd.camera.LensDistortion import LensDistortion
    cal = CameraCalibration()
    cal.setCamera('cam0', 12)
    ff_old, ff_new = np.full((6, 8), 0.5), np.full((6, 8), 0.8)
    cal.addFlatField(ff_old, date='01 Nov 15 - 10:00', info='old')
    cal.addFlatField(ff_new, date='30 Nov 15 - 13:20', info='new', light_spectrum='IR')
    cal.addFlatField(ff_new, date='30 Nov 15 - 13:20', info='new')
    assert cal.coeffs['shape'] == (6, 8) and cal.coeffs['light spectra'] == ['visible', 'IR']
    assert cal.dates('flat field', 'visible') == ['30 Nov 15 - 13:20', '01 Nov 15 - 10:00']
    assert cal.infos('flat field', 'visible') == ['new', 'old']
    with pytest.raises(Exception):
        cal.addFlatField(np.ones((8, 6)))
    # newest entry for date=None; for a date, the entry just NEWER than it (l[i-1], :37-49)
    assert cal.getCoeff('flat field', 'visible')[1] == 'new'
    assert cal.getCoeff('flat field', 'visible', '15 Nov 15 - 00:00')[1] == 'new'
    assert cal.getCoeff('flat field', 'visible', '15 Oct 15 - 00:00')[1] == 'old'
    assert cal.getCoeff('flat field', 'UV')[1] == 'new'        # falls back to the first light
    assert cal.getCoeff('psf', 'visible') is None
    assert _History([['x']]).at('not a date') == ['x']
    cal.addDarkCurrent(np.full((6, 8), 3.0), date='02 Nov 15 - 10:00')
    assert np.array_equal(cal.calcDarkCurrent(1.5), np.full((6, 8), 3.0))
    cal.addDarkCurrent(np.ones((6, 8)), np.zeros((6, 8)), date='03 Nov 15 - 10:00')
    assert type(cal.calcDarkCurrent(1.5)) is tuple  # as written at :509 - never evaluated
    ld = LensDistortion()
    ld.setCameraParams(8., 8., 3.5, 2.5, -0.1, 0.0, 0.0, 0.0, 0.0)
    ld.coeffs['shape'] = (6, 8)
    cal.addLens(ld, date='04 Nov 15 - 10:00')
    path = cal.saveToFile(str(tmp_path / 'cam'))
    assert path.endswith('.cal')
    cal2 = CameraCalibration.loadFromFile(str(tmp_path / 'cam'))
    assert cal2.coeffs['name'] == 'cam0' and cal2.coeffs['depth'] == 12
    assert np.array_equal(cal2.getCoeff('flat field', 'IR')[2], ff_new)
    assert np.array_equal(cal2.getCoeff('lens', 'visible')[2]['cameraMatrix'],
                          ld.coeffs['cameraMatrix'])
    cal2.transpose()
    assert cal2.coeffs['shape'] == (8, 6) and cal2.getCoeff('flat field', 'IR')[2].shape == (8, 6)
    cal.addNoise([1., 2., 3.], date='05 Nov 15 - 10:00')
    cal.clearOldCalibrations()  # keeps entry [-1] of the newest-first lists (:289-298)
    assert cal.infos('flat field', 'visible') == ['old']
    with pytest.raises(NotImplementedError):
        cal.correct(np.zeros((6, 8)), deblur=True)
    with pytest.raises(NotImplementedError):
        cal.correct([np.zeros((6, 8)), np.zeros((6, 8))])


# ---------------------------------------------------------------- sharding --
def test_frame_blocks():
    from imgprocessor_amd.sharding import frame_block, all_blocks
    for n in (0, 1, 7, 8, 9, 512, 513):
        for g in (1, 2, 3, 4, 8):
            blocks = all_blocks(n, g)
            covered = [i for (a, b) in blocks for i in range(a, b)]
            assert covered == list(range(n)), (n, g)
            assert max(b - a for a, b in blocks) == -(-n // g)
    assert frame_block(512, 8, 3) == (192, 256)  # C4: 64 frames per GPU
    with pytest.raises(ValueError):
        frame_block(4, 2, 2)


WORKER = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from imgprocessor_amd.sharding import frame_block
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
n = 37
a, b = frame_block(n, world, rank)
mine = torch.zeros(n, dtype=torch.int64)
mine[a:b] = 1
dist.barrier()
# the only cross-rank traffic of the benchmark: barrier + max of the elapsed time
el = torch.tensor([0.25 if rank == 0 else 0.75], dtype=torch.float64)
dist.all_reduce(el, op=dist.ReduceOp.MAX)
dist.all_reduce(mine, op=dist.ReduceOp.SUM)  # test-only: check the partition
if rank == 0:
    print(json.dumps({'max': float(el[0]), 'cover': mine.tolist()}))
dist.destroy_process_group()
'''


def test_row_bands():
    """single-frame split (SURVEY §8e alternative): bands tile the rows, halos are clipped"""
    from imgprocessor_amd.sharding import row_bands
    for h, g, halo in ((4320, 8, 5), (10, 3, 2), (7, 8, 1), (1, 2, 3)):
        bands = row_bands(h, g, halo)
        assert len(bands) == g
        rows = [r for (a, b, lo, hi) in bands for r in range(a, b)]
        assert rows == list(range(h))
        for (a, b, lo, hi) in bands:
            assert 0 <= lo <= a <= b <= hi <= h
            if b > a:
                assert lo == max(0, a - halo) and hi == min(h, b + halo)
    assert row_bands(4320, 8, 5)[1] == (540, 1080, 535, 1085)


def test_two_rank_partition_gloo(tmp_path):
    """world_size-2 run on CPU (gloo): ranks own disjoint contiguous blocks that cover the
    batch, and the timing reduction bench.py uses (barrier + MAX) works"""
    script = tmp_path / 'worker.py'
    script.write_text(WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29617', WORLD_SIZE='2')
    procs = []
    for r in range(2):
        procs.append(subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    import json
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert res['max'] == 0.75
    assert res['cover'] == [1] * 37


BENCH_WORKER = r'''
import os, sys, json, importlib.util
spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(%(root)r, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
barrier, mx, gather = bench.collectives(world, rank)
barrier()
ms = 10.0 + 5.0 * rank          # what a rank's end-to-end leg would report
every = gather(ms)
worst = mx(ms)
plan = bench.e2e_plan(world, 64, 2160, 3840)
if rank == 0:
    print(json.dumps({'every': every, 'worst': worst, 'plan': plan}))
import torch.distributed as dist
dist.destroy_process_group()
'''


def test_bench_end_to_end_plumbing_two_ranks_gloo(tmp_path):
    """the N > 1 half of bench.py that is not a kernel: per-rank times gathered over gloo, MAX over ranks,
    and the end-to-end plan (every rank streams its own 64 uint16 4K frames host -> device -> host: 6 bytes
    per pixel over PCIe) - world_size 2 on CPU; the streaming itself needs the GPU box"""
    import json
    script = tmp_path / 'bworker.py'
    script.write_text(BENCH_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29633', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e.decode()[-2000:]
    res = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert res['every'] == [10.0, 15.0] and res['worst'] == 15.0
    plan = res['plan']
    assert plan['frames_total'] == 128 and plan['pcie_bytes_per_rank'] == 6 * 64 * 2160 * 3840
    assert plan['pcie_bytes'] == 2 * plan['pcie_bytes_per_rank']
    # world 1: plain functions, no process group
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod1', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    b, m, g = bench.collectives(1, 0)
    assert b() is None and m(3.5) == 3.5 and g(2.0) == [2.0]


def test_bench_gpus_n_means_n(tmp_path):
    """`python bench.py --gpus N` without a launcher around it starts the N ranks itself (review
    item 6 of round 4: it used to measure one GPU and print n_gpus 1): the child command is the
    driver's own launch line, and the parent hands its arguments through and ends with the
    children's exit code - checked with a stand-in for the Python interpreter that records how it
    was called (no GPU here: the ranks themselves cannot run)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.self_launch_cmd(4, ['--gpus', '4', '--steps', '7'], 29555)
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1']
    assert cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29555'
    assert cmd[-5:] == [os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '7']
    # the parent: a fake interpreter in place of sys.executable
    fake = tmp_path / 'fakepython'
    log = tmp_path / 'called.txt'
    fake.write_text('#!/bin/sh\necho "$@" > %s\nexit 7\n' % log)
    fake.chmod(0o755)
    code = ('import sys, runpy; sys.executable = %r; sys.argv = [%r, "--gpus", "2", "--steps", "3"]; '
            'runpy.run_path(%r, run_name="__main__")' % (str(fake), os.path.join(ROOT, 'bench.py'),
                                                         os.path.join(ROOT, 'bench.py')))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, timeout=120)
    assert r.returncode == 7, r.stderr.decode()[-2000:]
    called = log.read_text().split()
    assert called[:2] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in called
    assert called[called.index('--nproc-per-node') + 1] == '2' and called[-4:] == ['--gpus', '2', '--steps', '3']


def test_optimal_new_camera_matrix_independent_restatement():
    """utils.getOptimalNewCameraMatrix (what LensDistortion uses for cv2.getOptimalNewCameraMatrix,
    camera/LensDistortion.py:350-353) against the second numpy restatement of OpenCV 4.x's
    definition in tests/golden/gen_golden.py (converged inverse lens model there, 20 fixed-point
    steps here): matrix within 1e-6 relative, roi identical (alpha = 0: within one pixel)"""
    import numpy as np
    from tests.conftest import load_golden
    from imgprocessor_amd.utils import getOptimalNewCameraMatrix
    g, pin = load_cv_golden('cv_modes.npz')
    print(pin)
    for name in ('barrel', 'pincushion', 'c2'):
        v = g['optK_in_' + name]
        K, d, (w, h) = v[:9].reshape(3, 3), v[9:14], (int(v[14]), int(v[15]))
        for alpha in (0, 1):
            M, roi = getOptimalNewCameraMatrix(K, d, (w, h), alpha, (w, h))
            want = g['optK_%s_a%d' % (name, alpha)]
            assert np.allclose(M, want, rtol=1e-6, atol=1e-6), (name, alpha, M, want)
            want_roi = [int(r) for r in g['optroi_%s_a%d' % (name, alpha)]]
            if alpha == 1:
                assert [int(r) for r in roi] == want_roi, (name, roi, want_roi)
            else:
                # alpha = 0 maps the inner rectangle EXACTLY onto [0, W-1]: ceil / floor of a value
                # that is an integer up to the convergence error of the inverse lens model
                assert all(abs(int(a) - b) <= 1 for a, b in zip(roi, want_roi)), (name, roi)


def test_cv2_golden_generator_is_one_command_away():
    """tests/golden/gen_cv2_golden.py: with an OpenCV it writes the cv2-generated vectors the parity tests
    then prefer; without one (this container, the GPU box) it says so and exits 0 without writing, and
    load_cv_golden labels every cv2-specific comparison "cv2-unpinned" """
    script = os.path.join(ROOT, 'tests', 'golden', 'gen_cv2_golden.py')
    have = subprocess.run([sys.executable, '-c', 'import cv2'], capture_output=True).returncode == 0
    g, pin = load_cv_golden('cv_modes.npz')
    if have:
        # (never run the generator from a test: it would change the fixtures the test run is reading)
        assert pin.startswith('cv2-pinned') or pin.startswith('cv2-unpinned')
        return
    before = sorted(os.listdir(os.path.dirname(script)))
    r = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert 'cv2 not available' in r.stdout
    assert sorted(os.listdir(os.path.dirname(script))) == before
    if not os.path.exists(os.path.join(ROOT, 'tests', 'golden', 'cv_modes_cv2.npz')):
        assert pin.startswith('cv2-unpinned')


def test_hand_scheduled_kernels_static_check(tmp_path):
    """every translation unit that carries hand-counted waits - the hand-scheduled loops of
    csrc/wave_pipe.hpp (inline-asm loads with counted vmcnt waits: the dense 3x3 / 5x5 / 7x7 kernels,
    the separable ones) and, since round 5, the Lanczos4 samplers of csrc/tile_warp.hpp (asm-issued
    ds_read_b64 groups released by counted lgkmcnt waits) - compiled here for gfx950 and checked
    statically: no instruction touches a register whose load / LDS read is still in flight, every
    vector-memory asm statement carries the SGPR hazard guard, no vector register is spilled, no
    kernel with counted LDS waits uses scratch memory (tools/check_pipe_asm.py; `make -C
    imgprocessor_amd/csrc check-asm` runs the same)"""
    import shutil
    import subprocess
    from concurrent.futures import ThreadPoolExecutor
    if not shutil.which('hipcc'):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'imgprocessor_amd', 'csrc')
    units = ['fused_k3', 'fused_k5', 'fused_k7', 'fused_sep_a', 'fused_sep_b', 'tile_warp_a', 'tile_warp_b']

    def one(u):
        d = tmp_path / u
        d.mkdir()
        cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-Wno-unused-function',
               '-save-temps=obj', '-c', os.path.join(src, u + '.hip'), '-I', src, '-o', str(d / (u + '.o'))]
        subprocess.run(cmd, check=True, cwd=str(d), capture_output=True)
        asm = d / (u + '-hip-amdgcn-amd-amdhsa-gfx950.s')
        assert asm.exists()
        return subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_pipe_asm.py'), str(asm)],
                              capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as ex:
        results = dict(zip(units, ex.map(one, units)))
    for u, r in results.items():
        assert r.returncode == 0, u + ':\n' + r.stdout + r.stderr
        if u.startswith('tile_warp'):
            # the Lanczos4 instantiations (float32 and uint16 frames) with their counted LDS waits
            assert 'asm LDS reads' in r.stdout and 'tile_warp_kernelILi4Ef' in r.stdout and \
                'tile_warp_kernelILi4Et' in r.stdout, u + ':\n' + r.stdout
            assert ' 0 bytes of scratch per lane' in r.stdout
        else:
            assert 'SampleRowSrc' in r.stdout, u + ':\n' + r.stdout
    assert 'LoadRowSrc' in results['fused_k5'].stdout
