import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def oracle():
    """the CPU oracle (test infrastructure, oracle/oracle.c)"""
    from oracle import oracle as orc
    orc.build()
    return orc


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def load_cv_golden(name):
    """cv_modes.npz / cv_resize.npz - the numpy restatements of OpenCV's published algorithms - with the
    cv2-GENERATED vectors laid over them when tests/golden/gen_cv2_golden.py has been run on a machine
    that has an OpenCV (files <name>_cv2.npz, same keys).  Returns (dict, label): the label says which
    of the two the comparison below it is pinned to, and the tests print it."""
    g = load_golden(name)
    cvp = os.path.join(GOLDEN, name.replace('.npz', '_cv2.npz'))
    if os.path.exists(cvp):
        c = dict(np.load(cvp))
        ver = str(c.pop('cv2_version', 'unknown'))
        g.update(c)
        return g, 'cv2-pinned (OpenCV %s, %d arrays from %s)' % (ver, len(c), os.path.basename(cvp))
    return g, 'cv2-unpinned (numpy restatement; run tests/golden/gen_cv2_golden.py where cv2 exists)'


def synth(shape, seed, dtype=np.float32):
    """SURVEY §8(d) synthetic image content"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:shape[0], 0:shape[1]].astype(np.float64)
    img = 0.5 + 0.25 * np.sin(2 * np.pi * x / 97) + 0.25 * np.cos(2 * np.pi * y / 61)
    img += 0.05 * rng.standard_normal(shape)
    return np.clip(img, 0, 1).astype(dtype)


def assert_close(a, b, rtol, atol=0.0, what=''):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), '%s: NaN pattern differs' % what
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = (err > tol) & ~nan_a
    if bad.any():
        i = np.unravel_index(np.argmax(np.where(bad, err - tol, -1)), a.shape)
        raise AssertionError('%s: %d/%d elements exceed rtol=%g atol=%g; worst at %s: got %r want %r'
                             % (what, bad.sum(), a.size, rtol, atol, i, a[i], b[i]))
