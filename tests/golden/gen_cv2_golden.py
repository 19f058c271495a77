#!/usr/bin/env python3
"""cv2-GENERATED vectors for the modes that are OpenCV's own:
one command away wherever an OpenCV is importable.

    python3 tests/golden/gen_cv2_golden.py          # writes cv_modes_cv2.npz, cv_resize_cv2.npz

`cv2` is an un-vendored, un-pinned dependency of the reference (setup.py:40) and is NOT installed
in the build container (no network): there this script prints "cv2 not available" and exits 0
without writing anything, and the parity tests say "cv2-unpinned".  On a machine that has any
opencv-python it re-runs the reference's own cv2 calls on the inputs the committed fixtures
cv_modes.npz / cv_resize.npz already hold (made by gen_golden.py) and stores the results under the
SAME keys plus the version string; tests/conftest.py::load_cv_golden then overlays them on the
numpy restatements and the tests report "cv2-pinned (OpenCV x.y.z)".  Only arrays are stored.

Reference call sites these vectors pin:
  camera/LensDistortion.py:323-326      cv2.remap(image, mapx, mapy, INTER_LINEAR, borderValue=...)
  camera/LensDistortion.py:350-357      cv2.getOptimalNewCameraMatrix(K, d, (w, h), 1, (w, h))
  camera/PerspectiveCorrection.py:401-405   cv2.warpPerspective(img, H, dsize, flags=INTER_LANCZOS4)
  camera/PerspectiveCorrection.py:377-378   cv2.warpPerspective(..., INTER_CUBIC | WARP_INVERSE_MAP)
  filters/fastMean.py:13-18             cv2.resize INTER_AREA down, INTER_LINEAR up
  filters/fastFilter.py:44-49           cv2.resize INTER_LANCZOS4
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# the homography of the extra warpPerspective vectors (96 x 128 fixtures): the reference demo's
# quad (camera/PerspectiveCorrection.py:866-869) scaled to the fixture size, as a matrix
WARP_H = np.array([[1.03, 0.021, -2.4], [-0.017, 0.985, 1.9], [4.1e-5, -2.3e-5, 1.0]])


def main():
    try:
        import cv2
    except Exception as ex:  # noqa: BLE001 - any import failure means "not here"
        print('cv2 not available (%s: %s): nothing written; the cv2-specific modes stay cv2-unpinned'
              % (type(ex).__name__, ex))
        return 0
    ver = cv2.__version__
    g = dict(np.load(os.path.join(HERE, 'cv_modes.npz')))
    out = {'cv2_version': np.array(ver)}
    img, img8, img16 = g['img'], g['img8'], g['img16']
    flags = {'linear': cv2.INTER_LINEAR, 'cubic': cv2.INTER_CUBIC, 'lanczos4': cv2.INTER_LANCZOS4}

    def remap(a, mx, my, kind, cv=0):
        return cv2.remap(a, mx, my, flags[kind], borderMode=cv2.BORDER_CONSTANT, borderValue=cv)

    for name in ('radial', 'strong'):
        mx, my = g['mapx_' + name], g['mapy_' + name]
        # float32 frames: cv2 rounds every coordinate to 1/32 px
        out['q5lin_%s_c0' % name] = remap(img, mx, my, 'linear', 0.0)
        out['q5lin_%s_c037' % name] = remap(img, mx, my, 'linear', 0.37)
        out['cubic075q5_' + name] = remap(img, mx, my, 'cubic')
        out['lanczos4_' + name] = remap(img, mx, my, 'lanczos4')
        # uint8 frames: the fixed-point paths
        out['u8fix_' + name] = remap(img8, mx, my, 'linear', 0)
        out['u8fix17_' + name] = remap(img8, mx, my, 'linear', 17)
        for kind in ('cubic', 'lanczos4'):
            out['u8tab_%s_%s' % (kind, name)] = remap(img8, mx, my, kind, 0)
            out['u8tab17_%s_%s' % (kind, name)] = remap(img8, mx, my, kind, 17)
        # uint16 frames: float32 table weights
        for kind in ('linear', 'cubic', 'lanczos4'):
            out['u16cv_%s_%s' % (kind, name)] = remap(img16, mx, my, kind, 0)
            out['u16cv1000_%s_%s' % (kind, name)] = remap(img16, mx, my, kind, 1000)
    # warpPerspective as PerspectiveCorrection calls it: correct() with H (cv2 inverts it),
    # uncorrect() / distort() with INTER_CUBIC | WARP_INVERSE_MAP (H used as dst -> src)
    out['warp_H'] = WARP_H
    hh, ww = img.shape
    for tag, a in (('f32', img), ('u8', img8), ('u16', img16)):
        for kind in ('linear', 'cubic', 'lanczos4'):
            out['warp_%s_%s' % (kind, tag)] = cv2.warpPerspective(a, WARP_H, (ww, hh), flags=flags[kind])
        out['warpinv_cubic_' + tag] = cv2.warpPerspective(a, WARP_H, (ww, hh),
                                                          flags=cv2.INTER_CUBIC | cv2.WARP_INVERSE_MAP)
    for key in g:
        if key.startswith('optK_in_'):
            name = key[len('optK_in_'):]
            v = g[key]
            K, d, w, h = v[:9].reshape(3, 3), v[9:14], int(v[14]), int(v[15])
            for alpha in (0, 1):
                M, roi = cv2.getOptimalNewCameraMatrix(K, d, (w, h), alpha, (w, h))
                out['optK_%s_a%d' % (name, alpha)] = np.asarray(M, np.float64)
                out['optroi_%s_a%d' % (name, alpha)] = np.asarray(roi, np.int64)
    np.savez_compressed(os.path.join(HERE, 'cv_modes_cv2.npz'), **out)

    r = dict(np.load(os.path.join(HERE, 'cv_resize.npz')))
    ro = {'cv2_version': np.array(ver)}
    rflags = {'linear': cv2.INTER_LINEAR, 'cubic': cv2.INTER_CUBIC, 'lanczos4': cv2.INTER_LANCZOS4,
              'area': cv2.INTER_AREA}
    for key in r:
        if key.startswith('img_') or key.startswith('aimg_'):
            continue
        kind, tag, size = key.split('_')
        dh, dw = (int(v) for v in size.split('x'))
        src = r[('aimg_' if kind == 'area' else 'img_') + tag]
        ro[key] = cv2.resize(src, (dw, dh), interpolation=rflags[kind])
    np.savez_compressed(os.path.join(HERE, 'cv_resize_cv2.npz'), **ro)
    print('OpenCV %s: wrote cv_modes_cv2.npz (%d arrays), cv_resize_cv2.npz (%d arrays)'
          % (ver, len(out) - 1, len(ro) - 1))
    return 0


if __name__ == '__main__':
    sys.exit(main())
