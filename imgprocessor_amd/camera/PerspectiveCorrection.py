"""Perspective correction (homography warp) on MI355X behind the call surface of
the reference's ``imgProcessor.camera.PerspectiveCorrection.PerspectiveCorrection``
(reference: imgProcessor/camera/PerspectiveCorrection.py — __init__ :39-91,
setReference :97-131, homography :133-191, uncorrect :374-378, correct :380-406,
correctPoints :408-414).

On the hot path: the warp itself (``correct`` = cv2.warpPerspective with
INTER_LANCZOS4, ``uncorrect`` = INTER_CUBIC | WARP_INVERSE_MAP) runs as a HIP
gather kernel that evaluates the homography per pixel in float64 — no map
arrays, 8 B/px of HBM traffic for float32.

Outside the accelerated path (raises NotImplementedError when requested):
pose estimation (solvePnP), tilt-intensity correction, reference-image
homographies from feature matching.  ``new_size=(sy, sx)`` must be given
explicitly; the reference's None-size branch needs the pose code and is
ill-defined.
"""
from __future__ import print_function

import numpy as np

from .. import ops
from ..device import DeviceArray, default_context
from ..utils.geometry import (genericCameraMatrix, sortCorners, getPerspectiveTransform,
                              perspectiveTransform)

_CV_BORDER = {0: 'constant', 1: 'replicate', 2: 'reflect', 3: 'wrap', 4: 'reflect101'}


class PerspectiveCorrection(object):

    def __init__(self, img_shape, cameraMatrix=None, distCoeffs=np.zeros((5, 1)),
                 do_correctIntensity=False, px_per_phys_unit=None, new_size=(None, None),
                 in_plane=False, border=0, maxShear=0.05, material='EL_Si_module', cv2_opts={},
                 interpolation=None, ctx=None):
        if do_correctIntensity:
            raise NotImplementedError('tilt-intensity correction (pose + emissivity model) is '
                                      'outside the accelerated hot path')
        self.opts = {'distCoeffs': np.asarray(distCoeffs).astype(np.float32),
                     'do_correctIntensity': False, 'new_size': new_size, 'in_plane': in_plane,
                     'cv2_opts': dict(cv2_opts), 'border': border, 'material': material,
                     'maxShear': maxShear, 'shape': tuple(img_shape[:2])}
        if cameraMatrix is None:
            cameraMatrix = genericCameraMatrix(img_shape)
        self.opts['cameraMatrix'] = np.asarray(cameraMatrix).astype(np.float32)
        self.refQuad = None
        self.quad = None
        self.px_per_phys_unit = px_per_phys_unit
        self._newBorders = self.opts['new_size']
        self._homography = None
        self._homography_is_fixed = True
        self.interpolation = interpolation  # None = the reference's flags per method
        self._ctx = ctx
        self.img = None

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def setReferenceQuad(self, refQuad):
        self.refQuad = sortCorners(refQuad)

    def setReference(self, ref):
        """ref: 3x3 homography, or the four (x,y) corners of the quad to rectify"""
        self.quad = None
        self._homography = None
        self._homography_is_fixed = True
        if isinstance(ref, np.ndarray) and ref.shape == (3, 3):
            self._homography = ref.astype(np.float64)
        elif len(ref) == 4:
            self.quad = sortCorners(ref)
        else:
            raise NotImplementedError('reference IMAGES need feature matching '
                                      '(ORB + RANSAC), outside the accelerated hot path')

    def _size(self):
        sy, sx = self._newBorders
        if sy is None or sx is None:
            raise NotImplementedError('new_size=(sy, sx) must be given explicitly '
                                      '(the aspect-ratio-from-pose branch is not implemented)')
        return int(sy), int(sx)

    @property
    def homography(self):
        if self._homography is None:
            if self.quad is None:
                raise RuntimeError('call setReference(quad or 3x3 homography) first')
            b = self.opts['border']
            if self.refQuad is not None:
                dst = self.refQuad.astype(np.float32)
            else:
                sy, sx = self._size()
                dst = np.float32([[b, b], [sx - b, b], [sx - b, sy - b], [b, sy - b]])
            self._homography = getPerspectiveTransform(self.quad.astype(np.float32), dst)
        return self._homography

    # ------------------------------------------------------------------
    def _border_kw(self):
        o = self.opts['cv2_opts']
        mode = o.get('borderMode', 0)
        mode = _CV_BORDER.get(mode, mode)
        val = o.get('borderValue', 0)
        if np.ndim(val):
            val = np.ravel(val)[0]
        return mode, float(val)

    def _warp(self, img, M_dst2src, out_shape, interpolation, border=None):
        mode, val = self._border_kw() if border is None else border

        def run(d):
            return ops.warp_perspective(d, M_dst2src, out_shape, interpolation, mode, val)
        if isinstance(img, DeviceArray):
            return run(img)
        img = np.asarray(img)
        if img.ndim == 3:   # (H, W, C): the layout copies both ways run on the device
            return ops.from_planes(run(ops.to_planes(img, ctx=self.ctx))).get()
        return run(self.ctx.to_device(img)).get()

    def correct(self, img):
        """perspective-rectify `img` into new_size=(sy,sx) — :380-406
        (cv2.warpPerspective(img, H, (sx,sy), flags=INTER_LANCZOS4))"""
        print("CORRECT PERSPECTIVE ...")
        self.img = img
        H = self.homography
        sy, sx = self._size()
        Minv = np.linalg.inv(np.asarray(H, dtype=np.float64))
        return self._warp(img, Minv, (sy, sx), self.interpolation or 'lanczos4')

    def uncorrect(self, img):
        """inverse warp back into an image of img's own shape — :374-378
        (flags=INTER_CUBIC | WARP_INVERSE_MAP: H itself maps destination -> source)"""
        s = img.shape[-2:] if isinstance(img, DeviceArray) else np.shape(img)[:2]
        # the reference passes only `flags` here: cv2's default border (constant 0), not cv2_opts
        return self._warp(img, np.asarray(self.homography, dtype=np.float64), tuple(s),
                          self.interpolation or 'cubic_cv_q5', border=('constant', 0.0))

    def distort(self, img, rotX=0, rotY=0, quad=None):
        """apply a perspective distortion: rectify `img` (correct), warp the result into `quad`
        and paste it centred into a zero image of img's shape - :193-270, the warp at :241-242
        (cv2.warpPerspective(corr, H, (w, h), flags=INTER_CUBIC | WARP_INVERSE_MAP)).
        Only the explicit-quad form is on the accelerated path: without `quad` the reference
        projects the stored quad through a pose (rotX / rotY), which needs its 3-D point and
        solvePnP code."""
        if quad is None:
            raise NotImplementedError('distort() without quad needs the pose / 3-D projection '
                                      'code, outside the accelerated hot path')
        if isinstance(img, DeviceArray):
            img = img.get()
        img = np.asarray(img)
        self.img = img
        corr = self.correct(img)
        s = img.shape
        wquad = sortCorners(quad)
        wquad -= wquad.min(axis=0)
        lx, ly = corr.shape[1], corr.shape[0]
        objP = np.array([[0, 0], [lx, 0], [lx, ly], [0, ly]], dtype=np.float32)
        homography = getPerspectiveTransform(wquad.astype(np.float32), objP)
        w = wquad[:, 0].max() - wquad[:, 0].min()
        h = wquad[:, 1].max() - wquad[:, 1].min()
        # WARP_INVERSE_MAP: the matrix itself maps destination -> source
        dist = self._warp(corr, np.asarray(homography, dtype=np.float64), (int(h), int(w)),
                          self.interpolation or 'cubic_cv_q5', border=('constant', 0.0))
        # move the middle of dist to the middle of the image
        bg = np.zeros(shape=s)
        rmn = (bg.shape[0] / 2, bg.shape[1] / 2)
        ss = dist.shape
        mn = (ss[0] / 2, ss[1] / 2)
        ref = (int(rmn[0] - mn[0]), int(rmn[1] - mn[1]))
        bg[ref[0]:ss[0] + ref[0], ref[1]:ss[1] + ref[1]] = dist
        # finally move the quad into position (the reference re-estimates its pose here too;
        # that call is outside the accelerated path and is not made)
        self.quad = wquad
        self.quad += (ref[1], ref[0])
        self.img = bg
        self._homography = None
        return self.img

    def correctPoints(self, pts):
        """cv2.perspectiveTransform(pts, H) — :408-414"""
        pts = np.asarray(pts)
        if pts.ndim == 2:
            pts = pts.reshape(1, *pts.shape)
        return perspectiveTransform(pts.astype(np.float32), self.homography).astype(np.float32)
