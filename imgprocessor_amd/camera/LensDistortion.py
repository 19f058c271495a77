"""Lens-distortion correction on MI355X behind the call surface of the
reference's ``imgProcessor.camera.LensDistortion.LensDistortion``
(reference: imgProcessor/camera/LensDistortion.py — correct :316-330,
getUndistortRectifyMap :342-358, distortImage :332-340, get/setCameraParams
:360-380, getDistortRectifyMap :382-389).

What runs where
  * maps: built once per image shape by a HIP kernel (initUndistortRectifyMap
    model, float64 arithmetic -> float32 maps) and cached on the device, like
    the reference caches ``self.mapx/self.mapy`` (:344-345);
  * ``correct``: HIP gather kernel (bilinear, BORDER_CONSTANT) — the hot loop;
  * camera calibration from pattern images (``calibrate``/``addImg`` …) is
    feature detection, outside this package's scope: load coefficients with
    ``setCameraParams`` / ``coeffs`` / ``readFromFile`` instead.

Differences a drop-in user should know (all stated in DESIGN.md):
  * interpolation is the exact-coordinate bilinear of scipy/skimage by
    default; pass ``interpolation='linear_cv_q5'`` for cv2's 1/32-px
    coordinate rounding;
  * ``newCameraMatrix='optimal'`` (default, = the reference's
    getOptimalNewCameraMatrix(alpha=1) call) follows OpenCV 4.x and is
    unpinned; ``'same'`` uses the camera matrix itself; a 3x3 array is taken
    verbatim.
"""
from collections import OrderedDict

import numpy as np

from .. import ops
from ..device import DeviceArray, default_context
from ..utils.geometry import getOptimalNewCameraMatrix, _undistort_points_normalized


class LensDistortion(object):
    ftype = 'npz'

    def __init__(self, coeffs=None, interpolation='linear', newCameraMatrix='optimal', ctx=None):
        self._coeffs = coeffs if coeffs is not None else {}
        self.opts = {}
        self.mapx, self.mapy = None, None
        self._d_mapx = self._d_mapy = None
        self._map_shape = None
        self.roi = None
        self.newCameraMatrix = None
        self.interpolation = interpolation
        self._newK_mode = newCameraMatrix
        self._ctx = ctx
        self.img = None

    # ------------------------------------------------------------------
    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    @property
    def coeffs(self):
        if not self._coeffs:
            raise RuntimeError('no calibration: set coeffs / setCameraParams / readFromFile '
                               '(calibration from images is outside this package)')
        return self._coeffs

    @coeffs.setter
    def coeffs(self, c):
        self._coeffs = c
        self._invalidate()

    def _invalidate(self):
        self.mapx = self.mapy = self._d_mapx = self._d_mapy = None
        self._map_shape = None

    def calibrate(self, *a, **k):
        raise NotImplementedError('pattern detection / cv2.calibrateCamera is outside the '
                                  'accelerated hot path; provide coefficients instead')

    # -- coefficient IO (LensDistortion.py:258-291) -----------------------
    def writeToFile(self, filename, saveOpts=False):
        if not filename.endswith('.%s' % self.ftype):
            filename += '.%s' % self.ftype
        s = {'coeffs': self.coeffs}
        if saveOpts:
            s['opts'] = self.opts
        np.savez(filename, **s)
        return filename

    def readFromFile(self, filename):
        s = dict(np.load(filename, allow_pickle=True))
        try:
            self.coeffs = s['coeffs'][()]
        except KeyError:
            self.coeffs = s  # legacy layout: flat npz
        try:
            self.opts = s['opts'][()]
        except KeyError:
            pass
        return self.coeffs

    # -- parameters (:360-380; note the k1,k2,k3,p1,p2 argument order) ------
    def getCameraParams(self):
        c = self.coeffs['cameraMatrix']
        k1, k2, p1, p2, k3 = tuple(np.ravel(self.coeffs['distortionCoeffs'])[:5].tolist())
        return c[0][0], c[1][1], c[0][2], c[1][2], k1, k2, k3, p1, p2

    def setCameraParams(self, fx, fy, cx, cy, k1, k2, k3, p1, p2):
        c = self._coeffs['cameraMatrix'] = np.zeros(shape=(3, 3))
        c[0, 0], c[1, 1], c[0, 2], c[1, 2], c[2, 2] = fx, fy, cx, cy, 1
        self._coeffs['distortionCoeffs'] = np.array([[k1, k2, p1, p2, k3]])
        self._invalidate()

    def _K_d(self):
        K = np.asarray(self.coeffs['cameraMatrix'], dtype=np.float64).reshape(3, 3)
        d = np.ravel(np.asarray(self.coeffs['distortionCoeffs'], dtype=np.float64))
        if d.size < 5:
            d = np.concatenate([d, np.zeros(5 - d.size)])
        if d.size > 5 and np.any(d[5:] != 0):
            raise NotImplementedError('rational / thin-prism coefficients (k4..) are not supported')
        return K, d[:5]

    # -- maps ---------------------------------------------------------------
    def _new_camera_matrix(self, imgWidth, imgHeight):
        K, d = self._K_d()
        mode = self._newK_mode
        if isinstance(mode, str):
            if mode == 'optimal':
                return getOptimalNewCameraMatrix(K, d, (imgWidth, imgHeight), 1,
                                                 (imgWidth, imgHeight))
            if mode == 'same':
                return K.copy(), (0, 0, imgWidth, imgHeight)
            raise ValueError("newCameraMatrix must be 'optimal', 'same' or a 3x3 array")
        return np.asarray(mode, dtype=np.float64).reshape(3, 3), (0, 0, imgWidth, imgHeight)

    def _device_maps(self, imgWidth, imgHeight):
        if self._d_mapx is None or self._map_shape != (imgHeight, imgWidth):
            K, d = self._K_d()
            self.newCameraMatrix, self.roi = self._new_camera_matrix(imgWidth, imgHeight)
            self._d_mapx, self._d_mapy = ops.build_undistort_map(
                K, d, self.newCameraMatrix, imgHeight, imgWidth, ctx=self.ctx, device=True)
            self._map_shape = (imgHeight, imgWidth)
            self.mapx = self.mapy = None
        return self._d_mapx, self._d_mapy

    def getUndistortRectifyMap(self, imgWidth, imgHeight):
        """(mapx, mapy) float32 host arrays, cached per shape (:342-358)"""
        dx, dy = self._device_maps(imgWidth, imgHeight)
        if self.mapx is None:
            self.mapx, self.mapy = dx.get(), dy.get()
        return self.mapx, self.mapy

    def getDistortRectifyMap(self, sizex, sizey):
        """first-order inverse map pos + (pos - map), as written at :382-389"""
        posy, posx = np.mgrid[0:sizey, 0:sizex].astype(np.float32)
        mapx, mapy = self.getUndistortRectifyMap(sizex, sizey)
        posx += posx - mapx
        posy += posy - mapy
        return posx, posy

    def getShift(self, width, height):
        mapx, mapy = self.getUndistortRectifyMap(width, height)
        posy, posx = np.mgrid[0:height, 0:width].astype(np.float32)
        return ((mapx - posx) ** 2 + (mapy - posy) ** 2) ** 0.5

    # -- the hot path ---------------------------------------------------------
    def _apply(self, image, fn):
        """run fn(device (n,h,w) or (h,w) array) for host (H,W[,C]) images or device arrays"""
        if isinstance(image, DeviceArray):
            return fn(image)
        image = np.asarray(image)
        if image.ndim == 3:  # (H, W, C) like cv2: channels are independent frames - the layout
            # copies both ways run on the device (no transposed copy on the host)
            return ops.from_planes(fn(ops.to_planes(image, ctx=self.ctx))).get()
        if image.ndim != 2:
            raise ValueError('expected a (H,W) or (H,W,C) image')
        return fn(self.ctx.to_device(image)).get()

    def correct(self, image, keepSize=False, borderValue=0):
        """remove lens distortion from `image` (ndarray (H,W[,C]) of uint8/uint16/
        float32/float64, or a DeviceArray (H,W) / batch (N,H,W)) — :316-330"""
        shape = image.shape
        if isinstance(image, DeviceArray):
            h, w = shape[-2:]
        else:
            h, w = np.shape(image)[:2]
        dx, dy = self._device_maps(w, h)
        roi = None if keepSize else self.roi
        if roi is not None and (roi[2] <= 0 or roi[3] <= 0):
            # the optimal camera matrix left no valid rectangle: the reference's crop
            # dst[y:y+h, x:x+w] (:327-329) is then an empty array
            if isinstance(image, DeviceArray):   # device in, device out - also when empty
                lead = tuple(shape[:-2])
                self.img = image.ctx.empty(lead + (max(roi[3], 0), max(roi[2], 0)), image.dtype)
            else:
                a = np.asarray(image)
                self.img = np.empty((max(roi[3], 0), max(roi[2], 0)) + a.shape[2:], a.dtype)
            return self.img

        def run(d):
            return ops.remap(d, dx, dy, self._interp_for(d.dtype), 'constant', borderValue,
                             map_roi=roi)
        self.img = self._apply(image, run)
        return self.img

    def _interp_for(self, dtype):
        """integer camera frames go through cv2.remap's own integer-frame arithmetic, as in the
        reference (:323-326): uint8 in 15-bit fixed point (any 'linear'), uint16 with float32
        table weights at 1/32 px ('linear_cv_q5').  Float frames keep the exact-coordinate
        bilinear of scipy / skimage unless the caller asked for the cv2 variant."""
        if self.interpolation == 'linear' and np.dtype(dtype) == np.uint16:
            return 'linear_cv_q5'
        return self.interpolation

    def distortImage(self, image):
        """opposite of `correct` (approximate inverse map, :332-340)"""
        if isinstance(image, DeviceArray):
            h, w = image.shape[-2:]
        else:
            h, w = np.shape(image)[:2]
        mx, my = self.getDistortRectifyMap(w, h)
        dmx, dmy = self.ctx.to_device(mx), self.ctx.to_device(my)
        return self._apply(image, lambda d: ops.remap(d, dmx, dmy, self._interp_for(d.dtype),
                                                      'constant', 0))

    def undistortPoints(self, points, keepSize=False):
        """ideal pixel positions of distorted `points` [(x,y), ...] (:293-314)"""
        K, d = self._K_d()
        h, w = self.img.shape[-2:] if self.img is not None else self.coeffs['shape'][:2]
        newK, roi = self._new_camera_matrix(w, h)
        pts = np.asarray(points, dtype=np.float32)
        if pts.ndim == 2:
            pts = np.expand_dims(pts, axis=0)
        pts = pts.copy()
        if not keepSize:
            # as written in the reference (:308-311): on the (1, N, 2) array this shifts BOTH
            # coordinates of point 0 by the roi's x offset and both of point 1 by its y offset
            # (and raises IndexError for a single point) - reproduced, not corrected
            xx, yy = roi[:2]
            pts[0, 0] -= xx
            pts[0, 1] -= yy
        pts = pts.reshape(-1, 2).astype(np.float64)
        n = _undistort_points_normalized(pts, K, d)
        return np.stack([n[:, 0] * newK[0, 0] + newK[0, 2],
                         n[:, 1] * newK[1, 1] + newK[1, 2]], axis=1)[None].astype(np.float32)

    def getCoeffStr(self):
        txt = ''
        for key, val in self.coeffs.items():
            txt += '%s = %s\n' % (key, val)
        return txt

    @staticmethod
    def makeCoeffs(cameraMatrix, distortionCoeffs, shape, reprojectionError=0.0,
                   apertureSize=None):
        """the coeffs dict layout calibrate() stores (:232-240)"""
        return OrderedDict([('reprojectionError', reprojectionError),
                            ('apertureSize', apertureSize),
                            ('cameraMatrix', np.asarray(cameraMatrix, dtype=np.float64)),
                            ('distortionCoeffs', np.asarray(distortionCoeffs,
                                                            dtype=np.float64).reshape(1, -1)),
                            ('shape', tuple(shape))])
