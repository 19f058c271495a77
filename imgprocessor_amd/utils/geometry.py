"""Host-side geometry helpers of the hot path: a handful of 3x3 / 4-point
computations per calibration (never per pixel), kept in numpy float64.
"""
import numpy as np


def genericCameraMatrix(shape, angularField=60):
    """generic pinhole matrix for an image shape — same formula as the reference's
    utils/genericCameraMatrix.py:7-28 (centre = int(shape/2), f = cx / tan(FOV/2))"""
    cy = int(shape[0] / 2)
    cx = int(shape[1] / 2)
    f = cx / np.tan(np.deg2rad(angularField / 2.0))
    return np.array([[f, 0, cx], [0, f, cy], [0, 0, 1]], dtype=np.float32)


def sortCorners(corners):
    """order a quadrilateral's corners clockwise on screen (y down) starting top-left,
    i.e. TL, TR, BR, BL — the order utils/sortCorners.py:8-50 produces"""
    c = np.asarray(corners, dtype=np.float64).reshape(4, 2)
    d = c - c.mean(axis=0)
    ang = np.arctan2(d[:, 1], d[:, 0])
    order = np.argsort(ang)
    c, ang = c[order], ang[order]
    start = int(np.abs(ang + 0.75 * np.pi).argmin())
    return np.roll(c, -start, axis=0)


def getPerspectiveTransform(src, dst):
    """H (3x3, h22 = 1) with H·(x,y,1) ~ (u,v,1) for the four pairs — the 8x8
    linear system cv2.getPerspectiveTransform solves (PerspectiveCorrection.py:149-150).
    Points go through float32 first, as the reference passes .astype(np.float32)."""
    s = np.asarray(src, dtype=np.float32).astype(np.float64).reshape(4, 2)
    d = np.asarray(dst, dtype=np.float32).astype(np.float64).reshape(4, 2)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        x, y = s[i]
        u, v = d[i]
        A[i] = (x, y, 1, 0, 0, 0, -x * u, -y * u)
        A[i + 4] = (0, 0, 0, x, y, 1, -x * v, -y * v)
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(A, b), 1.0).reshape(3, 3)


def perspectiveTransform(pts, H):
    """cv2.perspectiveTransform for an (..., 2) point array"""
    p = np.asarray(pts, dtype=np.float64)
    x, y = p[..., 0], p[..., 1]
    w = H[2, 0] * x + H[2, 1] * y + H[2, 2]
    out = np.empty_like(p)
    out[..., 0] = (H[0, 0] * x + H[0, 1] * y + H[0, 2]) / w
    out[..., 1] = (H[1, 0] * x + H[1, 1] * y + H[1, 2]) / w
    return out


def _undistort_points_normalized(pts, K, dist5, iters=20):
    """pixel -> ideal normalised coordinates (iterative inverse of the radial/tangential model)"""
    k1, k2, p1, p2, k3 = [float(v) for v in np.ravel(dist5)[:5]]
    x0 = (pts[:, 0] - K[0, 2]) / K[0, 0]
    y0 = (pts[:, 1] - K[1, 2]) / K[1, 1]
    x, y = x0.copy(), y0.copy()
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
        dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return np.stack([x, y], axis=1)


def _rectangles(K, dist5, newK, size, N=9):
    w, h = size
    gx, gy = np.meshgrid(np.arange(N), np.arange(N))
    # OpenCV 4.x: grid over [0, W-1] x [0, H-1] in double (2.4 / 3.0 sampled [0, W] in float)
    pts = np.stack([gx.ravel() * (w - 1) / (N - 1.0), gy.ravel() * (h - 1) / (N - 1.0)], axis=1)
    p = _undistort_points_normalized(pts, K, dist5)
    if newK is not None:
        p = np.stack([p[:, 0] * newK[0, 0] + newK[0, 2], p[:, 1] * newK[1, 1] + newK[1, 2]], 1)
    p = p.reshape(N, N, 2)
    ox0, ox1 = p[..., 0].min(), p[..., 0].max()
    oy0, oy1 = p[..., 1].min(), p[..., 1].max()
    ix0, ix1 = p[:, 0, 0].max(), p[:, -1, 0].min()
    iy0, iy1 = p[0, :, 1].max(), p[-1, :, 1].min()
    return (ix0, iy0, ix1 - ix0, iy1 - iy0), (ox0, oy0, ox1 - ox0, oy1 - oy0)


def getOptimalNewCameraMatrix(K, dist5, imageSize, alpha, newImgSize=None):
    """cv2.getOptimalNewCameraMatrix(K, d, (w,h), alpha, (w,h)) as OpenCV 4.x
    documents/implements it (9x9 point grid over [0, W-1] x [0, H-1], inner/outer rectangles,
    (W-1) scaling) — LensDistortion.py:350-353.  OpenCV's result changed across 3.x/4.x and
    cv2 is not available to pin it: this is the 4.x definition, checked against a second
    independent restatement (tests/golden/gen_golden.py::optimal_new_camera_matrix_np),
    unpinned against cv2 itself.
    Returns (newK float64 3x3, roi (x, y, w, h))."""
    K = np.asarray(K, dtype=np.float64).reshape(3, 3)
    w, h = imageSize
    nw, nh = newImgSize or imageSize
    inner, outer = _rectangles(K, dist5, None, (w, h))
    fx0, fy0 = (nw - 1) / inner[2], (nh - 1) / inner[3]
    cx0, cy0 = -fx0 * inner[0], -fy0 * inner[1]
    fx1, fy1 = (nw - 1) / outer[2], (nh - 1) / outer[3]
    cx1, cy1 = -fx1 * outer[0], -fy1 * outer[1]
    a = float(alpha)
    M = np.array([[fx0 * (1 - a) + fx1 * a, 0, cx0 * (1 - a) + cx1 * a],
                  [0, fy0 * (1 - a) + fy1 * a, cy0 * (1 - a) + cy1 * a],
                  [0, 0, 1.0]])
    inner, _ = _rectangles(K, dist5, M, (w, h))
    x0, y0 = int(np.ceil(inner[0])), int(np.ceil(inner[1]))
    x1 = x0 + int(np.floor(inner[2]))
    y1 = y0 + int(np.floor(inner[3]))
    x0c, y0c, x1c, y1c = max(x0, 0), max(y0, 0), min(x1, nw), min(y1, nh)
    roi = (x0c, y0c, max(x1c - x0c, 0), max(y1c - y0c, 0))
    return M, roi
