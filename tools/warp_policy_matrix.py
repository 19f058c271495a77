"""The measurements behind remap_impl.hpp::tile_warp_pays: homography warps on the row-walking
kernels (tile_warp = 0), under the default policy (1) and on the tile kernel wherever the
homography fits (2), over frame sizes, batch sizes and geometries.  ms per launch.

    python3 tools/warp_policy_matrix.py [interp ...]      (default: linear cubic lanczos4)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402
from imgprocessor_amd.utils import getPerspectiveTransform  # noqa: E402
from tools.angle_sweep import rot_persp, timed  # noqa: E402


def quad(h, w):
    q = np.array([(0.05 * w, 0.05 * h), (0.95 * w, 0.025 * h), (0.975 * w, 0.975 * h), (0.025 * w, 0.95 * h)], float)
    r = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    return np.linalg.inv(getPerspectiveTransform(q, r))


def zoom(h, w, s):
    return np.array([[s, 0, (1 - s) * w / 2], [0, s, (1 - s) * h / 2], [0, 0, 1.0]])


def main():
    interps = sys.argv[1:] or ['linear', 'cubic', 'lanczos4']
    ctx = ia.default_context(0)
    rng = np.random.default_rng(1)
    for (h, w, B) in ((2160, 3840, 16), (2160, 3840, 8), (2160, 3840, 4), (2160, 3840, 2), (2160, 3840, 1),
                      (1080, 1920, 16), (1080, 1920, 4), (1080, 1920, 1), (4320, 7680, 4), (480, 640, 8),
                      (2161, 3839, 8)):
        src = ctx.to_device(rng.random((B, h, w), dtype=np.float32))
        dst = ctx.empty((B, h, w), np.float32)
        for name, M in (('quad', quad(h, w)), ('rot0', rot_persp(h, w, 0)), ('rot2', rot_persp(h, w, 2)),
                        ('rot7', rot_persp(h, w, 7)), ('rot30', rot_persp(h, w, 30)), ('zoom0.7', zoom(h, w, 0.7)),
                        ('zoom1.2', zoom(h, w, 1.2)), ('zoom1.4', zoom(h, w, 1.4)),
                        ('r20z1.4', rot_persp(h, w, 20) @ zoom(h, w, 1.4))):
            row = []
            for it in interps:
                r = []
                for tw in (0, 1, 2):
                    ctx.set_tuning(tile_warp=tw)
                    r.append(timed(ctx, lambda: ops.warp_perspective(src, M, (h, w), it, out=dst), n=10, warm=5))
                flag = '' if r[1] <= 1.04 * min(r[0], r[2]) else '  <-- policy'
                row.append('%s %.3f/%.3f/%.3f%s' % (it, r[0], r[1], r[2], flag))
            print('%dx%d x%-2d %-8s ' % (h, w, B, name) + '   '.join(row), flush=True)
        ctx.set_tuning(tile_warp=1)
        del src, dst


if __name__ == '__main__':
    main()
