"""Run the 4K fused headline a few times with given context knobs (for rocprofv3 passes).

    python3 tools/run_one.py [--batch 16] [--steps 3] [--analytic] [--case cubic_maps|cubic_h|lanczos_h] knob=value ...
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402


def main():
    args = sys.argv[1:]
    batch, steps, analytic, knobs, case = 16, 3, False, {}, ''
    i = 0
    while i < len(args):
        a = args[i]
        if a == '--batch':
            batch = int(args[i + 1]); i += 1
        elif a == '--steps':
            steps = int(args[i + 1]); i += 1
        elif a == '--case':
            case = args[i + 1]; i += 1
        elif a == '--analytic':
            analytic = True
        elif '=' in a:
            k, v = a.split('=')
            knobs[k] = int(v)
        i += 1
    ctx = ia.default_context(0)
    ctx.set_tuning(**knobs)
    h, w = 2160, 3840
    K = np.array([[float(w), 0, (w - 1) / 2.0], [0, float(w), (h - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.outer(g, g)
    dmx, dmy = ops.build_undistort_map(K, dist, K, h, w, ctx=ctx, device=True)
    src = ctx.to_device(np.random.default_rng(0).random((batch, h, w), dtype=np.float32))
    dst = ctx.empty((batch, h, w), np.float32)
    H = np.array([[1.02, 0.03, -40.0], [-0.02, 0.98, 30.0], [4e-6, -3e-6, 1.0]])
    if case.startswith('rot'):   # rot<deg>_<interp>: the homography of tools/angle_sweep.py
        from angle_sweep import rot_persp
        deg, interp = case[3:].split('_', 1)
        H = rot_persp(h, w, float(deg))
        if interp.endswith('@u16'):   # uint16 frames
            interp = interp[:-4]
            src = ctx.to_device((src.get() * 65535).astype(np.uint16))
            dst = ctx.empty((batch, h, w), np.uint16)
    from imgprocessor_amd.utils import getPerspectiveTransform
    quad = np.array([(192, 108), (3648, 54), (3744, 2106), (96, 2052)], float)
    rect = np.array([(0, 0), (w - 1, 0), (w - 1, h - 1), (0, h - 1)], float)
    Hq = np.linalg.inv(getPerspectiveTransform(quad, rect))
    g9 = ops.gaussian_kernel1d(1.0)
    k7 = np.random.default_rng(123).random((7, 7))
    k7 /= k7.sum()
    u16 = ctx.to_device(np.round(src.get() * 4095).astype(np.uint16)) if case == 'c4' else None
    xsrc = xdst = None
    if case == 'lz16q':
        xsrc = ctx.to_device(np.round(src.get() * 65535).astype(np.uint16)); xdst = ctx.empty((batch, h, w), np.uint16)
    if case == 'lz8q':
        xsrc = ctx.to_device(np.round(src.get() * 255).astype(np.uint8)); xdst = ctx.empty((batch, h, w), np.uint8)
    k11 = np.random.default_rng(321).random((11, 11))
    k11 /= k11.sum()
    ang = np.deg2rad(7.0)
    R7 = np.array([[np.cos(ang), -np.sin(ang), 150.0], [np.sin(ang), np.cos(ang), -100.0], [4e-6, -2e-6, 1.0]])
    for _ in range(steps):
        if case == 'cubic_maps':
            ops.remap(src, dmx, dmy, interpolation='cubic', out=dst)
        elif case == 'cubic_h':
            ops.warp_perspective(src, H, (h, w), interpolation='cubic', out=dst)
        elif case.startswith('rot'):
            ops.warp_perspective(src, H, (h, w), interpolation=interp, out=dst)
        elif case == 'c4':
            ops.remap_conv2d(u16, dmx, dmy, k7, out=dst)
        elif case == 'fused7':
            ops.remap_conv2d(src, dmx, dmy, k7, out=dst)
        elif case == 'conv7':
            ops.conv2d(src, k7, out=dst)
        elif case == 'conv5':
            ops.conv2d(src, k5, out=dst)
        elif case == 'copy':
            dst.copy_from(src)
        elif case in ('lz16q', 'lz8q'):   # PerspectiveCorrection's default on the camera's uint16 / uint8 frames
            ops.warp_perspective(xsrc, Hq, (h, w), interpolation='lanczos4', out=xdst)
        elif case == 'remaplin':          # LensDistortion.correct itself: cv2.remap from the map pair (tile kernel)
            ops.remap(src, dmx, dmy, out=dst)
        elif case == 'remaplz4':
            ops.remap(src, dmx, dmy, interpolation='lanczos4', out=dst)
        elif case == 'c5':                # bicubic warp (rotation + perspective) + dense 11 x 11
            ops.warp_perspective_conv2d(src, R7, (h, w), k11, 'cubic', out=dst)
        elif case == 'conv11':
            ops.conv2d(src, k11, out=dst)
        elif case in ('lz4q', 'cubicq', 'linq'):   # bench.py's C3 homography (quad -> full frame), standalone warp
            ops.warp_perspective(src, Hq, (h, w), interpolation={'lz4q': 'lanczos4', 'cubicq': 'cubic', 'linq': 'linear'}[case], out=dst)
        elif case in ('c3lin', 'c3cubic'):
            ops.warp_perspective_sepconv2d(src, Hq, (h, w), g9, g9, 'linear' if case == 'c3lin' else 'cubic', out=dst)
        elif case == 'lanczos_h':
            ops.warp_perspective(src, H, (h, w), interpolation='lanczos4', out=dst)
        elif analytic:
            ops.undistort_conv2d(src, K, dist, K, k5, out=dst)
        else:
            ops.remap_conv2d(src, dmx, dmy, k5, out=dst)
    ctx.synchronize()


if __name__ == '__main__':
    main()
