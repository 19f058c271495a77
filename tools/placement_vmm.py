"""What about an ALLOCATION makes the strip-shaped kernels faster or slower on it (round 5, review
item 2)?  Source / result batches of the headline composed from physical chunks through HIP's
virtual-memory API (tools/vmm_shim.hip): chunk size x the order in which the chunks are dealt to
the two buffers; each pair probed with the plain 3x3 (the pool's probe, source -> result) and the
headline launch, next to hipMalloc pairs of the same process.  GPU box only.

    hipcc -O2 -shared -fPIC tools/vmm_shim.hip -o tools/libvmm_shim.so
    python tools/placement_vmm.py [repeats]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import _lib as L  # noqa: E402
from imgprocessor_amd import ops  # noqa: E402

B, H, W = 64, 2160, 3840
DENSE = B * H * W * 4


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    shim = C.CDLL(os.path.join(ROOT, 'tools', 'libvmm_shim.so'))
    shim.vmm_alloc_many.argtypes = [C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_uint64, C.c_int,
                                    C.POINTER(C.c_void_p)]
    shim.vmm_free.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    ctx = ia.default_context(0)
    ctx._place_n = 1
    lib = ctx._lib
    K = np.array([[float(W), 0, (W - 1) / 2.0], [0, float(W), (H - 1) / 2.0], [0, 0, 1.0]])
    dist = np.array([-0.12, 0.03, 1e-3, -5e-4, 0.0])
    g = np.exp(-0.5 * np.arange(-2, 3) ** 2)
    g /= g.sum()
    k5 = np.ascontiguousarray(np.outer(g, g), dtype=np.float64)
    k3 = np.full((3, 3), 1.0 / 9)
    dmx, dmy = ops.build_undistort_map(K, dist, K, H, W, ctx=ctx, device=True)

    def fused(sp, dp):
        L.check(lib.ipa_remap_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, W, dmx.ptr, dmy.ptr, W,
            k5.ctypes.data_as(C.POINTER(C.c_double)), 5, 5, dp, L.F32, H, W, W, B, H * W, H * W,
            L.INTER_LINEAR, L.BORDER_CONSTANT, 0.0, L.BORDER_REFLECT, L.BORDER_REFLECT),
            ctx.handle, 'remap_conv2d')

    def conv3(sp, dp):
        L.check(lib.ipa_conv2d_dev(
            ctx.handle, sp, L.F32, H, W, W, k3.ctypes.data_as(C.POINTER(C.c_double)), 3, 3, None, 0,
            dp, W, B, H * W, H * W, L.BORDER_REFLECT, L.BORDER_REFLECT, 0.0), ctx.handle, 'conv2d')

    def copy(sp, dp):
        L.check(lib.ipa_memcpy_d2d(ctx.handle, dp, sp, DENSE), ctx.handle, 'copy')

    def timeit(fn, sp, dp, warm=6, n=20):
        for _ in range(warm):
            fn(sp, dp)
        ctx.synchronize()
        e0, e1 = ctx.event(), ctx.event()
        e0.record()
        for _ in range(n):
            fn(sp, dp)
        e1.record()
        ctx.synchronize()
        return e0.elapsed_ms(e1) / n

    def report(name, sp, dp):
        L.check(lib.ipa_memset(ctx.handle, sp, 0x3c, DENSE), ctx.handle, 'memset')
        L.check(lib.ipa_memset(ctx.handle, dp, 0x3c, DENSE), ctx.handle, 'memset')
        print('%-44s 0x%x 0x%x  conv3 %.4f  fused %.4f  reversed: conv3 %.4f fused %.4f  copy %.4f'
              % (name, sp.value, dp.value, timeit(conv3, sp, dp), timeit(fused, sp, dp),
                 timeit(conv3, dp, sp), timeit(fused, dp, sp), timeit(copy, sp, dp)), flush=True)

    # clocks
    a, b = C.c_void_p(), C.c_void_p()
    L.check(lib.ipa_malloc(ctx.handle, DENSE, C.byref(a)), ctx.handle, 'malloc')
    L.check(lib.ipa_malloc(ctx.handle, DENSE, C.byref(b)), ctx.handle, 'malloc')
    for _ in range(300):
        fused(a, b)
    ctx.synchronize()
    report('hipMalloc pair (first of the process)', a, b)
    held = [a, b]
    orders = {0: 'buffer after buffer', 1: 'chunks dealt round-robin', 2: 'chunks shuffled', 3: 'reverse order'}
    for rep in range(reps):
        for chunk_mib in (0, 1024, 256, 32, 2):
            for order in ((0,) if chunk_mib == 0 else (0, 1, 2, 3)):
                out = (C.c_void_p * 2)()
                rc = shim.vmm_alloc_many(0, DENSE, chunk_mib << 20, order, 12345 + rep, 2, out)
                if rc:
                    print('vmm_alloc_many failed for chunk %d order %d' % (chunk_mib, order))
                    continue
                sp, dp = C.c_void_p(out[0]), C.c_void_p(out[1])
                report('vmm chunk %4d MiB, %s' % (chunk_mib, orders[order]) if chunk_mib else 'vmm one handle per buffer',
                       sp, dp)
                held += [sp, dp]   # kept mapped: a freed chunk would be handed out again
        a, b = C.c_void_p(), C.c_void_p()
        L.check(lib.ipa_malloc(ctx.handle, DENSE, C.byref(a)), ctx.handle, 'malloc')
        L.check(lib.ipa_malloc(ctx.handle, DENSE, C.byref(b)), ctx.handle, 'malloc')
        report('hipMalloc pair', a, b)
        held += [a, b]


if __name__ == '__main__':
    main()
