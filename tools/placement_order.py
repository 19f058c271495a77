"""In which ORDER do the classes of fresh allocations come (review item 2, round 5)?  Blocks of one
size are drawn as they come (placement off), probed with the pool's 3x3 probe over the FIRST
`probe_mib` MiB of each and listed in allocation order with their addresses.

    python tools/placement_order.py size_mib n_blocks [probe_mib]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd import _lib as L  # noqa: E402


def main():
    size = int(float(sys.argv[1]) * (1 << 20))
    n = int(sys.argv[2])
    probe = int(float(sys.argv[3]) * (1 << 20)) if len(sys.argv) > 3 else size
    ctx = ia.default_context(0)
    ctx._place_n = 1
    lib = ctx._lib
    ptrs = []
    for _ in range(n):
        p = C.c_void_p()
        try:
            L.check(lib.ipa_malloc(ctx.handle, size, C.byref(p)), ctx.handle, 'malloc')
        except MemoryError:
            break
        ptrs.append(p)
    t = [ctx._probe_block(p, probe) for p in ptrs]
    t2 = [ctx._probe_block(p, probe) for p in ptrs]
    print('blocks of %d bytes, probe over the first %d: order, address, ms, ms' % (size, probe))
    for i, p in enumerate(ptrs):
        print('  %2d  0x%x  %.4f  %.4f' % (i, p.value, t[i], t2[i]))
    for p in ptrs:
        lib.ipa_free(ctx.handle, p)


if __name__ == '__main__':
    main()
