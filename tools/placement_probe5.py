"""Is "good" / "bad" a property of REGIONS of memory?  One large slab; every 1 GiB chunk of it is
read (hipMemcpy into one fixed scratch chunk) and written (from the scratch chunk), several times:
per-chunk rates.  GPU box only.   python tools/placement_probe5.py [slab GiB]"""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imgprocessor_amd as ia  # noqa: E402
from imgprocessor_amd.device import DeviceArray  # noqa: E402

ctx = ia.default_context(0)
GB = 1 << 30
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
chunk = GB
slab = ctx.empty((n * GB,), np.uint8)
scratch = ctx.empty((chunk,), np.uint8)


def view(off):
    v = DeviceArray.__new__(DeviceArray)
    v.ctx, v.shape, v.dtype, v.nbytes = ctx, (chunk,), np.dtype(np.uint8), chunk
    v.ptr = C.c_void_p(slab.ptr.value + off)
    v._owner = False
    v._base = slab
    return v


def rate(fn, reps=6):
    fn()
    ctx.synchronize()
    e0, e1 = ctx.event(), ctx.event()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    ctx.synchronize()
    return 2 * chunk * reps / (e0.elapsed_ms(e1) * 1e-3) / 1e12


for _ in range(20):
    scratch.copy_from(view(0))
print('chunk (GiB offset): read TB/s  write TB/s   (hipMemcpy with a fixed 1 GiB scratch chunk, bytes read + written)')
rd, wr = [], []
for i in range(n):
    v = view(i * GB)
    r, w_ = rate(lambda: scratch.copy_from(v)), rate(lambda: v.copy_from(scratch))
    rd.append(r)
    wr.append(w_)
print(' read : ' + ' '.join('%.2f' % x for x in rd))
print(' write: ' + ' '.join('%.2f' % x for x in wr))
rd2 = [rate(lambda: scratch.copy_from(view(i * GB))) for i in range(n)]
print(' read again: ' + ' '.join('%.2f' % x for x in rd2))
