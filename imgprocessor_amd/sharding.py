"""Frame-batch sharding over the GPUs of one node.

The hot path has no cross-frame state except the read-only maps / matrices
(the reference processes one image per call: camera/LensDistortion.py:316-330),
so a batch of N frames splits into contiguous blocks of ceil(N/G) frames per
GPU with NO collective: every rank / device replicates the tiny read-only
state (camera matrix, distortion coefficients, filter kernel; optionally the
maps) and writes its own slice of the output.

Two ways to use it:
  * one process per GPU (torchrun / bench.py): ``frame_block(n, world, rank)``
    tells each rank which frames are its own;
  * one process, several GPUs: ``ShardedRunner`` drives one context per device
    from one host thread each (ctypes releases the GIL during library calls).
"""
import os
import threading

from .device import Context, device_count


def frame_block(n_frames, world_size, rank):
    """[start, stop) of the contiguous block of frames owned by `rank`"""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError('bad world_size/rank %r/%r' % (world_size, rank))
    per = -(-int(n_frames) // world_size)  # ceil
    start = min(rank * per, n_frames)
    return start, min(start + per, n_frames)


def all_blocks(n_frames, world_size):
    return [frame_block(n_frames, world_size, r) for r in range(world_size)]


class ShardedRunner(object):
    """run ``fn(ctx, start, stop)`` once per device, each on its own thread/context"""

    def __init__(self, devices=None):
        if devices is None:
            devices = list(range(device_count()))
        if not devices:
            raise RuntimeError('no gfx950 device visible (there is no CPU fallback)')
        self.devices = list(devices)
        self.contexts = [Context(d) for d in self.devices]

    def run(self, n_frames, fn):
        errors = [None] * len(self.contexts)
        results = [None] * len(self.contexts)

        def work(i):
            try:
                start, stop = frame_block(n_frames, len(self.contexts), i)
                if stop > start:
                    results[i] = fn(self.contexts[i], start, stop)
                self.contexts[i].synchronize()
            except Exception as e:  # re-raised on the caller's thread
                errors[i] = e
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(self.contexts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errors:
            if e is not None:
                raise e
        return results


class FramePipeline(object):
    """Host-to-host streaming of independent frames through ONE device with copies and
    kernels overlapped.

    ``depth`` workers, each with its own context (= its own HIP stream) on the same device
    and its own device buffers, take the frames round-robin: while one worker's kernel runs,
    the others' H2D / D2H copies use the copy engines.  ``fn(ctx, d_in, d_out)`` is the
    device-side work on one frame, written into ``d_out`` (e.g. ``lambda ctx, d, o:
    ops.remap_conv2d(d, mx[ctx], my[ctx], k, out=o)``); per-context read-only state (maps) is
    the caller's to build once per context (``pipe.contexts``).  Device buffers are allocated
    once per worker.

    For PCIe-rate transfers the host arrays should be page-locked: allocate them with
    ``pipe.pinned_empty`` (or ``Context.pinned_empty``); pageable arrays work but pass through
    the driver's staging copies.
    """

    def __init__(self, device=0, depth=3, cpus=None):
        """cpus: optional set of host CPUs the worker threads pin themselves to (e.g. the
        CPUs of the GPU's NUMA node, `numa_cpus_of_device`): the page-locked staging
        buffers are then touched and the copies driven from the socket the GPU hangs on"""
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.contexts = [Context(device) for _ in range(depth)]
        self.cpus = set(int(c) for c in cpus) if cpus else None

    def pinned_empty(self, shape, dtype):
        return self.contexts[0].pinned_empty(shape, dtype)

    def run(self, frames, out, fn):
        """for every frame i: upload frames[i], fn(ctx, d_in, d_out), download into out[i];
        `frames` and `out` are indexable by frame (ndarrays (N, ...) or lists of ndarrays)"""
        n = len(frames)
        if len(out) != n:
            raise ValueError('frames and out must hold the same number of frames')
        errors = [None] * len(self.contexts)

        def work(t):
            ctx = self.contexts[t]
            d_in = d_out = None
            try:
                if self.cpus:
                    os.sched_setaffinity(0, self.cpus)   # the calling thread only (Linux)
                for i in range(t, n, len(self.contexts)):
                    f, o = frames[i], out[i]
                    if d_in is None or d_in.shape != f.shape or d_in.dtype != f.dtype:
                        d_in = ctx.empty(f.shape, f.dtype)
                    if d_out is None or d_out.shape != o.shape or d_out.dtype != o.dtype:
                        d_out = ctx.empty(o.shape, o.dtype)
                    d_in.set(f)
                    fn(ctx, d_in, d_out)
                    d_out.get(o)
            except Exception as e:  # re-raised on the caller's thread
                errors[t] = e
        threads = [threading.Thread(target=work, args=(t,)) for t in range(len(self.contexts))]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        for e in errors:
            if e is not None:
                raise e
        return out


def numa_cpus_of_device(device=0):
    """host CPUs of the NUMA node GPU `device` is attached to: the device's PCI address as HIP
    numbers it in THIS process (so HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES are honoured and
    other render nodes do not shift the count), then /sys/bus/pci/devices/<address>/numa_node;
    None when the platform does not say"""
    import ctypes as C
    from . import _lib as L
    try:
        buf = C.create_string_buffer(64)
        if L.lib().ipa_device_pci_bus_id(int(device), buf, 64) != 0:
            return None
        bdf = buf.value.decode().strip().lower()
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
        if node < 0:
            return None
        spec = open('/sys/devices/system/node/node%d/cpulist' % node).read().strip()
        cpus = set()
        for part in spec.split(','):
            a, _, b = part.partition('-')
            cpus.update(range(int(a), int(b or a) + 1))
        return cpus
    except (OSError, ValueError, IndexError):
        return None


# ----------------------------------------------------------------- one huge frame --
def row_bands(height, world_size, halo=0):
    """Split the OUTPUT rows of one frame into `world_size` contiguous bands (the
    single-8K-frame alternative).  Returns per rank (out_start, out_stop, lo, hi): the band's
    own rows and the rows [lo, hi) it has to compute so that a filter of half-size `halo`
    sees true neighbours at the inner band edges (clipped at the frame, where the filter's
    border mode applies as usual)."""
    bands = []
    for r in range(world_size):
        a, b = frame_block(height, world_size, r)
        bands.append((a, b, max(0, a - halo), min(height, b + halo)))
    return bands


def remap_filter_band(src, mapx, mapy, kernel, band, interpolation='linear',
                      border_mode='constant', border_value=0.0, conv_mode='reflect'):
    """rows [band[0], band[1]) of filter(remap(src)) for ONE frame, computed from the band's
    own map rows plus a halo — what one GPU of a row-band split runs.  `src`, `mapx`, `mapy`
    are DeviceArrays of that GPU (the source frame and the maps are replicated; only the
    output is split, no collective).  Equal to the same rows of the whole-frame result."""
    import numpy as np
    from . import ops
    a, b, lo, hi = band
    if b <= a:
        return None
    mh, mw = mapx.shape
    part = ops.remap(src, mapx, mapy, interpolation, border_mode, border_value,
                     out_dtype=np.float32, map_roi=(0, lo, mw, hi - lo))
    k = np.asarray(kernel, dtype=np.float64)
    hk = k.shape[0] // 2
    if (lo > 0 and a - lo < hk) or (hi < mh and hi - b < hk):
        raise ValueError('the band needs a halo of %d rows for a %d-row kernel' % (hk, k.shape[0]))
    if (lo > 0 or hi < mh) and conv_mode in ('wrap', 'grid-wrap'):
        raise NotImplementedError('a wrapping filter border needs the whole frame')
    full = ops.conv2d(part, k, conv_mode)
    out = part.ctx.empty((b - a, mw), np.float32)
    part.ctx._check(part.ctx._lib.ipa_memcpy_d2d(
        part.ctx.handle, out.ptr, _offset_ptr(full, (a - lo) * mw), out.nbytes), 'memcpy_d2d')
    return out


def _offset_ptr(arr, elems):
    import ctypes as C
    return C.c_void_p(arr.ptr.value + elems * arr.dtype.itemsize)
