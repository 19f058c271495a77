"""Device context and device-resident arrays on top of the C ABI.

``Context`` wraps one ``ipa_ctx`` (one per GPU: a HIP stream + staging
workspace).  ``DeviceArray`` is a thin owner of a device allocation with a
numpy-like shape/dtype, so frames can stay in HBM between calls
(`lens.correct(d_img)` -> `filter(d_img, k)` never touches the host).
"""
import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _lib as L

_DT = {np.dtype(np.uint8): L.U8, np.dtype(np.uint16): L.U16,
       np.dtype(np.float32): L.F32, np.dtype(np.float64): L.F64}


def dtype_id(dt):
    try:
        return _DT[np.dtype(dt)]
    except KeyError:
        raise TypeError('imgprocessor_amd: unsupported dtype %s (uint8, uint16, float32, '
                        'float64 are)' % np.dtype(dt))


class Context(object):
    """one per GPU; create with Context(device_id) or use default_context()"""

    def __init__(self, device_id=0):
        self._lib = L.lib()
        h = C.c_void_p()
        L.check(self._lib.ipa_ctx_create(int(device_id), C.byref(h)), None, 'ipa_ctx_create')
        self.handle = h
        self.device_id = int(device_id)
        # freed device blocks are kept for reuse (exact size match): hipMalloc / hipFree cost
        # 100s of microseconds and synchronise the device, an output array per call would
        # otherwise dominate every small op.  Reuse is stream-ordered like the kernels.
        self._pool = {}
        self._pool_bytes = 0
        self._pool_limit = int(os.environ.get('IMGPROC_HIP_POOL_MB', '8192')) << 20
        self._pool_lock = threading.Lock()
        # Placement of large blocks: OPT-IN since round 5 (IMGPROC_HIP_PLACE=2; default 1 = off).
        # WHERE in the device memory a multi-GB batch buffer lies moves the strip-shaped streaming
        # kernels of this library by up to 10 % on MI355X (64 x 4K undistort + 5x5: 0.95 - 1.09 ms).
        # Round 5 (profiles/r05_micro.txt) narrowed it down to a property of physical REGIONS of the
        # HBM, 16 - 32 GB each - fresh allocations of one process fall into their classes in runs of
        # 8 - 16 in allocation order; padding the frame stride / the pitch, the distance between the
        # source and the result, and the way the buffer is put together from physical chunks
        # (hipMemCreate / hipMemMap) do not move a block out of its class; a linear copy does not see
        # the classes at all.  A caller cannot choose the region; what it can do is take the better
        # of TWO allocations by the probe below (`_probe_block`: a plain 3x3 filter from one half of
        # the block into the other, ~3 ms per 2 GiB) - with the hard limits that no more than one
        # extra block of the requested size is ever held (2 x the requested bytes in all) and that
        # the extra block fits into a quarter of the device's free memory.  Rounds 3 - 4 drew up to
        # 24 candidates (64 GiB held at once) by default: reliable it was not (the runs are longer
        # than that), and a library that allocates 24 x what it was asked for is no design.
        self._place_n = min(2, max(1, int(os.environ.get('IMGPROC_HIP_PLACE', '1'))))
        self._place_min = 256 << 20
        self._place_max = 16 << 30
        self._place_tls = threading.local()   # .active: this thread is choosing a block right now
        self._placed_ptrs = set()             # blocks the probe chose
        self._no_place_sizes = set()          # sizes whose placed block went back to the driver: a caller
                                              # that cycles through such blocks does not pay the probe again
        self.placement_log = []   # one entry per placed block: nbytes, probe ms per candidate, kept

    # -- info -------------------------------------------------------------
    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int()
        mem = C.c_size_t()
        self._check(self._lib.ipa_ctx_device_info(self.handle, name, 256, C.byref(cu),
                                                  C.byref(mem)), 'device_info')
        return {'name': name.value.decode(), 'cu_count': cu.value, 'total_mem': mem.value}

    def _check(self, status, what=''):
        L.check(status, self.handle, what)

    def synchronize(self):
        self._check(self._lib.ipa_ctx_synchronize(self.handle), 'synchronize')

    # -- launch-shape knobs ----------------------------
    def set_tuning(self, **knobs):
        """e.g. ctx.set_tuning(strip_h=48, frames_wg=0); returns the previous values"""
        old = {k: self.get_tuning(k) for k in knobs}
        for k, v in knobs.items():
            self._check(self._lib.ipa_ctx_set_tuning(self.handle, k.encode(), int(v)),
                        'set_tuning')
        return old

    def get_tuning(self, name):
        import ctypes
        v = ctypes.c_int(0)
        self._check(self._lib.ipa_ctx_get_tuning(self.handle, name.encode(), ctypes.byref(v)),
                    'get_tuning')
        return v.value

    # -- memory -----------------------------------------------------------
    def _alloc(self, nbytes):
        with self._pool_lock:
            blocks = self._pool.get(nbytes)
            if blocks:
                self._pool_bytes -= nbytes
                return blocks.pop()
        if (self._place_n > 1 and not getattr(self._place_tls, 'active', False) and
                self._place_min <= nbytes <= self._place_max and nbytes not in self._no_place_sizes):
            return self._alloc_placed(nbytes)
        return self._alloc_raw(nbytes)

    def _alloc_raw(self, nbytes, trim=True):
        p = C.c_void_p()
        try:
            self._check(self._lib.ipa_malloc(self.handle, nbytes, C.byref(p)), 'ipa_malloc')
        except MemoryError:
            if not trim:
                raise
            self.trim()  # give the pooled blocks back and retry once
            self._check(self._lib.ipa_malloc(self.handle, nbytes, C.byref(p)), 'ipa_malloc')
        return p

    def mem_info(self):
        """(free, total) bytes of the context's device"""
        free, total = C.c_size_t(), C.c_size_t()
        self._check(self._lib.ipa_mem_info(self.handle, C.byref(free), C.byref(total)), 'mem_info')
        return free.value, total.value

    def _alloc_placed(self, nbytes):
        """the better of two allocations of `nbytes` by `_probe_block`.  Never more than one extra
        block is held (and none when it would take more than a quarter of the free device memory);
        whatever is not returned goes back to the driver, also when a probe raises; a failed
        candidate allocation leaves the block pool alone.  (The block pool hands a placed block out
        again; once one had to go back to the driver - larger than the pool's limit - its size is
        not placed a second time.)"""
        self._place_tls.active = True
        first = extra = keep = None
        ok = False
        try:
            first = self._alloc_raw(nbytes)
            times = [self._probe_block(first, nbytes)]
            keep = first
            try:
                if nbytes <= self.mem_info()[0] // 4:
                    extra = self._alloc_raw(nbytes, trim=False)
            except MemoryError:
                extra = None
            if extra is not None:
                times.append(self._probe_block(extra, nbytes))
                if times[1] < times[0]:
                    keep = extra
            self.synchronize()
            with self._pool_lock:
                self._placed_ptrs.add(keep.value)
            self.placement_log.append({'nbytes': int(nbytes), 'ms': [round(t, 4) for t in times],
                                       'kept': 0 if keep is first else 1})
            ok = True
            return keep
        finally:
            self._place_tls.active = False
            # on success the candidate that was not chosen goes back to the driver; when anything above
            # raised (a probe, mem_info, the synchronize) BOTH do - nothing is returned to the caller then
            for p in (first, extra):
                if p is not None and (not ok or p is not keep):
                    if not ok:
                        with self._pool_lock:
                            self._placed_ptrs.discard(p.value)
                    self._lib.ipa_free(self.handle, p)

    def _probe_block(self, ptr, nbytes):
        """milliseconds of a plain 3x3 filter streaming the first half of the block into the
        second (float32 rows of 3840 px, whatever the block will hold; the contents do not
        matter): the strip-shaped access pattern of the hot kernels, reading and writing the
        candidate"""
        from . import ops
        w = 3840
        rows = (nbytes // 2) // (4 * w)

        def view(off):
            v = DeviceArray.__new__(DeviceArray)
            v.ctx, v.shape, v.dtype, v.nbytes = self, (rows, w), np.dtype(np.float32), rows * w * 4
            v.ptr = C.c_void_p(ptr.value + off)
            v._owner = False
            return v
        lo, hi = view(0), view(rows * w * 4)
        k3 = np.full((3, 3), 1.0 / 9)
        for _ in range(2):
            ops.conv2d(lo, k3, out=hi)
        e0, e1 = self.event(), self.event()
        e0.record()
        for _ in range(4):
            ops.conv2d(lo, k3, out=hi)
        e1.record()
        self.synchronize()
        return e0.elapsed_ms(e1) / 4

    def _release(self, ptr, nbytes):
        if self.handle is None:
            return
        with self._pool_lock:
            if self._pool_bytes + nbytes <= self._pool_limit:
                self._pool.setdefault(nbytes, []).append(ptr)
                self._pool_bytes += nbytes
                return
            if ptr.value in self._placed_ptrs:
                self._placed_ptrs.discard(ptr.value)
                self._no_place_sizes.add(nbytes)
        self._lib.ipa_free(self.handle, ptr)

    def trim(self):
        """return every pooled device block to the driver"""
        with self._pool_lock:
            blocks = [p for lst in self._pool.values() for p in lst]
            self._pool.clear()
            self._pool_bytes = 0
            for p in blocks:
                self._placed_ptrs.discard(p.value)
            # a trim is the caller's "start over": sizes whose placed block once went back to the driver may
            # be placed again
            self._no_place_sizes.clear()
        for p in blocks:
            self._lib.ipa_free(self.handle, p)

    def empty(self, shape, dtype):
        return DeviceArray(self, shape, dtype)

    def empty_placed(self, shape, dtype, probe, candidates=4, fill=None):
        """An uninitialised array like `empty`, chosen among `candidates` allocations by what
        `probe(array)` measures (milliseconds of the caller's own streaming launch on it; the
        smallest wins), the others freed.  Where a large buffer lands in physical memory moves
        the streaming kernels of this library by up to 10 % on MI355X; a
        caller that allocates its frame batches once can afford the few launches this costs.
        `fill(array)` (e.g. ``lambda a: a.copy_from(src)``) initialises every candidate before
        it is probed.  Returns (array, [milliseconds per candidate])."""
        held, times = [], []
        for _ in range(max(1, int(candidates))):
            try:
                a = DeviceArray(self, shape, dtype)   # (every candidate is held until the choice
            except MemoryError:                       # is made: a freed one would be handed out
                if not held:                          # again)
                    raise
                break   # out of device memory: choose among what there is
            if fill is not None:
                fill(a)
            times.append(float(probe(a)))
            held.append(a)
        best = held[int(np.argmin(times))]
        del held, a
        self.trim()   # the candidates that were not kept go back to the driver
        return best, times

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        d = DeviceArray(self, arr.shape, arr.dtype)
        d.set(arr)
        return d

    def pinned_empty(self, shape, dtype):
        """page-locked host ndarray (ipa_host_alloc): copies to / from it run at PCIe rate and
        do not pass through the driver's staging buffer.  Freed when the array is collected."""
        dtype = np.dtype(dtype)
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.ipa_host_alloc(self.handle, max(nbytes, 1), C.byref(p)), 'host_alloc')
        buf = (C.c_char * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape, dtype=np.int64)))
        arr = arr.reshape(shape)
        lib, handle = self._lib, self.handle
        weakref.finalize(buf, lambda: lib.ipa_host_free(handle, p))
        return arr

    # -- events (HIP events on the stream the kernels run on) --------------
    def event(self):
        return Event(self)

    def close(self):
        if self.handle is not None:
            self.trim()
            self._lib.ipa_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Event(object):
    def __init__(self, ctx):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(ctx._lib.ipa_event_create(ctx.handle, C.byref(h)), 'event_create')
        self.handle = h

    def record(self):
        self.ctx._check(self.ctx._lib.ipa_event_record(self.ctx.handle, self.handle), 'event_record')
        return self

    def elapsed_ms(self, later):
        """milliseconds from this event to `later` (synchronises on `later`)"""
        ms = C.c_float()
        self.ctx._check(self.ctx._lib.ipa_event_elapsed_ms(self.ctx.handle, self.handle,
                                                           later.handle, C.byref(ms)), 'elapsed')
        return ms.value

    def __del__(self):
        try:
            if self.ctx.handle is not None:
                self.ctx._lib.ipa_event_destroy(self.ctx.handle, self.handle)
        except Exception:
            pass


class DeviceArray(object):
    """C-contiguous array in HBM ([frame,] y, x)"""

    def __init__(self, ctx, shape, dtype):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        dtype_id(self.dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self.ptr = ctx._alloc(self.nbytes)
        self._owner = True

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def set(self, arr):
        arr = np.ascontiguousarray(arr, dtype=self.dtype)
        if arr.shape != self.shape:
            raise ValueError('shape mismatch %s vs %s' % (arr.shape, self.shape))
        self.ctx._check(self.ctx._lib.ipa_memcpy_h2d(self.ctx.handle, self.ptr,
                                                     arr.ctypes.data_as(C.c_void_p), self.nbytes),
                        'memcpy_h2d')
        return self

    def get(self, out=None):
        if out is None:
            out = np.empty(self.shape, self.dtype)
        assert out.flags.c_contiguous and out.nbytes == self.nbytes
        self.ctx._check(self.ctx._lib.ipa_memcpy_d2h(self.ctx.handle,
                                                     out.ctypes.data_as(C.c_void_p), self.ptr,
                                                     self.nbytes), 'memcpy_d2h')
        return out

    def copy_from(self, other):
        """stream-ordered device-to-device copy of an equally shaped DeviceArray"""
        if other.shape != self.shape or other.dtype != self.dtype:
            raise ValueError('copy_from needs equal shape and dtype')
        self.ctx._check(self.ctx._lib.ipa_memcpy_d2d(self.ctx.handle, self.ptr, other.ptr,
                                                     self.nbytes), 'memcpy_d2d')
        return self

    def frame(self, i):
        """view of frame i of a (n, h, w) batch (no copy, not owning)"""
        if self.ndim != 3:
            raise ValueError('frame() needs a (n,h,w) batch')
        v = DeviceArray.__new__(DeviceArray)
        v.ctx, v.shape, v.dtype = self.ctx, self.shape[1:], self.dtype
        v.nbytes = self.nbytes // self.shape[0]
        v.ptr = C.c_void_p(self.ptr.value + i * v.nbytes)
        v._owner = False
        v._base = self
        return v

    def free(self):
        if getattr(self, '_owner', False) and self.ptr is not None and self.ctx.handle is not None:
            self.ctx._release(self.ptr, self.nbytes)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default = threading.local()


def default_context(device_id=0):
    """the calling THREAD's context of a device (created on first use).  A context owns one
    stream and one staging workspace and is not re-entrant (ctypes releases the GIL during a
    call), so the implicit context every drop-in wrapper falls back to must not be shared between
    host threads: each thread gets its own.  Pass an explicit ``ctx=`` (or DeviceArrays, which
    carry theirs) to share one deliberately."""
    d = getattr(_default, 'ctx', None)
    if d is None:
        d = _default.ctx = {}
    c = d.get(device_id)
    if c is None or c.handle is None:
        c = d[device_id] = Context(device_id)
    return c


def device_count():
    n = C.c_int()
    rc = L.lib().ipa_device_count(C.byref(n))
    return n.value if rc == 0 else 0


def as_frames(a):
    """(shape-normalised) -> n_frames, h, w for a 2-D image or 3-D batch"""
    if len(a.shape) == 2:
        return 1, a.shape[0], a.shape[1]
    if len(a.shape) == 3:
        return a.shape
    raise ValueError('expected a (h,w) image or (n,h,w) batch, got shape %s' % (a.shape,))
