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
