"""Test-time data pipeline behind the reference's config surface.

``cfg.data.test.pipeline`` of the reference (configs/_base_/datasets/gaze360.py:27-36 and the L2CS variant
configs/multiclue_gaze/multiclue_gaze_r50_l2cs.py:31-39) is a list of transform dicts that mmdet's ``Compose``
(mmdet/datasets/pipelines/compose.py) runs one frame at a time on the CPU.  Here the same dicts build the same-named classes
from a ``PIPELINES`` registry, but the classes only carry the validated options and the GEOMETRY of their transform (crop
window, resized size, padded size, meta keys, and the RNG draws the reference makes, in the reference's order).  The pixel
work of the whole chain runs as ONE HIP launch per clip batch (``mcg_preprocess_frames``): decoded uint8 BGR frames go to the
device as they are, and the kernel writes the model's ``img`` tensor.  There is no CPU pixel path: without the library
``DevicePipeline.__call__`` raises.

    pipe = DevicePipeline(cfg.data.test.pipeline)
    img, img_metas = pipe(frames, device='cuda:0')            # frames: list of HxWx3 uint8 BGR arrays or file names
    model(img=[img], img_metas=[img_metas], return_loss=False, rescale=True, clip_length=7)

Randomness: the reference's ``CenterCrop(crop_type='relative_range')`` draws the crop size from the global numpy RNG at TEST
time too (transforms.py:1126-1130), and ``RandomFlip(flip_ratio=0.0)`` consumes one more uniform
(np.random.choice, transforms.py:463-497).  ``rng`` (default: the global ``np.random``) is drawn from in that order per frame,
so a seeded single-threaded run reproduces the reference's windows; ``crop_u=<float>`` pins the draw instead.
"""
import collections
import concurrent.futures
import ctypes as C
import os
import struct
import threading
import time

import numpy as np
import torch

from . import lib as L
from .registry import Registry

PIPELINES = Registry('pipeline')


class FramePlan:
    """Geometry of one frame on its way through the chain (what the reference keeps in the ``results`` dict)."""

    def __init__(self, shape, filename=None, ori_filename=None):
        self.ori_shape = tuple(shape)
        self.filename, self.ori_filename = filename, ori_filename
        self.crop = (0, 0, shape[0], shape[1])      # y, x, h, w inside the decoded frame
        self.img_shape = tuple(shape)                # current (h, w, c)
        self.pad_shape = None
        self.scale_factor = None
        self.flip, self.flip_direction = False, None
        self.img_norm_cfg = None
        self.resized = False


@PIPELINES.register_module()
class LoadImageFromFile:
    """loading.py:36-82.  Decode happens on the host (PIL); the array stays uint8 BGR like cv2.imread's."""

    def __init__(self, to_float32=False, color_type='color', channel_order='bgr', file_client_args=dict(backend='disk')):
        if to_float32 or color_type != 'color' or channel_order != 'bgr' or file_client_args.get('backend', 'disk') != 'disk':
            raise NotImplementedError('LoadImageFromFile: only uint8 colour BGR frames from disk feed the device pipeline')

    @staticmethod
    def load(filename, rgb=False):
        """-> HxWx3 uint8, BGR like cv2.imread's (rgb=True: the decoder's RGB order as it is -- the pixel kernel swaps channels
        either way, and the per-frame flip copy is saved)."""
        from PIL import Image
        with Image.open(filename) as im:
            arr = np.asarray(im.convert('RGB'))
        return arr if rgb else np.ascontiguousarray(arr[..., ::-1])

    def plan(self, p, rng):
        pass


class _DecodeProcs:
    """``n`` decode helper processes (mcgaze_amd/_decode_worker.py) driven through a MAILBOX in shared memory: a control file next to
    the pixel ring holds one request queue per helper (fixed-size records: ring slot + path; ``head`` advanced by this side, ``tail`` by the
    helper) and one status word per ring slot (0 = being decoded, 1 = done with h, w; 2 = does not fit a slot; 3 = failed, message in the
    slot).  Nobody sleeps in a system call the other side has to wake: the helpers poll their queue (200 us naps when idle), the
    consumer polls the status word of the frame it needs.  Round 4 measured the pipe protocol this replaces at 220 us per ``write`` on
    the GPU box -- a virtual machine, where waking a task on another vCPU is an inter-processor interrupt through the hypervisor -- plus
    one reader thread per helper competing for the interpreter lock.  A request goes to the helper with the shortest queue."""

    REC = 1024                                                   # bytes per request record: int64 slot, int32 path length, path

    def __init__(self, n, ring_fd, slot_bytes, slots, ring, shm_dir=None):
        import mmap
        import tempfile
        import subprocess
        import sys
        self.n, self.slots, self.ring, self.slot_bytes = int(n), int(slots), ring, int(slot_bytes)
        self.R = 1 << max(int(slots) + 2, 2).bit_length()        # records per queue: more than there are slots, so a queue never overflows
        self.req_base = (64 + self.n * 128 + 4095) // 4096 * 4096
        self.stat_base = self.req_base + self.n * self.R * self.REC
        size = self.stat_base + self.slots * 16
        fd, path = tempfile.mkstemp(prefix='mcg_ctl_', dir=shm_dir)
        os.unlink(path)
        os.ftruncate(fd, size)                                   # sparse: only the records in use ever get pages
        self.map = mmap.mmap(fd, size)
        self.stat = np.frombuffer(self.map, dtype=np.int32, count=self.slots * 4, offset=self.stat_base).reshape(self.slots, 4)
        self.tails = np.frombuffer(self.map, dtype=np.int64, count=self.n * 16, offset=64).reshape(self.n, 16)[:, 8]
        self.heads = np.zeros(self.n, dtype=np.int64)
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_decode_worker.py')
        try:
            # -E -s: the helper ignores PYTHON* variables and the user site (a stray sitecustomize must not slow or break seven start-ups), but it
            # is TOLD where the parent's own PIL lives -- a Pillow reachable only through PYTHONPATH or `pip install --user` works
            import PIL
            pil_dir = os.path.dirname(os.path.dirname(os.path.abspath(PIL.__file__)))
            self.procs = [subprocess.Popen([sys.executable, '-E', '-s', worker, str(ring_fd), str(slot_bytes), str(fd), str(k), str(self.n), str(self.R), str(self.slots), pil_dir],
                                           stdin=subprocess.DEVNULL, pass_fds=(ring_fd, fd)) for k in range(self.n)]
        finally:
            os.close(fd)

    def submit(self, path, slot):
        """Queue ``path`` for decoding into ring slot ``slot``.  False: the path does not fit a record (the caller decodes in line)."""
        b = os.fsencode(path)
        if len(b) > self.REC - 12:
            return False
        k = int(np.argmin(self.heads - self.tails))
        h = int(self.heads[k])
        struct.pack_into(f'<qi{len(b)}s', self.map, self.req_base + (k * self.R + h % self.R) * self.REC, slot, len(b), b)
        self.stat[slot, 0] = 0
        self.heads[k] = h + 1
        struct.pack_into('<q', self.map, 64 + k * 128, h + 1)    # publish: the record and the status word are in place
        return True

    def wait(self, slot):
        """-> (state, h, w) of ring slot ``slot`` once its helper is done with it."""
        st = self.stat[slot]
        spins = 0
        while st[0] == 0:
            spins += 1
            if spins > 64:
                time.sleep(5e-5)
                if spins % 4096 == 0 and any(p.poll() is not None for p in self.procs):
                    raise RuntimeError('decode worker exited')
        if st[0] == 3:
            raise RuntimeError('decode worker: ' + bytes(self.ring[slot * self.slot_bytes:slot * self.slot_bytes + int(st[3])]).decode('utf-8', 'replace'))
        return int(st[0]), int(st[1]), int(st[2])

    def close(self):
        struct.pack_into('<q', self.map, 0, 1)                   # stop flag: the helpers leave their loop at the next poll
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
                p.wait()                                         # reap it: no zombie behind a helper that had to be killed
        self.stat = self.tails = self.ring = None
        self.map = None


class FrameCache:
    """Decoded frames by path: each file is decoded once while it stays cached (LRU), optionally ahead of the consumer.

    The reference harness decodes a frame again for every window it belongs to (windows overlap by three frames,
    tools/test_gaze360_gaze.py:88-100).  Here a decode is shared by the windows that use the frame.  With ``workers`` > 0 a pool of
    host threads decodes what ``prefetch`` names ahead of the consumer (the reference uses seven loader threads); measured on the
    GPU box this LOSES to in-line decoding for small frames (a 360x360 JPEG decodes in 0.5 ms; the threads contend with the
    consumer for the interpreter lock and the allocator), so the default is in-line.  ``processes=True`` (round 4) decodes in ``workers``
    helper PROCESSES instead (``_decode_worker.py``, plain children started with subprocess: a process that holds a HIP context is
    not forked -- measured: forked workers made every later host -> device copy of the parent 58 ms slow -- and the caller's main
    module is not re-imported) that write the RGB pixels into a memory-mapped ring file in /dev/shm of ``capacity`` slots (``slot_bytes``
    each; a larger frame is decoded in line); the consumer gets numpy views of the ring (valid while the frame stays cached: run_many
    copies them into its pinned staging buffer at once).
    A ring view is only as good as its slot: a frame handed out as a view is therefore PINNED -- never evicted -- until ``release()``
    (run_many calls it once its staging copy is made); a call that needs more distinct frames at once than the cache has slots raises
    instead of handing out a view whose slot a later decode overwrites.  A caller that uses the cache directly calls ``release()`` when it
    is done with the views it holds.  ``processes=True`` needs (capacity + 1) * slot_bytes of /dev/shm at most (pages are touched as slots
    are used: ``frame_bytes``, the decoded size the caller expects, prices a slot in the check; a node's ranks share the tmpfs,
    LOCAL_WORLD_SIZE); when the tmpfs cannot back that (a container's default /dev/shm is 64 MB), the cache falls back to IN-LINE decoding
    with a warning -- a tmpfs that runs out under a mapped write is a SIGBUS in a helper, not an exception.
    Only the DECODE is shared: crop draws, geometry and the pixel kernel still run per window, in the caller's order -- results do
    not depend on the cache, the worker count or the worker kind."""

    def __init__(self, workers=0, capacity=512, loader=None, processes=False, slot_bytes=3 << 20, frame_bytes=None):
        self.rgb = loader is None                   # our own decode keeps the decoder's RGB order (DevicePipeline.run_many swaps in the kernel)
        self.loader = loader or (lambda path: LoadImageFromFile.load(path, rgb=True))
        self.capacity = max(int(capacity), 1)
        self.ring, self._map, self.procs, self.free, self.slot_bytes = None, None, None, [], int(slot_bytes)
        self.pool = None
        if workers > 0 and processes:
            if loader is not None:
                raise ValueError('FrameCache(processes=True) decodes with its own worker (PIL, RGB)')
            import tempfile
            # both shared files are ANONYMOUS: created in /dev/shm, unlinked at once, handed to the helpers as inherited descriptors -- nothing
            # is left behind there however this process ends
            shm = '/dev/shm' if os.path.isdir('/dev/shm') else None
            # The ring file is sparse: a slot costs the pages its frame touches.  The check prices a slot at ``frame_bytes`` -- the decoded size
            # the caller expects (harness: the first frame of the run, 390 KB for 360 x 360) -- or, unknown, at the whole slot; the tmpfs is
            # shared by the ranks of a node, so each rank may count on its share only (LOCAL_WORLD_SIZE).  
            per_slot = min(int(frame_bytes), self.slot_bytes) if frame_bytes else self.slot_bytes
            per_slot = (max(per_slot, 1) + 4095) // 4096 * 4096
            ranks = max(int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1), 1)
            try:
                vfs = os.statvfs(shm or tempfile.gettempdir())
                fit = int(vfs.f_bavail * vfs.f_frsize * 0.8 / ranks) // per_slot - 1     # slots the file system can back, with a margin
            except OSError:
                fit = self.capacity
            if fit < self.capacity + 1:                          # (no half measures -- a ring smaller than one run_many call's frames would only move the failure)
                import warnings
                warnings.warn(f'FrameCache: {shm or tempfile.gettempdir()} cannot back {self.capacity + 1} ring slots of {per_slot} bytes for each of {ranks} rank(s) '
                              f'(room for {max(fit, 0)}); decoding in line instead of in helper processes')
                processes, workers = False, 0                    # in line, not threads: decode threads LOSE to in-line decoding (class docstring)
        if workers > 0 and processes:
            import tempfile
            fd, path = tempfile.mkstemp(prefix='mcg_ring_', dir=shm)
            os.unlink(path)
            import mmap
            try:
                os.ftruncate(fd, (self.capacity + 1) * self.slot_bytes)           # sparse: a slot gets pages when a frame is written into it
                self._map = mmap.mmap(fd, (self.capacity + 1) * self.slot_bytes)
                self.ring = np.frombuffer(self._map, dtype=np.uint8)   # a plain ndarray: np.memmap's subclass machinery cost 5 us per slice
                self.free = list(range(self.capacity + 1))
                self.procs = _DecodeProcs(int(workers), fd, self.slot_bytes, self.capacity + 1, self.ring, shm)
            finally:
                os.close(fd)
        elif workers > 0:
            self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=int(workers), thread_name_prefix='mcg-decode')
        self.items = collections.OrderedDict()      # path -> Future (threads) / ring slot (processes) / array (in line), least recently used first
        self.used = collections.OrderedDict()       # paths the consumer has already asked for, oldest use first: the eviction candidates
        self.held = set()                           # paths handed out as ring views since the last release(): not evictable
        self.lock = threading.Lock()
        self.decodes = 0
        self.waits, self.wait_s, self.first_wait_s = 0, 0.0, 0.0            # times the consumer found a frame not decoded yet, and how long it then waited

    def _entry(self, path, wanted_now):
        with self.lock:
            e = self.items.get(path)
            if wanted_now:
                self.used[path] = True
                self.used.move_to_end(path)
            if e is not None:
                self.items.move_to_end(path)
                return e
            if self.pool is None and self.procs is None and not wanted_now:
                return None                         # in-line mode: prefetch is a no-op, the decode happens when the frame is asked for
            if self.procs is not None:
                while len(self.items) >= self.capacity:          # make room first: the new decode needs a ring slot
                    self._evict()
                slot = self.free.pop()
                if self.procs.submit(path, slot):
                    e = self.items[path] = slot
                else:                                            # a path too long for a request record: decoded here
                    self.free.append(slot)
                    e = self.items[path] = self.loader(path)
            else:
                e = self.items[path] = self.pool.submit(self.loader, path) if self.pool is not None else self.loader(path)
            self.decodes += 1
            while len(self.items) > self.capacity:
                self._evict()
            return e

    def _evict(self):
        # A frame decoded AHEAD and not yet asked for is the one the consumer needs next: plain LRU threw exactly those out (they are the
        # oldest entries once the cache is full -- every prefetch then cost a second, synchronous decode: round 3's "threads are slower").
        # Victims are frames already consumed, oldest use first; an unconsumed one only when nothing else is left.
        victim, kept = None, []
        while self.used and victim is None:
            p, _ = self.used.popitem(last=False)
            if p in self.held:
                kept.append(p)                                   # its view is still out: not a candidate (back in line below)
            elif p in self.items:
                victim = p
        for p in reversed(kept):
            self.used[p] = True
            self.used.move_to_end(p, last=False)
        if victim is None:
            victim = next((p for p in self.items if p not in self.held), None)
            if victim is None:
                raise RuntimeError(f'FrameCache: all {len(self.items)} cached frames are handed out as views of the decode ring; raise capacity '
                                   f'({self.capacity}) above the distinct frames of one run_many call, or release() between uses')
        old = self.items.pop(victim)
        if isinstance(old, int):                                 # ring slot: its writer must be done before the slot is handed out again
            try:
                self.procs.wait(old)
            except RuntimeError:
                pass
            self.free.append(old)

    def prefetch(self, paths):
        """Start decoding ``paths`` ahead of their use (a no-op for in-line decoding)."""
        if self.pool is not None or self.procs is not None:
            for p in paths:
                self._entry(p, False)

    def _wait(self, e):
        """The decode behind entry ``e`` (a Future of the thread pool or a ring slot of the helper processes), counting the times the
        consumer had to wait for it."""
        if isinstance(e, int):
            if self.procs.stat[e, 0] != 0:
                return self.procs.wait(e)
            wait = lambda: self.procs.wait(e)
        else:
            if e.done():
                return e.result()
            wait = e.result
        t0 = time.perf_counter()
        r = wait()
        dt = time.perf_counter() - t0
        if self.waits == 0:
            self.first_wait_s = dt                               # mostly the helpers' start-up (interpreter + PIL import)
        self.waits += 1
        self.wait_s += dt
        return r

    def __call__(self, path):
        e = self.items.get(path)
        if e is None:
            e = self._entry(path, True)
        else:                                                    # the common case (decoded ahead), without _entry's bookkeeping
            self.used[path] = True
            self.used.move_to_end(path)
        if isinstance(e, int):                                   # decoded by a helper process into ring slot e
            state, h, w = self._wait(e)
            if state == 2:
                return self.loader(path)                         # larger than a slot: decoded here
            o = e * self.slot_bytes
            with self.lock:
                self.held.add(path)                              # a view of the ring: the slot stays until release()
            return self.ring[o:o + h * w * 3].reshape(h, w, 3)
        return self._wait(e) if isinstance(e, concurrent.futures.Future) else e

    def release(self):
        """The views handed out so far are no longer read (run_many: copied into the staging buffer): their slots may be evicted again."""
        with self.lock:
            self.held.clear()

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=False, cancel_futures=True)
        if self.procs is not None:
            self.procs.close()
            self.procs = None
        self.items.clear()
        self.used.clear()
        self.ring = None                                         # (views handed out earlier keep the mapping alive; it goes with the last of them)
        self._map = None


@PIPELINES.register_module()
class CenterCrop:
    """transforms.py:953-1160: a CENTRED window whose size follows ``crop_type`` (the random offsets are commented out
    upstream, :1040-1043); 'relative_range' draws ONE uniform for both sides (:1126-1130)."""

    def __init__(self, crop_size, crop_type='absolute', allow_negative_crop=False, recompute_bbox=False, bbox_clip_border=True, crop_u=None):
        if crop_type not in ['relative_range', 'relative', 'absolute', 'absolute_range']:
            raise ValueError(f'Invalid crop_type {crop_type}.')
        if crop_type in ['absolute', 'absolute_range']:
            assert crop_size[0] > 0 and crop_size[1] > 0
            assert isinstance(crop_size[0], int) and isinstance(crop_size[1], int)
        else:
            assert 0 < crop_size[0] <= 1 and 0 < crop_size[1] <= 1
        self.crop_size, self.crop_type, self.crop_u = crop_size, crop_type, crop_u
        if crop_type == 'relative_range':
            # upstream: cs = float32 array, u = float64 array -> cs + u * (1 - cs) with (1 - cs) rounded in f32 and the rest in f64.
            # The same values as python floats (one rounding per operation, like the array expression), without three array temporaries per frame.
            cs = np.asarray(crop_size, dtype=np.float32)
            self._cs = [float(v) for v in cs]
            self._rest = [float(v) for v in (1 - cs)]

    def _get_crop_size(self, h, w, rng):
        if self.crop_type == 'absolute':
            return min(self.crop_size[0], h), min(self.crop_size[1], w)
        if self.crop_type == 'absolute_range':
            assert self.crop_size[0] <= self.crop_size[1]
            crop_h = rng.randint(min(h, self.crop_size[0]), min(h, self.crop_size[1]) + 1)
            crop_w = rng.randint(min(w, self.crop_size[0]), min(w, self.crop_size[1]) + 1)
            return crop_h, crop_w
        if self.crop_type == 'relative':
            return int(h * self.crop_size[0] + 0.5), int(w * self.crop_size[1] + 0.5)
        u = rng.random_sample() if self.crop_u is None else float(self.crop_u)        # ONE uniform for both sides (:1126-1130; rand(1) draws the same double)
        crop_h, crop_w = self._cs[0] + u * self._rest[0], self._cs[1] + u * self._rest[1]
        return int(h * crop_h + 0.5), int(w * crop_w + 0.5)

    def plan(self, p, rng):
        if p.resized:
            raise NotImplementedError('CenterCrop after Resize is not a chain the device kernel fuses')
        y, x, h, w = p.crop
        ch, cw = self._get_crop_size(h, w, rng)
        assert ch > 0 and cw > 0
        oy, ox = int(max(h - ch, 0) / 2 + 0.5), int(max(w - cw, 0) / 2 + 0.5)
        ch, cw = min(ch, h - oy), min(cw, w - ox)
        p.crop = (y + oy, x + ox, ch, cw)
        p.img_shape = (ch, cw) + p.img_shape[2:]


@PIPELINES.register_module()
class Resize:
    """transforms.py:31-330 for the single-scale test-time use: keep_ratio -> mmcv.imrescale's size rule, else exact size;
    bilinear (cv2.INTER_LINEAR), which is what the kernel implements."""

    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, bbox_clip_border=True, backend='cv2',
                 override=False, interpolation='bilinear'):
        if isinstance(img_scale, list):
            if len(img_scale) != 1:
                raise NotImplementedError('Resize: multi-scale sampling is a training option')
            img_scale = img_scale[0]
        if img_scale is None or ratio_range is not None or backend != 'cv2' or interpolation != 'bilinear':
            raise NotImplementedError('Resize: only img_scale=(w, h) with the cv2 bilinear backend is supported')
        assert isinstance(img_scale, tuple) and len(img_scale) == 2
        self.img_scale, self.keep_ratio = img_scale, keep_ratio

    def plan(self, p, rng):
        if p.resized:
            raise NotImplementedError('two Resize transforms in one pipeline')
        h, w = p.img_shape[:2]
        if self.keep_ratio:
            f = min(max(self.img_scale) / max(h, w), min(self.img_scale) / min(h, w))
            new_w, new_h = int(w * float(f) + 0.5), int(h * float(f) + 0.5)
        else:
            new_w, new_h = self.img_scale
        p.scale_factor = np.array([new_w / w, new_h / h, new_w / w, new_h / h], dtype=np.float32)
        p.img_shape = (new_h, new_w) + p.img_shape[2:]
        p.pad_shape = p.img_shape
        p.resized = True


@PIPELINES.register_module()
class RandomFlip:
    """transforms.py:333-530.  Only the never-flipping test-time setting is supported; it still consumes the uniform
    np.random.choice draws in the reference (:485)."""

    def __init__(self, flip_ratio=None, direction='horizontal'):
        if isinstance(flip_ratio, list) or direction != 'horizontal':
            raise NotImplementedError('RandomFlip: list ratios / non-horizontal directions are training options')
        if isinstance(flip_ratio, float):
            assert 0 <= flip_ratio <= 1
        elif flip_ratio is not None:
            raise ValueError('flip_ratios must be None, float, or list of float')
        if flip_ratio:
            raise NotImplementedError('RandomFlip: flip_ratio > 0 is a training option; the test pipeline uses 0.0')
        self.flip_ratio = flip_ratio

    def plan(self, p, rng):
        if self.flip_ratio is not None:
            rng.random_sample()
        p.flip, p.flip_direction = False, None


@PIPELINES.register_module()
class Normalize:
    """transforms.py:722-760."""

    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb
        self._cfg = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)       # one object for every frame's meta (read-only by convention)

    def plan(self, p, rng):
        p.img_norm_cfg = self._cfg


@PIPELINES.register_module()
class Pad:
    """transforms.py:623-720: zeros below / right, up to a multiple of size_divisor or a fixed size."""

    def __init__(self, size=None, size_divisor=None, pad_to_square=False, pad_val=dict(img=0, masks=0, seg=255)):
        if pad_to_square:
            raise NotImplementedError('Pad: pad_to_square')
        assert size is not None or size_divisor is not None, 'only one of size and size_divisor should be valid'
        assert size is None or size_divisor is None
        v = pad_val.get('img', 0) if isinstance(pad_val, dict) else pad_val
        if v != 0:
            raise NotImplementedError('Pad: the device kernel pads with zeros')
        self.size, self.size_divisor = size, size_divisor

    def plan(self, p, rng):
        h, w = p.img_shape[:2]
        if self.size is not None:
            ph, pw = max(self.size[0], h), max(self.size[1], w)
        else:
            d = self.size_divisor
            ph, pw = -(-h // d) * d, -(-w // d) * d                      # int(np.ceil(h / d)) * d
        p.pad_shape = (ph, pw) + p.img_shape[2:]


@PIPELINES.register_module()
class DefaultFormatBundle:
    """formatting.py:175-260: HWC -> CHW tensor; the kernel writes that layout."""

    def __init__(self, img_to_float=True, pad_val=dict(img=0, masks=0, seg=255)):
        pass

    def plan(self, p, rng):
        if p.pad_shape is None:
            p.pad_shape = p.img_shape
        if p.scale_factor is None:
            p.scale_factor = 1.0


@PIPELINES.register_module()
class ImageToTensor:
    def __init__(self, keys):
        self.keys = keys

    def plan(self, p, rng):
        pass


@PIPELINES.register_module()
class Collect:
    """formatting.py:279-352."""

    def __init__(self, keys, meta_keys=('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'flip',
                                        'flip_direction', 'img_norm_cfg')):
        if list(keys) != ['img']:
            raise NotImplementedError(f'Collect: the test pipeline collects only img (got {keys})')
        self.keys, self.meta_keys = keys, meta_keys

    def plan(self, p, rng):
        pass

    def meta(self, p):
        return {k: getattr(p, k) for k in self.meta_keys}


_DESC = np.dtype([('src', np.uint64)] + [(f, np.int32) for f in ('src_h', 'src_w', 'src_pitch', 'crop_y', 'crop_x', 'crop_h', 'crop_w', 'out_h', 'out_w')],
                 align=True)                       # lib.FrameDesc / mcg_frame_desc, as a numpy record
assert _DESC.itemsize == C.sizeof(L.FrameDesc) and all(_DESC.fields[n][1] == getattr(L.FrameDesc, n).offset for n in _DESC.names)


class DevicePipeline:
    """``Compose(cfg.data.test.pipeline)`` (mmdet/datasets/pipelines/compose.py:11-51) whose pixel work is one HIP launch."""

    def __init__(self, transforms):
        self.transforms = [PIPELINES.build(dict(t)) if isinstance(t, dict) else t for t in transforms]
        kinds = [type(t).__name__ for t in self.transforms]
        if not kinds or kinds[0] != 'LoadImageFromFile':
            raise ValueError('the pipeline must start with LoadImageFromFile')
        if 'Normalize' not in kinds or not any(k in kinds for k in ('DefaultFormatBundle', 'ImageToTensor')) or kinds[-1] != 'Collect':
            raise ValueError(f'unsupported test pipeline {kinds}: need ... Normalize ... DefaultFormatBundle, Collect')
        self.collect = self.transforms[-1]
        self._planners = [t.plan for t in self.transforms if type(t) not in (LoadImageFromFile, ImageToTensor, Collect)]   # the rest plan nothing
        # upload stages, used in turn: a pinned host buffer, its device twin, the event of the copy between them (on the copy stream) and the
        # event of the pixel kernels that read the device twin (on the caller's stream)
        self._stages = [dict(pin=None, dev=None, copied=None, read=None) for _ in range(self.STAGES)]
        self._stage_i, self._copy_stream = 0, {}

    def plan(self, shape, rng=np.random, filename=None, ori_filename=None):
        p = FramePlan(shape, filename, ori_filename)
        for plan in self._planners:
            plan(p, rng)
        if p.pad_shape is None:
            p.pad_shape = p.img_shape
        if p.scale_factor is None:
            p.scale_factor = 1.0
        return p

    def __call__(self, frames, device='cuda:0', rng=np.random, img_prefix=None, stream=None, loader=None):
        """frames: list of HxWx3 uint8 BGR arrays (cv2 order) or file names.  Returns (img [N,3,Hp,Wp] f32 on `device`,
        img_metas list of N dicts) -- one clip batch, padded to the largest padded frame like mmcv's collate.
        loader: path -> HxWx3 uint8 array (default LoadImageFromFile.load; a FrameCache decodes ahead on host threads)."""
        return self.run_many([frames], device, rng, img_prefix, stream, loader)[0]

    STAGES = 4

    def _staging(self, nbytes, dev, main):
        """The next upload stage, its buffers at least nbytes large: a pinned host buffer (a fresh ``pin_memory()`` per call costs
        milliseconds once decode threads compete for the allocator) and a device buffer that only the copy stream writes and only the
        pixel kernels read.  The upload runs on a stream of its own: on the caller's stream it would queue behind the forwards already
        launched there and the PCIe transfer would take its turn with the compute instead of running under it.  STAGES of them, used in
        turn, so that the host fills stage i while the copies and kernels of the stages before it are still queued."""
        st = self._stages[self._stage_i]
        self._stage_i = (self._stage_i + 1) % self.STAGES
        if st['copied'] is not None:
            st['copied'].synchronize()                            # the pinned buffer is about to be overwritten
        cs = self._copy_stream.get(dev)
        if cs is None:
            cs = self._copy_stream[dev] = torch.cuda.Stream(dev)
        if st['pin'] is None or st['pin'].numel() < nbytes or st['dev'].device != dev:
            size = max(int(nbytes * 1.25), 1 << 20)
            st['pin'] = torch.empty(size, dtype=torch.uint8).pin_memory()
            st['dev'] = torch.empty(size, dtype=torch.uint8, device=dev)
            st['read'] = None
            cs.wait_stream(main)                                  # the allocator may hand out memory that work queued on `main` still uses
        return st, cs

    def run_many(self, windows, device='cuda:0', rng=np.random, img_prefix=None, stream=None, loader=None):
        """Several clip batches (windows) through ONE staging copy and one launch per distinct padded size: windows = list of lists
        of frames (arrays or file names) -> list of (img, img_metas), each exactly what ``__call__`` returns for that window -- frames
        are planned window by window, frame by frame, so the RNG draws fall where they do one window at a time; padding is per
        window (mmcv's collate pads a clip to ITS largest frame)."""
        lib = L.load()
        dev = torch.device(device)
        if dev.type != 'cuda':
            raise L.McgError('DevicePipeline runs its pixel work on the GPU (mcg_preprocess_frames); there is no CPU path')
        if dev.index is None:                                     # 'cuda': the current device, by index -- the staging buffers are compared with it
            dev = torch.device('cuda', torch.cuda.current_device())
        rgb_source = all(isinstance(f, str) for w in windows for f in w)      # our own decode: keep PIL's RGB order, the kernel swaps
        if loader is not None:
            load = loader
            rgb_source = rgb_source and bool(getattr(loader, 'rgb', False))
        else:
            load = (lambda path: LoadImageFromFile.load(path, rgb=True)) if rgb_source else LoadImageFromFile.load
        # arrays: the DISTINCT decoded frames of this call (a file named by several windows -- they overlap by three frames -- is staged and
        # uploaded once); src[k]: the array the k-th (window, frame) reads; plans stay per (window, frame): every use draws its own crop
        arrays, src, plans, bounds, seen = [], [], [], [0], {}
        rngs = rng if isinstance(rng, (list, tuple)) else [rng] * len(windows)   # one generator per window (harness: one per VIDEO), or one for all
        if len(rngs) != len(windows):
            raise ValueError(f'run_many: {len(rngs)} generators for {len(windows)} windows')
        try:
            return self._run_many_loaded(windows, rngs, img_prefix, loader, load, rgb_source, arrays, src, plans, bounds, seen, lib, dev, stream)
        except BaseException:
            # a TypeError for a non-uint8 frame, a decode error, a staging failure: the ring views handed out so far are not read any more --
            # without this they stayed pinned and a caller that carried on met 'all cached frames are handed out' later
            if hasattr(loader, 'release'):
                loader.release()
            raise

    def _run_many_loaded(self, windows, rngs, img_prefix, loader, load, rgb_source, arrays, src, plans, bounds, seen, lib, dev, stream):
        for frames, rng in zip(windows, rngs):
            for f in frames:
                if isinstance(f, str):
                    path = os.path.join(img_prefix, f) if img_prefix is not None else f
                    k = seen.get(path)
                    if k is None:
                        arr = load(path)
                        if not rgb_source and getattr(loader, 'rgb', False):
                            arr = np.ascontiguousarray(arr[..., ::-1])       # mixed with caller-supplied BGR arrays: one order per call
                        k = seen[path] = len(arrays)
                        arrays.append(arr)
                    arr, names = arrays[k], (path, f)
                else:
                    arr, names = np.ascontiguousarray(f), (None, None)
                    k = len(arrays)
                    arrays.append(arr)
                if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
                    raise TypeError(f'frames must be HxWx3 uint8 arrays, got {arr.dtype} {arr.shape}')
                src.append(k)
                plans.append(self.plan(arr.shape, rng, *names))
            bounds.append(len(src))
        out = [None] * len(windows)
        n = len(src)
        for wi in range(len(windows)):
            if bounds[wi] == bounds[wi + 1]:
                out[wi] = (torch.empty(0, 3, 0, 0, dtype=torch.float32, device=dev), [])
        if n == 0:
            return out
        offs = np.cumsum([0] + [(a.size + 255) // 256 * 256 for a in arrays])
        # the frame descriptors travel in the same pinned staging buffer, behind the pixels: one non-blocking copy for everything (a
        # pageable .to() per descriptor table blocked the host until the stream had drained -- the engine's previous batch)
        groups = {}                                           # padded size -> windows
        for wi in range(len(windows)):
            a, b = bounds[wi], bounds[wi + 1]
            if b > a:
                pad = (max(p.pad_shape[0] for p in plans[a:b]), max(p.pad_shape[1] for p in plans[a:b]))
                groups.setdefault(pad, []).append(wi)
        dsz = C.sizeof(L.FrameDesc)
        desc_off, total = {}, int(offs[-1])
        for pad, wis in groups.items():
            desc_off[pad] = total
            total += (sum(bounds[wi + 1] - bounds[wi] for wi in wis) * dsz + 255) // 256 * 256
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev) if stream is None else torch.cuda.ExternalStream(stream, device=dev)
            st, cs = self._staging(total, dev, main)
            host, raw = st['pin'], st['dev']
            host_np = host.numpy()
            for a, o in zip(arrays, offs):
                host_np[int(o):int(o) + a.size] = a.reshape(-1)
            if hasattr(loader, 'release'):
                loader.release()                                  # ring views of a FrameCache: copied, their slots may go
            # descriptors (mcg_frame_desc), built as one structured array per padded size
            src_a = np.asarray(src, dtype=np.int64)
            shp = np.asarray([a.shape[:2] for a in arrays], dtype=np.int32)
            geo = np.asarray([p.crop + p.img_shape[:2] for p in plans], dtype=np.int32)          # crop y, x, h, w, out h, w
            for pad, wis in groups.items():
                idx = np.concatenate([np.arange(bounds[wi], bounds[wi + 1]) for wi in wis])
                si = src_a[idx]
                desc = np.zeros(len(idx), dtype=_DESC)
                desc['src'] = raw.data_ptr() + offs[si]
                desc['src_h'], desc['src_w'], desc['src_pitch'] = shp[si, 0], shp[si, 1], shp[si, 1] * 3
                for j, f in enumerate(('crop_y', 'crop_x', 'crop_h', 'crop_w', 'out_h', 'out_w')):
                    desc[f] = geo[idx, j]
                host_np[desc_off[pad]:desc_off[pad] + len(idx) * dsz] = desc.view(np.uint8)
            if st['read'] is not None:
                cs.wait_event(st['read'])                         # the kernels that read this device buffer STAGES calls ago
            with torch.cuda.stream(cs):
                raw[:total].copy_(host[:total], non_blocking=True)
            st['copied'] = torch.cuda.Event()
            st['copied'].record(cs)
            main.wait_event(st['copied'])
        norm = plans[0].img_norm_cfg
        mean = (C.c_float * 3)(*[float(v) for v in norm['mean']])
        stdinv = (C.c_float * 3)(*[float(np.float32(1.0 / np.float64(v))) for v in norm['std']])
        swap = int(bool(norm['to_rgb']) != rgb_source)        # channel swap the kernel performs: wanted order differs from the source's
        s = main.cuda_stream
        for (pad_h, pad_w), wis in groups.items():
            n_idx = sum(bounds[wi + 1] - bounds[wi] for wi in wis)
            img = torch.empty(n_idx, 3, pad_h, pad_w, dtype=torch.float32, device=dev)
            L.check(lib.mcg_preprocess_frames(C.c_void_p(s), C.c_void_p(raw.data_ptr() + desc_off[(pad_h, pad_w)]), n_idx, C.c_void_p(img.data_ptr()), pad_h, pad_w,
                                              mean, stdinv, swap), 'mcg_preprocess_frames')
            at = 0
            for wi in wis:
                cnt = bounds[wi + 1] - bounds[wi]
                out[wi] = (img[at:at + cnt], [self.collect.meta(p) for p in plans[bounds[wi]:bounds[wi + 1]]])
                at += cnt
        st['read'] = torch.cuda.Event()
        st['read'].record(main)
        return out


def build_pipeline(transforms):
    return DevicePipeline(transforms)
