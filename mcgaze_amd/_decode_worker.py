"""Decode worker of pipeline.FrameCache(processes=True).  Started as a plain child process (`python _decode_worker.py <ring fd> <slot
bytes> <control fd> <k> <n> <records per queue> <slots> [<directory of the parent's PIL>]`, the two files being unlinked /dev/shm files inherited as descriptors, not through multiprocessing: no fork of a process that holds a HIP context, no
re-import of the caller's main module, and this file imports neither torch, numpy nor the package).  It serves queue ``k`` of the
mailbox pipeline._DecodeProcs lays out in the control file: a request record names a ring slot and a path; the file is decoded with
PIL, its RGB uint8 pixels go into the memory-mapped ring file (/dev/shm) at that slot, then h, w and LAST the state word of the slot
are written (1 = done; 2 = the frame does not fit a slot, the consumer decodes it in line; 3 = failed, the message is in the slot).  The
queue is polled (200 us naps while it is empty, 1 ms after 40 ms of idleness); the loop ends when the stop word is set or the parent
is gone.  Only what the host side of the test pipeline did in line before (LoadImageFromFile.load,
mmdet/datasets/pipelines/loading.py:36-82) -- no arithmetic of the hot path."""
import mmap
import os
import struct
import sys
import time

REC = 1024                             # pipeline._DecodeProcs.REC


def main():
    if len(sys.argv) > 8 and sys.argv[8] and sys.argv[8] not in sys.path:
        sys.path.append(sys.argv[8])   # where the parent found ITS Pillow (the helper runs with -E -s: no PYTHONPATH, no user site)
    from PIL import Image              # no numpy here: its import alone is 0.2 s of start-up per helper; PIL hands out the pixel bytes itself
    ring_fd, slot_bytes, ctl_fd, k, n, R, slots = map(int, sys.argv[1:8])
    ring = mmap.mmap(ring_fd, 0)
    os.close(ring_fd)
    ctl = mmap.mmap(ctl_fd, 0)
    os.close(ctl_fd)
    req_base = (64 + n * 128 + 4095) // 4096 * 4096
    stat_base = req_base + n * R * REC
    head_off, tail_off = 64 + k * 128, 64 + k * 128 + 64
    parent = os.getppid()
    tail, idle = 0, 0
    while True:
        if struct.unpack_from('<q', ctl, head_off)[0] == tail:
            if struct.unpack_from('<q', ctl, 0)[0] or os.getppid() != parent:
                break
            idle += 1
            time.sleep(0.0002 if idle < 200 else 0.001)
            continue
        idle = 0
        off = req_base + (k * R + tail % R) * REC
        slot, plen = struct.unpack_from('<qi', ctl, off)
        path = os.fsdecode(ctl[off + 12:off + 12 + plen])
        st = stat_base + slot * 16
        try:
            with Image.open(path) as im:
                rgb = im.convert('RGB')
                w, h = rgb.size
                if h * w * 3 > slot_bytes:
                    state = 2
                else:
                    ring[slot * slot_bytes:slot * slot_bytes + h * w * 3] = rgb.tobytes()      # rows of RGB triples: what np.asarray(rgb) holds
                    struct.pack_into('<ii', ctl, st + 4, h, w)
                    state = 1
        except Exception as e:         # reported to the consumer, which raises it where the frame is asked for
            msg = repr(e).replace('\n', ' ').encode('utf-8', 'replace')[:slot_bytes]
            ring[slot * slot_bytes:slot * slot_bytes + len(msg)] = msg
            struct.pack_into('<i', ctl, st + 12, len(msg))
            state = 3
        struct.pack_into('<i', ctl, st, state)                   # last: the consumer polls this word
        tail += 1
        struct.pack_into('<q', ctl, tail_off, tail)


if __name__ == '__main__':
    main()
