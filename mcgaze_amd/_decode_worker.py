"""Decode worker of pipeline.FrameCache(processes=True).  Started as a plain child process (`python _decode_worker.py <ring file> <slot
bytes>`, not through multiprocessing: no fork of a process that holds a HIP context, no re-import of the caller's main module, and
this file imports neither torch, numpy nor the package), it reads one request per line on stdin -- "<byte offset> <path>" -- decodes the file
with PIL and writes the RGB uint8 pixels into the memory-mapped ring file (/dev/shm) at that offset, then answers "<h> <w>" on stdout
("-1 <message>" on failure, "0 0" when the frame does not fit a slot: the consumer then decodes it in line).  Only what the host side of
the test pipeline did in line before (LoadImageFromFile.load, mmdet/datasets/pipelines/loading.py:36-82) -- no arithmetic of the hot path."""
import mmap
import os
import sys


def main():
    from PIL import Image          # no numpy here: its import alone is 0.2 s of start-up per helper; PIL hands out the pixel bytes itself
    fd = os.open(sys.argv[1], os.O_RDWR)
    ring = mmap.mmap(fd, 0)
    os.close(fd)
    slot_bytes = int(sys.argv[2])
    out = sys.stdout
    for line in sys.stdin:
        off, path = line.rstrip('\n').split(' ', 1)
        try:
            with Image.open(path) as im:
                rgb = im.convert('RGB')
                w, h = rgb.size
                if h * w * 3 > slot_bytes:
                    out.write('0 0\n')
                else:
                    o = int(off)
                    ring[o:o + h * w * 3] = rgb.tobytes()      # rows of RGB triples: what np.asarray(rgb) holds
                    out.write(f'{h} {w}\n')
        except Exception as e:   # reported to the consumer, which raises it where the frame is asked for
            out.write('-1 ' + repr(e).replace('\n', ' ') + '\n')
        out.flush()


if __name__ == '__main__':
    main()
