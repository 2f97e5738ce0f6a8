"""Decode worker of pipeline.FrameCache(processes=True).  Started as a plain child process (`python _decode_worker.py <ring file> <slot
bytes>`, not through multiprocessing: no fork of a process that holds a HIP context, no re-import of the caller's main module, and
this file imports neither torch nor the package), it reads one request per line on stdin -- "<byte offset> <path>" -- decodes the file
with PIL and writes the RGB uint8 pixels into the memory-mapped ring file (/dev/shm) at that offset, then answers "<h> <w>" on stdout
("-1 <message>" on failure, "0 0" when the frame does not fit a slot: the consumer then decodes it in line).  Only what the host side of
the test pipeline did in line before (LoadImageFromFile.load, mmdet/datasets/pipelines/loading.py:36-82) -- no arithmetic of the hot path."""
import sys

import numpy as np


def main():
    from PIL import Image
    ring = np.memmap(sys.argv[1], dtype=np.uint8, mode='r+')
    slot_bytes = int(sys.argv[2])
    out = sys.stdout
    for line in sys.stdin:
        off, path = line.rstrip('\n').split(' ', 1)
        try:
            with Image.open(path) as im:
                arr = np.asarray(im.convert('RGB'))
            if arr.size > slot_bytes:
                out.write('0 0\n')
            else:
                o = int(off)
                ring[o:o + arr.size] = arr.reshape(-1)
                out.write(f'{arr.shape[0]} {arr.shape[1]}\n')
        except Exception as e:   # reported to the consumer, which raises it where the frame is asked for
            out.write('-1 ' + repr(e).replace('\n', ' ') + '\n')
        out.flush()


if __name__ == '__main__':
    main()
