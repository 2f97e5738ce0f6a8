"""GPU tool (lab): can a /dev/shm mapping be registered as pinned host memory and used as the source of an asynchronous H2D copy?
usage: python tools/lab/shm_pin_probe.py [MiB=256]"""
import mmap, os, sys, tempfile, time
import numpy as np
import torch
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
size = mib << 20
fd, path = tempfile.mkstemp(prefix='mcg_pin_', dir='/dev/shm')
os.unlink(path)
os.ftruncate(fd, size)
m = mmap.mmap(fd, size)
a = np.frombuffer(m, dtype=np.uint8)
a[:] = 7                                    # touch every page
t = torch.from_numpy(a)
torch.cuda.init()
d = torch.empty(size, dtype=torch.uint8, device='cuda')
def h2d(tag):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    d.copy_(t, non_blocking=True)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{tag}: is_pinned={t.is_pinned()}  call returned after {(t1 - t0) * 1e3:.2f} ms, done after {(t2 - t0) * 1e3:.2f} ms ({size / (t2 - t0) / 1e9:.1f} GB/s)')
h2d('pageable shm mapping')
t0 = time.perf_counter()
rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), size, 0)
print(f'cudaHostRegister -> {rc} in {(time.perf_counter() - t0) * 1e3:.1f} ms')
h2d('registered shm mapping')
h2d('registered shm mapping (again)')
# a second process writing into the same file would see the same pages: write through the mmap object and copy again
a[:16] = 9
d.copy_(t, non_blocking=True); torch.cuda.synchronize()
print('device sees the new bytes:', d[:16].tolist() == [9] * 16)
rc = torch.cuda.cudart().cudaHostUnregister(t.data_ptr())
print('cudaHostUnregister ->', rc)
