"""Helper for the GPU lab tools: loop a callable for a few seconds while a thread samples rocm-smi (shader clock, package power).
-> (ms per call, mean sclk MHz, mean W, samples).  On a power-capped chip only time, clock and power TOGETHER say what a change did: a
variant that saves cycles at the same energy per call runs at a lower clock and takes the same time (profiles/r05_a_power_map.md)."""
import re, subprocess, threading, time
import torch


def _sample(out, stop):
    while not stop.is_set():
        t = subprocess.run(['rocm-smi', '--showclocks', '--showpower'], capture_output=True, text=True).stdout
        m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
        pw = re.search(r'Power \(W\): ([\d.]+)', t)
        if m and pw:
            out.append((int(m.group(1)), float(pw.group(1))))
        time.sleep(0.08)


def sampled(fn, secs=2.5, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while n < 10 or time.perf_counter() - t0 < 0.25:
        fn()
        n += 1
    torch.cuda.synchronize()
    est = (time.perf_counter() - t0) / n
    iters = max(30, int(secs / est))
    out, stop = [], threading.Event()
    th = threading.Thread(target=_sample, args=(out, stop))
    t0 = time.perf_counter()
    for i in range(iters):
        fn()
        if i == iters // 5:
            th.start()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    stop.set()
    th.join()
    out = out[1:-1] if len(out) > 4 else out
    n = max(1, len(out))
    return ms, sum(o[0] for o in out) / n, sum(o[1] for o in out) / n, len(out)
