#!/usr/bin/env python
"""bench.py -- clips/s of the MCGaze per-clip forward path on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (R-50 + FPN per frame, 4 decoder stages, gaze head)
over one batch of synthetic clips already resident in HBM:  --clips-per-gpu clips (default 64 =
BASELINE.json configs[2]) x 7 frames x 3 x 224 x 224 per GPU.  With N > 1 every rank processes its
own 64 clips (weak scaling, configs[3]: 8 x 64 = 512 clips) -- or, with --global-clips G, its share
of a FIXED G clips (strong scaling) -- and the per-rank results are exchanged with ONE fused RCCL
all_gather per step.

One run reports, in ONE JSON line printed by rank 0:
  * the headline (`value`, `dtype`, `ms_per_step`, `roofline`): the --precision engine, default **f16x3** -- the engine that meets
    north_star's 1e-3 on (yaw, pitch) (f32 activations, split-fp16 x 3 MFMA contraction; the library's default) -- with its
    measured deviation from the CPU oracle on clip 0 (`max_abs_dev_yaw_pitch_clip0`, `within_tolerance`);
  * `verified`: after the timed loop, the last batch is re-run strictly serially (one trunk stream, no batch pipeline) and
    must reproduce the timed schedule's outputs BIT FOR BIT;
  * `throughput_engine`: the first --second-engine (default f16: fp16 storage and MFMA, MCG_F16) timed the same way, flagged
    `within_tolerance: false` when its measured deviation exceeds the tolerance, with the MAE clause of north_star measured beside it
    (`mae_shift_deg`, `within_0p05_deg`: from `mae_proxy`) -- reported, never the headline; `other_engines`: further ones (default bf16);
  * `roofline`: dominant contraction kernel, from HIP events around every contraction launch of one UNTIMED sampling step, with the
    algorithmic HBM bytes of every launch beside the PMC traffic;
  * `backbone`: BASELINE.json configs[1] -- R-50 backbone only at 32 clips x 7 frames, both engines;
  * `mae_proxy`: synthetic videos -> device preprocessing -> 7-frame windows (stride 4) -> engine -> overlap merge -> smooth_filter
    -> mean angular error (degrees), every engine against the fp32 CPU oracle's outputs taken as ground truth, and the SHIFT of
    the MAE against a synthetic ground truth placed ~10.7 degrees from the oracle (north_star: within +-0.05 degrees);
  * `latency_single_clip`: one 7-frame clip per forward (BASELINE.json configs[0], the reference harness's actual usage);
  * `host_input`: the same 64-clip step with every batch coming from (pinned) HOST memory -- f32 NCHW as the reference's collate hands it
    over, and uint8 frames normalised on the device -- H2D on a copy stream, double-buffered under the batch pipeline, results
    copied back: the PCIe-inclusive rate (never `value`);
  * `cpu_baseline`: the fp32 oracle on the host cores (rank 0, N = 1 only), best thread count of a sweep, median of >= 10 forwards;
  * N > 1: `world_size` (torch.distributed's), `rccl_ranks_verified` (every rank recomputes clip 0 of its ring neighbour and compares
    it bit for bit with what the all_gather delivered) and `strong_scaling` (a FIXED 512-clip batch sharded over the ranks).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import platform
import sys
import time

# multi-process GPU work on this ROCm stack needs dmabuf IPC (RCCL / tensor sharing across ranks fail with
# `hipIpcGetMemHandle: invalid argument` otherwise); set before the runtime loads, never overriding the caller
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOPS_PER_CLIP_TRUNK = 97.01e9    # backbone + FPN only
FLOPS_PER_CLIP_BACKBONE = 57.22e9  # R-50 alone, 8.174 GFLOP per frame
FLOPS_PER_CLIP = 99.55e9          # 2*MAC over convs + linears + bmms, 7x3x224x224 clip
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
POWER_LIMITED_MATRIX_TFLOPS = 1700.0   # what a pure MFMA loop on dense random fp16 operands holds on this chip (profiles/r05_d_mfma_energy.md)
PARITY_TOL = 1e-3                 # north_star: (yaw, pitch) within 1e-3 rad of the reference CPU path
CFG_NAMES = {0: 'igemm_kernel<float,128,64,64,4,1>', 1: 'igemm_kernel<float,128,64,128,4,1>', 2: 'igemm_kernel<float,128,128,64,2,2>',
             3: 'igemm_kernel<float,128,128,128,2,2>', 4: 'igemm_kernel<bf16,128,64,64,4,1>', 5: 'igemm_kernel<bf16,128,64,128,4,1>',
             6: 'igemm_kernel<bf16,128,128,64,2,2>', 7: 'igemm_kernel<bf16,128,128,128,2,2>',
             # LDS-DMA pipelined kernel: <dtype, BM, BN, K-slice bytes, waves M, waves N, stages>
             15: 'igemm_dma_kernel<bf16,256,64,64,4,1,2>', 25: 'igemm_dma_kernel<bf16,256,128,64,4,2,2>',
             27: 'igemm_dma_kernel<bf16,128,128,64,4,2,2>', 28: 'igemm_dma_kernel<bf16,256,256,64,4,4,3>',
             30: 'igemm_dma_kernel<bf16,256,256,128,4,4,2>', 31: 'igemm_dma_kernel<bf16,128,128,64,4,2,3>', 40: 'conv3x3_c64_kernel', 60: 'pw_pair_kernel (conv3 + next conv1, layer1)',
             61: 'pw_single_kernel (HBM-bound 1x1 convs, register-resident weights)', 62: 'pw_single_kernel<16,2,0,128> (dynamic_layer)',
             # f16x3 contraction (f32 activations, split-packed weights, 3 fp16 MFMAs per product)
             70: 'bneck_x3_kernel (conv2 3x3 + conv3 + next conv1, layer1 / layer2 tails)',
             71: 'pw_single_x3_kernel (HBM-bound 256 -> 256 / 1024 convs, register-resident split weights)',
             72: 'pw_single_x3_kernel<16,0,256> (dynamic_layer)',
             50: 'igemm_dma_kernel<float,256,256,128,4,2,2,2,x3>', 51: 'igemm_dma_kernel<float,128,128,128,2,2,2,2,x3>', 52: 'igemm_dma_kernel<float,256,64,128,4,1,2,2,x3>',
             53: 'igemm_dma_kernel<float,128,128,128,4,2,4,1,x3>',
             73: 'wino_x3w_kernel<NB> + small-grid wino_x3_kernel tiles (3x3 / stride 1 as 1-D Winograd F(2,3); FLOPs booked as the direct convolution)'}   # the Winograd FAMILY: wino_x3w_kernel<NB> (one wave per SIMD; grids of >= 130 workgroups) and the small-grid tiles of wino_x3_kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=250, help='timed steps (default: >= 2 s of timed region at ~10 ms per step)')
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--clips-per-gpu', type=int, default=64)
    ap.add_argument('--global-clips', type=int, default=0,
                    help='strong scaling: a FIXED number of clips per step, sharded over the ranks (e.g. 512 = BASELINE.json configs[3]); 0 = weak scaling with --clips-per-gpu')
    ap.add_argument('--clip-length', type=int, default=7)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--precision', default='f16x3', choices=['bf16', 'f16', 'fp32', 'f16x3'],
                    help='the HEADLINE engine; f16x3 (default) is the one inside north_star\'s 1e-3 tolerance')
    ap.add_argument('--second-engine', '--parity-engine', dest='second_engine', default=None,
                    help="engines timed in the same run beside the headline, comma-separated (of bf16, f16, f16x3, fp32; default f16,bf16 -- none under --fake-engine; 'none' = no second engine).  The first "
                         "is reported as `throughput_engine` (default f16: fp16 storage and MFMA, MCG_F16), the others under `other_engines`; one equal to --precision is skipped")
    ap.add_argument('--exact-steps', type=int, default=10, help='timed steps of the exact_engine leg (fp32 engine, rank 0, N = 1 only; 0 disables)')
    ap.add_argument('--second-steps', '--parity-steps', dest='second_steps', type=int, default=0, help='timed steps of the second engine (0 = steps)')
    ap.add_argument('--backbone-clips', type=int, default=32, help='BASELINE.json configs[1]: clips of the backbone-only sub-measurement (0 disables)')
    ap.add_argument('--mae-videos', type=int, default=8, help='synthetic videos of the mae_proxy leg (rank 0, N = 1 only; 0 disables)')
    ap.add_argument('--strong-clips', type=int, default=512, help='N > 1: fixed global batch of the strong-scaling sub-measurement (0 disables)')
    ap.add_argument('--chunk-frames', type=int, default=0)
    ap.add_argument('--workload', default='full', choices=['full', 'backbone_fpn', 'backbone'],
                    help="'full' = BASELINE.json configs[2] (the metric's configuration); 'backbone_fpn' = configs[1], the trunk alone "
                         "(use --clips-per-gpu 32 for its 32 x 7 frames); 'backbone' = the same without the FPN")
    ap.add_argument('--pipeline', type=int, default=1, choices=[0, 1],
                    help='1: two-deep batch pipeline (decoder of step k overlaps trunk of step k+1 on a second stream; '
                         'every batch is fully processed inside the timed region), 0: one stream, strictly serial')
    ap.add_argument('--trunk-streams', type=int, default=2, help='concurrent frame ranges of the trunk (engine option)')
    ap.add_argument('--decoder-priority', type=int, default=-1, help='HIP stream priority of the batch pipeline\'s decoder stream (-1 = high: the default; 0 = the trunk\'s)')
    ap.add_argument('--cpu-seconds', type=float, default=25.0, help='budget of the cpu_baseline leg (rank 0, N=1 only); 0 disables')
    ap.add_argument('--latency', type=int, default=1, choices=[0, 1], help='1: report single-clip latency (rank 0, N=1 only)')
    ap.add_argument('--power-seconds', type=float, default=1.5, help='N = 1: seconds of the product schedule sampled for package power / shader clock after the timed region (0 disables)')
    ap.add_argument('--host-input-steps', type=int, default=20, help='timed steps of the host_input leg (rank 0, N=1 only; 0 disables)')
    ap.add_argument('--engine-option', action='append', default=[], metavar='NAME=INT',
                    help='mcg_engine_set_option on every engine of the run (A/B of a kernel variant, e.g. winograd=2); recorded in config.engine_options')
    ap.add_argument('--fake-engine', action='store_true',
                    help='a CPU stand-in engine + gloo instead of HipEngine + RCCL: executes THIS FILE\'s N > 1 control flow (clip sharding, '
                         'fused exchange, ring-neighbour check, strong-scaling leg, stdout hand-over) in a container without GPUs '
                         '(tests/test_dist_cpu.py); every number in the line is meaningless and the line says so')
    ap.add_argument('--kernel-events', default='sample', choices=['sample', 'none'],
                    help="'sample': bracket every contraction-kernel launch of one UNTIMED step with HIP events (roofline)")
    return ap.parse_args()


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown.  A host that shows 256
    logical CPUs behind a 16-core quota runs 8 x 16 threads SLOWER than 1 x 16 (CFS throttling): the whole-host leg is sized by this."""
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        return None if q == 'max' else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or 'unknown'


def _cpu_worker(threads, seconds, clip_length, size, seed, ready, go, out, cpus=None):
    """One process of the cpu_baseline's processes x threads leg: single-clip oracle forwards for `seconds` after the common start."""
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)      # before the first parallel region: the OpenMP team inherits the mask
        except OSError:
            pass
    torch.set_num_threads(threads)
    from mcgaze_amd import synth
    from oracle import mcgaze_oracle as orc
    sd = orc.as_torch(synth.make_state_dict(0))
    metas = synth.make_img_metas(clip_length, (size, size, 3))
    clips = synth.make_clips(3 + seed, 2, clip_length, size, size)
    orc.forward(sd, clips[:clip_length], metas, clip_length)
    ready.put(seed)
    go.wait()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        orc.forward(sd, clips[(n % 2) * clip_length:(n % 2 + 1) * clip_length], metas, clip_length)
        n += 1
    out.put((n, time.perf_counter() - t0))


def cpu_multiprocess(procs, threads, seconds, clip_length, size, timeout=90.0):
    """`procs` processes x `threads` torch threads, each running single-clip oracle forwards (the reference harness's usage, one model
    per process) for the same `seconds` window: the host's aggregate rate.  Spawned, not forked (the parent holds a HIP context)."""
    import multiprocessing as mp
    ctx = mp.get_context('spawn')
    ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
    # each process on its own block of CPUs: unpinned, 16 x 16 spinning OpenMP threads ran 3.3 clips/s on a host where ONE 16-thread process runs 10.8
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else list(range(os.cpu_count() or 1))
    blocks = [avail[i * threads:(i + 1) * threads] for i in range(procs)]
    ps = [ctx.Process(target=_cpu_worker, args=(threads, seconds, clip_length, size, i, ready, go, out, blocks[i] if len(blocks[i]) == threads else None), daemon=True)
          for i in range(procs)]
    t_spawn = time.perf_counter()
    for q in ps:
        q.start()
    try:
        for _ in ps:
            ready.get(timeout=timeout)
        go.set()
        res = [out.get(timeout=seconds + timeout) for _ in ps]
    except Exception as e:   # a child that died or a host without the memory for `procs` models: the leg is reported as absent, with the reason
        for q in ps:
            q.terminate()
        return {'error': f'{type(e).__name__}: {e}', 'processes': procs, 'threads_per_process': threads}
    for q in ps:
        q.join(timeout=10)
    n = sum(r[0] for r in res)
    el = max(r[1] for r in res)
    return {'value': round(n / el, 3), 'unit': 'clips/s', 'processes': procs, 'threads_per_process': threads, 'cores': procs * threads,
            'sample': f'{n} single-clip forwards in {el:.1f} s over {procs} processes (one oracle model each, each pinned to its own {threads} CPUs, started together)',
            'startup_s': round(time.perf_counter() - t_spawn - el, 1)}


def cpu_baseline(seconds, clip_length, size):
    """The CPU oracle (fp32 torch restatement proven equal to the reference, tests/test_oracle.py) timed on the host cores on a
    bounded sample of the same synthetic workload: a thread-count sweep picks the fastest setting, then
    >= 10 timed single-clip forwards (the reference harness's usage, tools/test_gaze360_gaze.py:77-111) and a few 8-clip forwards;
    medians are reported."""
    from mcgaze_amd import synth
    from oracle import mcgaze_oracle as orc
    sd = orc.as_torch(synth.make_state_dict(0))
    metas = synth.make_img_metas(clip_length, (size, size, 3))
    clips = synth.make_clips(3, 2, clip_length, size, size)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()

    def one(i):
        t = time.perf_counter()
        orc.forward(sd, clips[(i % 2) * clip_length:(i % 2 + 1) * clip_length], metas, clip_length)
        return time.perf_counter() - t

    t_start = time.perf_counter()
    sweep = {}
    for th in sorted({t for t in (8, 16, 32, 64, 128) if t <= ncpu} | {min(default_threads, ncpu)}):
        torch.set_num_threads(th)
        one(0)                                   # warm-up at this thread count
        sweep[th] = min(one(1), one(2))
        if time.perf_counter() - t_start > seconds * 0.4:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    one(0)
    ts = [one(i) for i in range(10)]
    while time.perf_counter() - t_start < seconds * 0.75 and len(ts) < 40:
        ts.append(one(len(ts)))
    med = float(np.median(ts))
    clips8 = synth.make_clips(3, 8, clip_length, size, size)
    metas8 = synth.make_img_metas(8 * clip_length, (size, size, 3))
    t8 = []
    for _ in range(3):
        t = time.perf_counter()
        orc.forward(sd, clips8, metas8, clip_length)
        t8.append(time.perf_counter() - t)
        if time.perf_counter() - t_start > seconds * 1.2:
            break
    med8 = float(np.median(t8[1:] or t8))
    torch.set_num_threads(default_threads)
    # the whole host: as many `best`-thread processes as the logical CPUs hold (a single torch process does not scale past ~8 threads on
    # this model: the sweep above); bounded to 16 processes and ~6 s
    quota = cpu_quota_cores()
    usable = ncpu // 2 if ncpu >= 32 else ncpu                                        # the first half of the logical CPUs: one hardware thread per core on an SMT-2 host
    if quota is not None:
        usable = min(usable, int(quota))
    procs = min(16, max(1, usable // max(best, 1)))
    multi = cpu_multiprocess(procs, best, min(6.0, seconds * 0.25), clip_length, size) if procs >= 2 else \
        {'skipped': f'{usable} usable cores (logical CPUs {ncpu}, cgroup CPU quota {quota}) hold one {best}-thread process only: the single-process figure IS the host\'s'}
    return {'value': round(1.0 / med, 3), 'unit': 'clips/s', 'cores': best, 'kind': 'port', 'whole_host': multi,
            'sample': f'{len(ts)} single-clip forwards of {clip_length}x3x{size}x{size} (median {med * 1e3:.0f} ms, min {min(ts) * 1e3:.0f} ms), fp32 oracle '
                      f'(oracle/mcgaze_oracle.py) with torch.set_num_threads({best}) -- the fastest of the sweep',
            'thread_sweep_s_per_clip': {str(k): round(v, 3) for k, v in sweep.items()},
            'logical_cpus': ncpu, 'cgroup_cpu_quota_cores': quota, 'cpu_model': cpu_model(), 'torch_default_threads': default_threads,
            'batched8_value': round(8.0 / med8, 3), 'batched8_sample': f'{len(t8)} forwards of 8 clips, median {med8:.2f} s (first is warm-up when more than one)'}


def _sync(dev):
    if dev.type == 'cuda':
        torch.cuda.synchronize(dev)


class FakeEngine:
    """--fake-engine: HipEngine's interface on the CPU with outputs that depend on each frame's pixels only (so a clip's result does not
    depend on its batch or rank, like the real engine's).  NOT a fallback: bench.py without the flag needs the HIP library and a GPU."""
    device = torch.device('cpu')

    def set_option(self, name, value):
        pass

    def forward(self, x, T, img_hw=None, chunk_frames=0, out=None):
        N = x.shape[0]
        m = torch.stack([f.double().sum() for f in x]).float() / x[0].numel()
        gaze = torch.stack([torch.stack([torch.sin(m + k), torch.cos(m * (k + 1)), -torch.ones_like(m)], dim=-1) for k in range(4)])
        boxes = (m[:, None, None] * torch.arange(1, 13, dtype=torch.float32).view(1, 3, 4)).abs() + 1
        scores = (torch.sigmoid(m)[:, None].expand(N, 3) * torch.tensor([1.0, 0.9, 0.4])).contiguous()
        res = dict(gaze=gaze, boxes=boxes, scores=scores)
        if out is None:
            return res
        for k in res:
            out[k].copy_(res[k])
        return out


_COMM_STREAM = {}


def comm_stream(dev):
    """One exchange stream per device for the whole process (streams are created once and shared: engine.hip, StreamPool)."""
    if dev.index not in _COMM_STREAM:
        _COMM_STREAM[dev.index] = torch.cuda.Stream(device=dev, priority=-1)
    return _COMM_STREAM[dev.index]


class Leg:
    """One engine + its schedule (batch pipeline, result exchange) -- the thing a timed region runs."""

    def __init__(self, a, precision, dev, world, rank, dist, img, B, T, workload=None, engine=None):
        from mcgaze_amd import synth
        from mcgaze_amd.dist import ResultGather
        self.a, self.dev, self.dist, self.img, self.B, self.T, self.N = a, dev, dist, img, B, T, B * T
        self.precision = precision
        self.workload = workload or a.workload
        self.fake = bool(a.fake_engine)
        if engine is not None:
            self.eng = engine
        elif self.fake:
            self.eng = FakeEngine()
        else:
            from mcgaze_amd.engine import HipEngine
            self.eng = HipEngine(synth.make_state_dict(0), precision=precision, device=dev)
        self.eng.set_option('trunk_streams', a.trunk_streams)
        for kv in a.engine_option:
            name, _, val = kv.partition('=')
            self.eng.set_option(name, int(val))
        self.gathers = [ResultGather(self.N, world, dev) for _ in range(2)]   # results double-buffered like the pipeline
        self.outs = [g.local_views() for g in self.gathers]                    # the engine writes straight into the fused exchange buffers
        self.runner = None
        if a.pipeline and self.workload == 'full' and not self.fake:
            from mcgaze_amd.engine import PipelinedRunner
            self.runner = PipelinedRunner(self.eng, self.N, a.size, a.size, T, a.chunk_frames, decoder_priority=a.decoder_priority)
        # The result exchange runs on its own stream, ordered only after the decoder that produced the slot: on the caller's stream
        # it would sit between successive submits and serialise batch k+1's trunk behind batch k's decoder (the pipeline's whole point).
        self.comm = comm_stream(dev) if (dist is not None and not self.fake) else None
        self.gathered = [None, None]
        self.k = 0

    def step(self):
        a, eng = self.a, self.eng
        if self.workload == 'backbone_fpn':
            eng.backbone_fpn(self.img, a.chunk_frames)
            return
        if self.workload == 'backbone':
            eng.backbone_only(self.img)
            return
        slot = self.k & 1
        self.k += 1
        if self.fake:   # host engine: the same engine -> fused buffer -> all_gather sequence, without streams
            eng.forward(self.img, self.T, chunk_frames=a.chunk_frames, out=self.outs[slot])
            if self.dist is not None:
                self.gathers[slot].all_gather()
            return
        cur = torch.cuda.current_stream(self.dev)
        if self.comm is not None and self.gathered[slot] is not None:
            cur.wait_event(self.gathered[slot])   # the slot's previous exchange has read the buffer the engine is about to rewrite
        if self.runner is not None:
            done = self.runner.submit(self.img, self.outs[slot])
        else:
            eng.forward(self.img, self.T, chunk_frames=a.chunk_frames, out=self.outs[slot])
            done = cur.record_event()
        if self.comm is not None:
            self.comm.wait_event(done)
            with torch.cuda.stream(self.comm):
                self.gathers[slot].all_gather()
            self.gathered[slot] = self.comm.record_event()

    def drain(self):
        if self.runner is not None:
            self.runner.flush()
        if self.comm is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.comm)

    def barrier(self):
        if self.dist is not None:
            if self.fake:
                self.dist.barrier()
            else:
                self.dist.barrier(device_ids=[self.dev.index])
        _sync(self.dev)

    def serial_forward(self):
        """The strictly serial schedule: one trunk stream, no batch pipeline, the caller's stream only."""
        self.eng.set_option('trunk_streams', 1)
        try:
            out = self.eng.forward(self.img, self.T, chunk_frames=self.a.chunk_frames)
            _sync(self.dev)
        finally:
            self.eng.set_option('trunk_streams', self.a.trunk_streams)
        return out

    def sample_kernels(self):
        """Per-launch durations of every contraction launch of ONE untimed step.  The sampled step runs its trunk on ONE stream:
        with two frame ranges in flight a launch's wall time includes the other range's kernels and says nothing about the kernel."""
        a, eng = self.a, self.eng
        eng.set_option('trunk_streams', 1)
        try:
            eng.profile_start(4096)
            if self.workload == 'full':
                eng.forward(self.img, self.T, chunk_frames=a.chunk_frames)
            elif self.workload == 'backbone_fpn':
                eng.backbone_fpn(self.img, a.chunk_frames)
            else:
                eng.backbone_only(self.img)
            torch.cuda.synchronize(self.dev)
            rec = eng.profile_stop(4096)
        finally:
            eng.set_option('trunk_streams', a.trunk_streams)
        return rec

    def timed(self, steps, warmup):
        """warmup untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; returns seconds (max over ranks)."""
        import contextlib
        # The loop submits from the pipeline's own trunk stream (what a serving loop that owns the runner does): the hand-over
        # "caller's stream -> trunk stream" of every submit is then no cross-queue dependency.  Measured: 2 ms per timed region
        # (one idle-queue hand-over at pipeline fill, one at drain) -- 11.4 -> 9.5 ms for a single step, 8.89 -> 8.74 ms/step at 20.
        ctx = torch.cuda.stream(self.runner.sa) if self.runner is not None else contextlib.nullcontext()
        with ctx:
            for _ in range(max(warmup, 1)):
                self.step()
            self.drain()
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step()
            self.drain()   # the last batch's decoder (and exchange) finishes inside the timed region
            self.barrier()
            elapsed = time.perf_counter() - t0
        if self.dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def run_for(self, seconds):
        """The product schedule for about `seconds` of wall time, outside any timed region (power_probe)."""
        import contextlib
        ctx = torch.cuda.stream(self.runner.sa) if self.runner is not None else contextlib.nullcontext()
        n, t0 = 0, time.perf_counter()
        with ctx:
            while time.perf_counter() - t0 < seconds:
                for _ in range(4):
                    self.step()
                n += 4
                if n % 8 == 0:              # the host enqueues a step in under a millisecond: without this it would run half a minute ahead
                    self.drain()
                    _sync(self.dev)
            self.drain()
        _sync(self.dev)
        return n, time.perf_counter() - t0

    def verify(self):
        """The timed schedule's last outputs (both pipeline slots) against the strictly serial schedule on the same batch: bitwise."""
        if self.workload != 'full':
            return None
        ref = self.serial_forward()
        slots = [0, 1] if self.k >= 2 else [0]
        return all(torch.equal(ref[k], self.outs[s][k]) for s in slots for k in ('gaze', 'boxes', 'scores'))


def roofline_of(rec, precision):
    """rec: engine.profile_stop() tuples (ms, algorithmic flops, cfg, (M, N, K), algorithmic HBM bytes) of ONE step's contraction launches."""
    by = {}
    for t, f, c, _, b in rec:
        d = by.setdefault(c, [0.0, 0.0, 0, 0.0])
        d[0] += t; d[1] += f; d[2] += 1; d[3] += b
    if not by:
        return None
    dom = max(by, key=lambda c: by[c][0])
    t_ms, flops, n, abytes = by[dom]
    achieved = flops / (t_ms * 1e-3) / 1e12
    peak = PEAK_F32_TFLOPS if dom < 4 else PEAK_BF16_TFLOPS
    traffic, step_bytes, covered, tj = None, 0.0, 0, {}
    from mcgaze_amd import lib as _lib
    build_id = _lib.build_id()
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic = (tj.get(CFG_NAMES.get(dom, '')) or {}).get('hbm_bytes_per_launch')
        for c, v in by.items():   # HBM-side bytes of one step's contraction launches: per-symbol PMC average x launches
            b = (tj.get(CFG_NAMES.get(c, '')) or {}).get('hbm_bytes_per_launch')
            if b:
                step_bytes += b * v[2]; covered += v[2]
    algo_step = sum(v[3] for v in by.values())
    r = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
         'traffic': traffic, 'algorithmic_bytes_per_launch': int(abytes / n),
         'traffic_over_algorithmic': round(traffic / (abytes / n), 3) if traffic else None,
         'kernel': CFG_NAMES.get(dom, str(dom)), 'launches_per_step': n,
         'avg_launch_ms': round(t_ms / n, 4), 'algorithmic_gflop_per_launch': round(flops / n / 1e9, 2),
         'all_contraction_launches': {CFG_NAMES.get(c, str(c)): {'launches': v[2], 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                                                 'algorithmic_GB': round(v[3] / 1e9, 3), 'algorithmic_GBps': round(v[3] / 1e9 / (v[0] * 1e-3), 0)}
                                      for c, v in sorted(by.items())},
         # per launch (layer), in execution order: [cfg id, M, N, K, ms, algorithmic TFLOP/s, algorithmic MB, algorithmic GB/s] -- the waste
         # ratio of a layer is its share of `traffic` over the MB here
         'launches': [[c, m[0], m[1], m[2], round(t, 4), round(f / (t * 1e-3) / 1e12, 1), round(b / 1e6, 1), round(b / 1e9 / (t * 1e-3), 0)]
                      for t, f, c, m, b in rec],
         'algorithmic_bytes_step': int(algo_step),
         'sampled': 'every contraction launch of one UNTIMED step after warm-up, HIP events on the launch stream, trunk on one stream '
                    '(trunk_streams=1) so a launch\'s duration is its own; the timed steps run two frame ranges on concurrent streams',
         # the PMC file is committed evidence, not measured in this run: say which library build it was taken on
         'traffic_build_id': tj.get('_build_id'), 'library_build_id': build_id,
         'traffic_build_matches': bool(tj.get('_build_id') == build_id and CFG_NAMES.get(dom, '') in tj.get('_symbols_of_this_build', [])) if traffic else None,
         'traffic_source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_bench_traffic.sh; FETCH_SIZE doubled per '
                           'MI355X_MICROARCH.md); bytes per launch, averaged over the symbol\'s launches; L2-miss traffic incl. Infinity-Cache hits',
         'algorithmic_bytes': 'layer-granular: inputs and residual read once, output written once, weights once (mcg_engine_profile_stop)'}
    # the three largest symbols of the step, each against BOTH roofs, with the committed counter evidence beside the in-run durations
    # (the worst large symbol must be visible in the driver's line, not only the dominant one)
    mj = {}
    mpath = os.path.join(ROOT, 'profiles', 'pmc_mfma.json')
    if os.path.exists(mpath):
        mj = json.load(open(mpath))
    top = []
    for c in sorted(by, key=lambda c: -by[c][0])[:3]:
        t_c, f_c, n_c, b_c = by[c]
        nm = CFG_NAMES.get(c, str(c))
        pk = PEAK_F32_TFLOPS if c < 4 else PEAK_BF16_TFLOPS
        tf = f_c / (t_c * 1e-3) / 1e12
        mult = (2.0 if c == 73 else 3.0) if precision == 'f16x3' else 1.0
        tb = (tj.get(nm) or {}).get('hbm_bytes_per_launch')
        ratio = round(tb / (b_c / n_c), 3) if (tb and b_c) else None
        top.append({'kernel': nm, 'launches': n_c, 'ms': round(t_c, 3), 'share_of_sampled_ms': round(t_c / sum(v[0] for v in by.values()), 3),
                    'tflops': round(tf, 1), 'frac': round(tf / pk, 4), 'matrix_pipe_frac': round(mult * tf / pk, 4),
                    'algorithmic_GBps': round(b_c / 1e9 / (t_c * 1e-3), 0), 'hbm_frac': round(b_c / 1e9 / (t_c * 1e-3) / PEAK_HBM_GBPS, 4),
                    'mfma_busy': (mj.get(nm) or {}).get('mfma_busy'), 'waves_waiting': (mj.get(nm) or {}).get('waves_waiting'),
                    'traffic_over_algorithmic': ratio if (ratio is None or ratio >= 0.95) else None})
    r['top_symbols'] = top
    r['top_symbols_counter_build_ids'] = {'pmc_mfma': mj.get('_build_id'), 'pmc_traffic': tj.get('_build_id'), 'library': build_id}
    # the chip's own power-limited matrix rate: a loop of nothing but dense-random-operand fp16 MFMAs is held at 1.81 GHz / 1.30 kW
    # (profiles/r05_d_mfma_energy.md, tools/lab/micro/mfma_energy.hip, library build 02f203997b6f148b): `frac` read against THAT instead of the nameplate
    r['power_limited_matrix_peak_tflops'] = POWER_LIMITED_MATRIX_TFLOPS
    r['power_limited_matrix_peak_source'] = 'profiles/r05_d_mfma_energy.md: pure v_mfma_f32_32x32x16_f16 loop on dense random operands, 1.70 PFLOP/s at 1.30 kW / 1.81 GHz (round 5 measurement, not re-measured in this run)'
    if precision == 'f16x3':
        r['matrix_pipe_frac_of_power_limited_peak'] = round((2.0 if dom == 73 else 3.0) * achieved / POWER_LIMITED_MATRIX_TFLOPS, 4)
    if precision == 'f16x3':
        # matrix-pipe FLOPs per algorithmic FLOP: three fp16 MFMAs per product; the Winograd F(2,3) family multiplies 6 instead of 9 times
        # per output (x 2 / 3; F(4,3), engine option winograd = 2: x 1 / 2)
        mult = 2.0 if dom == 73 else 3.0
        r['mfma_per_algorithmic_flop'] = mult
        r['note'] = (f'achieved = ALGORITHMIC FLOP/s (a 3x3 conv booked as the direct convolution); this symbol issues {mult:g} matrix-pipe FLOPs per algorithmic '
                     f'FLOP (three fp16 MFMAs per product' + (', 6 instead of 9 products per output: 1-D Winograd F(2,3)' if dom == 73 else '') + '), so the matrix pipe runs at '
                     f'{round(mult * achieved, 1)} TFLOP/s = {round(mult * achieved / peak, 4)} of the 16-bit MFMA peak.  The chip runs every launch class of this path on its '
                     '1.4 kW package cap at 1.75 - 2.2 GHz (profiles/r05_a_power_map.md): a pure MFMA loop on dense random fp16 operands is throttled to 1.81 GHz = 0.68 of the nameplate peak (profiles/r05_d_mfma_energy.md)')
    if r['traffic_over_algorithmic'] is not None and r['traffic_over_algorithmic'] < 0.95:
        # PMC bytes below the algorithmic bytes: a calibration or bookkeeping error (round 4: dynamic_layer launches averaged into the
        # pw_single_x3 symbol), not a kernel that moves less than it must -- not printed as a ratio
        r['traffic_note'] = f"PMC traffic / algorithmic bytes = {r['traffic_over_algorithmic']} < 0.95: counter calibration or symbol bookkeeping is off; ratio withheld"
        r['traffic_over_algorithmic'] = None
    if precision == 'f16':   # the 16-bit cfg ids are named after their bf16 instantiation; this engine runs the fp16 one of the same template
        r = json.loads(json.dumps(r).replace('<bf16,', '<f16,'))
    if step_bytes:
        r['hbm_step'] = {'bytes': int(step_bytes), 'contraction_launches_covered': covered, 'of': len(rec), 'peak_GBps': PEAK_HBM_GBPS,
                         'algorithmic_bytes': int(algo_step), 'over_algorithmic': round(step_bytes / algo_step, 3) if algo_step else None,
                         'note': 'divide by ms_per_step for the whole-path HBM rate; stem, RoIAlign and the small decoder kernels are not in it'}
    return r


def oracle_clip0(img_np, T, size):
    from mcgaze_amd import synth
    from oracle import mcgaze_oracle as orc
    _, ref = orc.forward(synth.make_state_dict(0), img_np[:T], synth.make_img_metas(T, (size, size, 3)), T)
    return orc.yaw_pitch(ref['gaze_score'])


def deviation(out, want_yp, T):
    from oracle import mcgaze_oracle as orc
    return float(orc.wrap_yaw(orc.yaw_pitch(out['gaze'][0][:T].float().cpu()) - want_yp).abs().max())


def single_clip_latency(precisions, dev, T, size, iters=60):
    """BASELINE.json configs[0]: one clip per forward, strictly serial, host-synchronised per call (what the reference harness does)."""
    from mcgaze_amd import synth
    from mcgaze_amd.engine import HipEngine
    img = torch.from_numpy(synth.make_clips(5, 1, T, size, size)).to(dev)
    res = {}
    for p in precisions:
        eng = HipEngine(synth.make_state_dict(0), precision=p, device=dev)
        for _ in range(10):
            eng.forward(img, T)
        torch.cuda.synchronize(dev)
        ts = []
        for _ in range(iters):
            t = time.perf_counter()
            eng.forward(img, T)
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t)
        from mcgaze_amd.engine import GraphedForward
        gf = GraphedForward(eng, T, size, size, T)
        for _ in range(10):
            gf(img)
        torch.cuda.synchronize(dev)
        tg = []
        for _ in range(iters):
            t = time.perf_counter()
            gf(img)
            torch.cuda.synchronize(dev)
            tg.append(time.perf_counter() - t)
        same = all(torch.equal(gf.out[k], eng.forward(img, T)[k]) for k in ('gaze', 'boxes', 'scores'))
        torch.cuda.synchronize(dev)
        rec = None
        eng.profile_start(1024)
        eng.forward(img, T)
        torch.cuda.synchronize(dev)
        rec = eng.profile_stop(1024)
        res[p] = {'ms_per_clip': round(float(np.median(tg)) * 1e3, 3), 'min_ms': round(min(tg) * 1e3, 3), 'clips_per_s': round(1.0 / float(np.median(tg)), 1),
                  'how': 'engine.GraphedForward: the whole forward captured once into a HIP graph, one graph launch + host synchronise per clip (input copy included)',
                  'graph_equals_eager_bitwise': bool(same),
                  'eager_ms_per_clip': round(float(np.median(ts)) * 1e3, 3), 'eager_min_ms': round(min(ts) * 1e3, 3),
                  'contraction_launches': len(rec), 'contraction_ms_sum': round(sum(r[0] for r in rec), 3)}
        del eng
    return res


class _OracleEngine:
    """The CPU oracle behind the engine interface harness._run_windows drives (forward(x, T, img_hw) -> gaze / boxes / scores):
    the checker of the mae_proxy leg, never the thing measured."""

    def __init__(self, sd):
        from oracle import mcgaze_oracle as orc
        self.orc, self.sd, self.device = orc, orc.as_torch(sd), torch.device('cpu')

    def forward(self, x, T, img_hw=None):
        N, _, H, W = x.shape
        gaze, boxes, scores = [], [], []
        for b in range(N // T):
            hw = [(H, W)] * T if img_hw is None else [tuple(int(v) for v in r) for r in np.asarray(img_hw).reshape(-1, 2)[b * T:(b + 1) * T]]
            metas = [dict(img_shape=(h, w, 3), ori_shape=(h, w, 3), pad_shape=(H, W, 3), scale_factor=np.ones(4, dtype=np.float32), flip=False) for h, w in hw]
            det, g = self.orc.forward(self.sd, x[b * T:(b + 1) * T].numpy(), metas, T)
            gaze.append(torch.stack([g['gaze_score'], g['face_gaze_score'], g['eyes_gaze_score'], g['head_gaze_score']]))
            boxes.append(det[..., :4]); scores.append(det[..., 4])
        return dict(gaze=torch.cat(gaze, dim=1), boxes=torch.cat(boxes), scores=torch.cat(scores))


def mae_proxy(engines, dev, n_videos, T, frames_per_video=31, src=320, cpu_threads=16, gt_sigma=0.155):
    """The MAE half of north_star without the dataset (absent here): what a precision mode does to the reference's METRIC.
    Synthetic videos (smooth frame-dependent uint8 content) -> the config's test pipeline on the device (seeded CenterCrop draws,
    Resize, Normalize, Pad: mcg_preprocess_frames) -> 7-frame windows with stride 4 -> engine -> overlap merge
    (tools/test_gaze360_gaze.py:72-206) -> smooth_filter + mean angular error (tools/calculate_mae_gaze360.py:16-29,77-94,136-187).
    Every engine and the fp32 CPU oracle see the SAME preprocessed frames.  Reported per engine, in degrees:
      vs_oracle_deg    MAE with the oracle's own smoothed predictions as ground truth (0 for the oracle itself)
      synthetic_gt_deg MAE against a synthetic ground truth drawn ~10.7 degrees from the oracle's predictions (the README's 10.74)
      shift_deg        synthetic_gt_deg minus the oracle's -- the quantity north_star bounds by +-0.05
    Random-init weights: a trained checkpoint's boxes sit on heads and faces inside the frame, these sprawl, so the
    16-bit engine's figure is an upper bound of unknown slack, not a verdict on the real model."""
    from PIL import Image
    from mcgaze_amd import Config, harness, metric, synth
    from mcgaze_amd.pipeline import DevicePipeline
    pipe = DevicePipeline(Config.fromfile(os.path.join(ROOT, 'configs', 'mcgaze', 'r50_clip7_gaze360.py')).data.test.pipeline)
    rs = np.random.RandomState(2024)
    videos = []
    for vid in range(n_videos):
        base = rs.randint(0, 256, (src // 8 + 1, src // 8 + 1, 3)).astype(np.uint8)
        frames = [np.ascontiguousarray(np.asarray(Image.fromarray(np.roll(base, (i, 2 * i), axis=(0, 1))).resize((src, src), Image.BILINEAR))[..., ::-1])
                  for i in range(frames_per_video)]
        img, metas = pipe(frames, device=dev, rng=np.random.RandomState(100 + vid))
        torch.cuda.synchronize(dev)
        videos.append(dict(id=vid, frames=img, img_hw=np.array([m['img_shape'][:2] for m in metas], dtype=np.int32)))
    sd = synth.make_state_dict(0)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(cpu_threads, os.cpu_count() or 1))
    t0 = time.perf_counter()
    try:
        ref = harness.run_videos(_OracleEngine(sd), [dict(id=v['id'], frames=v['frames'].cpu(), img_hw=v['img_hw']) for v in videos], clip_len=T, batch_clips=8)
    finally:
        torch.set_num_threads(threads)
    cpu_s = time.perf_counter() - t0
    gt_rs = np.random.RandomState(7)
    oracle_gt, synth_gt = dict(annotations=[]), dict(annotations=[])
    for r in ref:
        g = metric.smooth_filter(torch.tensor(r['fusion_gazes']))
        oracle_gt['annotations'].append(dict(gaze=g.tolist()))
        noisy = g + gt_sigma * torch.from_numpy(gt_rs.standard_normal(tuple(g.shape)).astype(np.float32))
        synth_gt['annotations'].append(dict(gaze=(noisy / noisy.norm(dim=1, keepdim=True)).tolist()))
    base = metric.gaze_error(ref, synth_gt, verbose=False)['mae_360']
    out = {'what': mae_proxy.__doc__.split('Reported per engine')[0].strip().replace('\n    ', ' '),
           'videos': n_videos, 'frames': n_videos * frames_per_video, 'windows': sum(len(harness.plan_windows(frames_per_video, T)) for _ in videos),
           'oracle_synthetic_gt_deg': round(base, 4), 'oracle_cpu_seconds': round(cpu_s, 1), 'tolerance_deg': 0.05, 'engines': {}}
    for name, eng in engines.items():
        rec = harness.run_videos(eng, videos, clip_len=T, batch_clips=64)
        torch.cuda.synchronize(dev)
        sg = metric.gaze_error(rec, synth_gt, verbose=False)['mae_360']
        # vs the oracle: the reference's formula is acos(dot) UNCLAMPED (NaN once rounding lifts a dot product of near-identical unit
        # vectors above 1) and has no resolution near zero; here the angle is 2 asin(|a - b| / 2) on the same smoothed predictions
        angs = torch.cat([torch.rad2deg(2 * torch.asin(((metric.smooth_filter(torch.tensor(a['fusion_gazes'])).double() - torch.tensor(b['gaze']).double()).norm(dim=-1) / 2).clamp(max=1)))
                          for a, b in zip(rec, oracle_gt['annotations'])])
        vs, worst = float(angs.mean()), float(angs.max())
        out['engines'][name] = {'vs_oracle_deg': round(vs, 5), 'max_frame_vs_oracle_deg': round(worst, 4), 'synthetic_gt_deg': round(sg, 4),
                                'shift_deg': round(sg - base, 5), 'within_0p05_deg': bool(abs(sg - base) <= 0.05)}
    return out


def host_input_leg(eng, dev, B, T, size, steps, warmup=4):
    """PCIe-inclusive rate of the headline engine: every batch starts in pinned HOST memory and its results end there.
      f32:   [B*T, 3, size, size] f32, normalised (what the reference's collate hands to the model: 4 bytes per value over PCIe)
      uint8: [B*T, size, size, 3] uint8 BGR frames (1 byte per value) -> Normalize (to_rgb) + HWC -> CHW as elementwise device work on the
             copy stream (the dataset path's mcg_preprocess_frames does this together with crop / resize / pad: harness.run_annotation)
    H2D on a copy stream (results return on another) into one of two device buffers (re-filled once the trunk that read it has finished), the two-deep batch pipeline
    (decoder of batch k beside the trunk of batch k + 1), gaze / boxes / scores copied back non-blocking; the timed region ends when the
    last batch's results are on the host."""
    from mcgaze_amd import synth
    from mcgaze_amd.engine import PipelinedRunner
    N = B * T
    out = {}
    for mode in ('f32', 'uint8'):
        runner = PipelinedRunner(eng, N, size, size, T)
        copy = torch.cuda.Stream(device=dev)      # host -> device
        ret = torch.cuda.Stream(device=dev)       # device -> host (its own stream: it waits for a decoder, the next upload must not)
        src = torch.from_numpy(synth.make_clips(3, B, T, size, size))
        if mode == 'uint8':
            host = [(src.permute(0, 2, 3, 1) * 40 + 128).clamp(0, 255).to(torch.uint8).contiguous().pin_memory() for _ in range(2)]
            stage = [torch.empty_like(host[0], device=dev) for _ in range(2)]
            mean = torch.tensor([123.675, 116.28, 103.53], device=dev)
            std = torch.tensor([58.395, 57.12, 57.375], device=dev)
        else:
            host = [src.clone().pin_memory() for _ in range(2)]
        devb = [torch.empty(N, 3, size, size, dtype=torch.float32, device=dev) for _ in range(2)]
        outs = [dict(gaze=torch.empty(4, N, 3, device=dev), boxes=torch.empty(N, 3, 4, device=dev), scores=torch.empty(N, 3, device=dev)) for _ in range(2)]
        hres = [{k: torch.empty(v.shape, dtype=v.dtype).pin_memory() for k, v in outs[0].items()} for _ in range(2)]
        copied = [torch.cuda.Event() for _ in range(2)]
        back = [torch.cuda.Event() for _ in range(2)]

        def step(k):
            slot = k & 1
            with torch.cuda.stream(copy):
                if k >= 2:
                    copy.wait_event(runner.trunk_done[slot])           # the trunk that read this buffer has finished
                if mode == 'uint8':
                    stage[slot].copy_(host[slot], non_blocking=True)
                    # Normalize (to_rgb) + HWC -> CHW on the device: elementwise, on the copy stream, part of what is timed
                    devb[slot].copy_(((stage[slot].flip(-1).to(torch.float32) - mean) / std).permute(0, 3, 1, 2))
                else:
                    devb[slot].copy_(host[slot], non_blocking=True)
                copied[slot].record(copy)
            with torch.cuda.stream(runner.sa):
                runner.sa.wait_event(copied[slot])
                if k >= 2:
                    runner.sb.wait_event(back[slot])                   # the previous results of this slot have left for the host
                done = runner.submit(devb[slot], outs[slot])
            with torch.cuda.stream(ret):
                ret.wait_event(done)
                for kk in outs[slot]:
                    hres[slot][kk].copy_(outs[slot][kk], non_blocking=True)
                back[slot].record(ret)
        for k in range(warmup):
            step(k)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(warmup, warmup + steps):
            step(k)
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        # what reached the host in the last step against a plain forward of the same input (outside the timed region)
        last = (warmup + steps - 1) & 1
        want = eng.forward(devb[last].clone(), T)
        torch.cuda.synchronize(dev)
        ok = all(torch.equal(hres[last][kk], want[kk].cpu()) for kk in want)
        out[mode] = {'value': round(B * steps / el, 2), 'unit': 'clips/s', 'steps': steps, 'ms_per_step': round(el / steps * 1e3, 3),
                     'host_to_device_MB_per_step': round(host[0].numel() * host[0].element_size() / 1e6, 1), 'verified': bool(ok)}
        del runner
    out['what'] = host_input_leg.__doc__.strip().replace('\n    ', ' ')
    return out


def power_probe(leg, seconds=1.5):
    """Package power and shader clock WHILE the product schedule runs (rocm-smi sampled from a thread, in a loop of its own AFTER the timed
    region, so the sampling cannot touch `value`).  The path runs on the package power cap (profiles/r05_a_power_map.md): a step takes its
    energy divided by the cap, and the clock below the chip's maximum is the firmware trading frequency for power -- this object is that
    statement measured in the same run as the headline."""
    import re
    import shutil
    import subprocess
    import threading
    if shutil.which('rocm-smi') is None:
        return None
    dev_arg = ['-d', str(leg.dev.index or 0)]
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                t = subprocess.run(['rocm-smi'] + dev_arg + ['--showclocks', '--showpower'], capture_output=True, text=True, timeout=5).stdout
            except Exception:
                return
            m = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', t)
            pw = re.search(r'Power \(W\): ([\d.]+)', t)
            if m and pw:
                samples.append((int(m.group(1)), float(pw.group(1))))
            time.sleep(0.1)
    th = threading.Thread(target=poll, daemon=True)
    leg.run_for(0.3)                        # clocks settled before the first sample
    th.start()
    n, el = leg.run_for(seconds)
    for _ in range(6):                      # a box whose rocm-smi call takes a second yields one sample per second: keep the schedule running until there are a few
        if len(samples) >= 4:
            break
        n2, el2 = leg.run_for(1.0)
        n, el = n + n2, el + el2
    stop.set()
    th.join(timeout=10)
    s = samples[1:-1] if len(samples) > 4 else samples
    if not s:
        return None
    cap = None
    try:
        t = subprocess.run(['rocm-smi'] + dev_arg + ['--showmaxpower'], capture_output=True, text=True, timeout=5).stdout
        m = re.search(r'Max Graphics Package Power \(W\): ([\d.]+)', t)
        cap = float(m.group(1)) if m else None
    except Exception:
        pass
    w = sum(x[1] for x in s) / len(s)
    return {'package_power_W': round(w, 0), 'power_cap_W': cap, 'frac_of_cap': round(w / cap, 3) if cap else None,
            'sclk_MHz': round(sum(x[0] for x in s) / len(s), 0), 'max_sclk_MHz': 2400, 'samples': len(s),
            'ms_per_step_while_sampled': round(el / n * 1e3, 3), 'joules_per_step': round(w * el / n, 2),
            'how': 'rocm-smi --showclocks --showpower every 0.1 s from a thread while the product schedule loops for ~1.5 s AFTER the timed region (not inside it)'}


def timed_leg(leg, steps, warmup, total_per_step, flops_per_clip, world):
    el = leg.timed(steps, warmup)
    v = total_per_step * steps / el
    return el, {'value': round(v, 2), 'unit': 'clips/s', 'steps': steps, 'ms_per_step': round(el / steps * 1e3, 3),
                'model_tflops': round(v * flops_per_clip / 1e12, 1), 'frac_of_bf16_mfma_peak': round(v * flops_per_clip / 1e12 / (PEAK_BF16_TFLOPS * world), 4)}


def verify_ring_neighbour(leg, world, rank, T, size, strong_b=None):
    """Proof that the exchange really crossed ranks: rank r recomputes clip 0 of rank (r+1) % N -- same seeded generator, its own engine --
    and compares it BIT FOR BIT with that rank's block of the gathered buffer of the last timed step.  -> 1 if equal else 0."""
    from mcgaze_amd import synth
    q = (rank + 1) % world
    slot = (leg.k - 1) & 1
    _sync(leg.dev)
    g = leg.gathers[slot]
    theirs = g.rank_views(q) if world > 1 else g.local_views()
    one = leg.eng.forward(torch.from_numpy(synth.make_clips(3 + q, 1, T, size, size)).to(leg.dev), T)
    _sync(leg.dev)
    ok = (torch.equal(one['gaze'], theirs['gaze'][:, :T]) and torch.equal(one['boxes'], theirs['boxes'][:T]) and torch.equal(one['scores'], theirs['scores'][:T]))
    return int(ok)


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run for N > 1'
    if a.second_engine is None:
        a.second_engine = 'none' if a.fake_engine else 'f16,bf16'
    if a.fake_engine:   # control-flow rehearsal on the host (tests/test_dist_cpu.py): second engines only when asked for (the same stand-in under another name:
                        # what is rehearsed is N ranks walking through several engines' legs, collectives included, in step), no GPU-only legs
        a.kernel_events, a.backbone_clips, a.mae_videos, a.host_input_steps, a.latency, a.cpu_seconds = 'none', 0, 0, 0, 0, 0.0
        a.power_seconds = 0.0
        a.exact_steps = 0
        dev = torch.device('cpu')
    else:
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    dist = None
    if world > 1 or os.environ.get('MCG_BENCH_FORCE_DIST') == '1':   # the env switch exercises the exchange path on a single GPU
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # collective kernels on the high-priority queue pool, next to the decoder and exchange streams and away from the trunk's
        # (streams that share a hardware queue execute in submission order, event waits included -- engine.hip)
        os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
        if a.fake_engine:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from mcgaze_amd import synth
    from mcgaze_amd.dist import shard_clips

    T = a.clip_length
    if a.global_clips > 0:   # strong scaling: this rank's contiguous share of a fixed batch
        c0, c1 = shard_clips(a.global_clips, world, rank)
        B, total_per_step, scaling = c1 - c0, a.global_clips, 'strong'
        assert a.global_clips % world == 0, '--global-clips must be a multiple of the number of GPUs (equal shards keep the result exchange one fused all_gather)'
    else:
        B, total_per_step, scaling = a.clips_per_gpu, a.clips_per_gpu * world, 'weak'
    img_np = synth.make_clips(3 + rank, B, T, a.size, a.size)
    img = torch.from_numpy(img_np).to(dev)
    flops_per_clip = {'full': FLOPS_PER_CLIP, 'backbone_fpn': FLOPS_PER_CLIP_TRUNK, 'backbone': FLOPS_PER_CLIP_BACKBONE}[a.workload]
    want_yp = oracle_clip0(img_np, T, a.size) if (rank == 0 and a.workload == 'full' and not a.fake_engine) else None

    def run_engine(precision, steps, warmup):
        """One engine under the product schedule: sampling step, timed region, bitwise verification, deviation from the oracle,
        the ring-neighbour check of the exchange, and the backbone-only sub-measurement (configs[1]) on the same engine."""
        leg = Leg(a, precision, dev, world, rank, dist, img, B, T)
        for _ in range(2):
            leg.step()
        leg.drain()
        _sync(dev)
        roof = roofline_of(leg.sample_kernels(), precision) if a.kernel_events == 'sample' else None
        el, res = timed_leg(leg, steps, warmup, total_per_step, flops_per_clip, world)
        res['timed_region_s'] = round(el, 3)
        res['verified'] = leg.verify()
        d = deviation(leg.outs[0], want_yp, T) if want_yp is not None else None
        res['max_abs_dev_yaw_pitch_clip0'] = d
        res['tolerance'] = PARITY_TOL
        res['within_tolerance'] = (d is not None and d <= PARITY_TOL) if want_yp is not None else None
        if dist is not None and a.workload == 'full':
            ok = torch.tensor([verify_ring_neighbour(leg, world, rank, T, a.size)], dtype=torch.int64, device=dev)
            dist.all_reduce(ok)
            res['rccl_ranks_verified'] = int(ok.item())
        if roof and 'hbm_step' in roof:
            gbps = roof['hbm_step']['bytes'] / (el / steps) / 1e9
            roof['hbm_step'].update({'achieved_GBps': round(gbps, 1), 'frac': round(gbps / PEAK_HBM_GBPS, 4)})
        res['roofline'] = roof
        if world == 1 and a.power_seconds > 0 and a.workload == 'full':
            res['power'] = power_probe(leg, a.power_seconds)
        back = None
        if world == 1 and a.backbone_clips > 0 and a.workload == 'full':
            del leg.runner
            leg.runner = None
            bimg = torch.from_numpy(synth.make_clips(11, a.backbone_clips, T, a.size, a.size)).to(dev)
            bleg = Leg(a, precision, dev, 1, 0, None, bimg, a.backbone_clips, T, workload='backbone', engine=leg.eng)
            _, back = timed_leg(bleg, max(10, min(steps, 50)), 3, a.backbone_clips, FLOPS_PER_CLIP_BACKBONE, 1)
            # SURVEY.md 8(d) asks for both readings of "R-50 FPN backbone": the same batch through backbone + FPN (mcg_backbone_fpn_forward)
            fleg = Leg(a, precision, dev, 1, 0, None, bimg, a.backbone_clips, T, workload='backbone_fpn', engine=leg.eng)
            _, with_fpn = timed_leg(fleg, max(10, min(steps, 50)), 3, a.backbone_clips, FLOPS_PER_CLIP_TRUNK, 1)
            back['with_fpn'] = {k: with_fpn[k] for k in ('value', 'unit', 'steps', 'ms_per_step', 'model_tflops', 'frac_of_bf16_mfma_peak')}
            del bleg, fleg, bimg
        return leg, res, back

    WHAT = {'f16x3': 'f32 activations, weights split-packed into fp16 high / low halves, three fp16 MFMAs per product, f32 accumulate (include/mcgaze_hip.h MCG_F16X3); the library default',
            'bf16': 'bf16 activations and weights, bf16 MFMA, f32 accumulate; f32 LayerNorm / softmax / boxes (MCG_BF16); explicit opt-in',
            'f16': 'fp16 activations and weights (11 significant bits; range +-65504), fp16 MFMA, f32 accumulate; f32 LayerNorm / softmax / boxes (MCG_F16, round 6): '
                   'the bf16 engine\'s kernels and layouts in fp16; explicit opt-in',
            'fp32': 'f32 storage and f32 MFMA (MCG_F32)'}
    leg, head, head_back = run_engine(a.precision, a.steps, a.warmup)
    engines_for_mae = {a.precision: leg.eng}
    second, second_back, others, others_back = None, None, {}, {}
    seconds = [p for p in dict.fromkeys(a.second_engine.split(',')) if p not in ('none', '', a.precision)]
    for p in seconds:
        assert p in WHAT, f'--second-engine: unknown engine {p!r}'
    a.second_engine = seconds[0] if seconds else 'none'
    for i, prec in enumerate(seconds if a.workload == 'full' else []):
        del leg.runner
        leg.runner = None
        sleg, res_i, back_i = run_engine(prec, a.second_steps or a.steps, max(2, a.warmup))
        if res_i.get('roofline'):
            res_i['roofline'].pop('launches', None)       # the per-launch listing is the headline engine's
        res_i = dict({'dtype': prec, 'what': WHAT[prec],
                      'oracle': 'oracle/mcgaze_oracle.py (fp32 CPU restatement pinned to the reference goldens) on clip 0 of this batch'}, **res_i)
        engines_for_mae[prec] = sleg.eng
        del sleg.runner
        sleg.runner = None
        if i == 0:
            second, second_back = res_i, back_i
        else:
            if res_i.get('roofline'):                    # the first two engines carry the full roofline object; further ones its headline numbers
                res_i['roofline'] = {k: res_i['roofline'].get(k) for k in ('kernel', 'achieved', 'peak', 'unit', 'frac', 'launches_per_step', 'avg_launch_ms')}
            others[prec], others_back[prec] = res_i, back_i

    # ------------------------------------------------------------------ the exact engine beside the headline
    exact = None
    if world == 1 and a.exact_steps > 0 and a.workload == 'full' and 'fp32' not in [a.precision] + seconds:
        del leg.runner
        leg.runner = None
        xleg, exact, _ = run_engine('fp32', a.exact_steps, 2)
        if exact.get('roofline'):
            exact['roofline'].pop('launches', None)
        exact = dict({'dtype': 'fp32', 'what': WHAT['fp32'] + ': the reference\'s own arithmetic (f32 fma chains), reported beside the headline because the '
                                                             'headline arithmetic carries 22-bit operands'}, **exact)
        engines_for_mae['fp32'] = xleg.eng
        del xleg.runner
        xleg.runner = None

    # ------------------------------------------------------------------ N > 1: the fixed-batch (strong scaling) figure in the same line
    strong = None
    if world > 1 and a.strong_clips > 0 and a.global_clips == 0 and a.workload == 'full' and a.strong_clips % world == 0:
        Bs = a.strong_clips // world
        if Bs == B:
            strong = {'global_clips': a.strong_clips, 'clips_per_gpu': Bs, 'value': head['value'], 'ms_per_step': head['ms_per_step'], 'scaling': 'strong',
                      'note': 'identical to the headline configuration at this N (strong clips / N = clips_per_gpu): not timed a second time'}
        else:
            simg = torch.from_numpy(synth.make_clips(3 + rank, Bs, T, a.size, a.size)).to(dev)
            stl = Leg(a, a.precision, dev, world, rank, dist, simg, Bs, T, engine=leg.eng)
            _, strong = timed_leg(stl, max(5, min(a.steps, 10)), 2, a.strong_clips, FLOPS_PER_CLIP, world)
            strong.update({'global_clips': a.strong_clips, 'clips_per_gpu': Bs, 'scaling': 'strong'})
            del stl.runner, stl, simg

    if rank == 0:
        line = {
            'metric': 'clips/sec (7x3x224x224)', 'value': head['value'], 'unit': 'clips/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True, 'scaling': scaling,
            'vs_baseline': None, 'dtype': a.precision,
            'data': 'synthetic (seeded N(0,1) clips, random-init weights, resident in HBM)' if not a.fake_engine else 'FAKE ENGINE (--fake-engine: host stand-in, gloo): a rehearsal of the multi-rank control flow, every number is meaningless',
            'config': {'workload': {'full': 'full multiclue_gaze_r50 forward (R-50 + FPN + 4 decoder stages + gaze head), ', 'backbone_fpn': 'R-50 backbone + FPN only (BASELINE.json configs[1]), ', 'backbone': 'R-50 backbone only, C2..C5 (BASELINE.json configs[1]; mcg_bench_backbone_forward), '}[a.workload] +
                                   f'{B} clips/GPU x {T} frames x 3x{a.size}x{a.size}, {total_per_step} clips/step',
                       'clips_per_gpu': B, 'clip_length': T, 'global_clips': total_per_step, 'chunk_frames': a.chunk_frames,
                       'engine': WHAT[a.precision],
                       'parallelism': f'dp{world} (clips sharded by rank, one fused all_gather of results per step)' if world > 1 else 'single GPU',
                       'batch_pipeline': 'decoder(step k) overlaps trunk(step k+1) on a second HIP stream; the loop submits from the trunk stream; all K batches complete inside the timed region' if (a.pipeline and a.workload == 'full') else 'none (serial)',
                       'trunk_streams': a.trunk_streams, 'engine_options': a.engine_option},
            'world_size': dist.get_world_size() if dist is not None else 1,
            'timed_region_s': head['timed_region_s'],
            'verified': head['verified'],
            'verified_how': 'after the timed loop the batch is re-run strictly serially (trunk_streams=1, no batch pipeline, one stream); gaze / boxes / scores of both pipeline slots must equal it bit for bit',
            'max_abs_dev_yaw_pitch_clip0': head['max_abs_dev_yaw_pitch_clip0'], 'tolerance': PARITY_TOL, 'within_tolerance': head['within_tolerance'],
            'oracle': 'oracle/mcgaze_oracle.py (fp32 CPU restatement pinned to the reference goldens) on clip 0 of this batch',
            'model_tflops': head['model_tflops'],
            'frac_of_bf16_mfma_peak': head['frac_of_bf16_mfma_peak'],
            'roofline': head['roofline'],
        }
        if head.get('power'):
            line['power'] = head['power']
        pfp = os.path.join(ROOT, 'profiles', 'parity_fuzz.json')
        if a.precision in ('f16x3', 'fp32') and os.path.exists(pfp) and not a.fake_engine:
            # how far the 1e-3 on (yaw, pitch) holds beyond the bench clip: committed evidence (tools/parity_fuzz.py against the CPU oracle),
            # quoted with the library build it was taken on -- not measured in this run
            from mcgaze_amd import lib as _lib
            pf = json.load(open(pfp))
            if a.precision in pf:
                line['parity_fuzz'] = {'n': pf[a.precision]['n'], 'within': pf[a.precision]['within'], 'worst_rad': pf[a.precision]['worst_rad'],
                                       'median_rad': pf[a.precision].get('median_rad'),
                                       'trained_family': {k: (pf.get('trained_family') or {}).get(a.precision, {}).get(k) for k in ('n', 'within', 'worst_rad', 'median_rad')} if pf.get('trained_family') else None,
                                       'build_id': pf.get('build_id'), 'build_matches': pf.get('build_id') == _lib.build_id(),
                                       'source': 'profiles/parity_fuzz.json: tools/parity_fuzz.py (random shapes, clip lengths, weight seeds) vs the CPU oracle, tolerance 1e-3 rad; '
                                                 'the inputs beyond it are pinned as tests (tests/test_gpu_forward.py::test_known_fuzz_exceptions_stay_what_they_are)'}
        if os.path.exists(pfp) and not a.fake_engine:
            # the throughput engines in the same fuzz (committed evidence, like the headline's): how far OUTSIDE 1e-3 they are, per population
            pf = json.load(open(pfp))
            from mcgaze_amd import lib as _lib
            for tgt in [second] + list(others.values()):
                if tgt is not None and tgt['dtype'] in pf and tgt['dtype'] != a.precision:
                    e = pf[tgt['dtype']]
                    tgt['parity_fuzz'] = {'n': e['n'], 'within': e['within'], 'worst_rad': e['worst_rad'], 'median_rad': e.get('median_rad'),
                                          'trained_family': (pf.get('trained_family') or {}).get(tgt['dtype']),
                                          'build_id': pf.get('build_id'), 'build_matches': pf.get('build_id') == _lib.build_id(),
                                          'source': 'profiles/parity_fuzz.json (tools/parity_fuzz.py vs the CPU oracle, tolerance 1e-3 rad): a THROUGHPUT engine -- this object says how far outside the tolerance it is'}
        if 'rccl_ranks_verified' in head:
            line['rccl_ranks_verified'] = head['rccl_ranks_verified']
            line['rccl_verified_how'] = verify_ring_neighbour.__doc__.split('->')[0].strip().replace('\n    ', ' ')
        if strong is not None:
            line['strong_scaling'] = strong
        if second is not None:
            line['throughput_engine' if a.second_engine in ('bf16', 'f16') else 'second_engine'] = second
        if others:
            line['other_engines'] = others
        if exact is not None:
            line['exact_engine'] = exact
        if head_back is not None:
            line['backbone'] = {'what': f'BASELINE.json configs[1]: R-50 backbone only (stem + layer1..4, C2..C5; mcg_bench_backbone_forward), {a.backbone_clips} clips x {T} frames x 3x{a.size}x{a.size}, '
                                        f'{FLOPS_PER_CLIP_BACKBONE / 1e9:.2f} GFLOP per clip, two concurrent frame ranges; `with_fpn`: the same batch through backbone + FPN ({FLOPS_PER_CLIP_TRUNK / 1e9:.2f} GFLOP per clip)', a.precision: head_back}
            if second_back is not None:
                line['backbone'][a.second_engine] = second_back
            for pk, bk in others_back.items():
                if bk is not None:
                    line['backbone'][pk] = bk
        if world == 1 and a.mae_videos > 0 and a.workload == 'full':
            from mcgaze_amd.engine import HipEngine
            if 'fp32' not in engines_for_mae:
                engines_for_mae['fp32'] = HipEngine(synth.make_state_dict(0), precision='fp32', device=dev)
            line['mae_proxy'] = mae_proxy(engines_for_mae, dev, a.mae_videos, T)
            for key in ('throughput_engine', 'second_engine'):   # north_star's MAE clause, measured in this run, beside the engine it is about
                if key in line and line[key]['dtype'] in line['mae_proxy']['engines']:
                    m = line['mae_proxy']['engines'][line[key]['dtype']]
                    line[key]['mae_shift_deg'] = m['shift_deg']
                    line[key]['within_0p05_deg'] = m['within_0p05_deg']
            for pk, ok in (line.get('other_engines') or {}).items():
                if pk in line['mae_proxy']['engines']:
                    ok['mae_shift_deg'] = line['mae_proxy']['engines'][pk]['shift_deg']
                    ok['within_0p05_deg'] = line['mae_proxy']['engines'][pk]['within_0p05_deg']
        engines_for_mae.clear()
        if world == 1 and a.host_input_steps > 0 and a.workload == 'full' and a.pipeline:
            line['host_input'] = host_input_leg(leg.eng, dev, B, T, a.size, a.host_input_steps)
        if world == 1 and a.latency and a.workload == 'full':
            del leg
            line['latency_single_clip'] = single_clip_latency([p for p in dict.fromkeys([a.precision] + seconds) if p != 'none'], dev, T, a.size)
        if world == 1 and a.cpu_seconds > 0:
            line['cpu_baseline'] = cpu_baseline(a.cpu_seconds, T, a.size)
            if 'latency_single_clip' in line:
                line['latency_single_clip']['cpu_oracle_ms_per_clip'] = round(1e3 / line['cpu_baseline']['value'], 1)
    else:
        line = None
    # The JSON line is the LAST thing on stdout, across all ranks: RCCL writes a banner ("RCCL version ... Librccl path") through C
    # stdio, which is block-buffered on a pipe and would otherwise surface at process exit, after Python's line.  Every rank pushes
    # what it has buffered out now; ranks other than 0 then point their stdout at stderr; rank 0 prints after the barrier and does
    # the same before the process group is torn down.
    sys.stdout.flush()
    C.CDLL(None).fflush(None)
    if rank != 0:
        os.dup2(2, 1)
    if dist is not None:
        if a.fake_engine:
            dist.barrier()
        else:
            dist.barrier(device_ids=[local])
    if line is not None:
        print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
