#!/usr/bin/env python
"""bench.py -- clips/s of the MCGaze per-clip forward path on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (R-50 + FPN per frame, 4 decoder stages, gaze head)
over one batch of synthetic clips already resident in HBM:  --clips-per-gpu clips (default 64 =
BASELINE.json configs[2]) x 7 frames x 3 x 224 x 224 per GPU, bf16 MFMA engine.  With N > 1 every
rank processes its own 64 clips (weak scaling, configs[3]: 8 x 64 = 512 clips) and the per-rank
results are exchanged with ONE fused RCCL all_gather per step.

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOPS_PER_CLIP_TRUNK = 97.01e9    # backbone + FPN only
FLOPS_PER_CLIP_BACKBONE = 57.22e9  # R-50 alone, 8.174 GFLOP per frame
FLOPS_PER_CLIP = 99.55e9          # 2*MAC over convs + linears + bmms, 7x3x224x224 clip
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
CFG_NAMES = {0: 'igemm_kernel<float,128,64,64,4,1>', 1: 'igemm_kernel<float,128,64,128,4,1>', 2: 'igemm_kernel<float,128,128,64,2,2>',
             3: 'igemm_kernel<float,128,128,128,2,2>', 4: 'igemm_kernel<bf16,128,64,64,4,1>', 5: 'igemm_kernel<bf16,128,64,128,4,1>',
             6: 'igemm_kernel<bf16,128,128,64,2,2>', 7: 'igemm_kernel<bf16,128,128,128,2,2>',
             # LDS-DMA pipelined kernel: <dtype, BM, BN, K-slice bytes, waves M, waves N, stages>
             15: 'igemm_dma_kernel<bf16,256,64,64,4,1,2>', 16: 'igemm_dma_kernel<bf16,128,128,64,2,2,4>',
             17: 'igemm_dma_kernel<bf16,256,128,64,2,2,3>', 18: 'igemm_dma_kernel<bf16,256,128,64,4,2,3>',
             19: 'igemm_dma_kernel<bf16,256,256,64,4,2,3>', 24: 'igemm_dma_kernel<bf16,128,128,64,2,2,2>',
             25: 'igemm_dma_kernel<bf16,256,128,64,4,2,2>', 26: 'igemm_dma_kernel<bf16,128,128,64,2,2,3>',
             27: 'igemm_dma_kernel<bf16,128,128,64,4,2,2>', 28: 'igemm_dma_kernel<bf16,256,256,64,4,4,3>',
             29: 'igemm_dma_kernel<bf16,256,256,64,4,4,2>', 30: 'igemm_dma_kernel<bf16,256,256,128,4,4,2>', 31: 'igemm_dma_kernel<bf16,128,128,64,4,2,3>', 32: 'igemm_dma_kernel<bf16,256,128,64,4,2,3> (<=128 VGPRs)', 40: 'conv3x3_c64_kernel',
             # bf16x3 contraction (f32 activations, split-packed weights, 3 bf16 MFMAs per product)
             50: 'igemm_dma_kernel<float,256,256,128,4,2,2,2,x3>', 51: 'igemm_dma_kernel<float,128,128,128,2,2,2,2,x3>', 52: 'igemm_dma_kernel<float,256,64,128,4,1,2,2,x3>'}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--clips-per-gpu', type=int, default=64)
    ap.add_argument('--clip-length', type=int, default=7)
    ap.add_argument('--size', type=int, default=224)
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32', 'bf16x3'])
    ap.add_argument('--chunk-frames', type=int, default=0)
    ap.add_argument('--workload', default='full', choices=['full', 'backbone_fpn', 'backbone'],
                    help="'full' = BASELINE.json configs[2] (the metric's configuration); 'backbone_fpn' = configs[1], the trunk alone "
                         "(use --clips-per-gpu 32 for its 32 x 7 frames); 'backbone' = the same without the FPN")
    ap.add_argument('--pipeline', type=int, default=1, choices=[0, 1],
                    help='1: two-deep batch pipeline (decoder of step k overlaps trunk of step k+1 on a second stream; '
                         'every batch is fully processed inside the timed region), 0: one stream, strictly serial')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='budget of the cpu_baseline leg (rank 0, N=1 only); 0 disables')
    ap.add_argument('--kernel-events', default='first', choices=['first', 'none'],
                    help="'first': bracket every contraction-kernel launch of the FIRST timed step with HIP events")
    return ap.parse_args()


def cpu_baseline(seconds, clip_length, size):
    """The CPU oracle (fp32 torch restatement proven equal to the reference, tests/test_oracle.py)
    timed on the host cores, one clip per forward like the reference harness
    (tools/test_gaze360_gaze.py:77-111), on a bounded sample of the same synthetic workload."""
    from mcgaze_amd import synth
    from oracle import mcgaze_oracle as orc
    sd = orc.as_torch(synth.make_state_dict(0))
    metas = synth.make_img_metas(clip_length, (size, size, 3))
    clips = synth.make_clips(3, 2, clip_length, size, size)
    orc.forward(sd, clips[:clip_length], metas, clip_length)  # warm-up
    n, t0 = 0, time.time()
    while time.time() - t0 < seconds * 2 / 3:
        orc.forward(sd, clips[(n % 2) * clip_length:(n % 2 + 1) * clip_length], metas, clip_length)
        n += 1
    dt = time.time() - t0
    # second leg: 8 clips per forward -- what the oracle gains from batching on the same cores
    clips8 = synth.make_clips(3, 8, clip_length, size, size)
    metas8 = synth.make_img_metas(8 * clip_length, (size, size, 3))
    n8, t1 = 0, time.time()
    while n8 == 0 or time.time() - t1 < seconds / 3:
        orc.forward(sd, clips8, metas8, clip_length)
        n8 += 8
    dt8 = time.time() - t1
    return {'value': round(n / dt, 3), 'unit': 'clips/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': f'{n} clips of {clip_length}x3x{size}x{size}, one clip per forward, fp32 oracle (oracle/mcgaze_oracle.py), '
                      f'{dt:.1f} s on {os.cpu_count()} logical CPUs',
            'batched8_value': round(n8 / dt8, 3), 'batched8_sample': f'{n8} clips, 8 per forward, {dt8:.1f} s'}


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == a.gpus, f'--gpus {a.gpus} but WORLD_SIZE={world}: launch through torch.distributed.run for N > 1'
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
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from mcgaze_amd import lib as L
    from mcgaze_amd import synth
    from mcgaze_amd.engine import HipEngine
    from mcgaze_amd.dist import ResultGather

    if a.workload == 'backbone':
        os.environ['MCG_TRUNK_STOP'] = 'backbone'
    lib = L.load()
    B, T = a.clips_per_gpu, a.clip_length
    N = B * T
    eng = HipEngine(synth.make_state_dict(0), precision=a.precision, device=dev)
    img = torch.from_numpy(synth.make_clips(3 + rank, B, T, a.size, a.size)).to(dev)
    from mcgaze_amd.engine import PipelinedRunner
    gathers = [ResultGather(N, world, dev) for _ in range(2)]   # results double-buffered like the pipeline
    outs = [g.local_views() for g in gathers]                    # the engine writes straight into the fused exchange buffers
    runner = PipelinedRunner(eng, N, a.size, a.size, T, a.chunk_frames) if a.pipeline and a.workload == 'full' else None
    eng.forward(img, T, chunk_frames=a.chunk_frames, out=outs[0])
    state = {'k': 0}

    # The result exchange runs on its own stream, ordered only after the decoder that produced the slot: on the caller's stream
    # it would sit between successive submits and serialise batch k+1's trunk behind batch k's decoder (the pipeline's whole point).
    comm = torch.cuda.Stream(device=dev, priority=-1) if dist is not None else None
    gathered = [None, None]

    def step():
        if a.workload != 'full':
            eng.backbone_fpn(img, a.chunk_frames)
            return
        slot = state['k'] & 1
        state['k'] += 1
        cur = torch.cuda.current_stream(dev)
        if comm is not None and gathered[slot] is not None:
            cur.wait_event(gathered[slot])   # the slot's previous exchange has read the buffer the engine is about to rewrite
        if runner is not None:
            done = runner.submit(img, outs[slot])
        else:
            eng.forward(img, T, chunk_frames=a.chunk_frames, out=outs[slot])
            done = cur.record_event()
        if comm is not None:
            comm.wait_event(done)
            with torch.cuda.stream(comm):
                gathers[slot].all_gather()
            gathered[slot] = comm.record_event()

    def drain():
        if runner is not None:
            runner.flush()
        if comm is not None:
            torch.cuda.current_stream(dev).wait_stream(comm)

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize(dev)

    for _ in range(max(a.warmup, 1)):
        step()
    drain()
    torch.cuda.synchronize(dev)
    launches = 0
    user_streams = os.environ.get('MCG_TRUNK_STREAMS')
    if a.kernel_events == 'first':  # count contraction-kernel launches per step (untimed), then arm for the first timed step
        # The sampled step runs its trunk on ONE stream: with two frame ranges in flight (the default, engine.hip) a launch's
        # wall time includes the other range's kernels and says nothing about the kernel itself.
        os.environ['MCG_TRUNK_STREAMS'] = '1'
        cnt = C.c_int()
        L.check(lib.mcg_profile_start(4096), 'mcg_profile_start')
        if a.workload == 'full':
            eng.forward(img, T, chunk_frames=a.chunk_frames, out=outs[0])
        else:
            eng.backbone_fpn(img, a.chunk_frames)
        torch.cuda.synchronize(dev)
        L.check(lib.mcg_profile_stop(C.byref(cnt), None, None, None, None, 4096), 'mcg_profile_stop')
        launches = cnt.value
        L.check(lib.mcg_profile_start(launches), 'mcg_profile_start')

    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step()
        if i == 0 and a.kernel_events == 'first':  # sampling done: back to the configured number of concurrent frame ranges
            if user_streams is None:
                os.environ.pop('MCG_TRUNK_STREAMS', None)
            else:
                os.environ['MCG_TRUNK_STREAMS'] = user_streams
    drain()  # the last batch's decoder finishes inside the timed region
    barrier()
    elapsed = time.perf_counter() - t0

    roofline = None
    if a.kernel_events == 'first':
        cnt = C.c_int()
        ms = (C.c_float * launches)(); fl = (C.c_double * launches)(); cf = (C.c_int * launches)()
        L.check(lib.mcg_profile_stop(C.byref(cnt), ms, fl, cf, None, launches), 'mcg_profile_stop')
        rec = [(ms[i], fl[i], cf[i]) for i in range(cnt.value)]
        by = {}
        for t, f, c in rec:
            d = by.setdefault(c, [0.0, 0.0, 0])
            d[0] += t; d[1] += f; d[2] += 1
        dom = max(by, key=lambda c: by[c][0])
        t_ms, flops, n = by[dom]
        achieved = flops / (t_ms * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if dom >= 4 else PEAK_F32_TFLOPS   # bf16x3 (50..52) is priced against the bf16 peak too: it issues 3 bf16 MFMAs per algorithmic product
        traffic, step_bytes, covered = None, 0.0, 0
        tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = (tj.get(CFG_NAMES[dom]) or {}).get('hbm_bytes_per_launch')
            for c, v in by.items():   # HBM-side bytes of one step's contraction launches: per-symbol PMC average x launches
                b = (tj.get(CFG_NAMES.get(c, '')) or {}).get('hbm_bytes_per_launch')
                if b:
                    step_bytes += b * v[2]; covered += v[2]
        roofline = {'bound': 'mfma', 'achieved': round(achieved, 2), 'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                    'traffic': traffic, 'kernel': CFG_NAMES[dom], 'launches_per_step': n,
                    'avg_launch_ms': round(t_ms / n, 4), 'algorithmic_gflop_per_launch': round(flops / n / 1e9, 2),
                    'all_contraction_launches': {CFG_NAMES[c]: {'launches': v[2], 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1)}
                                                 for c, v in sorted(by.items())},
                    'sampled': "every contraction launch of the first timed step, HIP events on the launch stream; that step's trunk runs on one stream (MCG_TRUNK_STREAMS=1) so a launch's duration is its own -- the other steps run two frame ranges on concurrent streams",
                    'traffic_source': 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/pmc_bench_traffic.sh); bytes per launch'}
        if step_bytes:
            roofline['hbm_step'] = {'bytes': int(step_bytes), 'contraction_launches_covered': covered, 'of': len(rec), 'peak_GBps': PEAK_HBM_GBPS,
                                    'note': 'divide by ms_per_step for the whole-path HBM rate; stem, RoIAlign and the small decoder kernels are not in it'}

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        total_clips = B * world * a.steps
        flops_per_clip = {'full': FLOPS_PER_CLIP, 'backbone_fpn': FLOPS_PER_CLIP_TRUNK, 'backbone': FLOPS_PER_CLIP_BACKBONE}[a.workload]
        value = total_clips / elapsed
        if roofline and 'hbm_step' in roofline:
            gbps = roofline['hbm_step']['bytes'] / (elapsed / a.steps) / 1e9
            roofline['hbm_step'].update({'achieved_GBps': round(gbps, 1), 'frac': round(gbps / PEAK_HBM_GBPS, 4)})
        line = {
            'metric': 'clips/sec (7x3x224x224)', 'value': round(value, 2), 'unit': 'clips/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(elapsed / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': a.precision, 'data': 'synthetic (seeded N(0,1) clips, random-init weights, resident in HBM)',
            'config': {'workload': {'full': 'full multiclue_gaze_r50 forward (R-50 + FPN + 4 decoder stages + gaze head), ', 'backbone_fpn': 'R-50 backbone + FPN only (BASELINE.json configs[1]), ', 'backbone': 'R-50 backbone only, C2..C5 (BASELINE.json configs[1]; MCG_TRUNK_STOP=backbone), '}[a.workload] +
                                   f'{B} clips/GPU x {T} frames x 3x{a.size}x{a.size}, {B * world} clips/step',
                       'clips_per_gpu': B, 'clip_length': T, 'global_clips': B * world, 'chunk_frames': a.chunk_frames,
                       'parallelism': f'dp{world} (clips sharded by rank, one fused all_gather of results per step)' if world > 1 else 'single GPU',
                       'batch_pipeline': 'decoder(step k) overlaps trunk(step k+1) on a second HIP stream; all K batches complete inside the timed region' if (a.pipeline and a.workload == 'full') else 'none (serial)',
                       'trunk_streams': int(os.environ.get('MCG_TRUNK_STREAMS', '2'))},
            'model_tflops': round(value * flops_per_clip / 1e12, 1),
            'frac_of_bf16_mfma_peak': round(value * flops_per_clip / 1e12 / (PEAK_BF16_TFLOPS * world), 4),
            'roofline': roofline,
        }
        if world == 1 and a.cpu_seconds > 0:
            line['cpu_baseline'] = cpu_baseline(a.cpu_seconds, T, a.size)
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
        dist.barrier(device_ids=[local])
    if line is not None:
        print(json.dumps(line), flush=True)
    os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
