"""world_size-2 gloo test of the only multi-GPU exchange on the path (mcgaze_amd/dist.py)."""
import os
import pytest
import socket

import torch
import torch.multiprocessing as mp

from mcgaze_amd.dist import FLOATS_PER_FRAME, ResultGather, shard_clips


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, ret):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    g = ResultGather(n, world, torch.device('cpu'))
    v = g.local_views()
    v['gaze'].copy_(torch.full((4, n, 3), float(rank + 1)))
    v['boxes'].copy_(torch.arange(n * 12, dtype=torch.float32).view(n, 3, 4) + 1000 * rank)
    v['scores'].copy_(torch.full((n, 3), 0.5 + rank))
    g.all_gather()
    m = g.merged()
    ok = m['gaze'].shape == (4, world * n, 3) and m['boxes'].shape == (world * n, 3, 4)
    for r in range(world):
        ok &= bool((m['gaze'][:, r * n:(r + 1) * n] == r + 1).all())
        ok &= bool((m['boxes'][r * n:(r + 1) * n] == torch.arange(n * 12, dtype=torch.float32).view(n, 3, 4) + 1000 * r).all())
        ok &= bool((m['scores'][r * n:(r + 1) * n] == 0.5 + r).all())
    ret[rank] = ok
    dist.barrier()
    dist.destroy_process_group()


def test_fused_all_gather_world2():
    world, n = 2, 14
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), n, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_shard_clips_partitions_exactly():
    for clips in (1, 7, 64, 512, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_clips(clips, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == clips
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_views_alias_the_fused_buffer():
    g = ResultGather(5, 1, torch.device('cpu'))
    v = g.local_views()
    v['scores'].fill_(2.0)
    assert g.local.numel() == FLOATS_PER_FRAME * 5 and float(g.local[-1]) == 2.0 and float(g.local[0]) == 0.0


# ------------------------------------------------------------------------------------ dataset run: video sharding + record gather
class _FakeEngine:
    """Stands in for HipEngine on the CPU: deterministic outputs that depend on the frames only, so a sharded run must reproduce a
    single-process run record for record.  (Host logic under test: window planning, bucketed streaming, merge, shard, gather.)"""
    device = torch.device('cpu')

    def forward(self, x, T, img_hw=None, **kw):
        N = x.shape[0]
        m = torch.stack([f.double().sum() for f in x]).float() / x[0].numel()   # per frame, independent of the batch it sits in
        gaze = torch.stack([torch.stack([torch.sin(m + k), torch.cos(m * (c + 1)), -torch.ones_like(m)], dim=-1) for k in range(4) for c in [k]])
        boxes = (m[:, None, None] * torch.arange(1, 13, dtype=torch.float32).view(1, 3, 4)).abs() + 1
        scores = torch.sigmoid(m)[:, None].expand(N, 3) * torch.tensor([1.0, 0.9, 0.4])
        return dict(gaze=gaze, boxes=boxes, scores=scores.contiguous())


def _fake_videos():
    lengths = [3, 7, 8, 30, 11, 19, 7, 45, 12]
    vids = []
    for i, L in enumerate(lengths):
        g = torch.Generator().manual_seed(50 + i)
        vids.append(dict(id=100 + i, file_names=[f'v{i}/{t:04d}.png' for t in range(L)], frames=torch.randn(L, 3, 32, 32, generator=g)))
    return vids


def _dataset_worker(rank, world, port, ret):
    import torch.distributed as dist
    from mcgaze_amd import harness
    from mcgaze_amd.dist import gather_records, shard_videos
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    vids = _fake_videos()
    idx = shard_videos(vids, world, rank)
    recs = harness.run_videos(_FakeEngine(), [vids[i] for i in idx], batch_clips=5)
    allrecs = gather_records(idx, recs, len(vids))
    ret[rank] = (idx, allrecs)
    dist.barrier()
    dist.destroy_process_group()


def test_dataset_run_shards_videos_and_gathers_records_in_annotation_order():
    """tools/test_gaze360_gaze.py under 2 ranks (gloo): whole videos per rank balanced by frame count, every rank streams its
    windows through the engine in small batches, one object gather, records back in annotation order and identical to a
    single-process run."""
    from mcgaze_amd import harness
    from mcgaze_amd.dist import shard_videos
    world = 2
    vids = _fake_videos()
    single = harness.run_videos(_FakeEngine(), vids, batch_clips=64)
    ret = mp.Manager().dict()
    mp.spawn(_dataset_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    shards = [ret[r][0] for r in range(world)]
    assert sorted(shards[0] + shards[1]) == list(range(len(vids))) and not set(shards[0]) & set(shards[1])
    loads = [sum(len(vids[i]['file_names']) for i in sh) for sh in shards]
    assert abs(loads[0] - loads[1]) <= max(len(v['file_names']) for v in vids) // 2 + 1, loads
    for r in range(world):
        got = ret[r][1]
        assert [g['video_id'] for g in got] == [v['id'] for v in vids]
        assert got == single
    for w in (1, 3, 8):   # every video exactly once for other world sizes too
        parts = [shard_videos(vids, w, r) for r in range(w)]
        assert sorted(i for p in parts for i in p) == list(range(len(vids)))


# ------------------------------------------------------------------------------------ bench.py's N > 1 branch, executed
@pytest.mark.parametrize('world', [2, 4])
def test_bench_control_flow_with_the_fake_engine(world):
    """bench.main() under `torch.distributed.run --nproc-per-node N` (the driver's launch line) with the host
    stand-in engine and gloo -- clip sharding by rank, the fused exchange, verify_ring_neighbour's rank_views branch, the strong-scaling leg
    and the stdout hand-over all execute here before an 8-GPU node ever runs them; N = 4 so that an off-by-one in the ring neighbour
    ((r + 1) % N) or in rank_views cannot hide behind N = 2, where the neighbour is simply "the other one".  The numbers are meaningless; the
    SHAPE of the line is what is asserted: one JSON line, last on stdout, world_size N, every rank's neighbour check passed, strong_scaling
    present with the fixed batch divided by N."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', str(world), '--fake-engine', '--steps', '3', '--warmup', '1',
           '--clips-per-gpu', '2', '--size', '32', '--strong-clips', '8']
    if world == 2:   # round 6: the driver's default line times three engines per rank (headline, throughput_engine, other_engines): every rank must walk them in step
        cmd += ['--second-engine', 'f16,bf16']
    r = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])                                   # nothing after the JSON line
    assert sum(1 for ln in lines if ln.lstrip().startswith('{"metric"')) == 1
    assert line['n_gpus'] == world and line['world_size'] == world and line['scaling'] == 'weak'
    assert line['rccl_ranks_verified'] == world                    # every rank matched its ring neighbour's block bit for bit
    assert line['verified'] is True
    st = line['strong_scaling']
    assert st['global_clips'] == 8 and st['clips_per_gpu'] == 8 // world and st['scaling'] == 'strong' and st['value'] > 0
    assert line['config']['global_clips'] == 2 * world and 'FAKE ENGINE' in line['data']
    if world == 2:
        assert line['throughput_engine']['dtype'] == 'f16' and line['throughput_engine']['rccl_ranks_verified'] == world
        assert list(line['other_engines']) == ['bf16'] and line['other_engines']['bf16']['rccl_ranks_verified'] == world
    else:
        assert 'throughput_engine' not in line
