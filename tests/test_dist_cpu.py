"""world_size-2 gloo test of the only multi-GPU exchange on the path (mcgaze_amd/dist.py)."""
import os
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
