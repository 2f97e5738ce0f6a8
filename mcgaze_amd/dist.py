"""Multi-GPU: clips are independent (no cross-clip state: spatial attention is within a frame,
temporal attention within a clip, BN is in eval mode -- SURVEY.md section 8(e)), so the path
shards by clip with NO data-path collective.  The only exchange is one all_gather of the small
per-rank result block per step: gaze [4,N,3] + boxes [N,3,4] + scores [N,3] = 27 floats/frame
(756 B/clip at T=7), fused into ONE buffer and ONE collective (RCCL over xGMI on the GPU box,
gloo in the CPU tests).  The engine writes its outputs directly into the fused buffer's views.
"""
import torch

FLOATS_PER_FRAME = 4 * 3 + 3 * 4 + 3


def shard_clips(num_clips, world, rank):
    """Contiguous block of clips for ``rank`` (sizes differ by at most one)."""
    base, rem = divmod(num_clips, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _views(buf, n):
    return dict(gaze=buf[:12 * n].view(4, n, 3), boxes=buf[12 * n:24 * n].view(n, 3, 4), scores=buf[24 * n:27 * n].view(n, 3))


class ResultGather:
    """Fused result buffer for ``frames_per_rank`` frames per rank + its all_gather."""

    def __init__(self, frames_per_rank, world, device):
        self.n, self.world = frames_per_rank, world
        self.local = torch.zeros(FLOATS_PER_FRAME * frames_per_rank, dtype=torch.float32, device=device)
        self.all = torch.zeros(world * FLOATS_PER_FRAME * frames_per_rank, dtype=torch.float32, device=device) if world > 1 else self.local

    def all_single(self):
        if not hasattr(self, '_single'):
            self._single = torch.zeros_like(self.local)
        return self._single

    def local_views(self):
        return _views(self.local, self.n)

    def all_gather(self, group=None):
        import torch.distributed as dist
        if self.world == 1:
            if dist.is_available() and dist.is_initialized():   # single-rank group: still a real collective call (self-copy)
                dist.all_gather_into_tensor(self.all_single(), self.local, group=group)
            return self.all
        dist.all_gather_into_tensor(self.all, self.local, group=group)
        return self.all

    def rank_views(self, rank):
        per = FLOATS_PER_FRAME * self.n
        return _views(self.all[rank * per:(rank + 1) * per], self.n)

    def merged(self):
        """Results of all ranks concatenated in rank (= clip) order."""
        parts = [self.rank_views(r) for r in range(self.world)]
        return dict(gaze=torch.cat([p['gaze'] for p in parts], dim=1), boxes=torch.cat([p['boxes'] for p in parts]),
                    scores=torch.cat([p['scores'] for p in parts]))


# ------------------------------------------------------------------------------------ dataset runs (tools/test_gaze360_gaze.py)
def shard_videos(videos, world, rank):
    """Whole videos per rank (the overlap merge needs all windows of a video on one rank), balanced by frame
    count: longest video first onto the least-loaded rank.  Returns this rank's video indices in annotation order."""
    order = sorted(range(len(videos)), key=lambda i: -len(videos[i]['file_names']))
    load, mine = [0] * world, []
    for i in order:
        r = min(range(world), key=lambda k: load[k])
        load[r] += len(videos[i]['file_names'])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def gather_records(idx, recs, num_videos, group=None):
    """The dataset run's only exchange (the reference's only gather is mmdet/apis/test.py:179-209, collect_results): every rank
    contributes its (video index, record) pairs; every rank gets the records of ALL videos back in annotation order.  Raises if a
    video is missing or was produced twice."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, list(zip(idx, recs)), group=group)
    merged = {}
    for part in parts:
        for i, rec in part:
            if i in merged:
                raise RuntimeError(f'video {i} was processed by more than one rank')
            merged[i] = rec
    missing = [i for i in range(num_videos) if i not in merged]
    if missing:
        raise RuntimeError(f'no rank processed videos {missing[:8]}')
    return [merged[i] for i in range(num_videos)]
