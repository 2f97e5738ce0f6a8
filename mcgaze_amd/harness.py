"""Caller of the hot path: the reference's inference harness logic --
sliding 7-frame windows with stride 4 over every video, one forward per window, overlap merging, result
records -- restated from tools/test_gaze360_gaze.py:60-269 so that whole videos can be pushed through the
engine in large batches (the reference runs one clip per forward).

Semantics kept exactly (pinned by tests/golden/harness_merge.json, produced by running the reference's own
``main`` on a seeded fake model):
  * windows (:72-86): ``clip_num = ceil((L-7)/4)+1``; the last window is the LAST 7 frames; a video with
    L <= 7 frames is one short clip of T = L frames (so T is variable);
  * merge (:129-206): the first window is taken verbatim; later windows append their new frames and average the
    overlapping ones ``(old+new)/2`` -- gaze vectors are NOT re-normalised; box coordinates are zeroed where the
    score is < 0.5 in either prediction (the stored score may itself be an average);
  * record schema (:210-260): per frame fused gaze, per-clue boxes as [x, y, w, h] or None when zeroed, gazes, scores.
"""
import collections
import json
import math
import os

import numpy as np
import torch

CLUES = ('face', 'eyes', 'head')
last_run_stats = {}       # run_annotation's frame-cache counters of the last call (tools/dataset_throughput.py prints them)


def plan_windows(video_length, clip_len=7, stride=4):
    """-> list of (start, stop, overlap_with_previous).  tools/test_gaze360_gaze.py:72-86."""
    L = video_length
    if L <= clip_len:
        return [(0, L, 0)]
    clip_num = math.ceil((L - clip_len) / stride) + 1
    out = []
    for i in range(clip_num):
        if i != clip_num - 1:
            out.append((i * stride, i * stride + clip_len, clip_len - stride))
        else:
            rem = (L - clip_len) % stride
            out.append((L - clip_len, L, clip_len - rem if rem else clip_len - stride))
    return out


def _host(t):
    """torch tensor (any device) or array -> numpy array on the host (a view where possible)."""
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def merge_video(windows, clip_outputs, person_threshold=0.5):
    """windows from plan_windows; clip_outputs[i] = (det_bboxes [T,3,5], fused [T,3], others [T,3,3]) of window i (torch tensors or
    numpy arrays, f32).  Returns per-frame numpy arrays (det_bboxes [L,3,5], fused [L,3], others [L,3,3]).
    tools/test_gaze360_gaze.py:129-206.  Host arithmetic in f32 -- sums and halvings, exactly rounded, the same bits the reference's
    torch expressions give; the windows of a video tile it from frame 0 with the last one flush to the end, so frame t of window
    (start, stop) lands at row start + t."""
    L = windows[-1][1]
    det, fused, others = np.empty((L, 3, 5), np.float32), np.empty((L, 3), np.float32), np.empty((L, 3, 3), np.float32)
    thr = np.float32(person_threshold)
    half = np.float32(2)
    for i, ((start, stop, overlap), (d, f, o)) in enumerate(zip(windows, clip_outputs)):
        d, f, o = _host(d), _host(f), _host(o)
        low = d[..., 4:] < thr
        d = np.concatenate([np.where(low, np.float32(0), d[..., :4]), d[..., 4:]], axis=-1)      # boxes of low-score detections are zeroed (:140-150)
        if i == 0:
            det[:stop], fused[:stop], others[:stop] = d, f, o
            continue
        # overlapping frames: the last `overlap` stored frames against the first `overlap` frames of this window
        ov = slice(start, start + overlap)
        old_d, cur_d = det[ov], d[:overlap]
        bad = (old_d[..., 4:] < thr) | low[:overlap]
        det[ov, :, :4] = np.where(bad, np.float32(0), (old_d[..., :4] + cur_d[..., :4]) / half)
        det[ov, :, 4:] = (old_d[..., 4:] + cur_d[..., 4:]) / half
        fused[ov] = (fused[ov] + f[:overlap]) / half
        others[ov] = (others[ov] + o[:overlap]) / half
        new = slice(start + overlap, stop)
        det[new], fused[new], others[new] = d[overlap:], f[overlap:], o[overlap:]
    return det, fused, others


def video_record(video_id, det, fused, others):
    """Result record of one video (tools/test_gaze360_gaze.py:210-260): python floats of the f32 values; a box is [x, y, w, h] with
    the differences taken in double (the reference subtracts python floats), or None where the four coordinates sum to zero."""
    det, fused, others = _host(det), _host(fused), _host(others)
    rec = dict(video_id=video_id, category_id=1, fusion_gazes=fused.tolist())
    m = det[..., :4].astype(np.float64)
    empty = (((m[..., 0] + m[..., 1]) + m[..., 2]) + m[..., 3] == 0).tolist()
    xywh = np.stack([m[..., 0], m[..., 1], m[..., 2] - m[..., 0], m[..., 3] - m[..., 1]], axis=-1).tolist()
    gazes, score = others.tolist(), det[..., 4].tolist()
    for ci, c in enumerate(CLUES):
        rec[f'{c}_bboxes'] = [None if e[ci] else b[ci] for e, b in zip(empty, xywh)]
        rec[f'{c}_gazes'] = [g[ci] for g in gazes]
        rec[f'{c}_score'] = [sc[ci] for sc in score]
    return rec


def result_file_name(config_path, json_path):
    """``results_<config stem>_<annotation file name>`` (tools/test_gaze360_gaze.py:268; note rstrip('.py') semantics)."""
    return f'results_{config_path.rstrip(".py").split("/")[-1]}_{json_path.split("/")[-1]}'


def _run_windows(engine, ids, plans, get_window, batch_clips, person_threshold):
    """Core of run_videos / run_annotation, STREAMING: windows are visited in (video, window) order -- the reference's order, so the
    crop RNG draws inside ``get_window`` fall where upstream's do -- and dropped into per-(T, H, W) buckets; a bucket runs through
    the engine (batched semantics: N = B*T frames, clip_length = T) as soon as it holds ``batch_clips`` clips and its inputs are
    released; a video is merged and turned into its record as soon as its last window has come back.  Device memory is bounded by
    ``batch_clips`` clips per distinct window shape, not by the size of the dataset.
    get_window(vi, wi) -> (frames [T,3,H,W] f32, img_hw [T,2] int or None, scale [T,4] f32 or None)."""
    dev = engine.device
    buckets = {}                                       # (T, H, W) -> list of (vi, wi, frames, hw, scale)
    outputs = [[None] * len(p) for p in plans]
    pending = [len(p) for p in plans]
    records = [None] * len(plans)

    keep = []                                          # pinned host tensors of the last flush: alive until their non-blocking copies ran
    inflight = collections.deque()                     # batches whose results are on their way to the host: (items, T, pinned buffer, event)
    spare = []                                         # pinned result buffers free for reuse

    def upload(t):
        # small per-batch tables (img_shape, scale_factor): pinned + non-blocking, so that the host does not wait for the device queue to
        # drain (a pageable .to() did: the host then idled for the previous batch's forward instead of preparing the next)
        if dev.type != 'cuda':
            return t.to(dev)
        p = t.pin_memory()
        keep.append(p)
        return p.to(dev, non_blocking=True)

    def collect(leave):
        # results of finished batches -> per-window host arrays -> (once a video's last window is back) merge + record, all on the host:
        # the consumer never waits for a forward it has just queued (round 3 / 4 did, in video_record's .cpu(): host and device took turns)
        while len(inflight) > leave or (inflight and (inflight[0][3] is None or inflight[0][3].query())):
            items, T, buf, ev = inflight.popleft()
            if ev is not None:
                ev.synchronize()
            res = buf[:len(items) * T].numpy().copy()  # [frames, 27] = det 3x5 | fused 3 | others 3x3; copied: the pinned buffer is reused
            if ev is not None:
                spare.append(buf)
            for bi, (vi, wi) in enumerate(items):
                r = res[bi * T:(bi + 1) * T]
                outputs[vi][wi] = (r[:, :15].reshape(T, 3, 5), r[:, 15:18], r[:, 18:].reshape(T, 3, 3))
                pending[vi] -= 1
                if pending[vi] == 0:                   # all windows of the video are back: merge, record, release
                    records[vi] = video_record(ids[vi], *merge_video(plans[vi], outputs[vi], person_threshold))
                    outputs[vi] = None

    def flush(key):
        items = buckets.pop(key, [])
        if not items:
            return
        T = key[0]
        collect(2)
        x = torch.cat([it[2] for it in items]).to(dev, torch.float32).contiguous()
        del keep[:max(0, len(keep) - 4)]
        hw = None if items[0][3] is None else upload(torch.cat([torch.as_tensor(it[3], dtype=torch.int32).reshape(-1, 2) for it in items]))
        out = engine.forward(x, T, img_hw=hw)
        boxes = out['boxes']
        if items[0][4] is not None:   # rescale=True: every frame's boxes by its own scale_factor (multiclue_gaze_roi_head.py:360-363)
            boxes = boxes / upload(torch.cat([torch.as_tensor(it[4], dtype=torch.float32) for it in items]))[:, None, :]
        n = len(items) * T
        gaze = out['gaze']
        packed = torch.cat([boxes.reshape(n, 3, 4), out['scores'].reshape(n, 3, 1)], dim=-1).reshape(n, 15)
        packed = torch.cat([packed, gaze[0].reshape(n, 3), gaze[1:].permute(1, 0, 2).reshape(n, 9)], dim=1).to(torch.float32)
        who = [(vi, wi) for vi, wi, _, _, _ in items]
        if dev.type != 'cuda':
            inflight.append((who, T, packed, None))
            return
        buf = next((b for b in spare if b.shape[0] >= n), None)
        if buf is not None:
            spare.remove(buf)
        else:
            buf = torch.empty(max(n, batch_clips * T), 27, dtype=torch.float32).pin_memory()
        buf[:n].copy_(packed, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(dev))
        inflight.append((who, T, buf, ev))

    for vi, plan in enumerate(plans):
        for wi in range(len(plan)):
            x, hw, sc = get_window(vi, wi)
            key = (plan[wi][1] - plan[wi][0], x.shape[-2], x.shape[-1])
            buckets.setdefault(key, []).append((vi, wi, x, hw, sc))
            if len(buckets[key]) >= batch_clips:
                flush(key)
    for key in sorted(buckets):
        flush(key)
    collect(0)
    return records


def run_videos(engine, videos, clip_len=7, stride=4, batch_clips=64, scale_factor=None, person_threshold=0.5):
    """Push whole videos through the HIP engine.

    videos: list of dict(id=…, frames=Tensor[L,3,H,W] f32 already preprocessed (normalised, padded to /32)[, img_hw=[L,2] int: the
    per-frame img_shape inside the padded frame -- all videos or none]).
    scale_factor (4 floats) divides the boxes like rescale=True does (multiclue_gaze_roi_head.py:360-363)."""
    plans = [plan_windows(v['frames'].shape[0], clip_len, stride) for v in videos]

    def get_window(vi, wi):
        a, b, _ = plans[vi][wi]
        sc = None if scale_factor is None else torch.as_tensor(scale_factor, dtype=torch.float32).expand(b - a, 4)
        hw = videos[vi].get('img_hw')
        return videos[vi]['frames'][a:b], (None if hw is None else hw[a:b]), sc

    return _run_windows(engine, [v['id'] for v in videos], plans, get_window, batch_clips, person_threshold)


def run_annotation(engine, anno, root, pipeline, clip_len=7, stride=4, batch_clips=64, person_threshold=0.5, rng=None, workers=0, lookahead=None,
                   processes=False, video_rng=None):
    """tools/test_gaze360_gaze.py:57-269 from the annotation file down: for every ``anno['videos']`` entry (``id``,
    ``file_names``) each window's frames are loaded and preprocessed ANEW through ``pipeline`` (a
    mcgaze_amd.pipeline.DevicePipeline built from cfg.data.test.pipeline) -- like the reference, which re-runs its test
    pipeline, random crop included, for every window (:88-100) -- then all windows go through the engine in large batches
    and are merged per video.  Frames of a window are drawn in file-name order (the reference sorts its threads' results by
    file name, :96); ``rng`` seeds the crop draws (default: the global numpy RNG, as upstream).
    Each file is decoded once while it stays cached (windows overlap by three frames); windows are preprocessed in groups (one
    pinned copy and one launch per padded size).  ``workers`` > 0 adds host threads that decode ``lookahead`` windows ahead of the
    consumer (the reference uses 7 loader threads per window, :88-95) -- measured slower than in-line decoding for small frames
    (pipeline.FrameCache), hence 0 by default; ``processes=True`` makes them helper PROCESSES that decode into a shared-memory ring (no
    interpreter lock in common with this loop).  The records do not depend on any of it.
    ``video_rng``: video id -> numpy RandomState.  With it every video draws its crops from its OWN generator (created when its first
    window is planned) instead of the shared ``rng``: the records then do not depend on which videos a process was given or in what
    order -- what a run sharded over several consumer processes needs to reproduce a single-process run record for record (the
    reference's single global generator makes its results depend on the shard layout)."""
    import numpy as np
    import os
    from .pipeline import FrameCache
    rng = np.random if rng is None else rng
    videos = anno['videos']
    plans = [plan_windows(len(v['file_names']), clip_len, stride) for v in videos]
    order = [(vi, wi) for vi, plan in enumerate(plans) for wi in range(len(plan))]     # the order _run_windows visits windows in
    pos = {vw: i for i, vw in enumerate(order)}
    lookahead = 2 * batch_clips if lookahead is None else lookahead
    group = max(1, min(batch_clips, 32))                                    # windows staged per preprocessing call
    frame_bytes = None
    if workers > 0 and processes and videos and videos[0]['file_names']:
        # what a ring slot will really hold: the run's first frame, from its header (no decode) -- the /dev/shm check of the decode ring prices
        # a slot at that instead of the 3 MiB worst case (a 360 x 360 frame is 390 KB)
        try:
            from PIL import Image
            with Image.open(os.path.join(img_prefix, videos[0]['file_names'][0])) as im:
                frame_bytes = 2 * im.size[0] * im.size[1] * 3        # x 2: frames of a dataset vary
        except Exception:
            frame_bytes = None
    cache = FrameCache(workers, capacity=((lookahead if workers > 0 else 0) + group + 2) * clip_len, processes=processes, frame_bytes=frame_bytes)   # in-line decoding never runs ahead: only the staged group (and the 3-frame overlap) is worth keeping
    state = dict(ahead=0)

    def names_of(vi, wi):
        a, b, _ = plans[vi][wi]
        return sorted(videos[vi]['file_names'][a:b])

    staged = {}
    per_video = {}

    def rng_of(vi):
        if video_rng is None:
            return rng
        if vi not in per_video:
            per_video[vi] = video_rng(videos[vi]['id'])
        return per_video[vi]

    def get_window(vi, wi):
        i = pos[(vi, wi)]
        if i not in staged:
            # stage the next `group` windows in one go (one pinned copy, one launch per padded size); planned in visiting order, so
            # the crop draws fall exactly where they do one window at a time
            todo = order[i:i + group]
            if workers > 0:
                while state['ahead'] < min(i + len(todo) + lookahead, len(order)):   # decoders stay `lookahead` windows ahead
                    v2, w2 = order[state['ahead']]
                    cache.prefetch(os.path.join(root, n) if root is not None else n for n in names_of(v2, w2))
                    state['ahead'] += 1
            res = pipeline.run_many([names_of(v2, w2) for v2, w2 in todo], device=engine.device, rng=[rng_of(v2) for v2, _ in todo], img_prefix=root, loader=cache)
            staged.update({i + k: r for k, r in enumerate(res)})
        img, metas = staged.pop(i)
        hw = [m['img_shape'][:2] for m in metas]
        return img, hw, np.stack([m['scale_factor'] for m in metas])

    try:
        return _run_windows(engine, [v['id'] for v in videos], plans, get_window, batch_clips, person_threshold)
    finally:
        last_run_stats.update(decodes=cache.decodes, decode_waits=cache.waits, decode_wait_s=round(cache.wait_s, 3), decode_first_wait_s=round(cache.first_wait_s, 3))
        cache.close()


def dump_results(records, config_path, json_path, out_dir='results'):
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, result_file_name(config_path, json_path))
    with open(path, 'w') as f:
        json.dump(records, f)
    return path
