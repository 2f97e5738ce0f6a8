"""Host-side driver of ``libmcgaze_hip.so``: owns the packed weights, the workspace and the
C engine handle, and exposes the path (and its individual operators, for the parity tests)
over torch device tensors.  PyTorch is used for device memory and streams only; all
arithmetic runs in the hand-written HIP kernels behind the C-ABI.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import lib as L
from .packing import PackedWeights

# precision -> (storage dtype of activations, mcg_dtype code).  'f16x3': f32 storage, split-fp16 x 3 MFMA contraction (parity-grade
# fast mode, include/mcgaze_hip.h MCG_F16X3)
# 'f16' (round 6): the 16-bit throughput mode in fp16 -- bf16's kernels and layouts with 11 significant bits instead of 8 (MCG_F16)
_TORCH_DT = {'bf16': torch.bfloat16, 'f16': torch.float16, 'fp16': torch.float16, 'fp32': torch.float32, 'f32': torch.float32, 'f16x3': torch.float32, 'bf16x3': torch.float32}
_CODE = {'bf16': L.MCG_BF16, 'f16': L.MCG_F16, 'fp16': L.MCG_F16, 'fp32': L.MCG_F32, 'f32': L.MCG_F32, 'f16x3': L.MCG_F16X3,
         'bf16x3': L.MCG_F16X3}   # 'bf16x3': the engine's name while its halves were bf16 -- accepted, runs the fp16-halves engine


def _code(dtype):
    return L.MCG_BF16 if dtype == torch.bfloat16 else (L.MCG_F16 if dtype == torch.float16 else L.MCG_F32)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device=None):
    """The caller's current HIP stream on ``device`` (default: the current device)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _require_gpu():
    if not torch.cuda.is_available():
        raise L.McgError('mcgaze_amd needs a HIP device (MI355X / gfx950); there is no CPU fallback path')


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------ operators
def to_nhwc(x_nchw, dtype):
    """[N,C,H,W] f32 device tensor -> NHWC ``dtype`` via mcg_nchw_to_nhwc."""
    _require_gpu()
    lib = L.load()
    N, Cc, H, W = x_nchw.shape
    x = x_nchw.contiguous().float()
    out = torch.empty(N, H, W, Cc, dtype=dtype, device=x.device)
    L.check(lib.mcg_nchw_to_nhwc(_stream(), _code(dtype), _ptr(x), _ptr(out), N, Cc, H, W), 'mcg_nchw_to_nhwc')
    return out


def to_nchw(x_nhwc):
    _require_gpu()
    lib = L.load()
    N, H, W, Cc = x_nhwc.shape
    out = torch.empty(N, Cc, H, W, dtype=torch.float32, device=x_nhwc.device)
    L.check(lib.mcg_nhwc_to_nchw(_stream(), _code(x_nhwc.dtype), _ptr(x_nhwc.contiguous()), _ptr(out), N, Cc, H, W), 'mcg_nhwc_to_nchw')
    return out


def conv2d(x, w, bias=None, stride=1, pad=0, relu=False, residual=None, residual_mode=L.RES_NONE, x2=None, stride2=1, split=False,
           tile=0, flags=0, prescale=True):
    """x NHWC, w OHWI (same dtype), bias f32 -> NHWC.  mcg_conv2d.  With x2: w = [Cout,1,1,Cin+Cin2], x2 sampled at stride2.
    split=True: the MCG_F16X3 contraction -- x / residual f32, w given as f32 OHWI and split-packed here (packing.split_pack), pre-scaled
    by a power of two unless prescale=False (packing.pow2_prescale; mcg_conv_desc.wscale).
    tile / flags: mcg_conv_desc.tile (force a contraction tile) and MCG_FLAG_* (lib.FLAG_*)."""
    _require_gpu()
    lib = L.load()
    N, H, W, Cin = x.shape
    Cout, KH, KW, _ = w.shape
    wscale = 0.0
    if split:
        from .packing import pow2_prescale, split_pack
        assert x.dtype == torch.float32 and w.dtype == torch.float32
        ws, wscale = pow2_prescale(w.reshape(Cout, -1).cpu()) if prescale else (w.reshape(Cout, -1).cpu(), 0.0)
        w = split_pack(ws).to(x.device)
    Cin2 = x2.shape[3] if x2 is not None else 0
    Ho, Wo = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    y = torch.empty(N, Ho, Wo, Cout, dtype=x.dtype, device=x.device)
    d = L.ConvDesc(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                   residual.data_ptr() if residual is not None else None, y.data_ptr(),
                   N, H, W, Cin, Cout, KH, KW, stride, pad, int(relu), residual_mode if residual is not None else L.RES_NONE,
                   residual.shape[1] if residual is not None else 0, residual.shape[2] if residual is not None else 0,
                   x2.data_ptr() if x2 is not None else None, Cin2, stride2, x2.shape[1] if x2 is not None else 0,
                   x2.shape[2] if x2 is not None else 0, tile, flags, wscale)
    L.check(lib.mcg_conv2d(_stream(), L.MCG_F16X3 if split else _code(x.dtype), C.byref(d)), 'mcg_conv2d')
    return y


def conv3x3_wino(x, w, bias=None, relu=False, tile=0, g=2):
    """3x3 / stride 1 / pad 1 conv by the 1-D Winograd F(2,3) kernel of the MCG_F16X3 engine (mcg_conv3x3_wino_x3, wino_x3.hpp):
    x NHWC f32, w OHWI f32 [Cout,3,3,Cin] (packed here by packing.wino_pack), bias f32 -> NHWC f32.  tile: 0 = by grid size, 1..4 forced
    (4 = the one-wave-per-SIMD 128 x 128 tile).
    g = 4: the F(4,3) form of the same kernel (maps whose width is a multiple of 4, at least 16)."""
    _require_gpu()
    lib = L.load()
    from .packing import pow2_prescale, wino_pack
    N, H, W, Cin = x.shape
    Cout = w.shape[0]
    ws, wscale = pow2_prescale(w.cpu())
    u = wino_pack(ws, g=g).to(x.device)
    assert u.numel() * 2 == lib.mcg_conv3x3_wino_x3_weight_bytes(Cin, Cout, g), (u.numel(), Cin, Cout)
    y = torch.empty(N, H, W, Cout, dtype=torch.float32, device=x.device)
    L.check(lib.mcg_conv3x3_wino_x3(_stream(), _ptr(x.contiguous()), _ptr(u), _ptr(bias), _ptr(y), N, H, W, Cin, Cout, int(relu), tile, wscale, g), 'mcg_conv3x3_wino_x3')
    return y


def bottleneck_x3(x, src2, wstream, bias, cn, nsrc, trace=None):
    """The fused bottleneck tail (mcg_bottleneck_x3): x [N,H,W,cm] f32 = conv2's input (cm = 64 / 128), src2 = residual [N,H,W,4 cm]
    (nsrc 1) or the downsample conv's input [N,H,W,64] (nsrc 2); wstream / bias from packing.bneck_stream.
    -> (y [N,H,W,4 cm], z [N,H,W,cn] or None)."""
    _require_gpu()
    lib = L.load()
    N, H, W, cm = x.shape
    y = torch.empty(N, H, W, 4 * cm, dtype=torch.float32, device=x.device)
    z = torch.empty(N, H, W, cn, dtype=torch.float32, device=x.device) if cn else None
    L.check(lib.mcg_bottleneck_x3(_stream(), _ptr(x.contiguous()), _ptr(src2.contiguous()), _ptr(wstream), _ptr(bias), _ptr(y), _ptr(z),
                                  N, H, W, cm, nsrc, cn, _ptr(trace)), 'mcg_bottleneck_x3')
    return y, z


def stem(img, w_stem, bias, dtype, flags=0, split=False):
    """img [N,3,H,W] f32 -> [N,H/4,W/4,64] NHWC.  mcg_stem_forward."""
    _require_gpu()
    lib = L.load()
    N, _, H, W = img.shape
    code = L.MCG_F16X3 if split else _code(dtype)
    ws = _ws(lib.mcg_stem_workspace_bytes(code, N, H, W), img.device)
    y = torch.empty(N, H // 4, W // 4, 64, dtype=dtype, device=img.device)
    L.check(lib.mcg_stem_forward(_stream(), code, _ptr(img.contiguous()), _ptr(w_stem), _ptr(bias), _ptr(y), N, H, W,
                                 _ptr(ws), ws.numel(), flags), 'mcg_stem_forward')
    return y


def roi_align(feats, boxes, strides=(4, 8, 16, 32)):
    """feats: 4 NHWC levels; boxes [N,P,4] f32 -> ([N*P,49,C], levels int32 [N*P]).  mcg_roi_align."""
    _require_gpu()
    lib = L.load()
    N, P = boxes.shape[:2]
    Cc = feats[0].shape[-1]
    out = torch.empty(N * P, 49, Cc, dtype=feats[0].dtype, device=boxes.device)
    lv = torch.empty(N * P, dtype=torch.int32, device=boxes.device)
    fp = (C.c_void_p * 4)(*[f.data_ptr() for f in feats])
    fh = (C.c_int * 4)(*[f.shape[1] for f in feats])
    fw = (C.c_int * 4)(*[f.shape[2] for f in feats])
    st = (C.c_int * 4)(*strides)
    b = boxes.contiguous().float()
    L.check(lib.mcg_roi_align(_stream(), _code(feats[0].dtype), fp, fh, fw, st, Cc, _ptr(b), N * P, P, _ptr(out), _ptr(lv)), 'mcg_roi_align')
    return out, lv


def _table(d, keys):
    return (C.c_void_p * len(keys))(*[d[k].data_ptr() for k in keys])


def stage_forward(stage_w, roi_feat, obj, boxes, clip_length, stds=(0.5, 0.5, 1.0, 1.0), split=False, flags=0):
    """One decoder stage.  roi_feat [R,49,256], obj [N,3,256], boxes [N,3,4] f32
    -> (obj' [N,3,256], boxes' [N,3,4], cls logits [N,3]).  mcg_stage_forward."""
    _require_gpu()
    lib = L.load()
    N = obj.shape[0]
    dt = L.MCG_F16X3 if split else _code(obj.dtype)   # split: stage_w from PackedWeights(split=True), f32 activations
    ws = _ws(lib.mcg_stage_workspace_bytes(dt, N), obj.device)
    obj_out = torch.empty_like(obj)
    boxes_out = torch.empty(N, 3, 4, dtype=torch.float32, device=obj.device)
    cls = torch.empty(N, 3, dtype=torch.float32, device=obj.device)
    sd = (C.c_float * 4)(*stds)
    L.check(lib.mcg_stage_forward(_stream(), dt, _table(stage_w, L.STAGE_KEYS), _ptr(roi_feat.contiguous()), _ptr(obj.contiguous()),
                                  _ptr(boxes.contiguous().float()), N, clip_length, _ptr(obj_out), _ptr(boxes_out), _ptr(cls), sd,
                                  _ptr(ws), ws.numel(), flags), 'mcg_stage_forward')
    return obj_out, boxes_out, cls


def gaze_head(gaze_w, obj, split=False):
    """obj [N,3,256] -> [4,N,3] f32 unit vectors (fused, face, eyes, head).  mcg_gaze_head."""
    _require_gpu()
    lib = L.load()
    N = obj.shape[0]
    dt = L.MCG_F16X3 if split else _code(obj.dtype)
    ws = _ws(lib.mcg_gaze_head_workspace_bytes(dt, N), obj.device)
    out = torch.empty(4, N, 3, dtype=torch.float32, device=obj.device)
    L.check(lib.mcg_gaze_head(_stream(), dt, _table(gaze_w, L.GAZE_KEYS), _ptr(obj.contiguous()), N, _ptr(out), _ptr(ws), ws.numel()), 'mcg_gaze_head')
    return out


# ------------------------------------------------------------------------------------ engine
class HipEngine:
    """The whole per-clip forward path behind one C call (mcg_clip_forward)."""

    def __init__(self, state_dict, depth=50, num_stages=4, precision='f16x3', device='cuda:0', bbox_stds=(0.5, 0.5, 1.0, 1.0), fuse_downsample=True):
        _require_gpu()
        self.lib = L.load()
        self.device = torch.device(device)
        self.dtype = _TORCH_DT[precision]
        self.code = _CODE[precision]
        self.precision = precision
        self.weights = PackedWeights(state_dict, depth=depth, num_stages=num_stages, dtype=self.dtype, device=self.device,
                                     fuse_downsample=fuse_downsample, split=self.code == L.MCG_F16X3)
        w = self.weights
        mk = lambda c: L.ConvWeights(c['w'].data_ptr(), c['bias'].data_ptr(), c['cin'], c['cout'], c['k'], c['stride'], c['pad'],
                                     c['wf'].data_ptr() if c.get('wf') is not None else None, float(c.get('wscale', 0.0)),
                                     c['wf4'].data_ptr() if c.get('wf4') is not None else None)
        self._convs = (L.ConvWeights * len(w.convs))(*[mk(c) for c in w.convs])
        self._stage_tab = (C.c_void_p * (num_stages * L.SW_COUNT))(*[st[k].data_ptr() for st in w.stages for k in L.STAGE_KEYS])
        self._gaze_tab = _table(w.gaze, L.GAZE_KEYS)
        mw = L.ModelWeights()
        mw.blocks = (C.c_int * 4)(*w.blocks)
        mw.stem = mk(w.stem)
        mw.convs = C.cast(self._convs, C.POINTER(L.ConvWeights))
        mw.num_convs = len(w.convs)
        mw.lateral = (L.ConvWeights * 4)(*[mk(c) for c in w.lateral])
        mw.fpn_out = (L.ConvWeights * 4)(*[mk(c) for c in w.fpn_out])
        if len(w.c3_ds) == 4:
            mw.c3_ds = (L.ConvWeights * 4)(*[mk(c) for c in w.c3_ds])
        mw.init_boxes = w.init_boxes.data_ptr()
        mw.init_feats = w.init_feats.data_ptr()
        mw.num_stages = num_stages
        mw.stage_weights = C.cast(self._stage_tab, C.POINTER(C.c_void_p))
        mw.gaze_weights = C.cast(self._gaze_tab, C.POINTER(C.c_void_p))
        mw.bbox_stds = (C.c_float * 4)(*bbox_stds)
        if w.fused:
            self._fused = (L.FusedBlock * len(w.fused))(*[L.FusedBlock(f['wstream'].data_ptr(), f['bias'].data_ptr(), f['conv2_index'], f['cm'], f['c'],
                                                                      f['cn'], f['nsrc']) for f in w.fused])
            mw.fused, mw.num_fused = C.cast(self._fused, C.POINTER(L.FusedBlock)), len(w.fused)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):   # the engine's side streams and events belong to THIS device
            L.check(self.lib.mcg_engine_create(C.byref(self._handle), C.byref(mw), self.code), 'mcg_engine_create')
        self._ws = None
        self._ws_key = None

    def set_option(self, name, value):
        """mcg_engine_set_option: 'trunk_streams', 'max_range_frames', 'tile', 'staged_gemm', 'conv3x3_c64', 'stem_fused',
        'decoder_chain', 'pointwise_pair', 'pointwise_stream', 'bottleneck_fused', 'bottleneck_blocked', 'winograd', 'range_audit' (include/mcgaze_hip.h)."""
        L.check(self.lib.mcg_engine_set_option(self._handle, name.encode(), int(value)), f'mcg_engine_set_option({name})')

    def range_audit(self, capacity=256):
        """Read and reset the range audit (set_option('range_audit', 1) first): [(tensor name, values with |x| > 65504, non-finite values)]
        over every frame the trunk processed since the last read.  mcg_engine_range_audit (a debug call: it synchronises the device)."""
        counts = (C.c_ulonglong * (2 * capacity))()
        names = (C.c_char_p * capacity)()
        n = C.c_int()
        L.check(self.lib.mcg_engine_range_audit(self._handle, counts, names, capacity, C.byref(n)), 'mcg_engine_range_audit')
        return [(names[i].decode(), int(counts[2 * i]), int(counts[2 * i + 1])) for i in range(n.value)]

    def profile_start(self, capacity=4096):
        L.check(self.lib.mcg_engine_profile_start(self._handle, capacity), 'mcg_engine_profile_start')

    def profile_stop(self, capacity=4096):
        """-> list of (ms, algorithmic flops, cfg id, (M, N, K), algorithmic HBM bytes) per contraction launch recorded since profile_start."""
        cnt = C.c_int()
        ms = (C.c_float * capacity)(); fl = (C.c_double * capacity)(); by = (C.c_double * capacity)()
        cf = (C.c_int * capacity)(); sh = (C.c_int * (3 * capacity))()
        L.check(self.lib.mcg_engine_profile_stop(self._handle, C.byref(cnt), ms, fl, by, cf, sh, capacity), 'mcg_engine_profile_stop')
        return [(ms[i], fl[i], cf[i], (sh[3 * i], sh[3 * i + 1], sh[3 * i + 2]), by[i]) for i in range(cnt.value)]

    def backbone_only(self, img, return_levels=False):
        """BASELINE.json configs[1] measurement: the trunk up to C5 (mcg_bench_backbone_forward).  return_levels: C2..C5 as NHWC views
        into the engine's workspace (valid until the next call; needs the batch to run as one frame range: set_option('trunk_streams', 1))."""
        self._check_img(img)
        N, _, H, W = img.shape
        with torch.cuda.device(self.device):
            ws = self._workspace(N, H, W, 0)
            L.check(self.lib.mcg_bench_backbone_forward(self._handle, _stream(self.device), _ptr(img), N, H, W, _ptr(ws), ws.numel()), 'mcg_bench_backbone_forward')
            if not return_levels:
                return None
            tab = (C.c_void_p * 4)()
            L.check(self.lib.mcg_bench_backbone_levels(self._handle, _ptr(ws), N, H, W, tab), 'mcg_bench_backbone_levels')
            es = ws.element_size() * (2 if self.dtype in (torch.bfloat16, torch.float16) else 4)
            out = []
            for i in range(4):
                shape = (N, (H // 4) >> i, (W // 4) >> i, 256 << i)
                n = shape[0] * shape[1] * shape[2] * shape[3]
                off = tab[i] - ws.data_ptr()
                out.append(ws[off:off + n * es].view(self.dtype).view(shape))
            return out

    def __del__(self):
        h = getattr(self, '_handle', None)
        if h is not None and h.value:
            self.lib.mcg_engine_destroy(h)
            self._handle = None

    def _workspace(self, N, H, W, chunk):
        key = (N, H, W, chunk)
        need = self.lib.mcg_engine_workspace_bytes(self._handle, N, H, W, chunk)
        if self._ws is None or self._ws.numel() < need or self._ws_key != key:
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = _ws(need, self.device)
            self._ws_key = key
        return self._ws

    def _check_img(self, img):
        if not (img.is_cuda and img.device == self.device and img.dtype == torch.float32 and img.is_contiguous() and img.dim() == 4):
            raise L.McgError(f'img must be a contiguous float32 [N,3,H,W] tensor on {self.device} (got {img.dtype} {tuple(img.shape)} on {img.device})')

    def backbone_fpn(self, img, chunk_frames=0):
        """img [N,3,H,W] f32 on the device -> [P2..P5] NHWC in the engine dtype."""
        self._check_img(img)
        N, _, H, W = img.shape
        with torch.cuda.device(self.device):
            ws = self._workspace(N, H, W, chunk_frames)
            pyr = [torch.empty(N, (H // 4) >> i, (W // 4) >> i, 256, dtype=self.dtype, device=self.device) for i in range(4)]
            tab = (C.c_void_p * 4)(*[p.data_ptr() for p in pyr])
            L.check(self.lib.mcg_backbone_fpn_forward(self._handle, _stream(self.device), _ptr(img), N, H, W, chunk_frames, tab, _ptr(ws), ws.numel()),
                    'mcg_backbone_fpn_forward')
        return pyr

    def forward(self, img, clip_length, img_hw=None, chunk_frames=0, out=None):
        """img [N,3,H,W] f32 (device, contiguous), N = clips*clip_length.
        Returns dict(gaze [4,N,3], boxes [N,3,4], scores [N,3]) -- f32 device tensors."""
        self._check_img(img)
        N, _, H, W = img.shape
        with torch.cuda.device(self.device):
            ws = self._workspace(N, H, W, chunk_frames)
            if out is None:
                out = dict(gaze=torch.empty(4, N, 3, dtype=torch.float32, device=self.device),
                           boxes=torch.empty(N, 3, 4, dtype=torch.float32, device=self.device),
                           scores=torch.empty(N, 3, dtype=torch.float32, device=self.device))
            hw = self.img_hw_tensor(img_hw, N)
            L.check(self.lib.mcg_clip_forward(self._handle, _stream(self.device), _ptr(img), N, clip_length, H, W, _ptr(hw), chunk_frames,
                                              _ptr(out['gaze']), _ptr(out['boxes']), _ptr(out['scores']), _ptr(ws), ws.numel()),
                    'mcg_clip_forward')
        return out

    def img_hw_tensor(self, img_hw, N):
        """img_shape (h, w) per frame -> the device int32 [N,2] tensor the C-ABI takes (None stays None = every frame fills H x W)."""
        if img_hw is None:
            return None
        if not isinstance(img_hw, torch.Tensor):
            img_hw = torch.as_tensor(np.asarray(img_hw, dtype=np.int32).reshape(N, 2))
        hw = img_hw.to(device=self.device, dtype=torch.int32).contiguous()
        if hw.numel() != 2 * N:
            raise L.McgError(f'img_hw must hold {N} (h, w) pairs, got {tuple(hw.shape)}')
        return hw


class GraphedForward:
    """Latency path (BASELINE.json configs[0], the reference harness's usage: ONE clip per forward, tools/test_gaze360_gaze.py:107-111).
    A single 7-frame clip is ~170 short launches whose GPU time is far below their launch cost, so the whole forward --
    trunk, 4 x (RoIAlign + decoder stage), gaze head -- is captured ONCE into a HIP graph for a fixed (N, H, W, clip_length) and
    replayed: one graph launch per clip.  Inputs are copied into the graph's static buffer on the caller's stream; the
    returned tensors are the graph's static outputs (valid until the next call).  Results are bit-identical to engine.forward
    (tests/test_gpu_forward.py::test_graphed_forward_is_bit_identical)."""

    def __init__(self, engine, num_frames, H, W, clip_length, with_img_hw=False):
        e = self.e = engine
        self.N, self.T = num_frames, clip_length
        dev = e.device
        with torch.cuda.device(dev):
            self.img = torch.zeros(num_frames, 3, H, W, dtype=torch.float32, device=dev)
            self.hw = torch.zeros(num_frames, 2, dtype=torch.int32, device=dev) if with_img_hw else None
            if self.hw is not None:
                self.hw[:, 0], self.hw[:, 1] = H, W
            self.out = dict(gaze=torch.zeros(4, num_frames, 3, device=dev), boxes=torch.zeros(num_frames, 3, 4, device=dev),
                            scores=torch.zeros(num_frames, 3, device=dev))
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):          # eager warm-up on the capture stream: workspace allocation, side-stream probe
                for _ in range(2):
                    e.forward(self.img, clip_length, img_hw=self.hw, out=self.out)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=side):
                e.forward(self.img, clip_length, img_hw=self.hw, out=self.out)

    def __call__(self, img, img_hw=None):
        self.e._check_img(img)
        if tuple(img.shape) != tuple(self.img.shape):
            raise L.McgError(f'GraphedForward was captured for {tuple(self.img.shape)}, got {tuple(img.shape)}')
        self.img.copy_(img, non_blocking=True)
        if img_hw is not None:
            if self.hw is None:
                raise L.McgError('GraphedForward was captured without img_hw (with_img_hw=True to enable)')
            self.hw.copy_(self.e.img_hw_tensor(img_hw, self.N), non_blocking=True)
        self.graph.replay()
        return self.out


_PIPELINE_STREAMS = {}


def pipeline_streams(device, decoder_priority=-1):
    """The two streams of the batch pipeline, created ONCE per (device, priority) and shared by every runner of the process:
    streams created later in a process's life land on hardware queues that serialise against earlier ones (engine.hip,
    StreamPool), so a second runner with fresh streams loses the overlap the first one had."""
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), decoder_priority)
    if key not in _PIPELINE_STREAMS:
        _PIPELINE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev, priority=decoder_priority))
    return _PIPELINE_STREAMS[key]


class PipelinedRunner:
    """Two-deep software pipeline over successive batches (throughput serving): the decoder of batch k
    (4 x [RoIAlign + stage] + gaze head: ~110 short, latency-bound launches) runs on a second HIP stream and
    overlaps the trunk of batch k+1, whose large contraction kernels leave CUs idle only at their tails.
    Pyramids are double-buffered; trunks serialise on stream A, decoders on stream B, ordered by events.
    Every submitted batch is fully processed once ``flush()`` returns control to the caller's stream.  A loop that owns the runner
    should submit from ``runner.sa`` itself (``with torch.cuda.stream(runner.sa): ...``): the caller-stream -> trunk-stream hand-over
    of each submit is then no cross-queue dependency (2 ms per pipeline fill + drain when the queues are idle; bench.py does this)."""

    def __init__(self, engine, num_frames, H, W, clip_length, chunk_frames=0, decoder_priority=-1):
        self.e, self.N, self.H, self.W, self.T, self.chunk = engine, num_frames, H, W, clip_length, chunk_frames
        dev, lib, h = engine.device, engine.lib, engine._handle
        # the decoder's short launches get the high-priority queue so they slot in between the trunk's long kernels
        self.sa, self.sb = pipeline_streams(dev, decoder_priority)
        self.pyr = [[torch.empty(num_frames, (H // 4) >> i, (W // 4) >> i, 256, dtype=engine.dtype, device=dev) for i in range(4)] for _ in range(2)]
        self.tabs = [(C.c_void_p * 4)(*[p.data_ptr() for p in lvl]) for lvl in self.pyr]
        self.trunk_ws = _ws(lib.mcg_trunk_workspace_bytes(h, num_frames, H, W, chunk_frames), dev)
        self.dec_ws = _ws(lib.mcg_decoder_workspace_bytes(h, num_frames), dev)
        self.trunk_done = [torch.cuda.Event() for _ in range(2)]
        self.dec_done = [torch.cuda.Event() for _ in range(2)]
        self.used = [False, False]
        self._hw_keep = [None, None]
        self.k = 0

    def submit(self, img, out, img_hw=None):
        """Enqueue one batch: img [N,3,H,W] f32 (must stay valid until its trunk ran), out = dict(gaze, boxes, scores)."""
        with torch.cuda.device(self.e.device):
            return self._submit(img, out, img_hw)

    def _submit(self, img, out, img_hw):
        e, lib, slot = self.e, self.e.lib, self.k & 1
        e._check_img(img)
        img_hw = e.img_hw_tensor(img_hw, self.N)
        if img_hw is not None:
            img_hw.record_stream(self.sb)       # allocated on the caller's stream, read by the decoder on sb: the caching allocator must
                                                # not hand the block out again before that read has happened
        self._hw_keep[slot] = img_hw
        cur = torch.cuda.current_stream(e.device)
        self.sa.wait_stream(cur)                       # input produced on the caller's stream
        if self.used[slot]:
            self.sa.wait_event(self.dec_done[slot])    # the decoder that read this pyramid slot has finished
        L.check(lib.mcg_backbone_fpn_forward(e._handle, C.c_void_p(self.sa.cuda_stream), _ptr(img), self.N, self.H, self.W, self.chunk,
                                             self.tabs[slot], _ptr(self.trunk_ws), self.trunk_ws.numel()), 'mcg_backbone_fpn_forward')
        self.trunk_done[slot].record(self.sa)
        self.sb.wait_event(self.trunk_done[slot])
        L.check(lib.mcg_decoder_forward(e._handle, C.c_void_p(self.sb.cuda_stream), self.tabs[slot], self.N, self.T, self.H, self.W,
                                        _ptr(img_hw), _ptr(out['gaze']), _ptr(out['boxes']), _ptr(out['scores']),
                                        _ptr(self.dec_ws), self.dec_ws.numel()), 'mcg_decoder_forward')
        self.dec_done[slot].record(self.sb)
        self.used[slot] = True
        self.k += 1
        return self.dec_done[slot]

    def flush(self):
        """Make the caller's stream wait for every submitted batch."""
        cur = torch.cuda.current_stream(self.e.device)
        for slot in range(2):
            if self.used[slot]:
                cur.wait_event(self.dec_done[slot])
