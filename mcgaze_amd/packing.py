"""Checkpoint -> kernel-ready weights (load time, not the hot path).

Takes the reference's ``state_dict`` layout (the dead ``fc_cls`` /
``fc_reg`` and the unused ``gaze_head.0-2`` are tolerated and ignored) and produces the
buffers ``libmcgaze_hip.so`` expects (``include/mcgaze_hip.h``):

* BN folded into the preceding conv: ``w' = w * g/sqrt(var+eps)``, ``b' = beta - mean*g/sqrt(var+eps)``
  (eval-mode BN, resnet.py:648-658), computed in float64;
* conv weights OIHW -> OHWI (K = (kh, kw, cin) contiguous) in the compute dtype;
* stem 7x7x3 -> [64][7][8][4] (kw and channel zero-padded: one kernel row is one 32-element tap);
* ``dynamic_layer`` rows permuted so the generated 1x1-conv weights come out in dynconv_kernel's MFMA-fragment-major read order;
* per-clue heads and the gaze-head branches stacked into the tables the kernels index.
"""
import numpy as np
import torch

from .synth import ARCH

CLUES = ('face', 'eyes', 'head')


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))


def normalize_state_dict(sd):
    """init_detector's key rewrite (mmdet/apis/inference.py:45): strip ``module.``; accept the
    mmcv ``{'meta':…, 'state_dict':…}`` envelope."""
    if 'state_dict' in sd and not any(k.startswith('backbone.') for k in sd):
        sd = sd['state_dict']
    out = {}
    for k, v in sd.items():
        if k.startswith('module.'):
            k = k[len('module.'):]
        out[k] = _t(v)
    return out


def fold_bn(sd, conv_key, bn_prefix, eps=1e-5):
    w = sd[conv_key].double()
    g, b = sd[bn_prefix + '.weight'].double(), sd[bn_prefix + '.bias'].double()
    mean, var = sd[bn_prefix + '.running_mean'].double(), sd[bn_prefix + '.running_var'].double()
    scale = g / torch.sqrt(var + eps)
    return (w * scale[:, None, None, None]).float(), (b - mean * scale).float()


def ohwi(w):
    return w.permute(0, 2, 3, 1).contiguous()


def frag_major(w):
    """[..., 32 nt out, 16 nk in] -> MFMA-fragment-major [..., t=nt][ks=nk][lane=64][e=8] with
    WF[t][ks][lane][e] = W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + e] (include/mcgaze_hip.h: MCG_SW_*_WF, mcg_conv_weights.wf)."""
    lead = w.shape[:-2]
    rows, K = w.shape[-2:]
    assert K % 16 == 0 and rows % 32 == 0, (rows, K)
    v = w.reshape(*lead, rows // 32, 32, K // 16, 2, 8)        # t, n, ks, half, e
    n = len(lead)
    v = v.permute(*range(n), n, n + 2, n + 3, n + 1, n + 4)    # t, ks, half, n, e  -> lane = 32 half + n
    return v.contiguous().reshape(*lead, rows, K)


def frag_major_split(w):
    """f32 [..., 32 nt out, 16 nk in] -> the MFMA-fragment-major SPLIT copy chain_x3.hpp reads: fp16 [..., t = nt][ks = nk][high, low][lane = 64][e = 8]
    with element (t, ks, hl, lane, e) = the fp16 high (hl = 0) / low part of W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + e]; 4 bytes
    per weight, one wave-wide 16-byte load per (tile, K-step, part) is 1 KiB contiguous."""
    w = w.double()
    hi = w.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (w - hi.double()).clamp(-65504.0, 65504.0).to(torch.float16)
    lead = w.shape[:-2]
    rows, K = w.shape[-2:]
    fh, fl = frag_major(hi).reshape(*lead, rows // 32, K // 16, 64, 8), frag_major(lo).reshape(*lead, rows // 32, K // 16, 64, 8)
    return torch.stack([fh, fl], dim=-3).reshape(*lead, rows, 2 * K).contiguous()


def pow2_prescale(w):
    """(w * 2^e, 2^-e) with e chosen so that max |w| * 2^e lies in (2^13, 2^14]: the exact power-of-two pre-scale of an MCG_F16X3 weight
    matrix (include/mcgaze_hip.h: mcg_conv_desc.wscale).  An fp16 low half is a normal number down to 2^-14, i.e. for every weight within
    2^-17 of the largest one -- unscaled, a BN-folded weight of 1e-3 keeps 15 of its 22 bits.  The kernels multiply the f32 sums by 2^-e
    before the bias (exact).  An all-zero matrix is left alone."""
    w = w.double()
    m = float(w.abs().max())
    if not (m > 0.0) or not np.isfinite(m):
        return w, 1.0
    e = 14 - int(np.ceil(np.log2(m)))
    return w * (2.0 ** e), float(2.0 ** -e)


def split_pack(w):
    """f32 [..., K] -> fp16 [..., 2K], the weight operand of the MCG_F16X3 contraction (include/mcgaze_hip.h): per 8 consecutive
    K elements a 16-byte chunk of fp16 high parts, then a 16-byte chunk of fp16 low parts; hi = f16(w), lo = f16(w - hi), so
    hi + lo = w to 2^-22 relative for |w| >= 0.125; a smaller weight has its low half in fp16's subnormal range (absolute error
    2^-25: |w| = 1e-2 keeps ~18 bits, 1e-3 ~15; values beyond +-65504 saturate per half; low parts below 3e-8 vanish).  4 bytes per element,
    like the f32 matrix it replaces."""
    w = w.double()
    K = w.shape[-1]
    assert K % 8 == 0, f'split_pack: K={K} must be a multiple of 8'
    hi = w.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (w - hi.double()).clamp(-65504.0, 65504.0).to(torch.float16)
    lead = w.shape[:-1]
    v = torch.stack([hi.reshape(*lead, K // 8, 8), lo.reshape(*lead, K // 8, 8)], dim=-2)   # [..., K/8, 2, 8]
    return v.reshape(*lead, 2 * K).contiguous()


WINO_G = {2: [[1, 0, 0], [0.5, 0.5, 0.5], [-0.5, 0.5, -0.5], [0, 0, 1]],                      # F(2,3); row 2 carries the kernel's sign flip
          4: [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]}   # F(4,3), points 0, +-1, +-2, inf


def wino_pack(w, g=2):
    """f32 / f64 OHWI [Cout][3][3][Cin] (BN folded) -> the weight operand of wino_x3.hpp (mcg_conv3x3_wino_x3; mcg_conv_weights.wf of
    a 3x3 conv, MCG_F16X3): the kernel row (kx) of every y tap transformed by the F(2,3) matrix G = [[1,0,0],[1/2,1/2,1/2],[-1/2,1/2,-1/2],
    [0,0,1]] IN FLOAT64 (row 2 carries the sign flip the kernel's input transform uses), split into fp16 high / low parts
    (hi = f16(u), lo = f16(u - hi)), laid out fp16 [Cout / 128][K step = 3 cs + ky][nu 4][channel tile 4][high, low][lane 64][8]
    with element (nt, 3 cs + ky, nu, ct, hl, lane, e) = part hl of U_nu[128 nt + 32 ct + (lane & 31)][ky][16 cs + 8 (lane >> 5) + e]:
    one K step's 32 KiB are contiguous and MFMA-fragment-major."""
    cout, kh, kw, cin = w.shape
    assert (kh, kw) == (3, 3) and cout % 128 == 0 and cin % 32 == 0 and g in WINO_G, (tuple(w.shape), g)
    gm = torch.tensor(WINO_G[g], dtype=torch.float64, device=w.device)                          # g = 4: F(4,3), six positions [Cout/128][K step][6][4][hl][lane][8]
    nv = gm.shape[0]
    u = torch.einsum('pk,oykc->poyc', gm, w.double())                          # [nu][co][ky][ci]
    hi = u.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (u - hi.double()).clamp(-65504.0, 65504.0).to(torch.float16)
    v = torch.stack([hi, lo])                                                  # [hl][nu][co][ky][ci]
    v = v.reshape(2, nv, cout // 128, 4, 32, 3, cin // 16, 2, 8)               # hl, nu, nt, ct, n, ky, cs, half, e
    v = v.permute(2, 6, 5, 1, 3, 0, 7, 4, 8)                                   # nt, cs, ky, nu, ct, hl, half, n, e
    return v.contiguous().reshape(-1)


def _slab(w64, chain):
    """f32 [64 out][64 k] -> one 16 KiB weight slab of bneck_x3.hpp: fp16 [2 channel tiles][4 K-steps][high, low][64 lanes][8], lane l
    holding row 32 ct + (l & 31) and, for e < 8, column 16 s + 8 (l >> 5) + e (chain = False: the 3x3 conv, whose B operand comes
    from the window planes in natural order) or 16 s + 4 (l >> 5) + (e & 3) + 8 (e >> 2) (chain = True: the B operand is the previous
    contraction's accumulator registers, which hold channels {0..3, 8..11} + 4 (l >> 5) of a K-step)."""
    assert tuple(w64.shape) == (64, 64)
    w64 = w64.double()
    hi = w64.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (w64 - hi.double()).clamp(-65504.0, 65504.0).to(torch.float16)
    lane, e, s, ct = torch.arange(64), torch.arange(8), torch.arange(4), torch.arange(2)
    half = lane >> 5
    if chain:
        col = 16 * s[:, None, None] + 4 * half[None, :, None] + (e & 3)[None, None, :] + 8 * (e >> 2)[None, None, :]
    else:
        col = 16 * s[:, None, None] + 8 * half[None, :, None] + e[None, None, :]
    row = 32 * ct[:, None, None, None] + (lane & 31)[None, None, :, None]                     # [ct, 1, lane, 1]
    col = col[None].expand(2, 4, 64, 8)
    row = row.expand(2, 4, 64, 8)
    return torch.stack([hi[row, col], lo[row, col]], dim=2)                                  # [ct][s][hl][lane][e]


def bneck_stream(w2, b2, w3, b3, w1n=None, b1n=None):
    """Weight stream + bias block of one fused bottleneck tail (include/mcgaze_hip.h: mcg_fused_block; bneck_x3.hpp).
    w2 [cm][3][3][cm] OHWI (cm = 64 or 128), w3 [4 cm][cm (+ 64: the downsample conv's K-concatenated input, cm = 64 only)], w1n
    [cn][4 cm] or None -- all f32 with BN folded.  Slab order = consumption order: conv2 per 64-channel K half, per tap, per 64-row
    output pair; then per 64-channel chunk oc of y: w3[oc chunk][K part] for each K part of 64, w1n[64-row pair][oc chunk] for each
    pair.  -> (fp16 tensor of 8192 halves per slab, f32 bias [cm | 4 cm | cn | descale of w2, w3, w1n, 0])."""
    cm = w2.shape[0]
    assert cm in (64, 128) and tuple(w2.shape) == (cm, 3, 3, cm) and w3.shape[0] == 4 * cm and w3.shape[1] in ((64, 128) if cm == 64 else (128,))
    cn = 0 if w1n is None else w1n.shape[0]
    assert cn % 64 == 0 and cn <= 128 and (w1n is None or w1n.shape[1] == 4 * cm)
    # each matrix pre-scaled by its own power of two (pow2_prescale); the descale factors ride behind the biases
    w2, d2 = pow2_prescale(w2)
    w3, d3 = pow2_prescale(w3)
    w1n, d1 = pow2_prescale(w1n) if cn else (None, 1.0)
    slabs = [_slab(w2[op * 64:(op + 1) * 64, kh, kw, kk * 64:(kk + 1) * 64], chain=False)
             for kk in range(cm // 64) for kh in range(3) for kw in range(3) for op in range(cm // 64)]
    for oc in range(4 * cm // 64):
        for part in range(w3.shape[1] // 64):
            slabs.append(_slab(w3[oc * 64:(oc + 1) * 64, part * 64:(part + 1) * 64], chain=True))
        for pair in range(cn // 64):
            slabs.append(_slab(w1n[pair * 64:(pair + 1) * 64, oc * 64:(oc + 1) * 64], chain=True))
    bias = torch.cat([b2.float(), b3.float()] + ([b1n.float()] if cn else []) + [torch.tensor([d2, d3, d1, 0.0], dtype=torch.float32, device=b2.device)])
    return torch.stack(slabs).reshape(-1).contiguous(), bias.contiguous()


def dyn_permutation(d=256, feat=64, epc=4):
    """Row permutation of dynamic_layer (transformer.py:1134-1137: params[:, :d*feat].view(d, feat) = param_in, params[:, -d*feat:]
    .view(feat, d) = param_out) so that a token's generated weights come out in the order dynconv_kernel READS them: MFMA-fragment-major,
    one contiguous KiB per wave-wide 16-byte load (round 5; before: K-contiguous rows, where a load touched 32 rows x 32 bytes).
    ``epc`` = K elements a lane holds per load group: 4 (fp32 engine: one 16-byte chunk), 8 (bf16: one chunk; f16x3: two chunks of f32 that
    dynconv_x3_kernel splits into the eight halves of a 32x32x16 step).
      stage 1, B operand = param_in^T [n < feat][k < d]: offset ((tn * (d / 2 epc) + j) * 64 + lane) * epc + e holds
               n = 32 tn + (lane & 31), k = (2 j + (lane >> 5)) * epc + e                          <- old row k * feat + n
      stage 2, B operand = param_out^T [n < d][k < feat], wave w = n / 64, b = (n / 32) & 1: offset d * feat +
               (((2 w + b) * (feat / 2 epc) + j) * 64 + lane) * epc + e holds n = 64 w + 32 b + (lane & 31), k = (2 j + (lane >> 5)) * epc + e
                                                                                                   <- old row d * feat + k * d + n"""
    lane, e = torch.arange(64), torch.arange(epc)
    def frag(tiles, pairs):   # -> n [tiles, pairs, 64, epc], k [tiles, pairs, 64, epc]
        t, j = torch.arange(tiles), torch.arange(pairs)
        n = (32 * t[:, None, None, None] + (lane & 31)[None, None, :, None]).expand(tiles, pairs, 64, epc)
        k = ((2 * j[None, :, None, None] + (lane >> 5)[None, None, :, None]) * epc + e[None, None, None, :]).expand(tiles, pairs, 64, epc)
        return n.reshape(-1), k.reshape(-1)
    n, k = frag(feat // 32, d // (2 * epc))
    p_in = k * feat + n
    n2, k2 = frag(d // 32, feat // (2 * epc))
    p_out = d * feat + k2 * d + n2
    return torch.cat([p_in, p_out])


class PackedWeights:
    """Device-resident packed weights + the geometry tables the engine needs."""

    def __init__(self, state_dict, depth=50, num_stages=4, dtype=torch.bfloat16, device='cuda:0', fuse_downsample=True, split=False):
        """``dtype``: storage type of activations (and of the matrices for MCG_F32 / MCG_BF16).  ``split=True`` (MCG_F16X3, dtype
        must be float32): every conv / linear matrix is split-packed fp16 (``split_pack``) over its flattened K = (kh, kw, cin)."""
        sd = normalize_state_dict(state_dict)
        self.dtype, self.device, self.depth, self.num_stages, self.split = dtype, torch.device(device), depth, num_stages, split
        assert not split or dtype == torch.float32
        self.blocks = ARCH[depth]
        self._keep = []
        # matrices: K is the trailing axis after flattening (kh, kw, cin) -- conv weights arrive here as OHWI
        mat = (lambda t: self._dev(split_pack(t))) if split else (lambda t: self._dev(t.to(dtype)))               # [..., out, in]
        cmat = (lambda t: self._dev(split_pack(t.reshape(t.shape[0], -1)))) if split else mat                      # OHWI conv weight

        # one mcg_conv_weights entry from the folded OIHW weight (or an OHWI one): w, the optional second copy wf and, f16x3 trunk convs,
        # the power-of-two pre-scale both copies carry (pow2_prescale) with its descale factor
        def entry(w, b, k, stride, pad, wf=None, w_ohwi=None, wf4=None):
            wo = ohwi(w) if w_ohwi is None else w_ohwi
            ws = 0.0
            if split:
                wo, ws = pow2_prescale(wo)
            return dict(w=cmat(wo), bias=vec(b), cin=wo.shape[3], cout=wo.shape[0], k=k, stride=stride, pad=pad,
                        wf=wf(wo.permute(0, 3, 1, 2)) if wf is not None else None, wscale=ws,
                        wf4=wf4(wo.permute(0, 3, 1, 2)) if wf4 is not None else None)
        vec = lambda t: self._dev(t.float())
        # fragment-major copies of the 1x1 convs (bf16 engine): operands of the register-resident-weight kernels (pw_pair.hpp, pw_single.hpp)
        if dtype in (torch.bfloat16, torch.float16):   # the 16-bit engines (MCG_BF16 / MCG_F16 share every layout)
            wf1x1 = lambda w: self._dev(frag_major(w.reshape(w.shape[0], -1).to(dtype)))
        elif split:   # f16x3: split fragment-major copies for the shapes pw_single_x3.hpp serves (256 -> 256 / 1024)
            wf1x1 = lambda w: (self._dev(frag_major_split(w.reshape(w.shape[0], -1))) if (w.shape[1] == 256 and w.shape[0] in (256, 1024) and w.reshape(w.shape[0], -1).shape[1] == 256) else None)
        else:
            wf1x1 = lambda w: None

        # f16x3: Winograd F(2,3) copies of the stride-1 3x3 convs whose channel counts wino_x3.hpp tiles (FPN outputs, layer3 / layer4 conv2)
        wf3x3 = (lambda w: self._dev(wino_pack(ohwi(w)))) if split else (lambda w: None)
        wf3x3 = (lambda f: (lambda w: f(w) if (w.shape[0] % 128 == 0 and w.shape[1] % 32 == 0 and w.shape[1] >= 256) else None))(wf3x3)
        # ... and F(4,3) copies for the FPN output convs (maps whose width is a multiple of 4 and >= 16: P2 / P3 of a 224 x 224 input)
        wf3x3_g4 = (lambda w: self._dev(wino_pack(ohwi(w), g=4))) if split else None
        w, b = fold_bn(sd, 'backbone.conv1.weight', 'backbone.bn1')
        stem = torch.zeros(64, 7, 8, 4)
        stem[:, :, :7, :3] = w.permute(0, 2, 3, 1)
        self.stem = dict(w=cmat(stem), bias=vec(b), cin=32, cout=64, k=7, stride=2, pad=3, wscale=0.0)   # mcg_stem_forward takes no descale: unscaled
        self.convs = []
        self.fused = []  # f16x3: fused bottleneck tails (bneck_stream), dicts of wstream / bias / conv2_index / cm / c / cn / nsrc
        folded = []      # (w OIHW f32, b f32) of every entry of self.convs, for the fused tails
        self.c3_ds = []  # per layer: first block's conv3 + downsample as one K-concatenated 1x1 conv
        for li, nb in enumerate(self.blocks):
            for bi in range(nb):
                p = f'backbone.layer{li + 1}.{bi}'
                stride = 2 if (bi == 0 and li > 0) else 1
                for conv, bn, k, s, pad in (('conv1', 'bn1', 1, 1, 0), ('conv2', 'bn2', 3, stride, 1), ('conv3', 'bn3', 1, 1, 0)):
                    w, b = fold_bn(sd, f'{p}.{conv}.weight', f'{p}.{bn}')
                    self.convs.append(entry(w, b, k, s, pad, wf=wf1x1 if k == 1 else (wf3x3 if s == 1 else None)))
                    folded.append((w, b))
                if f'{p}.downsample.0.weight' in sd:
                    w3, b3 = w, b  # conv3 of this block (last of the loop above)
                    w, b = fold_bn(sd, f'{p}.downsample.0.weight', f'{p}.downsample.1')
                    self.convs.append(entry(w, b, 1, stride, 0))
                    folded.append((w, b))
                    if fuse_downsample:
                        wcat = torch.cat([ohwi(w3), ohwi(w)], dim=3)  # [Cout,1,1,planes + inplanes]
                        self.c3_ds.append(entry(None, b3 + b, 1, 1, 0, wf=wf1x1, w_ohwi=wcat))
        if split and fuse_downsample and depth >= 50:
            # layer1 (64 mid channels; every block, the first with its downsample conv as a second K source) and layer2 (128; the
            # identity blocks -- the first block's conv2 has stride 2): conv2 -> conv3 (+ downsample | + residual) -> next conv1
            ci = 0
            for li in range(2):
                cm = 64 << li
                for bi in range(self.blocks[li]):
                    has_ds = bi == 0
                    nxt = ci + (4 if has_ds else 3)
                    (w2, b2), (w3, b3) = folded[ci + 1], folded[ci + 2]
                    ok = tuple(w2.shape) == (cm, cm, 3, 3) and self.convs[ci + 1]['stride'] == 1 and w3.shape[0] == 4 * cm and not (has_ds and li > 0)
                    if ok:
                        w3m, b3m = w3.reshape(w3.shape[0], -1), b3
                        if has_ds:
                            wd, bd = folded[ci + 3]
                            ok = self.convs[ci + 3]['stride'] == 1 and wd.shape[1] == 64
                            w3m, b3m = torch.cat([w3m, wd.reshape(wd.shape[0], -1)], dim=1), b3 + bd
                    if ok:
                        w1n = b1n = None
                        if nxt < len(folded) and self.convs[nxt]['k'] == 1 and self.convs[nxt]['stride'] == 1 and self.convs[nxt]['cout'] in ((64, 128) if cm == 64 else (128,)):
                            w1n, b1n = folded[nxt][0].reshape(folded[nxt][0].shape[0], -1), folded[nxt][1]
                        ws, bs = bneck_stream(ohwi(w2), b2, w3m, b3m, w1n, b1n)
                        self.fused.append(dict(wstream=self._dev(ws), bias=self._dev(bs), conv2_index=ci + 1, cm=cm, c=4 * cm,
                                               cn=0 if w1n is None else w1n.shape[0], nsrc=2 if has_ds else 1))
                    ci = nxt
        self.lateral, self.fpn_out = [], []
        for i in range(4):
            w = sd[f'neck.lateral_convs.{i}.conv.weight']
            self.lateral.append(entry(w, sd[f'neck.lateral_convs.{i}.conv.bias'], 1, 1, 0, wf=wf1x1))
            w = sd[f'neck.fpn_convs.{i}.conv.weight']
            self.fpn_out.append(entry(w, sd[f'neck.fpn_convs.{i}.conv.bias'], 3, 1, 1, wf=wf3x3, wf4=wf3x3_g4))
        self.init_boxes = vec(sd['rpn_head.init_proposal_bboxes.weight'])
        self.init_feats = self._dev(sd['rpn_head.init_proposal_features.weight'].to(dtype))   # read by a non-GEMM kernel: storage dtype
        perm = dyn_permutation(epc=8 if (dtype in (torch.bfloat16, torch.float16) or split) else 4)   # f16x3: dynconv_x3_kernel takes eight K elements per lane and step
        self.stages = []
        for s in range(num_stages):
            p = f'roi_head.bbox_head.{s}'
            q = p + '.instance_interactive_conv'
            st = dict(
                IN_PROJ_W=mat(sd[p + '.attention.attn.in_proj_weight']), IN_PROJ_B=vec(sd[p + '.attention.attn.in_proj_bias']),
                OUT_PROJ_W=mat(sd[p + '.attention.attn.out_proj.weight']), OUT_PROJ_B=vec(sd[p + '.attention.attn.out_proj.bias']),
                ATTN_LN_G=vec(sd[p + '.attention_norm.weight']), ATTN_LN_B=vec(sd[p + '.attention_norm.bias']),
                DYN_W=mat(sd[q + '.dynamic_layer.weight'][perm]), DYN_B=vec(sd[q + '.dynamic_layer.bias'][perm]),
                NORM_IN_G=vec(sd[q + '.norm_in.weight']), NORM_IN_B=vec(sd[q + '.norm_in.bias']),
                NORM_OUT_G=vec(sd[q + '.norm_out.weight']), NORM_OUT_B=vec(sd[q + '.norm_out.bias']),
                FC_W=mat(sd[q + '.fc_layer.weight']), FC_B=vec(sd[q + '.fc_layer.bias']),
                FC_LN_G=vec(sd[q + '.fc_norm.weight']), FC_LN_B=vec(sd[q + '.fc_norm.bias']),
                IIC_LN_G=vec(sd[p + '.instance_interactive_conv_norm.weight']), IIC_LN_B=vec(sd[p + '.instance_interactive_conv_norm.bias']),
                FFN1_W=mat(sd[p + '.ffn.layers.0.0.weight']), FFN1_B=vec(sd[p + '.ffn.layers.0.0.bias']),
                FFN2_W=mat(sd[p + '.ffn.layers.1.weight']), FFN2_B=vec(sd[p + '.ffn.layers.1.bias']),
                FFN_LN_G=vec(sd[p + '.ffn_norm.weight']), FFN_LN_B=vec(sd[p + '.ffn_norm.bias']),
                CLS_FC_W=mat(sd[p + '.cls_fcs.0.weight']), CLS_LN_G=vec(sd[p + '.cls_fcs.1.weight']), CLS_LN_B=vec(sd[p + '.cls_fcs.1.bias']),
                REG_FC_W=mat(torch.stack([sd[p + f'.reg_fcs.{3 * j}.weight'] for j in range(3)])),
                REG_LN_G=vec(torch.stack([sd[p + f'.reg_fcs.{3 * j + 1}.weight'] for j in range(3)])),
                REG_LN_B=vec(torch.stack([sd[p + f'.reg_fcs.{3 * j + 1}.bias'] for j in range(3)])),
                HEAD_CLS_W=vec(torch.cat([sd[p + f'.{c}_fc_cls.weight'] for c in CLUES])),
                HEAD_CLS_B=vec(torch.cat([sd[p + f'.{c}_fc_cls.bias'] for c in CLUES])),
                HEAD_REG_W=vec(torch.stack([sd[p + f'.{c}_fc_reg.weight'] for c in CLUES])),
                HEAD_REG_B=vec(torch.stack([sd[p + f'.{c}_fc_reg.bias'] for c in CLUES])))
            raw = dict(OUT_PROJ_W=sd[p + '.attention.attn.out_proj.weight'], CLS_FC_W=sd[p + '.cls_fcs.0.weight'],
                       IN_PROJ_W=sd[p + '.attention.attn.in_proj_weight'],       # ABI 13: the f16x3 attention block (attn_block_x3.hpp)
                       REG_FC_W=torch.stack([sd[p + f'.reg_fcs.{3 * j}.weight'] for j in range(3)]),
                       DYN_W=sd[q + '.dynamic_layer.weight'][perm])
            for k in ('OUT_PROJ_W', 'CLS_FC_W', 'REG_FC_W', 'IN_PROJ_W', 'DYN_W'):   # fragment-major copies for the fused chain / attention-block kernels
                if dtype in (torch.bfloat16, torch.float16):
                    st[k + 'F'] = frag_major(st[k])
                elif split and k in raw:
                    st[k + 'F'] = self._dev(frag_major_split(raw[k]))              # f16x3: the chains' split fragment-major operands (chain_x3.hpp)
                else:
                    st[k + 'F'] = st[k]
            assert st['HEAD_CLS_W'].shape == (3, 256), 'use_sigmoid=True heads expected (gaze_stqi_head.py:72-75)'
            self.stages.append(st)
        # only the LAST stage's gaze head runs at inference (multiclue_gaze_roi_head.py:367,378)
        g = f'roi_head.gaze_head.{num_stages - 1}'
        branches = [f'gaze_{c}_fcs' for c in CLUES] + [f'gaze_{c}_confidence' for c in CLUES]  # branch = 3*k + clue
        outs = [f'fc_{c}' for c in CLUES] + [f'fc_{c}_confidence' for c in CLUES]
        self.gaze = dict(
            FC_W=mat(torch.stack([torch.stack([sd[g + f'.{br}.{3 * j}.weight'] for j in range(2)]) for br in branches])),
            LN_G=vec(torch.stack([torch.stack([sd[g + f'.{br}.{3 * j + 1}.weight'] for j in range(2)]) for br in branches])),
            LN_B=vec(torch.stack([torch.stack([sd[g + f'.{br}.{3 * j + 1}.bias'] for j in range(2)]) for br in branches])),
            OUT_W=vec(torch.stack([sd[g + f'.{o}.weight'] for o in outs])), OUT_B=vec(torch.stack([sd[g + f'.{o}.bias'] for o in outs])),
            FUSE_W=vec(sd[g + '.fc_gaze.weight']), FUSE_B=vec(sd[g + '.fc_gaze.bias']))

    def _dev(self, t):
        t = t.contiguous().to(self.device)
        self._keep.append(t)
        return t

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._keep)
