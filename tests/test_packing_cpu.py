"""CPU: the weight layouts the round-3 kernels read (bneck_x3.hpp's slab stream, chain_x3.hpp's split fragment-major matrices) against
their index formulas restated element by element, the fused-tail table PackedWeights builds, and bench.py's roofline bookkeeping."""
import os
import sys

import numpy as np
import torch

from mcgaze_amd import packing, synth


def _parts(w):
    hi = w.float().to(torch.float16)
    lo = (w.float() - hi.float()).to(torch.float16)
    return hi, lo


def test_split_halves_reconstruct_to_22_bits():
    """hi + lo = w to 2^-22 relative while the low half is a normal fp16 (|w| >= 0.125); below, the low half is subnormal and the
    error is absolute: 2^-25 (include/mcgaze_hip.h MCG_F16X3)."""
    w = torch.randn(64, 64) * 0.7
    hi, lo = _parts(w)
    err = ((hi.double() + lo.double()) - w.double()).abs()
    big = w.abs() >= 0.125
    assert float((err / w.double().abs().clamp_min(1e-30))[big].max()) <= 2.0 ** -22
    assert float(err[~big].max()) <= 2.0 ** -25


def test_bneck_slab_layout_matches_the_kernel_formulas():
    """packing._slab: [2 channel tiles][4 K-steps][high, low][64 lanes][8]; natural K order for conv2 (the B operand comes from the window
    planes), the chain permutation 16 s + 4 (lane >> 5) + (e & 3) + 8 (e >> 2) for conv3 / next conv1 (the B operand is the previous
    contraction's accumulators: channels {0..3, 8..11} + 4 half of a K-step) -- bneck_x3.hpp."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(64, 64, generator=g)
    hi, lo = _parts(w)
    for chain in (False, True):
        sl = packing._slab(w, chain)
        assert tuple(sl.shape) == (2, 4, 2, 64, 8) and sl.dtype == torch.float16
        rs = np.random.RandomState(1)
        for _ in range(400):
            ct, s, hl, lane, e = (int(rs.randint(n)) for n in (2, 4, 2, 64, 8))
            row, half = 32 * ct + (lane & 31), lane >> 5
            col = 16 * s + (4 * half + (e & 3) + 8 * (e >> 2) if chain else 8 * half + e)
            assert sl[ct, s, hl, lane, e] == (hi, lo)[hl][row, col]
        cols = sorted({16 * s + (4 * h + (e & 3) + 8 * (e >> 2) if chain else 8 * h + e) for s in range(4) for h in range(2) for e in range(8)})
        assert cols == list(range(64))          # every K index exactly once: a permutation inside each K-step


def test_bneck_stream_order_and_sizes():
    """Slab order = consumption order (include/mcgaze_hip.h mcg_fused_block): conv2 per K half / tap / output pair, then per 64-channel
    chunk of y the conv3 parts and the next conv1's pairs; 16 KiB per slab; bias [cm | 4 cm | cn | the three descale factors, 0]: every
    matrix is packed pre-scaled by its own power of two (packing.pow2_prescale)."""
    g = torch.Generator().manual_seed(6)
    for cm, k2, cn in ((64, 0, 64), (64, 64, 128), (64, 0, 0), (128, 0, 128), (128, 0, 0)):
        w2 = torch.randn(cm, 3, 3, cm, generator=g)
        w3 = torch.randn(4 * cm, cm + k2, generator=g)
        w1 = torch.randn(cn, 4 * cm, generator=g) if cn else None
        b = [torch.randn(n, generator=g) for n in (cm, 4 * cm, cn)]
        ws, bs = packing.bneck_stream(w2, b[0], w3, b[1], w1, b[2] if cn else None)
        nslab = 9 * (cm // 64) ** 2 + (cm // 16) * ((cm + k2) // 64 + cn // 64)
        assert ws.numel() == nslab * 8192 and ws.dtype == torch.float16 and bs.numel() == 5 * cm + cn + 4
        (w2, d2), (w3, d3) = packing.pow2_prescale(w2), packing.pow2_prescale(w3)        # what the slabs hold
        w1, d1 = packing.pow2_prescale(w1) if cn else (None, 1.0)
        for m, d in ((w2, d2), (w3, d3)) + (((w1, d1),) if cn else ()):
            assert 2.0 ** 13 < float(m.abs().max()) <= 2.0 ** 14 and np.log2(d) == round(np.log2(d))
        ws = ws.reshape(nslab, 2, 4, 2, 64, 8)
        # conv2: slab index ((kk * 9 + tap) * pairs + op)
        pairs = cm // 64
        kk, tap, op = pairs - 1, 5, pairs - 1
        want = packing._slab(w2[op * 64:(op + 1) * 64, tap // 3, tap % 3, kk * 64:(kk + 1) * 64], chain=False)
        assert torch.equal(ws[(kk * 9 + tap) * pairs + op], want)
        # chunk oc: parts of w3, then pairs of w1
        per = (cm + k2) // 64 + cn // 64
        oc = 4 * cm // 64 - 1
        base = 9 * pairs * pairs + oc * per
        part = (cm + k2) // 64 - 1
        assert torch.equal(ws[base + part], packing._slab(w3[oc * 64:(oc + 1) * 64, part * 64:(part + 1) * 64], chain=True))
        if cn:
            pr = cn // 64 - 1
            assert torch.equal(ws[base + (cm + k2) // 64 + pr], packing._slab(w1[pr * 64:(pr + 1) * 64, oc * 64:(oc + 1) * 64], chain=True))
        assert torch.equal(bs, torch.cat([b[0], b[1]] + ([b[2]] if cn else []) + [torch.tensor([d2, d3, d1, 0.0])]))


def test_frag_major_split_layout():
    g = torch.Generator().manual_seed(7)
    w = torch.randn(3, 256, 256, generator=g)
    f = packing.frag_major_split(w)
    assert tuple(f.shape) == (3, 256, 512) and f.dtype == torch.float16
    f = f.reshape(3, 8, 16, 2, 64, 8)
    hi, lo = _parts(w)
    rs = np.random.RandomState(2)
    for _ in range(400):
        m, t, ks, hl, lane, e = (int(rs.randint(n)) for n in (3, 8, 16, 2, 64, 8))
        assert f[m, t, ks, hl, lane, e] == (hi, lo)[hl][m, 32 * t + (lane & 31), 16 * ks + 8 * (lane >> 5) + e]


def test_packed_weights_fused_tail_table():
    """f16x3: layer1's three blocks (the first with its downsample conv as second K source) and layer2's identity blocks; next conv1
    fused where its width fits (64 / 128); none for the other engines."""
    sd = synth.make_state_dict(0)
    w = packing.PackedWeights(sd, dtype=torch.float32, device='cpu', split=True)
    got = [(f['conv2_index'], f['cm'], f['c'], f['cn'], f['nsrc']) for f in w.fused]
    assert got == [(1, 64, 256, 64, 2), (5, 64, 256, 64, 1), (8, 64, 256, 128, 1), (15, 128, 512, 128, 1), (18, 128, 512, 128, 1), (21, 128, 512, 0, 1)]
    for f in w.fused:
        nslab = 9 * (f['cm'] // 64) ** 2 + (f['cm'] // 16) * (f['cm'] // 64 + f['nsrc'] - 1 + f['cn'] // 64)
        assert f['wstream'].numel() == nslab * 8192 and f['bias'].numel() == 5 * f['cm'] + f['cn'] + 4   # + descale x 3, pad
        c2 = w.convs[f['conv2_index']]
        assert c2['k'] == 3 and c2['stride'] == 1 and c2['cin'] == f['cm']
    assert packing.PackedWeights(sd, dtype=torch.bfloat16, device='cpu').fused == []
    assert packing.PackedWeights(sd, dtype=torch.float32, device='cpu').fused == []
    st = w.stages[0]
    assert st['OUT_PROJ_WF'].dtype == torch.float16 and tuple(st['REG_FC_WF'].shape) == (3, 256, 512)
    # ABI 13: the f16x3 attention block reads in_proj in the split fragment-major form (24 column tiles); the 16-bit engines the plain one, in THEIR format
    assert st['IN_PROJ_WF'].dtype == torch.float16 and tuple(st['IN_PROJ_WF'].shape) == (768, 512)
    assert torch.equal(st['IN_PROJ_WF'], packing.frag_major_split(torch.as_tensor(sd['roi_head.bbox_head.0.attention.attn.in_proj_weight'])))
    for dt in (torch.bfloat16, torch.float16):
        w16 = packing.PackedWeights(sd, dtype=dt, device='cpu')
        s16 = w16.stages[0]
        assert s16['IN_PROJ_WF'].dtype == dt and tuple(s16['IN_PROJ_WF'].shape) == (768, 256)
        assert torch.equal(s16['IN_PROJ_WF'], packing.frag_major(torch.as_tensor(sd['roi_head.bbox_head.0.attention.attn.in_proj_weight']).to(dt)))
        assert w16.convs[0]['w'].dtype == dt and w16.convs[0]['wf'] is not None and w16.convs[0]['wf'].dtype == dt    # MCG_F16 shares every layout with MCG_BF16


def test_bench_roofline_bookkeeping():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    rec = [(0.5, 1.0e11, 50, (1000, 256, 2304), 2.0e8), (0.25, 0.5e11, 50, (1000, 256, 2304), 1.0e8), (0.1, 1.0e9, 51, (100, 256, 256), 1.0e6),
           (0.9, 2.0e11, 70, (5000, 320, 896), 3.6e9)]
    r = bench.roofline_of(rec, 'f16x3')
    assert r['kernel'] == bench.CFG_NAMES[70] and r['launches_per_step'] == 1          # dominant = the largest total time
    assert abs(r['achieved'] - 2.0e11 / 0.9e-3 / 1e12) < 0.01 and r['algorithmic_bytes_per_launch'] == int(3.6e9)
    assert len(r['launches']) == 4 and r['launches'][0][:4] == [50, 1000, 256, 2304] and r['algorithmic_bytes_step'] == int(2.0e8 + 1.0e8 + 1.0e6 + 3.6e9)
    assert 'three fp16 MFMAs' in r['note']


def test_wino_pack_layout_and_transform():
    """packing.wino_pack against the statement of wino_x3.hpp: decode the fragment-major split layout back into U_nu[co][ky][ci] and run the
    1-D Winograd F(2,3) recurrence in float64 -- it must reproduce the direct 3x3 convolution (so the transform matrix, the sign flip of
    position 2, the K-step order and the lane mapping are the kernel's)."""
    import torch.nn.functional as F
    from mcgaze_amd.packing import wino_pack
    g = torch.Generator().manual_seed(5)
    cout, cin, H, W = 128, 32, 5, 6
    w = torch.randn(cout, cin, 3, 3, generator=g).double()
    x = torch.randn(2, cin, H, W, generator=g).double()
    u = wino_pack(w.permute(0, 2, 3, 1).contiguous())
    assert u.dtype == torch.float16 and u.numel() * 2 == (cout // 128) * (3 * cin // 16) * 32768
    v = u.reshape(cout // 128, cin // 16, 3, 4, 4, 2, 2, 32, 8).double()          # nt, cs, ky, nu, ct, hl, half, n, e
    v = v.sum(dim=5)                                                               # hi + lo
    U = v.permute(3, 0, 4, 6, 2, 1, 5, 7).reshape(4, cout, 3, cin)                 # nu, (nt ct n), ky, (cs half e)
    gk = w.permute(0, 2, 3, 1)                                                     # co, ky, kx, ci
    want = torch.stack([gk[:, :, 0], (gk[:, :, 0] + gk[:, :, 1] + gk[:, :, 2]) / 2, -(gk[:, :, 0] - gk[:, :, 1] + gk[:, :, 2]) / 2, gk[:, :, 2]])
    assert float((U - want).abs().max()) < 2e-6                                    # fp16 hi + lo: 2^-22 relative of O(1) values
    xp = F.pad(x, (1, 2, 1, 1))                                                    # x halo (one extra column for an odd tail), y halo
    y = torch.zeros(2, cout, H, W, dtype=torch.float64)
    for x0 in range(0, W, 2):
        for ky in range(3):
            d = [xp[:, :, ky:ky + H, x0 + i] for i in range(4)]                    # [N, ci, H] each: input columns x0 - 1 .. x0 + 2
            V = [d[0] - d[2], d[1] + d[2], d[1] - d[2], d[1] - d[3]]
            M = [torch.einsum('nch,oc->noh', V[nu], want[nu][:, ky]) for nu in range(4)]
            y[:, :, :, x0] += M[0] + M[1] + M[2]
            if x0 + 1 < W:
                y[:, :, :, x0 + 1] += M[1] - M[2] - M[3]
    ref = F.conv2d(x, w, padding=1)
    assert float((y - ref).abs().max()) < 1e-9


def test_pow2_prescale_keeps_small_weights_at_22_bits():
    """hi + lo of the PRE-SCALED matrix reproduces every weight to 2^-21 relative for weight scales 1e-4 .. 1e2 (unscaled, a
    weight of 1e-3 keeps ~15 bits: its low half is an fp16 subnormal), the scale is an exact power of two and the descale is its inverse."""
    g = torch.Generator().manual_seed(8)
    for scale in (1e-4, 1e-3, 1e-2, 1.0, 1e2):
        w = torch.randn(64, 256, generator=g).double() * scale
        ws, d = packing.pow2_prescale(w)
        assert np.log2(d) == round(np.log2(d)) and torch.equal(ws * d, w)
        assert 2.0 ** 13 < float(ws.abs().max()) <= 2.0 ** 14
        p = packing.split_pack(ws).double().reshape(64, 32, 2, 8)
        back = (p[:, :, 0] + p[:, :, 1]).reshape(64, 256) * d
        big = w.abs() > 2.0 ** -17 * w.abs().max()          # weights within 2^-17 of the largest keep full precision
        rel = ((back - w).abs() / w.abs())[big]
        assert float(rel.max()) < 2.0 ** -21, (scale, float(rel.max()))
        raw = packing.split_pack(w).double().reshape(64, 32, 2, 8)
        rel_raw = (((raw[:, :, 0] + raw[:, :, 1]).reshape(64, 256) - w).abs() / w.abs())[big]
        if scale <= 1e-3:
            assert float(rel_raw.max()) > 2.0 ** -17             # the unscaled split really is worse there


def test_wino_pack_f43_transform():
    """packing.wino_pack(g=4): the decoded U with the kernel's F(4,3) input and output matrices (wino_x3.hpp header) reproduces the direct 3x3
    convolution in float64."""
    import torch.nn.functional as F
    from mcgaze_amd.packing import wino_pack
    g = torch.Generator().manual_seed(9)
    cout, cin, H, W = 128, 32, 4, 8
    w = torch.randn(cout, cin, 3, 3, generator=g).double()
    x = torch.randn(2, cin, H, W, generator=g).double()
    u = wino_pack(w.permute(0, 2, 3, 1).contiguous(), g=4)
    assert u.numel() * 2 == (cout // 128) * (3 * cin // 16) * 6 * 4 * 2 * 1024
    v = u.reshape(cout // 128, cin // 16, 3, 6, 4, 2, 2, 32, 8).double().sum(dim=5)      # nt, cs, ky, nu, ct, half, n, e
    U = v.permute(3, 0, 4, 6, 2, 1, 5, 7).reshape(6, cout, 3, cin)
    Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
    At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
    xp = F.pad(x, (1, 4, 1, 1))
    y = torch.zeros(2, cout, H, W, dtype=torch.float64)
    for x0 in range(0, W, 4):
        for ky in range(3):
            d = torch.stack([xp[:, :, ky:ky + H, x0 + i] for i in range(6)])                # [6, N, ci, H]
            V = torch.einsum('pf,fnch->pnch', Bt, d)
            M = torch.stack([torch.einsum('nch,oc->noh', V[nu], U[nu][:, ky]) for nu in range(6)])
            Y = torch.einsum('jp,pnoh->jnoh', At, M)
            for j in range(4):
                if x0 + j < W:
                    y[:, :, :, x0 + j] += Y[j]
    ref = F.conv2d(x, w, padding=1)
    assert float((y - ref).abs().max()) < 2e-5 * float(ref.abs().max())    # fp16 hi + lo of U: 2^-22 per weight, amplified by the x8 of the output transform
