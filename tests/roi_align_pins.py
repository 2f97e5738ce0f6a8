"""Independent pins for RoIAlign (mmcv.ops.RoIAlign(7, 1/stride, sampling_ratio=2, 'avg', aligned=True)), shared by the CPU
and GPU tests.  They are derived from the PUBLISHED kernel (mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh, v1.4.8:
`roi_align_forward_cuda_kernel` + `bilinear_interpolate`), not from the oracle's code:

    start = x1 * scale - 0.5 ; bin = (end - start) / 7 ; sample (iy, ix) of bin (ph, pw) at  start + ph*bin + (iy + .5)*bin/2
    sample outside  y < -1 || y > H || x < -1 || x > W  contributes 0 ;  y <= 0 -> 0 ;  y_low >= H-1 -> y_low = y_high = H-1, y = H-1
    output = sum of the 4 samples / 4

On an AFFINE map  f(y, x) = base + 10 y + x  bilinear interpolation is exact, so a sample's value is known in closed form:
0 if outside, else base + 10 clamp(y, 0, H-1) + clamp(x, 0, W-1).  `expected_affine` below evaluates that formula -- no gather, no
interpolation weights -- and the box list enumerates the kernel's edge cases with coordinates that are exact in binary.
"""
import numpy as np

# (name, box xyxy in IMAGE pixels) for a level of stride 4 with a 16 x 16 map (64 x 64 px frame); box sizes < 112 px -> level 0
EDGE_BOXES = [
    ('interior', (10.0, 14.0, 38.0, 42.0)),                 # start 2.0 / 3.0, bin 1.0: samples at k + .25 / .75
    ('sample rows in (-1, 0): clamped to row 0, still counted', (2.0, -1.5, 30.0, 12.5)),   # y start -0.875, bin 0.5
    ('sample exactly at y == -1 is inside (y < -1 is the test)', (2.0, -2.5, 30.0, 11.5)),    # y start -1.125, bin 0.5: first sample row at -1.0
    ('samples beyond y < -1 drop to zero', (2.0, -16.0, 30.0, 12.0)),
    ('far edge: y in [H-1, H] collapses onto the last row', (2.0, 50.0, 30.0, 64.0)),        # y start 12, ends 15.5
    ('sample exactly at y == H is inside (y > H is the test)', (2.0, 39.0, 30.0, 67.0)),     # y start 9.25, bin 1.0: last sample row at 16.0
    ('samples beyond y > H drop to zero', (2.0, 50.0, 30.0, 78.0)),
    ('x: left of -1 and right of W in one box', (-20.0, 10.0, 84.0, 38.0)),
    ('zero-area box: every sample at one point', (22.0, 26.0, 22.0, 26.0)),
    ('inverted box (x2 < x1, y2 < y1): negative bin size, samples walk backwards', (38.0, 42.0, 10.0, 14.0)),
    ('box entirely outside the map', (-300.0, -300.0, -200.0, -200.0)),
    ('one-pixel box', (31.0, 31.0, 32.0, 32.0)),
]

# level routing (single_level_roi_extractor.py:36-55): floor(log2(sqrt(w*h)/56 + 1e-6)) clamped to [0, 3]
LEVEL_EDGE_BOXES = [
    ((0.0, 0.0, 111.75, 111.75), 0), ((0.0, 0.0, 112.0, 112.0), 1), ((8.0, 8.0, 120.0, 120.0), 1),
    ((0.0, 0.0, 223.75, 223.75), 1), ((0.0, 0.0, 224.0, 224.0), 2), ((0.0, 0.0, 447.75, 447.75), 2),
    ((0.0, 0.0, 448.0, 448.0), 3), ((0.0, 0.0, 2000.0, 2000.0), 3), ((0.0, 0.0, 56.0, 224.0), 1), ((0.0, 0.0, 28.0, 447.0), 0),
    ((5.0, 5.0, 5.0, 900.0), 0),   # zero area: sqrt(0) -> log2(1e-6) -> clamped to level 0
]


def affine_map(H, W, base=0.0, channels=1):
    y, x = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    m = base + 10.0 * y + x
    return np.stack([m + 1000.0 * c for c in range(channels)])[None].astype(np.float32)   # [1, C, H, W]


def expected_affine(box, stride, H, W, base=0.0, channels=1):
    """Closed-form RoIAlign output [C, 7, 7] on affine_map for one box (float64 arithmetic on binary-exact inputs)."""
    x1, y1, x2, y2 = (float(v) / stride - 0.5 for v in box)
    bw, bh = (x2 - x1) / 7.0, (y2 - y1) / 7.0
    out = np.zeros((channels, 7, 7))
    for ph in range(7):
        for pw in range(7):
            acc = np.zeros(channels)
            for iy in range(2):
                y = y1 + ph * bh + (iy + 0.5) * bh / 2.0
                for ix in range(2):
                    x = x1 + pw * bw + (ix + 0.5) * bw / 2.0
                    if y < -1.0 or y > H or x < -1.0 or x > W:
                        continue
                    acc += base + 10.0 * min(max(y, 0.0), H - 1.0) + min(max(x, 0.0), W - 1.0) + 1000.0 * np.arange(channels)
            out[:, ph, pw] = acc / 4.0
    return out


# A NON-affine map with numbers worked out by hand (v[y][x] = (4y + x)^2 on a 4 x 4 map, stride 4):
#   box (2, 2, 30, 30) -> start 0, end 7, bin 1.0; bin (0,0) samples (.25,.25) (.25,.75) (.75,.25) (.75,.75) with corners 0, 1, 16, 25:
#     .5625*0 + .1875*1 + .1875*16 + .0625*25 = 4.75 ;  .1875*0 + .5625*1 + .0625*16 + .1875*25 = 6.25
#     .1875*0 + .0625*1 + .5625*16 + .1875*25 = 13.75 ; .0625*0 + .1875*1 + .1875*16 + .5625*25 = 17.25   -> mean 10.5
#   bin (3,3): samples at 3.25 / 3.75 >= H-1 collapse onto v[3][3] = 225 ; bin (4,4): 4.25 > H -> all four samples 0 ;
#   bin (3,4): rows collapse onto row 3, x = 4.25 / 4.75 > W -> 0 ; bin (0,3): rows .25/.75 between v[0][3] = 9 and v[1][3] = 49,
#     x collapsed onto column 3: (9 + .25*40 + 9 + .75*40) / 2 = 29.0
SQUARE_MAP = np.array([[(4 * y + x) ** 2 for x in range(4)] for y in range(4)], dtype=np.float32)[None, None]
SQUARE_BOX = (2.0, 2.0, 30.0, 30.0)
SQUARE_HAND = {(0, 0): 10.5, (3, 3): 225.0, (4, 4): 0.0, (3, 4): 0.0, (0, 3): 29.0, (6, 6): 0.0}
