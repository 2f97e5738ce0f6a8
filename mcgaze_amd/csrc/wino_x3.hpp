// 3x3 / stride 1 / pad 1 convolution of the MCG_F16X3 engine as a ONE-DIMENSIONAL Winograd F(2,3) contraction along x with the three
// y taps walked directly (fpn.py:178-180 output convs, resnet.py:263-302 conv2 of layer3): 6 matrix products per output pixel and
// channel pair instead of 9, in the same split arithmetic as igemm_dma.hpp's X3 mode (f32 activations, fp16 high / low weight parts,
// three v_mfma_f32_32x32x16_f16 per product).
//
//   output pair (x0, x0 + 1) of a row, input pixels d0..d3 = x0 - 1 .. x0 + 2 of row y + ky - 1, taps g0..g2 of kernel row ky:
//     V0 = d0 - d2   U0 = g0                  M_nu = sum over (ky, ci) V_nu . U_nu        (four accumulators per pair and channel)
//     V1 = d1 + d2   U1 = (g0 + g1 + g2) / 2  y(x0)     = M0 + M1 + M2
//     V2 = d1 - d2   U2 = -(g0 - g1 + g2) / 2 y(x0 + 1) = M1 - M2 - M3
//     V3 = d1 - d3   U3 = g2                  (V2 / U2 carry the sign flip of the textbook form so that three of the four input
//                                              transforms are the same subtraction)
//   U is computed in f64 and split into fp16 high / low parts at pack time (packing.py::wino_pack); V is one f32 VALU operation on
//   two LDS reads, then the contraction kernel's in-register split.  The error is that of the direct kernel with operands up to twice as
//   large as the result's terms (parity: tests/test_gpu_kernels.py::test_conv3x3_wino_x3).
//
// Why 1-D and not F(2x2, 3x3) (DESIGN.md 3.1h has the numbers): the 2-D form needs 16 accumulators per 2x2 output
// tile alive through the whole K loop.  With 64 .. 96 Ki accumulator registers per CU that is a tile of at most 64 x 64 (tiles x
// channels): 440 bytes of operands per MFMA against the direct kernel's 170 (L2 -> LDS 14 TB/s at the present matrix-pipe rate), and
// the input transform + split costs 512 / Nt = 8 VALU per MFMA at Nt = 64, more than a wave can issue beside them.  The 1-D form keeps
// 4 accumulators per PAIR: a 128-pair x 128-channel tile, 2.7 VALU and 220 bytes per MFMA, one workgroup of 8 waves per CU like tile 50.
//
// Structure (the large tile; wino_x3_kernel<NB, RH, RT, CT> has RH x RT row tiles of 32 pairs and CT column tiles of 32 channels).  A
// workgroup owns 128 consecutive output pairs in raster order over (frame, row, pair) and 128 output channels; wave w computes
// transform position nu = w & 3 for the row tiles 2 (w >> 2), 2 (w >> 2) + 1 (2 x 4 MFMA tiles = 128 accumulators).  Grids of fewer than
// ~130 such workgroups (a single clip's P3 / P4 / layer3) run 64 x 64 or 32 x 64 tiles: more workgroups, the same arithmetic per output
// -- K order, positions, output transform -- so every tile shape gives the same bits (test_conv3x3_wino_x3_tiles_are_bit_identical) and
// a clip's result does not depend on the batch it came in.
//   * WINDOW: the input rows the tile touches (its rows, one halo row above and below, ONE shared zero row between two frames, zero
//     columns left and right) are staged per 16-channel slice by LDS-DMA into one of two window buffers and serve all three y taps
//     and all four positions -- the A operand is read from HBM / L2 once per workgroup instead of once per tap.  A window row is a
//     sequence of 1 KiB blocks of 16 pixels = 8 even pixels x 64 B, then 8 odd pixels x 64 B, the four 16-byte chunks of a pixel XOR-
//     swizzled with (pixel pair index >> 2) & 3: the lanes of a fragment read (consecutive pairs = every second pixel) hit 16
//     distinct 16-byte bank groups, and a DMA piece still fetches whole 64-byte pixel slices.
//   * WEIGHTS: per K step (16 channels of one y tap) one 32 KiB stage [nu][channel tile][high, low][lane][16 B], MFMA-fragment-major
//     and contiguous in global memory in consumption order, through a two-stage ring.
//   * one s_barrier per K step; everything staged is waited for with vmcnt(0) (two-stage ring: nothing else is in flight).
//   * epilogue: the four positions of a pair live in four waves; they meet in LDS (two passes of 64 pairs x 128 channels x 4), the
//     output transform + bias (+ ReLU) is applied on 16-byte channel chunks and both pixels of a pair are stored.
// Batch invariance: a pair's arithmetic does not depend on the tile it falls into, so a clip's result is independent of the batch.
#pragma once
#include "igemm_dma.hpp"

namespace wnx {
constexpr int KS = 16;                              // input channels per K step
constexpr int UNT = 128;                            // output channels per weight block of the packed operand (packing.py::wino_pack)
constexpr int USTAGE = 4 * 4 * 2 * 1024;            // one K step of one block: [nu 4][channel tile 4][high, low][64 lanes][16 B]
constexpr int WIN_CAP = 48 * 1024;                  // one window buffer (pieces of 1 KiB; a workgroup may use fewer)
}  // namespace wnx

struct WinoParams {
  const float* x;        // [frames][H][W][Cin] f32
  const void* u;         // packing.py::wino_pack: fp16 [Cout / 128][3 Cin / 16 K steps][nu][ct][high, low][lane][8]
  const float* bias;     // [Cout] or NULL
  float* y;              // [frames][H][W][Cout] f32
  int H, W, PW, frames, Cin, Cout, relu;
  int total_pairs, n_tiles;
  float wscale;          // the transformed sums are multiplied by this power of two before the bias (pre-scaled weights); 0 = 1
};

// NB: 1 KiB blocks per window row = ceil((2 PW + 2) / 16).  Tile: RH x RT row tiles of 32 pairs, CT column tiles of 32 channels; 4 RH waves.
template <int NB, int RH, int RT, int CT>
__global__ __launch_bounds__(256 * RH, 1) void wino_x3_kernel(const WinoParams p) {
  using namespace wnx;
  constexpr int NW = 4 * RH, NTHREADS = 64 * NW, MT = 32 * RH * RT, NT = 32 * CT;
  constexpr int ROWB = NB * 1024;
  constexpr int BSTAGE = 4 * CT * 2 * 1024;           // this workgroup's share of a K step's weights
  constexpr int BPW = 8 * CT / NW;                    // weight pieces per wave and K step
  constexpr int MAXP = WIN_CAP / 1024 / NW;           // window pieces per wave and slice (upper bound)
  static_assert(8 * CT % NW == 0 && UNT % NT == 0, "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_win = smem;
  char* const s_b = smem + 2 * WIN_CAP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = wave & 3, rh = wave >> 2;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = logical / p.n_tiles, ntile = logical - mt * p.n_tiles;
  const int q0 = mt * MT;
  const int q_last = min(q0 + MT - 1, p.total_pairs - 1);
  const int H1 = p.H + 1;
  // window slot of global output row R = f H + y: R + f + 1 (slot f (H + 1) is the zero row in front of frame f)
  auto slot_of = [&](int q) { const int R = q / p.PW; return R + R / p.H + 1; };
  const int sig_b = __builtin_amdgcn_readfirstlane(slot_of(q0) - 1);
  const int NP = __builtin_amdgcn_readfirstlane((slot_of(q_last) - sig_b + 2) * NB);   // window pieces per slice
  const int NSL = p.Cin / KS, KT = 3 * NSL;

  const int n0 = ntile * NT;                                                    // first output channel of this workgroup
  const u32x4 srd_x = make_srd(p.x);
  const u32x4 srd_u = make_srd((const char*)p.u + (size_t)(n0 / UNT) * KT * USTAGE + (size_t)((n0 % UNT) / 32) * 2048);
  const uint32_t lds_win = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)s_win;
  const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)s_b;

  // ---- this wave's window pieces: piece pi = wave + NW n covers block pi % NB of window row pi / NB (uniform, SGPRs)
  uint32_t prow[MAXP];
  int pblk[MAXP];
  bool pok[MAXP], prowok[MAXP];
  static_for<MAXP>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    const int pi = wave + NW * n, j = pi / NB, b = pi - j * NB;
    const int sg = sig_b + j, f = sg / H1, r = sg - f * H1;
    pok[n] = pi < NP;
    prowok[n] = __builtin_amdgcn_readfirstlane((r != 0 && f < p.frames) ? 1 : 0) != 0;
    prow[n] = __builtin_amdgcn_readfirstlane((uint32_t)(((long long)(f * p.H + r - 1) * p.W) * p.Cin * 4));
    pblk[n] = b;
  });
  // lane l of a piece: parity l >> 5, pixel pair (l >> 2) & 7 of the block, LDS chunk slot l & 3 (holds channel chunk slot ^ swizzle)
  const int l_par = lane >> 5, l_i = (lane >> 2) & 7, l_cs = lane & 3;
  auto win_voff = [&](int b) -> uint32_t {
    const int x = 16 * b + 2 * l_i + l_par - 1;
    const int c = l_cs ^ ((2 * b + (l_i >> 2)) & 3);
    return (unsigned)x < (unsigned)p.W ? (uint32_t)((x * p.Cin + 4 * c) * 4) : MCG_OOB_OFFSET;
  };
  auto issue_window = [&](int cs, uint32_t dst) {
    static_for<MAXP>([&](auto nc) {
      constexpr int n = decltype(nc)::value;
      if (pok[n]) {
        const uint32_t v = prowok[n] ? win_voff(pblk[n]) : MCG_OOB_OFFSET;
        lds_dma16<0>(v, srd_x, prow[n] + (uint32_t)cs * (KS * 4), dst + (uint32_t)(wave + NW * n) * 1024u);
      }
    });
  };
  // weight pieces: piece pc = wave BPW + i of the stage [nu][CT][high, low] <- the packed block's [nu][4][high, low] (srd_u starts at this
  // workgroup's first channel tile)
  const uint32_t b_voff = (uint32_t)lane * 16u;
  auto issue_b = [&](int k, uint32_t dst) {
    static_for<BPW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      const int pc = wave * BPW + i, pnu = pc / (2 * CT), rest = pc - pnu * (2 * CT);
      lds_dma16<i * 1024>(b_voff, srd_u, (uint32_t)k * USTAGE + (uint32_t)pnu * 8192u + (uint32_t)rest * 1024u, dst + (uint32_t)wave * (BPW * 1024u));
    });
  };

  // ---- A operand addresses: pair (lane & 31) of row tile RT rh + rt, channels 8 (lane >> 5) .. + 7 of the slice = chunks 2 h, 2 h + 1
  const int pl = lane & 31, h = lane >> 5;
  const int o_a = nu == 0 ? 0 : 1, o_b = nu == 3 ? 3 : 2;
  const float sgn = nu == 1 ? 1.f : -1.f;            // V = a + sgn b
  const char* ap[RT][2][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int q = min(q0 + (RT * rh + rt) * 32 + pl, p.total_pairs - 1);   // pairs beyond the end repeat the last one; never stored
    const int R = q / p.PW, xp = q - R * p.PW;
    const int jrow = (R + R / p.H + 1) - sig_b - 1;                          // window row of tap ky = 0
#pragma unroll
    for (int px = 0; px < 2; ++px) {
      const int wx = 2 * xp + (px ? o_b : o_a);
      const int pp = wx >> 1, par = wx & 1, swz = (pp >> 2) & 3;
      const int off = jrow * ROWB + (pp >> 3) * 1024 + par * 512 + (pp & 7) * 64;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) ap[rt][px][ch] = s_win + off + (((2 * h + ch) ^ swz) << 4);
    }
  }
  const char* const bp = s_b + nu * (CT * 2 * 1024) + lane * 16;

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A fragments of one K step: two LDS reads per pixel, V = a + sgn b, the contraction kernel's in-register split
  auto prep = [&](auto wbc, auto kyc, bf16x8 (&ah)[RT], bf16x8 (&al)[RT]) {
    constexpr int AOFF = decltype(wbc)::value * WIN_CAP + decltype(kyc)::value * ROWB;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      const uint4 a0 = *(const uint4*)(ap[rt][0][0] + AOFF), a1 = *(const uint4*)(ap[rt][0][1] + AOFF);
      const uint4 b0 = *(const uint4*)(ap[rt][1][0] + AOFF), b1 = *(const uint4*)(ap[rt][1][1] + AOFF);
      uint4 v0, v1;
      v0.x = __float_as_uint(fmaf(sgn, __uint_as_float(b0.x), __uint_as_float(a0.x)));
      v0.y = __float_as_uint(fmaf(sgn, __uint_as_float(b0.y), __uint_as_float(a0.y)));
      v0.z = __float_as_uint(fmaf(sgn, __uint_as_float(b0.z), __uint_as_float(a0.z)));
      v0.w = __float_as_uint(fmaf(sgn, __uint_as_float(b0.w), __uint_as_float(a0.w)));
      v1.x = __float_as_uint(fmaf(sgn, __uint_as_float(b1.x), __uint_as_float(a1.x)));
      v1.y = __float_as_uint(fmaf(sgn, __uint_as_float(b1.y), __uint_as_float(a1.y)));
      v1.z = __float_as_uint(fmaf(sgn, __uint_as_float(b1.z), __uint_as_float(a1.z)));
      v1.w = __float_as_uint(fmaf(sgn, __uint_as_float(b1.w), __uint_as_float(a1.w)));
      split_f32x8(v0, v1, ah[rt], al[rt]);
    }
  };
  auto mma = [&](auto slc, const bf16x8 (&ah)[RT], const bf16x8 (&al)[RT]) {
    constexpr int BOFF = decltype(slc)::value * BSTAGE;
    bf16x8 bh[CT], bl[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      bh[ct] = __builtin_bit_cast(bf16x8, *(const uint4*)(bp + BOFF + ct * 2048));
      bl[ct] = __builtin_bit_cast(bf16x8, *(const uint4*)(bp + BOFF + ct * 2048 + 1024));
    }
    // small terms first; consecutive MFMAs never share an accumulator (igemm_dma.hpp, X3 mode: the same order)
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[i][j] = x3_mfma(al[i], bh[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[i][j] = x3_mfma(ah[i], bl[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
      for (int j = 0; j < CT; ++j) acc[i][j] = x3_mfma(ah[i], bh[j], acc[i][j]);
  };

  // ---- K loop: step k = 3 cs + ky (slice cs of 16 channels, y tap ky); weights of step k in ring stage k & 1, window slice cs in
  // buffer cs & 1.  Unrolled by six steps (two slices) so that stage, buffer and tap are immediates.  After a barrier every wave
  // of the workgroup stands at the same instruction, so whatever a step does before its MFMAs idles the matrix pipe of all four
  // SIMDs at once: the A fragments of step k + 1 are therefore prepared UNDER step k's MFMAs (the window slice of step k + 1 is
  // resident by then: a slice is issued at its predecessor's first step and waited for -- vmcnt(0) -- at the second).
  issue_window(0, lds_win);
  issue_b(0, lds_b);
  bf16x8 fh[2][RT], fl[2][RT];                            // [step parity][row tile]
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  prep(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, fh[0], fl[0]);
  constexpr int NM = 3 * RT * CT, VPM = (32 * RT + NM - 4) / (NM - 3);   // MFMAs per step; VALU issued behind each of the first NM - 3
#pragma nounroll
  for (int s2 = 0; s2 < NSL / 2; ++s2) {
    static_for<6>([&](auto uc) {
      constexpr int U = decltype(uc)::value, KY = U % 3, SL = U & 1, WB = U / 3;
      constexpr int U1 = (U + 1) % 6, KY1 = U1 % 3, WB1 = U1 / 3;
      const int k = 6 * s2 + U, cs = 2 * s2 + WB;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces landed; its reads of the previous step are done
      __builtin_amdgcn_s_barrier();                                  // ... and everyone's
      issue_b(min(k + 1, KT - 1), lds_b + (SL ^ 1) * BSTAGE);       // (the last step re-fetches its own stage: no branch in the step body)
      if (KY == 0 && cs + 1 < NSL) issue_window(cs + 1, lds_win + (WB ^ 1) * WIN_CAP);
      // one basic block from here to the next barrier: this step's MFMAs with the next step's A preparation (4 RT LDS reads, 32 RT VALU)
      // issued between them -- the last step prepares fragments nobody uses (valid addresses) rather than branch
      prep(std::integral_constant<int, WB1>{}, std::integral_constant<int, KY1>{}, fh[SL ^ 1], fl[SL ^ 1]);
      mma(std::integral_constant<int, SL>{}, fh[SL], fl[SL]);
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * CT + 4 * RT, 0);   // the B fragments of this step, then the raw A pixels of the next
      static_for<NM - 3>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      });
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    });
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: output transform across the four positions through LDS, one pass per wave row group (RT x 32 pairs)
  constexpr int PP = 32 * RT, CPR = NT / 4;                // pairs per pass, 16-byte chunks per row
  static_assert(4 * PP * NT * 4 <= 2 * WIN_CAP + 2 * BSTAGE && NTHREADS % CPR == 0 && (PP * CPR) % NTHREADS == 0, "epilogue staging");
  float* const C = (float*)smem;                           // [nu][PP pairs][NT]
  const int ch4 = (tid % CPR) * 4;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bv = *(const float4*)(p.bias + n0 + ch4);
  const float wsc = p.wscale > 0.f ? p.wscale : 1.f;
#pragma unroll 1
  for (int pass = 0; pass < RH; ++pass) {
    if (rh == pass) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            C[(nu * PP + rt * 32 + mfma32_row(r, lane)) * NT + ct * 32 + (lane & 31)] = acc[rt][ct][r];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < PP * CPR / NTHREADS; ++it) {
      const int prl = tid / CPR + it * (NTHREADS / CPR);
      const int q = q0 + pass * PP + prl;
      if (q < p.total_pairs) {
        const float4 m0 = *(const float4*)(C + (0 * PP + prl) * NT + ch4), m1 = *(const float4*)(C + (1 * PP + prl) * NT + ch4);
        const float4 m2 = *(const float4*)(C + (2 * PP + prl) * NT + ch4), m3 = *(const float4*)(C + (3 * PP + prl) * NT + ch4);
        float4 y0 = make_float4(((m0.x + m1.x) + m2.x) * wsc + bv.x, ((m0.y + m1.y) + m2.y) * wsc + bv.y, ((m0.z + m1.z) + m2.z) * wsc + bv.z, ((m0.w + m1.w) + m2.w) * wsc + bv.w);
        float4 y1 = make_float4(((m1.x - m2.x) - m3.x) * wsc + bv.x, ((m1.y - m2.y) - m3.y) * wsc + bv.y, ((m1.z - m2.z) - m3.z) * wsc + bv.z, ((m1.w - m2.w) - m3.w) * wsc + bv.w);
        if (p.relu) {
          y0.x = fmaxf(y0.x, 0.f); y0.y = fmaxf(y0.y, 0.f); y0.z = fmaxf(y0.z, 0.f); y0.w = fmaxf(y0.w, 0.f);
          y1.x = fmaxf(y1.x, 0.f); y1.y = fmaxf(y1.y, 0.f); y1.z = fmaxf(y1.z, 0.f); y1.w = fmaxf(y1.w, 0.f);
        }
        const int R = q / p.PW, xo = 2 * (q - R * p.PW);
        float* yp = p.y + ((long long)R * p.W + xo) * p.Cout + n0 + ch4;
        *(float4*)yp = y0;
        if (xo + 1 < p.W) *(float4*)(yp + p.Cout) = y1;
      }
    }
    __syncthreads();
  }
}

// Tile shapes: 0 = 128 pairs x 128 channels (8 waves), 1 = 64 x 64 (8 waves), 2 = 32 x 64 (4 waves)
static inline int wino_x3_tile_pairs(int shape) { return shape == 0 ? 128 : (shape == 1 ? 64 : 32); }
static inline int wino_x3_tile_channels(int shape) { return shape == 0 ? 128 : 64; }
// blocks per window row and the largest window (rows) a tile of mt pairs can need: its rows, the zero rows between frames, two halo rows
static inline int wino_x3_blocks(int W) { return (2 * ((W + 1) / 2) + 2 + 15) / 16; }
static inline int wino_x3_max_rows(int H, int W, int mt) {
  const int PW = (W + 1) / 2;
  const int rows = (PW - 1 + mt - 1) / PW + 1;
  const int cross = rows >= 2 ? (rows - 2) / H + 1 : 0;
  return rows + cross + 2;
}
// 3x3 / stride 1 / pad 1, f16x3: channel counts the tiles divide, a window that fits its buffer, operands inside the 2 GiB descriptor.
// Decided by the LAYER's shape only (never by the batch): a layer either runs this arithmetic or the direct kernel's for every batch size.
static inline bool wino_x3_applicable(int frames, int H, int W, int Cin, int Cout) {
  if (Cin % 32 != 0 || Cout % wnx::UNT != 0 || H < 1 || W < 2 || frames < 1) return false;
  int nb = wino_x3_blocks(W);
  if (nb == 3) nb = 4;
  if (nb > 4) return false;
  if (wino_x3_max_rows(H, W, 128) * nb * 1024 > wnx::WIN_CAP) return false;
  const long long px = (long long)frames * H * W;
  return px * Cin * 4 < MCG_DMA_MAX_BYTES && px * Cout * 4 < MCG_DMA_MAX_BYTES && px < 0x7fffffffLL;
}
static inline size_t wino_x3_weight_bytes(int Cin, int Cout) { return (size_t)(Cout / wnx::UNT) * (3 * Cin / wnx::KS) * wnx::USTAGE; }

template <int NB, int RH, int RT, int CT>
static inline int launch_wino_x3_t(hipStream_t s, const WinoParams& p, int grid) {
  constexpr int kLds = 2 * wnx::WIN_CAP + 2 * (4 * CT * 2 * 1024);
  static bool raised[MCG_MAX_DEVICES] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!raised[dev]) {
    if (hipFuncSetAttribute((const void*)wino_x3_kernel<NB, RH, RT, CT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
    raised[dev] = true;
  }
  hipLaunchKernelGGL((wino_x3_kernel<NB, RH, RT, CT>), dim3(grid), dim3(256 * RH), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int NB>
static inline int launch_wino_x3_nb(hipStream_t s, const WinoParams& p, int shape, int grid) {
  if (shape == 0) return launch_wino_x3_t<NB, 2, 2, 4>(s, p, grid);
  if (shape == 1) return launch_wino_x3_t<NB, 2, 1, 2>(s, p, grid);
  return launch_wino_x3_t<NB, 1, 1, 2>(s, p, grid);
}
// returns 0 on success; the caller has checked wino_x3_applicable.  shape: -1 = by grid size (the largest tile that still makes ~half
// a chip's worth of workgroups), else forced (tests)
static const int kWinoMinGrid = 130;
static inline int launch_wino_x3(hipStream_t s, WinoParams p, int shape = -1) {
  p.PW = (p.W + 1) / 2;
  p.total_pairs = p.frames * p.H * p.PW;
  auto grid_of = [&](int sh) { return ((p.total_pairs + wino_x3_tile_pairs(sh) - 1) / wino_x3_tile_pairs(sh)) * (p.Cout / wino_x3_tile_channels(sh)); };
  if (shape < 0) shape = grid_of(0) >= kWinoMinGrid ? 0 : (grid_of(1) >= kWinoMinGrid ? 1 : 2);
  p.n_tiles = p.Cout / wino_x3_tile_channels(shape);
  const int grid = grid_of(shape);
  const int nb = wino_x3_blocks(p.W);
  if (nb == 1) return launch_wino_x3_nb<1>(s, p, shape, grid);
  if (nb == 2) return launch_wino_x3_nb<2>(s, p, shape, grid);
  return launch_wino_x3_nb<4>(s, p, shape, grid);
}
