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
// Structure (the large tile; wino_x3_kernel<NB, RH, RT, CT, G> has RH x RT row tiles of 32 pairs and CT column tiles of 32 channels).  A
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
//   * one s_barrier per K step; everything staged is waited for with vmcnt(0) (two-stage ring: nothing else is in flight).  A step starts its
//     MFMAs right behind the barrier and issues the next stage's DMA pieces and the next step's A preparation behind them: the kernel is bound
//     by what two in-order waves per SIMD can ISSUE (measured: MFMA skeleton 2.0 ms + DMA issue 0.7 + transform 0.37 on the FPN P2 conv), not
//     by the matrix pipe -- DESIGN.md 3.1h.  Tried and not kept: transposed MFMAs with 16-byte epilogue stores (neutral), one M0 write per four
//     pieces (neutral), opposite phase orders for the two waves of a SIMD (slower).
//   * epilogue: the four positions of a pair live in four waves; they meet in LDS (two passes of 64 pairs x 128 channels x 4), the
//     output transform + bias (+ ReLU) is applied on 16-byte channel chunks and both pixels of a pair are stored.
// Batch invariance: a pair's arithmetic does not depend on the tile it falls into, so a clip's result is independent of the batch.
//
// G = 4 (round 4, second half): the same kernel with F(4,3) -- groups of FOUR output pixels, six positions (points 0, +-1, +-2, inf),
//     V = B^T d with B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]   (<= 4 terms: 1 mul + 3 fma)
//     U = G g,  G = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//     y = A^T M, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// 4.5 instead of 6 products per output.  Six positions = 12 waves of one row tile (64 groups x 128 channels) or 6 waves (32 x 64); a K step's
// weights are 48 KiB, which leaves 2 x 32 KiB of window: where a tile that straddles two frames would need more (56-pixel maps), tiles are
// cut at frame boundaries (tiles_per_frame).  The window's 1 KiB pieces hold two PHASES (pixel index mod G) of eight groups each; for G = 2
// that is the even / odd layout above.  Accuracy: the transform constants cost ~1.5 bits (emulated 1.9e-6 of scale against 6e-7 for the direct
// kernel) -- acceptable only because the weights are pre-scaled (without: 1.5e-5); DESIGN.md 3.1h.
#pragma once
#include "igemm_dma.hpp"

namespace wnx {
constexpr int KS = 16;                              // input channels per K step
constexpr int UNT = 128;                            // output channels per weight block of the packed operand (packing.py::wino_pack)
constexpr int ustage(int g) { return (g + 2) * 4 * 2 * 1024; }   // one K step of one block: [nu G + 2][channel tile 4][high, low][64 lanes][16 B]
constexpr int wcap(int g, int ct) { return (g == 4 && ct == 4) ? 32 * 1024 : 48 * 1024; }   // one window buffer (1 KiB pieces; a workgroup may use fewer)
}  // namespace wnx

struct WinoParams {
  const float* x;        // [frames][H][W][Cin] f32
  const void* u;         // packing.py::wino_pack: fp16 [Cout / 128][3 Cin / 16 K steps][nu][ct][high, low][lane][8]
  const float* bias;     // [Cout] or NULL
  float* y;              // [frames][H][W][Cout] f32
  int H, W, PW, frames, Cin, Cout, relu;   // PW: groups (of G output pixels) per row
  int total_pairs, n_tiles;                // groups in all frames; channel tiles
  int tpf, gpf;                            // tiles per frame (0: tiles run over frame boundaries), groups per frame
  float wscale;          // the transformed sums are multiplied by this power of two before the bias (pre-scaled weights); 0 = 1
};

// NB: 1 KiB pieces per window row.  Tile: RH x RT row tiles of 32 groups, CT column tiles of 32 channels; (G + 2) RH waves.
template <int NB, int RH, int RT, int CT, int G>
__global__ __launch_bounds__(64 * (G + 2) * RH, 1) void wino_x3_kernel(const WinoParams p) {
  using namespace wnx;
  constexpr int NV = G + 2, NW = NV * RH, NTHREADS = 64 * NW, MT = 32 * RH * RT, NT = 32 * CT;
  constexpr int ROWB = NB * 1024, WIN_CAP = wcap(G, CT), USTAGE = ustage(G);
  constexpr int BSTAGE = NV * CT * 2 * 1024;          // this workgroup's share of a K step's weights
  constexpr int BPW = 2 * NV * CT / NW;               // weight pieces per wave and K step
  constexpr int MAXP = (WIN_CAP / 1024 + NW - 1) / NW;   // window pieces per wave and slice (upper bound)
  constexpr int NTERM = G == 2 ? 2 : 4;               // input pixels per transformed value
  static_assert((2 * NV * CT) % NW == 0 && UNT % NT == 0 && (G == 2 || G == 4), "tile shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_win = smem;
  char* const s_b = smem + 2 * WIN_CAP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = wave % NV, rh = wave / NV;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = logical / p.n_tiles, ntile = logical - mt * p.n_tiles;
  int q0, q_end;
  if (p.tpf > 0) {                                    // tiles cut at frame boundaries
    const int f = mt / p.tpf, t = mt - f * p.tpf;
    q0 = f * p.gpf + t * MT;
    q_end = min(q0 + MT, (f + 1) * p.gpf);
  } else {
    q0 = mt * MT;
    q_end = min(q0 + MT, p.total_pairs);
  }
  const int q_last = q_end - 1;
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

  // ---- this wave's window pieces: piece pi = wave + NW n covers piece pi % NB of window row pi / NB (uniform, SGPRs)
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
  // A window row is a sequence of blocks of 8 G pixels: phase ph = pixel % G, group index i = pixel / G: [ph][i (8)][64 B].  A 1 KiB piece
  // carries two phases of a block (G = 2: the whole block); lane l: phase 2 (piece % (G / 2)) + (l >> 5), group (l >> 2) & 7 of the block,
  // LDS chunk slot l & 3 (holds channel chunk slot ^ swizzle).
  const int l_ph = lane >> 5, l_i = (lane >> 2) & 7, l_cs = lane & 3;
  auto win_voff = [&](int pc) -> uint32_t {
    const int b = pc / (G / 2), hph = pc - b * (G / 2);
    const int x = (8 * b + l_i) * G + 2 * hph + l_ph - 1;
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

  // ---- input transform of this wave's position: V = sum_t coef[t] * d[off[t]] (pixel offsets inside the group's G + 2 pixel footprint)
  int toff[NTERM];
  float tco[NTERM];
  if constexpr (G == 2) {
    toff[0] = nu == 0 ? 0 : 1; toff[1] = nu == 3 ? 3 : 2;
    tco[0] = 1.f; tco[1] = nu == 1 ? 1.f : -1.f;            // V = a + sgn b
  } else {
    const int o4[6][4] = {{0, 2, 4, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 2, 3, 4}, {1, 3, 5, 5}};
    const float c4[6][4] = {{4.f, -5.f, 1.f, 0.f}, {-4.f, -4.f, 1.f, 1.f}, {4.f, -4.f, -1.f, 1.f}, {-2.f, -1.f, 2.f, 1.f}, {2.f, -1.f, -2.f, 1.f}, {4.f, -5.f, 1.f, 0.f}};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      toff[t] = o4[0][t]; tco[t] = c4[0][t];
#pragma unroll
      for (int v = 1; v < 6; ++v)
        if (nu == v) { toff[t] = o4[v][t]; tco[t] = c4[v][t]; }
    }
  }
  // ---- A operand addresses: group (lane & 31) of row tile RT rh + rt, channels 8 (lane >> 5) .. + 7 of the slice = chunks 2 h, 2 h + 1
  const int pl = lane & 31, h = lane >> 5;
  const char* ap[RT][NTERM][2];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int q = min(q0 + (RT * rh + rt) * 32 + pl, q_last);               // groups beyond the end repeat the last one; never stored
    const int R = q / p.PW, xg = q - R * p.PW;
    const int jrow = (R + R / p.H + 1) - sig_b - 1;                          // window row of tap ky = 0
#pragma unroll
    for (int t = 0; t < NTERM; ++t) {
      const int wx = G * xg + toff[t];
      const int pp = wx / G, ph = wx - pp * G, swz = (pp >> 2) & 3;
      const int off = jrow * ROWB + (pp >> 3) * (G * 512) + ph * 512 + (pp & 7) * 64;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) ap[rt][t][ch] = s_win + off + (((2 * h + ch) ^ swz) << 4);
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

  // A fragments of one K step, in two halves: the raw pixel reads (LDS), then the f32 transform + the contraction kernel's in-register split
  struct Raw { uint4 d[RT][NTERM][2]; };
  auto load_raw = [&](auto wbc, auto kyc, Raw& r) {
    constexpr int AOFF = decltype(wbc)::value * WIN_CAP + decltype(kyc)::value * ROWB;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < NTERM; ++t) { r.d[rt][t][0] = *(const uint4*)(ap[rt][t][0] + AOFF); r.d[rt][t][1] = *(const uint4*)(ap[rt][t][1] + AOFF); }
  };
  auto transform = [&](const Raw& r, bf16x8 (&ah)[RT], bf16x8 (&al)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t e0[4] = {r.d[rt][0][c].x, r.d[rt][0][c].y, r.d[rt][0][c].z, r.d[rt][0][c].w};
        const uint32_t e1[4] = {r.d[rt][1][c].x, r.d[rt][1][c].y, r.d[rt][1][c].z, r.d[rt][1][c].w};
        if constexpr (G == 2) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * c + e] = fmaf(tco[1], __uint_as_float(e1[e]), __uint_as_float(e0[e]));     // a + sgn b
        } else {
          const uint32_t e2[4] = {r.d[rt][2][c].x, r.d[rt][2][c].y, r.d[rt][2][c].z, r.d[rt][2][c].w};
          const uint32_t e3[4] = {r.d[rt][3][c].x, r.d[rt][3][c].y, r.d[rt][3][c].z, r.d[rt][3][c].w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[4 * c + e] = fmaf(tco[3], __uint_as_float(e3[e]), fmaf(tco[2], __uint_as_float(e2[e]), fmaf(tco[1], __uint_as_float(e1[e]), tco[0] * __uint_as_float(e0[e]))));
        }
      }
      const uint4 v0 = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
      const uint4 v1 = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
      split_f32x8(v0, v1, ah[rt], al[rt]);
    }
  };

  // ---- K loop: step k = 3 cs + ky (slice cs of 16 channels, y tap ky); weights of step k in ring stage k & 1, window slice cs in
  // buffer cs & 1.  Unrolled by six steps (two slices) so that stage, buffer and tap are immediates.  After a barrier every wave
  // of the workgroup stands at the same instruction, so whatever a step does BEFORE its MFMAs idles the matrix pipe of all four SIMDs
  // at once.  A step therefore starts its MFMAs as soon as its weight fragments are read, and everything that serves the NEXT step is
  // issued behind them: the LDS-DMA pieces of the next weight stage / window slice one at a time behind the first MFMAs (a piece costs
  // 60 - 180 issue cycles; five to ten of them in front of the MFMAs cost the step a quarter of its time), the next step's A fragments
  // (raw reads, transform, split) under the rest.  (The window slice of step k + 1 is resident by then: a slice is issued at its
  // predecessor's first step and waited for -- vmcnt(0) -- at the second.)
  issue_window(0, lds_win);
  issue_b(0, lds_b);
  bf16x8 fh[2][RT], fl[2][RT];                            // [step parity][row tile]
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    Raw r0;
    load_raw(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r0);
    transform(r0, fh[0], fl[0]);
  }
  constexpr int NM1 = RT * CT;                            // MFMAs of one term
  constexpr int PVALU = (G == 2 ? 32 : 56) * RT, VPM = (PVALU + 2 * NM1 - 1) / (2 * NM1);   // transform + split VALU behind each MFMA of the last two terms
#pragma nounroll
  for (int s2 = 0; s2 < NSL / 2; ++s2) {
    static_for<6>([&](auto uc) {
      constexpr int U = decltype(uc)::value, KY = U % 3, SL = U & 1, WB = U / 3;
      constexpr int U1 = (U + 1) % 6, KY1 = U1 % 3, WB1 = U1 / 3;
      constexpr int BOFF = SL * BSTAGE;
      const int k = 6 * s2 + U, cs = 2 * s2 + WB;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces landed; its reads of the previous step are done
      __builtin_amdgcn_s_barrier();                                  // ... and everyone's
      bf16x8 bh[CT], bl[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        bh[ct] = __builtin_bit_cast(bf16x8, *(const uint4*)(bp + BOFF + ct * 2048));
        bl[ct] = __builtin_bit_cast(bf16x8, *(const uint4*)(bp + BOFF + ct * 2048 + 1024));
      }
      Raw raw;
      load_raw(std::integral_constant<int, WB1>{}, std::integral_constant<int, KY1>{}, raw);   // (the last step reads pixels nobody uses: valid addresses, no branch)
      __builtin_amdgcn_sched_barrier(0);
      // term 1 (activation low x weight high; small terms first, consecutive MFMAs never share an accumulator: igemm_dma.hpp's order), one
      // DMA piece of the next weight stage behind each of its first MFMAs, then the next window slice's pieces
      const uint32_t kn = (uint32_t)min(k + 1, KT - 1) * USTAGE, bdst = lds_b + (SL ^ 1) * BSTAGE + (uint32_t)wave * (BPW * 1024u);
      static_for<NM1>([&](auto mc) {
        constexpr int M = decltype(mc)::value, I = M / CT, J = M % CT;
        acc[I][J] = x3_mfma(fl[SL][I], bh[J], acc[I][J]);
        if constexpr (M < BPW) {                                      // (the last step re-fetches its own stage rather than branch)
          const int pc = wave * BPW + M, pnu = pc / (2 * CT), rest = pc - pnu * (2 * CT);
          lds_dma16<M * 1024>(b_voff, srd_u, kn + (uint32_t)pnu * 8192u + (uint32_t)rest * 1024u, bdst);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      if constexpr (NM1 < BPW) {
        static_for<BPW - NM1>([&](auto mc) {
          constexpr int M = NM1 + decltype(mc)::value;
          const int pc = wave * BPW + M, pnu = pc / (2 * CT), rest = pc - pnu * (2 * CT);
          lds_dma16<M * 1024>(b_voff, srd_u, kn + (uint32_t)pnu * 8192u + (uint32_t)rest * 1024u, bdst);
        });
      }
      if (KY == 0 && cs + 1 < NSL) issue_window(cs + 1, lds_win + (WB ^ 1) * WIN_CAP);
      // terms 2 and 3 in one basic block with the next step's transform + split
      transform(raw, fh[SL ^ 1], fl[SL ^ 1]);
#pragma unroll
      for (int i2 = 0; i2 < RT; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < CT; ++j2) acc[i2][j2] = x3_mfma(fh[SL][i2], bl[j2], acc[i2][j2]);
#pragma unroll
      for (int i2 = 0; i2 < RT; ++i2)
#pragma unroll
        for (int j2 = 0; j2 < CT; ++j2) acc[i2][j2] = x3_mfma(fh[SL][i2], bh[j2], acc[i2][j2]);
      static_for<2 * NM1 - 1>([&](auto) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
      });
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
    });
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- epilogue: output transform across the positions through LDS, one pass per wave row group (RT x 32 groups)
  constexpr int PP = 32 * RT, CPR = NT / 4;                // groups per pass, 16-byte chunks per row
  static_assert(NV * PP * NT * 4 <= 2 * WIN_CAP + 2 * BSTAGE, "epilogue staging");
  float* const C = (float*)smem;                           // [nu][PP groups][NT]
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
    for (int item = tid; item < PP * CPR; item += NTHREADS) {
      const int prl = item / CPR, ch4 = (item - prl * CPR) * 4;
      const int q = q0 + pass * PP + prl;
      if (q >= q_end) continue;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bv = *(const float4*)(p.bias + n0 + ch4);
      float4 m[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) m[v] = *(const float4*)(C + (v * PP + prl) * NT + ch4);
      float4 y[G];
      if constexpr (G == 2) {
        y[0] = make_float4(((m[0].x + m[1].x) + m[2].x) * wsc + bv.x, ((m[0].y + m[1].y) + m[2].y) * wsc + bv.y, ((m[0].z + m[1].z) + m[2].z) * wsc + bv.z, ((m[0].w + m[1].w) + m[2].w) * wsc + bv.w);
        y[1] = make_float4(((m[1].x - m[2].x) - m[3].x) * wsc + bv.x, ((m[1].y - m[2].y) - m[3].y) * wsc + bv.y, ((m[1].z - m[2].z) - m[3].z) * wsc + bv.z, ((m[1].w - m[2].w) - m[3].w) * wsc + bv.w);
      } else {
        // y0 = m0 + (m1 + m2) + (m3 + m4); y1 = (m1 - m2) + 2 (m3 - m4); y2 = (m1 + m2) + 4 (m3 + m4); y3 = (m1 - m2) + 8 (m3 - m4) + m5
#define MCG_W4(F)                                                                                                            \
        {                                                                                                                    \
          const float s12 = m[1].F + m[2].F, d12 = m[1].F - m[2].F, s34 = m[3].F + m[4].F, d34 = m[3].F - m[4].F;            \
          y[0].F = ((m[0].F + s12) + s34) * wsc + bv.F;                                                                      \
          y[1].F = fmaf(2.f, d34, d12) * wsc + bv.F;                                                                         \
          y[2].F = fmaf(4.f, s34, s12) * wsc + bv.F;                                                                         \
          y[3].F = (fmaf(8.f, d34, d12) + m[5].F) * wsc + bv.F;                                                              \
        }
        MCG_W4(x) MCG_W4(y) MCG_W4(z) MCG_W4(w)
#undef MCG_W4
      }
      const int R = q / p.PW, xo = G * (q - R * p.PW);
      float* yp = p.y + ((long long)R * p.W + xo) * p.Cout + n0 + ch4;
#pragma unroll
      for (int j = 0; j < G; ++j) {
        if (xo + j >= p.W) break;
        float4 o = y[j];
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *(float4*)(yp + (long long)j * p.Cout) = o;
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the 128-pair x 128-channel F(2,3) tile with ONE WAVE PER SIMD.  The 8-wave kernel above is bound by what
// two lock-stepped in-order waves per SIMD can issue: every wave issues its share of a K step's 32 KiB weight stage and of the window, reads
// the weight fragments back from LDS, and all eight meet at a barrier every step.  Here a workgroup is four waves, wave nu = transform
// position nu for ALL four row tiles of the tile (4 x 4 MFMA tiles = 256 accumulators: the accumulator half of the SIMD's 512-entry
// register file), and
//   * the WEIGHT fragments of position nu have exactly one reader -- this wave -- so they never touch LDS: eight 1 KiB buffer loads per K
//     step straight into registers (the packed operand is fragment-major already), double-buffered by step parity, issued one step ahead;
//     no ds_read of B, no barrier for B, half the LDS-DMA issue;
//   * the WINDOW slice (shared by the four positions) stays in LDS, double-buffered; ONE barrier per 16-channel slice (three K steps)
//     instead of one per step; a slice's pieces are issued two to three steps before the barrier that publishes them;
//   * LDS holds only the two window buffers (96 KiB) in the K loop; the epilogue's staging (two passes of 64 pairs x 128 channels x 4
//     positions = 128 KiB) overlays them.
// Per accumulator the arithmetic is the 8-wave kernel's -- K steps in the same order, lo.hi, hi.lo, hi.hi per step, the same output
// transform -- so the two kernels give the same bits (test_conv3x3_wino_x3_tiles_are_bit_identical covers shape 3 = this kernel).
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
template <int IMM>
__device__ __forceinline__ void buf_load16(u32x4v& dst, uint32_t voffset, const u32x4& srd, uint32_t soffset_uniform) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4" : "=&v"(dst) : "v"(voffset), "s"(srd), "s"(soffset_uniform), "n"(IMM) : "memory");
}
template <int NB>
__global__ __launch_bounds__(256, 1) void wino_x3w_kernel(const WinoParams p) {
  using namespace wnx;
  constexpr int G = 2, NV = 4, NW = 4, RT = 4, CT = 4, MT = 128, NT = 128;
  constexpr int ROWB = NB * 1024, WIN_CAP = 48 * 1024, USTAGE = ustage(2);
  constexpr int MAXP = WIN_CAP / 1024 / NW;            // window pieces per wave and slice (upper bound): 12, issued as two batches of 6
  constexpr int HB = MAXP / 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_win = smem;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nu = wave;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = logical / p.n_tiles, ntile = logical - mt * p.n_tiles;
  int q0, q_end;
  if (p.tpf > 0) {
    const int f = mt / p.tpf, t = mt - f * p.tpf;
    q0 = f * p.gpf + t * MT;
    q_end = min(q0 + MT, (f + 1) * p.gpf);
  } else {
    q0 = mt * MT;
    q_end = min(q0 + MT, p.total_pairs);
  }
  const int q_last = q_end - 1;
  const int H1 = p.H + 1;
  auto slot_of = [&](int q) { const int R = q / p.PW; return R + R / p.H + 1; };
  const int sig_b = __builtin_amdgcn_readfirstlane(slot_of(q0) - 1);
  const int NP = __builtin_amdgcn_readfirstlane((slot_of(q_last) - sig_b + 2) * NB);
  const int NSL = p.Cin / KS, KT = 3 * NSL;
  const int n0 = ntile * NT;
  const u32x4 srd_x = make_srd(p.x);
  const u32x4 srd_u = make_srd((const char*)p.u + (size_t)(n0 / UNT) * KT * USTAGE);
  const uint32_t lds_win = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)s_win;

  // ---- window pieces of this wave: piece pi = wave + 4 n = block pi % NB (= wave % NB for every n: NB divides 4) of window row pi / NB
  uint32_t prow[MAXP];
  uint32_t pokm = 0, prowokm = 0;
  static_for<MAXP>([&](auto nc) {
    constexpr int n = decltype(nc)::value;
    const int pi = wave + NW * n, j = pi / NB;
    const int sg = sig_b + j, f = sg / H1, r = sg - f * H1;
    if (pi < NP) pokm |= 1u << n;
    if (r != 0 && f < p.frames) prowokm |= 1u << n;
    prow[n] = __builtin_amdgcn_readfirstlane((uint32_t)(((long long)(f * p.H + r - 1) * p.W) * p.Cin * 4));
  });
  pokm = __builtin_amdgcn_readfirstlane(pokm);
  prowokm = __builtin_amdgcn_readfirstlane(prowokm);
  uint32_t w_voff;
  {
    const int l_ph = lane >> 5, l_i = (lane >> 2) & 7, l_cs = lane & 3, b = wave & (NB - 1);
    const int x = (8 * b + l_i) * G + l_ph - 1;
    const int c = l_cs ^ ((2 * b + (l_i >> 2)) & 3);
    w_voff = (unsigned)x < (unsigned)p.W ? (uint32_t)((x * p.Cin + 4 * c) * 4) : MCG_OOB_OFFSET;
  }
  auto issue_window = [&](auto hc, int cs, uint32_t dst) {       // batch hc (0, 1) of slice cs
    constexpr int h0 = decltype(hc)::value * HB;
    static_for<HB>([&](auto nc) {
      constexpr int n = h0 + decltype(nc)::value;
      if (pokm & (1u << n)) {
        const uint32_t v = (prowokm & (1u << n)) ? w_voff : MCG_OOB_OFFSET;
        lds_dma16<0>(v, srd_x, prow[n] + (uint32_t)cs * (KS * 4), dst + (uint32_t)(wave + NW * n) * 1024u);
      }
    });
  };
  // ---- weights of position nu: K step k = 8 KiB [channel tile][high, low][lane][16 B] at k USTAGE + nu 8192 of this channel block
  const uint32_t b_voff = (uint32_t)lane * 16u + (uint32_t)nu * 8192u;
  u32x4v bh[2][CT], bl[2][CT];
  auto issue_b1 = [&](auto slc, auto jc, uint32_t koff) {        // one of the eight loads of a step: j = 2 ct + (0 high, 1 low)
    constexpr int SLX = decltype(slc)::value, J = decltype(jc)::value;
    if constexpr (J & 1) buf_load16<(J & 3) * 1024>(bl[SLX][J >> 1], b_voff, srd_u, koff + (J >> 2) * 4096u);
    else buf_load16<(J & 3) * 1024>(bh[SLX][J >> 1], b_voff, srd_u, koff + (J >> 2) * 4096u);
  };

  // ---- input transform of this wave's position: V = d[toff0] + sgn d[toff1]
  const int toff0 = nu == 0 ? 0 : 1, toff1 = nu == 3 ? 3 : 2;
  const float tsg = nu == 1 ? 1.f : -1.f;
  const int pl = lane & 31, h = lane >> 5;
  const char* ap[RT][2][2];                               // window buffer 0, tap ky = 0
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const int q = min(q0 + rt * 32 + pl, q_last);
    const int R = q / p.PW, xg = q - R * p.PW;
    const int jrow = (R + R / p.H + 1) - sig_b - 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int wx = G * xg + (t == 0 ? toff0 : toff1);
      const int pp = wx / G, ph = wx - pp * G, swz = (pp >> 2) & 3;
      const int off = jrow * ROWB + (pp >> 3) * (G * 512) + ph * 512 + (pp & 7) * 64;
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) ap[rt][t][ch] = s_win + off + (((2 * h + ch) ^ swz) << 4);
    }
  }

  f32x16 acc[RT][CT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  struct Raw { uint4 d[RT][2][2]; };
  auto load_raw = [&](auto wbc, auto kyc, Raw& r) {
    constexpr int AOFF = decltype(wbc)::value * WIN_CAP + decltype(kyc)::value * ROWB;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < 2; ++t) { r.d[rt][t][0] = *(const uint4*)(ap[rt][t][0] + AOFF); r.d[rt][t][1] = *(const uint4*)(ap[rt][t][1] + AOFF); }
  };
  auto transform = [&](const Raw& r, bf16x8 (&ah)[RT], bf16x8 (&al)[RT]) {
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
      float v[8];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint32_t e0[4] = {r.d[rt][0][c].x, r.d[rt][0][c].y, r.d[rt][0][c].z, r.d[rt][0][c].w};
        const uint32_t e1[4] = {r.d[rt][1][c].x, r.d[rt][1][c].y, r.d[rt][1][c].z, r.d[rt][1][c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * c + e] = fmaf(tsg, __uint_as_float(e1[e]), __uint_as_float(e0[e]));
      }
      const uint4 v0 = make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
      const uint4 v1 = make_uint4(__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7]));
      split_f32x8(v0, v1, ah[rt], al[rt]);
    }
  };

  // ---- prologue: slice 0 (both batches), the weights of step 0, the first batch of slice 1
  issue_window(std::integral_constant<int, 0>{}, 0, lds_win);
  issue_window(std::integral_constant<int, 1>{}, 0, lds_win);
  static_for<8>([&](auto jc) { issue_b1(std::integral_constant<int, 0>{}, jc, 0u); });
  if (NSL > 1) issue_window(std::integral_constant<int, 0>{}, 1, lds_win + WIN_CAP);
  bf16x8 fh[2][RT], fl[2][RT];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  {
    Raw r0;
    load_raw(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, r0);
    transform(r0, fh[0], fl[0]);
  }
  constexpr int NM1 = RT * CT;
  uint32_t nh[RT][4], nl[RT][4];                          // the next step's A fragments while they are being made
#pragma nounroll
  for (int s2 = 0; s2 < NSL / 2; ++s2) {
    static_for<6>([&](auto uc) {
      constexpr int U = decltype(uc)::value, KY = U % 3, SL = U & 1, WB = U / 3;
      constexpr int U1 = (U + 1) % 6, KY1 = U1 % 3, WB1 = U1 / 3;
      const int k = 6 * s2 + U, cs = 2 * s2 + WB;
      // this step's weights (issued one step ago) have landed -- and with them every window piece issued before them
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(bh[SL][0]), "+v"(bh[SL][1]), "+v"(bh[SL][2]), "+v"(bh[SL][3]), "+v"(bl[SL][0]), "+v"(bl[SL][1]), "+v"(bl[SL][2]), "+v"(bl[SL][3])
                   :
                   : "memory");
      if constexpr (KY == 2) __builtin_amdgcn_s_barrier();   // slice cs + 1 is complete in its buffer; nobody reads slice cs's buffer any more
      Raw raw;
      load_raw(std::integral_constant<int, WB1>{}, std::integral_constant<int, KY1>{}, raw);
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t kn = (uint32_t)min(k + 1, KT - 1) * USTAGE;
      // The step's vector-memory instructions -- the eight weight loads of the next step, then (ky = 2: first batch of slice cs + 2 into the
      // buffer just released; ky = 0: second batch of slice cs + 1) six window pieces -- are issued ONE PER THREE OR FOUR MFMAs: the CU's address
      // unit takes ~16 cycles per 1 KiB request and the four waves run the same stream, so requests behind consecutive MFMAs (32 cycles apart)
      // queue up four deep and every wave stalls in its issue while its matrix pipe drains (measured: 70 cycles per weight load, 150 per
      // window piece).  Three MFMAs apart, the waves fall into a rotation after the first collision and the issue hides under the MFMAs.
      auto vmem_slot = [&](auto vc) {
        constexpr int V = decltype(vc)::value;
        if constexpr (V < 8) {
          issue_b1(std::integral_constant<int, SL ^ 1>{}, std::integral_constant<int, V>{}, kn);
        } else if constexpr (KY != 1 && V < 8 + HB) {
          constexpr int n = (KY == 0 ? HB : 0) + (V - 8);
          const int csn = KY == 0 ? cs + 1 : cs + 2;
          if (csn < NSL && (pokm & (1u << n))) {
            const uint32_t v = (prowokm & (1u << n)) ? w_voff : MCG_OOB_OFFSET;
            lds_dma16<0>(v, srd_x, prow[n] + (uint32_t)csn * (KS * 4), lds_win + (KY == 0 ? (WB ^ 1) : WB) * WIN_CAP + (uint32_t)(wave + NW * n) * 1024u);
          }
        }
      };
      // term 1 (activation low x weight high): 16 MFMAs, a memory instruction behind MFMAs 0, 3, 6, 9, 12, 15
      static_for<NM1>([&](auto mc) {
        constexpr int M = decltype(mc)::value, I = M / CT, J = M % CT;
        acc[I][J] = x3_mfma(fl[SL][I], __builtin_bit_cast(bf16x8, bh[SL][J]), acc[I][J]);
        if constexpr (M % 3 == 0) {
          __builtin_amdgcn_sched_barrier(0);
          vmem_slot(std::integral_constant<int, M / 3>{});
          __builtin_amdgcn_sched_barrier(0);
        }
      });
      __builtin_amdgcn_sched_barrier(0);
      // terms 2 and 3: eight units of four MFMAs + a sixteenth of the next step's transform + split (16 VALU), a memory instruction behind each
      static_for<8>([&](auto qc) {
        constexpr int Q = decltype(qc)::value, I = Q & 3, TERM3 = Q >> 2;
        constexpr int XR = Q >> 1, XC = Q & 1;               // transform unit: row tile XR, 16-byte chunk XC of the next step's A fragment
        {
          const uint32_t e0[4] = {raw.d[XR][0][XC].x, raw.d[XR][0][XC].y, raw.d[XR][0][XC].z, raw.d[XR][0][XC].w};
          const uint32_t e1[4] = {raw.d[XR][1][XC].x, raw.d[XR][1][XC].y, raw.d[XR][1][XC].z, raw.d[XR][1][XC].w};
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaf(tsg, __uint_as_float(e1[e]), __uint_as_float(e0[e]));
          split_pair(v[0], v[1], nh[XR][2 * XC], nl[XR][2 * XC]);
          split_pair(v[2], v[3], nh[XR][2 * XC + 1], nl[XR][2 * XC + 1]);
        }
#pragma unroll
        for (int j2 = 0; j2 < CT; ++j2)
          acc[I][j2] = x3_mfma(fh[SL][I], __builtin_bit_cast(bf16x8, TERM3 ? bh[SL][j2] : bl[SL][j2]), acc[I][j2]);
        // (the unit's results are only read one step later: without this pin instruction selection sinks all of a step's transform
        // behind its last MFMA, where nothing covers it)
        asm volatile("" : "+v"(nh[XR][2 * XC]), "+v"(nh[XR][2 * XC + 1]), "+v"(nl[XR][2 * XC]), "+v"(nl[XR][2 * XC + 1]));
        static_for<4>([&](auto) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        });
        __builtin_amdgcn_sched_barrier(0);
        vmem_slot(std::integral_constant<int, 6 + Q>{});
        __builtin_amdgcn_sched_barrier(0);
      });
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        fh[SL ^ 1][rt] = __builtin_bit_cast(bf16x8, make_uint4(nh[rt][0], nh[rt][1], nh[rt][2], nh[rt][3]));
        fl[SL ^ 1][rt] = __builtin_bit_cast(bf16x8, make_uint4(nl[rt][0], nl[rt][1], nl[rt][2], nl[rt][3]));
      }
    });
  }
  // (the last step's weight loads -- a re-fetch of its own stage, nobody reads them -- are still in flight: their registers stay tied to
  // this wait, or the compiler hands them to the epilogue's address arithmetic and the late data lands on a pointer.  Seen as a memory
  // fault under a second engine's load only, where a load takes long enough.)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
               : "+v"(bh[0][0]), "+v"(bh[0][1]), "+v"(bh[0][2]), "+v"(bh[0][3]), "+v"(bl[0][0]), "+v"(bl[0][1]), "+v"(bl[0][2]), "+v"(bl[0][3]),
                 "+v"(bh[1][0]), "+v"(bh[1][1]), "+v"(bh[1][2]), "+v"(bh[1][3]), "+v"(bl[1][0]), "+v"(bl[1][1]), "+v"(bl[1][2]), "+v"(bl[1][3])
               :
               : "memory");
  __syncthreads();

  // ---- epilogue: the 8-wave kernel's, two passes of two row tiles
  constexpr int PP = 64, CPR = NT / 4;
  float* const C = (float*)smem;                           // [nu][PP pairs][NT]
  const float wsc = p.wscale > 0.f ? p.wscale : 1.f;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          C[(nu * PP + rt * 32 + mfma32_row(r, lane)) * NT + ct * 32 + (lane & 31)] = acc[2 * pass + rt][ct][r];
    __syncthreads();
    for (int item = tid; item < PP * CPR; item += 256) {
      const int prl = item / CPR, ch4 = (item - prl * CPR) * 4;
      const int q = q0 + pass * PP + prl;
      if (q >= q_end) continue;
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bv = *(const float4*)(p.bias + n0 + ch4);
      float4 m[NV];
#pragma unroll
      for (int v = 0; v < NV; ++v) m[v] = *(const float4*)(C + (v * PP + prl) * NT + ch4);
      float4 y[2];
      y[0] = make_float4(((m[0].x + m[1].x) + m[2].x) * wsc + bv.x, ((m[0].y + m[1].y) + m[2].y) * wsc + bv.y, ((m[0].z + m[1].z) + m[2].z) * wsc + bv.z, ((m[0].w + m[1].w) + m[2].w) * wsc + bv.w);
      y[1] = make_float4(((m[1].x - m[2].x) - m[3].x) * wsc + bv.x, ((m[1].y - m[2].y) - m[3].y) * wsc + bv.y, ((m[1].z - m[2].z) - m[3].z) * wsc + bv.z, ((m[1].w - m[2].w) - m[3].w) * wsc + bv.w);
      const int R = q / p.PW, xo = G * (q - R * p.PW);
      float* yp = p.y + ((long long)R * p.W + xo) * p.Cout + n0 + ch4;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (xo + j >= p.W) break;
        float4 o = y[j];
        if (p.relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *(float4*)(yp + (long long)j * p.Cout) = o;
      }
    }
    __syncthreads();
  }
}

// Tile shapes.  G = 2: 0 = 128 pairs x 128 channels (8 waves), 1 = 64 x 64 (8 waves), 2 = 32 x 64 (4 waves), 3 = 128 x 128 with one wave per
// SIMD (wino_x3w_kernel).  G = 4: 0 = 64 groups x 128
// channels (12 waves), 1 or 2 = 32 x 64 (6 waves).
static inline int wino_x3_tile_groups(int g, int shape) { return g == 2 ? ((shape == 0 || shape == 3) ? 128 : (shape == 1 ? 64 : 32)) : (shape == 0 ? 64 : 32); }
static inline int wino_x3_tile_channels(int shape) { return (shape == 0 || shape == 3) ? 128 : 64; }
// 1 KiB pieces per window row, and the largest window (rows) a tile of mt groups can need: its rows, the zero rows between frames (cross:
// tiles run over frame boundaries), two halo rows
static inline int wino_x3_pieces(int W, int g) { return (g * ((W + g - 1) / g) + 2 + 8 * g - 1) / (8 * g) * (g / 2); }
static inline int wino_x3_max_rows(int H, int W, int mt, int g, bool cross) {
  const int PW = (W + g - 1) / g;
  int rows = (PW - 1 + mt - 1) / PW + 1;
  if (!cross && rows > H) rows = H;
  const int nz = cross ? (rows >= 2 ? (rows - 2) / H + 1 : 0) : 0;
  return rows + nz + 2;
}
static inline int wino_x3_nb_template(int pieces) { return pieces <= 1 ? 1 : (pieces <= 2 ? 2 : 4); }
// 3x3 / stride 1 / pad 1, f16x3: channel counts the tiles divide, a window that fits its buffer, operands inside the 2 GiB descriptor.
// Decided by the LAYER's shape only (never by the batch): a layer either runs this arithmetic or the direct kernel's for every batch size.
// -> 0 not applicable, 1 tiles may run over frame boundaries, 2 tiles must be cut at frame boundaries (the window buffer holds no zero row)
static inline int wino_x3_mode(int frames, int H, int W, int Cin, int Cout, int g) {
  if (Cin % 32 != 0 || Cout % wnx::UNT != 0 || H < 1 || W < 2 || frames < 1 || (g != 2 && g != 4)) return 0;
  if (g == 4 && (W % 4 != 0 || W < 16)) return 0;       // F(4,3) only where a row is whole groups and wide enough to pay
  const int pcs = wino_x3_pieces(W, g);
  if (pcs > 4) return 0;
  const int nb = wino_x3_nb_template(pcs);
  const long long px = (long long)frames * H * W;
  if (!(px * Cin * 4 < MCG_DMA_MAX_BYTES && px * Cout * 4 < MCG_DMA_MAX_BYTES && px < 0x7fffffffLL)) return 0;
  const int mt = wino_x3_tile_groups(g, 0), cap = wnx::wcap(g, 4);
  if (wino_x3_max_rows(H, W, mt, g, true) * nb * 1024 <= cap) return 1;
  if (wino_x3_max_rows(H, W, mt, g, false) * nb * 1024 <= cap) return 2;
  return 0;
}
static inline bool wino_x3_applicable(int frames, int H, int W, int Cin, int Cout, int g = 2) { return wino_x3_mode(frames, H, W, Cin, Cout, g) != 0; }
static inline size_t wino_x3_weight_bytes(int Cin, int Cout, int g = 2) { return (size_t)(Cout / wnx::UNT) * (3 * Cin / wnx::KS) * wnx::ustage(g); }

template <int NB, int RH, int RT, int CT, int G>
static inline int launch_wino_x3_t(hipStream_t s, const WinoParams& p, int grid) {
  constexpr int kLds = 2 * wnx::wcap(G, CT) + 2 * ((G + 2) * CT * 2 * 1024);
  static bool raised[MCG_MAX_DEVICES] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!raised[dev]) {
    if (hipFuncSetAttribute((const void*)wino_x3_kernel<NB, RH, RT, CT, G>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
    raised[dev] = true;
  }
  hipLaunchKernelGGL((wino_x3_kernel<NB, RH, RT, CT, G>), dim3(grid), dim3(64 * (G + 2) * RH), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int NB>
static inline int launch_wino_x3w(hipStream_t s, const WinoParams& p, int grid) {
  constexpr int kLds = 4 * 64 * 128 * 4;                 // epilogue staging (the K loop uses the first 96 KiB: two window buffers)
  static bool raised[MCG_MAX_DEVICES] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!raised[dev]) {
    if (hipFuncSetAttribute((const void*)wino_x3w_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
    raised[dev] = true;
  }
  hipLaunchKernelGGL((wino_x3w_kernel<NB>), dim3(grid), dim3(256), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
template <int NB>
static inline int launch_wino_x3_nb(hipStream_t s, const WinoParams& p, int g, int shape, int grid) {
  if (g == 2 && shape == 3) return launch_wino_x3w<NB>(s, p, grid);
  if (g == 4) return shape == 0 ? launch_wino_x3_t<NB, 2, 1, 4, 4>(s, p, grid) : launch_wino_x3_t<NB, 1, 1, 2, 4>(s, p, grid);
  if (shape == 0) return launch_wino_x3_t<NB, 2, 2, 4, 2>(s, p, grid);
  if (shape == 1) return launch_wino_x3_t<NB, 2, 1, 2, 2>(s, p, grid);
  return launch_wino_x3_t<NB, 1, 1, 2, 2>(s, p, grid);
}
// returns 0 on success; the caller has checked wino_x3_applicable(.., g).  shape: -1 = by grid size (the largest tile that still makes ~half
// a chip's worth of workgroups), else forced (tests)
static const int kWinoMinGrid = 130;
static inline int launch_wino_x3(hipStream_t s, WinoParams p, int shape = -1, int g = 2) {
  const int mode = wino_x3_mode(p.frames, p.H, p.W, p.Cin, p.Cout, g);
  if (mode == 0) return 1;
  p.PW = (p.W + g - 1) / g;
  p.gpf = p.H * p.PW;
  p.total_pairs = p.frames * p.gpf;
  auto mtiles = [&](int sh) {
    const int mt = wino_x3_tile_groups(g, sh);
    return mode == 2 ? p.frames * ((p.gpf + mt - 1) / mt) : (p.total_pairs + mt - 1) / mt;
  };
  auto grid_of = [&](int sh) { return mtiles(sh) * (p.Cout / wino_x3_tile_channels(sh)); };
  if (shape < 0) shape = grid_of(0) >= kWinoMinGrid ? (g == 2 ? 3 : 0) : ((g == 2 && grid_of(1) >= kWinoMinGrid) ? 1 : 2);
  if (g == 4 && (shape == 1 || shape == 3)) shape = 2;
  p.n_tiles = p.Cout / wino_x3_tile_channels(shape);
  p.tpf = mode == 2 ? (p.gpf + wino_x3_tile_groups(g, shape) - 1) / wino_x3_tile_groups(g, shape) : 0;
  const int grid = grid_of(shape);
  const int nb = wino_x3_nb_template(wino_x3_pieces(p.W, g));
  if (nb == 1) return launch_wino_x3_nb<1>(s, p, g, shape, grid);
  if (nb == 2) return launch_wino_x3_nb<2>(s, p, g, shape, grid);
  return launch_wino_x3_nb<4>(s, p, g, shape, grid);
}
