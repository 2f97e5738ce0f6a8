// Fused bottleneck tail of the MCG_F16X3 engine (resnet.py:263-302, two consecutive Bottleneck.forward calls), layer1 (CM = 64 mid
// channels) and layer2 (CM = 128) of a ResNet-50; C = 4 CM:
//
//     t = relu(conv2_3x3(o1) + b2)                         [M][CM]    never leaves the CU
//     y = relu([t | x0] . W3^T + b3 (+ res))               [M][C]     written: the block's output (next residual, C2 / C3)
//     z = relu(y . W1n^T + b1n)                            [M][CN]    written: the NEXT block's conv1 output
//
// Layer-granular execution moves, per identity block at f32 activations, 5.76 GB at 448 frames (conv1 reads y, conv2 reads and
// writes the 64-channel maps, conv3 re-reads y as the residual and writes y'); this kernel reads o1 and the residual once and
// writes y' and the next o1 once: 3.6 GB.  The three contractions are CHAINED IN REGISTERS:
//
//   * every contraction runs TRANSPOSED (MFMA A operand = weight rows = output channels, B operand = pixels), so a lane's
//     accumulators hold, for ITS pixel (lane & 31), the channels (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of a 32-channel tile;
//   * those registers -- after bias / residual / ReLU in f32 and the fp16 high / low split (split_pair: the bits the contraction
//     kernel's in-register A split produces from the stored f32 activation) -- ARE the B operand of the next contraction: a K-step of
//     16 channels takes, from lane half h, the eight channels 4 h + {0..3, 8..11}.  That is a fixed permutation of the 16 K indices of
//     the step; the host packs W3 and W1n with the same permutation (packing.py::bneck_stream), so the products are the reference's
//     and only the order of the 16 terms inside one MFMA differs from the layer-granular kernel (parity: tests/test_gpu_kernels.py);
//   * weights travel as 16 KiB SLABS (2 channel tiles x 4 K-steps x {high, low} x 64 lanes x 16 B, MFMA-fragment-major) in the order
//     the tile consumes them: 9 taps of W2, then per 64-channel chunk of y the W3 slab(s) and the W1n slab(s).  ONE loader wave
//     streams them HBM/L2 -> LDS (`buffer_load ... lds`) through a ring of five slots, filling whatever is free; the slabs that share a B
//     operand form a GROUP (conv2's output pairs of one tap, conv3's K parts, the next conv1's output pairs: 1 or 2 slabs) and the seven
//     compute waves meet the loader at one s_barrier per group (the loader's vmcnt covers its DMA, the barrier publishes it; nothing
//     else orders LDS-DMA -- and a compute wave drains its own LDS reads before it arrives: bnx_barrier_drained);
//   * a workgroup owns an 8 x 28 pixel tile = 7 groups of 32 pixels = 7 compute waves.  The 10 x 30 x 64 input window of conv2 sits
//     in LDS split ONCE into fp16 high / low chunk planes (the nine taps are immediates, conv3x3_c64.hpp's layout); the next tile's
//     window is fetched and parked by the compute waves during the 1x1 phases, when nobody reads the window.  CM = 128: the planes hold 64
//     channels at a time -- conv2's K is walked as two halves, the second half's window (prefetched into registers at the start of the
//     tile) is parked between them behind one extra barrier;
//   * residual rows and y / z rows move between HBM and registers in the accumulator layout (16-byte pieces, the two lane halves
//     adjacent: 32 contiguous bytes per pixel row per instruction, a whole 128-byte line per four).
//
// LDS: 16 planes x 4832 B window + 5 x 16 KiB ring + biases = 157.3 KiB: one workgroup (8 waves, <= 256 VGPRs) per CU, persistent.
#pragma once
#include "igemm_dma.hpp"

// Measurement builds only (tools/lab/bneck_ablate.sh compiles engine.hip with -DBNX_ABLATE=<mask> into a side library; the product build
// has 0 and none of this exists in it): which memory stream of the kernel costs what, in time AND joules.  A stream is switched off behind
// a condition that is false at run time but unknown to the compiler (p.H < 0), so the code and its registers stay what they are:
//   1 = the weight ring is filled once and never again (the slabs of the first five stay in LDS: real values, wrong weights)
//   2 = no residual loads     4 = no y / z stores     8 = no window loads after the first tile (its planes stay)
#ifndef BNX_ABLATE
#define BNX_ABLATE 0
#endif
namespace bnx {
constexpr int TH = 8, TW = 28, WH = TH + 2, WW = TW + 2;
constexpr int NPIX = TH * TW, NG = NPIX / 32, WPIX = WH * WW;       // 224 px, 7 groups, 300 window px
constexpr int PLANE = WPIX * 16 + 32;                               // one chunk plane: [window px][8 halves], padded
constexpr int NPLANE = 16;                                          // (K-step 0..3) x (lane half) x (high, low)
constexpr int WIN_BYTES = NPLANE * PLANE;
constexpr int SLAB = 16384, NSLOT = 5, RING_BYTES = NSLOT * SLAB;
constexpr int NCOMP = NG, NT = 64 * (NCOMP + 1);                    // 7 compute waves + 1 loader wave
constexpr int WIN_ROUNDS = (WPIX * 16 + NCOMP * 64 - 1) / (NCOMP * 64);   // 16-byte f32 chunks of a window per compute lane: 11
static_assert(NPIX % 32 == 0, "tile must be whole 32-pixel groups");
}  // namespace bnx

struct BneckParams {
  const float* x;        // conv2 input [frames][H][W][CM] f32 (the block's conv1 output)
  const float* res;      // NSRC == 1: residual [M][C];  NSRC == 2: the downsample conv's input [M][64] (block input, stride 1)
  const char* wstream;   // weight slabs in consumption order (packing.py::bneck_stream)
  const float* bias;     // [CM conv2 | C conv3 (+ downsample) | CN next conv1 | descale of W2, W3, W1n, 0]: the three matrices are packed pre-scaled by
                         // powers of two (fp16 low halves normal), their accumulators are multiplied back before the bias (exact)
  float* y;              // [M][C]
  float* z;              // [M][CN] (CN > 0)
  int H, W, tiles_x, tiles_per_frame, total_tiles;
  int res_blocked, y_blocked;  // the residual / y tensor is in the kernel's own BLOCKED layout (below) instead of [M][C]: only between two launches of
                               // this kernel over the same H x W grid (engine.hip: the blocks inside layer1 / layer2); NSRC == 1 only for the residual
  unsigned long long* trace;   // optional (measurement aid): workgroup 0 writes s_memtime stamps -- [0..2047] compute wave 0 (tile start, after
                               // conv2, after each chunk), [2048..4095] the loader wave (each slab's arrival); NULL in the product path
};

// eight f32 of one lane -> the fp16 high / low B-operand fragments of one K-step
__device__ __forceinline__ void bnx_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// One K-step of a slab: both channel tiles' weight fragments (high, low) against one pixel fragment (high, low); the six MFMAs
// alternate between the two accumulators (small terms first) so that no MFMA waits for the one issued just before it.
__device__ __forceinline__ void bnx_step(const char* sl, const bf16x8& xh, const bf16x8& xl, f32x16& a0, f32x16& a1) {
  const bf16x8 w0h = __builtin_bit_cast(bf16x8, *(const uint4*)(sl));
  const bf16x8 w0l = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + 1024));
  const bf16x8 w1h = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + 8192));
  const bf16x8 w1l = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + 8192 + 1024));
  a0 = x3_mfma(w0l, xh, a0);
  a1 = x3_mfma(w1l, xh, a1);
  a0 = x3_mfma(w0h, xl, a0);
  a1 = x3_mfma(w1h, xl, a1);
  a0 = x3_mfma(w0h, xh, a0);
  a1 = x3_mfma(w1h, xh, a1);
}

// LDS reads in flight across a barrier.  The compiler sinks the last K-step's MFMAs of a slab below the next slab's barrier and leaves
// their ds_reads in flight across it.  Two things are written behind a barrier, and neither may meet such a read:
//   * a ring slot.  The first version refilled, behind barrier kt, the slot of slab kt - 1: its reads usually return long before the DMA
//     data does (an L2 round trip later), but not always -- with a second engine busy on the device one launch in ~200 came back with a
//     pixel group computed from a half-overwritten weight fragment (tools/lab/bneck_contention_probe.py, profiles/r03_x_lds_war.md).
//     Now every compute-wave barrier drains the wave's LDS reads first, so a slot is free the moment the barrier that ends its group
//     has been passed.  (Also tried and equally safe: refilling only the slot of the slab before the previous one, bare barriers.
//     Same box: 0.945 / 0.950 / 0.940 ms for the layer1 identity block without a fix / slack slot / drain + slack slot.)
//   * the window planes at the K-half switch of CM = 128, re-parked right behind a barrier (drained like every other; the next tile's
//     window is parked two barriers after the last read of this tile's).
__device__ __forceinline__ void bnx_barrier_drained() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <int CM, int NSRC, int CN, int LAYOUT>   // LAYOUT: bit 0 = the residual is blocked, bit 1 = y is blocked (BneckParams; compile-time: the
                                                  // address forms as run-time values cost the CN = 128 instantiations 8 - 22 spilled registers)
__global__ __launch_bounds__(bnx::NT, 1) void bneck_x3_kernel(const BneckParams p) {
  using namespace bnx;
  constexpr bool RES_BLK = (LAYOUT & 1) != 0, Y_BLK = (LAYOUT & 2) != 0;
  static_assert(!RES_BLK || NSRC == 1, "only a residual can arrive blocked");
  constexpr int C = 4 * CM, KH = CM / 64, NCH = C / 64;          // output channels, 64-channel K halves of conv2, 64-channel chunks of y
  constexpr int PARTS = CM / 64 + NSRC - 1;                      // 64-wide K parts of conv3 (t [| second source])
  constexpr int NS1 = 9 * KH * KH;                               // conv2 slabs: K half x tap x 64-channel output pair
  constexpr int NS = NS1 + NCH * (PARTS + CN / 64);              // slabs per tile
  constexpr int CT1 = CM / 32, CT3 = CN / 32;                    // channel tiles of t and z
  constexpr int WPC = (WIN_ROUNDS + NCH - 1) / NCH;              // window rounds fetched per chunk iteration
  // Residual prefetch depth in 64-channel chunks (32 VGPRs each); the chunk loop is unrolled by RD only, so the ring slots are
  // compile-time (fully unrolled, the compiler hoists loads across chunks and spills).  Measured (profiles/r03_g_bneck_phases.md): depth 2
  // moves the waiting from the 1x1 phases into the 3x3 phase and leaves the tile time where it was (0.933 vs 0.932 ms, + 22 registers):
  // whenever the kernel touches memory it runs at what one CU can pull.  Depth 1.
  constexpr int RD = 1;
  static_assert(NCH % RD == 0, "chunk loop is unrolled by the prefetch depth");
  static_assert(CM == 64 || CM == 128, "64 or 128 mid channels");
  static_assert(NSRC == 1 || CM == 64, "the second source is layer1's first block only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_win = smem;
  char* const s_ring = smem + WIN_BYTES;
  float* const s_bias = (float*)(smem + WIN_BYTES + RING_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < CM + C + CN + 4; i += NT) s_bias[i] = p.bias[i];
  __syncthreads();
  const int first = blockIdx.x, stride = gridDim.x;
  if (first >= p.total_tiles) return;

  if (wave == NCOMP) {
    // ------------------------------------------------------------------ loader wave: the weight stream, three slabs ahead
    const u32x4 srd_w = make_srd(p.wstream);
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)s_ring;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t kt_issue = 0, slot_issue = 0;          // slab-in-tile and ring slot of the next slab to issue
    int issued_total = 0;
    auto issue = [&]() {
      const uint32_t src = kt_issue * SLAB, dst = ring_base + slot_issue * SLAB;
      if (!(BNX_ABLATE & 1) || issued_total < NSLOT || p.H < 0)
        static_for<16>([&](auto pc) { lds_dma16<decltype(pc)::value * 1024>(voff, srd_w, src + decltype(pc)::value * 1024, dst); });
      if (BNX_ABLATE & 1) ++issued_total;
      kt_issue = kt_issue + 1 == NS ? 0 : kt_issue + 1;
      slot_issue = slot_issue + 1 == NSLOT ? 0 : slot_issue + 1;
    };
    // Slabs are consumed in GROUPS (the slabs that share a B operand: conv2's output pairs of one tap, conv3's K parts, the next
    // conv1's output pairs) with one barrier per group.  occ = slabs issued and not yet released, prev = size of the group being read.
    // At a group's barrier the previous group's slots are released (its readers drained their LDS reads before arriving) and refilled.
    int occ = 0, prev = 0;
    long long to_issue = (long long)((p.total_tiles - first + stride - 1) / stride) * NS;
    auto fill = [&]() {
      while (occ < NSLOT && to_issue > 0) { issue(); ++occ; --to_issue; }
    };
    fill();
    int tr_n = 0;
    auto boundary = [&](auto gc) {
      constexpr int G = decltype(gc)::value;
      const int beyond = occ - prev - G;             // slabs issued after this group's: their pieces may still be in flight
      if (beyond <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (beyond == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (beyond == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
      if (p.trace && blockIdx.x == 0 && lane == 0 && tr_n < 2040) { p.trace[2048 + tr_n] = __builtin_amdgcn_s_memtime(); ++tr_n; }
      __builtin_amdgcn_s_barrier();                  // the group is published; every compute wave has drained its reads of the previous group
      if (p.trace && blockIdx.x == 0 && lane == 0 && tr_n < 2040) { p.trace[2048 + tr_n] = __builtin_amdgcn_s_memtime(); ++tr_n; }
      occ -= prev;
      prev = G;
      fill();
    };
    for (int tile = first; tile < p.total_tiles; tile += stride) {
#pragma unroll 1
      for (int kh = 0; kh < KH; ++kh) {
        if (kh == 1) __builtin_amdgcn_s_barrier();   // the compute waves' window switch (second K half)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) boundary(std::integral_constant<int, CM / 64>{});
      }
#pragma unroll 1
      for (int oc = 0; oc < NCH; ++oc) {
        boundary(std::integral_constant<int, PARTS>{});
        if constexpr (CN > 0) boundary(std::integral_constant<int, CN / 64>{});
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  // -------------------------------------------------------------------- compute waves: one 32-pixel group each
  const int g = wave, pl = lane & 31, half = lane >> 5;
  const int m = g * 32 + pl, ty = m / TW, tx = m - ty * TW;
  const char* const bw = s_win + half * 2 * PLANE + (ty * WW + tx) * 16;   // this lane's B-operand base (tap (0,0), K-step 0, high)
  const float* const s_b2 = s_bias;
  const float* const s_b3 = s_bias + CM;
  const float* const s_b1 = s_bias + CM + C;
  const float ws2 = s_bias[CM + C + CN], ws3 = s_bias[CM + C + CN + 1], ws1 = s_bias[CM + C + CN + 2];   // descale factors (exact powers of two)
  const int ctid = tid;                                                       // 0 .. 447 among the compute lanes

  auto origin = [&](int t, int& n, int& ty0, int& tx0) {
    n = t / p.tiles_per_frame;
    const int r = t - n * p.tiles_per_frame, tyi = r / p.tiles_x;
    ty0 = tyi * TH;
    tx0 = (r - tyi * p.tiles_x) * TW;
  };
  // window chunk `it` of this lane: 4 consecutive channels (c16) of window pixel wp, K half kh
  auto win_load = [&](int t, int kh, int it) -> float4 {
    int n, ty0, tx0;
    origin(t, n, ty0, tx0);
    const int idx = it * (NCOMP * 64) + ctid;
    const int wp = idx >> 4, c16 = idx & 15;
    const int wy = wp / WW, wx = wp - wy * WW;
    const int gy = ty0 - 1 + wy, gx = tx0 - 1 + wx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < WPIX * 16 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
      v = *(const float4*)(p.x + (((long long)n * p.H + gy) * p.W + gx) * CM + kh * 64 + c16 * 4);
    return v;
  };
  auto win_park = [&](int it, const float4& v) {
    const int idx = it * (NCOMP * 64) + ctid;
    if (idx >= WPIX * 16) return;
    const int wp = idx >> 4, c16 = idx & 15;
    uint32_t h0, l0, h1, l1;
    split_pair(v.x, v.y, h0, l0);
    split_pair(v.z, v.w, h1, l1);
    // plane = ((K-step * 2 + half) * 2 + high/low); 8 bytes at position (c16 & 1) of the pixel's 16-byte slot
    char* dst = s_win + (((c16 >> 2) * 2 + ((c16 >> 1) & 1)) * 2) * PLANE + wp * 16 + (c16 & 1) * 8;
    *(uint2*)dst = make_uint2(h0, h1);
    *(uint2*)(dst + PLANE) = make_uint2(l0, l1);
  };

  // first window
  for (int it = 0; it < WIN_ROUNDS; ++it) win_park(it, win_load(first, 0, it));

  const uint32_t a_lane = (uint32_t)lane * 16u;
  uint32_t slot = 0;                                                          // ring slot of the next slab to consume
  auto slab_at = [&](int i) { const uint32_t t = slot + i; return s_ring + (t >= NSLOT ? t - NSLOT : t) * SLAB + a_lane; };   // i-th slab of the group
  auto slab_adv = [&](int n) { slot = slot + n >= NSLOT ? slot + n - NSLOT : slot + n; };

  int tr_c = 0;
  auto stamp = [&]() {
    if (p.trace && blockIdx.x == 0 && tid == 0 && tr_c < 2040) { p.trace[tr_c] = __builtin_amdgcn_s_memtime(); ++tr_c; }
  };
  for (int tile = first; tile < p.total_tiles; tile += stride) {
    stamp();
    int n, ty0, tx0;
    origin(tile, n, ty0, tx0);
    const int gy = ty0 + ty, gx = tx0 + tx;
    const bool valid = gy < p.H && gx < p.W;
    const bool st_ok = valid && ((BNX_ABLATE & 4) ? p.H < 0 : true), ld_ok = valid && ((BNX_ABLATE & 2) ? p.H < 0 : true);
    const long long row = ((long long)n * p.H + (valid ? gy : 0)) * p.W + (valid ? gx : 0);
    // Residual / y addressing.  [M][C]: a lane's 16-byte piece of (chunk oc, channel tile c, quad q) is at row * C + 64 oc + 32 c + 8 q + 4 half -- one
    // instruction touches 32 rows, 32 bytes of each (the two lane halves adjacent).  BLOCKED: per (tile, pixel group) a block of 32 x C floats ordered
    // [oc][c][q][pixel][half][4], so the 64 lanes of one instruction write ONE contiguous KiB -- eight whole lines instead of 32 quarter lines.
    // Measured on the layer1 identity block: a CU moves 11.5 B/clk with the quarter-line pattern, 18.5 with whole lines.
    // (wave-uniform block base + a per-lane constant: no vector registers per tile)
    const long long sblk = ((long long)tile * NG + g) * (32 * C);
    const uint32_t lane_off = (uint32_t)(pl * 8 + 4 * half);
    const int next = tile + stride;
    const bool has_next = next < p.total_tiles && ((BNX_ABLATE & 8) ? p.H < 0 : true);

    // residual rows of chunk 0 (NSRC == 1) / the second source's fragments (NSRC == 2): in flight under the 3x3 phase
    float4 rres[RD][2][4];
    bf16x8 xfh[4], xfl[4];
    auto res_load = [&](auto slotc, int chunk) {
      constexpr int SL = decltype(slotc)::value;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rres[SL][c][q] = !ld_ok ? make_float4(0.f, 0.f, 0.f, 0.f)
                                  : (RES_BLK ? *(const float4*)(p.res + sblk + lane_off + chunk * 2048 + c * 1024 + q * 256)
                                                   : *(const float4*)(p.res + row * C + chunk * 64 + c * 32 + 8 * q + 4 * half));
    };
    if constexpr (NSRC == 1) {
      static_for<RD>([&](auto d) { res_load(d, decltype(d)::value); });
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 a = valid ? *(const float4*)(p.res + row * 64 + 16 * s + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = valid ? *(const float4*)(p.res + row * 64 + 16 * s + 4 * half + 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bnx_split8(v, xfh[s], xfl[s]);
      }
    }

    // ---------------- conv2: per K half, per tap, per 64-channel output pair one slab (4 K-steps of 16 channels)
    float4 w2nd[KH == 2 ? WIN_ROUNDS : 1];                  // CM = 128: the second K half's window, in flight under the first half
    const bool win2 = (BNX_ABLATE & 8) ? (p.H < 0 || tile == first) : true;
    if constexpr (KH == 2) {
      if (win2) {
#pragma unroll
        for (int it = 0; it < WIN_ROUNDS; ++it) w2nd[it] = win_load(tile, 1, it);
      }
    }
    f32x16 acc1[CT1];
#pragma unroll
    for (int c = 0; c < CT1; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[c][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's window writes (parked during the previous tile) are done
    static_for<KH>([&](auto khc) {
      constexpr int KHI = decltype(khc)::value;
      if constexpr (KHI == 1) {                             // every wave is done with the first half's planes: park the second half
        bnx_barrier_drained();
        if (win2) {
#pragma unroll
          for (int it = 0; it < WIN_ROUNDS; ++it) win_park(it, w2nd[it]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      static_for<9>([&](auto tc) {
        constexpr int TAP = decltype(tc)::value;
        constexpr int TOFF = ((TAP / 3) * WW + TAP % 3) * 16;
        // one group per tap: its CM / 64 output-pair slabs share the window fragments
        bnx_barrier_drained();
        const char* sl[CM / 64];
#pragma unroll
        for (int op = 0; op < CM / 64; ++op) sl[op] = slab_at(op);
        static_for<4>([&](auto jc) {
          constexpr int J = decltype(jc)::value;
          const bf16x8 xh = __builtin_bit_cast(bf16x8, *(const uint4*)(bw + (4 * J) * PLANE + TOFF));
          const bf16x8 xl = __builtin_bit_cast(bf16x8, *(const uint4*)(bw + (4 * J + 1) * PLANE + TOFF));
          static_for<CM / 64>([&](auto opc) {
            constexpr int OP = decltype(opc)::value;
            bnx_step(sl[OP] + J * 2048, xh, xl, acc1[2 * OP], acc1[2 * OP + 1]);
          });
        });
        slab_adv(CM / 64);
      });
    });
    stamp();
    // t = relu(acc1 + b2) -> B fragments of conv3's K-steps (K-step s = channels 16 s .. 16 s + 15)
    bf16x8 tfh[CM / 16], tfl[CM / 16];
#pragma unroll
    for (int s = 0; s < CM / 16; ++s) {
      const int c = s >> 1, q0 = 2 * (s & 1);
      float v[8];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const float4 b = *(const float4*)(s_b2 + c * 32 + 8 * (q0 + qq) + 4 * half);
        v[4 * qq] = fmaxf(acc1[c][4 * (q0 + qq)] * ws2 + b.x, 0.f);
        v[4 * qq + 1] = fmaxf(acc1[c][4 * (q0 + qq) + 1] * ws2 + b.y, 0.f);
        v[4 * qq + 2] = fmaxf(acc1[c][4 * (q0 + qq) + 2] * ws2 + b.z, 0.f);
        v[4 * qq + 3] = fmaxf(acc1[c][4 * (q0 + qq) + 3] * ws2 + b.w, 0.f);
      }
      bnx_split8(v, tfh[s], tfl[s]);
    }

    // ---------------- per 64-channel chunk of y: conv3 slab(s) -> y chunk (stored) -> next conv1 slab(s)
    f32x16 acc3[CT3 > 0 ? CT3 : 1];
#pragma unroll
    for (int c = 0; c < (CT3 > 0 ? CT3 : 1); ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[c][r] = 0.f;
#pragma unroll 1
    for (int oc0 = 0; oc0 < NCH; oc0 += RD)
    static_for<RD>([&](auto occ) {
      constexpr int SLOT = decltype(occ)::value;
      const int oc = oc0 + SLOT;
      // the next tile's window (first K half), WPC rounds per chunk: loads now, parked after this chunk's contractions
      float4 wv[WPC];
      if (has_next) {
#pragma unroll
        for (int j = 0; j < WPC; ++j)
          if (oc * WPC + j < WIN_ROUNDS) wv[j] = win_load(next, 0, oc * WPC + j);
      }
      f32x16 acc2[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0.f;
      bnx_barrier_drained();                               // one group: the K parts of this chunk's conv3
      static_for<PARTS>([&](auto pc) {
        constexpr int PART = decltype(pc)::value;
        const char* sl = slab_at(PART);
        static_for<4>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
          const bf16x8 bh = PART < CM / 64 ? tfh[(PART < CM / 64 ? PART : 0) * 4 + S] : xfh[S];
          const bf16x8 bl = PART < CM / 64 ? tfl[(PART < CM / 64 ? PART : 0) * 4 + S] : xfl[S];
          bnx_step(sl + S * 2048, bh, bl, acc2[0], acc2[1]);
        });
      });
      slab_adv(PARTS);
      // y chunk = relu(acc2 + b3 (+ res)), finished IN PLACE in the accumulators and stored
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = oc * 64 + c * 32 + 8 * q + 4 * half;
          const float4 b = *(const float4*)(s_b3 + ch);
          float4 o = make_float4(acc2[c][4 * q] * ws3 + b.x, acc2[c][4 * q + 1] * ws3 + b.y, acc2[c][4 * q + 2] * ws3 + b.z, acc2[c][4 * q + 3] * ws3 + b.w);
          if constexpr (NSRC == 1) { const float4 r = rres[SLOT][c][q]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
          if (st_ok) {
            if constexpr (Y_BLK) *(float4*)(p.y + sblk + lane_off + oc * 2048 + c * 1024 + q * 256) = o;
            else *(float4*)(p.y + row * C + ch) = o;
          }
          acc2[c][4 * q] = o.x; acc2[c][4 * q + 1] = o.y; acc2[c][4 * q + 2] = o.z; acc2[c][4 * q + 3] = o.w;
        }
      if constexpr (NSRC == 1) {
        if (oc + RD < NCH) res_load(std::integral_constant<int, SLOT>{}, oc + RD);   // refill the slot just consumed
      }
      // next conv1: this chunk's 64 channels of y are four K-steps; a step's B fragment is split from the accumulators when it is
      // needed (per slab: the split is 24 VALU per step, the registers of four fragments would not fit beside 128 output channels)
      if constexpr (CN > 0) {
        bnx_barrier_drained();                             // one group: the CN / 64 output-pair slabs share the split of y
        const char* sl[CN / 64];
#pragma unroll
        for (int pr = 0; pr < CN / 64; ++pr) sl[pr] = slab_at(pr);
        static_for<4>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
          constexpr int YC = S >> 1, R0 = 8 * (S & 1);
          const float v[8] = {acc2[YC][R0], acc2[YC][R0 + 1], acc2[YC][R0 + 2], acc2[YC][R0 + 3], acc2[YC][R0 + 4], acc2[YC][R0 + 5], acc2[YC][R0 + 6], acc2[YC][R0 + 7]};
          bf16x8 yh, yl;
          bnx_split8(v, yh, yl);
          static_for<CN / 64>([&](auto prc) {
            constexpr int PR = decltype(prc)::value;
            bnx_step(sl[PR] + S * 2048, yh, yl, acc3[2 * PR], acc3[2 * PR + 1]);
          });
        });
        slab_adv(CN / 64);
      }
      if (has_next) {
#pragma unroll
        for (int j = 0; j < WPC; ++j)
          if (oc * WPC + j < WIN_ROUNDS) win_park(oc * WPC + j, wv[j]);
      }
      stamp();
    });
    // ---------------- z = relu(acc3 + b1n)
    if constexpr (CN > 0) {
#pragma unroll
      for (int c = 0; c < CT3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = c * 32 + 8 * q + 4 * half;
          const float4 b = *(const float4*)(s_b1 + ch);
          const float4 o = make_float4(fmaxf(acc3[c][4 * q] * ws1 + b.x, 0.f), fmaxf(acc3[c][4 * q + 1] * ws1 + b.y, 0.f), fmaxf(acc3[c][4 * q + 2] * ws1 + b.z, 0.f),
                                       fmaxf(acc3[c][4 * q + 3] * ws1 + b.w, 0.f));
          if (st_ok) *(float4*)(p.z + row * CN + ch) = o;
        }
    }
  }
}

// Applicable: layer1 / layer2 of a ResNet-50 (64 / 128 mid channels, 4 x that out), stride 1; a second source (the first block's
// downsample input) only for 64 mid channels, 64 channels wide, at stride 1 on the same grid.
static inline bool bneck_x3_applicable(int cm, int c, int cn, int nsrc, int k2, int stride2) {
  if (cm == 64) return c == 256 && (cn == 0 || cn == 64 || cn == 128) && (nsrc == 1 || (nsrc == 2 && k2 == 64 && stride2 == 1));
  return cm == 128 && c == 512 && (cn == 0 || cn == 128) && nsrc == 1;
}
static inline size_t bneck_x3_stream_bytes(int cm, int nsrc, int cn) {
  return (size_t)(9 * (cm / 64) * (cm / 64) + (cm / 16) * (cm / 64 + nsrc - 1 + cn / 64)) * bnx::SLAB;
}

template <int CM, int NSRC, int CN, int LAYOUT>
static inline int launch_bneck_x3_t(hipStream_t s, const BneckParams& p) {
  constexpr int kLds = bnx::WIN_BYTES + bnx::RING_BYTES + (CM + 4 * CM + CN + 4) * 4;
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    if (hipFuncSetAttribute((const void*)bneck_x3_kernel<CM, NSRC, CN, LAYOUT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int grid = p.total_tiles < cus_of[dev] ? p.total_tiles : cus_of[dev];
  hipLaunchKernelGGL((bneck_x3_kernel<CM, NSRC, CN, LAYOUT>), dim3(grid), dim3(bnx::NT), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
// Which tensors may be blocked: y only where the next block's conv1 is this kernel's z on the same grid (cn = cm: a block inside a layer), the
// residual only for nsrc = 1.  Anything else is refused (1) rather than run in the wrong layout.
static inline bool bneck_x3_layout_ok(int cm, int nsrc, int cn, int res_blocked, int y_blocked) {
  return (!res_blocked || nsrc == 1) && (!y_blocked || (cn == cm));
}
// frames x H x W pixels; returns 0 on success
static inline int launch_bneck_x3(hipStream_t s, BneckParams p, int frames, int cm, int nsrc, int cn) {
  const int tiles_y = (p.H + bnx::TH - 1) / bnx::TH;
  p.tiles_x = (p.W + bnx::TW - 1) / bnx::TW;
  p.tiles_per_frame = tiles_y * p.tiles_x;
  const long long total = (long long)p.tiles_per_frame * frames;
  if (total <= 0 || total > 0x7fffffffLL) return 1;
  p.total_tiles = (int)total;
  if (!bneck_x3_layout_ok(cm, nsrc, cn, p.res_blocked, p.y_blocked)) return 1;
  const int lay = (p.res_blocked ? 1 : 0) | (p.y_blocked ? 2 : 0);
  if (cm == 128) {
    if (cn == 0) return lay ? launch_bneck_x3_t<128, 1, 0, 1>(s, p) : launch_bneck_x3_t<128, 1, 0, 0>(s, p);
    switch (lay) {
      case 0: return launch_bneck_x3_t<128, 1, 128, 0>(s, p);
      case 1: return launch_bneck_x3_t<128, 1, 128, 1>(s, p);
      case 2: return launch_bneck_x3_t<128, 1, 128, 2>(s, p);
      default: return launch_bneck_x3_t<128, 1, 128, 3>(s, p);
    }
  }
  if (nsrc == 1) {
    if (cn == 0) return lay ? launch_bneck_x3_t<64, 1, 0, 1>(s, p) : launch_bneck_x3_t<64, 1, 0, 0>(s, p);
    if (cn == 128) return lay ? launch_bneck_x3_t<64, 1, 128, 1>(s, p) : launch_bneck_x3_t<64, 1, 128, 0>(s, p);
    switch (lay) {
      case 0: return launch_bneck_x3_t<64, 1, 64, 0>(s, p);
      case 1: return launch_bneck_x3_t<64, 1, 64, 1>(s, p);
      case 2: return launch_bneck_x3_t<64, 1, 64, 2>(s, p);
      default: return launch_bneck_x3_t<64, 1, 64, 3>(s, p);
    }
  }
  if (cn == 0) return launch_bneck_x3_t<64, 2, 0, 0>(s, p);
  if (cn == 64) return lay ? launch_bneck_x3_t<64, 2, 64, 2>(s, p) : launch_bneck_x3_t<64, 2, 64, 0>(s, p);
  return launch_bneck_x3_t<64, 2, 128, 0>(s, p);
}
