// Fused bottleneck tail of the MCG_F16X3 engine (resnet.py:263-302, two consecutive Bottleneck.forward calls), layer1 of a ResNet-50:
//
//     t = relu(conv2_3x3(o1) + b2)                         [M][64]    never leaves the CU
//     y = relu([t | x0] . W3^T + b3 (+ res))               [M][256]   written: the block's output (next residual, C2)
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
//     streams them HBM/L2 -> LDS (`buffer_load ... lds`) through a ring of five slots, four slabs ahead; the seven compute waves meet
//     it at one s_barrier per slab (the loader's vmcnt covers its DMA, the barrier publishes it; nothing else orders LDS-DMA);
//   * a workgroup owns an 8 x 28 pixel tile = 7 groups of 32 pixels = 7 compute waves.  The 10 x 30 x 64 input window of conv2 sits
//     in LDS split ONCE into fp16 high / low chunk planes (the nine taps are immediates, conv3x3_c64.hpp's layout); the next tile's
//     window is fetched and parked by the compute waves during the 1x1 phases, when nobody reads the window;
//   * residual rows and y / z rows move between HBM and registers in the accumulator layout (16-byte pieces, the two lane halves
//     adjacent: 32 contiguous bytes per pixel row per instruction, a whole 128-byte line per four).
//
// LDS: 16 planes x 4832 B window + 5 x 16 KiB ring + biases = 157.3 KiB: one workgroup (8 waves, <= 256 VGPRs) per CU, persistent.
#pragma once
#include "igemm_dma.hpp"

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
  const float* x;        // conv2 input [frames][H][W][64] f32 (the block's conv1 output)
  const float* res;      // NSRC == 1: residual [M][256];  NSRC == 2: the downsample conv's input [M][64] (block input, stride 1)
  const char* wstream;   // weight slabs in consumption order (packing.py::bneck_stream)
  const float* bias;     // [64 conv2 | 256 conv3 (+ downsample) | CN next conv1]
  float* y;              // [M][256]
  float* z;              // [M][CN] (CN > 0)
  int H, W, tiles_x, tiles_per_frame, total_tiles;
};

// eight f32 of one lane -> the fp16 high / low B-operand fragments of one K-step
__device__ __forceinline__ void bnx_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(v[2 * q], v[2 * q + 1], h[q], l[q]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

template <int NSRC, int CN>
__global__ __launch_bounds__(bnx::NT, 1) void bneck_x3_kernel(const BneckParams p) {
  using namespace bnx;
  constexpr int NS = 9 + 4 * (NSRC + CN / 64);      // slabs per tile
  constexpr int CT3 = CN / 32;                      // channel tiles of z
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_win = smem;
  char* const s_ring = smem + WIN_BYTES;
  float* const s_bias = (float*)(smem + WIN_BYTES + RING_BYTES);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 + 256 + CN; i += NT) s_bias[i] = p.bias[i];
  __syncthreads();
  const int first = blockIdx.x, stride = gridDim.x;
  if (first >= p.total_tiles) return;

  if (wave == NCOMP) {
    // ------------------------------------------------------------------ loader wave: the weight stream, four slabs ahead
    const u32x4 srd_w = make_srd(p.wstream);
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)s_ring;
    const uint32_t voff = (uint32_t)lane * 16u;
    uint32_t kt_issue = 0, slot_issue = 0;          // slab-in-tile and ring slot of the next slab to issue
    auto issue = [&]() {
      const uint32_t src = kt_issue * SLAB, dst = ring_base + slot_issue * SLAB;
      static_for<16>([&](auto pc) { lds_dma16<decltype(pc)::value * 1024>(voff, srd_w, src + decltype(pc)::value * 1024, dst); });
      kt_issue = kt_issue + 1 == NS ? 0 : kt_issue + 1;
      slot_issue = slot_issue + 1 == NSLOT ? 0 : slot_issue + 1;
    };
    static_for<NSLOT - 1>([&](auto) { issue(); });
    for (int tile = first; tile < p.total_tiles; tile += stride) {
#pragma unroll 1
      for (int kt = 0; kt < NS; ++kt) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(16 * (NSLOT - 2)) : "memory");   // this slab's 16 pieces have landed
        __builtin_amdgcn_s_barrier();                                                // published; the previous slab's slot is free
        issue();   // always (past the last tile it wraps to slabs nobody reads): the outstanding-piece count stays uniform
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
  const float* const s_b3 = s_bias + 64;
  const float* const s_b1 = s_bias + 64 + 256;
  const int ctid = tid;                                                       // 0 .. 447 among the compute lanes

  auto origin = [&](int t, int& n, int& ty0, int& tx0) {
    n = t / p.tiles_per_frame;
    const int r = t - n * p.tiles_per_frame, tyi = r / p.tiles_x;
    ty0 = tyi * TH;
    tx0 = (r - tyi * p.tiles_x) * TW;
  };
  // window chunk `it` of this lane: 4 consecutive channels (c16) of window pixel wp
  auto win_load = [&](int t, int it) -> float4 {
    int n, ty0, tx0;
    origin(t, n, ty0, tx0);
    const int idx = it * (NCOMP * 64) + ctid;
    const int wp = idx >> 4, c16 = idx & 15;
    const int wy = wp / WW, wx = wp - wy * WW;
    const int gy = ty0 - 1 + wy, gx = tx0 - 1 + wx;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < WPIX * 16 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
      v = *(const float4*)(p.x + (((long long)n * p.H + gy) * p.W + gx) * 64 + c16 * 4);
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
  for (int it = 0; it < WIN_ROUNDS; ++it) win_park(it, win_load(first, it));

  const uint32_t a_lane = (uint32_t)lane * 16u;
  uint32_t slot = 0;                                                          // ring slot of the next slab to consume
  auto slab_ptr = [&]() { return s_ring + slot * SLAB + a_lane; };
  auto slab_next = [&]() { slot = slot + 1 == NSLOT ? 0 : slot + 1; };

  for (int tile = first; tile < p.total_tiles; tile += stride) {
    int n, ty0, tx0;
    origin(tile, n, ty0, tx0);
    const int gy = ty0 + ty, gx = tx0 + tx;
    const bool valid = gy < p.H && gx < p.W;
    const long long row = ((long long)n * p.H + (valid ? gy : 0)) * p.W + (valid ? gx : 0);
    const int next = tile + stride;
    const bool has_next = next < p.total_tiles;

    // residual rows of chunk 0 (NSRC == 1) / the second source's fragments (NSRC == 2): in flight under the 3x3 phase
    float4 rres[2][4];
    bf16x8 xfh[4], xfl[4];
    if constexpr (NSRC == 1) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rres[c][q] = valid ? *(const float4*)(p.res + row * 256 + c * 32 + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 a = valid ? *(const float4*)(p.res + row * 64 + 16 * s + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 b = valid ? *(const float4*)(p.res + row * 64 + 16 * s + 4 * half + 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bnx_split8(v, xfh[s], xfl[s]);
      }
    }

    // ---------------- conv2: 9 tap slabs, K = 64 channels per tap (4 K-steps), both channel tiles of t
    f32x16 acc1[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[c][r] = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's window writes (parked during the previous tile) are done
    static_for<9>([&](auto tc) {
      constexpr int TAP = decltype(tc)::value;
      constexpr int TOFF = ((TAP / 3) * WW + TAP % 3) * 16;
      __builtin_amdgcn_s_barrier();
      const char* sl = slab_ptr();
      static_for<4>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        const bf16x8 xh = __builtin_bit_cast(bf16x8, *(const uint4*)(bw + (4 * J) * PLANE + TOFF));
        const bf16x8 xl = __builtin_bit_cast(bf16x8, *(const uint4*)(bw + (4 * J + 1) * PLANE + TOFF));
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bf16x8 wh = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + J * 2048));
          const bf16x8 wl = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + J * 2048 + 1024));
          acc1[c] = x3_mfma(wl, xh, acc1[c]);
          acc1[c] = x3_mfma(wh, xl, acc1[c]);
          acc1[c] = x3_mfma(wh, xh, acc1[c]);
        }
      });
      slab_next();
    });
    // t = relu(acc1 + b2) -> B fragments of conv3's four K-steps (K-step s = channels 16 s .. 16 s + 15)
    bf16x8 tfh[4], tfl[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int c = s >> 1, q0 = 2 * (s & 1);
      float v[8];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const float4 b = *(const float4*)(s_b2 + c * 32 + 8 * (q0 + qq) + 4 * half);
        v[4 * qq] = fmaxf(acc1[c][4 * (q0 + qq)] + b.x, 0.f);
        v[4 * qq + 1] = fmaxf(acc1[c][4 * (q0 + qq) + 1] + b.y, 0.f);
        v[4 * qq + 2] = fmaxf(acc1[c][4 * (q0 + qq) + 2] + b.z, 0.f);
        v[4 * qq + 3] = fmaxf(acc1[c][4 * (q0 + qq) + 3] + b.w, 0.f);
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
    for (int oc = 0; oc < 4; ++oc) {
      // the next tile's window, three rounds per chunk: loads now, parked after this chunk's contractions
      float4 wv[3];
      if (has_next) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (oc * 3 + j < WIN_ROUNDS) wv[j] = win_load(next, oc * 3 + j);
      }
      f32x16 acc2[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[c][r] = 0.f;
      static_for<NSRC>([&](auto pc) {
        constexpr int PART = decltype(pc)::value;
        __builtin_amdgcn_s_barrier();
        const char* sl = slab_ptr();
        static_for<4>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
          const bf16x8 bh = PART == 0 ? tfh[S] : xfh[S];
          const bf16x8 bl = PART == 0 ? tfl[S] : xfl[S];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + S * 2048));
            const bf16x8 wl = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + S * 2048 + 1024));
            acc2[c] = x3_mfma(wl, bh, acc2[c]);
            acc2[c] = x3_mfma(wh, bl, acc2[c]);
            acc2[c] = x3_mfma(wh, bh, acc2[c]);
          }
        });
        slab_next();
      });
      // y chunk = relu(acc2 + b3 (+ res)); stored; split into the next conv1's B fragments
      bf16x8 yfh[4], yfl[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int c = s >> 1, q0 = 2 * (s & 1);
        float v[8];
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
          const int q = q0 + qq, ch = oc * 64 + c * 32 + 8 * q + 4 * half;
          const float4 b = *(const float4*)(s_b3 + ch);
          float4 o = make_float4(acc2[c][4 * q] + b.x, acc2[c][4 * q + 1] + b.y, acc2[c][4 * q + 2] + b.z, acc2[c][4 * q + 3] + b.w);
          if constexpr (NSRC == 1) { o.x += rres[c][q].x; o.y += rres[c][q].y; o.z += rres[c][q].z; o.w += rres[c][q].w; }
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
          if (valid) *(float4*)(p.y + row * 256 + ch) = o;
          v[4 * qq] = o.x; v[4 * qq + 1] = o.y; v[4 * qq + 2] = o.z; v[4 * qq + 3] = o.w;
        }
        if constexpr (CN > 0) bnx_split8(v, yfh[s], yfl[s]);
      }
      if constexpr (NSRC == 1) {   // the next chunk's residual rows
        if (oc < 3) {
#pragma unroll
          for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              rres[c][q] = valid ? *(const float4*)(p.res + row * 256 + (oc + 1) * 64 + c * 32 + 8 * q + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      static_for<CN / 64>([&](auto prc) {
        constexpr int PR = decltype(prc)::value;
        __builtin_amdgcn_s_barrier();
        const char* sl = slab_ptr();
        static_for<4>([&](auto sc) {
          constexpr int S = decltype(sc)::value;
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const bf16x8 wh = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + S * 2048));
            const bf16x8 wl = __builtin_bit_cast(bf16x8, *(const uint4*)(sl + c * 8192 + S * 2048 + 1024));
            acc3[2 * PR + c] = x3_mfma(wl, yfh[S], acc3[2 * PR + c]);
            acc3[2 * PR + c] = x3_mfma(wh, yfl[S], acc3[2 * PR + c]);
            acc3[2 * PR + c] = x3_mfma(wh, yfh[S], acc3[2 * PR + c]);
          }
        });
        slab_next();
      });
      if (has_next) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
          if (oc * 3 + j < WIN_ROUNDS) win_park(oc * 3 + j, wv[j]);
      }
    }
    // ---------------- z = relu(acc3 + b1n)
    if constexpr (CN > 0) {
#pragma unroll
      for (int c = 0; c < CT3; ++c)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ch = c * 32 + 8 * q + 4 * half;
          const float4 b = *(const float4*)(s_b1 + ch);
          const float4 o = make_float4(fmaxf(acc3[c][4 * q] + b.x, 0.f), fmaxf(acc3[c][4 * q + 1] + b.y, 0.f), fmaxf(acc3[c][4 * q + 2] + b.z, 0.f),
                                       fmaxf(acc3[c][4 * q + 3] + b.w, 0.f));
          if (valid) *(float4*)(p.z + row * CN + ch) = o;
        }
    }
  }
}

// Applicable: layer1 of a ResNet-50 (64 mid channels, 256 out), stride 1, the second source (first block) at stride 1 on the same grid.
static inline bool bneck_x3_applicable(int cm, int c, int cn, int nsrc, int k2, int stride2) {
  return cm == 64 && c == 256 && (cn == 0 || cn == 64 || cn == 128) && (nsrc == 1 || (nsrc == 2 && k2 == 64 && stride2 == 1));
}
static inline size_t bneck_x3_stream_bytes(int nsrc, int cn) { return (size_t)(9 + 4 * (nsrc + cn / 64)) * bnx::SLAB; }

template <int NSRC, int CN>
static inline int launch_bneck_x3_t(hipStream_t s, const BneckParams& p) {
  constexpr int kLds = bnx::WIN_BYTES + bnx::RING_BYTES + (64 + 256 + CN) * 4;
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    if (hipFuncSetAttribute((const void*)bneck_x3_kernel<NSRC, CN>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds) != hipSuccess) return 1;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int grid = p.total_tiles < cus_of[dev] ? p.total_tiles : cus_of[dev];
  hipLaunchKernelGGL((bneck_x3_kernel<NSRC, CN>), dim3(grid), dim3(bnx::NT), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
// frames x H x W pixels; returns 0 on success
static inline int launch_bneck_x3(hipStream_t s, BneckParams p, int frames, int nsrc, int cn) {
  const int tiles_y = (p.H + bnx::TH - 1) / bnx::TH;
  p.tiles_x = (p.W + bnx::TW - 1) / bnx::TW;
  p.tiles_per_frame = tiles_y * p.tiles_x;
  const long long total = (long long)p.tiles_per_frame * frames;
  if (total <= 0 || total > 0x7fffffffLL) return 1;
  p.total_tiles = (int)total;
  if (nsrc == 1) {
    if (cn == 0) return launch_bneck_x3_t<1, 0>(s, p);
    if (cn == 64) return launch_bneck_x3_t<1, 64>(s, p);
    return launch_bneck_x3_t<1, 128>(s, p);
  }
  if (cn == 0) return launch_bneck_x3_t<2, 0>(s, p);
  if (cn == 64) return launch_bneck_x3_t<2, 64>(s, p);
  return launch_bneck_x3_t<2, 128>(s, p);
}
