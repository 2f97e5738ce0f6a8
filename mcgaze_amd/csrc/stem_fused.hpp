// Fused stem for the bf16 engine: NCHW f32 frame -> conv 7x7/s2 (+ folded BN) -> ReLU -> max-pool 3x3/s2 -> NHWC bf16, ONE kernel
// (resnet.py:636-639 with deep_stem=False; the three-kernel path in igemm.hip -- pack, implicit GEMM, pool -- moves the
// 112x112x64 conv map through HBM twice: 719 MB written and read back per 448 frames, against 270 MB in + 180 MB out of
// algorithmic traffic).
//
// One workgroup (4 waves) owns an 8x8 tile of POOLED pixels of one frame:
//   1. the 39x39 input window behind it (17x17 conv pixels, 7x7/s2 taps) is read straight from the NCHW f32 frame
//      (coalesced along x), converted to bf16 and parked in LDS as [39][40][4 ch] (c = 3 is zero) -- 8 bytes per pixel, so
//      the 16-byte MFMA A fragment of conv pixel (cy, cx), tap row kh, tap columns 2j..2j+1 is ONE aligned ds_read_b128 at
//      ((2 cy + kh) * 40 + 2 cx + 2 j) * 8: the im2col is the address, as in the big kernel; image borders are zeros here;
//   2. the 64 x 224 weight matrix (K = 7 rows x 8 px x 4 ch, kw = 7 and c = 3 carry zeros: the same packing as the unfused
//      path, and the same K order, so results are bit-identical to it) lives in REGISTERS: every lane keeps its 16-byte B
//      chunk of all 14 K-steps of its wave's 32-channel half (56 VGPRs), loaded once from L2;
//   3. 289 conv pixels = 10 MFMA row blocks (padded to 320) x 2 channel halves = 20 units of 14 v_mfma_f32_32x32x16_bf16,
//      5 per wave; + bias in f32, rounded to bf16 (v_cvt_pk_bf16_f32), stored to LDS as [pixel][64];
//   4. each thread reduces two (pooled pixel, 8-channel chunk) windows from LDS with packed signed-16-bit max against 0 --
//      ReLU and max-pool in one (ReLU commutes with the rounding and with max); taps outside the conv map are skipped, which is
//      the pool's padding -- and writes 16 coalesced bytes.
// The grid is persistent (3 workgroups per CU, LDS 12.2 + 40 KiB each): a workgroup walks a strided list of tiles, keeps its
// weight fragments, and fetches the NEXT tile's pixels into registers before computing the current one, so the HBM latency of
// the input hides under the MFMA phase.  SQ counters (profiles/r01_g_stem.md) put the kernel at the VALU issue rate, not at
// HBM: what is left is instruction count.
#pragma once
#include "igemm_dma.hpp"

typedef short s16x8 __attribute__((ext_vector_type(8)));

namespace stemf {
constexpr int PT = 8;                 // pooled tile edge
constexpr int CT = 2 * PT + 1;        // conv tile edge (17)
constexpr int IT = 2 * (CT - 1) + 7;  // input tile edge (39)
constexpr int ITW = 40;               // padded row length (pixels) of the LDS input tile: keeps fragment reads 16-byte aligned
constexpr int NPIX = CT * CT;         // 289 conv pixels
constexpr int MB = (NPIX + 31) / 32;  // 10 MFMA row blocks
static_assert(MB % 2 == 0, "row blocks split evenly over two wave pairs");
constexpr int KS = 14;                // K-steps of 16: (kh, kw half)
constexpr int IN_BYTES = IT * ITW * 8;
constexpr int CONV_BYTES = MB * 32 * 128;
}  // namespace stemf

template <typename F>   // bf16_t or f16_t
__global__ __launch_bounds__(256) void stem_fused_kernel(const float* __restrict__ img, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                         bf16_t* __restrict__ y, int H, int W, int Hc, int Wc, int Ho, int Wo, int tiles_x, int tiles,
                                                         int total) {
  using namespace stemf;
  __shared__ __attribute__((aligned(16))) char s_in[IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_conv[CONV_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)H * W;

  // ---- 2. weight fragments into registers, once per workgroup (the grid is persistent: each workgroup walks a strided list of
  // tiles).  Waves 0-1 own channels 0..31, waves 2-3 channels 32..63; each wave runs 5 of the 10 row blocks against its half.
  const int nb = wave >> 1, mb0 = (wave & 1) * (MB / 2);
  uint4 bfrag[KS];
  {
    const bf16_t* wl = w + (size_t)(nb * 32 + (lane & 31)) * (KS * 16) + (lane >> 5) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfrag[ks] = *(const uint4*)(wl + ks * 16);
  }
  const int row = lane & 31, half = lane >> 5;
  const float b = bias[nb * 32 + row];

  // ---- 1. input window of tile t: global -> registers (all loads in flight at once) ... -> LDS (bf16, 4 channels per pixel)
  constexpr int TRIPS = (IT * ITW + 255) / 256;
  float v[TRIPS][3];
  auto origin = [&](int t, int& n, int& py0, int& px0) {
    n = t / tiles;
    const int r = t - n * tiles, ty = r / tiles_x;
    py0 = ty * PT;
    px0 = (r - ty * tiles_x) * PT;
  };
  auto fetch = [&](int t) {
    int n, py0, px0;
    origin(t, n, py0, px0);
    const int iy0 = 4 * py0 - 5, ix0 = 4 * px0 - 5;  // conv origin 2 p0 - 1 (pool pad 1), input origin 2 c0 - 3 (conv pad 3)
    const float* f0 = img + (size_t)n * 3 * plane;
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      const int r = idx / ITW, c = idx - r * ITW;
      const int gy = iy0 + r, gx = ix0 + c;
      v[j][0] = v[j][1] = v[j][2] = 0.f;
      if (idx < IT * ITW && c < IT && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        const float* p = f0 + (size_t)gy * W + gx;
        v[j][0] = p[0]; v[j][1] = p[plane]; v[j][2] = p[2 * plane];
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      if (idx < IT * ITW) *(uint2*)(s_in + idx * 8) = make_uint2(H16<F>::pack2(v[j][0], v[j][1]), H16<F>::pack2(v[j][2], 0.f));
    }
  };

  int t = blockIdx.x;
  if (t < total) fetch(t);
  for (; t < total; t += gridDim.x) {
    park();
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < total) fetch(tn);  // the next tile's pixels travel while this one is computed
    int n, py0, px0;
    origin(t, n, py0, px0);
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;

    // ---- 3. conv as MFMA: 20 (row block, channel half) units, 5 per wave
#pragma unroll 1
    for (int mb = mb0; mb < mb0 + MB / 2; ++mb) {
      const int m = min(mb * 32 + row, NPIX - 1);  // padded rows recompute the last pixel; never read back
      const int cy = m / CT, cx = m - cy * CT;
      const char* a0 = s_in + ((2 * cy) * ITW + 2 * cx + 2 * half) * 8;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const uint4 a = *(const uint4*)(a0 + ((ks >> 1) * ITW + (ks & 1) * 4) * 8);
        Mma<F>::run(acc, a, bfrag[ks]);
      }
      // C rows of this lane: base + {0,1,2,3, 8..11, 16..19, 24..27}.  Rows 289..319 of the last block are padding: s_conv has
      // room for them, so all 16 values are stored unpredicated, row offsets as ds_write immediates.  Only the bias is applied
      // here: ReLU commutes with the bf16 rounding and with max, so the pool's "max with 0" below performs it.
      const int mbase = mb * 32 + 4 * half;
      char* crow = s_conv + mbase * 128 + (nb * 32 + row) * 2;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const uint32_t pk = H16<F>::pack2(acc[r] + b, acc[r + 1] + b);
        *(bf16_t*)(crow + ((r & 3) + 8 * (r >> 2)) * 128) = (bf16_t)(pk & 0xffffu);
        *(bf16_t*)(crow + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * 128) = (bf16_t)(pk >> 16);
      }
    }
    __syncthreads();  // conv tile complete; s_in free for the next window

    // ---- 4. 3x3/s2 max over the conv tile, 16-byte channel chunks
    for (int idx = tid; idx < PT * PT * 8; idx += 256) {
      const int cc = idx & 7, pp = idx >> 3;
      const int py = pp / PT, px = pp - py * PT;
      if (py0 + py >= Ho || px0 + px >= Wo) continue;
      // max(0, window) = ReLU then max-pool.  bf16 bit patterns compare like sign-magnitude integers: a signed 16-bit max against
      // 0 maps every negative value (sign bit set) to +0 and orders the non-negative ones numerically.  Conv pixels outside the
      // conv map (top / left border tiles, ragged bottom / right) are the pool's padding: their taps are skipped.
      s16x8 mx = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if ((unsigned)(cy0 + 2 * py + dy) >= (unsigned)Hc) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if ((unsigned)(cx0 + 2 * px + dx) >= (unsigned)Wc) continue;
          mx = __builtin_elementwise_max(mx, *(const s16x8*)(s_conv + ((2 * py + dy) * CT + 2 * px + dx) * 128 + cc * 16));
        }
      }
      *(uint4*)(y + (((size_t)n * Ho + py0 + py) * Wo + px0 + px) * 64 + cc * 8) = __builtin_bit_cast(uint4, mx);
    }
    // the next iteration's park() touches only s_in; its barrier orders these s_conv reads before the next tile's s_conv writes
  }
}

static inline int launch_stem_fused(hipStream_t s, const float* img, const void* w_stem, const float* bias, void* y, int N, int H, int W, bool fp16 = false) {
  const int Hc = H / 2, Wc = W / 2, Ho = (Hc + 2 - 3) / 2 + 1, Wo = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Ho + stemf::PT - 1) / stemf::PT, tiles_x = (Wo + stemf::PT - 1) / stemf::PT;
  const long long total = (long long)tiles_y * tiles_x * N;
  if (total > 0x7fffffffLL) return 1;
  const int grid = (int)(total < 256 * 3 ? total : 256 * 3);  // persistent: 3 workgroups per CU (LDS-limited)
  if (fp16) hipLaunchKernelGGL(stem_fused_kernel<f16_t>, dim3(grid), dim3(256), 0, s, img, (const bf16_t*)w_stem, bias, (bf16_t*)y, H, W, Hc, Wc, Ho, Wo, tiles_x,
                               tiles_y * tiles_x, (int)total);
  else hipLaunchKernelGGL(stem_fused_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, img, (const bf16_t*)w_stem, bias, (bf16_t*)y, H, W, Hc, Wc, Ho, Wo, tiles_x,
                          tiles_y * tiles_x, (int)total);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// The same for the f16x3 engine (f32 activations, split-packed weights, three MFMAs per product): NCHW f32 frame -> conv 7x7/s2
// + BN -> ReLU -> max-pool -> NHWC f32 in one kernel.  The three-kernel path costs that engine 1.38 ms per 448 frames (pack 0.11,
// contraction 0.87, pool 0.39: the f32 conv map is 1.44 GB written and read back).  Differences from the bf16 kernel above:
//   * the input window is split ONCE when it is parked in LDS -- fp16 high and low parts (split_pair, the contraction kernel's
//     arithmetic) into two planes of [rows][40][4 ch] -- instead of once per fragment read;
//   * a lane keeps the high AND low 16-byte chunks of its weight row for all 14 K-steps (112 VGPRs, from the split-packed matrix
//     of packing.py::split_pack); each K-step issues lo.hi, hi.lo, hi.hi in the contraction kernel's order: bit-identical to it;
//   * the conv tile is f32 ([pixel][64] x 4 bytes), so a workgroup owns 8 x 4 pooled pixels (17 x 9 conv pixels = 5 row blocks,
//     40 KiB) to keep two workgroups per CU; the pool is a float4 max against 0.
namespace stemx {
constexpr int PTX = 8, PTY = 4;                   // pooled tile
constexpr int CTX = 2 * PTX + 1, CTY = 2 * PTY + 1;   // conv tile 17 x 9
constexpr int ITX = 2 * (CTX - 1) + 7, ITY = 2 * (CTY - 1) + 7;   // input tile 39 x 23
constexpr int ITW = 40;
constexpr int NPIX = CTX * CTY;                   // 153 conv pixels
constexpr int MB = (NPIX + 31) / 32;              // 5 row blocks
constexpr int KS = 14;
constexpr int IN_BYTES = ITY * ITW * 8;           // one plane (hi or lo)
constexpr int CONV_BYTES = MB * 32 * 256;
}  // namespace stemx

__global__ __launch_bounds__(256, 2) void stem_fused_x3_kernel(const float* __restrict__ img, const char* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, int H, int W, int Hc, int Wc, int Ho, int Wo, int tiles_x,
                                                               int tiles, int total) {
  using namespace stemx;
  __shared__ __attribute__((aligned(16))) char s_hi[IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_lo[IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_conv[CONV_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)H * W;
  // waves 0-1 own channels 0..31, waves 2-3 channels 32..63; of the 5 row blocks the even wave of a pair runs 0..2, the odd one 3..4
  const int nb = wave >> 1, mb0 = (wave & 1) ? 3 : 0, mb1 = (wave & 1) ? MB : 3;
  const int row = lane & 31, half = lane >> 5;
  bf16x8 bh[KS], bl[KS];
  {
    const char* wl = w + (size_t)(nb * 32 + row) * (KS * 16 * 4) + half * 32;   // split-packed row: per 8 K elements 16 B of hi, 16 B of lo
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + ks * 64));
      bl[ks] = __builtin_bit_cast(bf16x8, *(const uint4*)(wl + ks * 64 + 16));
    }
  }
  const float b = bias[nb * 32 + row];

  constexpr int TRIPS = (ITY * ITW + 255) / 256;
  float v[TRIPS][3];
  auto origin = [&](int t, int& n, int& py0, int& px0) {
    n = t / tiles;
    const int r = t - n * tiles, ty = r / tiles_x;
    py0 = ty * PTY;
    px0 = (r - ty * tiles_x) * PTX;
  };
  auto fetch = [&](int t) {
    int n, py0, px0;
    origin(t, n, py0, px0);
    const int iy0 = 4 * py0 - 5, ix0 = 4 * px0 - 5;
    const float* f0 = img + (size_t)n * 3 * plane;
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      const int r = idx / ITW, c = idx - r * ITW;
      const int gy = iy0 + r, gx = ix0 + c;
      v[j][0] = v[j][1] = v[j][2] = 0.f;
      if (idx < ITY * ITW && c < ITX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) {
        const float* p = f0 + (size_t)gy * W + gx;
        v[j][0] = p[0]; v[j][1] = p[plane]; v[j][2] = p[2 * plane];
      }
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      if (idx < ITY * ITW) {
        uint32_t h01, l01, h2, l2;
        split_pair(v[j][0], v[j][1], h01, l01);
        split_pair(v[j][2], 0.f, h2, l2);
        *(uint2*)(s_hi + idx * 8) = make_uint2(h01, h2);
        *(uint2*)(s_lo + idx * 8) = make_uint2(l01, l2);
      }
    }
  };

  int t = blockIdx.x;
  if (t < total) fetch(t);
  for (; t < total; t += gridDim.x) {
    park();
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < total) fetch(tn);
    int n, py0, px0;
    origin(t, n, py0, px0);
    const int cy0 = 2 * py0 - 1, cx0 = 2 * px0 - 1;
#pragma unroll 1
    for (int mb = mb0; mb < mb1; ++mb) {
      const int m = min(mb * 32 + row, NPIX - 1);
      const int cy = m / CTX, cx = m - cy * CTX;
      const int a0 = ((2 * cy) * ITW + 2 * cx + 2 * half) * 8;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = a0 + ((ks >> 1) * ITW + (ks & 1) * 4) * 8;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(s_hi + off));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(s_lo + off));
        acc = x3_mfma(al, bh[ks], acc);
        acc = x3_mfma(ah, bl[ks], acc);
        acc = x3_mfma(ah, bh[ks], acc);
      }
      const int mbase = mb * 32 + 4 * half;
      float* crow = (float*)(s_conv + mbase * 256) + nb * 32 + row;
#pragma unroll
      for (int r = 0; r < 16; ++r) crow[((r & 3) + 8 * (r >> 2)) * 64] = acc[r] + b;
    }
    __syncthreads();
    for (int idx = tid; idx < PTX * PTY * 16; idx += 256) {
      const int cc = idx & 15, pp = idx >> 4;
      const int py = pp / PTX, px = pp - py * PTX;
      if (py0 + py >= Ho || px0 + px >= Wo) continue;
      float4 mx = make_float4(0.f, 0.f, 0.f, 0.f);   // max with 0 = ReLU, then the pool's max
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        if ((unsigned)(cy0 + 2 * py + dy) >= (unsigned)Hc) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          if ((unsigned)(cx0 + 2 * px + dx) >= (unsigned)Wc) continue;
          const float4 c = *(const float4*)(s_conv + ((2 * py + dy) * CTX + 2 * px + dx) * 256 + cc * 16);
          mx.x = fmaxf(mx.x, c.x); mx.y = fmaxf(mx.y, c.y); mx.z = fmaxf(mx.z, c.z); mx.w = fmaxf(mx.w, c.w);
        }
      }
      *(float4*)(y + (((size_t)n * Ho + py0 + py) * Wo + px0 + px) * 64 + cc * 4) = mx;
    }
  }
}

static inline int launch_stem_fused_x3(hipStream_t s, const float* img, const void* w_stem, const float* bias, void* y, int N, int H, int W) {
  const int Hc = H / 2, Wc = W / 2, Ho = (Hc + 2 - 3) / 2 + 1, Wo = (Wc + 2 - 3) / 2 + 1;
  const int tiles_y = (Ho + stemx::PTY - 1) / stemx::PTY, tiles_x = (Wo + stemx::PTX - 1) / stemx::PTX;
  const long long total = (long long)tiles_y * tiles_x * N;
  if (total > 0x7fffffffLL) return 1;
  const int grid = (int)(total < 256 * 2 ? total : 256 * 2);  // persistent: 2 workgroups per CU (14.4 + 40 KiB of LDS each)
  hipLaunchKernelGGL(stem_fused_x3_kernel, dim3(grid), dim3(256), 0, s, img, (const char*)w_stem, bias, (float*)y, H, W, Hc, Wc, Ho, Wo, tiles_x,
                     tiles_y * tiles_x, (int)total);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
