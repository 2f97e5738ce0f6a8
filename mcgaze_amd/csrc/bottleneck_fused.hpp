// One identity bottleneck of layer1 (resnet.py:246-300; conv1 1x1 256->64, conv2 3x3 64->64, conv3 1x1 64->256, + input, ReLU) as ONE
// kernel for the bf16 engine: the 64-channel intermediates never leave the CU, the block's HBM traffic drops from
// 2.88 GB (three launches) to the 719 MB input + 719 MB output per 448 frames.
//
// A workgroup (16 waves) owns an 8 x 28 tile of output pixels of one frame (224 = 7 x 32) and its 10 x 30 halo window.  The three
// GEMMs run TRANSPOSED -- A operand = weight rows (output channels), B operand = pixels -- so that a lane of the 32x32 MFMA
// result owns 4 consecutive channels of ONE pixel: bias / ReLU / residual are applied in registers, pairs convert with
// v_cvt_pk_bf16_f32, and the 8 bytes go straight into the LDS operand tile of the next GEMM.  No f32 staging anywhere.
// Every LDS tile is stored as 16-byte chunk PLANES [chunk][pixel or row][16 B] (+32 bytes of padding per plane): fragment reads
// are 512 contiguous bytes per 32 lanes for any 3x3 tap, which is a plain address offset ((kh * 30 + kw) * 16).
//
//   phase 1  out1[300 window px][64] = ReLU(X[px][256] . W1^T + b1), zero outside the image (conv2's padding)   K = 256 in 4 slices
//   phase 2  out2[224 px][64]        = ReLU(sum over 9 taps out1[px + tap][64] . W2[tap]^T + b2)                 K = 576, one tap per slice
//   phase 3  Y[224 px][256]          = ReLU(out2 . W3^T + b3 + X[px])                                            4 passes of 64 channels
//
// K orders, rounding points and the order of the final additions are those of the three generic launches, so the result is
// BIT-IDENTICAL to them (tests/test_gpu_forward.py::test_fused_bottleneck_is_bit_identical).
//
// STATUS: experimental, off by default (MCG_FUSED_BLOCK=1 in engine.hip).  Correct, but at 726 us per 448-frame launch it is
// twice as slow as the three launches it replaces (DESIGN.md section 8.2 has the counters): one 16-wave workgroup per CU walking
// 35 barriers per tile has nothing to hide a stall behind.  Global memory is read two slices ahead into registers; the next step
// is two tiles in flight per CU.
#pragma once
#include "common.hpp"
#include <type_traits>

namespace bnf {
constexpr int TH = 8, TW = 28, WH = TH + 2, WW = TW + 2;
constexpr int NPIX = TH * TW, WPIX = WH * WW;       // 224, 300
constexpr int WPAD = 320;                             // window pixels padded to whole 32-column MFMA tiles
constexpr int P1 = WPAD * 16 + 32;                    // plane strides (bytes)
constexpr int P2 = NPIX * 16 + 32;
constexpr int PW = 64 * 16 + 32;                      // weight planes: 64 rows
constexpr int O1_BYTES = 8 * P1, O2_BYTES = 8 * P2, X_BYTES = 8 * P1, W_BYTES = 8 * PW;
constexpr int NW = 16, NT = 64 * NW;                 // waves / threads per workgroup
constexpr int XTRIPS = (WPAD * 8 + NT - 1) / NT;     // 16-byte items of an X slice per thread
constexpr int LDS_BYTES = O1_BYTES + O2_BYTES + X_BYTES + W_BYTES;
static_assert(8 * P2 <= O1_BYTES, "output staging aliases the out1 planes");
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace bnf

template <int N, typename F>
__device__ __forceinline__ void bnf_for(F&& f) {
  if constexpr (N > 0) {
    bnf_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

struct BottleneckParams {
  const bf16_t* x;      // [N][H][W][256]
  bf16_t* y;            // [N][H][W][256]
  const bf16_t* w1;     // [64][256]
  const bf16_t* w2;     // [64][3][3][64]
  const bf16_t* w3;     // [256][64]
  const float *b1, *b2, *b3;
  int H, W, tiles_x, tiles, total;
};

__global__ __launch_bounds__(bnf::NT) void bottleneck_fused_kernel(const BottleneckParams p) {
  using namespace bnf;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  char* s_o1 = smem;                       // out1 planes [8][WPAD]; later the output staging planes [8][NPIX]
  char* s_o2 = smem + O1_BYTES;            // out2 planes [8][NPIX]
  char* s_x = s_o2 + O2_BYTES;             // X slice planes [8][WPAD]: 64 input channels of the window
  char* s_w = s_x + X_BYTES;               // weight slice planes [8][64 rows]: 64 K-columns of 64 output channels
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, half = lane >> 5;

  // Global memory is read TWO slices ahead of its use, into registers (16 waves leave a thread 128 of them):
  //  * the 17 weight slices of a tile (4 of W1, 9 taps of W2, 4 passes of W3; 512 16-byte items each, threads 0..511) form one
  //    stream w_slice(0..16) that wraps into the next tile;
  //  * the 4 X slices of a tile (2560 items, 2.5 per thread); the next tile's first two are issued when phase 1 ends and travel
  //    under phases 2 and 3.
  const int w_row = (tid & 511) >> 3, w_c = tid & 7;
  const bool w_thread = tid < 512;
  auto fetch_w = [&](int i) -> uint4 {   // i-th weight slice of the per-tile stream
    if (!w_thread) return make_uint4(0, 0, 0, 0);
    const bf16_t* src = i < 4 ? p.w1 + (size_t)w_row * 256 + i * 64 : (i < 13 ? p.w2 + (size_t)w_row * 576 + (i - 4) * 64 : p.w3 + (size_t)((i - 13) * 64 + w_row) * 64);
    return *(const uint4*)(src + w_c * 8);
  };
  auto park_w = [&](uint4 v) { if (w_thread) *(uint4*)(s_w + w_c * PW + w_row * 16) = v; };
  uint4 wq0, wq1;            // weight slices i and i + 1 of the stream (i = the one about to be parked)
  uint4 xq0[XTRIPS], xq1[XTRIPS];
  auto fetch_x = [&](int t, int sl, uint4 (&dst)[XTRIPS]) {
    const int n = t / p.tiles, r0 = t - n * p.tiles, tyi = r0 / p.tiles_x;
    const int ty0 = tyi * TH, tx0 = (r0 - tyi * p.tiles_x) * TW;
    const bf16_t* X = p.x + (size_t)n * p.H * p.W * 256 + sl * 64;
#pragma unroll
    for (int j = 0; j < XTRIPS; ++j) {
      const int idx = tid + j * NT;
      const int px = idx >> 3, c = idx & 7;
      const int wy = px / WW, wx = px - wy * WW;
      const int gy = ty0 - 1 + wy, gx = tx0 - 1 + wx;
      dst[j] = make_uint4(0, 0, 0, 0);
      if (idx < WPAD * 8 && px < WPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) dst[j] = *(const uint4*)(X + ((size_t)gy * p.W + gx) * 256 + c * 8);
    }
  };
  auto park_x = [&](const uint4 (&src)[XTRIPS]) {
#pragma unroll
    for (int j = 0; j < XTRIPS; ++j) {
      const int idx = tid + j * NT;
      if (idx < WPAD * 8) *(uint4*)(s_x + (idx & 7) * P1 + (idx >> 3) * 16) = src[j];
    }
  };
  // residual rows of one phase-3 pass (64 channels of the 224 tile pixels): read coalesced (128 contiguous bytes per pixel), one pass
  // ahead, and parked in the X slice buffer, which is idle in phase 3 -- a per-lane 8-byte gather from global costs a cache-line
  // request per lane
  constexpr int RTRIPS = (NPIX * 8 + NT - 1) / NT;
  uint4 rq[RTRIPS];
  auto fetch_res = [&](const bf16_t* X, int ty0, int tx0, int pass) {
#pragma unroll
    for (int j = 0; j < RTRIPS; ++j) {
      const int idx = tid + j * NT;
      const int px = idx >> 3, c = idx & 7;
      const int py = px / TW, pxx = px - py * TW;
      const int gy = ty0 + py, gx = tx0 + pxx;
      rq[j] = make_uint4(0, 0, 0, 0);
      if (idx < NPIX * 8 && gy < p.H && gx < p.W) rq[j] = *(const uint4*)(X + ((size_t)gy * p.W + gx) * 256 + pass * 64 + c * 8);
    }
  };
  auto park_res = [&]() {
#pragma unroll
    for (int j = 0; j < RTRIPS; ++j) {
      const int idx = tid + j * NT;
      if (idx < NPIX * 8) *(uint4*)(s_x + (idx & 7) * P1 + (idx >> 3) * 16) = rq[j];
    }
  };
  if ((int)blockIdx.x < p.total) {
    fetch_x(blockIdx.x, 0, xq0);
    fetch_x(blockIdx.x, 1, xq1);
  }
  wq0 = fetch_w(0);
  wq1 = fetch_w(1);

  for (int t = blockIdx.x; t < p.total; t += gridDim.x) {
    const int n = t / p.tiles, r0 = t - n * p.tiles, tyi = r0 / p.tiles_x;
    const int ty0 = tyi * TH, tx0 = (r0 - tyi * p.tiles_x) * TW;
    const bf16_t* X = p.x + (size_t)n * p.H * p.W * 256;
    bf16_t* Y = p.y + (size_t)n * p.H * p.W * 256;
    const int tnext = t + (int)gridDim.x;
    int wi = 0;  // index of the weight slice in wq0
    auto next_w = [&]() {  // park slice wi, shift the queue, fetch slice wi + 2 (wrapping into the next tile's stream)
      park_w(wq0);
      wq0 = wq1;
      wq1 = fetch_w((wi + 2) % 17);
      ++wi;
    };

    // ================================================================ phase 1: conv1 over the window, K = 256 in 4 slices of 64
    {
      // 20 result tiles (channel tile ct, pixel tile pt) = 2 x 10: wave w owns tile w, waves 0..3 also tile 16 + w
      const bool two = wave < 4;
      const int ct0 = wave / 10, pt0 = wave - ct0 * 10, ct1 = (wave + 16) / 10, pt1 = (wave + 16) - ct1 * 10;
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      for (int sl = 0; sl < 4; ++sl) {
        if (sl & 1) park_x(xq1); else park_x(xq0);
        next_w();
        __syncthreads();
        // refill the X register that was just parked: slice sl + 2 of this tile, or slice sl - 2 of the next one
        if (sl + 2 < 4) { if (sl & 1) fetch_x(t, sl + 2, xq1); else fetch_x(t, sl + 2, xq0); }
        else if (tnext < p.total) { if (sl & 1) fetch_x(tnext, sl - 2, xq1); else fetch_x(tnext, sl - 2, xq0); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {   // the two accumulators interleave: no back-to-back dependent MFMAs
          const uint4 a0 = *(const uint4*)(s_w + (2 * ks + half) * PW + (ct0 * 32 + col) * 16);
          const uint4 b0 = *(const uint4*)(s_x + (2 * ks + half) * P1 + (pt0 * 32 + col) * 16);
          Mma<bf16_t>::run(acc0, a0, b0);
          if (two) {
            const uint4 a1 = *(const uint4*)(s_w + (2 * ks + half) * PW + (ct1 * 32 + col) * 16);
            const uint4 b1 = *(const uint4*)(s_x + (2 * ks + half) * P1 + (pt1 * 32 + col) * 16);
            Mma<bf16_t>::run(acc1, a1, b1);
          }
        }
        __syncthreads();
      }
      auto finish = [&](const f32x16& acc, int ct, int pt) {
        const int px = pt * 32 + col;
        const int wy = px / WW, wx = px - wy * WW;
        const int gy = ty0 - 1 + wy, gx = tx0 - 1 + wx;
        const bool inside = px < WPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch0 = ct * 32 + 8 * g + 4 * half;
          const float4 bb = *(const float4*)(p.b1 + ch0);
          uint2 o = make_uint2(pack2bf(fmaxf(acc[4 * g] + bb.x, 0.f), fmaxf(acc[4 * g + 1] + bb.y, 0.f)),
                               pack2bf(fmaxf(acc[4 * g + 2] + bb.z, 0.f), fmaxf(acc[4 * g + 3] + bb.w, 0.f)));
          if (!inside) o = make_uint2(0, 0);
          *(uint2*)(s_o1 + (ct * 4 + g) * P1 + px * 16 + half * 8) = o;
        }
      };
      finish(acc0, ct0, pt0);
      if (two) finish(acc1, ct1, pt1);
    }

    // ================================================================ phase 2: conv2 from the out1 planes, one tap (64 channels) per slice
    const bool active = wave < 14;                  // 14 result tiles = 2 channel tiles x 7 pixel tiles
    const int ct = wave / 7, pt = wave - ct * 7;
    const int m = pt * 32 + col;
    const int ty = m / TW, tx = m - ty * TW;
    {
      fetch_res(X, ty0, tx0, 0);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int tap = 0; tap < 9; ++tap) {
        next_w();
        __syncthreads();  // tap weights (and, at tap 0, the out1 planes) visible
        if (active) {
          const int kh = tap / 3, kw = tap - kh * 3;
          const int wp = (ty + kh) * WW + tx + kw;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint4 a = *(const uint4*)(s_w + (2 * ks + half) * PW + (ct * 32 + col) * 16);
            const uint4 b = *(const uint4*)(s_o1 + (2 * ks + half) * P1 + wp * 16);
            Mma<bf16_t>::run(acc, a, b);
          }
        }
        __syncthreads();
      }
      if (active) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch0 = ct * 32 + 8 * g + 4 * half;
          const float4 bb = *(const float4*)(p.b2 + ch0);
          const uint2 o = make_uint2(pack2bf(fmaxf(acc[4 * g] + bb.x, 0.f), fmaxf(acc[4 * g + 1] + bb.y, 0.f)),
                                     pack2bf(fmaxf(acc[4 * g + 2] + bb.z, 0.f), fmaxf(acc[4 * g + 3] + bb.w, 0.f)));
          *(uint2*)(s_o2 + (ct * 4 + g) * P2 + m * 16 + half * 8) = o;
        }
      }
    }

    // ================================================================ phase 3: conv3 + residual, 64 output channels per pass
    {
      char* s_st = s_o1;  // out1 is dead after phase 2: [8 chunks][NPIX][16 B] staging for one pass
      for (int pass = 0; pass < 4; ++pass) {
        next_w();
        park_res();
        __syncthreads();  // pass weights, residual rows (and, at pass 0, the out2 planes) visible; previous pass's staging fully stored
        if (pass + 1 < 4) fetch_res(X, ty0, tx0, pass + 1);
        if (active) {
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint4 a = *(const uint4*)(s_w + (2 * ks + half) * PW + (ct * 32 + col) * 16);
            const uint4 b = *(const uint4*)(s_o2 + (2 * ks + half) * P2 + m * 16);
            Mma<bf16_t>::run(acc, a, b);
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int ch0 = pass * 64 + ct * 32 + 8 * g + 4 * half;
            const float4 bb = *(const float4*)(p.b3 + ch0);
            float v[4] = {acc[4 * g] + bb.x, acc[4 * g + 1] + bb.y, acc[4 * g + 2] + bb.z, acc[4 * g + 3] + bb.w};
            const uint2 rr = *(const uint2*)(s_x + (ct * 4 + g) * P1 + m * 16 + half * 8);   // zero outside the image
            v[0] += __uint_as_float(rr.x << 16); v[1] += __uint_as_float(rr.x & 0xffff0000u);
            v[2] += __uint_as_float(rr.y << 16); v[3] += __uint_as_float(rr.y & 0xffff0000u);
            const uint2 o = make_uint2(pack2bf(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)), pack2bf(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)));
            *(uint2*)(s_st + (ct * 4 + g) * P2 + m * 16 + half * 8) = o;
          }
        }
        __syncthreads();
        for (int idx = tid; idx < NPIX * 8; idx += NT) {
          const int px = idx >> 3, c = idx & 7;
          const int py = px / TW, pxx = px - py * TW;
          const int oy = ty0 + py, ox = tx0 + pxx;
          if (oy < p.H && ox < p.W) *(uint4*)(Y + ((size_t)oy * p.W + ox) * 256 + pass * 64 + c * 8) = *(const uint4*)(s_st + c * P2 + px * 16);
        }
      }
      __syncthreads();  // the last pass's staging reads finish before the next tile rewrites the out1 planes
    }
  }
}

static inline int launch_bottleneck_fused(hipStream_t s, const void* x, void* y, const void* w1, const float* b1, const void* w2, const float* b2,
                                          const void* w3, const float* b3, int N, int H, int W) {
  BottleneckParams p;
  p.x = (const bf16_t*)x; p.y = (bf16_t*)y; p.w1 = (const bf16_t*)w1; p.w2 = (const bf16_t*)w2; p.w3 = (const bf16_t*)w3;
  p.b1 = b1; p.b2 = b2; p.b3 = b3; p.H = H; p.W = W;
  const int tiles_y = (H + bnf::TH - 1) / bnf::TH;
  p.tiles_x = (W + bnf::TW - 1) / bnf::TW;
  p.tiles = tiles_y * p.tiles_x;
  const long long total = (long long)p.tiles * N;
  if (total > 0x7fffffffLL) return 1;
  p.total = (int)total;
  const int grid = (int)(total < 256 ? total : 256);
  hipLaunchKernelGGL(bottleneck_fused_kernel, dim3(grid), dim3(bnf::NT), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
