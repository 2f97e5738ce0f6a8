// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (+ folded BN bias, ReLU), bf16 NHWC: layer1's conv2
// (resnet.py:262-266), three launches per forward on the 56x56 maps.  In the generic implicit-GEMM kernel this shape is the worst
// of the trunk (550 TFLOP/s, 1.9 TB/s): with only 64 output channels every activation byte is fetched nine times through
// L2 -> LDS-DMA for 64 MACs each, and that stream, not HBM or the matrix pipe, bounds it.  Here the input window of a spatial tile
// is staged ONCE and the nine taps are taken from LDS by address, as in stem_fused.hpp:
//
//   * a workgroup (4 waves) owns an 8 x 28 tile of output pixels (224 = 7 MFMA row blocks); its 10 x 30 input window (64 channels =
//     eight 16-byte chunks per pixel) sits in LDS as eight chunk PLANES [chunk][pixel][16 B] -- an A fragment (32 consecutive
//     pixels x one chunk) is then 512 contiguous bytes, conflict-free for any tap, and tap (kh, kw) / channel group j are pure
//     immediates: (kh * 30 + kw) * 16 + 2 j * plane.  The plane stride is padded by 32 bytes so that the eight chunks of one pixel
//     (written by eight neighbouring threads from one coalesced 128-byte global read) fall on different banks;
//   * the 64 x 576 weight matrix lives in registers: a wave keeps its 32-channel half for all 36 K-steps (144 VGPRs; the
//     kernel runs two waves per SIMD, which is also what its LDS allows);
//   * K order = tap-major, channels ascending, 16 per MFMA -- the generic kernel's order, so results are BIT-IDENTICAL to it;
//   * bias + ReLU in f32, rounded by v_cvt_pk_bf16_f32, staged through LDS as [pixel][64] and written as coalesced 16-byte chunks;
//   * persistent grid (2 workgroups per CU), the next tile's window is fetched into registers before the MFMA phase of the
//     current one.
#pragma once
#include "common.hpp"

namespace c64 {
constexpr int TH = 8, TW = 28;
constexpr int WH = TH + 2, WW = TW + 2;        // window
constexpr int NPIX = TH * TW;                   // 224
constexpr int MB = NPIX / 32;                   // 7 row blocks
constexpr int WPIX = WH * WW;                   // 300
constexpr int PLANE = WPIX * 16 + 32;           // bytes, padded (see above)
constexpr int IN_BYTES = 8 * PLANE;
constexpr int OUT_BYTES = NPIX * 128;
constexpr int KS = 36;                          // 9 taps x 4 channel groups of 16
constexpr int TRIPS = (WPIX * 8 + 255) / 256;   // 16-byte chunks per thread per window
static_assert(NPIX % 32 == 0, "tile must be whole MFMA row blocks");
}  // namespace c64

template <typename F>   // number format of the 2-byte elements: bf16_t or f16_t (pointers stay raw 2-byte pointers)
__global__ __launch_bounds__(256) void conv3x3_c64_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, const float* __restrict__ bias,
                                                          bf16_t* __restrict__ y, int H, int W, int tiles_x, int tiles, int total, int relu) {
  using namespace c64;
  __shared__ __attribute__((aligned(16))) char s_in[IN_BYTES];
  __shared__ __attribute__((aligned(16))) char s_out[OUT_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = lane & 31, half = lane >> 5;

  // weights of this wave's 32 output channels: W is [64][3][3][64] = [64][576], K index = (kh*3 + kw)*64 + c
  const int nb = wave >> 1;
  uint4 bfrag[KS];
  {
    const bf16_t* wl = w + (size_t)(nb * 32 + row) * (KS * 16) + half * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) bfrag[ks] = *(const uint4*)(wl + ks * 16);
  }
  const float b = bias[nb * 32 + row];
  const int mb0 = (wave & 1) ? 4 : 0, mb1 = (wave & 1) ? MB : 4;  // row blocks 0..3 | 4..6

  uint4 v[TRIPS];
  auto origin = [&](int t, int& n, int& ty0, int& tx0) {
    n = t / tiles;
    const int r = t - n * tiles, ty = r / tiles_x;
    ty0 = ty * TH;
    tx0 = (r - ty * tiles_x) * TW;
  };
  auto fetch = [&](int t) {
    int n, ty0, tx0;
    origin(t, n, ty0, tx0);
    const bf16_t* f0 = x + (size_t)n * H * W * 64;
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      const int p = idx >> 3, c = idx & 7;
      const int wy = p / WW, wx = p - wy * WW;
      const int gy = ty0 - 1 + wy, gx = tx0 - 1 + wx;
      v[j] = make_uint4(0, 0, 0, 0);
      if (idx < WPIX * 8 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v[j] = *(const uint4*)(f0 + ((size_t)gy * W + gx) * 64 + c * 8);
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int j = 0; j < TRIPS; ++j) {
      const int idx = tid + j * 256;
      if (idx < WPIX * 8) *(uint4*)(s_in + (idx & 7) * PLANE + (idx >> 3) * 16) = v[j];
    }
  };

  int t = blockIdx.x;
  if (t < total) fetch(t);
  for (; t < total; t += gridDim.x) {
    park();
    __syncthreads();
    const int tn = t + gridDim.x;
    if (tn < total) fetch(tn);
    int n, ty0, tx0;
    origin(t, n, ty0, tx0);

#pragma unroll 1
    for (int mb = mb0; mb < mb1; ++mb) {
      const int m = mb * 32 + row;
      const int ty = m / TW, tx = m - ty * TW;
      const char* a0 = s_in + half * PLANE + (ty * WW + tx) * 16;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int tap = ks >> 2, j = ks & 3;
        const uint4 a = *(const uint4*)(a0 + ((tap / 3) * WW + tap % 3) * 16 + 2 * j * PLANE);
        Mma<F>::run(acc, a, bfrag[ks]);
      }
      char* orow = s_out + (mb * 32 + 4 * half) * 128 + (nb * 32 + row) * 2;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        float v0 = acc[r] + b, v1 = acc[r + 1] + b;
        if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        const uint32_t pk = H16<F>::pack2(v0, v1);
        *(bf16_t*)(orow + ((r & 3) + 8 * (r >> 2)) * 128) = (bf16_t)(pk & 0xffffu);
        *(bf16_t*)(orow + (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * 128) = (bf16_t)(pk >> 16);
      }
    }
    __syncthreads();  // output tile complete; the window is free for the next tile

    bf16_t* y0 = y + (size_t)n * H * W * 64;
    for (int idx = tid; idx < NPIX * 8; idx += 256) {
      const int p = idx >> 3, c = idx & 7;
      const int ty = p / TW, tx = p - ty * TW;
      const int gy = ty0 + ty, gx = tx0 + tx;
      if (gy < H && gx < W) *(uint4*)(y0 + ((size_t)gy * W + gx) * 64 + c * 8) = *(const uint4*)(s_out + p * 128 + c * 16);
    }
    // the next iteration's park() writes s_in only; its barrier orders these s_out reads before the next tile's s_out writes
  }
}

static inline bool conv3x3_c64_applicable(int KH, int KW, int stride, int pad, int Cin, int Cout, bool has_residual, bool has_x2) {
  return KH == 3 && KW == 3 && stride == 1 && pad == 1 && Cin == 64 && Cout == 64 && !has_residual && !has_x2;
}
static inline int launch_conv3x3_c64(hipStream_t s, const void* x, const void* w, const float* bias, void* y, int N, int H, int W, int relu, bool fp16 = false) {
  const int tiles_y = (H + c64::TH - 1) / c64::TH, tiles_x = (W + c64::TW - 1) / c64::TW;
  const long long total = (long long)tiles_y * tiles_x * N;
  if (total > 0x7fffffffLL) return 1;
  const int grid = (int)(total < 512 ? total : 512);
  if (fp16) hipLaunchKernelGGL(conv3x3_c64_kernel<f16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, H, W, tiles_x,
                               tiles_y * tiles_x, (int)total, relu);
  else hipLaunchKernelGGL(conv3x3_c64_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, H, W, tiles_x,
                          tiles_y * tiles_x, (int)total, relu);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
