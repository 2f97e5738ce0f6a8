// Test-time frame preprocessing on the device: one launch turns a batch of decoded uint8 BGR frames
// into the model's input tensor.  Fuses, per frame, what the reference does on the CPU one frame and one transform at a time:
//   CenterCrop window        mmdet/datasets/pipelines/transforms.py:1036-1052 (window chosen by the host: mcgaze_amd/pipeline.py)
//   Resize(keep_ratio=True)  transforms.py:216-242 -> mmcv.imrescale -> cv2.resize(INTER_LINEAR), 8-bit fixed-point bilinear
//   Normalize(to_rgb=True)   transforms.py:739-755 -> mmcv.imnormalize: (float(rgb) - mean) * (1 / std) in float32
//   Pad(size_divisor=32) + collate of the clip: zeros to the right / below (transforms.py:665-683)
//   DefaultFormatBundle      formatting.py:229-231: HWC -> CHW
// The resize arithmetic is OpenCV's published INTER_LINEAR path for 8-bit images, restated (not linked): source coordinate
// (d + 0.5) * (src / dst) - 0.5 in double, cast to float, floor + fraction, clamped to the image with zero fraction at the
// borders; coefficients rounded to 11-bit fixed point (round half to even); horizontal pass in 32-bit integers; vertical pass
// uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2).  Integer work is bit-exact against
// oracle/preprocess_oracle.py by construction; floating-point contraction is disabled so the float steps are too.
// Memory-bound and tiny (a 64-clip batch reads <= 0.3 GB of pixels and writes 270 MB): one thread per output pixel, channel
// planes written coalesced.
#include "common.hpp"

struct LinCoef {
  int s0, s1, a0, a1;
};

__device__ __forceinline__ LinCoef lin_coef(int d, int dst, int src) {
#pragma clang fp contract(off)
  const double scale = 1.0 / ((double)dst / (double)src);
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= src - 1) { f = 0.f; s = src - 1; }
  LinCoef c;
  c.s0 = s;
  c.s1 = min(s + 1, src - 1);
  c.a1 = __float2int_rn(f * 2048.f);
  c.a0 = __float2int_rn((1.f - f) * 2048.f);
  return c;
}

__global__ __launch_bounds__(256) void preprocess_kernel(const mcg_frame_desc* __restrict__ frames, float* __restrict__ dst, int pad_h,
                                                         int pad_w, float m0, float m1, float m2, float s0, float s1, float s2, int to_rgb) {
#pragma clang fp contract(off)
  const mcg_frame_desc fd = frames[blockIdx.y];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= pad_h * pad_w) return;
  const int y = idx / pad_w, x = idx - y * pad_w;
  float* out = dst + (size_t)blockIdx.y * 3 * pad_h * pad_w + idx;
  const size_t plane = (size_t)pad_h * pad_w;
  if (y >= fd.out_h || x >= fd.out_w) {
    out[0] = 0.f; out[plane] = 0.f; out[2 * plane] = 0.f;
    return;
  }
  const LinCoef cx = lin_coef(x, fd.out_w, fd.crop_w), cy = lin_coef(y, fd.out_h, fd.crop_h);
  const unsigned char* base = (const unsigned char*)fd.src + (size_t)fd.crop_y * fd.src_pitch + (size_t)fd.crop_x * 3;
  const unsigned char* r0 = base + (size_t)cy.s0 * fd.src_pitch;
  const unsigned char* r1 = base + (size_t)cy.s1 * fd.src_pitch;
  int v[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int h0 = (int)r0[cx.s0 * 3 + c] * cx.a0 + (int)r0[cx.s1 * 3 + c] * cx.a1;
    const int h1 = (int)r1[cx.s0 * 3 + c] * cx.a0 + (int)r1[cx.s1 * 3 + c] * cx.a1;
    v[c] = (((cy.a0 * (h0 >> 4)) >> 16) + ((cy.a1 * (h1 >> 4)) >> 16) + 2) >> 2;
  }
  const int c0 = to_rgb ? 2 : 0, c2 = to_rgb ? 0 : 2;  // source is BGR
  out[0] = ((float)v[c0] - m0) * s0;
  out[plane] = ((float)v[1] - m1) * s1;
  out[2 * plane] = ((float)v[c2] - m2) * s2;
}

extern "C" int mcg_preprocess_frames(mcg_stream stream, const mcg_frame_desc* frames_dev, int num_frames, float* dst, int pad_h, int pad_w,
                                     const float mean[3], const float stdinv[3], int to_rgb) {
  MCG_CHECK_ARG(frames_dev && dst && mean && stdinv, "mcg_preprocess_frames: null pointer");
  MCG_CHECK_ARG(num_frames >= 0 && pad_h > 0 && pad_w > 0, "mcg_preprocess_frames: bad sizes n=%d pad=%dx%d", num_frames, pad_h, pad_w);
  MCG_CHECK_ARG(num_frames <= 65535, "mcg_preprocess_frames: at most 65535 frames per call (got %d)", num_frames);
  if (num_frames == 0) return MCG_OK;
  dim3 grid((pad_h * pad_w + 255) / 256, num_frames);
  hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, (hipStream_t)stream, frames_dev, dst, pad_h, pad_w, mean[0], mean[1], mean[2],
                     stdinv[0], stdinv[1], stdinv[2], to_rgb);
  MCG_CHECK_LAUNCH("mcg_preprocess_frames");
  return MCG_OK;
}
