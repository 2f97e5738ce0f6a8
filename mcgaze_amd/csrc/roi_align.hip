// RoIAlign over all pyramid levels in ONE launch, level chosen per box on the device (no host
// nonzero()/sync per level as in single_level_roi_extractor.py:86-104).
//
// Definition implemented (mmcv.ops.RoIAlign, pool_mode='avg', aligned=True, sampling_ratio=2,
// output 7x7 -- the published mmcv-full 1.4.x / Detectron2 kernel):
//   start = coord*spatial_scale - 0.5; bin = (end-start)/7; sample (iy,ix) of bin (ph,pw) at
//   y = start_h + ph*bin_h + (iy+.5)*bin_h/2;  a sample with y < -1 or y > H (x likewise) adds 0;
//   otherwise y = max(y,0); y_low = int(y); if y_low >= H-1: y_low = y_high = H-1, y = y_low;
//   value = bilinear; output = mean of the 4 samples.
// Features are NHWC so one sample touches C contiguous channels; output is [box][49][C].
#include "common.hpp"

struct RoiLevels {
  const void* feat[4];
  int h[4], w[4];
  float scale[4];
};

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(RoiLevels lv, int C, const float* __restrict__ boxes, int boxes_per_frame,
                                                        T* __restrict__ out, int32_t* __restrict__ levels_out, float finest_scale) {
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int P = 7, S = 2, NS = P * S;
  __shared__ int s_lo[2][NS], s_hi[2][NS];
  __shared__ float s_l[2][NS], s_h[2][NS];
  __shared__ int s_valid[2][NS];
  const int box = blockIdx.x, tid = threadIdx.x;
  const float x1 = boxes[box * 4 + 0], y1 = boxes[box * 4 + 1], x2 = boxes[box * 4 + 2], y2 = boxes[box * 4 + 3];
  // map_roi_levels (single_level_roi_extractor.py:51-54)
  const float sc = sqrtf((x2 - x1) * (y2 - y1));
  int level = (int)fminf(fmaxf(floorf(log2f(sc / finest_scale + 1e-6f)), 0.f), 3.f);
  if (!(sc == sc)) level = 0;  // NaN guard
  const int H = lv.h[level], W = lv.w[level];
  const float ss = lv.scale[level];
  if (tid < 2 * NS) {
    const int axis = tid / NS, i = tid % NS;  // axis 0 = y, 1 = x
    const float start = (axis == 0 ? y1 : x1) * ss - 0.5f, end = (axis == 0 ? y2 : x2) * ss - 0.5f;
    const float bin = (end - start) / (float)P;
    const int L = axis == 0 ? H : W;
    float c = start + (float)(i / S) * bin + ((float)(i % S) + 0.5f) * bin / (float)S;
    const int valid = !(c < -1.0f || c > (float)L);
    c = fmaxf(c, 0.f);
    int lo = (int)c, hi;
    if (lo >= L - 1) {
      lo = hi = L - 1;
      c = (float)lo;
    } else {
      hi = lo + 1;
    }
    const float l = c - (float)lo;
    s_lo[axis][i] = lo; s_hi[axis][i] = hi; s_l[axis][i] = l; s_h[axis][i] = 1.f - l; s_valid[axis][i] = valid && (c == c);
  }
  if (tid == 0) {
    if (levels_out) levels_out[box] = level;
  }
  __syncthreads();
  const int frame = box / boxes_per_frame;
  const T* __restrict__ F = (const T*)lv.feat[level] + (long long)frame * H * W * C;
  const int groups = C / EPC, bins_per_pass = 256 / groups;
  const int cg = tid % groups, slot = tid / groups;
  for (int bin = slot; bin < P * P; bin += bins_per_pass) {
    const int ph = bin / P, pw = bin % P;
    float acc[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int iy = 0; iy < S; ++iy) {
      const int yi = ph * S + iy;
#pragma unroll
      for (int ix = 0; ix < S; ++ix) {
        const int xi = pw * S + ix;
        if (!(s_valid[0][yi] && s_valid[1][xi])) continue;
        const int yl = s_lo[0][yi], yh = s_hi[0][yi], xl = s_lo[1][xi], xh = s_hi[1][xi];
        const float ly = s_l[0][yi], hy = s_h[0][yi], lx = s_l[1][xi], hx = s_h[1][xi];
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        float v1[EPC], v2[EPC], v3[EPC], v4[EPC];
        chunk_to_f32(*(const uint4*)(F + ((long long)yl * W + xl) * C + cg * EPC), v1, (T*)nullptr);
        chunk_to_f32(*(const uint4*)(F + ((long long)yl * W + xh) * C + cg * EPC), v2, (T*)nullptr);
        chunk_to_f32(*(const uint4*)(F + ((long long)yh * W + xl) * C + cg * EPC), v3, (T*)nullptr);
        chunk_to_f32(*(const uint4*)(F + ((long long)yh * W + xh) * C + cg * EPC), v4, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) acc[e] += w1 * v1[e] + w2 * v2[e] + w3 * v3[e] + w4 * v4[e];
      }
    }
#pragma unroll
    for (int e = 0; e < EPC; ++e) acc[e] *= (1.0f / (float)(S * S));
    *(uint4*)(out + ((long long)box * (P * P) + bin) * C + cg * EPC) = f32_to_chunk(acc, (T*)nullptr);
  }
}

int launch_roi_align(hipStream_t s, mcg_dtype dt, const void* const feats[4], const int feat_h[4], const int feat_w[4],
                     const int strides[4], int C, const float* boxes, int num_boxes, int boxes_per_frame, void* out,
                     int32_t* levels_out) {
  RoiLevels lv;
  for (int i = 0; i < 4; ++i) {
    lv.feat[i] = feats[i]; lv.h[i] = feat_h[i]; lv.w[i] = feat_w[i];
    lv.scale[i] = 1.0f / (float)strides[i];
  }
  const int epc = mcg_is16(dt) ? 8 : 4;
  MCG_CHECK_ARG(C % epc == 0 && C / epc <= 256 && 256 % (C / epc) == 0, "roi_align: unsupported channel count %d", C);
  if (dt == MCG_BF16)
    hipLaunchKernelGGL(roi_align_kernel<bf16_t>, dim3(num_boxes), dim3(256), 0, s, lv, C, boxes, boxes_per_frame, (bf16_t*)out, levels_out, 56.f);
  else if (dt == MCG_F16)
    hipLaunchKernelGGL(roi_align_kernel<f16_t>, dim3(num_boxes), dim3(256), 0, s, lv, C, boxes, boxes_per_frame, (f16_t*)out, levels_out, 56.f);
  else
    hipLaunchKernelGGL(roi_align_kernel<float>, dim3(num_boxes), dim3(256), 0, s, lv, C, boxes, boxes_per_frame, (float*)out, levels_out, 56.f);
  MCG_CHECK_LAUNCH("roi_align");
  return MCG_OK;
}

extern "C" int mcg_roi_align(mcg_stream s, mcg_dtype dt, const void* const feats[4], const int feat_h[4], const int feat_w[4],
                             const int strides[4], int C, const float* boxes, int num_boxes, int boxes_per_frame,
                             void* out, int32_t* levels_out) {
  MCG_CHECK_ARG(feats && feat_h && feat_w && strides && boxes && out, "mcg_roi_align: null pointer");
  MCG_CHECK_ARG(num_boxes > 0 && boxes_per_frame > 0, "mcg_roi_align: empty box set");
  for (int i = 0; i < 4; ++i) MCG_CHECK_ARG(feats[i] && feat_h[i] > 0 && feat_w[i] > 0 && strides[i] > 0, "mcg_roi_align: bad level %d", i);
  return launch_roi_align((hipStream_t)s, dt, feats, feat_h, feat_w, strides, C, boxes, num_boxes, boxes_per_frame, out, levels_out);
}
