// pw_single.hpp's persistent streaming kernel for the HBM-bound 1x1 convolutions, in the MCG_F16X3 arithmetic (f32 activations, weights
// as fp16 high / low fragments, three fp16 MFMAs per product): the P2 lateral 256 -> 256 with its nearest-upsampled top-down term
// (fpn.py:164-174) and layer3's conv3 256 -> 1024 + residual (resnet.py:283-298) of the f16x3 engine,
//
//     y = [relu]( A . W^T + b (+ res | + up(res)) )        A [M][256] f32, W [N][256]; per workgroup 128 output channels
//
// The generic x3 contraction kernel runs these at 3.3 - 3.5 TB/s of algorithmic bytes (one workgroup per CU; a 256 x 256 tile's
// 8 K-tiles sit between a cold prologue and a 512 KB epilogue).  Here a workgroup keeps its 128 x 256 slice of the weights -- high and
// low fragments, 128 VGPRs per wave -- in registers and walks 32-pixel tiles; the next tile's A rows and residual rows travel HBM -> LDS by
// `buffer_load ... lds` while the current tile is contracted and stored; two workgroups per CU; N / 128 workgroups on one XCD share
// an A tile through L2.  The A fragment (eight f32 from the swizzled LDS tile) is split in registers -- 24 VALU per three MFMAs, far
// under what the tile's bytes take.  K order, term order and f32 rounding points are the contraction kernel's.
#pragma once
#include "pw_single.hpp"

template <int KS, int RES, int NSPLIT>
__global__ __launch_bounds__(256, 2) void pw_single_x3_kernel(const PwSingleParams p) {
  constexpr int PX = 32, K = 16 * KS, N = 128, NF = N * NSPLIT, AROWB = 4 * K, YROWB = 4 * N, GROWB = 4 * NF, ACH = AROWB / 16, YCH = YROWB / 16;
  constexpr int ABYTES = PX * AROWB, YBYTES = PX * YROWB;
  constexpr int NYB = RES ? 2 : 1;
  constexpr int AOFF = NYB * YBYTES;
  constexpr int A_PIECES = ABYTES / 1024 / 4, Y_PIECES = YBYTES / 1024 / 4;
  static_assert(ACH <= 64 && YCH <= 64 && A_PIECES >= 1 && Y_PIECES >= 1 && AOFF + ABYTES <= 80 * 1024, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = blockIdx.x, nwalkers = gridDim.x / NSPLIT;
  // ids 8 apart sit on the same XCD; many_slices (dynamic_layer: 256 slices): workgroup id = walker * NSPLIT + slice
  const int nsp = p.many_slices ? bid % NSPLIT : (bid >> 3) % NSPLIT, walker = p.many_slices ? bid / NSPLIT : (bid / (8 * NSPLIT)) * 8 + (bid & 7);
  const float wsc = p.wscale > 0.f ? p.wscale : 1.f;   // exact power of two: the weights were packed pre-scaled by its inverse
  float4 breg[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) breg[q] = *(const float4*)(p.bias + nsp * N + wave * 32 + 8 * q + 4 * half);
  auto a_off = [](int r, int chunk) { return r * AROWB + ((chunk ^ (r & (ACH >= 32 ? 31 : 15))) << 4); };
  auto y_off = [](int r, int chunk) { return r * YROWB + ((chunk ^ (r & (YCH >= 32 ? 31 : 15))) << 4); };
  // weights of this wave's 32 channels, once per workgroup: split fragment-major [tile][K-step][high, low][lane][8] (packing.frag_major_split)
  uint4 w[KS][2];
  {
    const char* wb = (const char*)p.wf + ((size_t)(nsp * 4 + wave) * KS * 2 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      w[ks][0] = *(const uint4*)(wb + (size_t)(2 * ks) * 1024);
      w[ks][1] = *(const uint4*)(wb + (size_t)(2 * ks + 1) * 1024);
    }
  }
  const u32x4 srd_a = make_srd(p.a), srd_r = make_srd(RES ? p.res : p.a);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int A_LPR = ACH < 64 ? ACH : 64, Y_LPR = YCH < 64 ? YCH : 64;
  constexpr int A_RPP = 64 / A_LPR, Y_RPP = 64 / Y_LPR;
  const int ntiles = (p.M + PX - 1) / PX;
  auto issue_a = [&](int tile) {
    const int rows_left = p.M - tile * PX;
    const uint32_t so = (uint32_t)tile * ABYTES;
    static_for<A_PIECES>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int row = (wave * A_PIECES + J) * A_RPP + lane / A_LPR, pos = lane % A_LPR;
      const uint32_t vo = row < rows_left ? (uint32_t)(row * AROWB + ((pos ^ (row & (ACH >= 32 ? 31 : 15))) << 4)) : MCG_OOB_OFFSET;
      lds_dma16<AOFF + J * 1024>(vo, srd_a, so, lds_base + wave * (A_PIECES * 1024));
    });
  };
  const int HoWo = p.Ho * p.Wo;
  auto issue_res = [&](int tile, int buf) {
    const int rows_left = p.M - tile * PX;
    static_for<Y_PIECES>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int row = (wave * Y_PIECES + J) * Y_RPP + lane / Y_LPR, pos = lane % Y_LPR;
      const int chunk = pos ^ (row & (YCH >= 32 ? 31 : 15));
      uint32_t vo = MCG_OOB_OFFSET, so = 0;
      if (RES == 1) {
        so = (uint32_t)tile * (PX * GROWB) + nsp * YROWB;
        if (row < rows_left) vo = (uint32_t)(row * GROWB + (chunk << 4));
      } else if (row < rows_left) {   // nearest-upsample gather (torch: src = min(floor(dst * in / out), in - 1))
        const int m = tile * PX + row;
        const int f = m / HoWo, rem = m - f * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int sh = min((int)floorf(ho * p.rscale_h), p.Hr - 1), sw = min((int)floorf(wo * p.rscale_w), p.Wr - 1);
        vo = (uint32_t)((((long long)f * p.Hr + sh) * p.Wr + sw) * GROWB + nsp * YROWB + (chunk << 4));
      }
      lds_dma16<J * 1024>(vo, srd_r, so, lds_base + buf * YBYTES + wave * (Y_PIECES * 1024));
    });
  };
  if (walker < ntiles) {
    if (RES) issue_res(walker, 0);
    issue_a(walker);
  }
  int it = 0;
  for (int tile = walker; tile < ntiles; tile += nwalkers, ++it) {
    const long long m0 = (long long)tile * PX;
    char* s_y = smem + (RES ? (it & 1) * YBYTES : 0);
    const char* s_a = smem + AOFF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the tile have landed (and its earlier stores left)
    __syncthreads();                                           // everyone's pieces landed; the other y buffer is free
    const bool more = tile + nwalkers < ntiles;
    if (more && RES) issue_res(tile + nwalkers, (it + 1) & 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 xh, xl;
      split_f32x8(*(const uint4*)(s_a + a_off(px, 4 * ks + 2 * half)), *(const uint4*)(s_a + a_off(px, 4 * ks + 2 * half + 1)), xh, xl);
      acc = x3_mfma(__builtin_bit_cast(bf16x8, w[ks][0]), xl, acc);   // activation low x weight high first: the contraction kernel's order
      acc = x3_mfma(__builtin_bit_cast(bf16x8, w[ks][1]), xh, acc);
      acc = x3_mfma(__builtin_bit_cast(bf16x8, w[ks][0]), xh, acc);
    }
    __syncthreads();                                           // the A tile has been consumed by every wave (and, RES == 0: the previous y tile is stored)
    if (more) issue_a(tile + nwalkers);                        // single A tile (two would not leave room for two workgroups per CU): refilled under the epilogue
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = wave * 32 + 8 * q + 4 * half;
      char* slot = s_y + y_off(px, c0 >> 2);
      float4 v = make_float4(acc[4 * q] * wsc + breg[q].x, acc[4 * q + 1] * wsc + breg[q].y, acc[4 * q + 2] * wsc + breg[q].z, acc[4 * q + 3] * wsc + breg[q].w);
      if (RES) {
        const float4 rr = *(const float4*)slot;
        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
      }
      if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      *(float4*)slot = v;
    }
    __syncthreads();                                           // y tile complete
    for (int idx = tid; idx < PX * YCH; idx += 256) {
      const int r = idx / YCH, c = idx - r * YCH;
      if (m0 + r < p.M) *(uint4*)((float*)p.y + (m0 + r) * NF + nsp * N + c * 4) = *(const uint4*)(s_y + y_off(r, c));
    }
  }
}

// (K, N) of the f16x3 trunk served here: 256 -> 256 (P2 lateral) and 256 -> 1024 (layer3's conv3)
static inline bool pw_single_x3_applicable(int K, int N, int res_mode, long long M, long long res_rows) {
  return K == 256 && (N == 256 || N == 1024) && M >= 64 * 1024 && M * 4 * (K > N ? K : N) < MCG_DMA_MAX_BYTES && res_rows * 4 * N < MCG_DMA_MAX_BYTES;
}
template <int KS, int RES, int NSPLIT>
static inline void launch_pw_single_x3_t(hipStream_t s, const PwSingleParams& p) {
  constexpr int kLds = (RES ? 2 : 1) * 32 * 512 + 32 * 64 * KS;
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    (void)hipFuncSetAttribute((const void*)pw_single_x3_kernel<KS, RES, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int ntiles = (p.M + 31) / 32, unit = 8 * NSPLIT;
  int wgs = 2 * cus_of[dev] / unit * unit;                      // two workgroups per CU, whole groups of NSPLIT slices x 8 XCDs
  const int need = (ntiles + 7) / 8 * unit;
  if (wgs < unit) wgs = unit;
  hipLaunchKernelGGL((pw_single_x3_kernel<KS, RES, NSPLIT>), dim3(need < wgs ? need : wgs), dim3(256), kLds, s, p);
}
static inline int launch_pw_single_x3(hipStream_t s, const PwSingleParams& p, int N, int res_mode) {
  if (N == 256) {
    if (res_mode == 0) launch_pw_single_x3_t<16, 0, 2>(s, p);
    else if (res_mode == 1) launch_pw_single_x3_t<16, 1, 2>(s, p);
    else launch_pw_single_x3_t<16, 2, 2>(s, p);
  } else {
    if (res_mode == 0) launch_pw_single_x3_t<16, 0, 8>(s, p);
    else if (res_mode == 1) launch_pw_single_x3_t<16, 1, 8>(s, p);
    else launch_pw_single_x3_t<16, 2, 8>(s, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// DynamicConv's `dynamic_layer` for the f16x3 engine (transformer.py:1131-1134): y[M][32768] = x[M][256] . W^T + b on M = 1344 tokens,
// 176 MB of f32 output: 256 slices of 128 columns, each slice's split weights resident in the registers of two workgroups that share
// the token tiles (pw_single.hpp's launch_pw_dyn).  Bit-identical to the x3 contraction kernel.
static inline int launch_pw_dyn_x3(hipStream_t s, PwSingleParams p) {
  constexpr int KS = 16, NSPLIT = 256, kLds = 32 * 512 + 32 * 64 * KS;
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    (void)hipFuncSetAttribute((const void*)pw_single_x3_kernel<KS, 0, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int ntiles = (p.M + 31) / 32;
  int walkers = 2 * cus_of[dev] / NSPLIT;                       // two workgroups per CU
  if (walkers < 1) walkers = 1;
  if (walkers > ntiles) walkers = ntiles;
  p.many_slices = 1;
  hipLaunchKernelGGL((pw_single_x3_kernel<KS, 0, NSPLIT>), dim3(walkers * NSPLIT), dim3(256), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
