// HBM-bound 1x1 convolutions of the bf16 engine -- layer2's conv1 512 -> 128 and conv3 128 -> 512 + residual, layer3's conv3
// 256 -> 1024 + residual and first conv1 512 -> 256, the FPN laterals P2 / P3 with their nearest-upsampled top-down term, and the
// decoder's `dynamic_layer` (256 -> 32768 on 1344 tokens) -- as a persistent streaming kernel: pw_pair.hpp's structure with one
// contraction,
//
//     y = [relu]( A . W^T + b (+ res | + up(res)) )        A [M][K], W [N][K]; per workgroup a slice of W of N_wg * K * 2 B <= 128 KB
//
// The generic contraction kernel runs these at 3.2-4.6 TB/s (a fresh prologue / epilogue per 256-row tile, K of 2-8 K-tiles); here a
// workgroup keeps the whole weight matrix in REGISTERS (128 VGPRs per wave), walks 32-pixel tiles with a grid-stride loop, and the
// next tile's A rows and residual rows travel HBM -> LDS by `buffer_load ... lds` while the current tile is contracted and stored:
// two workgroups per CU, no global load inside the loop other than those DMAs (biases sit in LDS -- a load queued behind the next
// tile's DMA would drain the prefetch, pw_pair.hpp), transposed MFMAs so that bias / residual / ReLU / bf16 rounding work on 8-byte
// pieces in the LDS tile that the coalesced store reads.  K order and rounding points are the generic kernel's: bit-identical.
#pragma once
#include "pw_pair.hpp"

struct PwSingleParams {
  const void* a;            // [M][K] bf16, rows contiguous
  const void* res;          // RES 1: [M][N]; RES 2: [frames][Hr][Wr][N] gathered at (y * Hr / Ho, x * Wr / Wo) (F.interpolate nearest)
  const void* wf;           // [N/32][K/16][64][8] fragment-major
  const float* bias;        // [N]
  void* y;                  // [M][N]
  int M, relu, Ho, Wo, Hr, Wr;
  float rscale_h, rscale_w;
  int many_slices;          // 1: far more channel slices than XCDs (dynamic_layer: 128): workgroup id = walker * NSPLIT + slice
  float wscale;             // f16x3 (pw_single_x3.hpp): accumulators x this power of two before the bias (pre-scaled weights); 0 = 1
};

// K = 16 KS; a workgroup owns N = 128 TPW output channels (TPW 32-channel tiles per wave, 4 waves) of the NSPLIT * N the layer has:
// NSPLIT workgroups on one XCD (one L2) walk the same pixel tiles, each with its own slice of the weights in registers -- the A tile
// comes from HBM once and from L2 for the others.  RES: 0 none, 1 same-shape add, 2 nearest-upsample add.
template <typename F, int KS, int TPW, int RES, int NSPLIT>
__global__ __launch_bounds__(256, 2) void pw_single_kernel(const PwSingleParams p) {
  constexpr int PX = 32, K = 16 * KS, N = 128 * TPW, NF = N * NSPLIT, AROWB = 2 * K, YROWB = 2 * N, GROWB = 2 * NF, ACH = AROWB / 16, YCH = YROWB / 16;
  constexpr int ABYTES = PX * AROWB, YBYTES = PX * YROWB;
  constexpr int NYB = RES ? 2 : 1;                                           // y / residual tiles (the residual of the next tile lands early)
  constexpr bool BIAS_REGS = TPW == 1;                                       // one channel tile per wave: its 16 biases stay in registers (read before any DMA)
  constexpr int NAB = (2 * ABYTES + NYB * YBYTES + (BIAS_REGS ? 0 : N * 4) <= 80 * 1024) ? 2 : 1;   // A tiles: double-buffered when two workgroups still fit a CU
  constexpr int AOFF = NYB * YBYTES, BOFF = AOFF + NAB * ABYTES;
  constexpr int A_PIECES = ABYTES / 1024 / 4, Y_PIECES = YBYTES / 1024 / 4;   // 1 KiB DMA pieces per wave
  static_assert(A_PIECES >= 1 && Y_PIECES >= 1, "tile geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_bias = (float*)(smem + BOFF);
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blockIdx -> (walker, channel slice): ids 8 apart sit on the same XCD (round-robin dispatch over the 8 XCDs)
  const int bid = blockIdx.x, nwalkers = gridDim.x / NSPLIT;
  const int nsp = p.many_slices ? bid % NSPLIT : (bid >> 3) % NSPLIT, walker = p.many_slices ? bid / NSPLIT : (bid / (8 * NSPLIT)) * 8 + (bid & 7);
  float4 breg[4];
  if constexpr (BIAS_REGS) {
#pragma unroll
    for (int q = 0; q < 4; ++q) breg[q] = *(const float4*)(p.bias + nsp * N + wave * 32 + 8 * q + 4 * (lane >> 5));
  } else {
    for (int i = tid; i < N; i += 256) s_bias[i] = p.bias[nsp * N + i];
  }
  auto a_off = [](int r, int chunk) { return r * AROWB + ((chunk ^ (r & (ACH >= 32 ? 31 : 15))) << 4); };
  auto y_off = [](int r, int chunk) { return r * YROWB + ((chunk ^ (r & (YCH >= 32 ? 31 : 15))) << 4); };
  // ---- weights, once per workgroup: wave -> channel tiles wave * TPW + i
  uint4 w[TPW][KS];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const char* wb = (const char*)p.wf + ((size_t)((nsp * 4 + wave) * TPW + i) * KS * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w[i][ks] = *(const uint4*)(wb + (size_t)ks * 1024);
  }
  // ---- DMA geometry: lane-linear on the LDS side, XOR swizzle on the source chunk; rows per 1 KiB piece = 1024 / row bytes
  const u32x4 srd_a = make_srd(p.a), srd_r = make_srd(RES ? p.res : p.a);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  constexpr int A_LPR = ACH < 64 ? ACH : 64, Y_LPR = YCH < 64 ? YCH : 64;     // lanes per row within a piece
  constexpr int A_RPP = 64 / A_LPR, Y_RPP = 64 / Y_LPR;                       // rows per piece (1 KiB = 64 chunks)
  static_assert(ACH <= 64 && YCH <= 64, "rows longer than 1 KiB are not laid out here");
  const int ntiles = (p.M + PX - 1) / PX;
  auto issue_a = [&](int tile, int buf) {
    const int rows_left = p.M - tile * PX;
    const uint32_t so = (uint32_t)tile * ABYTES;
    static_for<A_PIECES>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int row = (wave * A_PIECES + J) * A_RPP + lane / A_LPR, pos = lane % A_LPR;
      const uint32_t vo = row < rows_left ? (uint32_t)(row * AROWB + ((pos ^ (row & (ACH >= 32 ? 31 : 15))) << 4)) : MCG_OOB_OFFSET;
      lds_dma16<AOFF + J * 1024>(vo, srd_a, so, lds_base + buf * ABYTES + wave * (A_PIECES * 1024));
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
      } else if (row < rows_left) {   // nearest-upsample gather: source row of output pixel m (torch: src = min(floor(dst * in / out), in - 1))
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
    issue_a(walker, 0);
  }
  __syncthreads();   // biases in LDS (the loop's first wait covers the register copies)
  int it = 0;
  for (int tile = walker; tile < ntiles; tile += nwalkers, ++it) {
    const long long m0 = (long long)tile * PX;
    char* s_y = smem + (RES ? (it & 1) * YBYTES : 0);
    const char* s_a = smem + AOFF + (NAB == 2 ? (it & 1) * ABYTES : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the tile have landed (and its earlier stores left)
    __syncthreads();                                           // everyone's pieces landed; the other buffers are free
    const bool more = tile + nwalkers < ntiles;
    if (more) {
      if (RES) issue_res(tile + nwalkers, (it + 1) & 1);
      if (NAB == 2) issue_a(tile + nwalkers, (it + 1) & 1);
    }
    // ---- contraction: wave -> TPW channel tiles x one pixel tile, K ascending
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const uint4 x = *(const uint4*)(s_a + a_off(px, 2 * ks + half));
#pragma unroll
      for (int i = 0; i < TPW; ++i) Mma<F>::run(acc[i], w[i][ks], x);
    }
    if (RES == 0) __syncthreads();                             // previous tile's y staging fully stored (single y tile)
    // ---- epilogue: (acc + bias) (+ res) -> [relu] -> bf16, 8 bytes (4 channels of one pixel) at a time, into the y tile
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = (wave * TPW + i) * 32 + 8 * q + 4 * half;
        float4 b4;
        if constexpr (BIAS_REGS) b4 = breg[q]; else b4 = *(const float4*)(s_bias + c0);
        char* slot = s_y + y_off(px, c0 >> 3) + (c0 & 7) * 2;
        float v[4] = {acc[i][4 * q] + b4.x, acc[i][4 * q + 1] + b4.y, acc[i][4 * q + 2] + b4.z, acc[i][4 * q + 3] + b4.w};
        if (RES) {
          const uint2 rr = *(const uint2*)slot;
          v[0] += H16<F>::lo(rr.x); v[1] += H16<F>::hi(rr.x);
          v[2] += H16<F>::lo(rr.y); v[3] += H16<F>::hi(rr.y);
        }
        *(uint2*)slot = p.relu ? relu_pack4<F>(v) : make_uint2(H16<F>::pack2(v[0], v[1]), H16<F>::pack2(v[2], v[3]));
      }
    }
    __syncthreads();                                           // y tile complete; the A tile has been consumed by every wave
    if (NAB == 1 && more) issue_a(tile + nwalkers, 0);        // single A tile: refill it now
    for (int idx = tid; idx < PX * YCH; idx += 256) {
      const int r = idx / YCH, c = idx - r * YCH;
      if (m0 + r < p.M) *(uint4*)((bf16_t*)p.y + (m0 + r) * NF + nsp * N + c * 8) = *(const uint4*)(s_y + y_off(r, c));
    }
  }
}

// (K, N) pairs of the R-50 trunk whose weights -- or a 128 / 256-channel slice of them -- fit the registers of one workgroup (128 KB)
static inline bool pw_single_applicable(int K, int N, int res_mode, long long M, long long res_rows) {
  const bool shape = (K == 256 && N == 256) || (K == 512 && N == 128 && res_mode == 0) || (K == 128 && N == 512) ||
                     (K == 256 && N == 1024) || (K == 512 && N == 256 && res_mode != 1);
  return shape && M >= 64 * 1024 && M * 2 * (K > N ? K : N) < MCG_DMA_MAX_BYTES && res_rows * 2 * N < MCG_DMA_MAX_BYTES;
}
template <typename F, int KS, int TPW, int RES, int NSPLIT>
static inline void launch_pw_single_t(hipStream_t s, const PwSingleParams& p) {
  constexpr int K = 16 * KS, N = 128 * TPW, AB = 32 * 2 * K, YB = 32 * 2 * N, NYB = RES ? 2 : 1;
  constexpr int BB = TPW == 1 ? 0 : N * 4;
  constexpr int NAB = (2 * AB + NYB * YB + BB <= 80 * 1024) ? 2 : 1;
  constexpr int kLds = NYB * YB + NAB * AB + BB;
  static_assert(kLds <= 80 * 1024, "two workgroups per CU");
  // per device (a process may hold engines on several GPUs): CU count, and the kernel's dynamic-LDS limit raised once
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    (void)hipFuncSetAttribute((const void*)pw_single_kernel<F, KS, TPW, RES, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int cus = cus_of[dev];
  const int ntiles = (p.M + 31) / 32, unit = 8 * NSPLIT;
  int wgs = 2 * cus / unit * unit;                              // two workgroups per CU, whole groups of NSPLIT slices x 8 XCDs
  const int need = (ntiles + 7) / 8 * unit;
  if (wgs < unit) wgs = unit;
  hipLaunchKernelGGL((pw_single_kernel<F, KS, TPW, RES, NSPLIT>), dim3(need < wgs ? need : wgs), dim3(256), kLds, s, p);
}
template <typename F>
static inline int launch_pw_single_f(hipStream_t s, const PwSingleParams& p, int K, int N, int res_mode) {
  if (K == 256 && N == 256) {
    if (res_mode == 0) launch_pw_single_t<F, 16, 2, 0, 1>(s, p);
    else if (res_mode == 1) launch_pw_single_t<F, 16, 2, 1, 1>(s, p);
    else launch_pw_single_t<F, 16, 2, 2, 1>(s, p);
  } else if (K == 256 && N == 1024) {
    if (res_mode == 0) launch_pw_single_t<F, 16, 2, 0, 4>(s, p);
    else if (res_mode == 1) launch_pw_single_t<F, 16, 2, 1, 4>(s, p);
    else launch_pw_single_t<F, 16, 2, 2, 4>(s, p);
  } else if (K == 512 && N == 128) {
    launch_pw_single_t<F, 32, 1, 0, 1>(s, p);
  } else if (K == 512 && N == 256) {
    if (res_mode == 0) launch_pw_single_t<F, 32, 1, 0, 2>(s, p);
    else launch_pw_single_t<F, 32, 1, 2, 2>(s, p);
  } else {
    if (res_mode == 0) launch_pw_single_t<F, 8, 4, 0, 1>(s, p);
    else if (res_mode == 1) launch_pw_single_t<F, 8, 4, 1, 1>(s, p);
    else launch_pw_single_t<F, 8, 4, 2, 1>(s, p);
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

// DynamicConv's `dynamic_layer` (transformer.py:1131-1134): y[M][32768] = x[M][256] . W^T + b with FEW rows (1344 tokens at 64 clips) and
// 88 MB of output.  The generic kernel spends it in prologues and epilogues (4 K-tiles per 256x128 output tile: 55 us, 1.6 TB/s of
// writes).  Here: 128 slices of 256 columns, each slice's weights resident in the registers of a few workgroups ("walkers") that
// split the token tiles between them; the tokens (688 KB) stream out of L2.  Same arithmetic as every pw_single launch: bit-identical.
static inline int launch_pw_single(hipStream_t s, const PwSingleParams& p, int K, int N, int res_mode, bool fp16 = false) {
  return fp16 ? launch_pw_single_f<f16_t>(s, p, K, N, res_mode) : launch_pw_single_f<bf16_t>(s, p, K, N, res_mode);
}
static inline bool pw_dyn_applicable(int M) { return M >= 256 && (long long)M * 512 < MCG_DMA_MAX_BYTES; }
template <typename F>
static inline int launch_pw_dyn_f(hipStream_t s, PwSingleParams p) {
  constexpr int KS = 16, TPW = 2, NSPLIT = 128, kLds = 32 * 512 + 2 * 32 * 512 + 256 * 4;   // y tile + two A tiles + biases
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    (void)hipFuncSetAttribute((const void*)pw_single_kernel<F, KS, TPW, 0, NSPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int ntiles = (p.M + 31) / 32;
  int walkers = 2 * cus_of[dev] / NSPLIT;                       // two workgroups per CU (4 walkers per slice on 256 CUs; 2, 3, 6, 8 measured slower)
  if (walkers < 1) walkers = 1;
  if (walkers > ntiles) walkers = ntiles;
  p.many_slices = 1;
  hipLaunchKernelGGL((pw_single_kernel<F, KS, TPW, 0, NSPLIT>), dim3(walkers * NSPLIT), dim3(256), kLds, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
static inline int launch_pw_dyn(hipStream_t s, const PwSingleParams& p, bool fp16 = false) {
  return fp16 ? launch_pw_dyn_f<f16_t>(s, p) : launch_pw_dyn_f<bf16_t>(s, p);
}
