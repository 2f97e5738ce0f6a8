// "Pointwise pair": a bottleneck's conv3 (1x1) + BN + residual + ReLU and the NEXT block's conv1 (1x1) + BN + ReLU as one kernel
// (resnet.py:263-302, two consecutive Bottleneck.forward calls), for the HBM-bound layer1 / layer2 of the bf16 engine:
//
//     y = relu([a1 | a2] . W3^T + b3 (+ res))          [M][C]     (written: the block's output -- next residual, C2 / C3)
//     z = relu(y . W1n^T + b1n)                        [M][C2]    (written: the next block's conv1 output)
//
// Layer-granular execution reads y back from HBM for the next conv1 (720 MB per layer1 block at 448 frames); here the y tile goes
// straight from the epilogue into the second contraction through LDS: per identity block 2.88 GB -> 2.16 GB of HBM traffic
// (profiles/r02_*: the 1x1 convs of layer1 / layer2 ran at 19 % matrix-pipe utilisation, 5.5 TB/s).
//
// One workgroup (4 waves) = 64 pixels.  Both contractions run TRANSPOSED -- MFMA A operand = weight rows (channels), B operand =
// pixels -- so that a lane's four consecutive accumulator registers are four consecutive channels of one pixel: bias, residual,
// ReLU and the bf16 conversion work on 8-byte pieces that go straight into the swizzled A-layout tile the second contraction
// (and the coalesced global store) reads; there is no f32 staging.  Weights come as MFMA-fragment-major copies (1 KiB contiguous
// per wave load, L2-resident); the residual tile is staged into the y tile's own LDS slots (read, then overwritten in place).
//
// K order, rounding points ((acc + bias) + res in f32, y rounded to bf16 before it feeds conv1) and operand bits are those of the
// two launches it replaces: BIT-IDENTICAL (tests/test_gpu_forward.py::test_pointwise_pair_fusion_is_bit_identical).
#pragma once
#include "igemm_dma.hpp"

// four f32 -> four bf16 (round to nearest even) with ReLU applied on the stored bf16 as a packed signed max (relu_chunk's form:
// the bits the contraction kernel's epilogue produces)
typedef short s16x4_hw __attribute__((ext_vector_type(4)));
template <typename F>   // bf16_t or f16_t: both are sign-magnitude, the packed signed max clears exactly the negative values
__device__ __forceinline__ uint2 relu_pack4(const float (&v)[4]) {
  const uint2 o = make_uint2(H16<F>::pack2(v[0], v[1]), H16<F>::pack2(v[2], v[3]));
  const s16x4_hw z4 = {0, 0, 0, 0};
  return __builtin_bit_cast(uint2, __builtin_elementwise_max(__builtin_bit_cast(s16x4_hw, o), z4));
}

struct PwPairParams {
  const void* a1; int K1;                          // [M][K1] bf16, rows contiguous (conv2's output)
  const void* a2; int K2, stride2, H2, W2;         // optional second source (block 0: the downsample conv's input), NHWC [frames][H2][W2][K2]
  const void* res;                                 // [M][C] bf16 or null
  const void* w3f; const float* b3;                // [C/32][(K1+K2)/16][64][8] fragment-major, [C]
  void* y;                                         // [M][C]
  const void* w1f; const float* b1;                // [C2/32][C/16][64][8] fragment-major, [C2]
  void* z;                                         // [M][C2]
  int M, C, C2, Ho, Wo;
};

// Persistent, C = 256 (layer1).  A workgroup keeps its weight fragments in REGISTERS (the first contraction's 2 channel tiles x KS1
// K-steps and the second's C2T tiles x 16 K-steps: 96 .. 192 VGPRs; fetching them per 64-pixel tile moved more bytes out of L2
// than the tile moves to and from HBM and ran at half the unfused speed -- layers with C = 512 would need 384 weight registers
// per wave and keep the layer-granular launches) and walks pixel tiles with a grid-stride loop, two workgroups per CU.  The next
// tile's operands (residual rows into the y tile's slots, the A tile(s)) travel HBM -> LDS by `buffer_load ... lds` into the other
// half of a double buffer while the current tile is contracted, stored and handed on: loads are always in flight.
template <typename F, int NSRC, int C2T>   // F: bf16_t or f16_t (number format of the 2-byte elements); NSRC: 1 = one 64-channel A source (identity block), 2 = two (block 0: conv2 output | downsample input)
__global__ __launch_bounds__(256, 2) void pw_pair_kernel(const PwPairParams p) {
  constexpr int PX = 64, YROWB = 512, AROWB = 128, KS1 = 4 * NSRC;
  // LDS.  NSRC == 1 (residual present): two y / residual tiles (the residual of the next tile lands while this one is used) + ONE A
  // tile (consumed by the first contraction early in the iteration; the next tile's A rows are fetched right after it) = 72 KiB.
  // NSRC == 2 (block 0, no residual): one y tile + two buffers of [A tile 1 | A tile 2] = 64 KiB.  + 1.5 KiB of biases.  Either way
  // two workgroups fit a CU: with ONE tile of loads in flight per CU the kernel is latency-bound (bytes in flight / memory latency
  // = 3.7 TB/s), with two it is not.
  constexpr int YBUF = NSRC == 1 ? PX * YROWB : 0;                 // stride of the y tile between the two buffers
  constexpr int ABASE = NSRC == 1 ? 2 * PX * YROWB : PX * YROWB;   // A tiles start behind the y tile(s)
  constexpr int ABUF = NSRC == 1 ? 0 : 2 * PX * AROWB;             // stride of the A tiles between the two buffers (single A tile for NSRC == 1)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, px_l = lane & 31, half = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // biases live in LDS: a global load inside the tile loop would queue behind the next tile's DMA (loads return in order) and the
  // compiler's wait for it would drain the prefetch before the epilogue could start
  constexpr int TILES_BYTES = NSRC == 1 ? 2 * PX * YROWB + PX * AROWB : PX * YROWB + 2 * 2 * PX * AROWB;
  float* s_b3 = (float*)(smem + TILES_BYTES);
  float* s_b1 = s_b3 + 256;
  for (int i = tid; i < 256; i += 256) s_b3[i] = p.b3[i];
  for (int i = tid; i < p.C2; i += 256) s_b1[i] = p.b1[i];
  auto a_off = [](int px, int chunk) { return px * AROWB + ((chunk ^ ((px >> 1) & 7)) << 4); };
  auto y_off = [](int px, int chunk) { return px * YROWB + ((chunk ^ (px & 31)) << 4); };
  // ---- weight fragments, once per workgroup
  uint4 w3[2][KS1], w1[C2T][16];
  {
    const char* wb = (const char*)p.w3f + ((size_t)(wave * 2) * KS1 * 64 + lane) * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) w3[i][ks] = *(const uint4*)(wb + ((size_t)i * KS1 + ks) * 1024);
#pragma unroll
    for (int t = 0; t < C2T; ++t) {
      const int ct2 = (wave * C2T + t) >> 1;
      const char* wb1 = (const char*)p.w1f + ((size_t)ct2 * 16 * 64 + lane) * 16;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) w1[t][ks] = *(const uint4*)(wb1 + (size_t)ks * 1024);
    }
  }
  // ---- DMA geometry.  One wave-instruction moves 1 KiB: 2 residual rows or 8 A rows; the LDS side is lane-linear, so the XOR
  // swizzle is applied to the SOURCE chunk.  Per-lane offsets are tile-invariant; the tile advances through the scalar offset.
  const u32x4 srd_res = make_srd(p.res ? p.res : p.a1), srd_a1 = make_srd(p.a1), srd_a2 = make_srd(NSRC == 2 ? p.a2 : p.a1);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  uint32_t vres[8], va[2];
#pragma unroll
  for (int j = 0; j < 8; ++j) {                                // wave's residual pieces: rows (wave * 8 + j) * 2 + {0, 1}
    const int row = (wave * 8 + j) * 2 + (lane >> 5), pos = lane & 31;
    vres[j] = row * YROWB + ((pos ^ (row & 31)) << 4);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {                                // wave's A pieces: rows (wave * 2 + j) * 8 + 0..7
    const int row = (wave * 2 + j) * 8 + (lane >> 3), pos = lane & 7;
    va[j] = row * AROWB + ((pos ^ ((row >> 1) & 7)) << 4);
  }
  const int ntiles = (p.M + PX - 1) / PX;
  const bool has_res = p.res != nullptr;
  auto issue_res = [&](int tile, uint32_t buf) {
    const int rows_left = p.M - tile * PX;                     // < 64 only on the last tile: rows beyond M read zeros (out-of-range offset)
    const uint32_t so_res = (uint32_t)tile * (PX * YROWB);
    static_for<8>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int row = (wave * 8 + J) * 2 + (lane >> 5);
      lds_dma16<J * 1024>(row < rows_left ? vres[J] : MCG_OOB_OFFSET, srd_res, so_res, lds_base + buf * YBUF + wave * 8192);
    });
  };
  auto issue_a = [&](int tile, uint32_t buf) {
    const int rows_left = p.M - tile * PX;
    const uint32_t so_a = (uint32_t)tile * (PX * AROWB);
    static_for<2>([&](auto jc) {
      constexpr int J = decltype(jc)::value;
      const int row = (wave * 2 + J) * 8 + (lane >> 3);
      const uint32_t vo = row < rows_left ? va[J] : MCG_OOB_OFFSET;
      lds_dma16<ABASE + J * 1024>(vo, srd_a1, so_a, lds_base + buf * ABUF + wave * 2048);
      if constexpr (NSRC == 2) lds_dma16<ABASE + PX * AROWB + J * 1024>(vo, srd_a2, so_a, lds_base + buf * ABUF + wave * 2048);
    });
  };
  const int zrowb = p.C2 * 2, zc = p.C2 / 8;
  int it = 0;
  if ((int)blockIdx.x < ntiles) {
    if (has_res) issue_res(blockIdx.x, 0);
    issue_a(blockIdx.x, 0);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
    const long long m0 = (long long)tile * PX;
    char* s_y = smem + (it & 1) * YBUF;
    const char* s_a = smem + ABASE + (it & 1) * ABUF;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's pieces of the tile have landed (and its earlier stores left)
    __syncthreads();                                           // everyone's pieces landed; everyone is done with the other buffer
    const bool more = tile + (int)gridDim.x < ntiles;
    if (more) {
      if (has_res) issue_res(tile + gridDim.x, (it + 1) & 1);
      if constexpr (NSRC == 2) issue_a(tile + gridDim.x, (it + 1) & 1);   // double-buffered A tiles; NSRC == 1: after the first contraction
    }
    // ---- first contraction: wave -> channel tiles {2 wave, 2 wave + 1} x both pixel tiles, K ascending (source 1, then source 2)
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      const char* src = s_a + (ks >> 2) * (PX * AROWB);
      const uint4 x0 = *(const uint4*)(src + a_off(px_l, 2 * (ks & 3) + half));
      const uint4 x1 = *(const uint4*)(src + a_off(32 + px_l, 2 * (ks & 3) + half));
      Mma<F>::run(acc[0][0], w3[0][ks], x0);
      Mma<F>::run(acc[0][1], w3[0][ks], x1);
      Mma<F>::run(acc[1][0], w3[1][ks], x0);
      Mma<F>::run(acc[1][1], w3[1][ks], x1);
    }
    // ---- epilogue 1: (acc + bias) + res -> relu -> bf16, 8 bytes (4 channels of one pixel) at a time, into the y tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = (wave * 2 + i) * 32 + 8 * q + 4 * half;
        const float4 b4 = *(const float4*)(s_b3 + c0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int px = j * 32 + px_l;
          char* slot = s_y + y_off(px, c0 >> 3) + (c0 & 7) * 2;
          float v[4] = {acc[i][j][4 * q] + b4.x, acc[i][j][4 * q + 1] + b4.y, acc[i][j][4 * q + 2] + b4.z, acc[i][j][4 * q + 3] + b4.w};
          if (has_res) {
            const uint2 rr = *(const uint2*)slot;
            v[0] += H16<F>::lo(rr.x); v[1] += H16<F>::hi(rr.x);
            v[2] += H16<F>::lo(rr.y); v[3] += H16<F>::hi(rr.y);
          }
          *(uint2*)slot = relu_pack4<F>(v);
        }
      }
    }
    __syncthreads();                                           // y tile complete; the A tile has been consumed by every wave
    if constexpr (NSRC == 1) {
      if (more) issue_a(tile + gridDim.x, 0);                  // single A tile: refill it now
    }
    // ---- y -> global (coalesced rows) and second contraction (K = the 256 channels of y)
    for (int idx = tid; idx < PX * 32; idx += 256) {
      const int px = idx >> 5, c = idx & 31;
      if (m0 + px < p.M) *(uint4*)((bf16_t*)p.y + (m0 + px) * 256 + c * 8) = *(const uint4*)(s_y + y_off(px, c));
    }
    f32x16 acc2[C2T];
#pragma unroll
    for (int t = 0; t < C2T; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
      const int pt = (wave * C2T + t) & 1;
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        const uint4 x = *(const uint4*)(s_y + y_off(pt * 32 + px_l, 2 * ks + half));
        Mma<F>::run(acc2[t], w1[t][ks], x);
      }
    }
    __syncthreads();                                           // y tile fully consumed
    // ---- epilogue 2: z = relu(acc2 + b1n) -> bf16, staged through LDS (the y tile's space) so that it leaves as whole rows
#pragma unroll
    for (int t = 0; t < C2T; ++t) {
      const int u = wave * C2T + t, ct2 = u >> 1, pt = u & 1;
      const int px = pt * 32 + px_l;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = ct2 * 32 + 8 * q + 4 * half;
        const float4 b4 = *(const float4*)(s_b1 + c0);
        const float v[4] = {acc2[t][4 * q] + b4.x, acc2[t][4 * q + 1] + b4.y, acc2[t][4 * q + 2] + b4.z, acc2[t][4 * q + 3] + b4.w};
        *(uint2*)(s_y + px * zrowb + (((c0 >> 3) ^ (px & 7)) << 4) + (c0 & 7) * 2) = relu_pack4<F>(v);
      }
    }
    __syncthreads();
    for (int idx = tid; idx < PX * zc; idx += 256) {
      const int px = idx / zc, c = idx - px * zc;
      if (m0 + px < p.M) *(uint4*)((bf16_t*)p.z + (m0 + px) * p.C2 + c * 8) = *(const uint4*)(s_y + px * zrowb + ((c ^ (px & 7)) << 4));
    }
  }
}

// Applicable: C = 256, 64-channel sources (layer1 of a ResNet-50), a second source only at stride 1 on the output grid, operands
// inside the 2 GiB window of the DMA descriptors.
static inline bool pw_pair_applicable(int K1, int K2, int stride2, int C, int C2, long long M) {
  return K1 == 64 && (K2 == 0 || (K2 == 64 && stride2 == 1)) && C == 256 && (C2 == 64 || C2 == 128) && M * 512 < MCG_DMA_MAX_BYTES;
}
template <typename F, int NSRC, int C2T>
static inline void launch_pw_pair_t(hipStream_t s, const PwPairParams& p) {
  constexpr int kLds = (NSRC == 1 ? 2 * 64 * 512 + 64 * 128 : 64 * 512 + 2 * 2 * 64 * 128) + (256 + 128) * 4;
  // per device (a process may hold engines on several GPUs): CU count, and the kernel's dynamic-LDS limit raised once
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    (void)hipFuncSetAttribute((const void*)pw_pair_kernel<F, NSRC, C2T>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  }
  const int cus = cus_of[dev];
  const int ntiles = (p.M + 63) / 64;
  const int wgs = 2 * cus;                                     // two workgroups per CU (2 x 80 KiB of LDS fit exactly)
  hipLaunchKernelGGL((pw_pair_kernel<F, NSRC, C2T>), dim3(ntiles < wgs ? ntiles : wgs), dim3(256), kLds, s, p);
}
template <typename F>
static inline int launch_pw_pair_f(hipStream_t s, const PwPairParams& p) {
  if (p.K2 == 0 && p.C2 == 64) launch_pw_pair_t<F, 1, 1>(s, p);
  else if (p.K2 == 0) launch_pw_pair_t<F, 1, 2>(s, p);
  else if (p.C2 == 64) launch_pw_pair_t<F, 2, 1>(s, p);
  else launch_pw_pair_t<F, 2, 2>(s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
static inline int launch_pw_pair(hipStream_t s, const PwPairParams& p, bool fp16 = false) {
  return fp16 ? launch_pw_pair_f<f16_t>(s, p) : launch_pw_pair_f<bf16_t>(s, p);
}
