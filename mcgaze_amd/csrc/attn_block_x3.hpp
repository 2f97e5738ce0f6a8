// Attention block of one decoder stage as ONE launch in the MCG_F16X3 arithmetic (round 6; attn_block.hpp is the bf16 engine's): the
// spatial pass and the temporal pass of gaze_stqi_head.py:148-166 -- each "in_proj -> 8-head softmax attention -> out_proj + residual ->
// LayerNorm", with the SAME weights and LayerNorm in both passes -- for one clip per workgroup.  Replaces six launches per stage
// (2 x [in_proj igemm_dma, attn_core_kernel, mlp_chain_x3 with one step]) of a few microseconds of work each.
//
// A clip's 3 T token rows (row = (frame, clue), frame-major) are contiguous and closed under both passes; with 3 T <= 32 they are one
// MFMA row tile and the whole block runs out of LDS (128 KiB, one workgroup of four waves per CU, one wave per SIMD: the 512-register file
// is the wave's -- a linear step's B fragments, high and low, are 256 of them):
//   s_x    32 rows x 1 KiB: the A operand of the next linear, SPLIT into fp16 high / low 16-byte chunks when it is written (chain_x3.hpp's
//          layout: chunk = K-step * 4 + lane half * 2 + (0 high, 1 low), XOR-swizzled by the row) -- the pass's input rows for in_proj, then
//          (in_proj done, nobody reads the rows any more: the residual comes from global memory / registers) the attention output for out_proj
//   s_qkv  [32][768] f32 (q | k | v); a head's 32 floats are stored with their eight 16-byte chunks ROTATED by the head index, so that the
//          eight (row, head) threads of a quarter-wave read eight different bank groups (unrotated, a 128-byte head stride puts all of them
//          on the same four banks); the out-projection's f32 slab aliases its first 32 KiB
// Linears are chain_x3.hpp's: wave w owns 64 output columns, K = 256 in 16 steps of lo.hi + hi.lo + hi.hi, B fragments straight from the
// fragment-major SPLIT copy of the weights (packing.py::frag_major_split; 1 KiB per wave load), the next sub-step's in flight while this
// one's results are written.  The attention core is attend_row_head -- the device function attn_core_kernel runs -- and the LayerNorm keeps
// ln_kernel's lane -> column ownership; every f32 rounding point of the six launches is kept (qkv, attention output, projection + bias,
// + residual), so the block is BIT-IDENTICAL to the sequence it replaces (tests/test_gpu_kernels.py::test_attn_block_x3_matches_unfused_bitwise).
// The pass-0 output rows stay in the registers of the lanes that normalised them (wave -> 8 rows, lane -> 4 columns): they ARE pass 1's residual.
#pragma once
#include "igemm_dma.hpp"
#include "attn_block.hpp"

namespace abx {
constexpr int D = 256, ROWS = 32, ROWB = D * 4, QKV_LD = 3 * D;
constexpr int LDS_BYTES = ROWS * ROWB + ROWS * QKV_LD * 4;   // 32 KiB + 96 KiB
__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ (row & 31)) << 4); }
// four consecutive channels c .. c + 3 of a row -> 8 bytes of high parts and 8 bytes of low parts at their place (chain_x3.hpp: park4)
__device__ __forceinline__ void park4(char* base, int row, int c, const float (&v)[4]) {
  uint32_t h0, l0, h1, l1;
  split_pair(v[0], v[1], h0, l0);
  split_pair(v[2], v[3], h1, l1);
  const int chunk = (c >> 4) * 4 + ((c >> 3) & 1) * 2, pos = ((c >> 2) & 1) * 8;
  *(uint2*)(base + swz(row, chunk) + pos) = make_uint2(h0, h1);
  *(uint2*)(base + swz(row, chunk + 1) + pos) = make_uint2(l0, l1);
}
}  // namespace abx

__global__ __launch_bounds__(256, 1) void attn_block_x3_kernel(const AttnBlockParams p) {
  using namespace abx;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const s_x = smem;
  float* const s_qkv = (float*)(smem + ROWS * ROWB);
  float* const s_t = s_qkv;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = 3 * p.T;                                   // <= 32 (checked by the launcher)
  const size_t m0 = (size_t)blockIdx.x * rows;
  const float* const X = (const float*)p.x;
  for (int idx = tid; idx < ROWS * 64; idx += 256) {          // token rows -> LDS, split (padding rows are zero; never stored)
    const int r = idx >> 6, c = (idx & 63) * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < rows) t = *(const float4*)(X + (m0 + r) * D + c);
    const float v[4] = {t.x, t.y, t.z, t.w};
    park4(s_x, r, c, v);
  }
  const int arow = lane & 31, half = lane >> 5;
  uint4 bfr[2][16][2];   // [column tile][K-step][high, low]
  auto load_b = [&](const void* W, int tile0) {               // fragments of column tiles tile0, tile0 + 1: 64 x 1 KiB wave loads
    const char* wb = (const char*)W + ((size_t)tile0 * 16 * 2 * 64 + lane) * 16;
    // opaque to loop-invariant code motion: left alone, the compiler hoists the 64-bit address of every 4 KiB group of every call site out of
    // both loops and keeps them all alive -- 263 spilled registers of nothing but addresses
    asm volatile("" : "+v"(wb));
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        bfr[j][ks][0] = *(const uint4*)(wb + ((size_t)(j * 16 + ks) * 2) * 1024);
        bfr[j][ks][1] = *(const uint4*)(wb + ((size_t)(j * 16 + ks) * 2 + 1) * 1024);
      }
  };
  auto mma2 = [&](f32x16 (&acc)[2]) {                         // chain_x3.hpp's linear: K order and the order of the three terms are the igemm kernel's
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(s_x + swz(arow, ks * 4 + half * 2)));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(s_x + swz(arow, ks * 4 + half * 2 + 1)));
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(al, __builtin_bit_cast(bf16x8, bfr[j][ks][0]), acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(ah, __builtin_bit_cast(bf16x8, bfr[j][ks][1]), acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(ah, __builtin_bit_cast(bf16x8, bfr[j][ks][0]), acc[j]);
    }
  };
  load_b(p.w_in, wave * 2);
  __syncthreads();
  float keep[8][4];                                           // pass 0's output rows of this lane: pass 1's residual
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int e = 0; e < 4; ++e) keep[a][e] = 0.f;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    // ---- q | k | v = x . Win^T + b (f32, what the unfused path stores): three sub-steps of 256 columns = 8 heads; wave -> heads 2 w, 2 w + 1
#pragma unroll 1
    for (int u = 0; u < 3; ++u) {
      f32x16 acc[2];
      mma2(acc);
      if (u < 2) load_b(p.w_in, (u + 1) * 8 + wave * 2);   // (the out-projection's fragments are NOT fetched here: 256 registers live across the
                                                           //  attention core put 310 of them into scratch; they are issued right behind it)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int head = wave * 2 + j;
        const float bb = p.b_in[u * D + head * 32 + arow];
        // element d = arow of the head: chunk (d >> 2) rotated by the head index
        float* dst = s_qkv + u * D + head * 32 + ((((arow >> 2) + head) & 7) << 2) + (arow & 3);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[mfma32_row(r, lane) * QKV_LD] = acc[j][r] + bb;
      }
    }
    __syncthreads();   // qkv complete; every wave is done reading the input rows in s_x
    // ---- attention core: thread = (query row, head) (attend_row_head); rows * 8 <= 256 pairs; its output, split, becomes the next A operand
    {
      const int L = pass == 0 ? 3 : p.T;
      const int i = tid >> 3, h = tid & 7;
      __attribute__((aligned(16))) float o[32];
      if (i < rows) {
        const int kbase = pass == 0 ? (i / 3) * 3 : i % 3, kstep = pass == 0 ? 1 : 3;
        attend_row_head<float>(s_qkv + i * QKV_LD + h * 32,
                               [&](int j) { return (const float*)(s_qkv + (kbase + j * kstep) * QKV_LD + D + h * 32); },
                               [&](int j) { return (const float*)(s_qkv + (kbase + j * kstep) * QKV_LD + 2 * D + h * 32); }, L, p.scale, o,
                               [&](const float* base, int c) { return *(const uint4*)(base + (((c + h) & 7) << 2)); });
      } else {
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float v[4] = {o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]};
        park4(s_x, i, h * 32 + 4 * c, v);
      }
    }
    load_b(p.w_out, wave * 2);
    __syncthreads();   // attention rows complete, s_qkv consumed (s_t aliases it)
    // ---- out-projection + bias (f32 slab), then + residual -> LayerNorm (mlp_chain_x3's step)
    {
      f32x16 acc[2];
      mma2(acc);
      if (pass == 0) load_b(p.w_in, wave * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wave * 64 + j * 32 + arow;
        const float bb = p.b_out[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) s_t[mfma32_row(r, lane) * D + col] = acc[j][r] + bb;
      }
    }
    __syncthreads();   // the slab is complete; every wave is done reading the attention rows in s_x
    const int c0 = lane * 4;
    const float4 g4 = *(const float4*)(p.g + c0), b4 = *(const float4*)(p.b + c0);
#pragma unroll
    for (int rr8 = 0; rr8 < 8; ++rr8) {
      const int r = wave * 8 + rr8;
      const float4 t4 = *(const float4*)(s_t + r * D + c0);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      if (r < rows) {  // residual = this pass's input row (the mmcv wrapper adds the identity, transformer.py MultiheadAttention)
        if (pass == 0) {
          const float4 rr = *(const float4*)(X + (m0 + r) * D + c0);
          v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += keep[rr8][e];
        }
      }
      const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bbv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = r < rows ? (v[e] - mean) * rstd * gg[e] + bbv[e] : 0.f;   // padding rows stay zero
      park4(s_x, r, c0, v);
#pragma unroll
      for (int e = 0; e < 4; ++e) keep[rr8][e] = v[e];
      if (pass == 1 && r < rows) *(float4*)((float*)p.y + (m0 + r) * D + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
  }
}

static inline int launch_attn_block_x3(hipStream_t s, const AttnBlockParams& p) {
  static bool attr_set[MCG_MAX_DEVICES] = {false};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!attr_set[dev]) {
    if (hipFuncSetAttribute((const void*)attn_block_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, abx::LDS_BYTES) != hipSuccess) return 1;
    attr_set[dev] = true;
  }
  hipLaunchKernelGGL(attn_block_x3_kernel, dim3(p.num_clips), dim3(256), abx::LDS_BYTES, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
