// Attention block of one decoder stage as ONE launch (bf16 engine): the spatial pass and the temporal pass of
// gaze_stqi_head.py:148-166 -- each "in_proj -> 8-head softmax attention -> out_proj + residual -> LayerNorm", with the SAME weights
// and LayerNorm in both passes -- for one clip per workgroup.  Replaces six launches per stage (2 x [in_proj igemm, attn_core_kernel,
// out_proj + LN chain]) whose GPU work is a few microseconds each.
//
// A clip's 3 T token rows (row = (frame, clue), frame-major) are contiguous and closed under both passes: the spatial groups are
// the 3 clues of a frame, the temporal groups the T frames of a clue.  With 3 T <= 32 they are one MFMA row tile, so the whole
// block runs out of LDS:
//   s_a / s_b   token rows, bf16, 16-byte K-chunks XOR-swizzled by the row (A operand; ping-pong between the passes)
//   s_qkv       [32][768] bf16 row-major (q | k | v), aliased by the f32 slab of the out-projection
//   s_att       attention output rows (A operand)
// Linears follow chain.hpp: wave w owns 64 output columns (two 32x32 tiles), K = 256 in 16 MFMA steps, B fragments straight from
// a fragment-major copy of the weights (1 KiB contiguous per wave load), the next sub-step's fragments in flight while this one's
// results are written.  Rounding points and orders are those of the launches it replaces (qkv, attention output and the
// projection's output are rounded to bf16 where the unfused path stores them; K order of the igemm kernel; the attention core
// is the same device function; LayerNorm with ln_kernel's lane -> column ownership), so the fused block is BIT-IDENTICAL to the
// unfused sequence (tests/test_gpu_kernels.py::test_attn_block_matches_unfused_bitwise).
#pragma once
#include "common.hpp"

// One (query row, head) of the 8-head / 32-dim attention: scores over the L keys of the row's group, softmax, weighted sum of the
// values.  One THREAD does the whole thing -- 32-term dot products as sequential fma chains, maximum first, then exp / sum /
// accumulate -- so there is no cross-lane reduction at all (the earlier (head, dim)-per-thread form spent its time in 32-lane
// butterflies: 5 LDS round trips per key with __shfl_xor).  Shared by attn_block_kernel (operands in LDS) and attn_core_kernel
// (operands in global memory), which therefore agree bit for bit.  key(j) / val(j) return a pointer to the 32 contiguous
// elements of key / value j for this head; out receives 32 elements.
// ld(base, c) loads 16-byte chunk c of the 32 elements at base: linear by default; attn_block_x3.hpp keeps a head's chunks rotated in LDS
// (bank conflicts) and passes its own -- where a value sits changes no arithmetic.
template <typename T, typename KeyF, typename ValF, typename LdF>
__device__ __forceinline__ void attend_row_head(const T* __restrict__ qp, KeyF key, ValF val, int L, float scale, T* __restrict__ out, LdF ld) {
  constexpr int EPC = Elem<T>::kPerChunk, HD = 32;
  float q[HD];
#pragma unroll
  for (int c = 0; c < HD / EPC; ++c) {
    float t[EPC];
    chunk_to_f32(ld(qp, c), t, (T*)nullptr);
#pragma unroll
    for (int e = 0; e < EPC; ++e) q[c * EPC + e] = t[e] * scale;
  }
  auto score = [&](int j) {
    const T* kp = key(j);
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD / EPC; ++c) {
      float t[EPC];
      chunk_to_f32(ld(kp, c), t, (T*)nullptr);
#pragma unroll
      for (int e = 0; e < EPC; ++e) s = fmaf(q[c * EPC + e], t[e], s);
    }
    return s;
  };
  float m = -INFINITY;
  for (int j = 0; j < L; ++j) m = fmaxf(m, score(j));
  float l = 0.f, o[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) o[d] = 0.f;
  for (int j = 0; j < L; ++j) {
    const float pj = expf(score(j) - m);   // recomputed: the same bits as in the first sweep, and no L-sized register array
    l += pj;
    const T* vp = val(j);
#pragma unroll
    for (int c = 0; c < HD / EPC; ++c) {
      float t[EPC];
      chunk_to_f32(ld(vp, c), t, (T*)nullptr);
#pragma unroll
      for (int e = 0; e < EPC; ++e) o[c * EPC + e] = fmaf(pj, t[e], o[c * EPC + e]);
    }
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int c = 0; c < HD / EPC; ++c) {
    float t[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) t[e] = o[c * EPC + e] * inv;
    *(uint4*)(out + c * EPC) = f32_to_chunk(t, (T*)nullptr);
  }
}

template <typename T, typename KeyF, typename ValF>
__device__ __forceinline__ void attend_row_head(const T* __restrict__ qp, KeyF key, ValF val, int L, float scale, T* __restrict__ out) {
  attend_row_head<T>(qp, key, val, L, scale, out, [](const T* base, int c) { return *(const uint4*)(base + c * Elem<T>::kPerChunk); });
}

struct AttnBlockParams {
  const void* x;        // [num_clips * 3T][256] bf16 token rows (attn_block_x3_kernel: f32)
  void* y;              // [num_clips * 3T][256] bf16: rows after both passes (attn_block_x3_kernel: f32)
  const void* w_in;     // in_proj weight [768][256], MFMA-fragment-major (24 column tiles; attn_block_x3_kernel: the SPLIT fragment-major form)
  const float* b_in;    // [768]
  const void* w_out;    // out_proj weight [256][256], fragment-major
  const float* b_out;   // [256]
  const float* g;       // attention_norm gamma / beta [256]
  const float* b;
  int num_clips, T;
  float scale;          // 1 / sqrt(head_dim)
};

template <typename T>   // bf16_t or f16_t
__global__ __launch_bounds__(256, 1) void attn_block_kernel(const AttnBlockParams p) {
  constexpr int D = 256, ROWS = 32, ROWB = D * 2, QKV_LD = 3 * D;
  __shared__ __attribute__((aligned(16))) char s_a[ROWS * ROWB];
  __shared__ __attribute__((aligned(16))) char s_b[ROWS * ROWB];
  __shared__ __attribute__((aligned(16))) char s_att[ROWS * ROWB];
  __shared__ __attribute__((aligned(16))) char s_big[ROWS * QKV_LD * 2];   // qkv (bf16) | out-projection slab (f32, 32 KiB of the 48)
  T* s_qkv = (T*)s_big;
  float* s_t = (float*)s_big;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rows = 3 * p.T;                                   // <= 32 (checked by the launcher)
  const size_t m0 = (size_t)blockIdx.x * rows;
  auto swz = [](int row, int chunk) { return row * ROWB + ((chunk ^ (row & 31)) << 4); };
  for (int idx = tid; idx < ROWS * 32; idx += 256) {          // token rows -> LDS (padding rows are zero; never stored)
    const int r = idx >> 5, c = idx & 31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < rows) v = *(const uint4*)((const char*)p.x + (m0 + r) * ROWB + c * 16);
    *(uint4*)(s_a + swz(r, c)) = v;
  }
  const int arow = lane & 31, half = lane >> 5;
  uint4 bfr[2][16];
  auto load_b = [&](const void* W, int tile0) {               // fragments of column tiles tile0, tile0 + 1: 32 x 1 KiB wave loads
    const char* wb = (const char*)W + ((size_t)tile0 * 16 * 64 + lane) * 16;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      bfr[0][ks] = *(const uint4*)(wb + ks * 1024);
      bfr[1][ks] = *(const uint4*)(wb + 16 * 1024 + ks * 1024);
    }
  };
  auto mma2 = [&](const char* A, f32x16 (&acc)[2]) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const uint4 a = *(const uint4*)(A + swz(arow, 2 * ks + half));
      Mma<T>::run(acc[0], a, bfr[0][ks]);
      Mma<T>::run(acc[1], a, bfr[1][ks]);
    }
  };
  load_b(p.w_in, wave * 2);
  __syncthreads();
  char* xin = s_a;
  char* xout = s_b;
  for (int pass = 0; pass < 2; ++pass) {
    // ---- q | k | v = x . Win^T + b, rounded to bf16 (the unfused path stores qkv): three sub-steps of 256 columns
#pragma unroll 1
    for (int u = 0; u < 3; ++u) {
      f32x16 acc[2];
      mma2(xin, acc);
      if (u < 2) load_b(p.w_in, (u + 1) * 8 + wave * 2);
      else load_b(p.w_out, wave * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = u * D + wave * 64 + j * 32 + arow;
        const float bb = p.b_in[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) Elem<T>::st(s_qkv + mfma32_row(r, lane) * QKV_LD + col, acc[j][r] + bb);
      }
    }
    __syncthreads();
    // ---- attention core: thread = (query row, head) (attend_row_head); rows * 8 <= 256 pairs
    {
      const int L = pass == 0 ? 3 : p.T;
      const int i = tid >> 3, h = tid & 7;
      if (i < rows) {
        const int kbase = pass == 0 ? (i / 3) * 3 : i % 3, kstep = pass == 0 ? 1 : 3;
        __attribute__((aligned(16))) T o[32];
        attend_row_head<T>(s_qkv + i * QKV_LD + h * 32,
                           [&](int j) { return (const T*)(s_qkv + (kbase + j * kstep) * QKV_LD + D + h * 32); },
                           [&](int j) { return (const T*)(s_qkv + (kbase + j * kstep) * QKV_LD + 2 * D + h * 32); }, L, p.scale, o);
#pragma unroll
        for (int c = 0; c < 4; ++c) *(uint4*)(s_att + swz(i, h * 4 + c)) = *(const uint4*)(o + c * 8);
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) *(uint4*)(s_att + swz(i, h * 4 + c)) = make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();   // s_att complete, s_qkv consumed (s_t aliases it)
    // ---- out-projection + bias (f32 slab), then + residual -> bf16 -> LayerNorm (chain.hpp's step)
    {
      f32x16 acc[2];
      mma2(s_att, acc);
      if (pass == 0) load_b(p.w_in, wave * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wave * 64 + j * 32 + arow;
        const float bb = p.b_out[col];
#pragma unroll
        for (int r = 0; r < 16; ++r) s_t[mfma32_row(r, lane) * D + col] = acc[j][r] + bb;
      }
    }
    __syncthreads();
    const int c0 = lane * 4;
    const float4 g4 = *(const float4*)(p.g + c0), b4 = *(const float4*)(p.b + c0);
#pragma unroll
    for (int rr8 = 0; rr8 < 8; ++rr8) {
      const int r = wave * 8 + rr8;
      const float4 t4 = *(const float4*)(s_t + r * D + c0);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      {  // residual = this pass's input row (the mmcv wrapper adds the identity, transformer.py MultiheadAttention)
        const uint2 rr = *(const uint2*)(xin + swz(r, c0 >> 3) + (c0 & 7) * 2);
        v[0] += H16<T>::lo(rr.x); v[1] += H16<T>::hi(rr.x);
        v[2] += H16<T>::lo(rr.y); v[3] += H16<T>::hi(rr.y);
      }
      {  // the unfused path stores the projection's output as bf16 before the LayerNorm kernel reads it
        const uint32_t lo = H16<T>::pack2(v[0], v[1]), hi = H16<T>::pack2(v[2], v[3]);
        v[0] = H16<T>::lo(lo); v[1] = H16<T>::hi(lo);
        v[2] = H16<T>::lo(hi); v[3] = H16<T>::hi(hi);
      }
      const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bbv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (v[e] - mean) * rstd * gg[e] + bbv[e];
      uint2 o = make_uint2(H16<T>::pack2(v[0], v[1]), H16<T>::pack2(v[2], v[3]));
      if (r >= rows) o = make_uint2(0, 0);   // padding rows stay zero
      *(uint2*)(xout + swz(r, c0 >> 3) + (c0 & 7) * 2) = o;
      if (pass == 1 && r < rows) *(uint2*)((char*)p.y + (m0 + r) * ROWB + c0 * 2) = o;
    }
    __syncthreads();
    char* t = xin; xin = xout; xout = t;
  }
}

static inline bool attn_block_applicable(int T) { return T >= 1 && 3 * T <= 32; }
static inline int launch_attn_block(hipStream_t s, const AttnBlockParams& p, bool fp16 = false) {
  if (fp16) hipLaunchKernelGGL(attn_block_kernel<f16_t>, dim3(p.num_clips), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(attn_block_kernel<bf16_t>, dim3(p.num_clips), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
