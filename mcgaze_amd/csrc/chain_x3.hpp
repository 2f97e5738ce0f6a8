// chain.hpp's row-block MLP chain in the MCG_F16X3 arithmetic (f32 storage; every product as three fp16 MFMAs on the operands' fp16
// high / low parts, f32 accumulate): up to four steps of
//     y = [ReLU] LayerNorm( src . W^T [+ bias] [+ residual] )        W: [256][256], src = the kernel's input rows or the previous y
// in ONE launch, for the f16x3 decoder's towers (gaze_stqi_head.py:185-188: 8 launches -> 1) and attention output projection +
// residual + LayerNorm (:151-155 / :162-166: 2 -> 1).  The layer-granular path costs 16.4 us per 1344 x 256 x 256 linear and 7.5 us
// per LayerNorm launch -- launch latency, not work (profiles/r03_l_decoder_x3.md).
//
// One workgroup (4 waves, one per SIMD: the whole 512-register file is the wave's) owns 32 token rows.  The rows sit in LDS SPLIT into
// fp16 high / low 16-byte chunks ([K-step][lane half][high, low], XOR-swizzled by the row), split once when they are written; per
// step wave w computes output columns [64 w, 64 w + 64): 2 tiles x 16 K-steps x 3 MFMAs, the B fragments (high and low, 256 VGPRs
// per step) straight from global in a fragment-major split copy of W (packing.py::frag_major_split), the next step's in flight under
// the LayerNorm phase.  K order, the order of the three terms and the f32 rounding points are the igemm kernel's; LayerNorm follows
// ln_kernel's lane -> column ownership.
#pragma once
#include "igemm_dma.hpp"
#include "chain.hpp"

__global__ __launch_bounds__(256, 1) void mlp_chain_x3_kernel(const ChainParams p) {
  constexpr int D = 256, ROWS = 32, ROWB = D * 4;
  __shared__ __attribute__((aligned(16))) char s_x[ROWS * ROWB];    // kernel input rows (A operand, split, swizzled)
  __shared__ __attribute__((aligned(16))) char s_y[ROWS * ROWB];    // previous step's output (A operand, split, swizzled)
  __shared__ __attribute__((aligned(16))) float s_t[ROWS * D];      // linear output (+ bias), f32, row-major (LayerNorm input)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * ROWS;
  // chunk = K-step * 4 + lane half * 2 + (0 high, 1 low): 64 chunks of 16 bytes per row
  auto swz = [](int row, int chunk) { return row * ROWB + ((chunk ^ (row & 31)) << 4); };
  // four consecutive channels c .. c + 3 of a row -> 8 bytes of high parts and 8 bytes of low parts at their place
  auto park4 = [&](char* base, int row, int c, const float (&v)[4]) {
    uint32_t h0, l0, h1, l1;
    split_pair(v[0], v[1], h0, l0);
    split_pair(v[2], v[3], h1, l1);
    const int chunk = (c >> 4) * 4 + ((c >> 3) & 1) * 2, pos = ((c >> 2) & 1) * 8;
    *(uint2*)(base + swz(row, chunk) + pos) = make_uint2(h0, h1);
    *(uint2*)(base + swz(row, chunk + 1) + pos) = make_uint2(l0, l1);
  };
  for (int idx = tid; idx < ROWS * 64; idx += 256) {   // input rows -> LDS (rows beyond M are zero; never stored)
    const int r = idx >> 6, c = (idx & 63) * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < p.M) t = *(const float4*)((const float*)p.x + (size_t)(m0 + r) * D + c);
    const float v[4] = {t.x, t.y, t.z, t.w};
    park4(s_x, r, c, v);
  }
  __syncthreads();
  const int arow = lane & 31, half = lane >> 5;
  uint4 bfr[2][16][2];   // [column tile][K-step][high, low]
  auto load_b = [&](int si) {
    const char* wb = (const char*)p.st[si].W + ((size_t)(wave * 2) * 16 * 2 * 64 + lane) * 16;  // tile 2 wave, K-step 0, high, this lane
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 16; ++ks) {
        bfr[j][ks][0] = *(const uint4*)(wb + ((size_t)(j * 16 + ks) * 2) * 1024);
        bfr[j][ks][1] = *(const uint4*)(wb + ((size_t)(j * 16 + ks) * 2 + 1) * 1024);
      }
  };
  load_b(0);
  for (int si = 0; si < p.steps; ++si) {
    const ChainStep& st = p.st[si];
    const char* A = st.from_input ? s_x : s_y;
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const bf16x8 ah = __builtin_bit_cast(bf16x8, *(const uint4*)(A + swz(arow, ks * 4 + half * 2)));
      const bf16x8 al = __builtin_bit_cast(bf16x8, *(const uint4*)(A + swz(arow, ks * 4 + half * 2 + 1)));
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(al, __builtin_bit_cast(bf16x8, bfr[j][ks][0]), acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(ah, __builtin_bit_cast(bf16x8, bfr[j][ks][1]), acc[j]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = x3_mfma(ah, __builtin_bit_cast(bf16x8, bfr[j][ks][0]), acc[j]);
    }
    if (si + 1 < p.steps) load_b(si + 1);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = wave * 64 + j * 32 + arow;
      const float bb = st.bias ? st.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s_t[mfma32_row(r, lane) * D + col] = acc[j][r] + bb;
    }
    __syncthreads();
    // ---- [+ residual] LayerNorm [ReLU]: wave -> 8 rows, lane -> 4 consecutive columns (ln_kernel's order)
    const int c0 = lane * 4;
    const float4 g4 = *(const float4*)(st.g + c0), b4 = *(const float4*)(st.b + c0);
#pragma unroll
    for (int rr8 = 0; rr8 < 8; ++rr8) {
      const int r = wave * 8 + rr8;
      const float4 t4 = *(const float4*)(s_t + r * D + c0);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      if (st.res && m0 + r < p.M) {
        const float4 rr = *(const float4*)((const float*)st.res + (size_t)(m0 + r) * D + c0);
        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
      }
      const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / D);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / D) + 1e-5f);
      const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, bbv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t = (v[e] - mean) * rstd * gg[e] + bbv[e];
        v[e] = st.relu ? fmaxf(t, 0.f) : t;
      }
      park4(s_y, r, c0, v);
      if (st.dst && m0 + r < p.M) *(float4*)((float*)st.dst + (size_t)(m0 + r) * D + c0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
  }
}

static inline int launch_mlp_chain_x3(hipStream_t s, const ChainParams& p) {
  hipLaunchKernelGGL(mlp_chain_x3_kernel, dim3((p.M + 31) / 32), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
