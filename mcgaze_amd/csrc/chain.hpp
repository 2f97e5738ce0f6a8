// Row-block MLP chain for the decoder's 256-wide token features (bf16 engine): up to four steps of
//     y = [ReLU] LayerNorm( src . W^T [+ bias] [+ residual] )        W: [256][256], src = the kernel's input rows or the previous y
// in ONE launch.  Replaces runs of (igemm linear -> ln_kernel) launch pairs on M = 3 x frames rows -- each pair is two launches of
// ~8 us for ~1 us of work -- e.g. the classification / regression towers of gaze_stqi_head.py:185-188 (8 launches -> 1) and the
// attention output projection + residual + LayerNorm of :151-155 / :162-166 (2 -> 1).
//
// One workgroup (4 waves) owns 32 token rows.  The rows sit in LDS as bf16 with their 16-byte K-chunks XOR-swizzled by the row index
// (a 512-byte row would otherwise put all 32 rows of an MFMA A fragment on one bank group).  Per step wave w computes output columns
// [64 w, 64 w + 64): 2 x 16 v_mfma_f32_32x32x16_bf16, B fragments straight from global in a fragment-major copy of W (1 KiB contiguous per wave load);
// results (+ bias) go to an f32 LDS slab; each wave then takes 8 rows, adds the residual, rounds to bf16 exactly where the unfused
// linear stores its output, and normalises with the reduction order of ln_kernel (lane owns 4 consecutive columns, wave_sum) -- so the chain is BIT-IDENTICAL to the
// launch sequence it replaces (tests/test_gpu_kernels.py::test_mlp_chain_matches_unfused_bitwise).
#pragma once
#include "common.hpp"

struct ChainStep {
  const void* W;        // 256x256 bf16, MFMA-fragment-major: [column tile 8][K-step 16][lane 64][8] (include/mcgaze_hip.h MCG_SW_*_WF)
  const float* bias;    // [256] or null
  const void* res;      // residual rows [M][256] bf16 added before the LayerNorm, or null
  const float* g;       // LayerNorm gamma / beta [256]
  const float* b;
  void* dst;            // [M][256] bf16 or null (intermediate only)
  int from_input;       // 1: source rows are the kernel's input, 0: the previous step's output
  int relu;
};
struct ChainParams {
  const void* x;        // [M][256] bf16
  int M, steps;
  ChainStep st[4];
};

template <typename T>   // bf16_t or f16_t (MCG_F16: the same chain in fp16)
__global__ __launch_bounds__(256, 1) void mlp_chain_kernel(const ChainParams p) {
  constexpr int D = 256, ROWS = 32, ROWB = D * 2;
  __shared__ __attribute__((aligned(16))) char s_x[ROWS * ROWB];    // kernel input rows (A operand, swizzled)
  __shared__ __attribute__((aligned(16))) char s_y[ROWS * ROWB];    // previous step's output (A operand, swizzled)
  __shared__ __attribute__((aligned(16))) float s_t[ROWS * D];      // linear output (+ bias), f32, row-major (LayerNorm input)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.x * ROWS;
  auto swz = [](int row, int chunk) { return row * ROWB + ((chunk ^ (row & 31)) << 4); };
  // input rows -> LDS (rows beyond M are zero; never stored)
  for (int idx = tid; idx < ROWS * 32; idx += 256) {
    const int r = idx >> 5, c = idx & 31;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m0 + r < p.M) v = *(const uint4*)((const char*)p.x + (size_t)(m0 + r) * ROWB + c * 16);
    *(uint4*)(s_x + swz(r, c)) = v;
  }
  __syncthreads();
  const int arow = lane & 31, half = lane >> 5;
  // B fragments of a whole step live in registers (2 column tiles x 16 K-steps x 16 bytes = 128 VGPRs; occupancy is irrelevant at
  // 42 workgroups): all 32 loads of a step are in flight together, and the NEXT step's are issued as soon as this step's MFMAs
  // have consumed theirs, so their L2 latency hides under the LayerNorm phase
  uint4 bfr[2][16];
  auto load_b = [&](int si) {
    const char* wb = (const char*)p.st[si].W + ((size_t)(wave * 2) * 16 * 64 + lane) * 16;  // tile 2 wave, K-step 0, this lane
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      bfr[0][ks] = *(const uint4*)(wb + ks * 1024);
      bfr[1][ks] = *(const uint4*)(wb + 16 * 1024 + ks * 1024);
    }
  };
  load_b(0);
  for (int si = 0; si < p.steps; ++si) {
    const ChainStep& st = p.st[si];
    const char* A = st.from_input ? s_x : s_y;
    // ---- linear: 2 column tiles per wave, K = 256 in 16 steps (same order as the igemm kernel)
    f32x16 acc[2];
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
    for (int rr8 = 0; rr8 < 8; ++rr8) {  // unrolled: the 8 rows' residual loads and reductions overlap
      const int r = wave * 8 + rr8;
      const float4 t4 = *(const float4*)(s_t + r * D + c0);
      float v[4] = {t4.x, t4.y, t4.z, t4.w};
      if (st.res && m0 + r < p.M) {  // (acc + bias) + residual in f32, as the igemm epilogue does
        const uint2 rr = *(const uint2*)((const char*)st.res + (size_t)(m0 + r) * ROWB + c0 * 2);
        v[0] += H16<T>::lo(rr.x); v[1] += H16<T>::hi(rr.x);
        v[2] += H16<T>::lo(rr.y); v[3] += H16<T>::hi(rr.y);
      }
      {  // the unfused path stores the linear's output as bf16 and the LayerNorm kernel reads that back: same rounding here
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
      for (int e = 0; e < 4; ++e) {
        const float t = (v[e] - mean) * rstd * gg[e] + bbv[e];
        v[e] = st.relu ? fmaxf(t, 0.f) : t;
      }
      const uint2 o = make_uint2(H16<T>::pack2(v[0], v[1]), H16<T>::pack2(v[2], v[3]));
      *(uint2*)(s_y + swz(r, c0 >> 3) + (c0 & 7) * 2) = o;
      if (st.dst && m0 + r < p.M) *(uint2*)((char*)st.dst + (size_t)(m0 + r) * ROWB + c0 * 2) = o;
    }
    __syncthreads();
  }
}

static inline int launch_mlp_chain(hipStream_t s, const ChainParams& p, bool fp16 = false) {
  if (fp16) hipLaunchKernelGGL(mlp_chain_kernel<f16_t>, dim3((p.M + 31) / 32), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(mlp_chain_kernel<bf16_t>, dim3((p.M + 31) / 32), dim3(256), 0, s, p);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
