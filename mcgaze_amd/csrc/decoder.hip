// Decoder-side kernels: multi-clue spatial/temporal attention core, fused LayerNorm chains,
// the DynamicConv instance-interaction core, per-clue box/score heads with delta2bbox, the
// gaze head tail, and query initialisation.  GEMM-shaped work goes through igemm.hpp.
#include "igemm.hpp"
#include "chain.hpp"
#include "attn_block.hpp"
#include "pw_single.hpp"
#include "igemm_dma.hpp"
#include "chain_x3.hpp"
#include "attn_block_x3.hpp"
#include "pw_single_x3.hpp"

#include <string.h>

// ------------------------------------------------------------------------------------------------
// Attention core (nn.MultiheadAttention inside mmcv's wrapper, gaze_stqi_head.py:151,162):
// 8 heads x 32 dims, softmax over the L tokens of one group.  One workgroup per group; thread =
// (query token, head), no cross-lane reduction (attn_block.hpp: attend_row_head).
// The bf16 engine runs both attention passes of a stage in attn_block_kernel instead whenever a clip's 3 T rows fit one MFMA tile.
//   spatial : group = frame,        tokens r = g*3 + i            (L = 3 clues)
//   temporal: group = (clip, clue), tokens r = (b*T + i)*3 + c    (L = T frames)
template <typename T>
__global__ __launch_bounds__(256) void attn_core_kernel(const T* __restrict__ qkv, T* __restrict__ out, int L, int temporal, int clip_len, float scale) {
  const int g = blockIdx.x;
  long long base; int step;
  if (temporal) { const int b = g / 3, c = g - b * 3; base = (long long)b * clip_len * 3 + c; step = 3; }
  else { base = (long long)g * 3; step = 1; }
  const int D = 256;
  for (int pair = threadIdx.x; pair < L * 8; pair += 256) {   // thread = (query token of the group, head)
    const int i = pair >> 3, h = pair & 7;
    const long long ri = base + (long long)i * step;
    attend_row_head<T>(qkv + ri * (3 * D) + h * 32,
                       [&](int j) { return qkv + (base + (long long)j * step) * (3 * D) + D + h * 32; },
                       [&](int j) { return qkv + (base + (long long)j * step) * (3 * D) + 2 * D + h * 32; }, L, scale, out + ri * D + h * 32);
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm chain over rows of width D (64 <= D <= 256, D % 64 == 0), one wave per row:
//   v = src row (T)  or  sum of split-K slabs (f32)  [+ bias]
//   if ln1: v = LN1(v) [ReLU]          if res: v += res row          if ln2: v = LN2(v) [ReLU]
// LN parameters may differ per row group (rows_per_group rows share one parameter set).
struct LnParams {
  const void* src; const float* partial; int slabs; long long slab_stride;
  const float* bias;
  const float* g1; const float* b1; int relu1;
  const void* res;
  const float* g2; const float* b2; int relu2;
  void* dst;
  int M, D; long long src_ld, res_ld, dst_ld;
  int rows_per_group; int param_stride;
};

template <typename T>
__global__ __launch_bounds__(256) void ln_kernel(const LnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + wave;
  if (row >= p.M) return;
  const int per = p.D / 64;  // elements per lane (<= 4), lane owns columns lane*per .. +per
  const int c0 = lane * per;
  const int grp = row / p.rows_per_group;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e >= per) { v[e] = 0.f; continue; }
    float t;
    if (p.partial) {
      t = 0.f;
      for (int s = 0; s < p.slabs; ++s) t += p.partial[(long long)s * p.slab_stride + (long long)row * p.D + c0 + e];
    } else {
      t = Elem<T>::ld((const T*)p.src + (long long)row * p.src_ld + c0 + e);
    }
    if (p.bias) t += p.bias[c0 + e];
    v[e] = t;
  }
  const float invD = 1.0f / (float)p.D;
  auto norm = [&](const float* g, const float* b, int relu) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < per) s += v[e];
    const float mean = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < per) { const float d = v[e] - mean; q += d * d; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * invD + 1e-5f);
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < per) {
      float t = (v[e] - mean) * rstd * g[grp * p.param_stride + c0 + e] + b[grp * p.param_stride + c0 + e];
      v[e] = relu ? fmaxf(t, 0.f) : t;
    }
  };
  if (p.g1) norm(p.g1, p.b1, p.relu1);
  if (p.res) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (e < per) v[e] += Elem<T>::ld((const T*)p.res + (long long)row * p.res_ld + c0 + e);
  }
  if (p.g2) norm(p.g2, p.b2, p.relu2);
#pragma unroll
  for (int e = 0; e < 4; ++e) if (e < per) Elem<T>::st((T*)p.dst + (long long)row * p.dst_ld + c0 + e, v[e]);
}

static int launch_ln(hipStream_t s, mcg_dtype dt, const LnParams& p) {
  MCG_CHECK_ARG(p.D % 64 == 0 && p.D <= 256 && p.M > 0, "layernorm: unsupported width %d", p.D);
  dim3 grid((p.M + 3) / 4);
  if (dt == MCG_BF16) hipLaunchKernelGGL(ln_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else if (dt == MCG_F16) hipLaunchKernelGGL(ln_kernel<f16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(ln_kernel<float>, grid, dim3(256), 0, s, p);
  MCG_CHECK_LAUNCH("layernorm");
  return MCG_OK;
}

static LnParams ln_simple(const void* src, const float* g, const float* b, int relu, void* dst, int M, int D) {
  LnParams p;
  memset(&p, 0, sizeof(p));
  p.src = src; p.g1 = g; p.b1 = b; p.relu1 = relu; p.dst = dst; p.M = M; p.D = D;
  p.src_ld = D; p.res_ld = D; p.dst_ld = D; p.rows_per_group = 1 << 30; p.param_stride = 0;
  return p;
}

// ------------------------------------------------------------------------------------------------
// DynamicConv core (transformer.py:1131-1148), one workgroup (4 waves) per token:
//   F1 = ReLU(LN64 (F[49x256]  . Win[256x64]))     Win^T  = params[r][0     .. 16384): [64][256] in MFMA-fragment-major order
//   F2 = ReLU(LN256(F1[49x64]  . Wout[64x256]))    Wout^T = params[r][16384 .. 32768): [256][64] likewise (packing.py::dyn_permutation)
// The 49 positions are padded to 64 rows (2 MFMA row tiles).  MFMA operand fragments for F and
// the generated weights are read straight from global/L2 (each is used by one token only);
// F1 goes through LDS (f32 for the LayerNorm, then dtype, chunk-swizzled, as the next A operand).
template <typename T>
__global__ __launch_bounds__(256) void dynconv_kernel(const T* __restrict__ roi, const T* __restrict__ params,
                                                      const float* __restrict__ g_in, const float* __restrict__ b_in,
                                                      const float* __restrict__ g_out, const float* __restrict__ b_out,
                                                      T* __restrict__ out) {
  constexpr int EPC = Elem<T>::kPerChunk;
  constexpr int P = 49, DI = 256, DF = 64;
  constexpr int D1_LD = DF + 4, D2_LD = DI + 4;
  constexpr int A2_ROWB = DF * (int)sizeof(T);            // bytes per F1 row as next A operand
  constexpr int A2_CPR = A2_ROWB / 16;
  constexpr int A2_RPB = (256 / A2_ROWB) > 0 ? (256 / A2_ROWB) : 1;
  constexpr int A2_BYTES = 64 * A2_ROWB;
  constexpr int D1_BYTES = 64 * D1_LD * 4;
  constexpr int D2_BYTES = P * D2_LD * 4;                  // D2 aliases D1 and A2 (barrier in between)
  constexpr int LDS_BYTES = D2_BYTES > D1_BYTES + A2_BYTES ? D2_BYTES : D1_BYTES + A2_BYTES;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  float* D1 = (float*)smem;
  float* D2 = (float*)smem;
  char* A2 = smem + D1_BYTES;

  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T* __restrict__ F = roi + (long long)r * P * DI;
  const T* __restrict__ Win = params + (long long)r * (2 * DI * DF);
  const T* __restrict__ Wout = Win + DI * DF;

  // ---- stage 1: wave -> one 32x32 tile of D1[64][64]
  {
    const int tm = wave >> 1, tn = wave & 1;
    const int prow = tm * 32 + (lane & 31);
    const bool pv = prow < P;
    const T* ap = F + (long long)prow * DI + (lane >> 5) * EPC;
    // the generated weights arrive MFMA-fragment-major (packing.py::dyn_permutation): chunk pair j of column tile tn = one contiguous KiB
    const T* bp = Win + ((long long)tn * (DI / (2 * EPC)) * 64 + lane) * EPC;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int PAIRS = DI / (2 * EPC);
    // all operand fragments of a group of K-steps are in flight together (one L2 latency per group, not one per step): the
    // kernel is latency-bound -- every operand is used by one token only and comes straight from L2
    constexpr int GRP = PAIRS < 8 ? PAIRS : 8;
#pragma unroll
    for (int j0 = 0; j0 < PAIRS; j0 += GRP) {
      uint4 a[GRP], b[GRP];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        a[j] = pv ? *(const uint4*)(ap + (j0 + j) * 2 * EPC) : make_uint4(0, 0, 0, 0);
        b[j] = *(const uint4*)(bp + (j0 + j) * 64 * EPC);
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) Mma<T>::run(acc, a[j], b[j]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) D1[(tm * 32 + mfma32_row(i, lane)) * D1_LD + tn * 32 + (lane & 31)] = acc[i];
  }
  // stage 2's weight fragments (wave -> columns [wave*64, +64) of Wout^T, K = 64): issued now, consumed after the LayerNorm below
  constexpr int PAIRS2 = DF / (2 * EPC);
  uint4 bf2[PAIRS2][2];
#pragma unroll
  for (int j = 0; j < PAIRS2; ++j)
#pragma unroll
    for (int b = 0; b < 2; ++b) bf2[j][b] = *(const uint4*)(Wout + ((long long)((wave * 2 + b) * PAIRS2 + j) * 64 + lane) * EPC);
  __syncthreads();
  // ---- LN over 64 features + ReLU, one wave per row, lane = feature; write F1 as dtype A operand
  for (int row = wave; row < 64; row += 4) {
    float v = D1[row * D1_LD + lane];
    const float mean = wave_sum(v) * (1.0f / DF);
    const float d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / DF) + 1e-5f);
    v = fmaxf(d * rstd * g_in[lane] + b_in[lane], 0.f);
    const int chunk = lane / EPC, within = lane % EPC;
    T* dst = (T*)(A2 + row * A2_ROWB + ((chunk ^ ((row / A2_RPB) % A2_CPR)) << 4)) + within;
    Elem<T>::st(dst, v);
  }
  __syncthreads();  // D1 fully consumed, A2 complete
  // ---- stage 2: wave -> columns [wave*64, +64) of D2[64][256], 2x2 tiles, K = 64
  {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    constexpr int PAIRS = DF / (2 * EPC);
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const int ch = 2 * j + (lane >> 5);
      uint4 af[2], bf[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = a * 32 + (lane & 31);
        af[a] = *(const uint4*)(A2 + row * A2_ROWB + ((ch ^ ((row / A2_RPB) % A2_CPR)) << 4));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) bf[b] = bf2[j][b];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) Mma<T>::run(acc[a][b], af[a], bf[b]);
    }
    __syncthreads();  // every wave is done reading A2 before D2 (which aliases it) is written
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = a * 32 + mfma32_row(i, lane);
          if (row < P) D2[row * D2_LD + wave * 64 + b * 32 + (lane & 31)] = acc[a][b][i];
        }
  }
  __syncthreads();
  // ---- LN over 256 channels + ReLU, one wave per position, lane owns 4 consecutive channels
  for (int row = wave; row < P; row += 4) {
    const float4 t = *(const float4*)(D2 + row * D2_LD + lane * 4);
    float v[4] = {t.x, t.y, t.z, t.w};
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / DI);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] -= mean; q += v[e] * v[e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / DI) + 1e-5f);
    T* dst = out + ((long long)r * P + row) * DI + lane * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) Elem<T>::st(dst + e, fmaxf(v[e] * rstd * g_out[lane * 4 + e] + b_out[lane * 4 + e], 0.f));
  }
}

// The same in the f16x3 arithmetic (MCG_F16X3 engine; round 5): f32 storage, both operands of both contractions split into fp16 high / low
// parts, three v_mfma_f32_32x32x16_f16 per product (lo.hi + hi.lo + hi.hi, the contraction kernels' order) instead of f32 MFMAs at 1/16 of
// the rate.  The generated weights arrive in THIS kernel's fragment order (packing.py::dyn_permutation(epc = 8): a lane's eight K elements of a
// 32x32x16 step are 32 contiguous bytes, a wave's two loads 2 KiB) and are split in registers; F1 is split ONCE when the LayerNorm writes
// it and parked in LDS as fp16 high / low planes (its four readers would each split it again).  Round 3 measured an x3 DynamicConv with
// K-contiguous parameter rows (a load touched 32 rows x 32 bytes) as slower than the f32 kernel; with whole-KiB loads it is the faster one.
__global__ __launch_bounds__(256) void dynconv_x3_kernel(const float* __restrict__ roi, const float* __restrict__ params,
                                                         const float* __restrict__ g_in, const float* __restrict__ b_in,
                                                         const float* __restrict__ g_out, const float* __restrict__ b_out,
                                                         float* __restrict__ out) {
  constexpr int P = 49, DI = 256, DF = 64;
  constexpr int D1_LD = DF + 4, D2_LD = DI + 4;
  constexpr int A2_ROWB = DF * 2;                          // bytes per F1 row of one fp16 plane
  constexpr int A2_PLANE = 64 * A2_ROWB;                   // 8 KiB
  constexpr int D1_BYTES = 64 * D1_LD * 4;
  constexpr int D2_BYTES = P * D2_LD * 4;                  // D2 aliases D1 and the planes (barrier in between)
  constexpr int LDS_BYTES = D2_BYTES > D1_BYTES + 2 * A2_PLANE ? D2_BYTES : D1_BYTES + 2 * A2_PLANE;
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];
  float* D1 = (float*)smem;
  float* D2 = (float*)smem;
  char* A2h = smem + D1_BYTES;
  char* A2l = A2h + A2_PLANE;

  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
  const float* __restrict__ F = roi + (long long)r * P * DI;
  const float* __restrict__ Win = params + (long long)r * (2 * DI * DF);
  const float* __restrict__ Wout = Win + DI * DF;

  // ---- stage 1: wave -> one 32x32 tile of D1[64][64], K = 256 = 16 steps
  {
    const int tm = wave >> 1, tn = wave & 1;
    const int prow = tm * 32 + (lane & 31);
    const bool pv = prow < P;
    const float* ap = F + (long long)prow * DI + 8 * h;
    const float* bp = Win + ((long long)tn * 16 * 64 + lane) * 8;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    constexpr int GRP = 4;                                 // K steps whose operand fragments are in flight together
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += GRP) {
      uint4 a[GRP][2], b[GRP][2];
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        a[j][0] = pv ? *(const uint4*)(ap + (s0 + j) * 16) : make_uint4(0, 0, 0, 0);
        a[j][1] = pv ? *(const uint4*)(ap + (s0 + j) * 16 + 4) : make_uint4(0, 0, 0, 0);
        b[j][0] = *(const uint4*)(bp + (s0 + j) * 512);
        b[j][1] = *(const uint4*)(bp + (s0 + j) * 512 + 4);
      }
#pragma unroll
      for (int j = 0; j < GRP; ++j) {
        bf16x8 ah, al, bh, bl;
        split_f32x8(a[j][0], a[j][1], ah, al);
        split_f32x8(b[j][0], b[j][1], bh, bl);
        acc = x3_mfma(al, bh, acc);
        acc = x3_mfma(ah, bl, acc);
        acc = x3_mfma(ah, bh, acc);
      }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) D1[(tm * 32 + mfma32_row(i, lane)) * D1_LD + tn * 32 + (lane & 31)] = acc[i];
  }
  // stage 2's weight fragments (wave -> columns [wave*64, +64) of Wout^T, K = 64 = 4 steps): issued now, consumed after the LayerNorm
  uint4 bf2[4][2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const float* q = Wout + ((long long)((wave * 2 + b) * 4 + s) * 64 + lane) * 8;
      bf2[s][b][0] = *(const uint4*)q;
      bf2[s][b][1] = *(const uint4*)(q + 4);
    }
  __syncthreads();
  // ---- LN over 64 features + ReLU, one wave per row, lane = feature; F1 split once into the fp16 planes (16-byte chunks XOR-swizzled
  // by row pair: a fragment read's 16 lanes hit 16 distinct bank groups)
  for (int row = wave; row < 64; row += 4) {
    float v = D1[row * D1_LD + lane];
    const float mean = wave_sum(v) * (1.0f / DF);
    const float d = v - mean;
    const float rstd = 1.0f / sqrtf(wave_sum(d * d) * (1.0f / DF) + 1e-5f);
    v = fmaxf(d * rstd * g_in[lane] + b_in[lane], 0.f);
    uint32_t hh, ll;
    split_pair(v, 0.f, hh, ll);                            // (the packed pair's second half is not stored)
    const int chunk = lane >> 3, within = lane & 7;
    const int off = row * A2_ROWB + ((chunk ^ ((row >> 1) & 7)) << 4) + within * 2;
    *(uint16_t*)(A2h + off) = (uint16_t)(hh & 0xffffu);
    *(uint16_t*)(A2l + off) = (uint16_t)(ll & 0xffffu);
  }
  __syncthreads();  // D1 fully consumed, the planes complete
  // ---- stage 2: wave -> columns [wave*64, +64) of D2[64][256], 2x2 tiles, K = 64
  {
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int ch = 2 * s + h;
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int row = a * 32 + (lane & 31);
        const int off = row * A2_ROWB + ((ch ^ ((row >> 1) & 7)) << 4);
        ah[a] = __builtin_bit_cast(bf16x8, *(const uint4*)(A2h + off));
        al[a] = __builtin_bit_cast(bf16x8, *(const uint4*)(A2l + off));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) split_f32x8(bf2[s][b][0], bf2[s][b][1], bh[b], bl[b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = x3_mfma(al[a], bh[b], acc[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = x3_mfma(ah[a], bl[b], acc[a][b]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = x3_mfma(ah[a], bh[b], acc[a][b]);
    }
    __syncthreads();  // every wave is done reading the planes before D2 (which aliases them) is written
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = a * 32 + mfma32_row(i, lane);
          if (row < P) D2[row * D2_LD + wave * 64 + b * 32 + (lane & 31)] = acc[a][b][i];
        }
  }
  __syncthreads();
  // ---- LN over 256 channels + ReLU, one wave per position, lane owns 4 consecutive channels
  for (int row = wave; row < P; row += 4) {
    const float4 t = *(const float4*)(D2 + row * D2_LD + lane * 4);
    float v[4] = {t.x, t.y, t.z, t.w};
    const float mean = wave_sum(v[0] + v[1] + v[2] + v[3]) * (1.0f / DI);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] -= mean; q += v[e] * v[e]; }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * (1.0f / DI) + 1e-5f);
    const float4 g = *(const float4*)(g_out + lane * 4), bb = *(const float4*)(b_out + lane * 4);
    float4 o;
    o.x = fmaxf(v[0] * rstd * g.x + bb.x, 0.f); o.y = fmaxf(v[1] * rstd * g.y + bb.y, 0.f);
    o.z = fmaxf(v[2] * rstd * g.z + bb.z, 0.f); o.w = fmaxf(v[3] * rstd * g.w + bb.w, 0.f);
    *(float4*)(out + ((long long)r * P + row) * DI + lane * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Per-clue score / box heads + delta2bbox (gaze_stqi_head.py:191-201, delta_xywh_bbox_coder.py:224-260),
// one wave per token; token r uses clue r % 3's weights.
template <typename T>
__global__ __launch_bounds__(256) void heads_kernel(const T* __restrict__ cls_feat, const T* __restrict__ reg_feat,
                                                    const float* __restrict__ wc, const float* __restrict__ bc,
                                                    const float* __restrict__ wr, const float* __restrict__ br,
                                                    const float* __restrict__ boxes_in, float* __restrict__ boxes_out,
                                                    float* __restrict__ cls_out, int R, float s0, float s1, float s2, float s3, float max_ratio) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave;
  if (r >= R) return;
  const int c = r % 3;
  float cf[4], rf[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    cf[e] = Elem<T>::ld(cls_feat + (long long)r * 256 + lane * 4 + e);
    rf[e] = Elem<T>::ld(reg_feat + (long long)r * 256 + lane * 4 + e);
  }
  float dots[5];
  {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += cf[e] * wc[c * 256 + lane * 4 + e];
    dots[0] = wave_sum(s) + bc[c];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) s += rf[e] * wr[(c * 4 + k) * 256 + lane * 4 + e];
    dots[1 + k] = wave_sum(s) + br[c * 4 + k];
  }
  if (lane == 0) {
    cls_out[r] = dots[0];
    const float x1 = boxes_in[r * 4], y1 = boxes_in[r * 4 + 1], x2 = boxes_in[r * 4 + 2], y2 = boxes_in[r * 4 + 3];
    const float dx = dots[1] * s0, dy = dots[2] * s1;
    const float dw = fminf(fmaxf(dots[3] * s2, -max_ratio), max_ratio), dh = fminf(fmaxf(dots[4] * s3, -max_ratio), max_ratio);
    const float px = (x1 + x2) * 0.5f, py = (y1 + y2) * 0.5f, pw = x2 - x1, ph = y2 - y1;
    const float gx = px + pw * dx, gy = py + ph * dy, gw = pw * expf(dw), gh = ph * expf(dh);
    boxes_out[r * 4] = gx - gw * 0.5f; boxes_out[r * 4 + 1] = gy - gh * 0.5f;
    boxes_out[r * 4 + 2] = gx + gw * 0.5f; boxes_out[r * 4 + 3] = gy + gh * 0.5f;
  }
}

// ------------------------------------------------------------------------------------------------
// Gaze head tail (gaze_head.py:172-200): feats [6][N][256] (branch = k*3 + clue, k=0 gaze, 1 confidence)
// -> per-clue gaze / confidence 3-vectors, confidence-weighted fusion, L2 normalisation (no eps).
template <typename T>
__global__ __launch_bounds__(256) void gaze_tail_kernel(const T* __restrict__ feats, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                                        const float* __restrict__ w_fuse, const float* __restrict__ b_fuse,
                                                        float* __restrict__ gaze_out, int N,
                                                        const float* __restrict__ cls_logits, float* __restrict__ scores_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 4 + wave;
  if (n >= N) return;
  // scores = sigmoid(last stage's logits) (multiclue_gaze_roi_head.py:351-352): the path's other tail, three values per frame
  if (scores_out && lane < 3) scores_out[n * 3 + lane] = 1.0f / (1.0f + expf(-cls_logits[n * 3 + lane]));
  float o[6][3];
#pragma unroll
  for (int br = 0; br < 6; ++br) {
    float f[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) f[e] = Elem<T>::ld(feats + ((long long)br * N + n) * 256 + lane * 4 + e);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) s += f[e] * w_out[(br * 3 + k) * 256 + lane * 4 + e];
      o[br][k] = wave_sum(s) + b_out[br * 3 + k];
    }
  }
  if (lane == 0) {
    float cat[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int k = 0; k < 3; ++k) cat[c * 3 + k] = o[3 + c][k] * o[c][k];  // confidence * gaze
    float fused[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float s = b_fuse[k];
#pragma unroll
      for (int j = 0; j < 9; ++j) s += w_fuse[k * 9 + j] * cat[j];
      fused[k] = s;
    }
    auto put = [&](int slot, const float* v) {
      const float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
#pragma unroll
      for (int k = 0; k < 3; ++k) gaze_out[((long long)slot * N + n) * 3 + k] = v[k] / nrm;
    };
    put(0, fused); put(1, o[0]); put(2, o[1]); put(3, o[2]);
  }
}

// ------------------------------------------------------------------------------------------------
// Query init (fixed_embedding_rpn_head.py:76-94): boxes = cxcywh->xyxy(E) * (w,h,w,h) per frame,
// object features = E_feat broadcast to every frame.  img_hw (device, [N][2]) may be null = (H, W).
template <typename T>
__global__ void init_queries_kernel(const float* __restrict__ init_boxes, const T* __restrict__ init_feats, const int* __restrict__ img_hw,
                                    int H, int W, float* __restrict__ boxes, T* __restrict__ obj, int N) {
  const long long total = (long long)N * 3 * 256;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % 256), q = (int)((i / 256) % 3);
    const int n = (int)(i / (3 * 256));
    obj[i] = init_feats[q * 256 + d];
    if (d < 4) {
      const float h = img_hw ? (float)img_hw[n * 2] : (float)H, w = img_hw ? (float)img_hw[n * 2 + 1] : (float)W;
      const float cx = init_boxes[q * 4], cy = init_boxes[q * 4 + 1], bw = init_boxes[q * 4 + 2], bh = init_boxes[q * 4 + 3];
      const float v = d == 0 ? (cx - 0.5f * bw) * w : d == 1 ? (cy - 0.5f * bh) * h : d == 2 ? (cx + 0.5f * bw) * w : (cy + 0.5f * bh) * h;
      boxes[((long long)n * 3 + q) * 4 + d] = v;
    }
  }
}

// ================================================================================================
// Host orchestration of one decoder stage and of the gaze head.
static inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }

struct StageWs {
  char *qkv, *att, *t, *x1, *x2, *x3, *params, *feat2, *h, *c1, *clsf, *r1, *r2;
  float* partial;
  size_t total;
};
static const int kFcSlices = 16;
static StageWs stage_layout(mcg_dtype dt, int N, char* base) {
  const size_t es = mcg_is16(dt) ? 2 : 4, R = (size_t)N * 3;
  StageWs w;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  w.qkv = take(R * 768 * es); w.att = take(R * 256 * es); w.t = take(R * 256 * es);
  w.x1 = take(R * 256 * es); w.x2 = take(R * 256 * es); w.x3 = take(R * 256 * es);
  w.params = take(R * 32768 * es); w.feat2 = take(R * 12544 * es);
  w.partial = (float*)take((size_t)(kFcSlices + 1) * R * 256 * 4);
  w.h = take(R * 2048 * es); w.c1 = take(R * 256 * es); w.clsf = take(R * 256 * es);
  w.r1 = take(R * 256 * es); w.r2 = take(R * 256 * es);
  w.total = off;
  return w;
}

extern "C" size_t mcg_stage_workspace_bytes(mcg_dtype dt, int num_frames) { return stage_layout(dt, num_frames, nullptr).total; }

int launch_roi_align(hipStream_t s, mcg_dtype dt, const void* const feats[4], const int feat_h[4], const int feat_w[4],
                     const int strides[4], int C, const float* boxes, int num_boxes, int boxes_per_frame, void* out,
                     int32_t* levels_out);

template <typename T>
static void launch_attn(hipStream_t s, const void* qkv, void* out, int groups, int L, int temporal, int clip_len) {
  hipLaunchKernelGGL(attn_core_kernel<T>, dim3(groups), dim3(256), 0, s, (const T*)qkv, (T*)out, L, temporal, clip_len, 1.0f / sqrtf(32.f));
}

extern "C" int mcg_stage_forward(mcg_stream s, mcg_dtype dt, const void* const W[MCG_SW_COUNT], const void* roi_feat,
                                 const void* obj_in, const float* boxes_in, int N, int clip_length, void* obj_out,
                                 float* boxes_out, float* cls_out, const float stds[4], void* ws, size_t ws_bytes, int flags) {
  return stage_forward_ctx((hipStream_t)s, dt, W, roi_feat, obj_in, boxes_in, N, clip_length, obj_out, boxes_out, cls_out, stds, ws, ws_bytes,
                           McgCtx::from_flags(0, flags));
}
int stage_forward_ctx(hipStream_t s, mcg_dtype dt, const void* const W[MCG_SW_COUNT], const void* roi_feat, const void* obj_in,
                      const float* boxes_in, int N, int clip_length, void* obj_out, float* boxes_out, float* cls_out,
                      const float stds[4], void* ws, size_t ws_bytes, const McgCtx& ctx) {
  MCG_CHECK_ARG(W && roi_feat && obj_in && boxes_in && obj_out && boxes_out && cls_out && stds && ws, "mcg_stage_forward: null pointer");
  MCG_CHECK_ARG(N > 0 && clip_length > 0 && N % clip_length == 0, "mcg_stage_forward: num_frames=%d is not a multiple of clip_length=%d", N, clip_length);
  for (int i = 0; i < MCG_SW_COUNT; ++i) MCG_CHECK_ARG(W[i], "mcg_stage_forward: weight table entry %d is null", i);
  StageWs w = stage_layout(dt, N, (char*)ws);
  if (ws_bytes < w.total) { mcg_set_error("mcg_stage_forward: workspace too small (%zu < %zu)", ws_bytes, w.total); return MCG_ERR_WORKSPACE; }
  const int R = N * 3, B = N / clip_length;
  const float* f32w[MCG_SW_COUNT];
  for (int i = 0; i < MCG_SW_COUNT; ++i) f32w[i] = (const float*)W[i];
  const bool bf = mcg_is16(dt), h16 = dt == MCG_F16;   // bf: 2-byte storage (MCG_BF16 or MCG_F16: the same launch sequence); h16: fp16 instantiations
  const size_t es = bf ? 2 : 4;

  // --- spatial then temporal self-attention with SHARED weights and LayerNorm (gaze_stqi_head.py:148-166)
  const void* xin = obj_in;
  char* xout[2] = {w.x1, w.x2};
  const bool chain_attn = bf && ctx.chain;
  const bool chain_x3 = dt == MCG_F16X3 && ctx.chain;   // f16x3: the row-block chains in the split arithmetic (chain_x3.hpp)
  // both passes as ONE launch, one clip per workgroup (bf16: attn_block.hpp; f16x3: attn_block_x3.hpp, round 6); bit-identical to the loop below
  const bool block_attn = (chain_attn || (chain_x3 && ctx.attn_block)) && attn_block_applicable(clip_length);
  if (block_attn) {
    AttnBlockParams ap;
    memset(&ap, 0, sizeof(ap));
    ap.x = obj_in; ap.y = w.x2; ap.w_in = W[MCG_SW_IN_PROJ_WF]; ap.b_in = f32w[MCG_SW_IN_PROJ_B];
    ap.w_out = W[MCG_SW_OUT_PROJ_WF]; ap.b_out = f32w[MCG_SW_OUT_PROJ_B]; ap.g = f32w[MCG_SW_ATTN_LN_G]; ap.b = f32w[MCG_SW_ATTN_LN_B];
    ap.num_clips = B; ap.T = clip_length; ap.scale = 1.0f / sqrtf(32.f);
    if (chain_x3 ? launch_attn_block_x3(s, ap) : launch_attn_block(s, ap, h16)) { mcg_set_error("attn_block launch failed"); return MCG_ERR_HIP; }
  }
  for (int pass = 0; pass < 2 && !block_attn; ++pass) {
    MCG_TRY(launch_linear(s, dt, xin, 256, W[MCG_SW_IN_PROJ_W], f32w[MCG_SW_IN_PROJ_B], nullptr, 0, w.qkv, 768, R, 256, 768, 0, ctx));
    if (h16) launch_attn<f16_t>(s, w.qkv, w.att, pass == 0 ? N : B * 3, pass == 0 ? 3 : clip_length, pass, clip_length);
    else if (bf) launch_attn<bf16_t>(s, w.qkv, w.att, pass == 0 ? N : B * 3, pass == 0 ? 3 : clip_length, pass, clip_length);
    else launch_attn<float>(s, w.qkv, w.att, pass == 0 ? N : B * 3, pass == 0 ? 3 : clip_length, pass, clip_length);
    MCG_CHECK_LAUNCH("attn_core");
    if (chain_attn || chain_x3) {  // out_proj + residual + LayerNorm as one launch
      ChainParams cp;
      memset(&cp, 0, sizeof(cp));
      cp.x = w.att; cp.M = R; cp.steps = 1;
      cp.st[0].W = W[MCG_SW_OUT_PROJ_WF]; cp.st[0].bias = f32w[MCG_SW_OUT_PROJ_B]; cp.st[0].res = xin;
      cp.st[0].g = f32w[MCG_SW_ATTN_LN_G]; cp.st[0].b = f32w[MCG_SW_ATTN_LN_B]; cp.st[0].dst = xout[pass]; cp.st[0].from_input = 1;
      if (chain_x3 ? launch_mlp_chain_x3(s, cp) : launch_mlp_chain(s, cp, h16)) { mcg_set_error("mlp_chain launch failed"); return MCG_ERR_HIP; }
    } else {
      MCG_TRY(launch_linear(s, dt, w.att, 256, W[MCG_SW_OUT_PROJ_W], f32w[MCG_SW_OUT_PROJ_B], xin, 256, w.t, 256, R, 256, 256, 0, ctx));
      MCG_TRY(launch_ln(s, dt, ln_simple(w.t, f32w[MCG_SW_ATTN_LN_G], f32w[MCG_SW_ATTN_LN_B], 0, xout[pass], R, 256)));
    }
    xin = xout[pass];
  }
  // --- DynamicConv (transformer.py:1116-1164)
  if (bf && ctx.chain && ctx.tile < 0 && !ctx.staged && pw_dyn_applicable(R)) {
    // few rows, 32768 columns: 256-column weight slices resident in registers, tokens streamed (pw_single.hpp); bit-identical to the generic kernel
    PwSingleParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.a = w.x2; pp.wf = W[MCG_SW_DYN_WF]; pp.bias = f32w[MCG_SW_DYN_B]; pp.y = w.params; pp.M = R; pp.Ho = 1; pp.Wo = 1;
    ProfRec* rec = prof_begin(ctx, s, 62, R, 32768, 256, 2.0 * R * 32768 * 256, 2.0 * ((double)R * (256 + 32768) + 32768.0 * 256));
    const int rc = launch_pw_dyn(s, pp, h16);
    prof_end(rec, s);
    if (rc) { mcg_set_error("pw_single (dynamic_layer) launch failed"); return MCG_ERR_HIP; }
  } else if (chain_x3 && ctx.tile < 0 && pw_dyn_applicable(R) && (long long)R * 1024 < MCG_DMA_MAX_BYTES) {
    // f16x3: the same with split weights, 256 slices of 128 columns (pw_single_x3.hpp); bit-identical to the x3 contraction kernel
    PwSingleParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.a = w.x2; pp.wf = W[MCG_SW_DYN_WF]; pp.bias = f32w[MCG_SW_DYN_B]; pp.y = w.params; pp.M = R; pp.Ho = 1; pp.Wo = 1;
    ProfRec* rec = prof_begin(ctx, s, 72, R, 32768, 256, 2.0 * R * 32768 * 256, 4.0 * ((double)R * (256 + 32768) + 32768.0 * 256));
    const int rc = launch_pw_dyn_x3(s, pp);
    prof_end(rec, s);
    if (rc) { mcg_set_error("pw_single_x3 (dynamic_layer) launch failed"); return MCG_ERR_HIP; }
  } else {
    MCG_TRY(launch_linear(s, dt, w.x2, 256, W[MCG_SW_DYN_W], f32w[MCG_SW_DYN_B], nullptr, 0, w.params, 32768, R, 256, 32768, 0, ctx));
  }
  if (h16) hipLaunchKernelGGL(dynconv_kernel<f16_t>, dim3(R), dim3(256), 0, s, (const f16_t*)roi_feat, (const f16_t*)w.params, f32w[MCG_SW_NORM_IN_G], f32w[MCG_SW_NORM_IN_B], f32w[MCG_SW_NORM_OUT_G], f32w[MCG_SW_NORM_OUT_B], (f16_t*)w.feat2);
  else if (bf) hipLaunchKernelGGL(dynconv_kernel<bf16_t>, dim3(R), dim3(256), 0, s, (const bf16_t*)roi_feat, (const bf16_t*)w.params, f32w[MCG_SW_NORM_IN_G], f32w[MCG_SW_NORM_IN_B], f32w[MCG_SW_NORM_OUT_G], f32w[MCG_SW_NORM_OUT_B], (bf16_t*)w.feat2);
  else if (dt == MCG_F16X3) hipLaunchKernelGGL(dynconv_x3_kernel, dim3(R), dim3(256), 0, s, (const float*)roi_feat, (const float*)w.params, f32w[MCG_SW_NORM_IN_G], f32w[MCG_SW_NORM_IN_B], f32w[MCG_SW_NORM_OUT_G], f32w[MCG_SW_NORM_OUT_B], (float*)w.feat2);
  else hipLaunchKernelGGL(dynconv_kernel<float>, dim3(R), dim3(256), 0, s, (const float*)roi_feat, (const float*)w.params, f32w[MCG_SW_NORM_IN_G], f32w[MCG_SW_NORM_IN_B], f32w[MCG_SW_NORM_OUT_G], f32w[MCG_SW_NORM_OUT_B], (float*)w.feat2);
  MCG_CHECK_LAUNCH("dynconv");
  int slabs = 1;
  MCG_TRY(launch_linear_splitk(s, dt, w.feat2, 12544, W[MCG_SW_FC_W], w.partial, R, 12544, 256, kFcSlices, &slabs, ctx));
  {
    LnParams p;
    memset(&p, 0, sizeof(p));
    p.partial = w.partial; p.slabs = slabs; p.slab_stride = (long long)R * 256; p.bias = f32w[MCG_SW_FC_B];
    p.g1 = f32w[MCG_SW_FC_LN_G]; p.b1 = f32w[MCG_SW_FC_LN_B]; p.relu1 = 1;
    p.res = w.x2; p.g2 = f32w[MCG_SW_IIC_LN_G]; p.b2 = f32w[MCG_SW_IIC_LN_B]; p.relu2 = 0;
    p.dst = w.x3; p.M = R; p.D = 256; p.src_ld = 256; p.res_ld = 256; p.dst_ld = 256; p.rows_per_group = 1 << 30;
    MCG_TRY(launch_ln(s, dt, p));
  }
  // --- FFN (mmcv FFN with add_identity, gaze_stqi_head.py:179-180)
  MCG_TRY(launch_linear(s, dt, w.x3, 256, W[MCG_SW_FFN1_W], f32w[MCG_SW_FFN1_B], nullptr, 0, w.h, 2048, R, 256, 2048, 1, ctx));
  {  // 2048 -> 256 at M = R rows is only a couple of dozen output tiles: split K so the whole chip works on it; the
     // LayerNorm kernel sums the slabs, adds bias and the residual, and normalises (deterministic, no atomics)
    int ffn_slabs = 1;
    MCG_TRY(launch_linear_splitk(s, dt, w.h, 2048, W[MCG_SW_FFN2_W], w.partial, R, 2048, 256, 8, &ffn_slabs, ctx));
    LnParams p;
    memset(&p, 0, sizeof(p));
    p.partial = w.partial; p.slabs = ffn_slabs; p.slab_stride = (long long)R * 256; p.bias = f32w[MCG_SW_FFN2_B];
    p.res = w.x3; p.g2 = f32w[MCG_SW_FFN_LN_G]; p.b2 = f32w[MCG_SW_FFN_LN_B];
    p.dst = obj_out; p.M = R; p.D = 256; p.src_ld = 256; p.res_ld = 256; p.dst_ld = 256; p.rows_per_group = 1 << 30;
    MCG_TRY(launch_ln(s, dt, p));
  }
  // --- towers (gaze_stqi_head.py:185-188)
  const void* rin = obj_out;
  const bool chain = (bf || dt == MCG_F16X3) && ctx.chain;
  if (chain) {  // cls tower + 3-layer reg tower: eight launches as one (chain.hpp: bit-identical; f16x3: chain_x3.hpp)
    ChainParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.x = obj_out; cp.M = R; cp.steps = 4;
    cp.st[0].W = W[MCG_SW_CLS_FC_WF]; cp.st[0].g = f32w[MCG_SW_CLS_LN_G]; cp.st[0].b = f32w[MCG_SW_CLS_LN_B]; cp.st[0].dst = w.clsf;
    cp.st[0].from_input = 1; cp.st[0].relu = 1;
    for (int j = 0; j < 3; ++j) {
      ChainStep& st = cp.st[1 + j];
      st.W = (const char*)W[MCG_SW_REG_FC_WF] + (size_t)j * 65536 * es;
      st.g = f32w[MCG_SW_REG_LN_G] + j * 256; st.b = f32w[MCG_SW_REG_LN_B] + j * 256;
      st.from_input = j == 0; st.relu = 1; st.dst = j == 2 ? w.r1 : nullptr;
    }
    if (bf ? launch_mlp_chain(s, cp, h16) : launch_mlp_chain_x3(s, cp)) { mcg_set_error("mlp_chain launch failed"); return MCG_ERR_HIP; }
    rin = w.r1;
  } else {
    MCG_TRY(launch_linear(s, dt, obj_out, 256, W[MCG_SW_CLS_FC_W], nullptr, nullptr, 0, w.c1, 256, R, 256, 256, 0, ctx));
    MCG_TRY(launch_ln(s, dt, ln_simple(w.c1, f32w[MCG_SW_CLS_LN_G], f32w[MCG_SW_CLS_LN_B], 1, w.clsf, R, 256)));
    char* rbuf[3] = {w.r1, w.r2, w.r1};
    for (int j = 0; j < 3; ++j) {
      MCG_TRY(launch_linear(s, dt, rin, 256, (const char*)W[MCG_SW_REG_FC_W] + (size_t)j * 65536 * es, nullptr, nullptr, 0, w.c1, 256, R, 256, 256, 0, ctx));
      MCG_TRY(launch_ln(s, dt, ln_simple(w.c1, f32w[MCG_SW_REG_LN_G] + j * 256, f32w[MCG_SW_REG_LN_B] + j * 256, 1, rbuf[j], R, 256)));
      rin = rbuf[j];
    }
  }
  const float max_ratio = 4.135166556742356f;  // |log(16/1000)|, delta_xywh_bbox_coder.py:236
  if (dt == MCG_BF16) hipLaunchKernelGGL(heads_kernel<bf16_t>, dim3((R + 3) / 4), dim3(256), 0, s, (const bf16_t*)w.clsf, (const bf16_t*)rin, f32w[MCG_SW_HEAD_CLS_W], f32w[MCG_SW_HEAD_CLS_B], f32w[MCG_SW_HEAD_REG_W], f32w[MCG_SW_HEAD_REG_B], boxes_in, boxes_out, cls_out, R, stds[0], stds[1], stds[2], stds[3], max_ratio);
  else if (dt == MCG_F16) hipLaunchKernelGGL(heads_kernel<f16_t>, dim3((R + 3) / 4), dim3(256), 0, s, (const f16_t*)w.clsf, (const f16_t*)rin, f32w[MCG_SW_HEAD_CLS_W], f32w[MCG_SW_HEAD_CLS_B], f32w[MCG_SW_HEAD_REG_W], f32w[MCG_SW_HEAD_REG_B], boxes_in, boxes_out, cls_out, R, stds[0], stds[1], stds[2], stds[3], max_ratio);
  else hipLaunchKernelGGL(heads_kernel<float>, dim3((R + 3) / 4), dim3(256), 0, s, (const float*)w.clsf, (const float*)rin, f32w[MCG_SW_HEAD_CLS_W], f32w[MCG_SW_HEAD_CLS_B], f32w[MCG_SW_HEAD_REG_W], f32w[MCG_SW_HEAD_REG_B], boxes_in, boxes_out, cls_out, R, stds[0], stds[1], stds[2], stds[3], max_ratio);
  MCG_CHECK_LAUNCH("heads");
  return MCG_OK;
}

extern "C" size_t mcg_gaze_head_workspace_bytes(mcg_dtype dt, int num_frames) {
  const size_t es = mcg_is16(dt) ? 2 : 4;
  return 3 * al256((size_t)6 * num_frames * 256 * es);
}

extern "C" int mcg_gaze_head(mcg_stream s, mcg_dtype dt, const void* const W[MCG_GW_COUNT], const void* obj, int N,
                             float* gaze_out, void* ws, size_t ws_bytes) {
  return gaze_head_ctx((hipStream_t)s, dt, W, obj, N, gaze_out, ws, ws_bytes, McgCtx(), nullptr, nullptr);
}
int gaze_head_ctx(hipStream_t s, mcg_dtype dt, const void* const W[MCG_GW_COUNT], const void* obj, int N, float* gaze_out,
                  void* ws, size_t ws_bytes, const McgCtx& ctx, const float* cls_logits, float* scores_out) {
  MCG_CHECK_ARG(W && obj && gaze_out && ws && N > 0, "mcg_gaze_head: bad argument");
  for (int i = 0; i < MCG_GW_COUNT; ++i) MCG_CHECK_ARG(W[i], "mcg_gaze_head: weight table entry %d is null", i);
  if (ws_bytes < mcg_gaze_head_workspace_bytes(dt, N)) { mcg_set_error("mcg_gaze_head: workspace too small"); return MCG_ERR_WORKSPACE; }
  const size_t es = mcg_is16(dt) ? 2 : 4;
  const size_t buf = al256((size_t)6 * N * 256 * es);
  char* raw = (char*)ws; char* h1 = raw + buf; char* h2 = h1 + buf;
  const char* fcw = (const char*)W[MCG_GW_FC_W];  // [6 branches = k*3+clue][2 layers][256][256]
  const float* lg = (const float*)W[MCG_GW_LN_G];
  const float* lb = (const float*)W[MCG_GW_LN_B];
  for (int layer = 0; layer < 2; ++layer) {
    {  // six branches (3 gaze MLPs, 3 confidence MLPs; branch = 3 k + clue) as ONE grouped launch: layer 0 reads token `clue` of
       // every frame for both k (x_g_period = 3), layer 1 its own branch's hidden rows
      IgemmParams p;
      memset(&p, 0, sizeof(p));
      p.M = N; p.Ho = 1; p.Wo = 1; p.H = 1; p.W = 1; p.Cin = 256; p.KH = 1; p.KW = 1; p.stride = 1; p.Cout = 256; p.nocheck = 1;
      p.splitk = 1; p.tiles_per_slice = 1 << 30;
      if (layer == 0) { p.x = obj; p.xs_n = 768; p.x_g = 256; p.x_g_period = 3; }
      else { p.x = h1; p.xs_n = 256; p.x_g = (long long)N * 256; }
      p.w = fcw + (size_t)layer * 65536 * es; p.w_g = 2 * 65536;
      p.y = raw; p.y_g = (long long)N * 256; p.y_row_stride = 256;
      MCG_TRY(launch_igemm(s, dt, p, 6, ctx));
    }
    LnParams q = ln_simple(raw, lg + layer * 256, lb + layer * 256, 1, layer == 0 ? h1 : h2, 6 * N, 256);
    q.rows_per_group = N; q.param_stride = 2 * 256;  // LN params are [6][2][256]
    MCG_TRY(launch_ln(s, dt, q));
  }
  if (dt == MCG_BF16) hipLaunchKernelGGL(gaze_tail_kernel<bf16_t>, dim3((N + 3) / 4), dim3(256), 0, s, (const bf16_t*)h2, (const float*)W[MCG_GW_OUT_W], (const float*)W[MCG_GW_OUT_B], (const float*)W[MCG_GW_FUSE_W], (const float*)W[MCG_GW_FUSE_B], gaze_out, N, cls_logits, scores_out);
  else if (dt == MCG_F16) hipLaunchKernelGGL(gaze_tail_kernel<f16_t>, dim3((N + 3) / 4), dim3(256), 0, s, (const f16_t*)h2, (const float*)W[MCG_GW_OUT_W], (const float*)W[MCG_GW_OUT_B], (const float*)W[MCG_GW_FUSE_W], (const float*)W[MCG_GW_FUSE_B], gaze_out, N, cls_logits, scores_out);
  else hipLaunchKernelGGL(gaze_tail_kernel<float>, dim3((N + 3) / 4), dim3(256), 0, s, (const float*)h2, (const float*)W[MCG_GW_OUT_W], (const float*)W[MCG_GW_OUT_B], (const float*)W[MCG_GW_FUSE_W], (const float*)W[MCG_GW_FUSE_B], gaze_out, N, cls_logits, scores_out);
  MCG_CHECK_LAUNCH("gaze_tail");
  return MCG_OK;
}

// exported to engine.hip
int launch_init_queries(hipStream_t s, mcg_dtype dt, const float* init_boxes, const void* init_feats, const int* img_hw, int H, int W,
                        float* boxes, void* obj, int N) {
  const long long total = (long long)N * 3 * 256;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (dt == MCG_BF16) hipLaunchKernelGGL(init_queries_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, init_boxes, (const bf16_t*)init_feats, img_hw, H, W, boxes, (bf16_t*)obj, N);
  else if (dt == MCG_F16) hipLaunchKernelGGL(init_queries_kernel<f16_t>, dim3(grid), dim3(256), 0, s, init_boxes, (const f16_t*)init_feats, img_hw, H, W, boxes, (f16_t*)obj, N);
  else hipLaunchKernelGGL(init_queries_kernel<float>, dim3(grid), dim3(256), 0, s, init_boxes, (const float*)init_feats, img_hw, H, W, boxes, (float*)obj, N);
  MCG_CHECK_LAUNCH("init_queries");
  return MCG_OK;
}
