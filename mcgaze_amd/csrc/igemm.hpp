// Implicit-GEMM convolution / linear kernel for gfx950: the one contraction kernel behind every
// conv of the R-50 + FPN trunk and every nn.Linear of the decoder.
//
//   D[m][n] = sum_k A[m][k] * W[n][k]      m = output pixel (frame, y, x)   n = output channel
//                                           k = (kh, kw, cin), cin contiguous (NHWC x OHWI)
//
// One workgroup = 256 threads = 4 waves computes a BM x BN output tile from 32x32 MFMA tiles.
// A and W K-slices (BKB bytes of K per row) are staged global -> registers -> LDS with a
// two-buffer software pipeline (loads for tile k+1 are in flight while tile k feeds the MFMAs,
// one barrier per K-tile).  LDS rows are XOR-swizzled in 16-byte chunks so that both the
// ds_write_b128 staging stores and the ds_read_b128 fragment loads are bank-conflict free.
// The epilogue stages the f32 accumulators through LDS so that bias + residual (+ nearest
// upsample for the FPN top-down path) + ReLU are applied on full, coalesced 16-byte rows.
#pragma once
#include "common.hpp"

struct IgemmParams {
  const void* x;
  const void* w;
  const float* bias;
  const void* res;
  void* y;
  float* partial;  // split-K slabs [slice][group][M][Cout] f32 (splitk > 1)
  int M, Ho, Wo, H, W, Cin, KH, KW, stride, pad, Cout;
  long long xs_n, xs_h;
  int xs_w;
  int nocheck;
  long long y_row_stride, res_row_stride;
  int relu, res_mode, Hr, Wr;
  float rscale_h, rscale_w;
  long long x_g, w_g, y_g, res_g;  // per-group element offsets (blockIdx.z)
  int x_g_period;                  // > 0: the A operand of group g is that of group g % x_g_period (several weight sets over the same inputs)
  int bias_g;
  int splitk, tiles_per_slice;
  // optional second A source, K-concatenated after the first (1x1 taps only): D = [A1 | A2] . [W1 | W2]^T.
  // Used to fuse a bottleneck's conv3 with its downsample conv (res_layer.py:51-61): A2 = block input sampled at stride2.
  const void* x2;
  int Cin2, stride2;
  long long xs2_n, xs2_h;
  int xs2_w;
  int algo_k;  // algorithmic K for FLOP accounting when the packed K carries zero padding (stem); 0 = KH*KW*Cin
  float wscale;  // f16x3: the accumulators are multiplied by this power of two before bias / residual (the weights were packed pre-scaled by its
                 // inverse so that their fp16 low halves are normal numbers; mcg_conv_desc.wscale).  launch_igemm turns 0 into 1.
};

template <typename T, int BM, int BN, int BKB, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CPR = BKB / 16;
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int RPB = 256 / BKB;  // rows per 256-byte LDS bank row
  constexpr int ROWS_PER_PASS = 256 / CPR;
  constexpr int A_ITERS = BM / ROWS_PER_PASS, B_ITERS = BN / ROWS_PER_PASS;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB, STAGE = A_BYTES + B_BYTES;
  constexpr int C_LD = BN + 4;
  constexpr int LDS_MAIN = 2 * STAGE, LDS_EPI = WTM * C_LD * 4;
  constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
  static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small for the staging pattern");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int g = blockIdx.z, slice = blockIdx.y;

  const T* __restrict__ X = (const T*)p.x + (long long)(p.x_g_period > 0 ? g % p.x_g_period : g) * p.x_g;
  const T* __restrict__ Wt = (const T*)p.w + (long long)g * p.w_g;
  const T* __restrict__ X2 = (const T*)p.x2;
  const long long K = (long long)p.KH * p.KW * p.Cin + (X2 ? p.Cin2 : 0);
  const int tiles_per_tap = p.Cin / BK;
  const int KT1 = p.KH * p.KW * tiles_per_tap;
  const int KT = KT1 + (X2 ? p.Cin2 / BK : 0);
  const int kt_begin = slice * p.tiles_per_slice;
  const int kt_end = min(KT, kt_begin + p.tiles_per_slice);

  // ---- per-thread staging coordinates
  const int chunk = tid % CPR, row_in_pass = tid / CPR;
  long long a_off[A_ITERS], a_off2[A_ITERS];
  int a_hi0[A_ITERS], a_wi0[A_ITERS];
  int a_lds[A_ITERS], b_lds[B_ITERS];
  long long b_off[B_ITERS];
  bool b_ok[B_ITERS];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int it = 0; it < A_ITERS; ++it) {
    const int row = row_in_pass + it * ROWS_PER_PASS;
    const int m = m0 + row;
    a_lds[it] = row * BKB + ((chunk ^ ((row / RPB) % CPR)) << 4);
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[it] = ho * p.stride - p.pad;
      a_wi0[it] = wo * p.stride - p.pad;
      a_off[it] = (long long)n * p.xs_n + (long long)a_hi0[it] * p.xs_h + (long long)a_wi0[it] * p.xs_w + chunk * EPC;
      a_off2[it] = (long long)n * p.xs2_n + (long long)(ho * p.stride2) * p.xs2_h + (long long)(wo * p.stride2) * p.xs2_w + chunk * EPC;
    } else {
      a_hi0[it] = -(1 << 28);  // never valid
      a_wi0[it] = 0;
      a_off[it] = 0;
      a_off2[it] = 0;
    }
  }
#pragma unroll
  for (int it = 0; it < B_ITERS; ++it) {
    const int row = row_in_pass + it * ROWS_PER_PASS;
    b_lds[it] = A_BYTES + row * BKB + ((chunk ^ ((row / RPB) % CPR)) << 4);
    b_ok[it] = (n0 + row) < p.Cout;
    b_off[it] = (long long)(n0 + row) * K + chunk * EPC;
  }

  uint4 ra[A_ITERS], rb[B_ITERS];
  auto load_tile = [&](int kt) {
    if (kt < KT1) {
      const int tap = kt / tiles_per_tap, c0 = (kt - tap * tiles_per_tap) * BK;
      const int kh = tap / p.KW, kw = tap - kh * p.KW;
      const long long tap_off = (long long)kh * p.xs_h + (long long)kw * p.xs_w + c0;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int hi = a_hi0[it] + kh, wi = a_wi0[it] + kw;
        const bool ok = p.nocheck ? (a_hi0[it] > -(1 << 27)) : ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W);
        ra[it] = ok ? *(const uint4*)(X + a_off[it] + tap_off) : make_uint4(0, 0, 0, 0);
      }
    } else {  // second source: 1x1, always in range
      const long long c0 = (long long)(kt - KT1) * BK;
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it)
        ra[it] = (a_hi0[it] > -(1 << 27)) ? *(const uint4*)(X2 + a_off2[it] + c0) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it)
      rb[it] = b_ok[it] ? *(const uint4*)(Wt + b_off[it] + (long long)kt * BK) : make_uint4(0, 0, 0, 0);
  };
  auto store_tile = [&](int buf) {
    char* base = smem + buf * STAGE;
#pragma unroll
    for (int it = 0; it < A_ITERS; ++it) *(uint4*)(base + a_lds[it]) = ra[it];
#pragma unroll
    for (int it = 0; it < B_ITERS; ++it) *(uint4*)(base + b_lds[it]) = rb[it];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment read offsets (constant over the K loop)
  int fa[TM], fb[TN], ka[TM], kb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = wm * WTM + i * 32 + (lane & 31);
    fa[i] = r * BKB;
    ka[i] = (r / RPB) % CPR;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int r = wn * WTN + j * 32 + (lane & 31);
    fb[j] = A_BYTES + r * BKB;
    kb[j] = (r / RPB) % CPR;
  }

  if (kt_begin < kt_end) {
    load_tile(kt_begin);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    const int cur = (kt - kt_begin) & 1;
    const bool more = (kt + 1) < kt_end;
    if (more) load_tile(kt + 1);
    const char* base = smem + cur * STAGE;
#pragma unroll
    for (int j2 = 0; j2 < CPR / 2; ++j2) {
      const int ch = 2 * j2 + (lane >> 5);
      uint4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *(const uint4*)(base + fa[i] + ((ch ^ ka[i]) << 4));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *(const uint4*)(base + fb[j] + ((ch ^ kb[j]) << 4));
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (f32) -> coalesced rows
  float* C = (float*)smem;
#pragma unroll 1
  for (int pass = 0; pass < WAVES_M; ++pass) {
    if (wm == pass) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            C[(i * 32 + mfma32_row(r, lane)) * C_LD + wn * WTN + j * 32 + (lane & 31)] = acc[i][j][r];
    }
    __syncthreads();
    const int mbase = m0 + pass * WTM;
    if (p.splitk > 1) {
      constexpr int CPRO = BN / 4;
      float* P = p.partial + ((long long)(slice * gridDim.z + g) * p.M) * p.Cout;
      for (int idx = tid; idx < WTM * CPRO; idx += 256) {
        const int r = idx / CPRO, c = (idx - r * CPRO) * 4;
        const int m = mbase + r, n = n0 + c;
        if (m < p.M && n < p.Cout) *(float4*)(P + (long long)m * p.Cout + n) = *(const float4*)(C + r * C_LD + c);
      }
    } else {
      constexpr int CPRO = BN / EPC;
      T* Y = (T*)p.y + (long long)g * p.y_g;
      const T* R = (const T*)p.res + (long long)g * p.res_g;
      const float* Bv = p.bias ? p.bias + (long long)g * p.bias_g : nullptr;
      for (int idx = tid; idx < WTM * CPRO; idx += 256) {
        const int r = idx / CPRO, c = (idx - r * CPRO) * EPC;
        const int m = mbase + r, n = n0 + c;
        if (m >= p.M || n >= p.Cout) continue;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 t = *(const float4*)(C + r * C_LD + c + e);
          v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        if (Bv) {
#pragma unroll
          for (int e = 0; e < EPC; e += 4) {
            const float4 t = *(const float4*)(Bv + n + e);
            v[e] += t.x; v[e + 1] += t.y; v[e + 2] += t.z; v[e + 3] += t.w;
          }
        }
        if (p.res_mode != MCG_RES_NONE) {
          long long rrow = m;
          if (p.res_mode == MCG_RES_UPSAMPLE_ADD) {
            const int f = m / HoWo, rem = m - f * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
            const int sh = min((int)floorf(ho * p.rscale_h), p.Hr - 1), sw = min((int)floorf(wo * p.rscale_w), p.Wr - 1);
            rrow = ((long long)f * p.Hr + sh) * p.Wr + sw;
          }
          float rv[EPC];
          const uint4 rc = *(const uint4*)(R + rrow * p.res_row_stride + n);
          chunk_to_f32(rc, rv, (T*)nullptr);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += rv[e];
        }
        if (p.relu) {
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        *(uint4*)(Y + (long long)m * p.y_row_stride + n) = f32_to_chunk(v, (T*)nullptr);
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// Launch context: everything that selects a kernel VARIANT or observes launches.  It travels explicitly from the C-ABI entry
// point down to the launchers -- the library reads no environment variable and keeps no mutable process-global state.
// An engine owns one (mcg_engine_set_option / mcg_engine_profile_start); the stand-alone operator entry points build one from
// their `tile` / `flags` arguments (include/mcgaze_hip.h, MCG_FLAG_*).
struct ProfRec { hipEvent_t a, b; double flops; double bytes; int cfg; int shape[3]; };  // bytes: ALGORITHMIC HBM bytes of the launch (inputs + residual read once, output written once, weights once)
struct Prof { ProfRec* recs = nullptr; int cap = 0, n = 0; };
struct McgCtx {
  int tile = -1;             // forced tile id of igemm_dma_kernel for Cout > 64 (bf16), -1 = heuristic (launch_typed)
  bool staged = false;       // bf16: register-staged igemm_kernel instead of the LDS-DMA kernel (the first path; A/B and > 2 GiB fallback)
  bool c64 = true;           // layer1's conv2 through conv3x3_c64.hpp
  bool stem_fused = true;    // bf16 stem through stem_fused.hpp
  bool chain = true;         // decoder row-block chains (chain.hpp)
  bool attn_block = true;    // f16x3: both attention passes of a stage as one launch (attn_block_x3.hpp); the bf16 engine's block follows `chain`
  Prof* prof = nullptr;      // armed: every contraction launch is bracketed by an event pair
  static McgCtx from_flags(int tile, int flags) {
    McgCtx c;
    c.tile = tile > 0 ? tile : -1;
    c.staged = (flags & MCG_FLAG_STAGED_GEMM) != 0;
    c.c64 = !(flags & MCG_FLAG_NO_SPECIALISED);
    c.stem_fused = !(flags & MCG_FLAG_NO_SPECIALISED);
    c.chain = !(flags & MCG_FLAG_NO_SPECIALISED);
    c.attn_block = !(flags & MCG_FLAG_NO_SPECIALISED) && !(flags & MCG_FLAG_NO_ATTN_BLOCK);
    return c;
  }
};

// Profiling hooks (igemm.hip): bracket one launch with the context's event pair, tagged with a configuration id and its work.
ProfRec* prof_begin(const McgCtx& ctx, hipStream_t s, int cfg, int M, int N, int K, double flops, double bytes = 0.0);
void prof_end(ProfRec* rec, hipStream_t s);

// Host-side launcher (igemm.hip).  Picks the tile shape from Cout / M.
int launch_igemm(hipStream_t s, mcg_dtype dt, const IgemmParams& p, int groups, const McgCtx& ctx);
// Convenience: y[M][Cout] = x[M][K] * w[Cout][K]^T (+bias)(+res)(relu), rows lda / ldy apart.
int launch_linear(hipStream_t s, mcg_dtype dt, const void* x, long long lda, const void* w, const float* bias,
                  const void* res, long long ldres, void* y, long long ldy, int M, int K, int Cout, int relu, const McgCtx& ctx);
// Split-K linear: partial slabs [splitk][M][Cout] f32; returns the slice count through *splitk_out.
int launch_linear_splitk(hipStream_t s, mcg_dtype dt, const void* x, long long lda, const void* w, float* partial,
                         int M, int K, int Cout, int want_slices, int* splitk_out, const McgCtx& ctx);
// The C-ABI operators with an explicit context (the engine calls these; the extern "C" wrappers build the context from flags).
int conv2d_ctx(hipStream_t s, mcg_dtype dt, const mcg_conv_desc* d, const McgCtx& ctx);
int stem_forward_ctx(hipStream_t s, mcg_dtype dt, const float* img, const void* w_stem, const float* bias, void* y, int N, int H, int W,
                     void* ws, size_t ws_bytes, const McgCtx& ctx);
int stage_forward_ctx(hipStream_t s, mcg_dtype dt, const void* const W[MCG_SW_COUNT], const void* roi_feat, const void* obj_in,
                      const float* boxes_in, int N, int clip_length, void* obj_out, float* boxes_out, float* cls_out,
                      const float stds[4], void* ws, size_t ws_bytes, const McgCtx& ctx);
// cls_logits / scores_out (optional, [N][3]): the last stage's logits -> sigmoid scores, written by the gaze tail kernel
int gaze_head_ctx(hipStream_t s, mcg_dtype dt, const void* const W[MCG_GW_COUNT], const void* obj, int N, float* gaze_out,
                  void* ws, size_t ws_bytes, const McgCtx& ctx, const float* cls_logits, float* scores_out);
