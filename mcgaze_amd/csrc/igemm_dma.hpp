// Implicit-GEMM kernel, LDS-DMA pipelined variant (the throughput path).
//
// Same contraction, data layouts and LDS swizzle as igemm.hpp, but the A / W K-slices travel
// HBM -> LDS directly (`global_load_lds_dwordx4`, 1 KiB per wave-instruction, no VGPR round trip,
// no ds_write), through a ring of STAGES LDS buffers with STAGES-1 K-tiles in flight:
//
//   iteration kt:  s_waitcnt vmcnt(PIECES_PER_WAVE*(STAGES-2))   this wave's pieces of tile kt landed
//                  s_barrier                                      everyone's pieces landed AND everyone
//                                                                 finished reading tile kt-1's buffer
//                  issue DMA of tile kt+STAGES-1 into the buffer tile kt-1 just released
//                  ds_read_b128 fragments of tile kt + MFMAs
//
// one barrier per K-tile, never vmcnt(0) inside the loop.  The DMA destination is lane-linear
// (LDS base + lane*16), so the bank-conflict-free XOR swizzle is applied on the per-lane SOURCE
// address: lane l of a piece fetches (row r0 + l/CPR, chunk (l%CPR) ^ key(row)).  Out-of-image
// taps / rows beyond M or Cout / tiles beyond K read a zero page instead of being predicated off,
// so every piece always writes its full 1 KiB and the vmcnt arithmetic stays uniform.  The DMA and
// its waits are inline asm (hipcc would otherwise drain vmcnt(0) before every ds_read).
//
// Epilogue: the residual rows this thread will need are fetched into registers BEFORE the K loop
// (their HBM latency hides under the whole contraction); accumulators are staged through LDS as
// f32 (unpadded [rows][BN]: ds_write_b32 by 32-lane row segments and ds_read_b128 along rows are
// both conflict-free) in as few passes as the LDS ring's footprint allows, then bias + residual
// (+ nearest-upsampled FPN top-down term) + ReLU are applied on coalesced 16-byte row chunks.
#pragma once
#include "igemm.hpp"

__device__ uint4 g_mcg_zero_page[4];  // zero-initialised; invalid lanes fetch from here

__device__ __forceinline__ void lds_dma16(const void* gsrc, uint32_t lds_dst_uniform) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst_uniform)
      : "memory");
}

template <typename T, int BM, int BN, int BKB, int WAVES_M, int WAVES_N, int STAGES>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2) void igemm_dma_kernel(const IgemmParams p) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int CPR = BKB / 16;
  constexpr int BK = BKB / (int)sizeof(T);
  constexpr int RPB = 256 / BKB;             // rows per 256-byte LDS bank row
  constexpr int RPP = 1024 / BKB;            // rows per DMA piece (one wave-instruction = 1 KiB)
  constexpr int A_PIECES = BM / RPP / NW, B_PIECES = BN / RPP / NW;
  constexpr int PIECES_PER_WAVE = A_PIECES + B_PIECES;
  constexpr int WTM = BM / WAVES_M, WTN = BN / WAVES_N, TM = WTM / 32, TN = WTN / 32;
  constexpr int A_BYTES = BM * BKB, B_BYTES = BN * BKB, STAGE = A_BYTES + B_BYTES;
  constexpr int LDS_BYTES = STAGES * STAGE;
  // epilogue passes: as many wave-rows (WTM output rows each) per pass as fit the ring's footprint
  constexpr int WR_FIT = LDS_BYTES / (WTM * BN * 4);
  constexpr int WR_PER_PASS = WR_FIT >= WAVES_M ? WAVES_M : (WR_FIT >= 1 ? WR_FIT : 1);
  constexpr int PASSES = (WAVES_M + WR_PER_PASS - 1) / WR_PER_PASS;
  constexpr int PASS_ROWS = WR_PER_PASS * WTM;
  constexpr int CPRO = BN / EPC;                         // output chunks per row
  constexpr int CH_PER_THREAD = PASS_ROWS * CPRO / NT;   // output chunks per thread per pass
  constexpr bool EARLY_RES = TM * TN * 16 + CH_PER_THREAD * 4 <= 144;  // register budget: prefetch residual before the K loop
  static_assert(A_PIECES >= 1 && B_PIECES >= 1 && A_PIECES * RPP * NW == BM && B_PIECES * RPP * NW == BN, "tile / wave count mismatch");
  static_assert(STAGES >= 3, "ring needs >= 3 stages");
  static_assert(PIECES_PER_WAVE * (STAGES - 2) <= 63, "vmcnt field");
  static_assert(WTM * BN * 4 <= LDS_BYTES, "epilogue staging does not fit");
  static_assert(PASS_ROWS * CPRO % NT == 0, "epilogue chunk split");
  __shared__ __attribute__((aligned(16))) char smem[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int tiles_n = (p.Cout + BN - 1) / BN;
  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (logical / tiles_n) * BM, n0 = (logical % tiles_n) * BN;
  const int g = blockIdx.z, slice = blockIdx.y;

  const T* __restrict__ X = (const T*)p.x + (long long)g * p.x_g;
  const T* __restrict__ Wt = (const T*)p.w + (long long)g * p.w_g;
  const long long K = (long long)p.KH * p.KW * p.Cin;
  const int tiles_per_tap = p.Cin / BK;
  const int KT = p.KH * p.KW * tiles_per_tap;
  const int kt_begin = slice * p.tiles_per_slice;
  const int kt_end = min(KT, kt_begin + p.tiles_per_slice);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const char* zero_page = (const char*)g_mcg_zero_page;
  const int HoWo = p.Ho * p.Wo;

  // ---- residual prefetch for epilogue pass 0 (latency hidden under the K loop)
  const T* __restrict__ R = (const T*)p.res + (long long)g * p.res_g;
  const bool has_res = p.res_mode != MCG_RES_NONE && p.splitk <= 1;
  uint4 rpre[CH_PER_THREAD];
  auto fetch_residual = [&](int pass) {
#pragma unroll
    for (int q = 0; q < CH_PER_THREAD; ++q) {
      const int idx = tid + q * NT;
      const int r = idx / CPRO, c = (idx - r * CPRO) * EPC;
      const int m = m0 + pass * PASS_ROWS + r, n = n0 + c;
      rpre[q] = make_uint4(0, 0, 0, 0);
      if (has_res && m < p.M && n < p.Cout) {
        long long rrow = m;
        if (p.res_mode == MCG_RES_UPSAMPLE_ADD) {
          const int f = m / HoWo, rem = m - f * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
          const int sh = min((int)floorf(ho * p.rscale_h), p.Hr - 1), sw = min((int)floorf(wo * p.rscale_w), p.Wr - 1);
          rrow = ((long long)f * p.Hr + sh) * p.Wr + sw;
        }
        rpre[q] = *(const uint4*)(R + rrow * p.res_row_stride + n);
      }
    }
  };
  if (EARLY_RES) fetch_residual(0);

  // ---- per-lane DMA source coordinates: piece i of this wave covers rows (wave*PIECES + i)*RPP .. +RPP-1
  const int drow = lane / CPR, dcs = lane % CPR;
  long long a_off[A_PIECES], b_off[B_PIECES];
  int a_hi0[A_PIECES], a_wi0[A_PIECES];
  bool b_ok[B_PIECES];
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int row = (wave * A_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int m = m0 + row;
    if (m < p.M) {
      const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
      a_hi0[i] = ho * p.stride - p.pad;
      a_wi0[i] = wo * p.stride - p.pad;
      a_off[i] = (long long)n * p.xs_n + (long long)a_hi0[i] * p.xs_h + (long long)a_wi0[i] * p.xs_w + chunk * EPC;
    } else {
      a_hi0[i] = -(1 << 28);
      a_wi0[i] = 0;
      a_off[i] = 0;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int row = (wave * B_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    b_ok[i] = (n0 + row) < p.Cout;
    b_off[i] = (long long)(n0 + row) * K + chunk * EPC;
  }

  // running (kh, kw, cin-tile) of the NEXT K-tile to issue: advanced incrementally, no division in the loop
  int nk_t = kt_begin;
  int nk_tap = kt_begin / tiles_per_tap;
  int nk_c = kt_begin - nk_tap * tiles_per_tap;
  int nk_kh = nk_tap / p.KW, nk_kw = nk_tap - nk_kh * p.KW;
  const int chk = p.nocheck ? 0 : 1;
  auto issue_next = [&](int buf) {
    const int live = nk_t < kt_end ? 1 : 0;
    const long long tap_off = (long long)nk_kh * p.xs_h + (long long)nk_kw * p.xs_w + nk_c * BK;
    const uint32_t dst = lds_base + buf * STAGE;
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
      const int hi = a_hi0[i] + nk_kh, wi = a_wi0[i] + nk_kw;
      const int inimg = ((unsigned)hi < (unsigned)p.H ? 1 : 0) & ((unsigned)wi < (unsigned)p.W ? 1 : 0);
      const int ok = live & (a_hi0[i] > -(1 << 27) ? 1 : 0) & (inimg | (chk ^ 1));
      const uintptr_t real = (uintptr_t)(X + a_off[i] + tap_off);
      const uintptr_t src = ok ? real : (uintptr_t)zero_page;
      lds_dma16((const void*)src, dst + (wave * A_PIECES + i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
      const int ok = live & (b_ok[i] ? 1 : 0);
      const uintptr_t real = (uintptr_t)(Wt + b_off[i] + (long long)nk_t * BK);
      const uintptr_t src = ok ? real : (uintptr_t)zero_page;
      lds_dma16((const void*)src, dst + A_BYTES + (wave * B_PIECES + i) * 1024);
    }
    ++nk_t;
    if (++nk_c == tiles_per_tap) {
      nk_c = 0;
      if (++nk_kw == p.KW) { nk_kw = 0; ++nk_kh; }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue_next(s);
  int cur = 0, nxt = STAGES - 1;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES_PER_WAVE * (STAGES - 2)) : "memory");
    __builtin_amdgcn_s_barrier();
    issue_next(nxt);
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
    cur = (cur + 1 == STAGES) ? 0 : cur + 1;
    nxt = (nxt + 1 == STAGES) ? 0 : nxt + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing (zero-page) pieces must land before LDS is reused
  __syncthreads();

  // ---- epilogue
  float* C = (float*)smem;
#pragma unroll 1
  for (int pass = 0; pass < PASSES; ++pass) {
    if (pass > 0 || !EARLY_RES) fetch_residual(pass);
    if (wm / WR_PER_PASS == pass) {
      const int wr = wm % WR_PER_PASS;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            C[(wr * WTM + i * 32 + mfma32_row(r, lane)) * BN + wn * WTN + j * 32 + (lane & 31)] = acc[i][j][r];
    }
    __syncthreads();
    const int mbase = m0 + pass * PASS_ROWS;
    if (p.splitk > 1) {
      constexpr int CPR4 = BN / 4;
      float* P = p.partial + ((long long)(slice * gridDim.z + g) * p.M) * p.Cout;
      for (int idx = tid; idx < PASS_ROWS * CPR4; idx += NT) {
        const int r = idx / CPR4, c = (idx - r * CPR4) * 4;
        const int m = mbase + r, n = n0 + c;
        if (m < p.M && n < p.Cout) *(float4*)(P + (long long)m * p.Cout + n) = *(const float4*)(C + r * BN + c);
      }
    } else {
      T* Y = (T*)p.y + (long long)g * p.y_g;
      const float* Bv = p.bias ? p.bias + (long long)g * p.bias_g : nullptr;
#pragma unroll
      for (int q = 0; q < CH_PER_THREAD; ++q) {
        const int idx = tid + q * NT;
        const int r = idx / CPRO, c = (idx - r * CPRO) * EPC;
        const int m = mbase + r, n = n0 + c;
        if (m >= p.M || n >= p.Cout) continue;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 t = *(const float4*)(C + r * BN + c + e);
          v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        if (Bv) {
#pragma unroll
          for (int e = 0; e < EPC; e += 4) {
            const float4 t = *(const float4*)(Bv + n + e);
            v[e] += t.x; v[e + 1] += t.y; v[e + 2] += t.z; v[e + 3] += t.w;
          }
        }
        if (has_res) {
          float rv[EPC];
          chunk_to_f32(rpre[q], rv, (T*)nullptr);
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
    if (pass + 1 < PASSES) __syncthreads();
  }
}
