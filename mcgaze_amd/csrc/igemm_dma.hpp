// Implicit-GEMM kernel, LDS-DMA pipelined variant (the throughput path).
//
// Same contraction, data layouts and LDS swizzle as igemm.hpp, but the A / W K-slices travel
// HBM -> LDS directly (`buffer_load_dwordx4 ... offen lds`, 1 KiB per wave-instruction, no VGPR round
// trip, no ds_write), through a ring of STAGES LDS buffers with STAGES-1 K-tiles in flight:
//
//   iteration kt:  s_waitcnt vmcnt(PIECES_PER_WAVE*(STAGES-2))   this wave's pieces of tile kt landed
//                  s_barrier                                      everyone's pieces landed AND everyone
//                                                                 finished reading tile kt-1's buffer
//                  issue DMA of tile kt+STAGES-1 into the buffer tile kt-1 just released
//                  ds_read_b128 fragments of tile kt + MFMAs
//
// one barrier per K-tile, never vmcnt(0) inside the loop.  The DMA destination is lane-linear
// (LDS base + lane*16), so the bank-conflict-free XOR swizzle is applied on the per-lane SOURCE
// offset: lane l of a piece fetches (row r0 + l/CPR, chunk (l%CPR) ^ key(row)).
//
// Address generation is kept off the VALU (PMC on the first DMA version showed ~60 VALU + ~70 SALU per
// 16 MFMAs with 64-bit per-lane pointers): both operands are read through buffer descriptors, the
// per-lane byte offset (frame / row / chunk) is loop-invariant, and the K-tile advance (tap (kh,kw) and
// channel slice) is a scalar.  Spatial zero padding uses the descriptor's range check: each lane holds a
// bit mask of the taps that fall inside the image, and an out-of-image tap swaps the lane's offset for
// one beyond num_records, which the hardware returns as zeros.  Rows beyond M / Cout and prefetches
// beyond the last K-tile are clamped to valid addresses instead (their accumulators are never stored),
// so every piece always writes its full 1 KiB and the vmcnt arithmetic stays uniform.  1x1 / unpadded
// convs and linears issue ZERO per-K-tile VALU for addressing (offset in `soffset`); padded convs 3.
//
// Epilogue: the residual rows this thread will need are fetched into registers BEFORE the K loop
// (their HBM latency hides under the whole contraction); accumulators are staged through LDS as
// f32 (unpadded [rows][BN]: ds_write_b32 by 32-lane row segments and ds_read_b128 along rows are
// both conflict-free) in as few passes as the LDS ring's footprint allows, then bias + residual
// (+ nearest-upsampled FPN top-down term) + ReLU are applied on coalesced 16-byte row chunks.
#pragma once
#include "igemm.hpp"

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raw buffer descriptor over [base, base + 2 GiB): offsets >= 0x80000000 read as zero.
__device__ __forceinline__ u32x4 make_srd(const void* base) {
  const uint64_t b = (uint64_t)base;
  u32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((uint32_t)b);
  r.y = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
  r.z = 0x80000000u;
  r.w = 0x00020000u;
  return r;
}
#define MCG_OOB_OFFSET 0xFFFFFF00u
#define MCG_DMA_MAX_BYTES 0x7FFFFF00ll  // operands must fit the 2 GiB descriptor window

__device__ __forceinline__ void lds_dma16(uint32_t voffset, const u32x4& srd, uint32_t soffset_uniform, uint32_t lds_dst_uniform) {
  asm volatile(
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voffset), "s"(srd), "s"(lds_dst_uniform), "s"(soffset_uniform)
      : "memory");
}

template <typename T, int BM, int BN, int BKB, int WAVES_M, int WAVES_N, int STAGES, int MINW = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void igemm_dma_kernel(const IgemmParams p) {
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int CPR = BKB / 16;
  constexpr int BK = BKB / ES;
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
  // register budget: prefetch the first pass's residual before the K loop only when it does not cost occupancy
  constexpr bool EARLY_RES = TM * TN * 16 + CH_PER_THREAD * 4 <= 144 && MINW <= 2;
  static_assert(A_PIECES >= 1 && B_PIECES >= 1 && A_PIECES * RPP * NW == BM && B_PIECES * RPP * NW == BN, "tile / wave count mismatch");
  static_assert(STAGES >= 2, "ring needs >= 2 stages");
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

  const int tiles_per_tap = p.Cin / BK;
  const int KT1 = p.KH * p.KW * tiles_per_tap;
  const int KT = KT1 + (p.x2 ? p.Cin2 / BK : 0);  // optional K-concatenated second A source (1x1 taps)
  const int kt_begin = slice * p.tiles_per_slice;
  const int kt_end = min(KT, kt_begin + p.tiles_per_slice);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const int HoWo = p.Ho * p.Wo;
  const u32x4 srd_a = make_srd((const T*)p.x + (long long)g * p.x_g);
  const u32x4 srd_b = make_srd((const T*)p.w + (long long)g * p.w_g);
  const u32x4 srd_a2 = make_srd(p.x2 ? p.x2 : p.x);

  // ---- residual prefetch for epilogue pass 0 (latency hidden under the K loop)
  const T* __restrict__ R = (const T*)p.res + (long long)g * p.res_g;
  const bool has_res = p.res_mode != MCG_RES_NONE && p.splitk <= 1;
  uint4 rpre_a[CH_PER_THREAD], rpre_b[CH_PER_THREAD];  // double-buffered: pass p+1's rows are in flight while pass p is stored
  auto fetch_residual = [&](int pass, uint4 (&rpre)[CH_PER_THREAD]) {
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
  if (EARLY_RES) fetch_residual(0, rpre_a);

  // ---- per-lane DMA source offsets (bytes, loop-invariant) and in-image tap masks
  const int drow = lane / CPR, dcs = lane % CPR;
  uint32_t a_voff[A_PIECES], a_voff2[A_PIECES], a_mask[A_PIECES], b_voff[B_PIECES];
  const int chk = p.nocheck ? 0 : 1;
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int row = (wave * A_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int m = min(m0 + row, p.M - 1);  // rows beyond M re-read the last row; their outputs are never stored
    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    const long long off = (long long)n * p.xs_n + (long long)hi0 * p.xs_h + (long long)wi0 * p.xs_w + chunk * EPC;
    a_voff[i] = (uint32_t)(off * ES);  // wraps below zero for padded border pixels; exact again (mod 2^32) once a valid tap is added
    uint32_t mask = 0;
    if (chk) {
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W) mask |= 1u << (kh * p.KW + kw);
    }
    a_mask[i] = mask;
    a_voff2[i] = (uint32_t)(((long long)n * p.xs2_n + (long long)(ho * p.stride2) * p.xs2_h + (long long)(wo * p.stride2) * p.xs2_w + chunk * EPC) * ES);
  }
  const long long K = (long long)p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0);
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int row = (wave * B_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int n = min(n0 + row, p.Cout - 1);
    b_voff[i] = (uint32_t)(((long long)n * K + chunk * EPC) * ES);
  }

  // running state of the NEXT K-tile to issue (all scalar): tap index, channel slice
  int nk_t = kt_begin;
  int nk_tap = kt_begin / tiles_per_tap;
  int nk_c = kt_begin - nk_tap * tiles_per_tap;
  int nk_kh = nk_tap / p.KW, nk_kw = nk_tap - nk_kh * p.KW;
  // One DMA piece of the next tile (q < A_PIECES: an A piece, else a W piece).  Issuing a piece costs the issuing wave
  // ~60-180 cycles (MI355X_MICROARCH: "LDS-DMA piece issue cost"), so in the main loop the pieces are spread between
  // the MFMAs of the current tile instead of being issued back to back right after the barrier.
  uint32_t is_tap_bytes = 0, is_tap_bit = 0, is_wk_bytes = 0, is_dst = 0;
  auto issue_begin = [&](int buf) {
    is_tap_bytes = (uint32_t)(((long long)nk_kh * p.xs_h + (long long)nk_kw * p.xs_w + nk_c * BK) * ES);
    is_tap_bit = 1u << nk_tap;
    is_wk_bytes = (uint32_t)nk_t * BKB;
    is_dst = lds_base + buf * STAGE;
  };
  auto issue_piece = [&](int q) {
    if (q < A_PIECES) {
      if (nk_t >= KT1) {  // second source (1x1): pure scalar K offset
        lds_dma16(a_voff2[q], srd_a2, (uint32_t)(nk_t - KT1) * BKB, is_dst + (wave * A_PIECES + q) * 1024);
      } else if (chk) {  // padded conv: fold the tap into the per-lane offset (exact mod 2^32), zero-fill out-of-image taps
        const uint32_t v = (a_mask[q] & is_tap_bit) ? a_voff[q] + is_tap_bytes : MCG_OOB_OFFSET;
        lds_dma16(v, srd_a, 0u, is_dst + (wave * A_PIECES + q) * 1024);
      } else {    // 1x1 / pre-padded / linear: the tap is a pure scalar offset
        lds_dma16(a_voff[q], srd_a, is_tap_bytes, is_dst + (wave * A_PIECES + q) * 1024);
      }
    } else {
      const int i = q - A_PIECES;
      lds_dma16(b_voff[i < B_PIECES ? i : 0], srd_b, is_wk_bytes, is_dst + A_BYTES + (wave * B_PIECES + i) * 1024);
    }
  };
  auto issue_end = [&]() {
    if (nk_t + 1 < kt_end) {  // prefetches beyond the last K-tile simply re-read it
      ++nk_t;
      if (nk_t < KT1 && ++nk_c == tiles_per_tap) {
        nk_c = 0;
        ++nk_tap;
        if (++nk_kw == p.KW) { nk_kw = 0; ++nk_kh; }
      }
    }
  };
  auto issue_next = [&](int buf) {
    issue_begin(buf);
#pragma unroll
    for (int q = 0; q < PIECES_PER_WAVE; ++q) issue_piece(q);
    issue_end();
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
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing prefetch pieces must land before LDS is reused
  __syncthreads();

  // ---- epilogue
  float* C = (float*)smem;
  if (!EARLY_RES) fetch_residual(0, rpre_a);
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    uint4 (&rpre)[CH_PER_THREAD] = (pass & 1) ? rpre_b : rpre_a;
    if (pass + 1 < PASSES) fetch_residual(pass + 1, (pass & 1) ? rpre_a : rpre_b);
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
