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
#include <type_traits>

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
#define MCG_MAX_DEVICES 64                // per-device launch state (CU count, raised LDS limit) of the persistent kernels
#define MCG_DMA_MAX_BYTES 0x7FFFFF00ll  // operands must fit the 2 GiB descriptor window

// One wave-wide 1 KiB HBM -> LDS piece: lane l's 16 bytes land at LDS (lds_base_uniform + IMM) + 16 * l.
template <int IMM>
__device__ __forceinline__ void lds_dma16(uint32_t voffset, const u32x4& srd, uint32_t soffset_uniform, uint32_t lds_base_uniform) {
  asm volatile(
      "s_add_u32 m0, %2, %4\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %0, %1, %3 offen lds"
      :
      : "v"(voffset), "s"(srd), "s"(lds_base_uniform), "s"(soffset_uniform), "n"(IMM)
      : "memory", "scc");
}

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// Eight f32 values (two 16-byte chunks) -> bf16 high parts and bf16 low parts: hi = RNE(x), lo = RNE(x - float(hi)); the
// subtraction is exact (Sterbenz), so hi + lo = x to 2^-17 relative.
typedef __fp16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
// Two f32 -> their fp16 HIGH parts and fp16 LOW parts, packed: hi = RTZ_f16(x), lo = RTZ_f16(x - hi).  x - hi is exact, so hi + lo = x
// to 2^-21 relative (22 significant bits) for |x| in [6e-5 * 2^11, 65504]; below that the low part is a subnormal half (absolute error
// <= 6e-8), above it the halves saturate at +-65504 each (round toward zero never produces an infinity from a finite value).
// Four VALU per pair: v_cvt_pkrtz, two v_fma_mix_f32 (x + (-1) * f32(half): the mixed-precision FMA reads the packed half directly, no
// v_cvt_f32_f16 in front of a v_sub), v_cvt_pkrtz.  The -1 goes through an opaque scalar: written as a literal, the compiler folds the FMA
// back into a subtraction of a converted value (six VALU per pair).  Same bits either way -- the difference is exact -- and the split is the
// one piece of VALU work every f16x3 kernel does per activation element; on a chip that runs this path on its power cap
// instructions are energy.
__device__ __forceinline__ float mcg_opaque_m1() {
  float m = -1.0f;
  asm("" : "+s"(m));
  return m;
}
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& h, uint32_t& l) {
  const f16x2_t hh = __builtin_amdgcn_cvt_pkrtz(a, b);
  h = __builtin_bit_cast(uint32_t, hh);
  const float m1 = mcg_opaque_m1();
  l = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(__builtin_fmaf((float)hh[0], m1, a), __builtin_fmaf((float)hh[1], m1, b)));
}
// one 32x32x16 product term of the split contraction (operands travel as 16-byte vectors; they hold eight halves)
__device__ __forceinline__ f32x16 x3_mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split_f32x8(const uint4& c0, const uint4& c1, bf16x8& hi, bf16x8& lo) {
  const uint32_t x[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  uint32_t h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair(__uint_as_float(x[2 * q]), __uint_as_float(x[2 * q + 1]), h[q], l[q]);
  hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
  lo = __builtin_bit_cast(bf16x8, make_uint4(l[0], l[1], l[2], l[3]));
}

// X3 != 0 (T = float): the "f16x3" contraction of the MCG_F16X3 engine.  Activations stay f32 in HBM and LDS; the weight
// operand is the split-packed form (packing.py::split_pack: per 8 consecutive K elements a 16-byte chunk of fp16 HIGH parts,
// then a 16-byte chunk of fp16 LOW parts, w = hi + lo to 2^-21 -- 4 bytes per element, so the DMA geometry is that of an f32
// matrix).  An A fragment is split in registers (split_pair: hi = RTZ_f16(x), lo = RTZ_f16(x - hi), the difference is exact)
// and each 32x32x16 product is three v_mfma_f32_32x32x16_f16: lo.hi + hi.lo + hi.hi, f32 accumulate (the dropped lo.lo term is
// 2^-22 relative).  Measured end to end: 1e-5 rad on (yaw, pitch) against the f32 oracle, at the fp16 matrix-pipe rate / 3.  (The
// first version split into bf16 halves -- 16 significant bits, 6e-5 rad -- at the same speed; fp16 halves carry 22.)
template <typename T, int BM, int BN, int BKB, int WAVES_M, int WAVES_N, int STAGES, int MINW = 2, int X3 = 0>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, MINW) void igemm_dma_kernel(const IgemmParams p) {
  static_assert(!X3 || (sizeof(T) == 4 && BKB == 128), "f16x3 mode: f32 storage, 128-byte K slices (32 channels)");
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
  constexpr int RING_BYTES = STAGES * STAGE;
  // the 256x128 / 8-wave tile rounds its 48 KiB ring up to 64 KiB (still two workgroups per CU): two wave-rows of epilogue staging
  // per pass instead of one halves the epilogue's barriers
  constexpr int LDS_BYTES = (BM == 256 && BN == 128 && NW == 8 && RING_BYTES == 49152 && MINW <= 2) ? 65536 : RING_BYTES;
  // epilogue passes: as many wave-rows (WTM output rows each) per pass as fit the ring's footprint
  constexpr int WR_FIT = LDS_BYTES / (WTM * BN * 4);
  // ... rounded down to a divisor of WAVES_M: a last pass with fewer wave-rows than the others would store stale staging rows
  constexpr int WR_WANT = WR_FIT >= WAVES_M ? WAVES_M : (WR_FIT >= 4 && WAVES_M % 4 == 0 ? 4 : (WR_FIT >= 2 && WAVES_M % 2 == 0 ? 2 : 1));
  // ... and small enough that the double-buffered residual rows stay within 2 x 8 chunks per thread (f32 tiles are wide in chunks)
  constexpr int WR_CAP = (8 * NT) / (WTM * (BN / EPC)) >= 1 ? (8 * NT) / (WTM * (BN / EPC)) : 1;
  constexpr int WR_PER_PASS = !X3 ? WR_WANT : (WR_WANT <= WR_CAP ? WR_WANT : (WR_CAP >= 2 && WAVES_M % 2 == 0 ? 2 : 1));
  constexpr int PASSES = (WAVES_M + WR_PER_PASS - 1) / WR_PER_PASS;
  constexpr int PASS_ROWS = WR_PER_PASS * WTM;
  constexpr int CPRO = BN / EPC;                         // output chunks per row
  constexpr int CH_PER_THREAD = PASS_ROWS * CPRO / NT;   // output chunks per thread per pass
  // register budget: prefetch the first pass's residual before the K loop only when it does not cost occupancy
  constexpr bool EARLY_RES = TM * TN * 16 + CH_PER_THREAD * 4 <= 144 && MINW <= 2 && !(LDS_BYTES != RING_BYTES);
  static_assert(A_PIECES >= 1 && B_PIECES >= 1 && A_PIECES * RPP * NW == BM && B_PIECES * RPP * NW == BN, "tile / wave count mismatch");
  static_assert(STAGES >= 2, "ring needs >= 2 stages");
  static_assert(PIECES_PER_WAVE * (STAGES - 2) <= 63, "vmcnt field");
  static_assert(WTM * BN * 4 <= LDS_BYTES, "epilogue staging does not fit");
  static_assert(PASS_ROWS * CPRO % NT == 0, "epilogue chunk split");
  static_assert(WAVES_M % WR_PER_PASS == 0, "epilogue passes must tile the wave-rows");
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
  const u32x4 srd_a = make_srd((const T*)p.x + (long long)(p.x_g_period > 0 ? g % p.x_g_period : g) * p.x_g);
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
  uint32_t a_voff[A_PIECES], a_mask[A_PIECES], a_eff[A_PIECES], b_voff[B_PIECES];
  const bool a_linear = p.pad == 0 && p.nocheck &&
                        (HoWo == 1 || (p.stride == 1 && p.xs_h == (long long)p.Wo * p.xs_w && p.xs_n == (long long)p.Ho * p.xs_h));
  const long long a_slope = HoWo == 1 ? p.xs_n : p.xs_w;
#pragma unroll
  for (int i = 0; i < A_PIECES; ++i) {
    const int row = (wave * A_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int m = min(m0 + row, p.M - 1);  // rows beyond M re-read the last row; their outputs are never stored
    uint32_t mask = 0xffffffffu;
    if (a_linear) {  // dense unpadded stride-1 input (every 1x1 conv and linear): the offset is affine in m, no divisions
      a_voff[i] = (uint32_t)(((long long)m * a_slope + chunk * EPC) * ES);
      a_mask[i] = mask;
      continue;
    }
    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    const long long off = (long long)n * p.xs_n + (long long)hi0 * p.xs_h + (long long)wi0 * p.xs_w + chunk * EPC;
    a_voff[i] = (uint32_t)(off * ES);  // wraps below zero for padded border pixels; exact again (mod 2^32) once a valid tap is added
    if (!p.nocheck) {
      mask = 0;
      for (int kh = 0; kh < p.KH; ++kh)
        for (int kw = 0; kw < p.KW; ++kw)
          if ((unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.W) mask |= 1u << (kh * p.KW + kw);
    }
    a_mask[i] = mask;
  }
  // second (K-concatenated, 1x1) source: only needed once, when the K loop crosses into it
  auto second_source_voff = [&](int i) -> uint32_t {
    const int row = (wave * A_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int m = min(m0 + row, p.M - 1);
    const int n = m / HoWo, rem = m - n * HoWo, ho = rem / p.Wo, wo = rem - ho * p.Wo;
    return (uint32_t)(((long long)n * p.xs2_n + (long long)(ho * p.stride2) * p.xs2_h + (long long)(wo * p.stride2) * p.xs2_w + chunk * EPC) * ES);
  };
  const long long K = (long long)p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0);
#pragma unroll
  for (int i = 0; i < B_PIECES; ++i) {
    const int row = (wave * B_PIECES + i) * RPP + drow;
    const int chunk = dcs ^ ((row / RPB) % CPR);
    const int n = min(n0 + row, p.Cout - 1);
    b_voff[i] = (uint32_t)(((long long)n * K + chunk * EPC) * ES);
  }

  // ---- issue stream.  K is walked as a sequence of SEGMENTS: one per filter tap (kh, kw) of the first source
  // (Cin / BK tiles each), then one for the optional second source.  Everything that depends on the segment -- the
  // descriptor, the per-lane offsets with the tap folded in and out-of-image taps swapped for the out-of-range offset
  // -- is recomputed only when a segment ends (rare, uniform branch).  Per K-tile the issue costs PIECES_PER_WAVE x
  // (s_add m0 / s_nop / buffer_load) plus four scalar updates: SQ counters on the earlier per-tile tap bookkeeping
  // showed ~60 SALU and a dozen branches per 8 MFMAs, the scalar unit as busy as the matrix pipe.
  const int n_tiles = kt_end - kt_begin;
  int sg_kh = 0, sg_kw = 0, sg_left = 0;
  uint32_t sg_bit = 1u, soff_a = 0, soff_b = (uint32_t)kt_begin * BKB;
  bool sg_second = false;
  u32x4 srd_cur = srd_a;
  auto segment_setup = [&]() {
    if (!sg_second) {
      const uint32_t tap_bytes = (uint32_t)(((long long)sg_kh * p.xs_h + (long long)sg_kw * p.xs_w) * ES);
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) a_eff[i] = (a_mask[i] & sg_bit) ? a_voff[i] + tap_bytes : MCG_OOB_OFFSET;
    } else {
      srd_cur = srd_a2;
#pragma unroll
      for (int i = 0; i < A_PIECES; ++i) a_eff[i] = second_source_voff(i);
    }
  };
  {
    int c0;
    if (kt_begin < KT1) {
      const int tap = kt_begin / tiles_per_tap;
      c0 = kt_begin - tap * tiles_per_tap;
      sg_kh = tap / p.KW;
      sg_kw = tap - sg_kh * p.KW;
      sg_bit = 1u << tap;
      sg_left = tiles_per_tap - c0;
    } else {
      c0 = kt_begin - KT1;
      sg_second = true;
      sg_left = KT - kt_begin;
    }
    soff_a = (uint32_t)c0 * BKB;
    segment_setup();
  }
  auto segment_advance = [&]() {
    soff_a = 0;
    if (!sg_second) {
      sg_bit <<= 1;
      if (++sg_kw == p.KW) { sg_kw = 0; ++sg_kh; }
      if (sg_kh == p.KH) { sg_second = true; sg_left = p.x2 ? p.Cin2 / BK : 0x7fffffff; if (!p.x2) return; }
      else sg_left = tiles_per_tap;
    } else {
      sg_left = 0x7fffffff;
      return;
    }
    segment_setup();
  };
  const uint32_t dst_a = lds_base + wave * (A_PIECES * 1024), dst_b = lds_base + A_BYTES + wave * (B_PIECES * 1024);
  // ring slot = compile-time byte offset IMM (main loop) + uniform run-time offset roff (remainder loop)
  auto issue_tile = [&](auto imm_c, uint32_t roff) {
    constexpr int IMM = decltype(imm_c)::value;
    static_for<A_PIECES>([&](auto i) { lds_dma16<IMM + decltype(i)::value * 1024>(a_eff[decltype(i)::value], srd_cur, soff_a, dst_a + roff); });
    static_for<B_PIECES>([&](auto i) { lds_dma16<IMM + decltype(i)::value * 1024>(b_voff[decltype(i)::value], srd_b, soff_b, dst_b + roff); });
    soff_a += BKB;
    soff_b += BKB;
    if (--sg_left == 0) segment_advance();
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment addresses: every 32-row fragment of a wave shares the swizzle key (32 / RPB is a multiple of CPR), so one
  // VGPR per K-chunk pair and operand; fragment index and ring stage are immediates of the ds_read
  static_assert((32 / RPB) % CPR == 0, "swizzle key must repeat every 32 rows");
  const int fkey = ((lane & 31) / RPB) % CPR;
  const char* fa[CPR / 2];
  const char* fb[CPR / 2];
#pragma unroll
  for (int j2 = 0; j2 < CPR / 2; ++j2) {
    // plain: K-chunk 2 j2 + half.  f16x3: the lane's 8 channels of MFMA step j = j2 / 2 are chunks 4 j + 2 half + (j2 & 1): for A
    // the first / last four f32 values, for W the fp16 high parts / low parts of the same 8 channels
    const int chunk = X3 ? 4 * (j2 >> 1) + 2 * (lane >> 5) + (j2 & 1) : 2 * j2 + (lane >> 5);
    const int cb = (chunk ^ fkey) << 4;
    fa[j2] = smem + (wm * WTM + (lane & 31)) * BKB + cb;
    fb[j2] = smem + A_BYTES + (wn * WTN + (lane & 31)) * BKB + cb;
  }
  // Big wave tiles (>= 8 MFMAs per K-step) run two waves per SIMD: too few to hide an LDS round trip per K-step behind the other
  // waves, so the fragments are double-buffered in registers -- step j+1's reads are issued before step j's MFMAs.
  constexpr bool SWP = TM * TN >= 8 && CPR / 2 >= 2;
  auto compute_tile = [&](auto imm_c, uint32_t roff) {
    constexpr int IMM = decltype(imm_c)::value;
    if constexpr (X3 != 0) {
      static_for<CPR / 4>([&](auto jc) {
        constexpr int J = decltype(jc)::value;
        bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          split_f32x8(*(const uint4*)(fa[2 * J] + roff + IMM + i * 32 * BKB), *(const uint4*)(fa[2 * J + 1] + roff + IMM + i * 32 * BKB), ah[i], al[i]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          bh[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(fb[2 * J] + roff + IMM + j * 32 * BKB));
          bl[j] = __builtin_bit_cast(bf16x8, *(const uint4*)(fb[2 * J + 1] + roff + IMM + j * 32 * BKB));
        }
        // small terms first; consecutive MFMAs never share an accumulator
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = x3_mfma(al[i], bh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = x3_mfma(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = x3_mfma(ah[i], bh[j], acc[i][j]);
      });
    } else if constexpr (SWP) {
      uint4 af[2][TM], bf[2][TN];
      auto load = [&](auto j2c) {
        constexpr int J2 = decltype(j2c)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[J2 & 1][i] = *(const uint4*)(fa[J2] + roff + IMM + i * 32 * BKB);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[J2 & 1][j] = *(const uint4*)(fb[J2] + roff + IMM + j * 32 * BKB);
      };
      load(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      static_for<CPR / 2>([&](auto j2c) {
        constexpr int J2 = decltype(j2c)::value;
        if constexpr (J2 + 1 < CPR / 2) load(std::integral_constant<int, J2 + 1>{});
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[J2 & 1][i], bf[J2 & 1][j]);
        // issue order: one ds_read of the next step behind each of this step's first MFMAs
        if constexpr (J2 + 1 < CPR / 2) {
          static_for<TM + TN>([&](auto) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          });
          __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - (TM + TN), 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
        }
      });
    } else {
#pragma unroll
      for (int j2 = 0; j2 < CPR / 2; ++j2) {
        uint4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *(const uint4*)(fa[j2] + roff + IMM + i * 32 * BKB);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *(const uint4*)(fb[j2] + roff + IMM + j * 32 * BKB);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) Mma<T>::run(acc[i][j], af[i], bf[j]);
      }
    }
  };
  typedef std::integral_constant<int, 0> zero_c;

  // tile t lives in ring slot t % STAGES; the slot tile t-1 just released receives tile t + STAGES - 1
  const int n_pre = min(n_tiles, STAGES - 1);
  static_for<STAGES - 1>([&](auto s) { if (decltype(s)::value < n_pre) issue_tile(std::integral_constant<int, decltype(s)::value * STAGE>{}, 0u); });
  const int n_main = n_tiles - n_pre;  // tiles whose step still has a successor to issue
  int kt = 0;
  for (; kt + STAGES <= n_main; kt += STAGES)  // steady state, unrolled over the ring so slots are immediates
    static_for<STAGES>([&](auto s) {
      constexpr int S = decltype(s)::value;
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES_PER_WAVE * (STAGES - 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_tile(std::integral_constant<int, ((S + STAGES - 1) % STAGES) * STAGE>{}, 0u);
      compute_tile(std::integral_constant<int, S * STAGE>{}, 0u);
    });
  // remainder (< STAGES issuing steps) and drain (n_pre steps, nothing left to issue): rolled, run-time slot
  uint32_t rs = 0, rn = (STAGES - 1) * STAGE;
#pragma nounroll
  for (; kt < n_tiles; ++kt) {
    if (kt < n_main) {
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PIECES_PER_WAVE * (STAGES - 2)) : "memory");
      __builtin_amdgcn_s_barrier();
      issue_tile(zero_c{}, rn);
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    compute_tile(zero_c{}, rs);
    rs = rs + STAGE == STAGES * STAGE ? 0 : rs + STAGE;
    rn = rn + STAGE == STAGES * STAGE ? 0 : rn + STAGE;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing prefetch pieces must land before LDS is reused
  __syncthreads();

  // ---- epilogue
  float* C = (float*)smem;
  if (!EARLY_RES) fetch_residual(0, rpre_a);
  // every chunk a thread stores lies in the same 16-byte column group (NT is a multiple of the chunks per row): its bias
  // values and column pointer are fetched once
  static_assert(NT % CPRO == 0, "epilogue column ownership");
  const int cth = (tid % CPRO) * EPC, rth = tid / CPRO;
  const bool col_ok = n0 + cth < p.Cout;
  T* yrow = (T*)p.y + (long long)g * p.y_g + n0 + cth + (long long)(m0 + rth) * p.y_row_stride;  // + uniform row steps per store
  const float wsc = X3 ? p.wscale : 1.f;   // exact power of two (f16x3 pre-scaled weights); 1 elsewhere
  float bv[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) bv[e] = 0.f;
  if (p.bias && col_ok && p.splitk <= 1) {
    const float* Bv = p.bias + (long long)g * p.bias_g + n0 + cth;
#pragma unroll
    for (int e = 0; e < EPC; e += 4) {
      const float4 t = *(const float4*)(Bv + e);
      bv[e] = t.x; bv[e + 1] = t.y; bv[e + 2] = t.z; bv[e + 3] = t.w;
    }
  }
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
        if (m < p.M && n < p.Cout) {
          float4 t = *(const float4*)(C + r * BN + c);
          if (X3) { t.x *= wsc; t.y *= wsc; t.z *= wsc; t.w *= wsc; }
          *(float4*)(P + (long long)m * p.Cout + n) = t;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < CH_PER_THREAD; ++q) {
        const int r = rth + q * (NT / CPRO);
        if (mbase + r >= p.M || !col_ok) continue;
        float v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; e += 4) {
          const float4 t = *(const float4*)(C + r * BN + cth + e);
          if (X3) {   // t * 2^k is exact, so this is (acc * wscale) + bias whether or not the compiler contracts it into an fma
            v[e] = t.x * wsc + bv[e]; v[e + 1] = t.y * wsc + bv[e + 1]; v[e + 2] = t.z * wsc + bv[e + 2]; v[e + 3] = t.w * wsc + bv[e + 3];
          } else {
            v[e] = t.x + bv[e]; v[e + 1] = t.y + bv[e + 1]; v[e + 2] = t.z + bv[e + 2]; v[e + 3] = t.w + bv[e + 3];
          }
        }
        if (has_res) {
          float rv[EPC];
          chunk_to_f32(rpre[q], rv, (T*)nullptr);
#pragma unroll
          for (int e = 0; e < EPC; ++e) v[e] += rv[e];
        }
        uint4 o = f32_to_chunk(v, (T*)nullptr);
        if (p.relu) o = relu_chunk(o, (T*)nullptr);
        *(uint4*)(yrow + (long long)(pass * PASS_ROWS + q * (NT / CPRO)) * p.y_row_stride) = o;
      }
    }
    if (pass + 1 < PASSES) __syncthreads();
  }
}
