// Shared device helpers for the gfx950 (CDNA4) kernels.  Wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mcgaze_hip.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef uint16_t bf16_t;  // storage type of a bf16 element

#define MCG_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// f32 -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (two values per instruction; a software RNE costs ~6 VALU each
// and the epilogues convert every output element)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// fp16 storage (MCG_F16, round 6): a distinct 2-byte type, so that the 16-bit kernels instantiate once per number format.  f32 -> fp16 is
// round-to-nearest-even (v_cvt_f16_f32 x 2 + pack); values beyond +-65504 become infinities -- the engine is for checkpoints whose activations
// stay inside the fp16 range (the f16x3 engine needs the same of its operand halves; mcg_engine_range_audit counts the offenders).
struct f16_t { uint16_t v; };
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8_hw __attribute__((ext_vector_type(8)));
typedef float f32x8_hw __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  const f32x2_hw v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_hw));
}
__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }

// A packed pair of 16-bit elements <-> f32, per number format: what the specialised 16-bit kernels (pw_pair, pw_single, conv3x3_c64,
// stem_fused, chain, attn_block) use where they round an f32 value to the storage type or read one back.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2bf(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t pk) { return __uint_as_float(pk << 16); }
  __device__ static __forceinline__ float hi(uint32_t pk) { return __uint_as_float(pk & 0xffff0000u); }
};
template <> struct H16<f16_t> {
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) { return pack2h(lo, hi); }
  __device__ static __forceinline__ float lo(uint32_t pk) { return h2f((uint16_t)(pk & 0xffffu)); }
  __device__ static __forceinline__ float hi(uint32_t pk) { return h2f((uint16_t)(pk >> 16)); }
};

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int kPerChunk = 4;  // elements in a 16-byte chunk
  static constexpr mcg_dtype kDtype = MCG_F32;
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int kPerChunk = 8;
  static constexpr mcg_dtype kDtype = MCG_BF16;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

template <> struct Elem<f16_t> {
  static constexpr int kPerChunk = 8;
  static constexpr mcg_dtype kDtype = MCG_F16;
  __device__ static __forceinline__ float ld(const f16_t* p) { return h2f(p->v); }
  __device__ static __forceinline__ void st(f16_t* p, float v) { p->v = (uint16_t)(pack2h(v, 0.f) & 0xffffu); }
};

// 16-byte chunk <-> floats
__device__ __forceinline__ void chunk_to_f32(const uint4& c, float (&v)[4], float*) {
  v[0] = __uint_as_float(c.x); v[1] = __uint_as_float(c.y); v[2] = __uint_as_float(c.z); v[3] = __uint_as_float(c.w);
}
__device__ __forceinline__ void chunk_to_f32(const uint4& c, float (&v)[8], bf16_t*) {
  v[0] = __uint_as_float(c.x << 16); v[1] = __uint_as_float(c.x & 0xffff0000u);
  v[2] = __uint_as_float(c.y << 16); v[3] = __uint_as_float(c.y & 0xffff0000u);
  v[4] = __uint_as_float(c.z << 16); v[5] = __uint_as_float(c.z & 0xffff0000u);
  v[6] = __uint_as_float(c.w << 16); v[7] = __uint_as_float(c.w & 0xffff0000u);
}
__device__ __forceinline__ void chunk_to_f32(const uint4& c, float (&v)[8], f16_t*) {
  const f32x8_hw f = __builtin_convertvector(__builtin_bit_cast(f16x8_hw, c), f32x8_hw);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = f[e];
}
__device__ __forceinline__ uint4 f32_to_chunk(const float (&v)[4], float*) {
  return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
}
__device__ __forceinline__ uint4 f32_to_chunk(const float (&v)[8], bf16_t*) {
  return make_uint4(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7]));
}

__device__ __forceinline__ uint4 f32_to_chunk(const float (&v)[8], f16_t*) {
  return make_uint4(pack2h(v[0], v[1]), pack2h(v[2], v[3]), pack2h(v[4], v[5]), pack2h(v[6], v[7]));
}

// ReLU on a stored 16-byte chunk.  bf16: sign-magnitude patterns order like signed 16-bit integers, so a packed signed max with 0
// clears exactly the negative values (v_pk_max_i16, 4 instructions per 8 elements).
typedef short s16x8_hw __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint4 relu_chunk(const uint4& c, float*) {
  return make_uint4(__float_as_uint(fmaxf(__uint_as_float(c.x), 0.f)), __float_as_uint(fmaxf(__uint_as_float(c.y), 0.f)),
                    __float_as_uint(fmaxf(__uint_as_float(c.z), 0.f)), __float_as_uint(fmaxf(__uint_as_float(c.w), 0.f)));
}
__device__ __forceinline__ uint4 relu_chunk(const uint4& c, bf16_t*) {
  const s16x8_hw z = {0, 0, 0, 0, 0, 0, 0, 0};
  return __builtin_bit_cast(uint4, __builtin_elementwise_max(__builtin_bit_cast(s16x8_hw, c), z));
}

__device__ __forceinline__ uint4 relu_chunk(const uint4& c, f16_t*) { return relu_chunk(c, (bf16_t*)nullptr); }   // fp16 is sign-magnitude too

// One 32x32 MFMA "chunk pair" step: each lane holds 16 bytes of A (row lane&31) and 16 bytes of B
// (column lane&31) taken from K-chunk 2j + (lane>>5).  bf16: one v_mfma_f32_32x32x16_bf16.
// fp16: one v_mfma_f32_32x32x16_f16.  f32: four v_mfma_f32_32x32x2_f32 (exact f32 fma chain); the K order inside the pair is
// permuted identically for A and B, which leaves the dot product's term set unchanged.
template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<f16_t> {
  __device__ static __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_hw, a), __builtin_bit_cast(f16x8_hw, b), acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  __device__ static __forceinline__ void run(f32x16& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
  }
};

// C/D element (reg r of lane l) of a 32x32 MFMA tile: col = l & 31, row = (r&3) + 8*(r>>2) + 4*(l>>5)
__device__ __forceinline__ int mfma32_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Wave-wide reductions on the DPP path (VALU-speed lane permutes; the __shfl_xor butterfly goes through ds_bpermute, an LDS round
// trip per step -- one LayerNorm row cost ~1 us of pure latency, and the decoder runs thousands of them).  Rows of 16 lanes reduce
// with quad_perm / row_half_mirror / row_mirror, rows combine with row_bcast15 / row_bcast31, lane 63 holds the result and is
// read back as a scalar.  Fixed order -> deterministic.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_mov(float v, float fill) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1>(v, 0.f);        // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v, 0.f);        // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v, 0.f);       // row_half_mirror
  v += dpp_mov<0x140>(v, 0.f);       // row_mirror: every lane holds its row's sum
  v += dpp_mov<0x142, 0xA>(v, 0.f);  // row_bcast15 into rows 1, 3
  v += dpp_mov<0x143, 0xC>(v, 0.f);  // row_bcast31 into rows 2, 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v, v));
  v = fmaxf(v, dpp_mov<0x4E>(v, v));
  v = fmaxf(v, dpp_mov<0x141>(v, v));
  v = fmaxf(v, dpp_mov<0x140>(v, v));
  v = fmaxf(v, dpp_mov<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_mov<0x143, 0xC>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// XCD-aware block remap (8 XCDs, block b runs on XCD b % 8): give each XCD a contiguous run of
// logical tile ids so that tiles sharing an operand panel meet in one L2.  Bijective for any grid.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// ---------------------------------------------------------------- dtype helpers of the host launchers
static inline bool mcg_is16(mcg_dtype dt) { return dt == MCG_BF16 || dt == MCG_F16; }   // 2-byte activation storage

// ---------------------------------------------------------------- host side error plumbing
void mcg_set_error(const char* fmt, ...);
#define MCG_CHECK_ARG(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      mcg_set_error(__VA_ARGS__);       \
      return MCG_ERR_ARG;               \
    }                                   \
  } while (0)
#define MCG_CHECK_LAUNCH(what)                                                       \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      mcg_set_error("%s: %s", what, hipGetErrorString(e__));                         \
      return MCG_ERR_HIP;                                                            \
    }                                                                                \
  } while (0)
#define MCG_TRY(expr)                 \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != MCG_OK) return rc__;  \
  } while (0)
