// Host launchers for the implicit-GEMM kernel + the trunk's memory-bound helpers
// (stem input packing, 3x3/s2 max-pool, NCHW<->NHWC).
#include "igemm_dma.hpp"
#include "stem_fused.hpp"
#include "conv3x3_c64.hpp"

#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";
void mcg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mcg_last_error(void) { return g_err; }
extern "C" int mcg_abi_version(void) { return MCG_ABI_VERSION; }
// (mcg_build_id lives in build_id.hip: the one object that is rebuilt whenever ANY kernel source or the public header changes)
extern "C" int mcg_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    mcg_set_error("mcg_device_info: no HIP device");
    return MCG_ERR_HIP;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return MCG_OK;
}


// ------------------------------------------------------------------------------------------------
// Per-launch profiling of the contraction kernel (bench.py's roofline line): when the context carries an armed Prof, every
// contraction launch is bracketed by a hipEvent pair on the launch stream and tagged with its algorithmic FLOPs
// (2*M*Cout*K, true K for the zero-padded stem) and tile configuration.  The records belong to the engine (engine.hip).
ProfRec* prof_begin(const McgCtx& ctx, hipStream_t s, int cfg, int M, int N, int K, double flops, double bytes) {
  Prof* pr = ctx.prof;
  if (!pr || pr->n >= pr->cap) return nullptr;
  ProfRec* rec = &pr->recs[pr->n++];
  rec->cfg = cfg;
  rec->shape[0] = M; rec->shape[1] = N; rec->shape[2] = K;
  rec->flops = flops;
  rec->bytes = bytes;
  (void)hipEventRecord(rec->a, s);
  return rec;
}
void prof_end(ProfRec* rec, hipStream_t s) {
  if (rec) (void)hipEventRecord(rec->b, s);
}
static double algo_flops(const IgemmParams& p, int groups) {
  return 2.0 * p.M * p.Cout * (p.algo_k > 0 ? (double)p.algo_k : (double)p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0)) * groups;
}
static int gemm_k(const IgemmParams& p) { return p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0); }
// Algorithmic HBM bytes of one launch (layer-granular): every input pixel the taps touch read ONCE (a 1x1 /
// stride-s conv needs only the sampled pixels; a KxK conv the whole map), the second source and the residual rows once, the output
// written once, the weights once.  es = bytes per activation element; split-K slabs count as f32 outputs.
static double algo_bytes(const IgemmParams& p, int groups, int es) {
  const double frames = (double)p.M / ((double)p.Ho * p.Wo);
  const double in_px = (p.KH == 1 && p.KW == 1) ? (double)p.M : frames * p.H * p.W;
  double b = in_px * p.Cin * es + (p.x2 ? (double)p.M * p.Cin2 * es : 0.0);
  b += p.splitk > 1 ? (double)p.splitk * p.M * p.Cout * 4 : (double)p.M * p.Cout * es;
  if (p.res_mode == MCG_RES_ADD) b += (double)p.M * p.Cout * es;
  if (p.res_mode == MCG_RES_UPSAMPLE_ADD) b += frames * p.Hr * p.Wr * p.Cout * es;
  b += (double)p.Cout * gemm_k(p) * es;
  return b * groups;
}

template <typename T, int BM, int BN, int BKB, int WM, int WN>
static void launch_cfg(hipStream_t s, const IgemmParams& p, int groups) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  dim3 grid(tiles, p.splitk, groups);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, BKB, WM, WN>), grid, dim3(256), 0, s, p);
}

template <typename T, int BM, int BN, int BKB, int WM, int WN, int ST, int MINW = 2, int X3 = 0>
static void launch_dma(hipStream_t s, const IgemmParams& p, int groups) {
  const int tiles = ((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  dim3 grid(tiles, p.splitk, groups);
  hipLaunchKernelGGL((igemm_dma_kernel<T, BM, BN, BKB, WM, WN, ST, MINW, X3>), grid, dim3(64 * WM * WN), 0, s, p);
}

// The DMA kernel addresses both operands through 2 GiB buffer descriptors with 32-bit offsets and keeps the
// in-image tap mask in 32 bits; anything larger falls back to the register-staged kernel (64-bit pointers).
static bool dma_eligible(const IgemmParams& p, int es) {
  const long long frames = p.M / ((long long)p.Ho * p.Wo);
  const long long x_extent = ((frames - 1) * p.xs_n + (long long)(p.H - 1) * p.xs_h + (long long)(p.W - 1) * p.xs_w + p.Cin) * es;
  const long long w_extent = (long long)p.Cout * ((long long)p.KH * p.KW * p.Cin + (p.x2 ? p.Cin2 : 0)) * es;
  if (p.x2) {
    const long long x2_extent = ((frames - 1) * p.xs2_n + (long long)(p.Ho - 1) * p.stride2 * p.xs2_h + (long long)(p.Wo - 1) * p.stride2 * p.xs2_w + p.Cin2) * es;
    if (x2_extent >= MCG_DMA_MAX_BYTES) return false;
  }
  return x_extent > 0 && x_extent < MCG_DMA_MAX_BYTES && w_extent < MCG_DMA_MAX_BYTES && (p.nocheck || p.KH * p.KW <= 32);
}

// Tile ids of igemm_dma_kernel for Cout > 64 (bf16) -- the ones the heuristic chooses; all walk K in the same order and are
// bit-identical to each other (tests/test_gpu_kernels.py::test_every_dma_tile_is_bit_identical forces each through `tile`).
//    9 = 256x128  8 waves, 64-byte K slices, 2 stages (48 -> 64 KiB LDS, two workgroups per CU): the default
//   11 = 128x128  8 waves, 2 stages: M <= 4096 rows (decoder linears)
//   15 = 128x128  8 waves, 3 stages: few workgroups (7x7 maps)
//   12 = 256x256 16 waves, 64-byte slices, 3 stages: Cout % 256 == 0, K >= 384, second source or Cin % 64 != 0
//   14 = 256x256 16 waves, 128-byte slices, 2 stages: the deep layers (3x3s, layer3/4 conv1): fewest LDS-DMA bytes per FLOP
// What the sweeps said (profiles/r01_d_tile_sweep.md, r01_f_tile_sweep.md, r01_i_tile16.md): occupancy beats tile size until the
// grid drops to about one workgroup per CU; 256x256 wins where K is deep and N fills it.
static const int kT12Min = 320, kT9Min = 300, kT14MinK = 384;
static bool tile_ok(int tile, const IgemmParams& p, int es) {
  if (tile == 9 || tile == 11 || tile == 15 || tile == 12) return true;
  return tile == 14 && (p.Cin * es) % 128 == 0 && !p.x2;
}

template <typename T>
static int launch_typed(hipStream_t s, const IgemmParams& p, int groups, const McgCtx& ctx) {
  constexpr int ES = (int)sizeof(T);
  const bool dma = ES == 2 && !ctx.staged && dma_eligible(p, ES);
  const bool wide = (p.Cin * ES) % 128 == 0 && (!p.x2 || (p.Cin2 * ES) % 128 == 0);
  MCG_CHECK_ARG((p.Cin * ES) % 64 == 0 && (!p.x2 || (p.Cin2 * ES) % 64 == 0), "igemm: Cin=%d must be a multiple of %d elements", p.Cin, 64 / ES);
  MCG_CHECK_ARG(p.Cout % (16 / ES) == 0, "igemm: Cout=%d must be a multiple of %d", p.Cout, 16 / ES);
  int tile = 9;
  if (dma && p.Cout > 64) {
    const long long blocks9 = (long long)((p.M + 255) / 256) * ((p.Cout + 127) / 128);
    const long long Kdim = gemm_k(p);
    if (ctx.tile >= 0) {
      MCG_CHECK_ARG(tile_ok(ctx.tile, p, ES), "igemm: tile %d is not available for this problem (Cin=%d%s)", ctx.tile, p.Cin, p.x2 ? ", second source" : "");
      tile = ctx.tile;
    } else if (p.Cout % 256 == 0 && Kdim >= 384 && blocks9 >= kT12Min) {
      tile = (Kdim >= kT14MinK && tile_ok(14, p, ES)) ? 14 : 12;
    } else {
      tile = blocks9 >= kT9Min ? 9 : (p.M <= 4096 ? 11 : 15);
    }
  }
  // cfg ids (bench.py CFG_NAMES): 0..3 f32 register-staged, 4..7 bf16 register-staged, 15 = 256x64 DMA, 16 + tile = DMA tiles
  const int cfg = dma ? (p.Cout <= 64 ? 15 : 16 + tile) : (ES == 2 ? 4 : 0) + (p.Cout <= 64 ? 0 : 2) + (wide ? 1 : 0);
  ProfRec* rec = prof_begin(ctx, s, cfg, p.M, p.Cout * groups, gemm_k(p), algo_flops(p, groups), algo_bytes(p, groups, ES));
  if (dma) {
    if (p.Cout <= 64) {  // 256x64 tile, 4 waves, 2 stages; the register-capped variant (4 workgroups/CU) pays for long-K (3x3) layers
      if ((long long)p.KH * p.KW * p.Cin >= 512) launch_dma<T, 256, 64, 64, 4, 1, 2, 4>(s, p, groups);
      else launch_dma<T, 256, 64, 64, 4, 1, 2>(s, p, groups);
    }
    else if (tile == 9) launch_dma<T, 256, 128, 64, 4, 2, 2>(s, p, groups);
    else if (tile == 11) launch_dma<T, 128, 128, 64, 4, 2, 2>(s, p, groups);
    else if (tile == 15) launch_dma<T, 128, 128, 64, 4, 2, 3>(s, p, groups);
    else if (tile == 12) launch_dma<T, 256, 256, 64, 4, 4, 3, 4>(s, p, groups);
    else launch_dma<T, 256, 256, 128, 4, 4, 2, 4>(s, p, groups);
  } else if (p.Cout <= 64) {
    if (wide) launch_cfg<T, 128, 64, 128, 4, 1>(s, p, groups);
    else launch_cfg<T, 128, 64, 64, 4, 1>(s, p, groups);
  } else {
    if (wide) launch_cfg<T, 128, 128, 128, 2, 2>(s, p, groups);
    else launch_cfg<T, 128, 128, 64, 2, 2>(s, p, groups);
  }
  prof_end(rec, s);
  MCG_CHECK_LAUNCH("igemm launch");
  return MCG_OK;
}

// MCG_F16X3: f32 activations x split-packed fp16 weights, three fp16 MFMAs per product (igemm_dma.hpp, X3 mode).
//   50 = 256x256 8 waves (64x128 wave tiles: the A split is shared by four column tiles), 2 stages, 128 KiB -- deep, wide layers
//   51 = 128x128 4 waves, 2 stages, 64 KiB (two workgroups per CU) -- Cout < 256, few rows (7x7 maps, decoder linears)
//   52 = 256x64 4 waves -- Cout <= 64
//   53 = 128x128 with EIGHT waves (32 x 64 wave tiles) and a four-stage ring (128 KiB) -- grids of at most one workgroup per CU with a
//        deep K (a single clip: M = 1372 rows in layer3, 343 in layer4; K = 1024 .. 4608).  There a K step of tile 51 is one wave per
//        SIMD doing its DMA wait, its A split and its MFMAs in sequence (0.9 us per 32 channels against 0.35 us of MFMA time); two waves
//        per SIMD overlap them: 0.070 -> 0.051 ms on layer3's 3x3, single-clip contraction time 2.76 -> 2.27 ms (the deeper ring alone:
//        2.66; sixteen waves: 2.24).  Same K order: bit-identical to 50 / 51 (tests/test_gpu_kernels.py::test_every_x3_tile_is_bit_identical),
//        so a clip's result does not depend on the batch it came in.
// "at most one workgroup per CU": the CU count of the current device, queried once per device
static int x3_deep_ring_max_wgs() {
  static int cus_of[MCG_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MCG_MAX_DEVICES) dev = 0;
  if (!cus_of[dev]) {
    hipDeviceProp_t prop;
    cus_of[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  return cus_of[dev];
}
static int launch_x3(hipStream_t s, const IgemmParams& p, int groups, const McgCtx& ctx) {
  MCG_CHECK_ARG(p.Cin % 32 == 0 && (!p.x2 || p.Cin2 % 32 == 0), "igemm (f16x3): Cin=%d must be a multiple of 32", p.Cin);
  MCG_CHECK_ARG(p.Cout % 4 == 0, "igemm (f16x3): Cout=%d must be a multiple of 4", p.Cout);
  if (!dma_eligible(p, 4)) {
    mcg_set_error("igemm (f16x3): operand beyond the 2 GiB descriptor window or more than 32 taps (M=%d Cin=%d %dx%d)", p.M, p.Cin, p.KH, p.KW);
    return MCG_ERR_UNSUPPORTED;
  }
  const long long t256 = (long long)((p.M + 255) / 256) * ((p.Cout + 255) / 256);
  int tile = p.Cout <= 64 ? 52 : (p.Cout % 256 == 0 && t256 >= 200 ? 50 : 51);
  const long long t128 = (long long)((p.M + 127) / 128) * ((p.Cout + 127) / 128) * groups;
  if (tile == 51 && t128 <= x3_deep_ring_max_wgs() && gemm_k(p) >= 512) tile = 53;
  if ((ctx.tile == 50 || ctx.tile == 51 || ctx.tile == 53) && p.Cout > 64) tile = ctx.tile;
  ProfRec* rec = prof_begin(ctx, s, tile, p.M, p.Cout * groups, gemm_k(p), algo_flops(p, groups), algo_bytes(p, groups, 4));
  if (tile == 50) launch_dma<float, 256, 256, 128, 4, 2, 2, 2, 1>(s, p, groups);
  else if (tile == 51) launch_dma<float, 128, 128, 128, 2, 2, 2, 2, 1>(s, p, groups);
  else if (tile == 53) launch_dma<float, 128, 128, 128, 4, 2, 4, 1, 1>(s, p, groups);
  else launch_dma<float, 256, 64, 128, 4, 1, 2, 2, 1>(s, p, groups);
  prof_end(rec, s);
  MCG_CHECK_LAUNCH("igemm (f16x3) launch");
  return MCG_OK;
}

int launch_igemm(hipStream_t s, mcg_dtype dt, const IgemmParams& p_in, int groups, const McgCtx& ctx) {
  IgemmParams p = p_in;
  if (!(p.wscale > 0.f)) p.wscale = 1.f;
  MCG_CHECK_ARG(p.M > 0 && p.Cout > 0 && groups > 0, "igemm: empty problem (M=%d Cout=%d groups=%d)", p.M, p.Cout, groups);
  MCG_CHECK_ARG(dt == MCG_F16X3 || p.wscale == 1.f, "igemm: wscale is an MCG_F16X3 operand (got %g)", (double)p.wscale);
  if (dt == MCG_F16X3) return launch_x3(s, p, groups, ctx);
  return dt == MCG_BF16 ? launch_typed<bf16_t>(s, p, groups, ctx) : (dt == MCG_F16 ? launch_typed<f16_t>(s, p, groups, ctx) : launch_typed<float>(s, p, groups, ctx));
}

static IgemmParams linear_params(const void* x, long long lda, const void* w, int M, int K, int Cout) {
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w;
  p.M = M; p.Ho = 1; p.Wo = 1; p.H = 1; p.W = 1; p.Cin = K; p.KH = 1; p.KW = 1; p.stride = 1; p.pad = 0; p.Cout = Cout;
  p.xs_n = lda; p.xs_h = 0; p.xs_w = 0; p.nocheck = 1;
  p.splitk = 1; p.tiles_per_slice = 1 << 30;
  return p;
}

int launch_linear(hipStream_t s, mcg_dtype dt, const void* x, long long lda, const void* w, const float* bias,
                  const void* res, long long ldres, void* y, long long ldy, int M, int K, int Cout, int relu, const McgCtx& ctx) {
  IgemmParams p = linear_params(x, lda, w, M, K, Cout);
  p.bias = bias; p.res = res; p.y = y;
  p.y_row_stride = ldy; p.res_row_stride = ldres;
  p.relu = relu; p.res_mode = res ? MCG_RES_ADD : MCG_RES_NONE;
  return launch_igemm(s, dt, p, 1, ctx);
}

int launch_linear_splitk(hipStream_t s, mcg_dtype dt, const void* x, long long lda, const void* w, float* partial,
                         int M, int K, int Cout, int want_slices, int* splitk_out, const McgCtx& ctx) {
  IgemmParams p = linear_params(x, lda, w, M, K, Cout);
  const int es = mcg_is16(dt) ? 2 : 4;
  const bool dma = mcg_is16(dt) && !ctx.staged;
  // K elements per K-tile of the kernel that will run: 32 (f16x3), 64-byte slices (bf16 DMA), 128- or 64-byte slices (register-staged)
  const int bk = dt == MCG_F16X3 ? 32 : (dma ? 64 : (((long long)K * es) % 128 == 0 ? 128 : 64)) / es;
  const int KT = K / bk;
  int slices = want_slices < 1 ? 1 : (want_slices > KT ? KT : want_slices);
  const int per = (KT + slices - 1) / slices;
  slices = (KT + per - 1) / per;
  p.partial = partial;
  p.splitk = slices; p.tiles_per_slice = per;
  if (slices == 1) p.splitk = 2;  // force the slab path; slice 1 is empty and writes zeros
  *splitk_out = p.splitk;
  return launch_igemm(s, dt, p, 1, ctx);
}

int conv2d_ctx(hipStream_t s, mcg_dtype dt, const mcg_conv_desc* d, const McgCtx& ctx) {
  MCG_CHECK_ARG(d && d->x && d->w && d->y, "mcg_conv2d: null pointer");
  MCG_CHECK_ARG(d->stride >= 1 && d->KH >= 1 && d->KW >= 1, "mcg_conv2d: bad geometry");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = d->x; p.w = d->w; p.bias = d->bias; p.res = d->residual; p.y = d->y;
  p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.Cout = d->Cout;
  p.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
  p.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
  MCG_CHECK_ARG(p.Ho > 0 && p.Wo > 0, "mcg_conv2d: empty output");
  p.M = d->N * p.Ho * p.Wo;
  p.xs_w = d->Cin; p.xs_h = (long long)d->W * d->Cin; p.xs_n = (long long)d->H * d->W * d->Cin;
  p.y_row_stride = d->Cout; p.res_row_stride = d->Cout;
  p.relu = d->relu; p.res_mode = d->residual ? d->residual_mode : MCG_RES_NONE;
  if (p.res_mode == MCG_RES_UPSAMPLE_ADD) {
    MCG_CHECK_ARG(d->Hr > 0 && d->Wr > 0, "mcg_conv2d: UPSAMPLE_ADD needs Hr, Wr");
    p.Hr = d->Hr; p.Wr = d->Wr;
    p.rscale_h = (float)d->Hr / (float)p.Ho;  // torch nearest: src = min(floor(dst * in/out), in-1)
    p.rscale_w = (float)d->Wr / (float)p.Wo;
  }
  p.nocheck = d->pad == 0 ? 1 : 0;  // without padding every tap of every output pixel is inside the image
  if (d->x2) {
    MCG_CHECK_ARG(d->KH == 1 && d->KW == 1 && d->pad == 0 && d->stride == 1, "mcg_conv2d: a second source needs a 1x1 / stride 1 / pad 0 primary conv");
    MCG_CHECK_ARG(d->Cin2 > 0 && d->stride2 >= 1 && d->H2 >= (p.Ho - 1) * d->stride2 + 1 && d->W2 >= (p.Wo - 1) * d->stride2 + 1, "mcg_conv2d: second source geometry");
    p.x2 = d->x2; p.Cin2 = d->Cin2; p.stride2 = d->stride2;
    p.xs2_w = d->Cin2; p.xs2_h = (long long)d->W2 * d->Cin2; p.xs2_n = (long long)d->H2 * d->W2 * d->Cin2;
  }
  p.splitk = 1; p.tiles_per_slice = 1 << 30;
  p.wscale = d->wscale;
  if (mcg_is16(dt) && d->bias && ctx.c64 && !ctx.staged && ctx.tile < 0 &&
      conv3x3_c64_applicable(d->KH, d->KW, d->stride, d->pad, d->Cin, d->Cout, p.res_mode != MCG_RES_NONE, d->x2 != nullptr)) {
    // layer1's conv2: window staged once, nine taps by address (conv3x3_c64.hpp); bit-identical to the generic kernel
    ProfRec* rec = prof_begin(ctx, s, 40, p.M, 64, 576, 2.0 * p.M * 64 * 576, algo_bytes(p, 1, 2));
    const int rc = launch_conv3x3_c64(s, d->x, d->w, d->bias, d->y, d->N, d->H, d->W, d->relu, dt == MCG_F16);
    prof_end(rec, s);
    if (rc) { mcg_set_error("conv3x3_c64 launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  return launch_igemm(s, dt, p, 1, ctx);
}
extern "C" int mcg_conv2d(mcg_stream s, mcg_dtype dt, const mcg_conv_desc* d) {
  MCG_CHECK_ARG(d, "mcg_conv2d: null descriptor");
  return conv2d_ctx((hipStream_t)s, dt, d, McgCtx::from_flags(d->tile, d->flags));
}

// ------------------------------------------------------------------------------------------------
// Stem.  The 7x7/s2 conv over 3 channels is run by the same implicit-GEMM kernel: the NCHW f32
// frame is first repacked to a zero-bordered NHWC4 image [N][H+6][W+8][4]; one "tap" is then a
// whole kernel row = 8 pixels x 4 channels = 32 contiguous elements (kw=7 and c=3 carry zero
// weights), so K = 7 taps x 32 and no bounds checks are needed.
template <typename T>
__global__ void stem_pack_kernel(const float* __restrict__ img, T* __restrict__ dst, int N, int H, int W, int Hp, int Wp) {
  const long long total = (long long)N * Hp * Wp;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int wp = (int)(i % Wp);
    const long long t = i / Wp;
    const int hp = (int)(t % Hp), n = (int)(t / Hp);
    const int h = hp - 3, w = wp - 3;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
      const float* src = img + ((long long)n * 3 * H + h) * W + w;
      v0 = src[0]; v1 = src[(long long)H * W]; v2 = src[2LL * H * W];
    }
    T* o = dst + i * 4;
    Elem<T>::st(o, v0); Elem<T>::st(o + 1, v1); Elem<T>::st(o + 2, v2); Elem<T>::st(o + 3, 0.f);
  }
}

// max-pool 3x3 stride 2 pad 1 over NHWC (resnet.py:611); one thread per 16-byte channel chunk.
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
  constexpr int EPC = Elem<T>::kPerChunk;
  const int cchunks = C / EPC;
  const long long total = (long long)N * Ho * Wo * cchunks;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % cchunks);
    long long t = i / cchunks;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float m[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) m[e] = -INFINITY;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int h = ho * 2 - 1 + dy;
      if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int w = wo * 2 - 1 + dx;
        if ((unsigned)w >= (unsigned)W) continue;
        float v[EPC];
        chunk_to_f32(*(const uint4*)(x + (((long long)n * H + h) * W + w) * C + cc * EPC), v, (T*)nullptr);
#pragma unroll
        for (int e = 0; e < EPC; ++e) m[e] = fmaxf(m[e], v[e]);
      }
    }
    *(uint4*)(y + i * EPC) = f32_to_chunk(m, (T*)nullptr);
  }
}

static inline int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  return (int)(g > 256 * 16 ? 256 * 16 : (g < 1 ? 1 : g));
}

extern "C" size_t mcg_stem_workspace_bytes(mcg_dtype dt, int N, int H, int W) {
  const size_t es = mcg_is16(dt) ? 2 : 4;
  const size_t packed = (size_t)N * (H + 6) * (W + 8) * 4 * es;
  const size_t conv = (size_t)N * (H / 2) * (W / 2) * 64 * es;
  return ((packed + 255) / 256) * 256 + ((conv + 255) / 256) * 256;
}

extern "C" int mcg_stem_forward(mcg_stream s, mcg_dtype dt, const float* img, const void* w_stem, const float* bias,
                                void* y, int N, int H, int W, void* ws, size_t ws_bytes, int flags) {
  return stem_forward_ctx((hipStream_t)s, dt, img, w_stem, bias, y, N, H, W, ws, ws_bytes, McgCtx::from_flags(0, flags));
}
int stem_forward_ctx(hipStream_t s, mcg_dtype dt, const float* img, const void* w_stem, const float* bias, void* y, int N, int H, int W,
                     void* ws, size_t ws_bytes, const McgCtx& ctx) {
  MCG_CHECK_ARG(img && w_stem && y && ws, "mcg_stem_forward: null pointer");
  MCG_CHECK_ARG(H % 4 == 0 && W % 4 == 0, "mcg_stem_forward: H, W must be multiples of 4 (got %dx%d)", H, W);
  if (ws_bytes < mcg_stem_workspace_bytes(dt, N, H, W)) {
    mcg_set_error("mcg_stem_forward: workspace too small (%zu < %zu)", ws_bytes, mcg_stem_workspace_bytes(dt, N, H, W));
    return MCG_ERR_WORKSPACE;
  }
  if (mcg_is16(dt) && ctx.stem_fused) {  // one kernel, no conv-map round trip (stem_fused.hpp); bit-identical to the path below
    if (launch_stem_fused(s, img, w_stem, bias, y, N, H, W, dt == MCG_F16)) { mcg_set_error("stem_fused launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  if (dt == MCG_F16X3 && ctx.stem_fused) {  // the f16x3 form of the same kernel; bit-identical to the three launches below
    if (launch_stem_fused_x3(s, img, w_stem, bias, y, N, H, W)) { mcg_set_error("stem_fused (f16x3) launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  const size_t es = mcg_is16(dt) ? 2 : 4;
  const int Hp = H + 6, Wp = W + 8, Hc = H / 2, Wc = W / 2;
  char* packed = (char*)ws;
  char* conv = packed + (((size_t)N * Hp * Wp * 4 * es + 255) / 256) * 256;
  const long long npix = (long long)N * Hp * Wp;
  if (dt == MCG_BF16) hipLaunchKernelGGL(stem_pack_kernel<bf16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, s, img, (bf16_t*)packed, N, H, W, Hp, Wp);
  else if (dt == MCG_F16) hipLaunchKernelGGL(stem_pack_kernel<f16_t>, dim3(grid_for(npix, 256)), dim3(256), 0, s, img, (f16_t*)packed, N, H, W, Hp, Wp);
  else hipLaunchKernelGGL(stem_pack_kernel<float>, dim3(grid_for(npix, 256)), dim3(256), 0, s, img, (float*)packed, N, H, W, Hp, Wp);
  MCG_CHECK_LAUNCH("stem_pack");
  IgemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = packed; p.w = w_stem; p.bias = bias; p.y = conv;
  p.Ho = Hc; p.Wo = Wc; p.M = N * Hc * Wc; p.H = Hp; p.W = Wp; p.Cin = 32; p.KH = 7; p.KW = 1; p.stride = 2; p.pad = 0; p.Cout = 64;
  p.xs_w = 4; p.xs_h = (long long)Wp * 4; p.xs_n = (long long)Hp * Wp * 4; p.nocheck = 1;
  p.y_row_stride = 64; p.relu = 1; p.res_mode = MCG_RES_NONE; p.splitk = 1; p.tiles_per_slice = 1 << 30;
  p.algo_k = 147;  // 7*7*3 real taps; the packed K of 224 carries zero weights
  MCG_TRY(launch_igemm(s, dt, p, 1, ctx));
  const int Ho = (Hc + 2 - 3) / 2 + 1, Wo = (Wc + 2 - 3) / 2 + 1;
  const long long nchunks = (long long)N * Ho * Wo * (64 / (16 / (int)es));
  if (dt == MCG_BF16) hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, dim3(grid_for(nchunks, 256)), dim3(256), 0, s, (const bf16_t*)conv, (bf16_t*)y, N, Hc, Wc, 64, Ho, Wo);
  else if (dt == MCG_F16) hipLaunchKernelGGL(maxpool3x3s2_kernel<f16_t>, dim3(grid_for(nchunks, 256)), dim3(256), 0, s, (const f16_t*)conv, (f16_t*)y, N, Hc, Wc, 64, Ho, Wo);
  else hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for(nchunks, 256)), dim3(256), 0, s, (const float*)conv, (float*)y, N, Hc, Wc, 64, Ho, Wo);
  MCG_CHECK_LAUNCH("maxpool");
  return MCG_OK;
}

// ------------------------------------------------------------------------------------------------
// Layout conversion through an LDS tile so both sides stay coalesced.
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, pix = p0 + tx;
    tile[j][tx] = (c < C && pix < HW) ? src[((long long)n * C + c) * HW + pix] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int pix = p0 + j, c = c0 + tx;
    if (c < C && pix < HW) Elem<T>::st(dst + ((long long)n * HW + pix) * C + c, tile[tx][j]);
  }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int HW) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int j = ty; j < 32; j += 8) {
    const int pix = p0 + j, c = c0 + tx;
    tile[j][tx] = (c < C && pix < HW) ? Elem<T>::ld(src + ((long long)n * HW + pix) * C + c) : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int c = c0 + j, pix = p0 + tx;
    if (c < C && pix < HW) dst[((long long)n * C + c) * HW + pix] = tile[tx][j];
  }
}

extern "C" int mcg_nchw_to_nhwc(mcg_stream s, mcg_dtype dt, const float* src, void* dst, int N, int C, int H, int W) {
  MCG_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "mcg_nchw_to_nhwc: bad argument");
  dim3 grid((H * W + 31) / 32, (C + 31) / 32, N);
  if (dt == MCG_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, src, (bf16_t*)dst, C, H * W);
  else if (dt == MCG_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, src, (f16_t*)dst, C, H * W);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, src, (float*)dst, C, H * W);
  MCG_CHECK_LAUNCH("nchw_to_nhwc");
  return MCG_OK;
}
extern "C" int mcg_nhwc_to_nchw(mcg_stream s, mcg_dtype dt, const void* src, float* dst, int N, int C, int H, int W) {
  MCG_CHECK_ARG(src && dst && N > 0 && C > 0 && H > 0 && W > 0, "mcg_nhwc_to_nchw: bad argument");
  dim3 grid((H * W + 31) / 32, (C + 31) / 32, N);
  if (dt == MCG_BF16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)s, (const bf16_t*)src, dst, C, H * W);
  else if (dt == MCG_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16_t>, grid, dim3(256), 0, (hipStream_t)s, (const f16_t*)src, dst, C, H * W);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)s, (const float*)src, dst, C, H * W);
  MCG_CHECK_LAUNCH("nhwc_to_nchw");
  return MCG_OK;
}
