/*
 * mcgaze_hip.h -- C-ABI of libmcgaze_hip.so, the MI355X (gfx950) implementation of the
 * MCGaze per-clip forward path.
 *
 * The reference has NO in-tree native code and no FFI for this path:
 * the arithmetic is reached through torch / mmcv Python calls.  Each entry point below
 * therefore cites the reference *Python* interface (file:line, relative to the upstream
 * repo root) whose work it replaces; INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add at that call site.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless named host_*
 *   - every call enqueues work on the caller's HIP stream and returns without host sync
 *   - no allocation on the hot path: scratch comes from a caller-provided workspace
 *   - return value: MCG_OK or an MCG_ERR_* code; mcg_last_error() gives the message
 *   - activations are NHWC ("channels last"): [frame][y][x][channel]; dtype is MCG_F32
 *     (reference mode, f32 MFMA, exact f32 accumulate chains), MCG_F16X3 (parity-grade fast
 *     mode: f32 storage, split-fp16 x 3 MFMA contraction) or MCG_BF16 (throughput mode,
 *     bf16 storage and MFMA, f32 accumulate).  Bias / LayerNorm parameters / boxes are always f32.
 *   - conv / linear weights are "OHWI": [Cout][KH][KW][Cin] (K contiguous), BN folded in.
 */
#ifndef MCGAZE_HIP_H
#define MCGAZE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCG_ABI_VERSION 13

enum { MCG_OK = 0, MCG_ERR_ARG = 1, MCG_ERR_HIP = 2, MCG_ERR_UNSUPPORTED = 3, MCG_ERR_WORKSPACE = 4 };
/* MCG_F16X3: the parity-grade fast mode.  Activations, biases and every non-GEMM kernel are exactly those of MCG_F32 (4-byte
 * f32 storage); only the contraction differs: every conv / linear weight matrix is handed over SPLIT-PACKED -- per 8 consecutive
 * K elements a 16-byte chunk of fp16 HIGH parts followed by a 16-byte chunk of fp16 LOW parts (w = hi + lo, lo = f16(w - hi):
 * 22 significant bits for |w| >= 0.125; below that the low half is an fp16 SUBNORMAL with an absolute error of 2^-25, so a weight of
 * magnitude 1e-2 keeps ~18 bits and one of 1e-3 ~15 -- tests/test_gpu_kernels.py::test_f16x3_small_weights measures it; 4 bytes per
 * element like f32) -- the f32 activations are split the same way in registers (round toward zero,
 * x - hi exact), and each product runs as three fp16 MFMAs (lo.hi + hi.lo + hi.hi) with f32 accumulation.  Operands beyond
 * +-65504 saturate per half (the reference's activations are orders of magnitude below); parts below 6e-8 flush to zero.
 * Measured: 1e-5 rad on (yaw, pitch) against the reference (north_star: 1e-3), within a factor 2 of the MCG_F32 engine. */
/* MCG_F16 (round 6, ABI 13): the 16-bit throughput mode in fp16 -- activations and weights stored as fp16 (11 significant bits instead of
 * bf16's 8; range +-65504), v_mfma_f32_32x32x16_f16, f32 accumulate, f32 LayerNorm / softmax / boxes.  Every layout, kernel and workspace size
 * is MCG_BF16's with "bf16" read as "fp16".  Outside north_star's 1e-3 rad on random-weight nets like MCG_BF16, but eight times closer. */
typedef enum { MCG_F32 = 0, MCG_BF16 = 1, MCG_F16X3 = 2, MCG_F16 = 3 } mcg_dtype;
/* bytes per activation element */
#define MCG_ELEM_BYTES(dt) (((dt) == MCG_BF16 || (dt) == MCG_F16) ? 2 : 4)
typedef void* mcg_stream; /* hipStream_t */

int mcg_abi_version(void);
/* First 16 hex digits of the sha256 over the library's kernel sources + this header at build time (csrc/Makefile); profiles/ files carry
 * it so that a reader can tell which build a measurement belongs to. */
const char* mcg_build_id(void);
const char* mcg_last_error(void);
/* Fills CU count, HBM bytes and the gcnArchName of the current HIP device. */
int mcg_device_info(int* cu_count, size_t* hbm_bytes, char* arch, int arch_len);

/* ---------------------------------------------------------------- layout helpers */
/* [N,C,H,W] f32 (the tensor the reference hands to backbone(img), multiclue_gaze.py:35)
 * -> NHWC dtype. */
int mcg_nchw_to_nhwc(mcg_stream s, mcg_dtype dt, const float* src, void* dst, int N, int C, int H, int W);
/* NHWC dtype -> [N,C,H,W] f32 (for handing a pyramid level back to NCHW consumers). */
int mcg_nhwc_to_nchw(mcg_stream s, mcg_dtype dt, const void* src, float* dst, int N, int C, int H, int W);

/* ---------------------------------------------------------------- conv / linear (implicit GEMM on MFMA)
 * Replaces, with BN folded and the activation fused:
 *   conv+BN+ReLU of Bottleneck.forward            mmdet/models/backbones/resnet.py:263-302
 *   downsample conv+BN                             mmdet/models/utils/res_layer.py:51-61
 *   FPN lateral / output convs + top-down add      mmdet/models/necks/fpn.py:157-180
 *   every nn.Linear of the decoder (H=W=1)         gaze_stqi_head.py:151-201, transformer.py:1131-1162
 */
enum { MCG_RES_NONE = 0, MCG_RES_ADD = 1, MCG_RES_UPSAMPLE_ADD = 2 };
/* Variant switches of the stand-alone operator entry points (an engine takes the same through mcg_engine_set_option).  The library
 * reads no environment variable; the default (0) is the product path.  Every variant computes the same function -- the bf16 ones
 * bit-identically (tests/test_gpu_kernels.py) -- they exist for A/B measurements and for those tests. */
enum {
  MCG_FLAG_STAGED_GEMM = 1,     /* bf16: register-staged contraction kernel instead of the LDS-DMA one */
  MCG_FLAG_NO_SPECIALISED = 2,  /* generic launch sequences instead of conv3x3_c64 / the fused stem / the decoder row-block chains */
  MCG_FLAG_NO_ATTN_BLOCK = 4    /* MCG_F16X3 mcg_stage_forward: the attention passes as in_proj / attention core / out_proj + LayerNorm launches instead of
                                 * ONE attention-block launch (attn_block_x3.hpp); same bits.  The block reads MCG_SW_IN_PROJ_WF (ABI 13) */
};
typedef struct {
  const void* x;        /* NHWC [N,H,W,Cin]                                         */
  const void* w;        /* OHWI [Cout,KH,KW,Cin]                                    */
  const float* bias;    /* [Cout] or NULL                                           */
  const void* residual; /* MCG_RES_ADD: [N,Ho,Wo,Cout]; UPSAMPLE_ADD: [N,Hr,Wr,Cout] */
  void* y;              /* NHWC [N,Ho,Wo,Cout]                                      */
  int N, H, W, Cin, Cout, KH, KW, stride, pad;
  int relu;             /* 1: y = max(y, 0) after bias and residual                 */
  int residual_mode;    /* MCG_RES_*                                                */
  int Hr, Wr;           /* residual spatial size for UPSAMPLE_ADD (nearest, F.interpolate size=) */
  /* Optional second input, K-concatenated after the first (needs KH=KW=1, stride 1, pad 0):
   *   y = [x | x2 sampled at stride2] . [w1 | w2]^T,  w = [Cout][Cin + Cin2].
   * Fuses a bottleneck's conv3 with its downsample conv (mmdet/models/utils/res_layer.py:51-61,
   * resnet.py:289-298): x2 = the block input, w2 = the BN-folded downsample weight. */
  const void* x2;       /* NHWC [N,H2,W2,Cin2] or NULL */
  int Cin2, stride2, H2, W2;
  int tile;             /* 0 = heuristic; else force this tile id of the contraction kernel (igemm.hip: 9, 11, 12, 14, 15; f16x3: 50, 51, 53) */
  int flags;            /* MCG_FLAG_* */
  float wscale;         /* MCG_F16X3 only, 0 = 1: y = wscale * (x . w) + bias ...  A power of two: the caller packs w PRE-SCALED by 1 / wscale so that
                           max |w| sits in (2^13, 2^14] and every fp16 low half down to 2^-17 of the largest weight is a NORMAL number (22 significant
                           bits for weights of any magnitude; unscaled, a weight of 1e-3 keeps 15).  mcgaze_amd/packing.py::pow2_prescale, then split_pack. */
} mcg_conv_desc;
int mcg_conv2d(mcg_stream s, mcg_dtype dt, const mcg_conv_desc* d);

/* The fused bottleneck tail as a stand-alone operator (MCG_F16X3 arithmetic; resnet.py:263-302): x = conv2's input
 * [frames][H][W][64] f32; src2 = the residual [frames][H][W][256] (nsrc = 1) or the downsample conv's input [frames][H][W][64]
 * (nsrc = 2); y [..][4 cm]; z [..][cn] (cn = 0: not written); cm = 64 or 128 mid channels (x then has cm channels, the residual 4 cm).
 * trace: NULL, or (measurement aid) a device buffer of 4096 uint64 that workgroup 0 fills with shader-clock stamps of its phases.
 * See* mcg_fused_block for wstream / bias. */
int mcg_bottleneck_x3(mcg_stream s, const float* x, const float* src2, const void* wstream, const float* bias, float* y, float* z,
                      int frames, int H, int W, int cm, int nsrc, int cn, void* trace);

/* 3x3 / stride 1 / pad 1 convolution as a ONE-DIMENSIONAL Winograd F(2,3) contraction along x (MCG_F16X3 arithmetic; wino_x3.hpp): the
 * FPN output convs (mmdet/models/necks/fpn.py:178-180) and a bottleneck's stride-1 conv2 (mmdet/models/backbones/resnet.py:263-302,
 * layer3 / layer4) with 6 instead of 9 matrix products per output.  x [frames][H][W][Cin] f32, y [frames][H][W][Cout] f32 = conv + bias
 * (+ ReLU); u = the weight in the kernel's transformed, split, fragment-major layout (mcgaze_amd/packing.py::wino_pack:
 * fp16 [Cout / 128][3 Cin / 16 K steps (16-channel slice major, y tap minor)][position 4][32-channel tile 4][high, low][lane 64][8],
 * mcg_conv3x3_wino_x3_weight_bytes(Cin, Cout) bytes = 16 / 9 of the f32 OHWI tensor).  Needs Cin % 32 == 0, Cout % 128 == 0,
 * 2 <= W <= 62 and a 128-pair tile's input window (its rows + halo rows, 16 channels) within 48 KiB -- every level of a 224 x 224
 * input; MCG_ERR_UNSUPPORTED otherwise (use mcg_conv2d).  The result differs from mcg_conv2d's in rounding only (both within 1e-6 of
 * scale of the f64 convolution); it does not depend on how the frames are batched.  tile: 0 = chosen by grid size; 1 / 2 / 3 force the
 * 128 x 128 / 64 x 64 / 32 x 64 (pairs x channels) workgroup tile of the 8-wave kernel, 4 the 128 x 128 tile with one wave per SIMD and the
 * weight fragments read straight from global memory (g = 2 only; what tile 0 picks for grids of >= 130 workgroups) -- all give the same bits.  wscale: as mcg_conv_desc.wscale (u packed
 * pre-scaled by its inverse; 0 = 1).  g: output pixels per transform group -- 2 = F(2,3) (the description above), 4 = F(4,3): six positions for
 * four outputs, 4.5 products per output; u = wino_pack(w, g=4) (24 / 9 of the OHWI tensor); additionally needs W % 4 == 0 and W >= 16; tile 1 = 64
 * groups x 128 channels, 2 / 3 = 32 x 64.  Its transform constants cost about 1.5 bits against the direct kernel (measured: tests). */
size_t mcg_conv3x3_wino_x3_weight_bytes(int Cin, int Cout, int g);
int mcg_conv3x3_wino_x3(mcg_stream s, const float* x, const void* u, const float* bias, float* y, int frames, int H, int W,
                        int Cin, int Cout, int relu, int tile, float wscale, int g);

/* Stem: conv 7x7 s2 p3 (3->64) + BN + ReLU then max-pool 3x3 s2 p1 (resnet.py:636-639).
 * img is the reference's NCHW f32 frame tensor.  w_stem is the packed stem weight
 * [64][7][8][4] (kw and channel zero-padded, BN folded).  ws >= mcg_stem_workspace_bytes. */
size_t mcg_stem_workspace_bytes(mcg_dtype dt, int N, int H, int W);
int mcg_stem_forward(mcg_stream s, mcg_dtype dt, const float* img, const void* w_stem, const float* bias,
                     void* y, int N, int H, int W, void* ws, size_t ws_bytes, int flags);

/* ---------------------------------------------------------------- RoIAlign, all levels, one launch
 * Replaces SingleRoIExtractor.forward (roi_extractors/single_level_roi_extractor.py:57-115:
 * map_roi_levels + per-level mmcv.ops.RoIAlign(7, 1/stride, sampling_ratio=2, 'avg', aligned=True))
 * and bbox2roi (mmdet/core/bbox/transforms.py:75-94).
 * boxes [N*P,4] f32 xyxy in image pixels, P boxes per frame, frame index = row / P.
 * out is [N*P][49][C] dtype (position-major, channel-minor: the layout DynamicConv consumes,
 * transformer.py:1131-1133).  levels_out (optional) receives the chosen pyramid level per box. */
int mcg_roi_align(mcg_stream s, mcg_dtype dt, const void* const feats[4], const int feat_h[4], const int feat_w[4],
                  const int strides[4], int C, const float* boxes, int num_boxes, int boxes_per_frame,
                  void* out, int32_t* levels_out);

/* ---------------------------------------------------------------- decoder stage / gaze head / whole path
 * Weight tables are arrays of device pointers indexed by the enums below.  Matrices are
 * dtype, [out][in] row-major exactly as in the checkpoint unless noted; vectors are f32.
 */
enum {
  MCG_SW_IN_PROJ_W = 0, MCG_SW_IN_PROJ_B, MCG_SW_OUT_PROJ_W, MCG_SW_OUT_PROJ_B, MCG_SW_ATTN_LN_G, MCG_SW_ATTN_LN_B,
  MCG_SW_DYN_W,      /* dynamic_layer.weight rows permuted (mcgaze_amd/packing.py::dyn_permutation(epc = elements per 16-byte chunk)) so that a token's
                        param_in^T [64][256] and param_out^T [256][64] come out in the MFMA-fragment-major order dynconv_kernel reads */
  MCG_SW_DYN_B,      /* permuted the same way, f32 */
  MCG_SW_NORM_IN_G, MCG_SW_NORM_IN_B, MCG_SW_NORM_OUT_G, MCG_SW_NORM_OUT_B,
  MCG_SW_FC_W, MCG_SW_FC_B, MCG_SW_FC_LN_G, MCG_SW_FC_LN_B, MCG_SW_IIC_LN_G, MCG_SW_IIC_LN_B,
  MCG_SW_FFN1_W, MCG_SW_FFN1_B, MCG_SW_FFN2_W, MCG_SW_FFN2_B, MCG_SW_FFN_LN_G, MCG_SW_FFN_LN_B,
  MCG_SW_CLS_FC_W, MCG_SW_CLS_LN_G, MCG_SW_CLS_LN_B,
  MCG_SW_REG_FC_W,   /* [3][256][256] */
  MCG_SW_REG_LN_G,   /* [3][256] */
  MCG_SW_REG_LN_B,
  MCG_SW_HEAD_CLS_W, /* f32 [3 clues][256]      (face, eyes, head)_fc_cls.weight */
  MCG_SW_HEAD_CLS_B, /* f32 [3]                                                  */
  MCG_SW_HEAD_REG_W, /* f32 [3 clues][4][256]   (face, eyes, head)_fc_reg.weight */
  MCG_SW_HEAD_REG_B, /* f32 [3][4]                                               */
  /* MFMA-fragment-major copies of [32 t][256] matrices for the fused row-block chain / attention block.
   *   MCG_BF16:  bf16 WF[t][ks][lane][e] = W[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + e], t < rows / 32, ks < 16, lane < 64, e < 8 --
   *              one wave-wide 16-byte load per (column tile, K-step) is then 1 KiB contiguous.
   *   MCG_F16X3: REQUIRED for OUT_PROJ / CLS_FC / REG_FC / DYN and -- since ABI 13 -- IN_PROJ (the default stage path -- chain_x3.hpp,
   *              attn_block_x3.hpp, pw_single_x3.hpp -- reads them;
   *              passing the row-major split-packed pointer again gives WRONG results, not an error): the SPLIT fragment-major form
   *              fp16 WF[t][ks][hl][lane][e], hl = 0 the fp16 high parts, hl = 1 the low parts of the same elements as above
   *              (mcgaze_amd/packing.py::frag_major_split; 4 bytes per weight).  IN_PROJ_WF ([768][256], t < 24) is read by the attention block whenever
   *              3 x clip_length <= 32 (ABI <= 12: not read); flags = MCG_FLAG_NO_ATTN_BLOCK keeps the launch sequence that does not need it.
   *              mcg_stage_forward(flags = MCG_FLAG_NO_SPECIALISED) runs the generic launch sequence, which reads only the row-major
   *              split-packed matrices.
   *   MCG_F32:   ignored (may be given the row-major pointers again). */
  MCG_SW_OUT_PROJ_WF, MCG_SW_CLS_FC_WF,
  MCG_SW_REG_FC_WF,  /* [3] x fragment-major */
  MCG_SW_IN_PROJ_WF, /* in_proj_weight [768][256] fragment-major (t < 24 column tiles) for the fused attention block (attn_block.hpp) */
  MCG_SW_DYN_WF,     /* dynamic_layer.weight [32768][256] (rows permuted like the row-major entry) fragment-major, t < 1024 (pw_single.hpp) */
  MCG_SW_COUNT
};
enum {
  MCG_GW_FC_W = 0,   /* [6 branches][2 layers][256][256]; branch = 3*k + clue, k = 0 gaze MLP, 1 confidence MLP */
  MCG_GW_LN_G,       /* f32 [6][2][256] */
  MCG_GW_LN_B,
  MCG_GW_OUT_W,      /* f32 [6][3][256]: fc_{clue} (gaze) / fc_{clue}_confidence */
  MCG_GW_OUT_B,      /* f32 [6][3] */
  MCG_GW_FUSE_W,     /* f32 [3][9]  fc_gaze */
  MCG_GW_FUSE_B,     /* f32 [3] */
  MCG_GW_COUNT
};

/* One GazeSTQIHead.forward + refine_bboxes (gaze_stqi_head.py:119-202, bbox_head.py:380-457,
 * delta_xywh_bbox_coder.py:224-260 with stds (.5,.5,1,1), clip_border=False).
 *   roi_feat [R][49][256] dtype (from mcg_roi_align), obj_in/obj_out [N][3][256] dtype,
 *   boxes_in/boxes_out [N][3][4] f32, cls_out [N][3] f32 (logits, pre-sigmoid).
 *   N = num_clips * clip_length frames; temporal attention spans clip_length frames. */
size_t mcg_stage_workspace_bytes(mcg_dtype dt, int num_frames);
int mcg_stage_forward(mcg_stream s, mcg_dtype dt, const void* const weights[MCG_SW_COUNT], const void* roi_feat,
                      const void* obj_in, const float* boxes_in, int num_frames, int clip_length,
                      void* obj_out, float* boxes_out, float* cls_out, const float bbox_stds[4],
                      void* ws, size_t ws_bytes, int flags);

/* GazeHead.forward (mask_heads/gaze_head.py:138-202): obj [N][3][256] dtype -> gaze [4][N][3] f32
 * unit vectors in the order fused, face, eyes, head. */
size_t mcg_gaze_head_workspace_bytes(mcg_dtype dt, int num_frames);
int mcg_gaze_head(mcg_stream s, mcg_dtype dt, const void* const weights[MCG_GW_COUNT], const void* obj,
                  int num_frames, float* gaze_out, void* ws, size_t ws_bytes);

/* ---------------------------------------------------------------- engine: the whole path in one call
 * Replaces MultiClueGaze.simple_test (mmdet/models/detectors/multiclue_gaze.py:105-131) with
 * the batched semantics of forward_train (:77-78): N = num_clips*clip_length frames.
 */
typedef struct {
  const void* w;      /* OHWI dtype, BN folded */
  const float* bias;  /* f32 [Cout] */
  int cin, cout, k, stride, pad;
  const void* wf;     /* optional second copy of w for a specialised kernel; NULL -> the contraction kernel on w (always correct):
                         MCG_BF16, 1x1 convs: MFMA-fragment-major bf16 [cout/32][cin/16][64 lanes][8],
                           wf[t][ks][lane][e] = w[32 t + (lane & 31)][16 ks + 8 (lane >> 5) + e] (pw_pair.hpp, pw_single.hpp);
                         MCG_F16X3, 1x1 convs 256 -> 256 / 256 -> 1024: the SPLIT fragment-major form fp16 [cout/32][cin/16][high, low][64][8]
                           (packing.py::frag_major_split; pw_single_x3.hpp).  A row-major pointer here gives wrong results;
                         MCG_F16X3, 3x3 / stride 1 / pad 1 convs with cin % 32 == 0, cout % 128 == 0: the Winograd F(2,3) operand of
                           mcg_conv3x3_wino_x3 (packing.py::wino_pack; wino_x3.hpp);
                         MCG_F32: ignored */
  float wscale;       /* MCG_F16X3: the power of two w AND wf were pre-scaled by the inverse of (mcg_conv_desc.wscale); 0 = 1 = unscaled */
  const void* wf4;    /* optional, MCG_F16X3, 3x3 / stride 1 / pad 1 convs: the F(4,3) operand of mcg_conv3x3_wino_x3 (wino_pack(w, g=4), same pre-scale);
                         used on maps whose width is a multiple of 4 and at least 16, wf (F(2,3)) elsewhere; NULL -> wf only */
} mcg_conv_weights;

/* A fused bottleneck tail of the MCG_F16X3 engine (bneck_x3.hpp): conv2 (3x3) -> conv3 (1x1, + downsample as a second K source or
 * + residual) -> the NEXT block's conv1 (1x1) in one kernel; the 64-channel intermediates never leave the CU and the block output is
 * read from HBM once less.  wstream: the three weight matrices as 16 KiB MFMA-fragment-major slabs of fp16 high / low parts in
 * the order the kernel consumes them (mcgaze_amd/packing.py::bneck_stream gives the exact layout; bytes =
 * 16384 * (9 (cm / 64)^2 + (cm / 16) (cm / 64 + nsrc - 1 + cn / 64))).  bias: f32 [cm | c | cn | 4]: the three bias vectors, then the
 * power-of-two descale factors of conv2's, conv3's (+ downsample) and the next conv1's matrix (each packed pre-scaled by the inverse,
 * see mcg_conv_desc.wscale) and one pad float.  Applies when cm = 64, c = 256, cn in
 * {0, 64, 128} (layer1 of a ResNet-50) or cm = 128, c = 512, cn in {0, 128}, nsrc = 1 (layer2's identity blocks); other layers keep
 * the layer-granular launches. */
typedef struct {
  const void* wstream;
  const float* bias;
  int conv2_index;                  /* index into convs[] of the conv2 this unit replaces (its conv3 [+ downsample] follow) */
  int cm, c, cn;                    /* mid channels, output channels, output channels of the next block's conv1 (0 = none) */
  int nsrc;                         /* 1: identity block (residual = block input); 2: first block (conv3 | downsample, no residual) */
} mcg_fused_block;

typedef struct {
  int blocks[4];                    /* bottlenecks per layer, (3,4,6,3) for R-50 */
  mcg_conv_weights stem;            /* packed stem, see mcg_stem_forward */
  const mcg_conv_weights* convs;    /* host array, execution order: per block conv1, conv2, conv3[, downsample] */
  int num_convs;
  mcg_conv_weights lateral[4];
  mcg_conv_weights fpn_out[4];
  mcg_conv_weights c3_ds[4];        /* per layer: first block's conv3 and downsample fused ([Cout][planes + inplanes], bias summed); w = NULL -> unfused */
  const float* init_boxes;          /* f32 [3][4] normalised cxcywh (rpn_head.init_proposal_bboxes.weight) */
  const void* init_feats;           /* dtype [3][256] */
  int num_stages;
  const void* const* stage_weights; /* host array [num_stages][MCG_SW_COUNT] of device pointers */
  const void* const* gaze_weights;  /* host array [MCG_GW_COUNT]: the LAST stage's gaze head */
  float bbox_stds[4];
  const mcg_fused_block* fused;     /* optional (MCG_F16X3): host array of fused bottleneck tails, or NULL */
  int num_fused;
} mcg_model_weights;

typedef struct mcg_engine mcg_engine;
/* Threading and streams (what a host that drives the library from several threads may rely on)
 *   - The stand-alone operators (mcg_conv2d ... mcg_gaze_head, mcg_preprocess_frames) keep no state: any thread, any stream.
 *     mcg_last_error() is thread-local.
 *   - An ENGINE runs ONE forward at a time.  mcg_backbone_fpn_forward / mcg_clip_forward / mcg_bench_backbone_forward /
 *     mcg_engine_set_option / mcg_engine_profile_* take the engine's mutex for the duration of the ENQUEUE (they never wait for
 *     the GPU): two host threads calling into the same engine are serialised silently, in lock order, and both calls are
 *     correct -- but they share the engine's fork / join events and the caller-provided workspace, so they must not pass the
 *     same workspace unless they also enqueue on the same stream.  mcg_decoder_forward touches no engine state beyond the
 *     (immutable) weight tables and does not take the mutex: it may run on another thread / stream beside the trunk of the
 *     next batch (mcgaze_amd/engine.py: PipelinedRunner) as long as its workspace and pyramid are its own.
 *     For concurrent forwards on one device use one engine per thread (weights may be shared: the engine copies only the
 *     tables); tests/test_gpu_forward.py::test_two_threads_two_engines_one_device asserts bit-identical results.
 *   - Streams: every call is ordered on the caller's stream `s` -- work queued on `s` before the call is visible to it, work
 *     queued on `s` after the call sees its results.  With trunk_streams > 1 the trunk forks frame ranges onto side streams
 *     and joins them back into `s` with events before returning, so this still holds; the side streams come from a
 *     per-DEVICE pool of 8 non-blocking streams created once per process and shared by every engine on that device (streams
 *     created later in a process's life serialise against earlier ones on this runtime), chosen per caller stream by a
 *     one-off concurrency probe (a ~1 ms host wait on the first call that sees a new caller stream; never on the hot path
 *     afterwards).  Two engines driven concurrently on one device therefore share side streams: results are unaffected,
 *     their trunks' frame ranges may serialise against each other.
 *   - The library reads no environment variable and never synchronises the device on the hot path; mcg_engine_profile_stop,
 *     mcg_engine_range_audit (debug) and the first-call probe are the only host waits. */
/* The engine copies the weight TABLES (not the weights); device buffers stay caller-owned. */
int mcg_engine_create(mcg_engine** out, const mcg_model_weights* w, mcg_dtype dt);
void mcg_engine_destroy(mcg_engine* e);
/* Per-engine options (integers; unknown names are an error).  An engine is driven by one host thread at a time.
 *   trunk_streams     1..4  concurrent frame ranges of the trunk (default 2: one range's kernel tails overlap the other's kernels)
 *   max_range_frames  >= 0  lowers the frames-per-range cap (0 = what fits the 2 GiB descriptor window)
 *   tile              forces a contraction tile id (0 = heuristic), see mcg_conv_desc.tile
 *   staged_gemm, conv3x3_c64, stem_fused, decoder_chain   0/1 kernel-variant switches (defaults 0, 1, 1, 1)
 *   decoder_attn_block 0/1 f16x3: both attention passes of a decoder stage (in_proj, attention core, out_proj + residual + LayerNorm, twice) as ONE
 *                     launch per stage, one clip per workgroup (attn_block_x3.hpp; clips of at most 10 frames); bit-identical to the six launches (default 1)
 *   pointwise_pair    0/1 conv3 (+ residual) of a block and conv1 of the next as one kernel in layer1 (bf16; default 1)
 *   pointwise_stream  0/1 HBM-bound 1x1 convs (layer2, layer3 conv3, P2 / P3 laterals) by the persistent register-resident-weight
 *                     kernel pw_single.hpp (bf16; default 1)
 *   bottleneck_fused  0/1 the fused bottleneck tails handed over in mcg_model_weights.fused (f16x3; default 1)
 *   bottleneck_blocked 0/1 a tensor that only travels from one fused tail to the next (the residual inside layer1 / layer2) is kept in that
 *                     kernel's blocked layout (whole-line stores and loads) instead of [M][C]; internal workspace only, results bit-identical (default 1)
 *   winograd          0/1/2 stride-1 3x3 convs that carry a Winograd copy by wino_x3.hpp (f16x3): 0 off, 1 (default) F(2,3) (mcg_conv_weights.wf), 2 F(4,3)
 *                     on maps whose shape allows it (mcg_conv_weights.wf4; 6 % faster there, four times the operator error), F(2,3) elsewhere
 *   wino_tile         -1..3 tile of the F(2,3) kernel: -1 (default) by grid size -- the one-wave-per-SIMD tile wino_x3w_kernel on grids of >= 130
 *                     workgroups --, 0 / 1 / 2 force the 8- / 4-wave tiles of wino_x3_kernel, 3 forces wino_x3w_kernel.  Every tile gives the same
 *                     bits; 0 is the run-time fallback for wino_x3w_kernel (its hand-issued register loads need a spill-free build, which
 *                     csrc/check_resources.py enforces at build time)
 *   range_audit       0/1 DEBUG (MCG_F32 / MCG_F16X3; default 0): after every activation tensor the trunk writes, a counting kernel tallies the
 *                     values beyond the fp16 range (|x| > 65504: an f16x3 operand half would saturate there) and the non-finite ones;
 *                     read and reset with mcg_engine_range_audit.  Turning it on allocates the counters (the library's only allocation,
 *                     at option time); it costs a pass over every activation, so it is off in the product path. */
int mcg_engine_set_option(mcg_engine* e, const char* name, int value);
/* Read-out of the range_audit option: synchronises the device (a debug call), copies the counters of the first min(capacity, tensors)
 * audited tensors to the host and resets them.  counts[2 i] = values with |x| > 65504, counts[2 i + 1] = non-finite values of tensor i,
 * summed over every frame processed since the last read; names (optional, [capacity]) receives engine-owned tensor names ("stem",
 * "layer3.2.conv1", "fpn.P2", ...) valid until the option changes.  A real checkpoint whose activations leave the fp16 range shows up
 * here layer by layer instead of as a silently saturated gaze vector. */
int mcg_engine_range_audit(mcg_engine* e, unsigned long long* counts, const char** names, int capacity, int* n_out);
/* chunk_frames: the trunk runs in chunks of this many frames so that layer outputs stay
 * resident in the 256 MiB Infinity Cache (0 = all frames in one pass). */
size_t mcg_engine_workspace_bytes(const mcg_engine* e, int num_frames, int H, int W, int chunk_frames);
/* Backbone + FPN only: img NCHW f32 [N,3,H,W] -> P2..P5 NHWC dtype. */
int mcg_backbone_fpn_forward(mcg_engine* e, mcg_stream s, const float* img, int num_frames, int H, int W,
                             int chunk_frames, void* const pyramid[4], void* ws, size_t ws_bytes);
/* Decoder only: 4 x (RoIAlign + GazeSTQIHead + box refinement) + GazeHead over an existing pyramid
 * (MultiClueGazeROIHead.simple_test, multiclue_gaze_roi_head.py:287-384).  Separate from the trunk so a caller can
 * overlap the decoder of batch k with the trunk of batch k+1 on a second stream (mcgaze_amd/engine.py: PipelinedRunner).
 * img_hw: DEVICE int [num_frames][2] = img_shape (h, w) per frame, or NULL when every frame fills the padded H x W
 * (query boxes scale with img_shape, fixed_embedding_rpn_head.py:80-89).  Outputs (device, f32):
 *   gaze_out [4][N][3] (fused, face, eyes, head), boxes_out [N][3][4], scores_out [N][3] (sigmoid). */
size_t mcg_trunk_workspace_bytes(const mcg_engine* e, int num_frames, int H, int W, int chunk_frames);
size_t mcg_decoder_workspace_bytes(const mcg_engine* e, int num_frames);
int mcg_decoder_forward(mcg_engine* e, mcg_stream s, const void* const pyramid[4], int num_frames, int clip_length, int H, int W,
                        const int* img_hw, float* gaze_out, float* boxes_out, float* scores_out, void* ws, size_t ws_bytes);
/* Whole path = mcg_backbone_fpn_forward + mcg_decoder_forward on one stream; ws >= mcg_engine_workspace_bytes. */
int mcg_clip_forward(mcg_engine* e, mcg_stream s, const float* img, int num_frames, int clip_length, int H, int W,
                     const int* img_hw, int chunk_frames, float* gaze_out, float* boxes_out, float* scores_out,
                     void* ws, size_t ws_bytes);

/* ---------------------------------------------------------------- test-time preprocessing
 * One launch replaces the per-frame CPU transforms the reference's test pipeline applies between image decode and the model
 * (configs/_base_/datasets/gaze360.py:27-36): CenterCrop (mmdet/datasets/pipelines/transforms.py:1036-1052; the window is
 * chosen by the caller, who owns the RNG draw of :1126-1130) -> Resize(keep_ratio) (transforms.py:216-242, cv2 INTER_LINEAR on
 * 8-bit pixels) -> Normalize(to_rgb) (:739-755) -> Pad + batch collate (zeros right/below, :665-683) -> HWC->CHW
 * (formatting.py:229-231).  frames_dev: DEVICE array of num_frames descriptors; every src points at a decoded uint8 frame in
 * device memory, 3 interleaved channels in cv2 order (BGR), rows src_pitch bytes apart.  The crop window must lie inside the
 * frame; (out_h, out_w) is the resized size, <= (pad_h, pad_w).  dst: [num_frames][3][pad_h][pad_w] f32 = the `img` tensor
 * mcg_clip_forward takes.  mean / stdinv: host arrays, channel order of the OUTPUT (RGB when to_rgb).  to_rgb is a plain swap of
 * channels 0 and 2 on the way out: a caller whose decoder already delivers RGB passes the frames as they are and to_rgb = 0. */
typedef struct mcg_frame_desc {
  const void* src;
  int src_h, src_w, src_pitch;
  int crop_y, crop_x, crop_h, crop_w;
  int out_h, out_w;
} mcg_frame_desc;
int mcg_preprocess_frames(mcg_stream stream, const mcg_frame_desc* frames_dev, int num_frames, float* dst, int pad_h, int pad_w,
                          const float mean[3], const float stdinv[3], int to_rgb);

/* ---------------------------------------------------------------- measurement aids (bench.py)
 * While armed, every launch of the contraction kernel made by THIS engine is bracketed by a hipEvent pair on its launch stream.
 * mcg_engine_profile_stop synchronises, returns per-launch duration (ms), algorithmic FLOPs, algorithmic HBM bytes (inputs and
 * residual read once, output written once, weights once -- what a layer-granular schedule cannot go below), tile-configuration id
 * (bench.py CFG_NAMES) and the GEMM shape (M, N, K) of every recorded launch (any output array may be NULL), and disarms. */
int mcg_engine_profile_start(mcg_engine* e, int capacity);
int mcg_engine_profile_stop(mcg_engine* e, int* count, float* ms, double* flops, double* algo_bytes, int* cfg, int* shape_mnk, int capacity);
/* BASELINE.json configs[1] "R-50 backbone-only": stem + layer1..4 (C2..C5 stay in the workspace), no FPN.  Not a product entry
 * point; ws >= mcg_trunk_workspace_bytes(e, num_frames, H, W, 0). */
int mcg_bench_backbone_forward(mcg_engine* e, mcg_stream s, const float* img, int num_frames, int H, int W, void* ws, size_t ws_bytes);
/* Where that call left C2..C5 (NHWC [num_frames][H/4 >> i][W/4 >> i][256 << i], engine dtype) inside ws, for a batch that ran as ONE
 * frame range (engine option trunk_streams = 1): tests/test_gpu_forward.py::test_backbone_only_matches_the_oracle. */
int mcg_bench_backbone_levels(const mcg_engine* e, void* ws, int num_frames, int H, int W, void* levels[4]);

#ifdef __cplusplus
}
#endif
#endif /* MCGAZE_HIP_H */
