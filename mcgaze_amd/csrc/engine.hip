// Engine: enqueues the whole per-clip forward path (trunk -> query init -> 4 x [RoIAlign + decoder
// stage] -> gaze head) from a single C-ABI call, ordered on the caller's HIP stream.  No allocation, no host sync:
// every intermediate lives in the caller-provided workspace.
//
// The trunk of a large batch runs as `trunk_streams` (option, default 2) frame ranges on concurrent streams -- the caller's plus
// side streams the engine owns, forked and joined with events so the call still behaves as one operation on the caller's
// stream.  Frames are independent, so results do not change; what changes is that the last, partly filled round of
// workgroups of one range's kernel (layer3 at 14x14 has 343 output tiles for 256 CUs) overlaps with the other range's
// kernels: 448 frames 9.36 -> 8.75 ms (tools/lab/trunk_two_streams.py).
#include "igemm.hpp"
#include "pw_pair.hpp"
#include "pw_single.hpp"
#include "bneck_x3.hpp"
#include "pw_single_x3.hpp"
#include "wino_x3.hpp"

#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

int launch_roi_align(hipStream_t s, mcg_dtype dt, const void* const feats[4], const int feat_h[4], const int feat_w[4],
                     const int strides[4], int C, const float* boxes, int num_boxes, int boxes_per_frame, void* out,
                     int32_t* levels_out);
int launch_init_queries(hipStream_t s, mcg_dtype dt, const float* init_boxes, const void* init_feats, const int* img_hw, int H, int W,
                        float* boxes, void* obj, int N);

struct mcg_engine {
  mcg_dtype dt;
  int blocks[4];
  mcg_conv_weights stem;
  std::vector<mcg_conv_weights> convs;
  mcg_conv_weights lateral[4], fpn_out[4], c3_ds[4];
  std::vector<mcg_fused_block> fused;   // f16x3: fused bottleneck tails (bneck_x3.hpp), looked up by conv2 index
  bool bneck_fused = true;
  bool bneck_blocked = true;            // tensors that only travel between two fused tails use the kernel's blocked layout (bneck_x3.hpp)
  int winograd = 1;            // f16x3: stride-1 3x3 convs with a Winograd-packed weight copy run wino_x3.hpp: 0 off, 1 (default) F(2,3) (mcg_conv_weights.wf),
                               // 2 F(4,3) where the layer's shape allows it and the weights carry that copy (wf4), F(2,3) elsewhere: 6 % faster on
                               // the 56-wide maps for four times the operator error -- opt-in
  int wino_tile = -1;          // tile of the F(2,3) kernel: -1 (default) by grid size (the one-wave-per-SIMD tile on grids >= 130 workgroups), 0 / 1 / 2 = the 8- and
                               // 4-wave tiles of wino_x3_kernel forced, 3 = wino_x3w_kernel forced.  Every tile gives the same bits; 0 is the run-time
                               // fallback for wino_x3w_kernel, whose hand-issued register loads depend on a spill-free build (csrc/check_resources.py)
  // range audit (debug option, f32-storage engines): per activation tensor the trunk writes, how many values lie beyond the fp16 range
  // (|x| > 65504: an f16x3 operand half would saturate) and how many are not finite.  Counters live on the device; read by mcg_engine_range_audit.
  static constexpr int kAuditCap = 256;
  unsigned long long* audit_dev = nullptr;       // [kAuditCap][2]
  std::vector<std::string> audit_names;          // filled by the first chunk that runs with the audit on
  const float* init_boxes;
  const void* init_feats;
  int num_stages;
  std::vector<const void*> stage_w;  // [num_stages][MCG_SW_COUNT]
  const void* gaze_w[MCG_GW_COUNT];
  float stds[4];
  static constexpr int kMaxSplit = 4, kCandidates = 8;
  struct StreamPool* pool = nullptr;             // this device's side-stream candidates (shared by every engine on the device)
  hipStream_t* cand = nullptr;                   // = pool->cand
  hipEvent_t ev_fork = nullptr, ev_join[kMaxSplit - 1] = {};
  // options (mcg_engine_set_option) and observers: read at create time or set explicitly, never from the environment
  McgCtx ctx;                  // kernel-variant switches + profiling sink handed to every launcher
  Prof prof;                   // per-launch event records (mcg_engine_profile_start / _stop)
  int trunk_streams = 2;       // concurrent frame ranges of the trunk
  int max_range_frames = 0;    // 0 = what fits the 2 GiB descriptor window
  bool pw_single = true;       // HBM-bound 1x1 convs of layer2 / the P2 lateral (bf16): persistent register-resident-weight kernel (pw_single.hpp)
  bool pw_pair = true;         // layer1 / layer2 (bf16): conv3 (+ residual) and the next block's conv1 as one kernel (pw_pair.hpp)
  std::mutex mu;               // one forward at a time per engine: the fork/join events and side streams are shared state
};
// Side-stream candidates are created ONCE per device and shared by every engine of the process.  Measured (tools/lab/leg_order_probe.py):
// streams created later in a process's life land on hardware queues that serialise against the earlier ones -- a second engine
// with its own fresh streams ran its trunk 20 % slower than the first (bf16 9.7 -> 11.7 ms per 64 clips) no matter whether the
// first had been destroyed.  The pool is immutable after creation (guarded by a mutex while it is built and while the per-caller-
// stream probe result is cached); it lives until process exit.
struct StreamPool {
  hipStream_t cand[mcg_engine::kCandidates] = {};
  hipEvent_t ev_probe[3] = {};
  std::map<hipStream_t, std::vector<int>> side_of;  // caller stream -> candidates on OTHER hardware queues (probed once)
  std::mutex mu;
  bool ok = false;
};
static StreamPool* stream_pool_for_current_device() {
  static std::mutex g_mu;
  static std::map<int, StreamPool*> g_pools;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_mu);
  auto it = g_pools.find(dev);
  if (it != g_pools.end()) return it->second;
  StreamPool* p = new StreamPool();
  bool ok = true;
  // The ROCm runtime packs the streams of a priority level onto at most GPU_MAX_HW_QUEUES (4) hardware queues, and two streams on
  // one hardware queue execute in submission order INCLUDING each other's event waits: a frame range on a stream that shares a
  // queue with the caller's runs after it, not beside it (10.1 ms instead of 8.75 for 448 frames).  Which streams share a queue
  // depends on every stream the process has created, so the pool keeps several candidates and, the first time it sees a caller
  // stream, measures which of them actually run concurrently with it (side_streams_for).
  for (int i = 0; i < mcg_engine::kCandidates; ++i) ok = ok && hipStreamCreateWithFlags(&p->cand[i], hipStreamNonBlocking) == hipSuccess;
  for (int i = 0; i < 3; ++i) ok = ok && hipEventCreate(&p->ev_probe[i]) == hipSuccess;
  p->ok = ok;
  g_pools[dev] = p;
  return p;
}
// ~150 us of wall clock (s_memrealtime ticks at 100 MHz), one wave
__global__ void probe_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
}
// Candidates that run CONCURRENTLY with caller stream s, best effort and cached per stream.  A first-call cost of about a
// millisecond and a host wait (set-up, not the hot path).
static const std::vector<int>& side_streams_for(mcg_engine* e, hipStream_t s) {
  StreamPool* p = e->pool;
  std::lock_guard<std::mutex> lock(p->mu);
  auto it = p->side_of.find(s);
  if (it != p->side_of.end()) return it->second;
  std::vector<int> good, rest;
  const long long ticks = 15000;
  for (int c = 0; c < mcg_engine::kCandidates; ++c) {
    bool concurrent = false;
    if (hipEventRecord(p->ev_probe[0], s) == hipSuccess) {
      hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, s, ticks);
      hipLaunchKernelGGL(probe_spin_kernel, dim3(1), dim3(64), 0, p->cand[c], ticks);
      float ms = 0.f;
      if (hipEventRecord(p->ev_probe[1], s) == hipSuccess && hipEventRecord(p->ev_probe[2], p->cand[c]) == hipSuccess &&
          hipEventSynchronize(p->ev_probe[1]) == hipSuccess && hipEventSynchronize(p->ev_probe[2]) == hipSuccess &&
          hipEventElapsedTime(&ms, p->ev_probe[0], p->ev_probe[2]) == hipSuccess)
        concurrent = ms < 0.15f * 1.6f;  // both spins inside ~1.6 spin lengths: they overlapped
    }
    (concurrent ? good : rest).push_back(c);
  }
  (void)hipGetLastError();
  good.insert(good.end(), rest.begin(), rest.end());  // fall back to serialised candidates rather than fail
  return p->side_of.emplace(s, good).first->second;
}
// Frames per launch sequence are capped so that the largest activation of a range ([n, H/4, W/4, 256]) stays inside the 2 GiB
// window of the contraction kernel's buffer descriptors (beyond it every conv would fall back to the slower register-staged
// kernel): 1337 frames at 224x224 bf16.  The `max_range_frames` option lowers the cap (tests).
static int range_frame_cap(const mcg_engine* e, int H, int W) {
  const long long per_frame = (long long)(H / 4) * (W / 4) * 256 * (mcg_is16(e->dt) ? 2 : 4);
  long long cap = 0x7FFFFF00ll / (per_frame > 0 ? per_frame : 1);
  if (e->max_range_frames > 0 && e->max_range_frames < cap) cap = e->max_range_frames;
  return cap < 1 ? 1 : (cap > 0x7fffffff ? 0x7fffffff : (int)cap);
}
static const int kMinFramesPerRange = 56;  // below this a range's kernels no longer fill the chip on their own
static int trunk_ranges(const mcg_engine* e, int frames) {
  int k = e->trunk_streams;
  if (k > mcg_engine::kMaxSplit) k = mcg_engine::kMaxSplit;
  while (k > 1 && frames / k < kMinFramesPerRange) --k;
  return k < 1 ? 1 : k;
}

static inline size_t al256(size_t b) { return (b + 255) / 256 * 256; }
static inline size_t esize(mcg_dtype dt) { return mcg_is16(dt) ? 2 : 4; }

extern "C" void mcg_engine_destroy(mcg_engine* e);
extern "C" int mcg_engine_create(mcg_engine** out, const mcg_model_weights* w, mcg_dtype dt) {
  MCG_CHECK_ARG(out && w, "mcg_engine_create: null pointer");
  MCG_CHECK_ARG(dt == MCG_F32 || dt == MCG_BF16 || dt == MCG_F16X3 || dt == MCG_F16, "mcg_engine_create: unknown dtype %d", (int)dt);
  int expect = 0;
  for (int l = 0; l < 4; ++l) {
    MCG_CHECK_ARG(w->blocks[l] > 0, "mcg_engine_create: blocks[%d]=%d", l, w->blocks[l]);
    expect += 3 * w->blocks[l] + 1;
  }
  MCG_CHECK_ARG(w->num_convs == expect && w->convs, "mcg_engine_create: expected %d bottleneck convs, got %d", expect, w->num_convs);
  MCG_CHECK_ARG(w->num_stages > 0 && w->stage_weights && w->gaze_weights && w->init_boxes && w->init_feats, "mcg_engine_create: decoder tables missing");
  for (int i = 0; i < w->num_convs; ++i)
    MCG_CHECK_ARG(w->convs[i].w && w->convs[i].bias, "mcg_engine_create: conv %d has a null pointer", i);
  mcg_engine* e = new mcg_engine();
  e->dt = dt;
  memcpy(e->blocks, w->blocks, sizeof(e->blocks));
  e->stem = w->stem;
  e->convs.assign(w->convs, w->convs + w->num_convs);
  memcpy(e->lateral, w->lateral, sizeof(e->lateral));
  memcpy(e->fpn_out, w->fpn_out, sizeof(e->fpn_out));
  memcpy(e->c3_ds, w->c3_ds, sizeof(e->c3_ds));
  if (w->fused && w->num_fused > 0 && dt == MCG_F16X3) e->fused.assign(w->fused, w->fused + w->num_fused);
  e->init_boxes = w->init_boxes;
  e->init_feats = w->init_feats;
  e->num_stages = w->num_stages;
  e->stage_w.assign(w->stage_weights, w->stage_weights + (size_t)w->num_stages * MCG_SW_COUNT);
  memcpy(e->gaze_w, w->gaze_weights, sizeof(e->gaze_w));
  memcpy(e->stds, w->bbox_stds, sizeof(e->stds));
  bool ok = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming) == hipSuccess;
  e->pool = stream_pool_for_current_device();
  ok = ok && e->pool && e->pool->ok;
  if (e->pool) e->cand = e->pool->cand;
  for (int i = 0; i < mcg_engine::kMaxSplit - 1; ++i) ok = ok && hipEventCreateWithFlags(&e->ev_join[i], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    mcg_engine_destroy(e);
    mcg_set_error("mcg_engine_create: could not create the side streams / events");
    return MCG_ERR_HIP;
  }
  *out = e;
  return MCG_OK;
}

// Options.  Everything that used to be an environment switch of the lab bench is an explicit, per-engine setting.
extern "C" int mcg_engine_set_option(mcg_engine* e, const char* name, int value) {
  MCG_CHECK_ARG(e && name, "mcg_engine_set_option: null pointer");
  std::lock_guard<std::mutex> lock(e->mu);
  if (!strcmp(name, "trunk_streams")) { MCG_CHECK_ARG(value >= 1 && value <= mcg_engine::kMaxSplit, "trunk_streams must be 1..%d", mcg_engine::kMaxSplit); e->trunk_streams = value; }
  else if (!strcmp(name, "max_range_frames")) { MCG_CHECK_ARG(value >= 0, "max_range_frames must be >= 0"); e->max_range_frames = value; }
  else if (!strcmp(name, "tile")) e->ctx.tile = value > 0 ? value : -1;
  else if (!strcmp(name, "staged_gemm")) e->ctx.staged = value != 0;
  else if (!strcmp(name, "conv3x3_c64")) e->ctx.c64 = value != 0;
  else if (!strcmp(name, "stem_fused")) e->ctx.stem_fused = value != 0;
  else if (!strcmp(name, "decoder_chain")) e->ctx.chain = value != 0;
  else if (!strcmp(name, "decoder_attn_block")) e->ctx.attn_block = value != 0;
  else if (!strcmp(name, "pointwise_pair")) e->pw_pair = value != 0;
  else if (!strcmp(name, "pointwise_stream")) e->pw_single = value != 0;
  else if (!strcmp(name, "bottleneck_fused")) e->bneck_fused = value != 0;
  else if (!strcmp(name, "bottleneck_blocked")) e->bneck_blocked = value != 0;
  else if (!strcmp(name, "winograd")) { MCG_CHECK_ARG(value >= 0 && value <= 2, "winograd must be 0, 1 or 2"); e->winograd = value; }
  else if (!strcmp(name, "wino_tile")) { MCG_CHECK_ARG(value >= -1 && value <= 3, "wino_tile must be -1 (by grid size) .. 3"); e->wino_tile = value; }
  else if (!strcmp(name, "range_audit")) {
    MCG_CHECK_ARG(!mcg_is16(e->dt) || !value, "range_audit: f32-storage engines only (MCG_F32, MCG_F16X3)");
    if (value && !e->audit_dev) {   // set-up, not the hot path: the only allocation the library ever makes
      if (hipMalloc((void**)&e->audit_dev, sizeof(unsigned long long) * 2 * mcg_engine::kAuditCap) != hipSuccess) { e->audit_dev = nullptr; mcg_set_error("range_audit: hipMalloc failed"); return MCG_ERR_HIP; }
    }
    if (value && hipMemset(e->audit_dev, 0, sizeof(unsigned long long) * 2 * mcg_engine::kAuditCap) != hipSuccess) { mcg_set_error("range_audit: hipMemset failed"); return MCG_ERR_HIP; }
    if (!value && e->audit_dev) { (void)hipFree(e->audit_dev); e->audit_dev = nullptr; }
    e->audit_names.clear();
  }
  else { mcg_set_error("mcg_engine_set_option: unknown option '%s'", name); return MCG_ERR_ARG; }
  return MCG_OK;
}

// Per-launch profiling, owned by the engine: while armed, every contraction launch the engine makes is bracketed by an event pair.
extern "C" int mcg_engine_profile_start(mcg_engine* e, int capacity) {
  MCG_CHECK_ARG(e, "mcg_engine_profile_start: null engine");
  std::lock_guard<std::mutex> lock(e->mu);
  if (e->prof.recs) { mcg_set_error("mcg_engine_profile_start: already armed"); return MCG_ERR_ARG; }
  MCG_CHECK_ARG(capacity > 0 && capacity <= (1 << 20), "mcg_engine_profile_start: bad capacity %d", capacity);
  e->prof.recs = new ProfRec[capacity];
  for (int i = 0; i < capacity; ++i) { e->prof.recs[i].a = nullptr; e->prof.recs[i].b = nullptr; }
  for (int i = 0; i < capacity; ++i) {
    if (hipEventCreate(&e->prof.recs[i].a) != hipSuccess || hipEventCreate(&e->prof.recs[i].b) != hipSuccess) {
      for (int j = 0; j <= i; ++j) {   // undo: the engine must stay re-armable
        if (e->prof.recs[j].a) (void)hipEventDestroy(e->prof.recs[j].a);
        if (e->prof.recs[j].b) (void)hipEventDestroy(e->prof.recs[j].b);
      }
      delete[] e->prof.recs;
      e->prof = Prof();
      mcg_set_error("mcg_engine_profile_start: hipEventCreate failed");
      return MCG_ERR_HIP;
    }
  }
  e->prof.cap = capacity; e->prof.n = 0;
  e->ctx.prof = &e->prof;
  return MCG_OK;
}
extern "C" int mcg_engine_profile_stop(mcg_engine* e, int* count, float* ms, double* flops, double* bytes, int* cfg, int* shape, int capacity) {
  MCG_CHECK_ARG(e, "mcg_engine_profile_stop: null engine");
  std::lock_guard<std::mutex> lock(e->mu);
  if (!e->prof.recs) { mcg_set_error("mcg_engine_profile_stop: not armed"); return MCG_ERR_ARG; }
  const int n = e->prof.n;
  int rc = MCG_OK;
  for (int i = 0; i < n; ++i) {
    const ProfRec& r = e->prof.recs[i];
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) rc = MCG_ERR_HIP;
    if (i < capacity) {
      if (ms) ms[i] = t;
      if (flops) flops[i] = r.flops;
      if (bytes) bytes[i] = r.bytes;
      if (cfg) cfg[i] = r.cfg;
      if (shape) { shape[3 * i] = r.shape[0]; shape[3 * i + 1] = r.shape[1]; shape[3 * i + 2] = r.shape[2]; }
    }
  }
  for (int i = 0; i < e->prof.cap; ++i) { (void)hipEventDestroy(e->prof.recs[i].a); (void)hipEventDestroy(e->prof.recs[i].b); }
  delete[] e->prof.recs;
  e->prof = Prof();
  e->ctx.prof = nullptr;
  if (count) *count = n < capacity ? n : capacity;
  if (rc != MCG_OK) mcg_set_error("mcg_engine_profile_stop: event query failed");
  return rc;
}
extern "C" void mcg_engine_destroy(mcg_engine* e) {
  if (!e) return;
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  for (int i = 0; i < mcg_engine::kMaxSplit - 1; ++i)
    if (e->ev_join[i]) (void)hipEventDestroy(e->ev_join[i]);
  if (e->prof.recs) {
    for (int i = 0; i < e->prof.cap; ++i) { (void)hipEventDestroy(e->prof.recs[i].a); (void)hipEventDestroy(e->prof.recs[i].b); }
    delete[] e->prof.recs;
  }
  if (e->audit_dev) (void)hipFree(e->audit_dev);
  delete e;
}

// ---------------------------------------------------------------- range audit (debug option)
__global__ void range_audit_kernel(const float* __restrict__ x, long long n, unsigned long long* __restrict__ cnt) {
  unsigned long long big = 0, bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float v = x[i];
    if (!(fabsf(v) <= 3.4028235e38f)) ++bad;       // inf or nan
    else if (fabsf(v) > 65504.f) ++big;
  }
  big = (unsigned long long)wave_sum((float)big) ;  // counts per wave stay far below 2^24: exact in f32
  bad = (unsigned long long)wave_sum((float)bad);
  if ((threadIdx.x & 63) == 0) {
    if (big) atomicAdd(cnt, big);
    if (bad) atomicAdd(cnt + 1, bad);
  }
}
// One audited tensor: slot idx of the chunk's sequence (the same for every chunk and frame range, so the counters add up over the batch)
struct AuditCursor { mcg_engine* e; hipStream_t s; int idx; bool naming; };
static void audit_tensor(AuditCursor& a, const char* name, const void* p, long long elements) {
  if (!a.e->audit_dev || a.idx >= mcg_engine::kAuditCap) return;
  if (a.naming) a.e->audit_names.push_back(name);
  const long long per_thread = 16;
  long long blocks = (elements + 256 * per_thread - 1) / (256 * per_thread);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(range_audit_kernel, dim3((int)blocks), dim3(256), 0, a.s, (const float*)p, elements, a.e->audit_dev + 2 * a.idx);
  ++a.idx;
}

// ---------------------------------------------------------------- trunk workspace for one chunk of n frames
struct TrunkWs {
  char *stem_ws, *x0, *xa, *xb, *o1, *o2, *ds, *c[4], *l[4];
  size_t stem_bytes, total;
};
static TrunkWs trunk_layout(mcg_dtype dt, int n, int H, int W, char* base) {
  const size_t es = esize(dt);
  const size_t h2 = H / 4, w2 = W / 4;
  TrunkWs t;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  t.stem_bytes = mcg_stem_workspace_bytes(dt, n, H, W);
  t.stem_ws = take(t.stem_bytes);
  const size_t big = (size_t)n * h2 * w2 * 256 * es;  // the largest activation: C2 = [n, H/4, W/4, 256]
  t.x0 = take((size_t)n * h2 * w2 * 64 * es);
  t.xa = take(big); t.xb = take(big);
  t.o1 = take(big / 2); t.o2 = take(big / 2);
  t.ds = take(big);
  for (int i = 0; i < 4; ++i) {
    const size_t hi = (H / 4) >> i, wi = (W / 4) >> i;
    t.c[i] = take((size_t)n * hi * wi * (256u << i) * es);
    t.l[i] = take((size_t)n * hi * wi * 256 * es);
  }
  t.total = off;
  return t;
}

static int conv_call(const mcg_engine* e, hipStream_t s, mcg_dtype dt, const mcg_conv_weights& cw, const void* x, int n, int h, int w, void* y,
                     int relu, const void* res, int res_mode, int hr, int wr) {
  mcg_conv_desc d;
  memset(&d, 0, sizeof(d));
  d.x = x; d.w = cw.w; d.bias = cw.bias; d.residual = res; d.y = y;
  d.N = n; d.H = h; d.W = w; d.Cin = cw.cin; d.Cout = cw.cout; d.KH = cw.k; d.KW = cw.k; d.stride = cw.stride; d.pad = cw.pad;
  d.relu = relu; d.residual_mode = res_mode; d.Hr = hr; d.Wr = wr;
  d.wscale = dt == MCG_F16X3 ? cw.wscale : 0.f;
  const long long M = (long long)n * h * w;
  const int rm = res ? res_mode : MCG_RES_NONE;
  if (mcg_is16(dt) && e->pw_single && e->ctx.tile < 0 && !e->ctx.staged && cw.wf && cw.bias && cw.k == 1 && cw.stride == 1 && cw.pad == 0 &&
      pw_single_applicable(cw.cin, cw.cout, rm, M, rm == MCG_RES_UPSAMPLE_ADD ? (long long)n * hr * wr : M) && M < 0x7fffffffll) {
    PwSingleParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.a = x; pp.res = res; pp.wf = cw.wf; pp.bias = cw.bias; pp.y = y;
    pp.M = (int)M; pp.relu = relu; pp.Ho = h; pp.Wo = w;
    if (rm == MCG_RES_UPSAMPLE_ADD) { pp.Hr = hr; pp.Wr = wr; pp.rscale_h = (float)hr / (float)h; pp.rscale_w = (float)wr / (float)w; }
    const double res_rows = rm == MCG_RES_NONE ? 0.0 : (rm == MCG_RES_ADD ? (double)M : (double)n * hr * wr);
    ProfRec* rec = prof_begin(e->ctx, s, 61, pp.M, cw.cout, cw.cin, 2.0 * M * cw.cin * cw.cout,
                              2.0 * ((double)M * (cw.cin + cw.cout) + res_rows * cw.cout + (double)cw.cin * cw.cout));
    const int prc = launch_pw_single(s, pp, cw.cin, cw.cout, rm == MCG_RES_NONE ? 0 : (rm == MCG_RES_ADD ? 1 : 2), dt == MCG_F16);
    prof_end(rec, s);
    if (prc) { mcg_set_error("pw_single launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  if (dt == MCG_F16X3 && e->pw_single && e->ctx.tile < 0 && cw.wf && cw.bias && cw.k == 1 && cw.stride == 1 && cw.pad == 0 &&
      pw_single_x3_applicable(cw.cin, cw.cout, rm, M, rm == MCG_RES_UPSAMPLE_ADD ? (long long)n * hr * wr : M) && M < 0x7fffffffll) {
    // f16x3: the HBM-bound 256 -> 256 / 256 -> 1024 convs by the persistent streaming kernel (pw_single_x3.hpp)
    PwSingleParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.a = x; pp.res = res; pp.wf = cw.wf; pp.bias = cw.bias; pp.y = y;
    pp.M = (int)M; pp.relu = relu; pp.Ho = h; pp.Wo = w; pp.wscale = cw.wscale;
    if (rm == MCG_RES_UPSAMPLE_ADD) { pp.Hr = hr; pp.Wr = wr; pp.rscale_h = (float)hr / (float)h; pp.rscale_w = (float)wr / (float)w; }
    const double res_rows = rm == MCG_RES_NONE ? 0.0 : (rm == MCG_RES_ADD ? (double)M : (double)n * hr * wr);
    ProfRec* rec = prof_begin(e->ctx, s, 71, pp.M, cw.cout, cw.cin, 2.0 * M * cw.cin * cw.cout,
                              4.0 * ((double)M * (cw.cin + cw.cout) + res_rows * cw.cout + (double)cw.cin * cw.cout));
    const int prc = launch_pw_single_x3(s, pp, cw.cout, rm == MCG_RES_NONE ? 0 : (rm == MCG_RES_ADD ? 1 : 2));
    prof_end(rec, s);
    if (prc) { mcg_set_error("pw_single_x3 launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  // F(4,3) where the layer's shape allows it and the weights carry that copy, else F(2,3), else the direct kernel -- by SHAPE only
  const int wg = (dt == MCG_F16X3 && e->winograd && e->ctx.tile < 0 && cw.k == 3 && cw.stride == 1 && cw.pad == 1 && rm == MCG_RES_NONE)
                     ? ((cw.wf4 && e->winograd >= 2 && wino_x3_applicable(n, h, w, cw.cin, cw.cout, 4)) ? 4 : ((cw.wf && wino_x3_applicable(n, h, w, cw.cin, cw.cout, 2)) ? 2 : 0))
                     : 0;
  if (wg) {
    // f16x3: the 3x3 / stride 1 convs (FPN outputs, layer3's conv2) as a 1-D Winograd F(2,3) contraction (wino_x3.hpp); FLOPs and bytes
    // are booked as the DIRECT convolution's: the roofline keeps counting the reference's arithmetic
    WinoParams wp;
    memset(&wp, 0, sizeof(wp));
    wp.x = (const float*)x; wp.u = wg == 4 ? cw.wf4 : cw.wf; wp.bias = cw.bias; wp.y = (float*)y;
    wp.H = h; wp.W = w; wp.frames = n; wp.Cin = cw.cin; wp.Cout = cw.cout; wp.relu = relu; wp.wscale = cw.wscale;
    ProfRec* rec = prof_begin(e->ctx, s, 73, (int)M, cw.cout, 9 * cw.cin, 2.0 * M * 9.0 * cw.cin * cw.cout,
                              4.0 * ((double)M * (cw.cin + cw.cout) + 9.0 * cw.cin * cw.cout));
    const int wrc = launch_wino_x3(s, wp, e->wino_tile, wg);
    prof_end(rec, s);
    if (wrc) { mcg_set_error("wino_x3 launch failed"); return MCG_ERR_HIP; }
    return MCG_OK;
  }
  return conv2d_ctx(s, dt, &d, e->ctx);
}

// Backbone + FPN over frames [f0, f0+n): writes pyramid level i at frame offset f0.
static int trunk_chunk(mcg_engine* e, hipStream_t s, const float* img, int f0, int n, int H, int W, void* const pyr[4], char* wsbase,
                       bool backbone_only = false) {
  const mcg_dtype dt = e->dt;
  const size_t es = esize(dt);
  TrunkWs t = trunk_layout(dt, n, H, W, wsbase);
  MCG_TRY(stem_forward_ctx(s, dt, img + (size_t)f0 * 3 * H * W, e->stem.w, e->stem.bias, t.x0, n, H, W, t.stem_ws, t.stem_bytes, e->ctx));
  AuditCursor au{e, s, 0, e->audit_dev != nullptr && e->audit_names.empty()};
  char aname[64];
  audit_tensor(au, "stem", t.x0, (long long)n * (H / 4) * (W / 4) * 64);
  const void* x = t.x0;
  int h = H / 4, w = W / 4, ci = 0;
  bool x_blocked = false;  // x is in the fused tail's blocked layout (written by the previous block's fused tail for the next one's only)
  bool o1_ready = false;   // o1 already holds this block's conv1 output (written by the previous block's pointwise-pair / fused-tail kernel)
  char *o1 = t.o1, *o2 = t.o2;   // conv1 / conv2 outputs; the fused tail writes the NEXT conv1 output into o2 and the two swap
  for (int l = 0; l < 4; ++l) {
    for (int b = 0; b < e->blocks[l]; ++b) {
      const mcg_conv_weights& c1 = e->convs[ci], &c2 = e->convs[ci + 1], &c3 = e->convs[ci + 2];
      const bool has_ds = b == 0;
      const int ho = (h + 2 * c2.pad - c2.k) / c2.stride + 1, wo = (w + 2 * c2.pad - c2.k) / c2.stride + 1;
      void* y = (b == e->blocks[l] - 1) ? (void*)t.c[l] : (x == t.xa ? (void*)t.xb : (void*)t.xa);
      if (!o1_ready) {
        MCG_TRY(conv_call(e, s, dt, c1, x, n, h, w, o1, 1, nullptr, MCG_RES_NONE, 0, 0));
        snprintf(aname, sizeof(aname), "layer%d.%d.conv1", l + 1, b);
        audit_tensor(au, aname, o1, (long long)n * h * w * c1.cout);
      }
      o1_ready = false;
      // f16x3: conv2 -> conv3 (+ downsample / + residual) -> the next block's conv1 as ONE kernel (bneck_x3.hpp)
      // fused_at(conv index of a block's conv1, block index): that block's fused tail if one was handed over and every shape matches
      auto fused_at = [&](int cj, int bj) -> const mcg_fused_block* {
        if (!(dt == MCG_F16X3 && e->bneck_fused && e->ctx.tile < 0) || cj + 2 >= (int)e->convs.size()) return nullptr;
        const mcg_fused_block* f = nullptr;
        for (const mcg_fused_block& q : e->fused)
          if (q.conv2_index == cj + 1) f = &q;
        if (!f) return nullptr;
        const bool ds = bj == 0;
        const mcg_conv_weights &d2 = e->convs[cj + 1], &d3 = e->convs[cj + 2];
        const int kk = ds ? e->convs[cj + 3].cin : 0, ss = ds ? e->convs[cj + 3].stride : 1;
        const int cn_i = cj + (ds ? 4 : 3);
        const bool nx = f->cn == 0 || (cn_i < (int)e->convs.size() && e->convs[cn_i].k == 1 && e->convs[cn_i].stride == 1 && e->convs[cn_i].cout == f->cn && e->convs[cn_i].cin == f->c);
        return (d2.k == 3 && d2.stride == 1 && d2.pad == 1 && d2.cin == f->cm && d2.cout == f->cm && d3.cout == f->c && f->nsrc == (ds ? 2 : 1) && nx &&
                bneck_x3_applicable(f->cm, f->c, f->cn, f->nsrc, kk, ss)) ? f : nullptr;
      };
      const mcg_fused_block* fb = fused_at(ci, b);
      if (x_blocked && !(fb && fb->nsrc == 1)) { mcg_set_error("trunk: a blocked tensor reached a kernel that does not read that layout"); return MCG_ERR_UNSUPPORTED; }
      if (fb) {
        const int k2 = has_ds ? e->convs[ci + 3].cin : 0;
        const int ci_nx = ci + (has_ds ? 4 : 3);
        {
          BneckParams bp;
          memset(&bp, 0, sizeof(bp));
          bp.x = (const float*)o1; bp.res = (const float*)x; bp.wstream = (const char*)fb->wstream; bp.bias = fb->bias;
          bp.y = (float*)y; bp.z = (float*)o2; bp.H = h; bp.W = w;
          // y goes to the NEXT block's fused tail only (as its residual; its conv1 is this kernel's z, cn > 0): both sides use the kernel's blocked
          // layout -- whole-line stores and loads (bneck_x3.hpp) -- where the block grid fits the ping-pong buffer; not under range_audit,
          // which reads tensors as [M][C]
          const long long blk_bytes = (long long)n * ((h + bnx::TH - 1) / bnx::TH) * ((w + bnx::TW - 1) / bnx::TW) * bnx::NPIX * fb->c * (long long)es;
          const mcg_fused_block* fnext = (b + 1 < e->blocks[l]) ? fused_at(ci_nx, b + 1) : nullptr;
          bp.res_blocked = x_blocked ? 1 : 0;
          bp.y_blocked = (e->bneck_blocked && !e->audit_dev && y != (void*)t.c[l] && fb->cn > 0 && fnext && fnext->nsrc == 1 && fnext->c == fb->c &&
                          blk_bytes <= (long long)((size_t)n * (H / 4) * (W / 4) * 256 * es)) ? 1 : 0;
          const double M = (double)n * h * w;
          ProfRec* rec = prof_begin(e->ctx, s, 70, n * h * w, fb->c + fb->cn, 9 * fb->cm + fb->cm + k2 + fb->c,
                                    2.0 * M * (9.0 * fb->cm * fb->cm + (double)(fb->cm + k2) * fb->c + (double)fb->c * fb->cn),
                                    4.0 * (M * (fb->cm + (has_ds ? k2 : fb->c) + fb->c + fb->cn) + 9.0 * fb->cm * fb->cm + (double)(fb->cm + k2) * fb->c + (double)fb->c * fb->cn));
          const int frc = launch_bneck_x3(s, bp, n, fb->cm, fb->nsrc, fb->cn);
          prof_end(rec, s);
          if (frc) { mcg_set_error("bneck_x3 launch failed"); return MCG_ERR_HIP; }
          snprintf(aname, sizeof(aname), "layer%d.%d.out (fused tail)", l + 1, b);
          audit_tensor(au, aname, y, (long long)n * h * w * fb->c);
          if (fb->cn > 0) {
            snprintf(aname, sizeof(aname), "layer%d.%d.next_conv1 (fused tail)", l + 1, b);
            audit_tensor(au, aname, o2, (long long)n * h * w * fb->cn);
          }
          if (fb->cn > 0) { char* tmp = o1; o1 = o2; o2 = tmp; o1_ready = true; }
          x = y; h = ho; w = wo;
          x_blocked = bp.y_blocked != 0;
          ci = ci_nx;
          continue;
        }
      }
      MCG_TRY(conv_call(e, s, dt, c2, o1, n, h, w, o2, 1, nullptr, MCG_RES_NONE, 0, 0));
      snprintf(aname, sizeof(aname), "layer%d.%d.conv2", l + 1, b);
      audit_tensor(au, aname, o2, (long long)n * ho * wo * c2.cout);
      // conv3 (+ downsample / + residual) together with the NEXT block's conv1 (pw_pair.hpp): layer1 / layer2, where both are
      // HBM-bound.  The next conv1 is the following block's, or the next layer's first (1x1, stride 1 on this block's output).
      const int ci_next = ci + (has_ds ? 4 : 3);
      const mcg_conv_weights* c3w = has_ds ? (e->c3_ds[l].w ? &e->c3_ds[l] : nullptr) : &c3;
      const mcg_conv_weights* c1n = ci_next < (int)e->convs.size() ? &e->convs[ci_next] : nullptr;
      const int k2 = has_ds ? e->convs[ci + 3].cin : 0;
      if (mcg_is16(dt) && e->pw_pair && l == 0 && c3w && c1n && c3w->wf && c1n->wf && c3w->bias && c1n->bias && c1n->k == 1 && c1n->stride == 1 &&
          c1n->cin == c3.cout && pw_pair_applicable(c3.cin, k2, has_ds ? e->convs[ci + 3].stride : 1, c3.cout, c1n->cout, (long long)n * ho * wo) && (long long)n * ho * wo < 0x7fffffffll) {
        PwPairParams pp;
        memset(&pp, 0, sizeof(pp));
        pp.a1 = o2; pp.K1 = c3.cin;
        if (has_ds) { pp.a2 = x; pp.K2 = k2; pp.stride2 = e->convs[ci + 3].stride; pp.H2 = h; pp.W2 = w; }
        else pp.res = x;
        pp.w3f = c3w->wf; pp.b3 = c3w->bias; pp.y = y;
        pp.w1f = c1n->wf; pp.b1 = c1n->bias; pp.z = o1;
        pp.M = n * ho * wo; pp.C = c3.cout; pp.C2 = c1n->cout; pp.Ho = ho; pp.Wo = wo;
        // cfg 60: both contractions of the pair count (2 M (K C + C C2))
        ProfRec* rec = prof_begin(e->ctx, s, 60, pp.M, pp.C + pp.C2, pp.K1 + pp.K2, 2.0 * pp.M * ((double)(pp.K1 + pp.K2) * pp.C + (double)pp.C * pp.C2),
                                  2.0 * ((double)pp.M * (pp.K1 + pp.K2 + (has_ds ? 0 : pp.C) + pp.C + pp.C2) + (double)(pp.K1 + pp.K2) * pp.C + (double)pp.C * pp.C2));
        const int prc = launch_pw_pair(s, pp, dt == MCG_F16);
        prof_end(rec, s);
        if (prc) { mcg_set_error("pw_pair launch failed"); return MCG_ERR_HIP; }
        o1_ready = true;
        x = y; h = ho; w = wo;
        ci = ci_next;
        continue;
      }
      if (has_ds && e->c3_ds[l].w) {
        // conv3 and the downsample conv as ONE K-concatenated GEMM: relu([o2 | x@stride] . [W3 | Wd]^T + b3 + bd);
        // the downsample output never goes to HBM and conv3 reads no residual.
        const mcg_conv_weights& f = e->c3_ds[l];
        mcg_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.x = o2; d.w = f.w; d.bias = f.bias; d.y = y;
        d.wscale = dt == MCG_F16X3 ? f.wscale : 0.f;
        d.N = n; d.H = ho; d.W = wo; d.Cin = c3.cin; d.Cout = c3.cout; d.KH = 1; d.KW = 1; d.stride = 1; d.pad = 0; d.relu = 1;
        d.x2 = x; d.Cin2 = e->convs[ci + 3].cin; d.stride2 = e->convs[ci + 3].stride; d.H2 = h; d.W2 = w;
        MCG_TRY(conv2d_ctx(s, dt, &d, e->ctx));
      } else {
        const void* identity = x;
        if (has_ds) {
          MCG_TRY(conv_call(e, s, dt, e->convs[ci + 3], x, n, h, w, t.ds, 0, nullptr, MCG_RES_NONE, 0, 0));
          identity = t.ds;
        }
        MCG_TRY(conv_call(e, s, dt, c3, o2, n, ho, wo, y, 1, identity, MCG_RES_ADD, 0, 0));
      }
      snprintf(aname, sizeof(aname), "layer%d.%d.out", l + 1, b);
      audit_tensor(au, aname, y, (long long)n * ho * wo * c3.cout);
      x = y; h = ho; w = wo;
      ci += has_ds ? 4 : 3;
    }
  }
  if (backbone_only) return MCG_OK;  // mcg_bench_backbone_forward: C2..C5 only
  // FPN (fpn.py:157-180): laterals top-down with the nearest-upsample add fused into the epilogue
  int hs[4], wsz[4];
  for (int i = 0; i < 4; ++i) { hs[i] = (H / 4) >> i; wsz[i] = (W / 4) >> i; }
  for (int i = 3; i >= 0; --i) {
    const void* res = i == 3 ? nullptr : t.l[i + 1];
    MCG_TRY(conv_call(e, s, dt, e->lateral[i], t.c[i], n, hs[i], wsz[i], t.l[i], 0, res, res ? MCG_RES_UPSAMPLE_ADD : MCG_RES_NONE,
                      i == 3 ? 0 : hs[i + 1], i == 3 ? 0 : wsz[i + 1]));
    snprintf(aname, sizeof(aname), "fpn.lateral%d (+ top-down)", i);
    audit_tensor(au, aname, t.l[i], (long long)n * hs[i] * wsz[i] * 256);
  }
  for (int i = 0; i < 4; ++i) {
    char* dst = (char*)pyr[i] + (size_t)f0 * hs[i] * wsz[i] * 256 * es;
    MCG_TRY(conv_call(e, s, dt, e->fpn_out[i], t.l[i], n, hs[i], wsz[i], dst, 0, nullptr, MCG_RES_NONE, 0, 0));
    snprintf(aname, sizeof(aname), "fpn.P%d", i + 2);
    audit_tensor(au, aname, dst, (long long)n * hs[i] * wsz[i] * 256);
  }
  return MCG_OK;
}

// Debug read-out of the range audit (engine option range_audit = 1): synchronises the device, copies the counters to the host and
// resets them.  counts[2 i] = values with |x| > 65504, counts[2 i + 1] = non-finite values of audited tensor i over every frame processed
// since the last read; names[i] (optional) points at the engine-owned name of tensor i (valid until the option changes).
extern "C" int mcg_engine_range_audit(mcg_engine* e, unsigned long long* counts, const char** names, int capacity, int* n_out) {
  MCG_CHECK_ARG(e && counts && n_out && capacity > 0, "mcg_engine_range_audit: bad argument");
  std::lock_guard<std::mutex> lock(e->mu);
  if (!e->audit_dev) { mcg_set_error("mcg_engine_range_audit: the range_audit option is off"); return MCG_ERR_ARG; }
  const int n = (int)e->audit_names.size() < capacity ? (int)e->audit_names.size() : capacity;
  if (hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(counts, e->audit_dev, sizeof(unsigned long long) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemset(e->audit_dev, 0, sizeof(unsigned long long) * 2 * mcg_engine::kAuditCap) != hipSuccess) {
    mcg_set_error("mcg_engine_range_audit: device read failed");
    return MCG_ERR_HIP;
  }
  if (names) for (int i = 0; i < n; ++i) names[i] = e->audit_names[i].c_str();
  *n_out = n;
  return MCG_OK;
}

// Decoder scratch for N frames (RoI features, query state, per-stage and gaze-head workspaces).
struct DecWs {
  char *roi, *obj_a, *obj_b, *stage_ws, *gaze_ws;
  float *boxes_a, *boxes_b, *cls;
  size_t stage_bytes, gaze_bytes, total;
};
static DecWs dec_layout(mcg_dtype dt, int N, char* base) {
  const size_t es = esize(dt), R = (size_t)N * 3;
  DecWs c;
  memset(&c, 0, sizeof(c));
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += al256(bytes); return p; };
  c.roi = take(R * 49 * 256 * es);
  c.obj_a = take(R * 256 * es); c.obj_b = take(R * 256 * es);
  c.boxes_a = (float*)take(R * 4 * 4); c.boxes_b = (float*)take(R * 4 * 4); c.cls = (float*)take(R * 4);
  c.stage_bytes = mcg_stage_workspace_bytes(dt, N); c.stage_ws = take(c.stage_bytes);
  c.gaze_bytes = mcg_gaze_head_workspace_bytes(dt, N); c.gaze_ws = take(c.gaze_bytes);
  c.total = off;
  return c;
}
static size_t pyramid_bytes(mcg_dtype dt, int N, int H, int W, int level) {
  return al256((size_t)N * ((H / 4) >> level) * ((W / 4) >> level) * 256 * esize(dt));
}

extern "C" size_t mcg_trunk_workspace_bytes(const mcg_engine* e, int N, int H, int W, int chunk) {
  if (!e || N <= 0) return 0;
  if (chunk <= 0 || chunk > N) chunk = N;
  if (chunk < N) return trunk_layout(e->dt, chunk, H, W, nullptr).total;
  // whole batch: one layout per concurrent frame range, sized for the maximum split so the answer does not depend on the environment
  size_t total = 0;
  const int cap = range_frame_cap(e, H, W);
  for (int k = 1; k <= mcg_engine::kMaxSplit; ++k) {
    const int per = (N + k - 1) / k < cap ? (N + k - 1) / k : cap;
    const size_t t = (size_t)k * al256(trunk_layout(e->dt, per, H, W, nullptr).total);
    if (t > total) total = t;
  }
  return total;
}
extern "C" size_t mcg_decoder_workspace_bytes(const mcg_engine* e, int N) {
  if (!e || N <= 0) return 0;
  return dec_layout(e->dt, N, nullptr).total;
}
extern "C" size_t mcg_engine_workspace_bytes(const mcg_engine* e, int N, int H, int W, int chunk) {
  if (!e || N <= 0) return 0;
  size_t total = mcg_trunk_workspace_bytes(e, N, H, W, chunk) + mcg_decoder_workspace_bytes(e, N);
  for (int i = 0; i < 4; ++i) total += pyramid_bytes(e->dt, N, H, W, i);
  return total;
}

static int check_shape(int N, int H, int W) {
  MCG_CHECK_ARG(N > 0, "num_frames must be positive (got %d)", N);
  MCG_CHECK_ARG(H >= 32 && W >= 32 && H % 32 == 0 && W % 32 == 0, "frame size %dx%d must be a multiple of 32 (Pad(size_divisor=32))", H, W);
  return MCG_OK;
}

static int trunk_forward(mcg_engine* e, mcg_stream s_, const float* img, int N, int H, int W, int chunk,
                         void* const pyramid[4], void* ws, size_t ws_bytes, bool backbone_only) {
  MCG_CHECK_ARG(e && img && (pyramid || backbone_only) && ws, "mcg_backbone_fpn_forward: null pointer");
  MCG_TRY(check_shape(N, H, W));
  std::lock_guard<std::mutex> lock(e->mu);
  if (chunk <= 0 || chunk > N) chunk = N;
  const size_t need = mcg_trunk_workspace_bytes(e, N, H, W, chunk);
  if (ws_bytes < need) { mcg_set_error("mcg_backbone_fpn_forward: workspace too small (%zu < %zu)", ws_bytes, need); return MCG_ERR_WORKSPACE; }
  hipStream_t s = (hipStream_t)s_;
  if (chunk < N) {  // sequential frame chunks in a small workspace
    for (int f0 = 0; f0 < N; f0 += chunk) MCG_TRY(trunk_chunk(e, s, img, f0, (N - f0) < chunk ? (N - f0) : chunk, H, W, pyramid, (char*)ws, backbone_only));
    return MCG_OK;
  }
  const int k = trunk_ranges(e, N);
  const int cap = range_frame_cap(e, H, W);
  const int per = (N + k - 1) / k < cap ? (N + k - 1) / k : cap;
  if (k == 1) {
    for (int f0 = 0; f0 < N; f0 += per) MCG_TRY(trunk_chunk(e, s, img, f0, (N - f0) < per ? (N - f0) : per, H, W, pyramid, (char*)ws, backbone_only));
    return MCG_OK;
  }
  // fork: the side streams start after everything already queued on the caller's stream (the input, the previous consumer of
  // the pyramid buffers); join: the caller's stream continues after every range.  Range i owns stream i and workspace slot i;
  // a batch larger than k capped ranges takes several rounds on the same streams and slots (in order per stream).
  const std::vector<int>& sides = side_streams_for(e, s);
  const size_t part_ws = al256(trunk_layout(e->dt, per, H, W, nullptr).total);
  if (hipEventRecord(e->ev_fork, s) != hipSuccess) { mcg_set_error("mcg_backbone_fpn_forward: hipEventRecord failed"); return MCG_ERR_HIP; }
  int rc = MCG_OK;
  for (int i = 1; i < k; ++i)
    if (hipStreamWaitEvent(e->cand[sides[i - 1]], e->ev_fork, 0) != hipSuccess) { mcg_set_error("mcg_backbone_fpn_forward: hipStreamWaitEvent failed"); return MCG_ERR_HIP; }
  for (int f0 = 0, i = 0; f0 < N && rc == MCG_OK; f0 += per, i = (i + 1) % k) {
    hipStream_t si = i == 0 ? s : e->cand[sides[i - 1]];
    rc = trunk_chunk(e, si, img, f0, (N - f0) < per ? (N - f0) : per, H, W, pyramid, (char*)ws + (size_t)i * part_ws, backbone_only);
  }
  for (int i = 1; i < k; ++i) {  // joined even after a failed launch, so the caller's stream never runs ahead of a side stream
    hipStream_t si = e->cand[sides[i - 1]];
    if (hipEventRecord(e->ev_join[i - 1], si) != hipSuccess || hipStreamWaitEvent(s, e->ev_join[i - 1], 0) != hipSuccess) {
      mcg_set_error("mcg_backbone_fpn_forward: join failed");
      rc = rc == MCG_OK ? MCG_ERR_HIP : rc;
    }
  }
  return rc;
}

extern "C" int mcg_bottleneck_x3(mcg_stream s, const float* x, const float* src2, const void* wstream, const float* bias, float* y, float* z,
                                 int frames, int H, int W, int cm, int nsrc, int cn, void* trace) {
  MCG_CHECK_ARG(x && src2 && wstream && bias && y && (z || cn == 0), "mcg_bottleneck_x3: null pointer");
  MCG_CHECK_ARG(frames > 0 && H > 0 && W > 0, "mcg_bottleneck_x3: empty problem");
  MCG_CHECK_ARG(bneck_x3_applicable(cm, 4 * cm, cn, nsrc, 64, 1), "mcg_bottleneck_x3: unsupported shape (cm=%d nsrc=%d cn=%d)", cm, nsrc, cn);
  BneckParams bp;
  memset(&bp, 0, sizeof(bp));
  bp.x = x; bp.res = src2; bp.wstream = (const char*)wstream; bp.bias = bias; bp.y = y; bp.z = z; bp.H = H; bp.W = W; bp.trace = (unsigned long long*)trace;
  if (launch_bneck_x3((hipStream_t)s, bp, frames, cm, nsrc, cn)) { mcg_set_error("mcg_bottleneck_x3: launch failed"); return MCG_ERR_HIP; }
  return MCG_OK;
}

extern "C" size_t mcg_conv3x3_wino_x3_weight_bytes(int Cin, int Cout, int g) {
  return (Cin > 0 && Cout > 0 && Cin % 32 == 0 && Cout % wnx::UNT == 0 && (g == 2 || g == 4)) ? wino_x3_weight_bytes(Cin, Cout, g) : 0;
}
extern "C" int mcg_conv3x3_wino_x3(mcg_stream s, const float* x, const void* u, const float* bias, float* y, int frames, int H, int W,
                                   int Cin, int Cout, int relu, int tile, float wscale, int g) {
  MCG_CHECK_ARG(x && u && y, "mcg_conv3x3_wino_x3: null pointer");
  MCG_CHECK_ARG(tile >= 0 && tile <= 4, "mcg_conv3x3_wino_x3: tile must be 0 (by grid size) .. 4");
  if (!wino_x3_applicable(frames, H, W, Cin, Cout, g)) {
    mcg_set_error("mcg_conv3x3_wino_x3: unsupported shape (frames=%d %dx%d, %d -> %d channels, F(%d,3)): Cin %% 32, Cout %% 128, W <= 62 (F(4,3): W %% 4 == 0, W >= 16), a tile's window within its buffer", frames, H, W, Cin, Cout, g);
    return MCG_ERR_UNSUPPORTED;
  }
  WinoParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.x = x; wp.u = u; wp.bias = bias; wp.y = y; wp.H = H; wp.W = W; wp.frames = frames; wp.Cin = Cin; wp.Cout = Cout; wp.relu = relu;
  wp.wscale = wscale;
  if (launch_wino_x3((hipStream_t)s, wp, tile - 1, g)) { mcg_set_error("mcg_conv3x3_wino_x3: launch failed"); return MCG_ERR_HIP; }
  return MCG_OK;
}

extern "C" int mcg_backbone_fpn_forward(mcg_engine* e, mcg_stream s, const float* img, int N, int H, int W, int chunk,
                                        void* const pyramid[4], void* ws, size_t ws_bytes) {
  return trunk_forward(e, s, img, N, H, W, chunk, pyramid, ws, ws_bytes, false);
}
// Measurement entry point (bench.py --workload backbone, BASELINE.json configs[1] "R-50 backbone-only"): the trunk up to C5,
// no FPN; C2..C5 stay in the workspace.
extern "C" int mcg_bench_backbone_forward(mcg_engine* e, mcg_stream s, const float* img, int N, int H, int W, void* ws, size_t ws_bytes) {
  return trunk_forward(e, s, img, N, H, W, 0, nullptr, ws, ws_bytes, true);
}

// Where mcg_bench_backbone_forward left C2..C5 (NHWC, engine dtype) for a batch that ran as ONE frame range (trunk_streams = 1, or fewer
// than 112 frames): pointers into the caller's workspace.  Test / measurement aid.
extern "C" int mcg_bench_backbone_levels(const mcg_engine* e, void* ws, int N, int H, int W, void* levels[4]) {
  MCG_CHECK_ARG(e && ws && levels, "mcg_bench_backbone_levels: null pointer");
  MCG_TRY(check_shape(N, H, W));
  MCG_CHECK_ARG(trunk_ranges(e, N) == 1 && N <= range_frame_cap(e, H, W), "mcg_bench_backbone_levels: the batch runs as several frame ranges (set trunk_streams = 1)");
  const TrunkWs t = trunk_layout(e->dt, N, H, W, (char*)ws);
  for (int i = 0; i < 4; ++i) levels[i] = t.c[i];
  return MCG_OK;
}

extern "C" int mcg_decoder_forward(mcg_engine* e, mcg_stream s_, const void* const pyramid[4], int N, int clip_length, int H, int W,
                                   const int* img_hw, float* gaze_out, float* boxes_out, float* scores_out, void* ws, size_t ws_bytes) {
  hipStream_t s = (hipStream_t)s_;
  MCG_CHECK_ARG(e && pyramid && gaze_out && boxes_out && scores_out && ws, "mcg_decoder_forward: null pointer");
  MCG_TRY(check_shape(N, H, W));
  MCG_CHECK_ARG(clip_length > 0 && N % clip_length == 0, "num_frames=%d is not a multiple of clip_length=%d", N, clip_length);
  DecWs c = dec_layout(e->dt, N, (char*)ws);
  if (ws_bytes < c.total) { mcg_set_error("mcg_decoder_forward: workspace too small (%zu < %zu)", ws_bytes, c.total); return MCG_ERR_WORKSPACE; }
  MCG_TRY(launch_init_queries(s, e->dt, e->init_boxes, e->init_feats, img_hw, H, W, c.boxes_a, c.obj_a, N));
  int fh[4], fw[4];
  const int strides[4] = {4, 8, 16, 32};
  for (int i = 0; i < 4; ++i) { fh[i] = (H / 4) >> i; fw[i] = (W / 4) >> i; }
  char* obj_in = c.obj_a; char* obj_out = c.obj_b;
  float* b_in = c.boxes_a; float* b_out = c.boxes_b;
  for (int st = 0; st < e->num_stages; ++st) {
    MCG_TRY(launch_roi_align(s, e->dt, pyramid, fh, fw, strides, 256, b_in, N * 3, 3, c.roi, nullptr));
    float* bdst = (st == e->num_stages - 1) ? boxes_out : b_out;
    MCG_TRY(stage_forward_ctx(s, e->dt, &e->stage_w[(size_t)st * MCG_SW_COUNT], c.roi, obj_in, b_in, N, clip_length, obj_out, bdst,
                              c.cls, e->stds, c.stage_ws, c.stage_bytes, e->ctx));
    char* t = obj_in; obj_in = obj_out; obj_out = t;
    if (st != e->num_stages - 1) { float* tb = b_in; b_in = b_out; b_out = tb; }
  }
  MCG_TRY(gaze_head_ctx(s, e->dt, e->gaze_w, obj_in, N, gaze_out, c.gaze_ws, c.gaze_bytes, e->ctx, c.cls, scores_out));
  return MCG_OK;
}

extern "C" int mcg_clip_forward(mcg_engine* e, mcg_stream s_, const float* img, int N, int clip_length, int H, int W,
                                const int* img_hw, int chunk, float* gaze_out, float* boxes_out, float* scores_out,
                                void* ws, size_t ws_bytes) {
  MCG_CHECK_ARG(e && img && gaze_out && boxes_out && scores_out && ws, "mcg_clip_forward: null pointer");
  MCG_TRY(check_shape(N, H, W));
  MCG_CHECK_ARG(clip_length > 0 && N % clip_length == 0, "mcg_clip_forward: num_frames=%d is not a multiple of clip_length=%d", N, clip_length);
  const size_t need = mcg_engine_workspace_bytes(e, N, H, W, chunk);
  if (ws_bytes < need) { mcg_set_error("mcg_clip_forward: workspace too small (%zu < %zu)", ws_bytes, need); return MCG_ERR_WORKSPACE; }
  char* base = (char*)ws;
  const size_t trunk_bytes = mcg_trunk_workspace_bytes(e, N, H, W, chunk);
  void* pyr[4];
  size_t off = trunk_bytes;
  for (int i = 0; i < 4; ++i) { pyr[i] = base + off; off += pyramid_bytes(e->dt, N, H, W, i); }
  MCG_TRY(mcg_backbone_fpn_forward(e, s_, img, N, H, W, chunk, pyr, base, trunk_bytes));
  return mcg_decoder_forward(e, s_, pyr, N, clip_length, H, W, img_hw, gaze_out, boxes_out, scores_out, base + off, ws_bytes - off);
}
