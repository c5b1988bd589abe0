// flutl_api.cpp -- host half of the extern "C" shim (include/flutl.h): display
// list validation, the per-element context (stream, atlas and list memory in
// HBM) and the render entry points used by the GStreamer element.  Kernels and
// the stateless launches live in flutl_kernels.hip.
//
// There is deliberately no CPU rendering path here: if HIP is unusable every
// compute entry point fails with FLUTL_ENODEV.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <new>
#include <vector>
#include "flutl.h"

namespace {

constexpr size_t kAtlasSlack = 16;   // readable bytes before/after any entry
constexpr size_t kAtlasAlign = 16;

inline bool hip_ok (hipError_t e) { return e == hipSuccess; }

}  // namespace

// (csrc/flutl_scan.hip; library internal: flutl_launch_scan with the statistics counter)
extern "C" int flutl_launch_scan_counted (void *stream_, const int32_t *edges_dev, uint32_t n_edges,
    const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride, int mode,
    uint32_t *counters_dev);
extern "C" size_t flutl_scan_zero_bytes (void);     // (the scratch's leading words that are zero between conversions)
extern "C" int flutl_launch_scan_hosted (void *stream_, const int32_t *edges, int32_t *raw_copy_dev,
    uint32_t n_edges, const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride,
    int mode, uint32_t *counters_dev, const void *carry_src, void *carry_dst, size_t carry_bytes);

struct flutl_list {
  uint32_t width, height;
  uint32_t n_regions, n_prims;
  flutl_region *regions;  // device
  flutl_prim *prims;      // device (same allocation)
  void *mem;
};

// One submission in flight: its display list in page-locked host memory (the
// caller's list memory is free again when submit returns, and a copy from
// page-locked memory is a true asynchronous DMA) and in HBM.
struct list_slot {
  void *host;
  void *host_dev;      // the device's address of `host`
  void *dev;
  size_t cap;
  hipEvent_t done;     // recorded after the last kernel that reads `dev`
  bool busy;
};
constexpr int kListSlots = 8;    // submissions in flight per context
constexpr int kEdgeBlocks = 32;          // page-locked blocks for outlines' edges ...
constexpr size_t kEdgeBlockBytes = (size_t) 1 << 20;     // ... of 37 000 edges each
constexpr int kEdgeEpochEvents = 16;
constexpr uint32_t kEdgeEpochPolygons = 4;       // an event behind every so many polygons

struct FrameArena;

// contexts alive per device in this process (see flutl_frame_stage: the lanes)
static std::atomic<int> g_contexts[16];

struct flutl_ctx {
  int device;
  hipStream_t stream;
  uint8_t *atlas;      // device
  size_t atlas_cap;    // allocated bytes
  size_t atlas_used;   // bytes in use (incl. leading slack)
  void *frame;         // device staging frame for render_to_host
  size_t frame_cap;
  void *listbuf;       // reusable device buffer for transient display lists
  size_t listbuf_cap;
  list_slot slots[kListSlots];
  uint32_t next_slot;
  uint8_t *stage;      // page-locked staging for atlas uploads (no wait per entry)
  size_t stage_cap, stage_used;
  uint64_t *delta_segments;   // device: segments (flutl_delta_segment_bytes) the delta read-backs wrote
  void *scan;          // device scratch of the polygon scan converter (edges + its own)
  size_t scan_cap;
  uint32_t *scan_counters;     // device: [0] polygons the sequential converter took
  // (ONE stream per context.  Outlines had a stream of their own for a while: a wait of the
  // main stream for another queue's kernel costs 10-15 us each time, tools/stream_trace.py;
  // DESIGN 5.7.)
  // An outline's edges stay in page-locked host memory: scan_prepare_kernel reads them from
  // there over PCIe (7 us more than from HBM for a 12 000-edge outline) -- no upload on
  // another queue and no wait for it.  The memory is an arena of blocks
  // (flutl_ctx_edge_block_acquire: the element's stroke workers write their polygon's edges
  // straight into one, so nobody copies them on the thread that feeds the device;
  // flutl_atlas_put_polygon takes a block of its own for edges that come from elsewhere).
  // A block is free again when the kernels that read it are done: every few polygons an
  // event is recorded behind them (an "epoch"), looked at only when blocks run short.
  uint8_t *edge_arena;
  std::mutex *edge_lock;        // block states (workers acquire / release; the device's owner queues)
  int8_t edge_state[kEdgeBlocks];       // 0 free, 1 out with a caller, 2 queued (edge_epoch)
  uint32_t edge_epoch[kEdgeBlocks];
  hipEvent_t epoch_done[kEdgeEpochEvents];
  uint32_t epoch_cur;           // epoch of the polygons being queued now
  uint32_t epoch_known;         // epochs below this one are known complete
  uint32_t epoch_uses;          // polygons queued in epoch_cur
  // A staged submission (flutl_frame_stage): its list lies in the slot's page-locked half and
  // is still to go up -- carried by the next scan conversion's launch (its last workgroup
  // copies it: no copy operation of its own on the stream, 4.5 us + a gap per cue), or by a
  // copy when flutl_frame_submit_staged finds it still pending.
  struct Staged {
    bool valid, copy_pending;
    flutl_frame *frame;
    int slot;
    size_t rb, pb, ob, copy_bytes;
    uint32_t n_regions, n_prims, n_order, y0, y1, now[4];
    int lane;
  } staged;
  std::vector<FrameArena *> frame_arenas;
  // The outlines' scan conversions are queued on a stream of their own: they hang on nothing
  // the compositor does -- an atlas entry is written once, into bytes nothing has read yet --
  // so cue k + 1's conversion runs beside cue k's compositor launch, whose stores wait for
  // PCIe (tools/overlap_bench.py: 48.7 us per pair in turn, 34.8 side by side).  scan_gen:
  // entries queued there; what each compositor lane has been made to wait for: atlas_join.
  // FLUTL_ONE_STREAM=1: everything on `stream`, as before round 4.
  hipStream_t scan_stream;
  hipEvent_t scan_ev;
  uint32_t scan_gen;            // atlas entries queued on the scan stream (one count per entry)
  uint32_t scan_ev_gen;         // ... what scan_ev stands behind
  uint32_t scan_joined[2];      // ... what each compositor lane has been made to wait for
  bool list_copy;               // FLUTL_LIST_COPY=1 (see flutl_frame_submit_staged)
  // Frames are submitted to two streams in turn ("lanes"; lane 0 is `stream`, where everything
  // else of the context is queued): a DELTA launch spends a third of its time waiting for its
  // stores to cross PCIe, the next frame's launch composites meanwhile
  // (tools/delta_two_streams_bench.py: 23.4 -> 15.1 us per launch).  What orders the lanes:
  // a frame's own event when it changes lanes (flutl_frame.lane), an event behind lane 0's
  // atlas uploads before a launch on lane 1 reads them (upload_gen), the scan stream's event
  // per lane (scan_joined); the atlas starting over or growing drains all of them.  Each lane
  // has a park buffer of its own.  FLUTL_ONE_LANE=1: lane 0 only.
  hipStream_t stream2;
  hipEvent_t lane_ev;
  void *frame2;
  size_t frame2_cap;
  uint32_t n_staged;            // frames staged on alternating lanes so far: the next one's lane is its low bit
  uint32_t upload_gen, upload_seen2;
  uint32_t staged_scan_gen, staged_upload_gen;  // scan_gen / upload_gen when the last frame was staged
  bool lanes_allowed;           // (not FLUTL_ONE_LANE)
  uint32_t alone_run;           // frames staged in a row while this was the device's only context here
  uint32_t lanes_after;         // ... of which so many before the second lane is used
  // FLUTL_API_TIMING=1: where flutl_frame_submit's time goes on the calling thread (ns)
  bool api_timing;
  uint64_t t_sub[6], n_sub;
};

// Frames made together (flutl_ctx_reserve_frames): ONE anonymous mapping advised to use huge
// pages, populated and registered with the device in one go, and one device allocation for the
// twins.  Page-locking a 4K frame of its own costs 5.5 ms (14 ms in a process whose other
// threads contend for the address space lock); eighteen of them this way cost 35 ms together
// (tools/microbench/pin_cost.cpp).  Counted: the frames may outlive the context.
struct FrameArena {
  std::atomic<int> refs;
  std::mutex lock;
  int device;
  void *map;
  size_t map_bytes;
  char *host, *twin;
  size_t block_bytes;
  uint32_t count;
  std::vector<uint8_t> state;   // per block: 0 free and zero, 1 free, 2 in a frame
};

static void frame_arena_unref (FrameArena *a)
{
  if (!a || a->refs.fetch_sub (1, std::memory_order_acq_rel) != 1)
    return;
  (void) hipSetDevice (a->device);
  if (a->host)
    (void) hipHostUnregister (a->host);
  if (a->map)
    munmap (a->map, a->map_bytes);
  if (a->twin)
    (void) hipFree (a->twin);
  delete a;
}

struct flutl_frame {
  int device;
  uint32_t width, height;
  FrameArena *arena;   // where host and twin come from (NULL: allocations of the frame's own)
  uint32_t arena_block;
  int lane;            // the compositor lane of the context its last operation was queued on
  void *host;          // page-locked, device mapped
  void *host_dev;      // its device address
  void *twin;          // device
  hipEvent_t done;
  // (atomics: flutl_frame_wait runs on whatever thread maps the buffer -- a tee or a
  // shared sub-memory lets two threads wait for one frame at once)
  std::atomic<bool> pending;   // a submission has been queued since the last wait
  std::atomic<int> error;      // sticky: a submission failed, the content is undefined
  std::atomic<bool> written;   // the host block was written into by its consumer (flutl_frame_mark_written)
  uint32_t ext[4];     // x0, y0, x1, y1: what the frame holds lies inside (empty: nothing)
};

extern "C" int flutl_abi_version (void) { return FLUTL_ABI_VERSION; }

extern "C" int flutl_device_count (void)
{
  int n = 0;
  if (!hip_ok (hipGetDeviceCount (&n)))
    return 0;
  return n;
}

extern "C" const char *flutl_strerror (int code)
{
  switch (code) {
    case FLUTL_OK: return "ok";
    case FLUTL_EINVAL: return "invalid argument or malformed display list";
    case FLUTL_ENOMEM: return "out of memory";
    case FLUTL_ENODEV: return "no usable HIP device / HIP runtime error";
    case FLUTL_ERANGE: return "primitive reads outside the atlas";
    default: return "unknown error";
  }
}

extern "C" int flutl_list_validate (uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, size_t atlas_bytes)
{
  if (width == 0 || height == 0 || width > 32767 || height > 32767)
    return FLUTL_EINVAL;
  if ((n_regions && !regions) || (n_prims && !prims))
    return FLUTL_EINVAL;
  for (uint32_t r = 0; r < n_regions; r++) {
    const flutl_region &rg = regions[r];
    if ((uint64_t) rg.first + rg.count > n_prims)
      return FLUTL_EINVAL;
    if (rg.x0 < 0 || rg.y0 < 0 || rg.x1 > (int) width || rg.y1 > (int) height ||
        rg.x0 > rg.x1 || rg.y0 > rg.y1)
      return FLUTL_EINVAL;
    int depth = 0;
    for (uint32_t i = rg.first; i < rg.first + rg.count; i++) {
      const flutl_prim &p = prims[i];
      if (p.cx0 > p.cx1 || p.cy0 > p.cy1 || p.cx0 < rg.x0 || p.cy0 < rg.y0 ||
          p.cx1 > rg.x1 || p.cy1 > rg.y1)
        return FLUTL_EINVAL;
      if (p.rule != FLUTL_RULE_PIXMAN && p.rule != FLUTL_RULE_LERP)
        return FLUTL_EINVAL;
      if (p.op == FLUTL_OP_RECT || p.op == FLUTL_OP_MASK) {
        // the kernel relies on a valid premultiplied colour (channel <= alpha):
        // it is what makes its non-saturating adds exact
        const uint32_t a = p.color >> 24;
        if (((p.color >> 16) & 255u) > a || ((p.color >> 8) & 255u) > a ||
            (p.color & 255u) > a)
          return FLUTL_EINVAL;
      }
      switch (p.op) {
        case FLUTL_OP_RECT:
          if (p.a > p.c || p.b > p.d)
            return FLUTL_EINVAL;
          break;
        case FLUTL_OP_MASK:
        case FLUTL_OP_IMAGE: {
          const int64_t bpp = p.op == FLUTL_OP_IMAGE ? 4 : 1;
          if (p.a < 0 || p.b <= 0)
            return FLUTL_EINVAL;
          if (p.op == FLUTL_OP_IMAGE && ((p.a & 3) || (p.b & 3)))
            return FLUTL_EINVAL;
          if (p.op == FLUTL_OP_MASK && (p.param & ~FLUTL_MASK_XCLIP))
            return FLUTL_EINVAL;
          // the kernel forms (row - d) * stride with a signed 24-bit multiply
          if (p.op == FLUTL_OP_MASK && (p.b > 0x7fffff || p.d < -0x400000))
            return FLUTL_ERANGE;
          if (p.cx0 == p.cx1 || p.cy0 == p.cy1)
            break;
          if (p.cx0 < p.c || p.cy0 < p.d)
            return FLUTL_ERANGE;
          if ((int64_t) (p.cx1 - p.c) * bpp > p.b)
            return FLUTL_ERANGE;
          const int64_t lo = (int64_t) p.a + (int64_t) (p.cy0 - p.d) * p.b +
              (int64_t) (p.cx0 - p.c) * bpp;
          const int64_t hi = (int64_t) p.a + (int64_t) (p.cy1 - 1 - p.d) * p.b +
              (int64_t) (p.cx1 - p.c) * bpp;
          // the kernel reads 4-byte windows that may start up to 3 bytes
          // before / end 3 bytes after the bbox: keep them inside the atlas
          if (lo < (int64_t) kAtlasSlack ||
              hi + (int64_t) kAtlasSlack > (int64_t) atlas_bytes)
            return FLUTL_ERANGE;
          break;
        }
        case FLUTL_OP_GROUP_BEGIN:
          if (depth != 0)
            return FLUTL_EINVAL;
          depth = 1;
          break;
        case FLUTL_OP_GROUP_END:
          if (depth != 1 || p.param > 255)
            return FLUTL_EINVAL;
          depth = 0;
          break;
        default:
          return FLUTL_EINVAL;
      }
    }
    if (depth != 0)
      return FLUTL_EINVAL;
  }
  return FLUTL_OK;
}

extern "C" int flutl_blur_kernel (int radius, double sigma, int32_t *out)
{
  if (radius < 0 || radius > FLUTL_BLUR_MAX_RADIUS || !out || !(sigma > 0.0))
    return FLUTL_EINVAL;
  const double scale2 = 2.0 * sigma * sigma;
  const double scale1 = 1.0 / (M_PI * scale2);
  const int size = 2 * radius + 1;
  std::vector<double> acc ((size_t) size * size);
  double sum = 0.0;
  int i = 0;
  for (int x = -radius; x <= radius; ++x)
    for (int y = -radius; y <= radius; ++y, ++i) {
      acc[i] = scale1 * exp (-((double) (x * x) + (double) (y * y)) / scale2);
      sum += acc[i];
    }
  for (i = 0; i < size * size; ++i)
    out[i] = (int32_t) ((acc[i] / sum) * 65536.0);  // pixman_double_to_fixed
  return FLUTL_OK;
}

extern "C" uint32_t flutl_row_order (uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, uint32_t row0, uint32_t row1, uint16_t *order_out)
{
  if (!order_out || row0 >= row1 || row1 > height || height > 32767)
    return 0;
  const uint32_t base = row0 / 16u, n = (row1 + 15u) / 16u - base;
  // (on the stack, sorted without a temporary buffer: this runs once per frame on the thread
  // that feeds the device -- a vector and std::stable_sort's buffer were two allocations and
  // 3 us of its 24 per frame)
  uint64_t key[2048];           // height <= 32767: at most 2048 supertile rows
  for (uint32_t k = 0; k < n; k++)
    key[k] = 0;
  for (uint32_t r = 0; r < n_regions; r++) {
    const flutl_region &rg = regions[r];
    for (uint32_t i = rg.first; i < rg.first + rg.count && i < n_prims; i++) {
      const flutl_prim &p = prims[i];
      if (p.cx1 <= p.cx0 || p.cy1 <= p.cy0)
        continue;
      // tiles of 64 px the primitive crosses in a row, times what a tile of it costs
      const uint32_t tiles = (uint32_t) (p.cx1 - 1) / 64u - (uint32_t) p.cx0 / 64u + 1u;
      const uint32_t cost = p.op == FLUTL_OP_MASK ? 8u : p.op == FLUTL_OP_RECT ? 1u : 6u;
      uint32_t a = (uint32_t) p.cy0 / 16u, b = (uint32_t) (p.cy1 - 1) / 16u;
      a = a < base ? base : a;
      b = b >= base + n ? base + n - 1 : b;
      for (uint32_t row = a; row <= b && row >= base; row++)
        key[row - base] += tiles * cost;
    }
  }
  // heaviest first, equal rows top to bottom: one ascending sort of (~weight, row)
  for (uint32_t k = 0; k < n; k++)
    key[k] = ((uint64_t) (0xffffffffu - (uint32_t) (key[k] > 0xffffffffull ? 0xffffffffull : key[k])) << 16) | k;
  std::sort (key, key + n);
  for (uint32_t k = 0; k < n; k++)
    order_out[k] = (uint16_t) (key[k] & 0xffffu);
  (void) width;
  return n;
}

// ---- context -------------------------------------------------------------

static int atlas_reserve (flutl_ctx *c, size_t need);
static void stage_reserve (flutl_ctx *c);

extern "C" int flutl_ctx_create (int device, flutl_ctx **out)
{
  if (!out)
    return FLUTL_EINVAL;
  *out = nullptr;
  int n = 0;
  if (!hip_ok (hipGetDeviceCount (&n)) || device < 0 || device >= n)
    return FLUTL_ENODEV;
  if (!hip_ok (hipSetDevice (device)))
    return FLUTL_ENODEV;
  flutl_ctx *c = new (std::nothrow) flutl_ctx ();
  if (!c)
    return FLUTL_ENOMEM;
  c->device = device;
  if (!hip_ok (hipStreamCreateWithFlags (&c->stream, hipStreamNonBlocking))) {
    delete c;
    return FLUTL_ENODEV;
  }
  c->scan_stream = nullptr;
  c->scan_ev = nullptr;
  c->scan_gen = c->scan_ev_gen = c->scan_joined[0] = c->scan_joined[1] = 0;
  c->stream2 = nullptr;
  c->lane_ev = nullptr;
  c->frame2 = nullptr;
  c->frame2_cap = 0;
  c->n_staged = 0;
  c->upload_gen = c->upload_seen2 = 0;
  c->staged_scan_gen = c->staged_upload_gen = 0;
  {
    const char *one = getenv ("FLUTL_ONE_LANE");
    c->lanes_allowed = !(one && *one && *one != '0');
    c->alone_run = 0;
    // (FLUTL_LANES_AFTER=n: after n frames instead of 8; the tests' way to lanes at once)
    const char *after = getenv ("FLUTL_LANES_AFTER");
    c->lanes_after = after && *after ? (uint32_t) atoi (after) : 8u;
    if (c->lanes_allowed) {
      if (!hip_ok (hipStreamCreateWithFlags (&c->stream2, hipStreamNonBlocking)) ||
          !hip_ok (hipEventCreateWithFlags (&c->lane_ev, hipEventDisableTiming))) {
        if (c->stream2)
          (void) hipStreamDestroy (c->stream2);
        c->stream2 = nullptr;
        c->lane_ev = nullptr;
        c->lanes_allowed = false;
      }
    }
  }
  {
    const char *lc = getenv ("FLUTL_LIST_COPY");
    c->list_copy = lc && *lc && *lc != '0';
  }
  {
    const char *one = getenv ("FLUTL_ONE_STREAM");
    if (!(one && *one && *one != '0')) {
      if (!hip_ok (hipStreamCreateWithFlags (&c->scan_stream, hipStreamNonBlocking)) ||
          !hip_ok (hipEventCreateWithFlags (&c->scan_ev, hipEventDisableTiming))) {
        if (c->scan_ev)
          (void) hipEventDestroy (c->scan_ev);
        if (c->scan_stream)
          (void) hipStreamDestroy (c->scan_stream);
        if (c->stream2) {
          (void) hipEventDestroy (c->lane_ev);
          (void) hipStreamDestroy (c->stream2);
        }
        (void) hipStreamDestroy (c->stream);
        delete c;
        return FLUTL_ENODEV;
      }
    }
  }
  c->atlas = nullptr;
  c->atlas_cap = 0;
  c->atlas_used = kAtlasSlack;
  c->frame = nullptr;
  c->frame_cap = 0;
  c->listbuf = nullptr;
  c->listbuf_cap = 0;
  for (auto &sl : c->slots)
    sl = list_slot { nullptr, nullptr, nullptr, 0, nullptr, false };
  c->next_slot = 0;
  c->stage = nullptr;
  c->stage_cap = c->stage_used = 0;
  c->delta_segments = nullptr;
  c->scan = nullptr;
  c->scan_cap = 0;
  c->scan_counters = nullptr;
  c->edge_arena = nullptr;
  c->edge_lock = new (std::nothrow) std::mutex ();
  memset (c->edge_state, 0, sizeof c->edge_state);
  memset (c->edge_epoch, 0, sizeof c->edge_epoch);
  c->epoch_cur = c->epoch_known = 1;
  c->epoch_uses = 0;
  memset (&c->staged, 0, sizeof c->staged);
  c->api_timing = getenv ("FLUTL_API_TIMING") != nullptr;
  memset (c->t_sub, 0, sizeof c->t_sub);
  c->n_sub = 0;
  for (auto &e : c->epoch_done)
    e = nullptr;
  {
    bool ok = c->edge_lock != nullptr;
    for (auto &e : c->epoch_done)
      ok = ok && hip_ok (hipEventCreateWithFlags (&e, hipEventDisableTiming));
    // (without the arena an outline's edges are uploaded and waited for instead)
    if (!ok || !hip_ok (hipHostMalloc (reinterpret_cast<void **> (&c->edge_arena),
            kEdgeBlocks * kEdgeBlockBytes, hipHostMallocDefault)))
      c->edge_arena = nullptr;
  }
  // The first masks of a stream come with its first frame, and the caller's
  // recording thread may be waiting on whoever uploads them: the atlas and the
  // upload staging are made now (10 ms of allocation calls), not then.  (Failure
  // is not an error here: both are made on demand as well.)
  (void) atlas_reserve (c, (size_t) 1 << 20);
  (void) stage_reserve (c);
  // (... and the runtime sets up its path for copies of 256 KB and more on the
  // first such copy, 8 ms measured: zeros over the zeros of the fresh atlas)
  if (c->atlas && c->stage) {
    memset (c->stage, 0, (size_t) 1 << 20);
    (void) hipMemcpyAsync (c->atlas, c->stage, (size_t) 1 << 20, hipMemcpyHostToDevice, c->stream);
    (void) hipStreamSynchronize (c->stream);
  }
  // (... and so is the scan converter's scratch: the first cue of a stream brings the
  // outlines of all its regions)
  if (hip_ok (hipMalloc (&c->scan, (size_t) 8 << 20)) &&
      hip_ok (hipMemsetAsync (c->scan, 0, flutl_scan_zero_bytes (), c->stream)))    // (the converter's status words, the rows' counts)
    c->scan_cap = (size_t) 8 << 20;
  else {
    if (c->scan)
      (void) hipFree (c->scan);
    c->scan = nullptr;
  }
  if (hip_ok (hipMalloc (reinterpret_cast<void **> (&c->scan_counters), 4 * sizeof (uint32_t))))
    (void) hipMemsetAsync (c->scan_counters, 0, 4 * sizeof (uint32_t), c->stream);
  else
    c->scan_counters = nullptr;
  if (c->scan) {
    // (the runtime loads a kernel's code when it is first launched, a millisecond each:
    // a triangle through the converter now, not with the first cue's outlines)
    static const int32_t tri[21] = { 256, 256, 256, 1536, 256, 1536, 1,
      256, 256, 1536, 1536, 256, 1536, -1, 0, 0, 0, 0, 0, 0, 0 };
    const int32_t box[4] = { 1, 1, 7, 7 };
    uint8_t *base = static_cast<uint8_t *> (c->scan);
    const size_t own = flutl_scan_scratch_bytes (2, 6, 6);
    if (hip_ok (hipMemcpyAsync (base + own, tri, sizeof tri, hipMemcpyHostToDevice, c->stream)))
      (void) flutl_launch_scan_counted (c->stream, reinterpret_cast<const int32_t *> (base + own), 2,
          box, 0, base, base + own + 256, 16, 0 | 16, nullptr);
  }
  (void) hipStreamSynchronize (c->stream);
  g_contexts[device & 15].fetch_add (1, std::memory_order_relaxed);
  *out = c;
  return FLUTL_OK;
}

// the stream the scan conversions are queued on
static inline hipStream_t scan_stream_of (flutl_ctx *c)
{
  return c->scan_stream ? c->scan_stream : c->stream;
}

static inline hipStream_t lane_stream (flutl_ctx *c, int lane)
{
  return lane && c->stream2 ? c->stream2 : c->stream;
}

// A compositor lane behind every atlas entry queued on the scan stream so far: before anything
// on it that reads the atlas (or writes where an entry was, or frees what a conversion uses)
static bool atlas_join_lane (flutl_ctx *c, int lane)
{
  if (!c->scan_stream || c->scan_joined[lane] == c->scan_gen)
    return true;
  if (c->scan_ev_gen != c->scan_gen) {
    if (!hip_ok (hipEventRecord (c->scan_ev, c->scan_stream))) {
      const bool ok = hip_ok (hipStreamSynchronize (c->scan_stream));
      c->scan_joined[0] = c->scan_joined[1] = c->scan_ev_gen = c->scan_gen;
      return ok;
    }
    c->scan_ev_gen = c->scan_gen;
  }
  if (!hip_ok (hipStreamWaitEvent (lane_stream (c, lane), c->scan_ev, 0)))
    return hip_ok (hipStreamSynchronize (c->scan_stream));
  c->scan_joined[lane] = c->scan_gen;
  return true;
}

static bool atlas_join (flutl_ctx *c) { return atlas_join_lane (c, 0); }

// lane 1 behind the atlas uploads queued on lane 0 so far (flutl_atlas_put, _put_blurred)
static bool uploads_join_lane (flutl_ctx *c, int lane)
{
  if (lane == 0 || !c->stream2 || c->upload_seen2 == c->upload_gen)
    return true;
  if (!hip_ok (hipEventRecord (c->lane_ev, c->stream)) ||
      !hip_ok (hipStreamWaitEvent (c->stream2, c->lane_ev, 0)))
    return hip_ok (hipStreamSynchronize (c->stream));
  c->upload_seen2 = c->upload_gen;
  return true;
}

// every stream of the context drained
static bool sync_streams (flutl_ctx *c)
{
  bool ok = true;
  if (c->scan_stream) {
    ok = hip_ok (hipStreamSynchronize (c->scan_stream));
    c->scan_joined[0] = c->scan_joined[1] = c->scan_ev_gen = c->scan_gen;
  }
  if (c->stream2) {
    ok = hip_ok (hipStreamSynchronize (c->stream2)) && ok;
    c->upload_seen2 = c->upload_gen;
  }
  return hip_ok (hipStreamSynchronize (c->stream)) && ok;
}

extern "C" void flutl_ctx_destroy (flutl_ctx *c)
{
  if (!c)
    return;
  if (c->api_timing && c->n_sub)
    fprintf (stderr, "flutl api timing: %llu submits; per submit us: validate+order %.1f slot-wait %.1f "
        "list-copy %.1f memcpyAsync %.1f launch %.1f events %.1f\n", (unsigned long long) c->n_sub,
        c->t_sub[0] / 1e3 / c->n_sub, c->t_sub[1] / 1e3 / c->n_sub, c->t_sub[2] / 1e3 / c->n_sub,
        c->t_sub[3] / 1e3 / c->n_sub, c->t_sub[4] / 1e3 / c->n_sub, c->t_sub[5] / 1e3 / c->n_sub);
  (void) hipSetDevice (c->device);
  (void) sync_streams (c);
  g_contexts[c->device & 15].fetch_sub (1, std::memory_order_relaxed);
  if (c->atlas)
    (void) hipFree (c->atlas);
  if (c->frame)
    (void) hipFree (c->frame);
  if (c->frame2)
    (void) hipFree (c->frame2);
  if (c->listbuf)
    (void) hipFree (c->listbuf);
  for (auto &sl : c->slots) {
    if (sl.host)
      (void) hipHostFree (sl.host);
    if (sl.dev)
      (void) hipFree (sl.dev);
    if (sl.done)
      (void) hipEventDestroy (sl.done);
  }
  if (c->stage)
    (void) hipHostFree (c->stage);
  if (c->delta_segments)
    (void) hipFree (c->delta_segments);
  if (c->scan)
    (void) hipFree (c->scan);
  if (c->scan_counters)
    (void) hipFree (c->scan_counters);
  for (auto &e : c->epoch_done)
    if (e)
      (void) hipEventDestroy (e);
  if (c->edge_arena)
    (void) hipHostFree (c->edge_arena);
  delete c->edge_lock;
  for (FrameArena *a : c->frame_arenas)
    frame_arena_unref (a);
  if (c->scan_stream) {
    (void) hipEventDestroy (c->scan_ev);
    (void) hipStreamDestroy (c->scan_stream);
  }
  if (c->stream2) {
    (void) hipEventDestroy (c->lane_ev);
    (void) hipStreamDestroy (c->stream2);
  }
  (void) hipStreamDestroy (c->stream);
  delete c;
}

extern "C" int flutl_ctx_sync (flutl_ctx *c)
{
  if (!c)
    return FLUTL_EINVAL;
  if (!sync_streams (c))
    return FLUTL_ENODEV;
  c->stage_used = 0;       // every staged upload has landed
  if (c->edge_lock) {      // (... and every outline's edges have been read)
    std::lock_guard<std::mutex> g (*c->edge_lock);
    for (int k = 0; k < kEdgeBlocks; k++)
      if (c->edge_state[k] == 2)
        c->edge_state[k] = 0;
    if (c->epoch_uses) {
      c->epoch_cur++;
      c->epoch_uses = 0;
    }
    c->epoch_known = c->epoch_cur;
  }
  return FLUTL_OK;
}

static void stage_reserve (flutl_ctx *c)
{
  constexpr size_t kStageBytes = (size_t) 8 << 20;
  if (!c->stage && hip_ok (hipHostMalloc (reinterpret_cast<void **> (&c->stage), kStageBytes,
          hipHostMallocDefault)))
    c->stage_cap = kStageBytes;
}

static int atlas_reserve (flutl_ctx *c, size_t need)
{
  if (need <= c->atlas_cap)
    return FLUTL_OK;
  // (growing costs a device allocation, a copy and a drained stream, on the frame
  // that brings a cue's first masks: start where 4K text with outlines ends up;
  // HBM is not what is short on this device)
  // (320 MB: the element starts its mask caches over at 256 MB of atlas, so a stream never
  // grows it -- a 4K stream of outlined text passes 64 MB after 116 cues, and growing is a
  // device allocation, a copy and a drained stream in the middle of it)
  size_t cap = c->atlas_cap ? c->atlas_cap : (size_t) 320 << 20;
  while (cap < need)
    cap *= 2;
  uint8_t *n = nullptr;
  if (!hip_ok (hipMalloc (reinterpret_cast<void **> (&n), cap)))
    return FLUTL_ENOMEM;
  // (conversions in flight write the old buffer, launches on either lane read it)
  (void) sync_streams (c);
  if (!hip_ok (hipMemsetAsync (n, 0, cap, c->stream))) {
    (void) hipFree (n);
    return FLUTL_ENODEV;
  }
  // lists in flight may still read the old buffer: drain first
  if ((c->atlas && !hip_ok (hipMemcpyAsync (n, c->atlas, c->atlas_used, hipMemcpyDeviceToDevice,
              c->stream))) ||
      !hip_ok (hipStreamSynchronize (c->stream))) {
    (void) hipFree (n);
    return FLUTL_ENODEV;
  }
  if (c->atlas)
    (void) hipFree (c->atlas);
  c->atlas = n;
  c->atlas_cap = cap;
  return FLUTL_OK;
}

extern "C" int flutl_atlas_put (flutl_ctx *c, const void *src, size_t bytes,
    uint32_t *offset_out)
{
  if (!c || !src || !offset_out || bytes == 0)
    return FLUTL_EINVAL;
  (void) hipSetDevice (c->device);
  const size_t off = (c->atlas_used + kAtlasAlign - 1) & ~(kAtlasAlign - 1);
  const size_t end = off + bytes;
  if (end + kAtlasSlack > 0x7fffffffu)
    return FLUTL_ERANGE;
  int rc = atlas_reserve (c, end + kAtlasSlack);
  if (rc)
    return rc;
  // src may be freed by the caller right after return: small entries (glyph
  // masks, outline masks) go through page-locked staging and are not waited
  // for -- the stream orders them before whatever reads them; the staging
  // memory is reclaimed when the stream is next drained
  stage_reserve (c);
  if (c->stage && bytes <= c->stage_cap / 2) {
    if (c->stage_used + bytes > c->stage_cap) {
      if (!hip_ok (hipStreamSynchronize (c->stream)))
        return FLUTL_ENODEV;
      c->stage_used = 0;
    }
    memcpy (c->stage + c->stage_used, src, bytes);
    if (!hip_ok (hipMemcpyAsync (c->atlas + off, c->stage + c->stage_used, bytes,
            hipMemcpyHostToDevice, c->stream)))
      return FLUTL_ENODEV;
    c->stage_used += (bytes + 63) & ~(size_t) 63;
  } else {
    if (!hip_ok (hipMemcpyAsync (c->atlas + off, src, bytes, hipMemcpyHostToDevice,
            c->stream)) ||
        !hip_ok (hipStreamSynchronize (c->stream)))
      return FLUTL_ENODEV;
    c->stage_used = 0;
  }
  c->atlas_used = end;
  c->upload_gen++;
  *offset_out = (uint32_t) off;
  return FLUTL_OK;
}

static int upload_list (flutl_ctx *c, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, flutl_region **rd, flutl_prim **pd);

// ---- the arena of page-locked edge blocks (see flutl_ctx) ---------------------------------
// epochs known complete -> their blocks free.  wait: block until at least the oldest
// outstanding epoch is done (its event is recorded first if it is the current one).
static void edge_blocks_reclaim (flutl_ctx *c, bool wait)
{
  // (the caller holds edge_lock)
  if (wait && c->epoch_known == c->epoch_cur && c->epoch_uses) {
    if (hip_ok (hipEventRecord (c->epoch_done[c->epoch_cur % kEdgeEpochEvents], scan_stream_of (c)))) {
      c->epoch_cur++;
      c->epoch_uses = 0;
    }
  }
  while (c->epoch_known < c->epoch_cur) {
    hipEvent_t ev = c->epoch_done[c->epoch_known % kEdgeEpochEvents];
    if (wait) {
      if (!hip_ok (hipEventSynchronize (ev)))
        break;
      wait = false;             // (one epoch waited for; the rest only if they are done already)
    } else if (hipEventQuery (ev) != hipSuccess) {
      break;
    }
    c->epoch_known++;
  }
  for (int k = 0; k < kEdgeBlocks; k++)
    if (c->edge_state[k] == 2 && c->edge_epoch[k] < c->epoch_known)
      c->edge_state[k] = 0;
}

static void *edge_block_take (flutl_ctx *c, bool may_wait)
{
  for (int pass = 0; pass < 3; pass++) {
    for (int k = 0; k < kEdgeBlocks; k++)
      if (c->edge_state[k] == 0) {
        c->edge_state[k] = 1;
        return c->edge_arena + (size_t) k * kEdgeBlockBytes;
      }
    if (pass == 2 || (pass == 1 && !may_wait))
      break;
    edge_blocks_reclaim (c, pass == 1);
  }
  return nullptr;
}

extern "C" int flutl_ctx_edge_block_acquire (flutl_ctx *c, size_t bytes, void **block_out)
{
  if (!c || !block_out)
    return FLUTL_EINVAL;
  *block_out = nullptr;
  if (!c->edge_arena || bytes > kEdgeBlockBytes)
    return FLUTL_ENOMEM;
  std::lock_guard<std::mutex> g (*c->edge_lock);
  // (a worker thread never waits for the device: without a free block its edges stay where
  // they are and are copied by flutl_atlas_put_polygon)
  *block_out = edge_block_take (c, false);
  return *block_out ? FLUTL_OK : FLUTL_ENOMEM;
}

extern "C" void flutl_ctx_edge_block_release (flutl_ctx *c, void *block)
{
  if (!c || !block || !c->edge_arena)
    return;
  const size_t off = static_cast<uint8_t *> (block) - c->edge_arena;
  if (off >= kEdgeBlocks * kEdgeBlockBytes || off % kEdgeBlockBytes)
    return;
  std::lock_guard<std::mutex> g (*c->edge_lock);
  if (c->edge_state[off / kEdgeBlockBytes] == 1)
    c->edge_state[off / kEdgeBlockBytes] = 0;
}

extern "C" size_t flutl_polygon_entry_bytes (uint32_t w, uint32_t h)
{
  return 4 + ((size_t) w + 4) * h;
}

extern "C" int flutl_atlas_put_polygon (flutl_ctx *c, const int32_t *edges, uint32_t n_edges,
    const int32_t box[4], int even_odd, uint32_t *offset_out)
{
  if (!c || !offset_out || !box || (n_edges && !edges) || box[2] <= box[0] || box[3] <= box[1] ||
      box[0] < 0 || box[1] < 0 || box[2] - box[0] > 32767 || box[3] - box[1] > 32767)
    return FLUTL_EINVAL;
  (void) hipSetDevice (c->device);
  // (a block of the arena is the context's from here on, whatever happens: one that never
  // gets queued goes back at once)
  struct BlockGuard {
    flutl_ctx *c;
    const void *p;
    ~BlockGuard () { flutl_ctx_edge_block_release (c, const_cast<void *> (p)); }
  } guard { c, edges };
  const uint32_t w = (uint32_t) (box[2] - box[0]), h = (uint32_t) (box[3] - box[1]);
  const size_t bytes = flutl_polygon_entry_bytes (w, h);
  const size_t off = (c->atlas_used + kAtlasAlign - 1) & ~(kAtlasAlign - 1);
  const size_t end = off + bytes;
  if (end + kAtlasSlack > 0x7fffffffu)
    return FLUTL_ERANGE;
  int rc = atlas_reserve (c, end + kAtlasSlack);
  if (rc)
    return rc;
  // scratch: the converter's own (its status words first: they are zero between calls),
  // then the edges
  const size_t edge_bytes = ((size_t) n_edges * 7 * sizeof (int32_t) + 255) & ~(size_t) 255;
  // (the converter takes boxes of up to 4096 pixels' width: a wider entry -- a line of 8K
  // text -- is converted in column windows, which is the same coverage: the conversion of
  // a window is the crop of the conversion of the whole, tests/test_gpu_scan.py)
  constexpr uint32_t kWindow = 4096;
  const size_t own_bytes = flutl_scan_scratch_bytes (n_edges, w < kWindow ? w : kWindow, h);
  const size_t need = edge_bytes + own_bytes;
  hipStream_t ms = scan_stream_of (c);
  if (need > c->scan_cap) {
    // (whatever still reads the old one is ordered before the free by the sync)
    if (c->scan) {
      if (!hip_ok (hipStreamSynchronize (ms)))
        return FLUTL_ENODEV;
      (void) hipFree (c->scan);
      c->scan = nullptr;
      c->scan_cap = 0;
    }
    size_t cap = need + need / 2;
    if (cap < ((size_t) 8 << 20))
      cap = (size_t) 8 << 20;   // (an outline of two 4K lines needs about 3 MB)
    if (!hip_ok (hipMalloc (&c->scan, cap)))
      return FLUTL_ENOMEM;
    if (!hip_ok (hipMemsetAsync (c->scan, 0, flutl_scan_zero_bytes (), ms))) {
      (void) hipFree (c->scan);
      c->scan = nullptr;
      return FLUTL_ENODEV;
    }
    c->scan_cap = cap;
  }
  if (!c->scan_counters) {
    if (!hip_ok (hipMalloc (reinterpret_cast<void **> (&c->scan_counters), 4 * sizeof (uint32_t))) ||
        !hip_ok (hipMemsetAsync (c->scan_counters, 0, 4 * sizeof (uint32_t), ms)))
      return FLUTL_ENOMEM;
  }
  // the edges' copy in device memory (made by the prepare kernel: the sequential sweep's)
  int32_t *edges_dev = reinterpret_cast<int32_t *> (static_cast<uint8_t *> (c->scan) + own_bytes);
  const int32_t *edges_src = edges_dev;         // what the prepare kernel reads
  int32_t *raw_copy = nullptr;
  if (n_edges) {
    const size_t eb = (size_t) n_edges * 7 * sizeof (int32_t);
    // edges in a block of the arena already (the caller's worker wrote them there), or
    // copied into one now
    int block = -1;
    if (c->edge_arena) {
      const uint8_t *e8 = reinterpret_cast<const uint8_t *> (edges);
      std::lock_guard<std::mutex> g (*c->edge_lock);
      if (e8 >= c->edge_arena && e8 < c->edge_arena + kEdgeBlocks * kEdgeBlockBytes) {
        block = (int) ((e8 - c->edge_arena) / kEdgeBlockBytes);
        if ((size_t) (e8 - c->edge_arena) % kEdgeBlockBytes || eb > kEdgeBlockBytes ||
            c->edge_state[block] != 1)
          return FLUTL_EINVAL;
        edges_src = edges;
      } else if (eb <= kEdgeBlockBytes) {
        uint8_t *at = static_cast<uint8_t *> (edge_block_take (c, true));
        if (at) {
          block = (int) ((at - c->edge_arena) / kEdgeBlockBytes);
          memcpy (at, edges, eb);
          edges_src = reinterpret_cast<const int32_t *> (at);
        }
      }
      if (block >= 0) {
        raw_copy = edges_dev;
        c->edge_state[block] = 2;
        c->edge_epoch[block] = c->epoch_cur;
      }
    }
    if (block < 0) {
      // (more edges than a block holds, or no arena: uploaded, and waited for)
      if (!hip_ok (hipMemcpyAsync (edges_dev, edges, eb, hipMemcpyHostToDevice, ms)) ||
          !hip_ok (hipStreamSynchronize (ms)))
        return FLUTL_ENODEV;
    }
  }
  // the entry: the coverage, and the zero bytes around and between its rows (mode 8); the
  // converter's status words are the first of its scratch and stay zero between calls (16)
  c->scan_gen++;
  if (n_edges == 0) {
    if (!hip_ok (hipMemsetAsync (c->atlas + off, 0, bytes, ms)))
      return FLUTL_ENODEV;
  } else {
    for (uint32_t x = 0; x < w && !rc; x += kWindow) {
      const uint32_t ww = w - x < kWindow ? w - x : kWindow;
      const int32_t wbox[4] = { box[0] + (int32_t) x, box[1], box[0] + (int32_t) (x + ww), box[3] };
      // (a window zeroes the 4 bytes after each of its rows; those of all but the last
      // window are the next window's first pixels, written behind it in stream order)
      const int pad = x == 0 ? 8 : (x + ww == w ? 32 : 0);
      // (the first window's prepare kernel makes the device copy; the others read that)
      // (a staged frame's list rides up with the first window's launch)
      const bool carry = x == 0 && c->staged.valid && c->staged.copy_pending;
      list_slot &csl = c->slots[c->staged.slot];
      rc = flutl_launch_scan_hosted (ms, x == 0 ? edges_src : edges_dev, x == 0 ? raw_copy : nullptr,
          n_edges, wbox, even_odd, c->scan, c->atlas + off + 4 + x, (size_t) w + 4, 0 | pad | 16,
          c->scan_counters, carry ? csl.host : nullptr, carry ? csl.dev : nullptr,
          carry ? c->staged.copy_bytes : 0);
      if (carry && rc == FLUTL_OK)
        c->staged.copy_pending = false;
    }
  }
  if (n_edges && c->edge_arena && edges_src != edges_dev) {
    // an event behind every few polygons: what frees their blocks (edge_blocks_reclaim)
    std::lock_guard<std::mutex> g (*c->edge_lock);
    if (++c->epoch_uses >= kEdgeEpochPolygons) {
      // (the event's slot comes round every kEdgeEpochEvents epochs: 64 polygons, twice
      // what the arena holds -- its earlier use has been seen complete long since)
      if (c->epoch_cur - c->epoch_known >= (uint32_t) kEdgeEpochEvents - 1)
        edge_blocks_reclaim (c, true);
      if (hip_ok (hipEventRecord (c->epoch_done[c->epoch_cur % kEdgeEpochEvents], ms))) {
        c->epoch_cur++;
        c->epoch_uses = 0;
      }
    }
  }
  if (rc)
    return rc;
  c->atlas_used = end;
  *offset_out = (uint32_t) off;
  return FLUTL_OK;
}

extern "C" uint32_t flutl_ctx_scan_fallbacks (flutl_ctx *c)
{
  uint32_t n = 0;
  if (!c || !c->scan_counters)
    return 0;
  (void) hipSetDevice (c->device);
  (void) atlas_join (c);
  if (!hip_ok (hipMemcpyAsync (&n, c->scan_counters, sizeof n, hipMemcpyDeviceToHost, c->stream)) ||
      !hip_ok (hipStreamSynchronize (c->stream)))
    return 0;
  return n;
}

extern "C" int flutl_atlas_put_blurred (flutl_ctx *c, uint32_t w, uint32_t h,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, int radius, double sigma, uint32_t distx, uint32_t disty,
    const int32_t *box, int rotate_cw, void *host_copy, uint32_t *offset_out)
{
  if (!c || !offset_out || w == 0 || h == 0 || w > 32766 || h > 32766 || radius < 0 ||
      radius > FLUTL_BLUR_MAX_RADIUS || distx > 127 || disty > 127)
    return FLUTL_EINVAL;
  int rc = flutl_list_validate (w, h, regions, n_regions, prims, n_prims,
      c->atlas_used + kAtlasSlack);
  if (rc)
    return rc;
  (void) hipSetDevice (c->device);
  (void) atlas_join (c);
  // a fractional blit position: the entry is the blurred surface resampled by
  // the constant bilinear weights, one pixel wider and higher, IN the box's
  // edge coverage.  rotate_cw: the 90-degree CTM of the vertical writing
  // modes; the entry is turned a quarter clockwise -- after the resample, which
  // is then in the surface's own axes, and before the box, which is in device
  // space.
  const bool resample = distx != 0 || disty != 0 || box != nullptr;
  const uint32_t rw = resample ? w + 1 : w, rh = resample ? h + 1 : h;   // before turning
  const uint32_t ew = rotate_cw ? rh : rw, eh = rotate_cw ? rw : rh;
  const size_t tight = (size_t) w * 4, bytes = tight * h;
  const size_t etight = (size_t) ew * 4, ebytes = etight * eh;
  // reserve the atlas entry first: growing the atlas moves it
  const size_t off = (c->atlas_used + kAtlasAlign - 1) & ~(kAtlasAlign - 1);
  const size_t end = off + ebytes;
  if (end + kAtlasSlack > 0x7fffffffu)
    return FLUTL_ERANGE;
  if ((rc = atlas_reserve (c, end + kAtlasSlack)))
    return rc;
  // temporary surface (and, when resampling, the blurred one) in the
  // context's staging frame
  // staging: [0] the composited surface, [1] its blur, [0] again the resample
  // when a turn follows (the composite is no longer needed then)
  const size_t rbytes = (size_t) rw * 4 * rh;
  const size_t need = (resample || rotate_cw) ? (rbytes > bytes ? rbytes : bytes) + bytes
                                              : bytes;
  if (need > c->frame_cap) {
    if (c->frame) {
      (void) hipStreamSynchronize (c->stream);
      (void) hipFree (c->frame);
      c->frame = nullptr;
      c->frame_cap = 0;
    }
    if (!hip_ok (hipMalloc (&c->frame, need)))
      return FLUTL_ENOMEM;
    c->frame_cap = need;
  }
  flutl_region *rd;
  flutl_prim *pd;
  if ((rc = upload_list (c, regions, n_regions, prims, n_prims, &rd, &pd)))
    return rc;
  if ((rc = flutl_launch_frame (c->stream, w, h, c->frame, tight, rd, n_regions, pd,
           c->atlas)))
    return rc;
  // Gaussian weights: host -> small device buffer
  std::vector<int32_t> weights ((size_t) (2 * radius + 1) * (2 * radius + 1));
  if ((rc = flutl_blur_kernel (radius, sigma, weights.data ())))
    return rc;
  const size_t wbytes = (size_t) (2 * radius + 1) * (2 * radius + 1) * 4;
  int32_t *wdev = nullptr;
  if (!hip_ok (hipMalloc (reinterpret_cast<void **> (&wdev), wbytes)))
    return FLUTL_ENOMEM;
  const size_t slot0 = rbytes > bytes ? rbytes : bytes;
  uint8_t *blurred = (resample || rotate_cw)
      ? static_cast<uint8_t *> (c->frame) + slot0 : c->atlas + off;
  bool ok = hip_ok (hipMemcpyAsync (wdev, weights.data (), wbytes, hipMemcpyHostToDevice, c->stream));
  if (ok)
    ok = flutl_launch_blur (c->stream, c->frame, blurred, w, h, tight, wdev,
        radius) == FLUTL_OK;
  if (ok && resample && !rotate_cw)
    ok = flutl_launch_resample (c->stream, blurred, c->atlas + off, w, h, tight, etight,
        distx, disty, box) == FLUTL_OK;
  if (ok && rotate_cw) {
    const uint8_t *turn_src = blurred;
    if (resample) {         // resample (no box) into staging slot 0, then turn that
      ok = flutl_launch_resample (c->stream, blurred, c->frame, w, h, tight, (size_t) rw * 4,
          distx, disty, nullptr) == FLUTL_OK;
      turn_src = static_cast<const uint8_t *> (c->frame);
    }
    if (ok)
      ok = flutl_launch_rotate_cw (c->stream, turn_src, c->atlas + off, rw, rh,
          (size_t) rw * 4, etight, box) == FLUTL_OK;
  }
  if (ok && host_copy)
    ok = hip_ok (hipMemcpyAsync (host_copy, c->atlas + off, ebytes, hipMemcpyDeviceToHost,
        c->stream));
  ok = hip_ok (hipStreamSynchronize (c->stream)) && ok;
  (void) hipFree (wdev);
  if (!ok)
    return FLUTL_ENODEV;
  c->atlas_used = end;
  *offset_out = (uint32_t) off;
  return FLUTL_OK;
}

extern "C" void flutl_atlas_reset (flutl_ctx *c)
{
  if (!c)
    return;
  // (the bytes are used again: whatever is written there next, on whichever stream, comes
  // behind everything queued so far on all of them -- drained: this is once per document, or
  // per 256 MB of masks)
  (void) hipSetDevice (c->device);
  (void) sync_streams (c);
  c->atlas_used = kAtlasSlack;
}

extern "C" size_t flutl_atlas_size (const flutl_ctx *c)
{
  return c ? c->atlas_used + kAtlasSlack : 0;
}

extern "C" int flutl_host_alloc (size_t bytes, void **out)
{
  if (!out || bytes == 0)
    return FLUTL_EINVAL;
  *out = nullptr;
  int n = 0;
  if (!hip_ok (hipGetDeviceCount (&n)) || n == 0)
    return FLUTL_ENODEV;
  return hip_ok (hipHostMalloc (out, bytes, hipHostMallocPortable)) ? FLUTL_OK
                                                                     : FLUTL_ENOMEM;
}

extern "C" void flutl_host_free (void *ptr)
{
  if (ptr)
    (void) hipHostFree (ptr);
}

extern "C" int flutl_list_create (flutl_ctx *c, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, flutl_list **out)
{
  if (!c || !out)
    return FLUTL_EINVAL;
  *out = nullptr;
  int rc = flutl_list_validate (width, height, regions, n_regions, prims, n_prims,
      c->atlas_used + kAtlasSlack);
  if (rc)
    return rc;
  (void) hipSetDevice (c->device);
  flutl_list *l = new (std::nothrow) flutl_list ();
  if (!l)
    return FLUTL_ENOMEM;
  const size_t rb = ((size_t) n_regions * sizeof (flutl_region) + 31) & ~(size_t) 31;
  const size_t pb = (size_t) n_prims * sizeof (flutl_prim);
  l->width = width;
  l->height = height;
  l->n_regions = n_regions;
  l->n_prims = n_prims;
  l->mem = nullptr;
  if (!hip_ok (hipMalloc (&l->mem, rb + pb + 32))) {
    delete l;
    return FLUTL_ENOMEM;
  }
  l->regions = static_cast<flutl_region *> (l->mem);
  l->prims = reinterpret_cast<flutl_prim *> (static_cast<uint8_t *> (l->mem) + rb);
  bool ok = true;
  if (n_regions)
    ok = ok && hip_ok (hipMemcpyAsync (l->regions, regions,
        (size_t) n_regions * sizeof (flutl_region), hipMemcpyHostToDevice, c->stream));
  if (n_prims)
    ok = ok && hip_ok (hipMemcpyAsync (l->prims, prims, pb, hipMemcpyHostToDevice,
        c->stream));
  ok = ok && hip_ok (hipStreamSynchronize (c->stream));
  if (!ok) {
    (void) hipFree (l->mem);
    delete l;
    return FLUTL_ENODEV;
  }
  *out = l;
  return FLUTL_OK;
}

extern "C" void flutl_list_destroy (flutl_ctx *c, flutl_list *l)
{
  if (!l)
    return;
  if (c) {
    (void) hipSetDevice (c->device);
    (void) hipStreamSynchronize (c->stream);
  }
  if (l->mem)
    (void) hipFree (l->mem);
  delete l;
}

extern "C" int flutl_render (flutl_ctx *c, const flutl_list *l, void *dst_dev,
    size_t dst_stride_bytes)
{
  if (!c || !l || !dst_dev)
    return FLUTL_EINVAL;
  (void) hipSetDevice (c->device);
  (void) atlas_join (c);
  return flutl_launch_frame (c->stream, l->width, l->height, dst_dev,
      dst_stride_bytes, l->regions, l->n_regions, l->prims, c->atlas);
}

static int render_lists_to_host (flutl_ctx *c, uint32_t width, uint32_t height,
    const flutl_region *regions_dev, uint32_t n_regions, const flutl_prim *prims_dev,
    void *dst_host, size_t dst_stride_bytes, const uint32_t *bands = nullptr,
    uint32_t n_bands = 0, uint32_t col0 = 0, uint32_t col1 = 0);

extern "C" int flutl_render_frame_to_host (flutl_ctx *c, uint32_t width,
    uint32_t height, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, void *dst_host,
    size_t dst_stride_bytes)
{
  return flutl_render_frame_rows_to_host (c, width, height, regions, n_regions, prims,
      n_prims, dst_host, dst_stride_bytes, nullptr, 0);
}

extern "C" int flutl_render_frame_rows_to_host (flutl_ctx *c, uint32_t width,
    uint32_t height, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, void *dst_host,
    size_t dst_stride_bytes, const uint32_t *row_bands, uint32_t n_bands)
{
  return flutl_render_frame_window_to_host (c, width, height, regions, n_regions, prims,
      n_prims, dst_host, dst_stride_bytes, row_bands, n_bands, 0, width);
}

extern "C" int flutl_render_frame_window_to_host (flutl_ctx *c, uint32_t width,
    uint32_t height, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, void *dst_host,
    size_t dst_stride_bytes, const uint32_t *row_bands, uint32_t n_bands,
    uint32_t col0, uint32_t col1)
{
  if (!c || !dst_host || dst_stride_bytes < (size_t) width * 4 || (n_bands && !row_bands) ||
      col0 >= col1 || col1 > width)
    return FLUTL_EINVAL;
  for (uint32_t i = 0; i < n_bands; i++)
    if (row_bands[2 * i] >= row_bands[2 * i + 1] || row_bands[2 * i + 1] > height)
      return FLUTL_EINVAL;
  int rc = flutl_list_validate (width, height, regions, n_regions, prims, n_prims,
      c->atlas_used + kAtlasSlack);
  if (rc)
    return rc;
  (void) hipSetDevice (c->device);
  flutl_region *rd;
  flutl_prim *pd;
  if ((rc = upload_list (c, regions, n_regions, prims, n_prims, &rd, &pd)))
    return rc;
  // (the final stream synchronise also covers the H2D of the caller's pageable
  // list memory, which hipMemcpyAsync stages before returning)
  return render_lists_to_host (c, width, height, rd, n_regions, pd, dst_host,
      dst_stride_bytes, row_bands, n_bands, col0, col1);
}

/* Copy a (validated) display list into the context's reusable list buffer. */
static int upload_list (flutl_ctx *c, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, flutl_region **rd_out, flutl_prim **pd_out)
{
  const size_t rb = ((size_t) n_regions * sizeof (flutl_region) + 31) & ~(size_t) 31;
  const size_t pb = (size_t) n_prims * sizeof (flutl_prim);
  if (rb + pb + 32 > c->listbuf_cap) {
    if (c->listbuf) {
      (void) hipStreamSynchronize (c->stream);
      (void) hipFree (c->listbuf);
      c->listbuf = nullptr;
      c->listbuf_cap = 0;
    }
    size_t cap = (size_t) 1 << 16;
    while (cap < rb + pb + 32)
      cap *= 2;
    if (!hip_ok (hipMalloc (&c->listbuf, cap)))
      return FLUTL_ENOMEM;
    c->listbuf_cap = cap;
  }
  flutl_region *rd = static_cast<flutl_region *> (c->listbuf);
  flutl_prim *pd = reinterpret_cast<flutl_prim *> (static_cast<uint8_t *> (c->listbuf) + rb);
  bool ok = true;
  if (n_regions)
    ok = ok && hip_ok (hipMemcpyAsync (rd, regions, (size_t) n_regions * sizeof (flutl_region),
        hipMemcpyHostToDevice, c->stream));
  if (n_prims)
    ok = ok && hip_ok (hipMemcpyAsync (pd, prims, pb, hipMemcpyHostToDevice, c->stream));
  if (!ok)
    return FLUTL_ENODEV;
  *rd_out = rd;
  *pd_out = pd;
  return FLUTL_OK;
}

extern "C" int flutl_render_to_host (flutl_ctx *c, const flutl_list *l,
    void *dst_host, size_t dst_stride_bytes)
{
  if (!c || !l || !dst_host || dst_stride_bytes < (size_t) l->width * 4)
    return FLUTL_EINVAL;
  (void) hipSetDevice (c->device);
  return render_lists_to_host (c, l->width, l->height, l->regions, l->n_regions,
      l->prims, dst_host, dst_stride_bytes);
}

static int render_lists_to_host (flutl_ctx *c, uint32_t width, uint32_t height,
    const flutl_region *regions_dev, uint32_t n_regions, const flutl_prim *prims_dev,
    void *dst_host, size_t dst_stride_bytes, const uint32_t *bands, uint32_t n_bands,
    uint32_t col0, uint32_t col1)
{
  if (col1 == 0)
    col1 = width;
  struct { uint32_t width, height, n_regions; const flutl_region *regions;
    const flutl_prim *prims; } lv = { width, height, n_regions, regions_dev, prims_dev };
  const auto *l = &lv;
  const size_t tight = (size_t) l->width * 4;
  const size_t need = tight * l->height;
  (void) atlas_join (c);
  if (need > c->frame_cap) {
    if (c->frame) {
      (void) hipStreamSynchronize (c->stream);
      (void) hipFree (c->frame);
      c->frame = nullptr;
      c->frame_cap = 0;
    }
    if (!hip_ok (hipMalloc (&c->frame, need)))
      return FLUTL_ENOMEM;
    c->frame_cap = need;
  }
  int rc = flutl_launch_frame (c->stream, l->width, l->height, c->frame, tight,
      l->regions, l->n_regions, l->prims, c->atlas);
  if (rc)
    return rc;
  // whole frame, or only the caller's row bands and columns (dirty-rectangle
  // copy: the caller guarantees the rest of dst_host already holds what it should)
  const uint32_t whole[2] = { 0, l->height };
  if (!n_bands) {
    bands = whole;
    n_bands = 1;
  }
  bool ok = true;
  for (uint32_t i = 0; i < n_bands && ok; i++) {
    const uint32_t y0 = bands[2 * i], rows = bands[2 * i + 1] - y0;
    uint8_t *d = static_cast<uint8_t *> (dst_host) + (size_t) y0 * dst_stride_bytes +
        (size_t) col0 * 4;
    const uint8_t *sp = static_cast<const uint8_t *> (c->frame) + (size_t) y0 * tight +
        (size_t) col0 * 4;
    const size_t wbytes = (size_t) (col1 - col0) * 4;
    if (dst_stride_bytes == tight && wbytes == tight)
      ok = hip_ok (hipMemcpyAsync (d, sp, tight * rows, hipMemcpyDeviceToHost, c->stream));
    else
      ok = hip_ok (hipMemcpy2DAsync (d, dst_stride_bytes, sp, tight, wbytes, rows,
          hipMemcpyDeviceToHost, c->stream));
  }
  if (!hip_ok (hipStreamSynchronize (c->stream)) || !ok)
    return FLUTL_ENODEV;
  return FLUTL_OK;
}

// ---- fused over-video on a host frame ---------------------------------------------

extern "C" int flutl_blend_over_host (flutl_ctx *c, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims, void *video_host, size_t stride_bytes)
{
  if (!c || !video_host || stride_bytes < (size_t) width * 4 || (stride_bytes & 3))
    return FLUTL_EINVAL;
  int rc = flutl_list_validate (width, height, regions, n_regions, prims, n_prims,
      c->atlas_used + kAtlasSlack);
  if (rc)
    return rc;
  uint32_t y0 = height, y1 = 0;
  for (uint32_t i = 0; i < n_regions; i++) {
    if (regions[i].x1 <= regions[i].x0 || regions[i].y1 <= regions[i].y0 || !regions[i].count)
      continue;
    y0 = (uint32_t) regions[i].y0 < y0 ? (uint32_t) regions[i].y0 : y0;
    y1 = (uint32_t) regions[i].y1 > y1 ? (uint32_t) regions[i].y1 : y1;
  }
  if (y1 <= y0)
    return FLUTL_OK;            // nothing to draw: the frame is the result
  (void) hipSetDevice (c->device);
  (void) atlas_join (c);
  // whole supertiles of 16 rows: the kernel writes video OVER'd (or copied) there
  const uint32_t fy0 = y0 & ~15u, fy1 = ((y1 + 15u) & ~15u) < height ? ((y1 + 15u) & ~15u) : height;
  const size_t tight = (size_t) width * 4, frame_bytes = tight * height;
  if (2 * frame_bytes > c->frame_cap) {
    (void) hipStreamSynchronize (c->stream);
    if (c->frame)
      (void) hipFree (c->frame);
    c->frame = nullptr;
    c->frame_cap = 0;
    if (!hip_ok (hipMalloc (&c->frame, 2 * frame_bytes)))
      return FLUTL_ENOMEM;
    c->frame_cap = 2 * frame_bytes;
  }
  uint8_t *dst = static_cast<uint8_t *> (c->frame), *vid = dst + frame_bytes;
  uint8_t *host = static_cast<uint8_t *> (video_host);
  flutl_region *rd;
  flutl_prim *pd;
  if ((rc = upload_list (c, regions, n_regions, prims, n_prims, &rd, &pd)))
    return rc;
  const size_t rows = fy1 - fy0;
  bool ok = hip_ok (hipMemcpy2DAsync (vid + (size_t) fy0 * tight, tight,
      host + (size_t) fy0 * stride_bytes, stride_bytes, tight, rows, hipMemcpyHostToDevice,
      c->stream));
  if (ok)
    ok = flutl_launch_frame_over_rows (c->stream, width, height, dst, vid, tight, rd, n_regions,
        pd, c->atlas, fy0, fy1) == FLUTL_OK;
  if (ok)
    ok = hip_ok (hipMemcpy2DAsync (host + (size_t) fy0 * stride_bytes, stride_bytes,
        dst + (size_t) fy0 * tight, tight, tight, rows, hipMemcpyDeviceToHost, c->stream));
  if (!hip_ok (hipStreamSynchronize (c->stream)) || !ok)
    return FLUTL_ENODEV;
  return FLUTL_OK;
}

// ---- pooled output frames, completed asynchronously ---------------------------

extern "C" int flutl_frame_create_quiet (flutl_ctx *c, uint32_t width, uint32_t height, int quiet,
    flutl_frame **out);

extern "C" int flutl_frame_create (flutl_ctx *c, uint32_t width, uint32_t height,
    flutl_frame **out)
{
  return flutl_frame_create_quiet (c, width, height, 0, out);
}

extern "C" int flutl_ctx_reserve_frames (flutl_ctx *c, uint32_t width, uint32_t height, uint32_t count)
{
  if (!c || width == 0 || height == 0 || width > 32767 || height > 32767 || count > 4096)
    return FLUTL_EINVAL;
  const size_t bytes = (size_t) width * height * 4;
  const size_t block = (bytes + 4095) & ~(size_t) 4095;
  uint32_t have = 0;
  for (FrameArena *a : c->frame_arenas)
    if (a->block_bytes == block) {
      std::lock_guard<std::mutex> g (a->lock);
      for (uint8_t st : a->state)
        have += st != 2;
    }
  if (have >= count)
    return FLUTL_OK;
  count -= have;
  (void) hipSetDevice (c->device);
  FrameArena *a = new (std::nothrow) FrameArena ();
  if (!a)
    return FLUTL_ENOMEM;
  const size_t huge = (size_t) 2 << 20;
  a->refs.store (1);
  a->device = c->device;
  a->block_bytes = block;
  a->count = count;
  a->host = a->twin = nullptr;
  const size_t total = (block * count + huge - 1) & ~(huge - 1);
  a->map_bytes = total + huge;
  a->map = mmap (nullptr, a->map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (a->map == MAP_FAILED) {
    delete a;
    return FLUTL_ENOMEM;
  }
  char *base = reinterpret_cast<char *> ((reinterpret_cast<uintptr_t> (a->map) + huge - 1) & ~(uintptr_t) (huge - 1));
  (void) madvise (base, total, MADV_HUGEPAGE);
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
  // (the pages brought in zeroed by the kernel, once; a kernel without the advice: by writing)
  if (madvise (base, total, MADV_POPULATE_WRITE) != 0)
    memset (base, 0, total);
  void *dev = nullptr;
  bool ok = hip_ok (hipHostRegister (base, total, hipHostRegisterPortable | hipHostRegisterMapped));
  if (ok)
    a->host = base;
  // (the frames' device address is taken per frame; it must be the host's for the kernels'
  // sake nowhere -- they are handed host_dev -- but a mapping that fails shows here)
  ok = ok && hip_ok (hipHostGetDevicePointer (&dev, base, 0));
  ok = ok && hip_ok (hipMalloc (reinterpret_cast<void **> (&a->twin), block * count));
  ok = ok && hip_ok (hipMemsetAsync (a->twin, 0, block * count, c->stream));
  if (!ok) {
    if (!a->host) {            // (never registered: unref would unregister it)
      munmap (a->map, a->map_bytes);
      a->map = nullptr;
    }
    frame_arena_unref (a);
    return FLUTL_ENOMEM;
  }
  a->state.assign (count, 0);
  c->frame_arenas.push_back (a);
  return FLUTL_OK;
}

// A free block of a reservation for frames of `bytes`: a zero one first.  *clean: it is zero
// (the host half; the twin is cleared by whoever takes it unless the block is fresh).
static FrameArena *frame_arena_take (flutl_ctx *c, size_t bytes, uint32_t *index, bool *clean)
{
  const size_t block = (bytes + 4095) & ~(size_t) 4095;
  for (int want = 0; want < 2; want++)
    for (FrameArena *a : c->frame_arenas) {
      if (a->block_bytes != block)
        continue;
      std::lock_guard<std::mutex> g (a->lock);
      for (uint32_t i = 0; i < a->count; i++)
        if (a->state[i] == want) {
          a->state[i] = 2;
          a->refs.fetch_add (1, std::memory_order_relaxed);
          *index = i;
          *clean = want == 0;
          return a;
        }
    }
  return nullptr;
}

// (quiet: the host block is cleared by the calling thread, not by the device over PCIe in
// stream order -- a frame made in the background while the stream is busy with cues must
// not put 1.3 ms of clearing in front of them; the twin's clearing is microseconds)
extern "C" int flutl_frame_create_quiet (flutl_ctx *c, uint32_t width, uint32_t height, int quiet,
    flutl_frame **out)
{
  if (!c || !out || width == 0 || height == 0 || width > 32767 || height > 32767)
    return FLUTL_EINVAL;
  *out = nullptr;
  (void) hipSetDevice (c->device);
  flutl_frame *f = new (std::nothrow) flutl_frame ();
  if (!f)
    return FLUTL_ENOMEM;
  const size_t bytes = (size_t) width * height * 4;
  f->device = c->device;
  f->width = width;
  f->height = height;
  f->host = f->host_dev = f->twin = nullptr;
  f->done = nullptr;
  f->pending = false;
  f->error = 0;
  f->written = false;
  f->lane = 0;
  f->ext[0] = f->ext[1] = f->ext[2] = f->ext[3] = 0;
  bool ok = true, clean = false;
  f->arena = frame_arena_take (c, bytes, &f->arena_block, &clean);
  if (f->arena) {
    f->host = f->arena->host + f->arena->block_bytes * f->arena_block;
    f->twin = f->arena->twin + f->arena->block_bytes * f->arena_block;
  } else {
    ok = hip_ok (hipHostMalloc (&f->host, bytes, hipHostMallocPortable | hipHostMallocMapped));
    ok = ok && hip_ok (hipMalloc (&f->twin, bytes));
  }
  ok = ok && hip_ok (hipHostGetDevicePointer (&f->host_dev, f->host, 0));
  // (system-scope release: what the kernel wrote to the host block is visible to
  // the CPU once the event has been waited for)
  ok = ok && hip_ok (hipEventCreateWithFlags (&f->done,
      hipEventDisableTiming | hipEventReleaseToSystem));
  // the transparent frame, in both copies: cleared by the device in stream order
  // (the host block over PCIe: no 33 MB memset on the caller's thread, and no
  // wait -- whatever is submitted into the frame comes later on the same stream)
  // (a fresh block of a reservation is zero in both halves: the twins' clearing was queued
  // on this stream when the reservation was made)
  if (ok && quiet && !clean)
    memset (f->host, 0, bytes);
  ok = ok && (clean || hip_ok (hipMemsetAsync (f->twin, 0, bytes, c->stream))) &&
      (quiet || clean || hip_ok (hipMemsetAsync (f->host_dev, 0, bytes, c->stream))) &&
      hip_ok (hipEventRecord (f->done, c->stream));
  if (ok)
    f->pending = true;
  if (!ok) {
    flutl_frame_destroy (f);
    return FLUTL_ENOMEM;
  }
  *out = f;
  return FLUTL_OK;
}

extern "C" void flutl_frame_destroy (flutl_frame *f)
{
  if (!f)
    return;
  if (f->done) {
    if (f->pending)
      (void) hipEventSynchronize (f->done);
    (void) hipEventDestroy (f->done);
  }
  if (f->arena) {
    {
      std::lock_guard<std::mutex> g (f->arena->lock);
      f->arena->state[f->arena_block] = 1;
    }
    frame_arena_unref (f->arena);
  } else {
    if (f->twin)
      (void) hipFree (f->twin);
    if (f->host)
      (void) hipHostFree (f->host);
  }
  delete f;
}

extern "C" void *flutl_frame_host (const flutl_frame *f) { return f ? f->host : nullptr; }

extern "C" void flutl_frame_mark_written (flutl_frame *f)
{
  if (f)
    f->written.store (true, std::memory_order_release);
}

extern "C" const void *flutl_frame_device (const flutl_frame *f)
{
  return f ? f->twin : nullptr;
}

extern "C" int flutl_frame_wait (flutl_frame *f)
{
  if (!f)
    return FLUTL_EINVAL;
  if (f->pending.load (std::memory_order_acquire)) {
    // (hipEventSynchronize does not depend on the calling thread's device, and
    // several threads may wait for one event)
    if (!hip_ok (hipEventSynchronize (f->done)))
      f->error.store (FLUTL_ENODEV);
    f->pending.store (false, std::memory_order_release);
  }
  return f->error.load ();
}

extern "C" uint64_t flutl_ctx_delta_bytes (flutl_ctx *c)
{
  uint64_t n = 0;
  if (!c || !c->delta_segments)
    return 0;
  (void) hipSetDevice (c->device);
  if (c->stream2)               // (launches on the other lane count into it too)
    (void) hipStreamSynchronize (c->stream2);
  if (!hip_ok (hipMemcpyAsync (&n, c->delta_segments, sizeof n, hipMemcpyDeviceToHost, c->stream)) ||
      !hip_ok (hipStreamSynchronize (c->stream)))
    return 0;
  return n * flutl_delta_segment_bytes ();
}

static inline uint64_t now_ns (void)
{
  struct timespec ts;
  clock_gettime (CLOCK_MONOTONIC, &ts);
  return (uint64_t) ts.tv_sec * 1000000000ull + (uint64_t) ts.tv_nsec;
}
#define SUB_MARK(k) do { if (c->api_timing) { const uint64_t n_ = now_ns (); c->t_sub[k] += n_ - t_mark; t_mark = n_; } } while (0)

extern "C" int flutl_frame_stage (flutl_ctx *c, flutl_frame *f,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims)
{
  if (!c || !f || f->device != c->device)
    return FLUTL_EINVAL;
  if (f->error)
    return f->error;
  c->staged.valid = false;
  const uint32_t width = f->width, height = f->height;
  uint64_t t_mark = c->api_timing ? now_ns () : 0;
  // (masks the list names that are still to be made -- the outline polygons of this very
  // frame, queued between stage and submit -- have their space in the atlas already: the
  // caller keeps offsets in step with its host-side atlas; what is checked is the layout)
  int rc = flutl_list_validate (width, height, regions, n_regions, prims, n_prims, 0x7fffffffu);
  if (rc)
    return rc;
  (void) hipSetDevice (c->device);
  // the rows the new content can occupy: the union of the region boxes
  uint32_t now[4] = { width, height, 0, 0 };
  for (uint32_t i = 0; i < n_regions; i++) {
    if (regions[i].x1 <= regions[i].x0 || regions[i].y1 <= regions[i].y0)
      continue;
    now[0] = (uint32_t) regions[i].x0 < now[0] ? (uint32_t) regions[i].x0 : now[0];
    now[1] = (uint32_t) regions[i].y0 < now[1] ? (uint32_t) regions[i].y0 : now[1];
    now[2] = (uint32_t) regions[i].x1 > now[2] ? (uint32_t) regions[i].x1 : now[2];
    now[3] = (uint32_t) regions[i].y1 > now[3] ? (uint32_t) regions[i].y1 : now[3];
  }
  if (now[2] <= now[0] || now[3] <= now[1])
    now[0] = now[1] = now[2] = now[3] = 0;
  // a list slot: the oldest submission must have finished with it
  const int slot = (int) c->next_slot;
  list_slot &sl = c->slots[slot];
  c->next_slot = (c->next_slot + 1) % kListSlots;
  SUB_MARK (0);
  if (sl.busy) {
    if (!hip_ok (hipEventSynchronize (sl.done)))
      return FLUTL_ENODEV;
    sl.busy = false;
  }
  SUB_MARK (1);
  const size_t rb = ((size_t) n_regions * sizeof (flutl_region) + 31) & ~(size_t) 31;
  const size_t pb = ((size_t) n_prims * sizeof (flutl_prim) + 31) & ~(size_t) 31;
  const size_t ob = (((size_t) height + 15) / 16 * sizeof (uint16_t) + 31) & ~(size_t) 31;   // row order
  if (rb + pb + ob + 32 > sl.cap) {
    if (sl.host)
      (void) hipHostFree (sl.host);
    if (sl.dev)
      (void) hipFree (sl.dev);
    sl.host = sl.host_dev = sl.dev = nullptr;
    sl.cap = 0;
    size_t cap = (size_t) 1 << 16;
    while (cap < rb + pb + ob + 32)
      cap *= 2;
    if (!hip_ok (hipHostMalloc (&sl.host, cap, hipHostMallocDefault)) ||
        !hip_ok (hipMalloc (&sl.dev, cap)))
      return FLUTL_ENOMEM;
    if (!hip_ok (hipHostGetDevicePointer (&sl.host_dev, sl.host, 0)))
      sl.host_dev = nullptr;
    sl.cap = cap;
  }
  if (!sl.done && !hip_ok (hipEventCreateWithFlags (&sl.done, hipEventDisableTiming)))
    return FLUTL_ENODEV;
  if (n_regions)
    memcpy (sl.host, regions, (size_t) n_regions * sizeof (flutl_region));
  if (n_prims)
    memcpy (static_cast<uint8_t *> (sl.host) + rb, prims, (size_t) n_prims * sizeof (flutl_prim));
  // the frame's lane: the two in turn -- while nothing new goes into the atlas (a frame that
  // brings masks or outlines waits for them on lane 0 anyway, and would pay a join per lane:
  // measured 4 % slower on a stream of new outlined text) and while this is the only context
  // on its device in the process (every context has three streams, the runtime a handful of
  // hardware queues per process: four elements with two lanes each were no faster, in one run
  // of two 15 % slower, than with one); behind whatever the frame's last use queued on the other lane (its clearing when it
  // was made, its last submission)
  const bool quiet_atlas = c->staged_scan_gen == c->scan_gen && c->staged_upload_gen == c->upload_gen;
  c->staged_scan_gen = c->scan_gen;
  c->staged_upload_gen = c->upload_gen;
  // (... for a few frames: the contexts of one pipeline's elements are made within
  // milliseconds of each other)
  const bool alone = g_contexts[c->device & 15].load (std::memory_order_relaxed) <= 1;
  c->alone_run = alone ? c->alone_run + 1 : 0;
  const int lane = c->stream2 && quiet_atlas && c->alone_run >= c->lanes_after ? (int) (c->n_staged++ & 1u) : 0;
  hipStream_t ls = lane_stream (c, lane);
  if (f->lane != lane) {
    if (!hip_ok (hipStreamWaitEvent (ls, f->done, 0)))
      return FLUTL_ENODEV;
    f->lane = lane;
  }
  // a frame its consumer wrote into: the twin is brought up to the host block (the whole
  // frame up, in stream order) and every row is composited, so that the comparison below
  // puts each pixel right again in both
  if (f->written.exchange (false, std::memory_order_acq_rel)) {
    if (!hip_ok (hipMemcpyAsync (f->twin, f->host, (size_t) width * 4 * height, hipMemcpyHostToDevice,
            ls)))
      return FLUTL_ENODEV;
    f->ext[0] = f->ext[1] = 0;
    f->ext[2] = width;
    f->ext[3] = height;
  }
  // rows of this submission: what the frame held before + what it holds now
  uint32_t y0 = now[1], y1 = now[3];
  if (f->ext[3] > f->ext[1]) {
    if (y1 <= y0) {
      y0 = f->ext[1];
      y1 = f->ext[3];
    } else {
      y0 = f->ext[1] < y0 ? f->ext[1] : y0;
      y1 = f->ext[3] > y1 ? f->ext[3] : y1;
    }
  }
  // their supertile rows, heaviest first (flutl_launch_frame_ordered)
  const uint32_t n_order = flutl_row_order (width, height, regions, n_regions, prims, n_prims,
      y0, y1, reinterpret_cast<uint16_t *> (static_cast<uint8_t *> (sl.host) + rb + pb));
  SUB_MARK (2);
  c->staged.frame = f;
  c->staged.lane = lane;
  c->staged.slot = slot;
  c->staged.rb = rb;
  c->staged.pb = pb;
  c->staged.ob = ob;
  c->staged.copy_bytes = rb + pb + (n_order ? ob : 0);
  c->staged.copy_pending = c->staged.copy_bytes != 0;
  c->staged.n_regions = n_regions;
  c->staged.n_prims = n_prims;
  c->staged.n_order = n_order;
  c->staged.y0 = y0;
  c->staged.y1 = y1;
  memcpy (c->staged.now, now, sizeof now);
  c->staged.valid = true;
  return FLUTL_OK;
}

extern "C" int flutl_frame_submit_staged (flutl_ctx *c, flutl_frame *f)
{
  if (!c || !f || !c->staged.valid || c->staged.frame != f)
    return FLUTL_EINVAL;
  c->staged.valid = false;
  (void) hipSetDevice (c->device);
  uint64_t t_mark = c->api_timing ? now_ns () : 0;
  const uint32_t width = f->width, height = f->height;
  list_slot &sl = c->slots[c->staged.slot];
  const size_t rb = c->staged.rb, pb = c->staged.pb;
  const uint32_t n_order = c->staged.n_order;
  // (now that the frame's atlas entries have been made: every primitive reads inside the atlas)
  int rc = flutl_list_validate (width, height, static_cast<const flutl_region *> (sl.host),
      c->staged.n_regions, reinterpret_cast<const flutl_prim *> (static_cast<uint8_t *> (sl.host) + rb),
      c->staged.n_prims, c->atlas_used + kAtlasSlack);
  if (rc)
    return rc;
  // scratch frame of the lane (here: where translucent groups park pixels)
  const int lane = c->staged.lane;
  hipStream_t ls = lane_stream (c, lane);
  void *&park = lane ? c->frame2 : c->frame;
  size_t &park_cap = lane ? c->frame2_cap : c->frame_cap;
  const size_t need = (size_t) width * 4 * height;
  if (need > park_cap) {
    (void) hipStreamSynchronize (ls);
    if (park)
      (void) hipFree (park);
    park = nullptr;
    park_cap = 0;
    if (!hip_ok (hipMalloc (&park, need)))
      return FLUTL_ENOMEM;
    park_cap = need;
  }
  bool ok = atlas_join_lane (c, lane) && uploads_join_lane (c, lane);
  // The list: up with a scan conversion's launch since it was staged; else the launch reads
  // it where it lies, in the slot's page-locked half -- a few KB that the first waves fetch
  // over PCIe and the others find in the L2: 3.2 us on the launch (tools/list_in_host_bench.py:
  // 22.4 -> 25.6 us) instead of a copy operation of 4.2 us, the gap behind it and 3-6 us of
  // hipMemcpyAsync on this thread.  The slot is not written again before its event has been
  // seen.  FLUTL_LIST_COPY=1: copied, as before.
  const uint8_t *lbase = static_cast<const uint8_t *> (sl.dev);
  if (c->staged.copy_pending) {
    if (sl.host_dev && !c->list_copy)
      lbase = static_cast<const uint8_t *> (sl.host_dev);
    else
      ok = hip_ok (hipMemcpyAsync (sl.dev, sl.host, c->staged.copy_bytes, hipMemcpyHostToDevice, ls));
    c->staged.copy_pending = false;
  }
  if (!ok)
    return FLUTL_ENODEV;
  SUB_MARK (3);
  const flutl_region *rd = reinterpret_cast<const flutl_region *> (lbase);
  const flutl_prim *pd = reinterpret_cast<const flutl_prim *> (lbase + rb);
  if (!c->delta_segments) {
    // (once per context, and waited for: launches on either lane count into it)
    if (!hip_ok (hipMalloc (reinterpret_cast<void **> (&c->delta_segments), sizeof (uint64_t))) ||
        !hip_ok (hipMemsetAsync (c->delta_segments, 0, sizeof (uint64_t), c->stream)) ||
        !hip_ok (hipStreamSynchronize (c->stream)))
      rc = FLUTL_ENOMEM;
  }
  // one launch: composite those rows, compare with the twin, write what differs
  // into the twin and the host block (flutl_launch_frame_delta)
  if (rc == FLUTL_OK && n_order)
    rc = flutl_launch_frame_delta (ls, width, height, f->twin, f->host_dev, park,
        (size_t) width * 4, rd, c->staged.n_regions, pd, c->atlas, c->staged.y0, c->staged.y1,
        reinterpret_cast<const uint16_t *> (lbase + rb + pb),
        c->delta_segments);
  SUB_MARK (4);
  ok = rc == FLUTL_OK && hip_ok (hipEventRecord (sl.done, ls)) &&
      hip_ok (hipEventRecord (f->done, ls));
  SUB_MARK (5);
  if (c->api_timing)
    c->n_sub++;
  sl.busy = true;
  f->pending = true;
  if (!ok) {
    // whatever was queued: the frame's content is unknown from here on
    (void) hipStreamSynchronize (ls);
    f->error = rc ? rc : FLUTL_ENODEV;
    return f->error;
  }
  memcpy (f->ext, c->staged.now, sizeof c->staged.now);
  return FLUTL_OK;
}

extern "C" int flutl_frame_submit (flutl_ctx *c, flutl_frame *f,
    const flutl_region *regions, uint32_t n_regions, const flutl_prim *prims,
    uint32_t n_prims)
{
  if (!c || !f || f->device != c->device)
    return FLUTL_EINVAL;
  // (everything the list names is in the atlas by now: checked against what is there)
  int rc = flutl_list_validate (f->width, f->height, regions, n_regions, prims, n_prims,
      c->atlas_used + kAtlasSlack);
  if (rc == FLUTL_OK)
    rc = flutl_frame_stage (c, f, regions, n_regions, prims, n_prims);
  return rc == FLUTL_OK ? flutl_frame_submit_staged (c, f) : rc;
}
