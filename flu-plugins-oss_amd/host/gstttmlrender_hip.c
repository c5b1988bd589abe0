/*
 * gstttmlrender_hip.c -- MI355X-native `ttmlrender` element.
 *
 * Drop-in replacement for the reference's plugins/ttml/gstttmlrender.c +
 * gstttmlblur.c: it provides the same GType (gst_ttmlrender_get_type, declared
 * by the reference's own gstttmlrender.h, which plugins/ttml/gstfluttml.c:57-60
 * registers as element "ttmlrender"), the same pad template, properties and
 * GstTTMLBaseClass vfuncs (plugins/ttml/gstttmlbase.h:71-98), and is compiled
 * together with the reference's UNTOUCHED base class / parser / data model.
 *
 * What differs is how a buffer is produced (gen_buffer, reference
 * gstttmlrender.c:1427-1478): the layout work that is inherently host side
 * (Pango markup -> PangoLayout -> glyph runs) stays here, but nothing is drawn
 * on the host.  Every drawing call of the reference is RECORDED as a display
 * list (flutl_recorder.c) and the frame is composited by the HIP kernel behind
 * the C-ABI of include/flutl.h.  There is no CPU rendering fallback: if the
 * HIP library cannot render, gen_buffer returns NULL (the base class's error
 * convention, gstttmlbase.c:161-163) after logging an error.
 *
 * Reference behaviours that affect pixels are reproduced on purpose, including
 * the odd ones; each is marked "reference:" below.
 *
 * Not implemented yet (counted and logged, never silently approximated):
 * (nothing the reference can render in a sane time; what is beyond a limit fails the buffer).
 */
#ifdef HAVE_CONFIG_H
#include "config.h"
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <gst/gst.h>
#include <gst/video/video.h>
#include <gst/video/video-overlay-composition.h>
#include <pango/pangocairo.h>
#include <libxml/parser.h>      /* xmlInitParser (gst_ttmlrender_class_init) */

#include "gstttmlbase.h"
#include "gstttmlrender.h"
#include "gstttmlstate.h"
#include "gstttmlspan.h"
#include "gstttmlstyle.h"
#include "gstttmlattribute.h"
#include "gstttmlutils.h"

#include "flutl.h"
#include "flutl_recorder.h"
#include "flutl_pack.h"

GST_DEBUG_CATEGORY_EXTERN (ttmlrender_debug);
#define GST_CAT_DEFAULT ttmlrender_debug

#define DEFAULT_WINDOW_WIDTH 720
#define DEFAULT_WINDOW_HEIGHT 576

enum
{
  PROP_0,
  PROP_DEFAULT_FONT_FAMILY,
  PROP_DEFAULT_FONT_SIZE,
  PROP_DEFAULT_TEXT_ALIGN,
  PROP_DEFAULT_DISPLAY_ALIGN,
  PROP_WINDOW_WIDTH,
  PROP_WINDOW_HEIGHT
};

/* Pool of output frames.  A frame (flutl_frame, include/flutl.h) is a page-locked
 * host block -- system memory as far as downstream is concerned -- plus its twin
 * in HBM; it is wrapped as a GstMemory of our own allocator, and when downstream
 * drops the buffer the frame comes back here.  The pool is reference counted
 * because buffers may outlive the element.
 *
 * Asynchronous completion + delta read-back: gen_buffer only QUEUES
 * the device work of a buffer (list upload and ONE kernel that composites, compares
 * with the twin and writes just the 64-byte segments that differ from what the
 * frame held before into the host block and the twin) and returns; the streaming thread records the
 * next buffer meanwhile.  Mapping the memory waits for the frame (any thread).
 * The memory is handed out READONLY so that nobody scribbles into the block --
 * block and twin must stay equal (a downstream element that wants to write
 * makes its copy, as GStreamer's writability rules require anyway).
 * A HIP-aware consumer finds the device address of the twin as qdata
 * "flutl-hip-device-frame" on the memory (valid once the memory has been mapped).
 * FLUTL_FULL_D2H=1 restores synchronous whole-frame copies into writable memory. */
#define MAX_IN_FLIGHT 16           /* buffers handed out but not yet queued on the device */
#define DEVICE_STROKES_AHEAD 6     /* ... with outlines made on the device (run_ahead) */
#define DEVICE_STROKES_AHEAD_MAX 16
#define VIDEO_CUES_AHEAD 64        /* fused over-video: cues the TTML side may be ahead of the video */
#define POOL_MAX_FREE 24
#define POOL_BUDGET_BYTES ((gsize) 640 << 20)   /* page-locked frames kept per element (+ twins in HBM) */

/* Frames of `size` bytes that fit the budget run_ahead() also uses, plus the one
 * being handed out: 20 at 4K (the caps below apply), 5 at 8K. */
static inline guint
pool_budget_frames (gsize size)
{
  return (guint) (POOL_BUDGET_BYTES / MAX (size, 1)) + 1;
}

typedef struct
{
  flutl_frame *frame;             /* asynchronous mode */
  gpointer block;                 /* FLUTL_FULL_D2H: a plain page-locked block */
  gboolean stale;                 /* a render into it failed: not to be recycled */
} FrameSlot;

typedef struct
{
  gint refs;
  GMutex lock;
  gsize block_size, prefilled_size;
  guint prefilled_count;
  GPtrArray *free_slots;          /* FrameSlot* */
} FramePool;

typedef struct
{
  FramePool *pool;
  FrameSlot *slot;
  gsize size;
} FrameBlock;

static FramePool *
frame_pool_new (void)
{
  FramePool *p = g_new0 (FramePool, 1);
  p->refs = 1;
  g_mutex_init (&p->lock);
  p->free_slots = g_ptr_array_new ();
  return p;
}

static void
frame_slot_free (FrameSlot *s)
{
  if (s->frame)
    flutl_frame_destroy (s->frame);
  if (s->block)
    flutl_host_free (s->block);
  g_free (s);
}

static void
frame_pool_unref (FramePool *p)
{
  if (!g_atomic_int_dec_and_test (&p->refs))
    return;
  for (guint i = 0; i < p->free_slots->len; i++)
    frame_slot_free (g_ptr_array_index (p->free_slots, i));
  g_ptr_array_free (p->free_slots, TRUE);
  g_mutex_clear (&p->lock);
  g_free (p);
}

static void
frame_block_release (gpointer data)
{
  FrameBlock *b = data;
  FramePool *p = b->pool;
  g_mutex_lock (&p->lock);
  /* keep at most a few frames, and only of the current size */
  if (b->size == p->block_size &&
      p->free_slots->len < MIN (POOL_MAX_FREE, pool_budget_frames (b->size)) &&
      !b->slot->stale)
    g_ptr_array_add (p->free_slots, b->slot);
  else
    frame_slot_free (b->slot);
  g_mutex_unlock (&p->lock);
  frame_pool_unref (p);
  g_free (b);
}

/* A buffer's promise: gen_buffer hands the buffer out when its display list is
 * recorded; the finisher thread (below) queues it on the device when the masks
 * of its outlines are there, and says so here. */
typedef struct
{
  gint refs;
  GMutex lock;
  GCond cond;
  gboolean ready;
  int error;
} Ticket;

static Ticket *
ticket_new (void)
{
  Ticket *t = g_new0 (Ticket, 1);
  t->refs = 1;
  g_mutex_init (&t->lock);
  g_cond_init (&t->cond);
  return t;
}

static Ticket *
ticket_ref (Ticket *t)
{
  g_atomic_int_inc (&t->refs);
  return t;
}

static void
ticket_unref (Ticket *t)
{
  if (!g_atomic_int_dec_and_test (&t->refs))
    return;
  g_mutex_clear (&t->lock);
  g_cond_clear (&t->cond);
  g_free (t);
}

static void
ticket_complete (Ticket *t, int error)
{
  g_mutex_lock (&t->lock);
  t->ready = TRUE;
  t->error = error;
  g_cond_broadcast (&t->cond);
  g_mutex_unlock (&t->lock);
}

static int
ticket_wait (Ticket *t)
{
  int error;
  g_mutex_lock (&t->lock);
  while (!t->ready)
    g_cond_wait (&t->cond, &t->lock);
  error = t->error;
  g_mutex_unlock (&t->lock);
  return error;
}

/* ---- GstMemory over a frame: map = wait for the frame ------------------------- */
typedef struct
{
  GstMemory mem;
  FrameBlock *block;              /* NULL for sub-memories (they hold the parent) */
  flutl_frame *frame;
  Ticket *ticket;                 /* NULL: queued on the device before it was handed out */
  guint8 *data;
} FlutlMemory;

typedef struct { GstAllocator parent; } FlutlAllocator;
typedef struct { GstAllocatorClass parent_class; } FlutlAllocatorClass;
GType flutl_allocator_get_type (void);
G_DEFINE_TYPE (FlutlAllocator, flutl_allocator, GST_TYPE_ALLOCATOR);

static gpointer
flutl_mem_map (GstMemory *mem, gsize maxsize, GstMapFlags flags)
{
  FlutlMemory *m = (FlutlMemory *) mem;
  if ((m->ticket && ticket_wait (m->ticket) != FLUTL_OK) ||
      flutl_frame_wait (m->frame) != FLUTL_OK) {
    GST_ERROR ("ttmlrender: the frame behind this memory failed to render");
    return NULL;
  }
  /* (FLUTL_WRITABLE: whoever maps for writing works in place in the block; block and twin
   * are brought together again when the frame is next used) */
  if (flags & GST_MAP_WRITE)
    flutl_frame_mark_written (m->frame);
  return m->data;
}

static void
flutl_mem_unmap (GstMemory *mem)
{
}

static GstMemory *
flutl_mem_share (GstMemory *mem, gssize offset, gssize size)
{
  FlutlMemory *m = (FlutlMemory *) mem, *sub;
  GstMemory *parent = mem->parent ? mem->parent : mem;
  if (size == -1)
    size = mem->size - offset;
  sub = g_new0 (FlutlMemory, 1);
  gst_memory_init (GST_MEMORY_CAST (sub),
      GST_MINI_OBJECT_FLAGS (parent) | GST_MINI_OBJECT_FLAG_LOCK_READONLY, mem->allocator,
      parent, mem->maxsize, mem->align, mem->offset + offset, size);
  sub->frame = m->frame;
  sub->ticket = m->ticket ? ticket_ref (m->ticket) : NULL;
  sub->data = m->data;
  return GST_MEMORY_CAST (sub);
}

static void
flutl_allocator_free (GstAllocator *allocator, GstMemory *mem)
{
  FlutlMemory *m = (FlutlMemory *) mem;
  if (m->block)
    frame_block_release (m->block);
  if (m->ticket)
    ticket_unref (m->ticket);
  g_free (m);
}

static void
flutl_allocator_class_init (FlutlAllocatorClass *klass)
{
  GstAllocatorClass *ac = GST_ALLOCATOR_CLASS (klass);
  ac->alloc = NULL;               /* frames come from the element's pool only */
  ac->free = flutl_allocator_free;
}

static void
flutl_allocator_init (FlutlAllocator *self)
{
  GstAllocator *a = GST_ALLOCATOR_CAST (self);
  a->mem_type = "FlutlHipFrame";
  a->mem_map = flutl_mem_map;
  a->mem_unmap = flutl_mem_unmap;
  a->mem_share = flutl_mem_share;
  GST_OBJECT_FLAG_SET (self, GST_ALLOCATOR_FLAG_CUSTOM_ALLOC);
}

static GstAllocator *
flutl_allocator_get (void)
{
  static gsize once = 0;
  static GstAllocator *a = NULL;
  if (g_once_init_enter (&once)) {
    a = g_object_new (flutl_allocator_get_type (), NULL);
    gst_object_ref_sink (a);
    g_once_init_leave (&once, 1);
  }
  return a;
}

static FrameSlot *
frame_slot_new_ex (flutl_ctx *ctx, gint width, gint height, gboolean async, gboolean quiet)
{
  FrameSlot *slot = g_new0 (FrameSlot, 1);
  if (async) {
    if (flutl_frame_create_quiet (ctx, width, height, quiet, &slot->frame) != FLUTL_OK) {
      g_free (slot);
      return NULL;
    }
  } else {
    if (flutl_host_alloc ((gsize) width * height * 4, &slot->block) != FLUTL_OK) {
      g_free (slot);
      return NULL;
    }
  }
  return slot;
}

static FrameSlot *
frame_slot_new (flutl_ctx *ctx, gint width, gint height, gboolean async)
{
  return frame_slot_new_ex (ctx, width, height, async, FALSE);
}

/* (pool lock held) frames of another size go */
static void
frame_pool_set_size (FramePool *p, gsize size)
{
  if (p->block_size == size)
    return;
  for (guint i = 0; i < p->free_slots->len; i++)
    frame_slot_free (g_ptr_array_index (p->free_slots, i));
  g_ptr_array_set_size (p->free_slots, 0);
  p->block_size = size;
}

/* A slot of the pool (recycled, or made on `ctx`), or NULL. */
static FrameSlot *
frame_pool_take (FramePool *p, flutl_ctx *ctx, gint width, gint height, gboolean async)
{
  FrameSlot *slot = NULL;
  g_mutex_lock (&p->lock);
  frame_pool_set_size (p, (gsize) width * height * 4);
  for (guint i = p->free_slots->len; i-- > 0 && !slot;) {
    FrameSlot *c = g_ptr_array_index (p->free_slots, i);
    if ((c->frame != NULL) == async)
      slot = g_ptr_array_remove_index_fast (p->free_slots, i);
  }
  g_mutex_unlock (&p->lock);
  return slot ? slot : frame_slot_new (ctx, width, height, async);
}

/* Frames of a new size are made ahead, when the first one is asked for: a frame
 * costs milliseconds (page-locking 4 bytes per pixel), which belongs to start-up
 * and not between two buffers of a running stream. */
static void
frame_pool_prefill (FramePool *p, flutl_ctx *ctx, gint width, gint height, guint n)
{
  const gsize size = (gsize) width * height * 4;
  guint have = 0;
  g_mutex_lock (&p->lock);
  if (p->prefilled_size == size)
    have = p->prefilled_count;
  if (have >= n) {
    g_mutex_unlock (&p->lock);
    return;
  }
  p->prefilled_size = size;
  p->prefilled_count = n;
  frame_pool_set_size (p, size);
  g_mutex_unlock (&p->lock);
  /* (their memory in one piece: FLUTL_NO_FRAME_ARENA=1 makes each frame's own, as rounds 1-3) */
  if (!g_getenv ("FLUTL_NO_FRAME_ARENA"))
    (void) flutl_ctx_reserve_frames (ctx, width, height, n - have);
  for (guint i = have; i < n; i++) {
    /* (cleared by this thread, at start-up: not by the device in front of the first cues) */
    FrameSlot *slot = frame_slot_new_ex (ctx, width, height, TRUE, TRUE);
    if (!slot)
      break;
    g_mutex_lock (&p->lock);
    g_ptr_array_add (p->free_slots, slot);
    g_mutex_unlock (&p->lock);
  }
}

/* A GstBuffer over a pooled slot; the slot returns to the pool with the buffer. */
static GstBuffer *
frame_pool_wrap (FramePool *p, FrameSlot *slot, gsize size, Ticket *ticket)
{
  FrameBlock *b = g_new0 (FrameBlock, 1);
  GstBuffer *buffer;
  b->pool = p;
  b->slot = slot;
  b->size = size;
  g_atomic_int_inc (&p->refs);
  if (slot->frame) {
    FlutlMemory *m = g_new0 (FlutlMemory, 1);
    /* READONLY, unless FLUTL_WRITABLE=1: the reference's buffers are writable
     * (gstttmlrender.c:1442-1444), so an element that works in place on its input takes them as
     * they are; READONLY memory makes it copy the frame first (33 MB at 4K: 8-9 ms per frame,
     * tools/map_cost.py).  The price of writable: a frame that was written into costs its next
     * user one copy of the frame up to the device and a composite of all its rows
     * (flutl_frame_mark_written) instead of the rows of the regions. */
    static gsize writable_once = 0;
    static gboolean writable = FALSE;
    if (g_once_init_enter (&writable_once)) {
      writable = g_getenv ("FLUTL_WRITABLE") != NULL;
      g_once_init_leave (&writable_once, 1);
    }
    gst_memory_init (GST_MEMORY_CAST (m), writable ? 0 : GST_MEMORY_FLAG_READONLY, flutl_allocator_get (),
        NULL, size, 0, 0, size);
    m->block = b;
    m->frame = slot->frame;
    m->ticket = ticket ? ticket_ref (ticket) : NULL;
    m->data = flutl_frame_host (slot->frame);
    gst_mini_object_set_qdata (GST_MINI_OBJECT_CAST (m),
        g_quark_from_static_string ("flutl-hip-device-frame"),
        (gpointer) flutl_frame_device (slot->frame), NULL);
    buffer = gst_buffer_new ();
    gst_buffer_append_memory (buffer, GST_MEMORY_CAST (m));
    return buffer;
  }
  return gst_buffer_new_wrapped_full (0, slot->block, size, 0, size, b, frame_block_release);
}

/* Device-side state of one element instance. */
typedef struct
{
  guint32 offset;
  gint w, h;
} BlurredImage;

typedef struct
{
  FramePool *pool;
  FlutlRecorder *rec;
  FlutlRecorder *sub;        /* child recorder for blurred-outline surfaces */
  GHashTable *blurred;       /* key (checksum) -> BlurredImage, per document */
  flutl_ctx *ctx;            /* NULL until first use; stays NULL without a GPU */
  gboolean ctx_failed;
  int device;                   /* the HIP device of this instance (pick_device) */
  gboolean edge_blocks_set;     /* the recorder's stroke workers write edges into the context's blocks */
  FlutlPackWriter *dump;     /* FLUTL_DUMP_PACK=<file>: scene-pack dump */
  gboolean dump_only;        /* FLUTL_DUMP_ONLY=1: record, do not render */
  gboolean dump_discard;     /* the pack goes to /dev/null: not written at all */
  guint8 *blank;             /* ... and hand out this blank frame */
  gsize blank_size;
  gboolean dump_defer;       /* FLUTL_DUMP_DEFER=1: dump with strokes made on worker threads */
  gsize atlas_limit;         /* FLUTL_ATLAS_LIMIT_MB (default 256): see gen_buffer */
  gboolean overlay_meta;     /* FLUTL_OVERLAY_COMPOSITION=1: also attach the frame as a
                              * GstVideoOverlayCompositionMeta (the reference has none,
                              * SURVEY D1/D2; an environment knob, not a property, so
                              * that the element's property surface stays identical) */
  gboolean full_d2h;         /* FLUTL_FULL_D2H=1: whole-frame copies, writable memory */
  gboolean timing;           /* FLUTL_TIMING=1: host / device split on stderr at dispose */
  gboolean timing_trace;     /* FLUTL_TIMING=2: ... and a line per buffer */
  gint64 t_record, t_device;
  gint64 t_first_return, t_last_return;   /* monotonic us: end of the first / last gen_buffer */
  guint n_timed;
  GHashTable *images;        /* id -> cairo_surface_t (decoded PNGs) */
  GHashTable *layouts;       /* layout cache: key string -> PangoLayout, per document */
  gboolean no_layout_cache;  /* FLUTL_NO_LAYOUT_CACHE=1 */
  GHashTable *plans;         /* SpanPlan * (a set): what span_to_region does, by its inputs */
  struct KeyBuf *plan_key;
  gboolean no_span_plans;    /* FLUTL_NO_SPAN_PLANS=1 */
  GPtrArray *region_cache;   /* RegionEntry *: recorded regions by what determines them */
  gboolean no_region_cache;  /* FLUTL_NO_REGION_CACHE=1 */
  gboolean no_recolour;      /* FLUTL_NO_RECOLOUR=1: regions that differ in colours are recorded anew */
  GArray *color_sources;     /* ColorSource: the colour attributes the region being recorded has drawn with */
  guint region_recoloured;   /* statistics (FLUTL_TIMING) */
  guint region_hits, region_misses;
  guint skipped;             /* outlines of this buffer that could not be drawn */
  /* the finisher: buffers are handed out when their list is recorded; this thread
   * waits for the outline masks still being rasterised (flutl_recorder_resolve),
   * uploads new atlas entries and queues the frame on the device, in order.
   * While it has work, it alone uses `ctx`. */
  gboolean sync_mode;        /* FLUTL_SYNC=1: everything on the streaming thread */
  guint pool_prefill;        /* FLUTL_POOL_PREFILL (default 8): frames made ahead */
  gboolean pool_prefill_set;
  guint max_ahead;           /* FLUTL_MAX_AHEAD: buffers ahead of the device (0: by size) */
  guint n_pushed;            /* buffers handed to the finisher since the last reset */
  gboolean queue_always;     /* every frame goes through the finisher (the default since
                              * round 4; FLUTL_QUEUE_ALWAYS=0: a frame with all its masks
                              * there is queued on the device by the streaming thread itself
                              * when the finisher is idle, as in rounds 2 and 3 -- at 7 k
                              * frames/s there was no clear winner; at 20-28 k the 13-16 us
                              * of a submit's runtime calls are a third of that thread's
                              * time per buffer: tools/r4_queue_ab.sh) */
  GThread *finisher;
  GAsyncQueue *jobs;
  GMutex flight_lock;
  GCond flight_cond;
  guint in_flight;
  guint waiting_in_flight;   /* ... of which wait for masks made on worker threads */
  gint64 t_submit, t_flush, t_resolve;   /* (finisher's clocks, read when it is idle) */
  gint64 t_build, t_layout;  /* parts of t_record: spans -> regions, paragraphs -> layouts */
  gint64 t_miss;             /* ... and recording the regions the region cache did not have */
  gint64 t_push_block;       /* streaming thread blocked on the run-ahead limit */
  /* fused over-video: with the optional request pad "video_sink"
   * linked, gen_buffer records cues instead of rendering frames, and every video
   * buffer gets the cue of its time composited into it on the device */
  GstPad *video_pad;
  GstPadEventFunction ttml_event;   /* the base class's handlers of the TTML pad */
  GstPadChainFunction ttml_chain;
  GMutex video_lock, ctx_lock;      /* ctx_lock: the device, between the two streaming threads */
  GCond video_cond;
  gboolean video_have_caps, ttml_eos, video_flushing, video_eos;
  gint video_width, video_height;
  GQueue cues;               /* VideoCue *, in time order */
  guint n_deferred;          /* buffers that went through the finisher */
  guint n_polygons;          /* outlines scan converted on the device (statistics) */
  guint64 polygon_edges;
  gboolean host_strokes;     /* FLUTL_HOST_STROKES=1: outlines rasterised by Cairo on the workers */
  guint64 delta_reported;
} GstTTMLRenderPrivate;

static GstStaticPadTemplate ttmlrender_src_template =
    GST_STATIC_PAD_TEMPLATE ("src", GST_PAD_SRC, GST_PAD_ALWAYS,
        GST_STATIC_CAPS ("video/x-raw, format=BGRA, width=(int)[1,MAX], "
            "height=(int)[1,MAX], framerate=(fraction)0/1, "
            "pixel-aspect-ratio=(fraction)[0/1, MAX]"));

G_DEFINE_TYPE_WITH_PRIVATE (GstTTMLRender, gst_ttmlrender, GST_TYPE_TTMLBASE);

static inline GstTTMLRenderPrivate *
get_priv (GstTTMLRender *render)
{
  return gst_ttmlrender_get_instance_private (render);
}

/* ---- styling that Pango markup cannot express ------------------------------
 * Carried through Pango as integer attributes of four private types. */
typedef enum
{
  HIP_ATTR_OVERLINE,
  HIP_ATTR_INVISIBLE,
  HIP_ATTR_REVERSE,
  HIP_ATTR_REVERSE_OBLIQUE,
  HIP_ATTR_LAST
} HipAttrKind;

static PangoAttribute *hip_attr_copy (const PangoAttribute *attr);
static void hip_attr_destroy (PangoAttribute *attr);
static gboolean hip_attr_equal (const PangoAttribute *a, const PangoAttribute *b);

static PangoAttrClass hip_attr_klass[HIP_ATTR_LAST] = {
  { PANGO_ATTR_INVALID, hip_attr_copy, hip_attr_destroy, hip_attr_equal },
  { PANGO_ATTR_INVALID, hip_attr_copy, hip_attr_destroy, hip_attr_equal },
  { PANGO_ATTR_INVALID, hip_attr_copy, hip_attr_destroy, hip_attr_equal },
  { PANGO_ATTR_INVALID, hip_attr_copy, hip_attr_destroy, hip_attr_equal },
};

static PangoAttribute *
hip_attr_new (const PangoAttrClass *klass, int value, guint start, guint end)
{
  PangoAttrInt *a = g_slice_new (PangoAttrInt);
  pango_attribute_init (&a->attr, klass);
  a->value = value;
  a->attr.start_index = start;
  a->attr.end_index = end;
  return &a->attr;
}

static PangoAttribute *
hip_attr_copy (const PangoAttribute *attr)
{
  return hip_attr_new (attr->klass, ((const PangoAttrInt *) attr)->value,
      attr->start_index, attr->end_index);
}

static void
hip_attr_destroy (PangoAttribute *attr)
{
  g_slice_free (PangoAttrInt, (PangoAttrInt *) attr);
}

static gboolean
hip_attr_equal (const PangoAttribute *a, const PangoAttribute *b)
{
  return ((const PangoAttrInt *) a)->value == ((const PangoAttrInt *) b)->value;
}

/* ---- per-buffer region model ------------------------------------------------ */
typedef struct
{
  gchar *id;
  gint zindex;
  gint ox, oy, ex, ey;       /* origin, extent: integers (reference :49-52) */
  gint pox, poy, pex, pey;   /* after padding */
  guint32 bg;                /* 0xRRGGBBAA */
  gdouble opacity;
  cairo_surface_t *image;
  gint image_x, image_y;
  GstTTMLDisplayAlign display_align;
  gboolean overflow_visible;
  GstTTMLWritingMode writing_mode;
  GstTTMLTextOutline outline; /* reference: a region attribute, from the
                               * first span that reaches the region (:62-64) */
  GList *layouts;            /* PangoLayout */
  GString *par;              /* markup of the paragraph being assembled */
  GstTTMLStyle par_style;    /* (slow path) style of the paragraph's first span ... */
  gboolean par_vals;         /* ... or (span plans) what region_flush_paragraph wants of it */
  gint par_wrap, par_align, par_spacing;
  PangoAttrList *par_attrs;
  GString *par_key;          /* the par_add_attr calls so far, as bytes (layout cache key) */
  gboolean par_unkeyable;    /* ... or attributes that are not plain data (shapes) */
  gint par_plain_len;
  GList *pending_attrs;      /* a region put in for its background (background_region) whose
                              * configuring waits for a span to do it: its attribute stack */
} HipRegion;

#define COMP(c, shift) ((((c) >> (shift)) & 255) / 255.0)

static gint
region_cmp_z (gconstpointer a, gconstpointer b)
{
  return ((const HipRegion *) a)->zindex - ((const HipRegion *) b)->zindex;
}

static gint
region_cmp_id (gconstpointer a, gconstpointer id)
{
  return g_strcmp0 (((const HipRegion *) a)->id, (const gchar *) id);
}

static void
region_free (gpointer data)
{
  HipRegion *r = data;
  if (r->par)
    g_string_free (r->par, TRUE);
  if (r->par_attrs)
    pango_attr_list_unref (r->par_attrs);
  if (r->par_key)
    g_string_free (r->par_key, TRUE);
  gst_ttml_style_reset (&r->par_style);
  g_list_free_full (r->layouts, g_object_unref);
  g_free (r->id);
  g_free (r);
}

/* ---- images (smpte:backgroundImage) ----------------------------------------- */
typedef struct
{
  const guint8 *p;
  gint left;
} MemReader;

static cairo_status_t
mem_read (void *closure, unsigned char *out, unsigned int len)
{
  MemReader *m = closure;
  if ((gint) len > m->left)
    return CAIRO_STATUS_READ_ERROR;
  memcpy (out, m->p, len);
  m->p += len;
  m->left -= len;
  return CAIRO_STATUS_SUCCESS;
}

static cairo_surface_t *
image_lookup (GstTTMLRender *render, const gchar *id)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  cairo_surface_t *s;
  guint8 *data = NULL;
  gint len = 0;
  gboolean owned = FALSE;

  if (!priv->images)
    priv->images = g_hash_table_new_full (g_str_hash, g_str_equal, g_free,
        (GDestroyNotify) cairo_surface_destroy);
  s = g_hash_table_lookup (priv->images, id);
  if (s)
    return s;

  if (id[0] == '#') {
    gst_ttml_state_restore_data (&render->base.state, id + 1, &data, &len);
    if (!data) {
      GST_WARNING_OBJECT (render, "no embedded image '%s'", id + 1);
      return NULL;
    }
  } else {
    /* reference :375-407: URL as is, or relative to the document location */
    gchar *scheme = g_uri_parse_scheme (id), *url;
    if (scheme) {
      url = g_strdup (id);
      g_free (scheme);
    } else {
      gchar *base = gst_ttmlbase_uri_get (render->base.sinkpad);
      gchar *dir = g_path_get_dirname (base);
      url = strncmp (base, "file://", 7) == 0 ? g_build_filename (dir, id, NULL)
                                             : g_strdup_printf ("%s/%s", dir, id);
      g_free (base);
      g_free (dir);
    }
    if (!fludownloader_helper_downloader_download_sync (render->downloader, url,
            &data, &len, NULL)) {
      GST_WARNING_OBJECT (render, "could not retrieve '%s'", id);
      g_free (url);
      return NULL;
    }
    owned = TRUE;
    g_free (url);
  }
  MemReader rd = { data, len };
  s = cairo_image_surface_create_from_png_stream (mem_read, &rd);
  if (owned)
    g_free (data);
  if (!s || cairo_surface_status (s) != CAIRO_STATUS_SUCCESS) {
    GST_WARNING_OBJECT (render, "could not decode image '%s'", id);
    if (s)
      cairo_surface_destroy (s);
    return NULL;
  }
  g_hash_table_insert (priv->images, g_strdup (id), s);
  return s;
}

/* ---- style -> region geometry (reference :434-561) --------------------------- */
static gfloat
norm_len (GstTTMLRender *render, GstTTMLAttribute *attr, int index, int dir)
{
  return gst_ttml_attribute_get_normalized_length (&render->base.state, NULL, attr,
      index, dir, NULL);
}

static void
region_configure (GstTTMLRender *render, HipRegion *r, GstTTMLStyle *style)
{
  GstTTMLAttribute *a;

  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_ZINDEX);
  r->zindex = a ? a->value.i : 0;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_OPACITY);
  r->opacity = a ? a->value.d : 1.0;

  /* lengths are floats, stored into ints: truncation (reference :450-489) */
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_ORIGIN);
  r->ox = a ? norm_len (render, a, 0, 0) : 0;
  r->oy = a ? norm_len (render, a, 1, 1) : 0;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_EXTENT);
  if (a) {
    r->ex = norm_len (render, a, 0, 0);
    r->ey = norm_len (render, a, 1, 1);
  } else {
    r->ex = render->base.state.frame_width;
    r->ey = render->base.state.frame_height;
  }
  r->pox = r->ox;
  r->poy = r->oy;
  r->pex = r->ex;
  r->pey = r->ey;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_PADDING);
  if (a) {
    /* reference: int += float, int -= float + float (each a float->int store) */
    gfloat px = norm_len (render, a, 3, 0), py = norm_len (render, a, 0, 1);
    r->pox += px;
    r->poy += py;
    r->pex -= px + norm_len (render, a, 1, 0);
    r->pey -= py + norm_len (render, a, 2, 1);
  }

  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_BACKGROUND_REGION_COLOR);
  r->bg = a ? a->value.color : 0;

  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_SMPTE_BACKGROUND_IMAGE);
  r->image = (a && a->value.string) ? image_lookup (render, a->value.string) : NULL;
  if (r->image) {
    gint iw = cairo_image_surface_get_width (r->image);
    gint ih = cairo_image_surface_get_height (r->image);
    a = gst_ttml_style_get_attr (style,
        GST_TTML_ATTR_SMPTE_BACKGROUND_IMAGE_HORIZONTAL);
    if (!a)
      r->image_x = r->pox + (r->pex - iw) / 2;
    else if (a->value.raw_length[0].unit == GST_TTML_LENGTH_UNIT_RELATIVE)
      r->image_x = r->pox + (r->pex - iw) * a->value.raw_length[0].f;
    else
      r->image_x = r->pox + a->value.raw_length[0].f;
    a = gst_ttml_style_get_attr (style,
        GST_TTML_ATTR_SMPTE_BACKGROUND_IMAGE_VERTICAL);
    if (!a)
      r->image_y = r->poy + (r->pey - ih) / 2;
    else if (a->value.raw_length[0].unit == GST_TTML_LENGTH_UNIT_RELATIVE)
      r->image_y = r->poy + (r->pey - ih) * a->value.raw_length[0].f;
    else
      r->image_y = r->poy + a->value.raw_length[0].f;
  }

  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_DISPLAY_ALIGN);
  r->display_align = a ? a->value.display_align : render->default_display_align;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_TEXTOUTLINE);
  if (a)
    r->outline = a->value.text_outline;
  else
    r->outline.length[0].unit = GST_TTML_LENGTH_UNIT_NOT_PRESENT;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_OVERFLOW);
  r->overflow_visible = a ? a->value.b : FALSE;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_WRITING_MODE);
  r->writing_mode = a ? a->value.writing_mode : GST_TTML_WRITING_MODE_LRTB;
}

static void par_values_from_style (GstTTMLRender *render, GstTTMLStyle *style, gint *wrap,
    gint *align, gint *spacing);

/* ---- active spans -> region paragraphs (reference :591-939) ------------------ */
static void
par_add_attr (HipRegion *r, PangoAttribute *attr)
{
  /* what the layout cache keys on: the calls, not Pango's merged list (walking
   * that copies every attribute) */
  gint32 rec[6] = { 0, (gint32) attr->start_index, (gint32) attr->end_index, 0, 0, 0 };
  if (!r->par_key)
    r->par_key = g_string_sized_new (8 * sizeof rec);
  if (attr->klass->type == PANGO_ATTR_FOREGROUND || attr->klass->type == PANGO_ATTR_BACKGROUND) {
    const PangoColor *c = &((PangoAttrColor *) attr)->color;
    rec[0] = attr->klass->type == PANGO_ATTR_FOREGROUND ? 1 : 2;
    rec[3] = c->red;
    rec[4] = c->green;
    rec[5] = c->blue;
  } else if (attr->klass >= &hip_attr_klass[0] && attr->klass < &hip_attr_klass[HIP_ATTR_LAST]) {
    rec[0] = 16 + (gint32) (attr->klass - hip_attr_klass);
    rec[3] = ((PangoAttrInt *) attr)->value;
  } else {
    r->par_unkeyable = TRUE;
  }
  g_string_append_len (r->par_key, (const gchar *) rec, sizeof rec);
  if (!r->par_attrs)
    r->par_attrs = pango_attr_list_new ();
  pango_attr_list_change (r->par_attrs, attr);
}

/* reference :861-898: Pango markup has no alpha, so a colour with alpha != 0
 * travels as a Pango colour attribute with the alpha hidden in the LSB of red */
static void
smuggled_color (guint32 c, guint16 *r, guint16 *g, guint16 *b)
{
  *r = ((c >> 16) & 0xFF00) | (c & 0x00FF);
  *g = (c >> 8) & 0xFF00;
  *b = c & 0xFF00;
}

/* Anamorphic font size "w h" (reference :745-857): every character of the span
 * becomes a Pango shape of the horizontally scaled glyph's size; the draw loop
 * later fills the scaled glyph outline itself.  The font is chosen here, from
 * the markup parsed so far -- it may not be the one Pango picks (reference
 * comment :753-756); reproduced as is. */
typedef struct
{
  unsigned long glyph;
  double hscale;
  cairo_scaled_font_t *font;
} HipShape;

static gpointer
hip_shape_copy (gconstpointer data)
{
  const HipShape *s = data;
  HipShape *n = g_new0 (HipShape, 1);
  *n = *s;
  cairo_scaled_font_reference (n->font);
  return n;
}

static void
hip_shape_free (gpointer data)
{
  HipShape *s = data;
  cairo_scaled_font_destroy (s->font);
  g_free (s);
}

static void
anamorphic_shapes (GstTTMLRender *render, HipRegion *r, GstTTMLSpan *span,
    GstTTMLAttribute *a)
{
  PangoAttrList *parsed = NULL;
  PangoAttrIterator *it;
  PangoAttribute *fd;
  PangoFontset *fontset;
  gint start = 0, end = 0;
  const gchar *p = span->chars, *stop = span->chars + span->length;
  double hscale = norm_len (render, a, 0, 0) / norm_len (render, a, 1, 1);

  if (!pango_parse_markup (r->par->str, -1, 0, &parsed, NULL, NULL, NULL) || !parsed)
    return;
  /* the attributes in force on the text just appended */
  it = pango_attr_list_get_iterator (parsed);
  while (end < r->par_plain_len) {
    pango_attr_iterator_next (it);
    pango_attr_iterator_range (it, &start, &end);
  }
  fd = pango_attr_iterator_get (it, PANGO_ATTR_FONT_DESC);
  if (fd) {
    fontset = pango_font_map_load_fontset (pango_cairo_font_map_get_default (),
        render->pango_context, ((PangoAttrFontDesc *) fd)->desc, NULL);
    while (p < stop) {
      const gchar *next = g_utf8_next_char (p);
      gunichar wc = g_utf8_get_char (p);
      PangoFont *font = pango_fontset_get_font (fontset, wc);
      cairo_scaled_font_t *sf =
          pango_cairo_font_get_scaled_font ((PangoCairoFont *) font);
      cairo_glyph_t *glyphs = NULL;
      int n_glyphs = 0;
      PangoRectangle ink, logical;
      HipShape *shape;
      PangoAttribute *sa;

      cairo_scaled_font_text_to_glyphs (sf, 0, 0, p, next - p, &glyphs, &n_glyphs,
          NULL, NULL, NULL);
      pango_font_get_glyph_extents (font, glyphs->index, &ink, &logical);
      /* int *= double: truncation, as in the reference (:816-819) */
      logical.width *= hscale;
      ink.width *= hscale;
      logical.x *= hscale;
      ink.x *= hscale;
      shape = g_new0 (HipShape, 1);
      shape->glyph = glyphs->index;
      shape->hscale = hscale;
      shape->font = cairo_scaled_font_reference (sf);
      cairo_glyph_free (glyphs);
      g_object_unref (font);
      sa = pango_attr_shape_new_with_data (&ink, &logical, shape, hip_shape_copy,
          hip_shape_free);
      sa->start_index = r->par_plain_len + (p - span->chars);
      sa->end_index = r->par_plain_len + (next - span->chars);
      if (!r->par_attrs)
        r->par_attrs = pango_attr_list_new ();
      r->par_unkeyable = TRUE;    /* (shape attributes carry objects: no cached layout) */
      pango_attr_list_insert (r->par_attrs, sa);
      p = next;
    }
    g_object_unref (fontset);
  }
  pango_attr_iterator_destroy (it);
  pango_attr_list_unref (parsed);
}

static void
span_to_region_slow (GstTTMLRender *render, GList **regions, GstTTMLSpan *span)
{
  GstTTMLBase *base = GST_TTMLBASE (render);
  GstTTMLStyle merged;
  GstTTMLAttribute *a;
  const gchar *rid;
  HipRegion *r;
  GList *link;
  gchar *head, *tail, *font_size;
  guint start, end;

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_DISPLAY);
  if (a && !a->value.b)
    return;

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_REGION);
  rid = a ? a->value.string : "anonymous";
  if (a) {
    /* span attributes override the region's own */
    GstTTMLStyle region_style;
    if (!g_hash_table_lookup_extended (base->state.saved_region_attr_stacks, rid,
            NULL, (gpointer *) &region_style.attributes))
      return;                   /* unknown region: the span is dropped */
    gst_ttml_style_copy (&merged, &region_style, FALSE);
    for (link = span->style.attributes; link; link = link->next)
      gst_ttml_attribute_free (
          gst_ttml_style_set_attr (&merged, (GstTTMLAttribute *) link->data));
  } else {
    gst_ttml_style_copy (&merged, &span->style, FALSE);
  }

  link = g_list_find_custom (*regions, rid, region_cmp_id);
  if (link) {
    r = link->data;
    /* reference :647-654: only the first contributor configures the region */
    if (!r->layouts && !r->par) {
      region_configure (render, r, &merged);
      r->pending_attrs = NULL;
    }
  } else {
    r = g_new0 (HipRegion, 1);
    r->id = g_strdup (rid);
    region_configure (render, r, &merged);
    *regions = g_list_insert_sorted (*regions, r, region_cmp_z);
  }

  /* default font size: one cell row (reference :677-685) */
  font_size = render->default_font_size
      ? g_strdup (render->default_font_size)
      : g_strdup_printf (" %dpx ",
            base->state.frame_height / base->state.cell_resolution_y);
  gst_ttml_style_gen_pango_markup (&base->state, &merged, &head, &tail,
      render->default_font_family, font_size);
  g_free (font_size);

  if (span->length) {
    if (!r->par)
      r->par = g_string_new (NULL);
    g_string_append (r->par, head);
    g_string_append_len (r->par, span->chars, span->length);
    g_string_append (r->par, tail);
  }
  g_free (head);
  g_free (tail);

  if (!r->par_style.attributes)
    gst_ttml_style_copy (&r->par_style, &merged, FALSE);

  start = r->par_plain_len;
  end = start + span->length;

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_TEXT_DECORATION);
  if (a && (a->value.text_decoration & GST_TTML_TEXT_DECORATION_OVERLINE))
    par_add_attr (r, hip_attr_new (&hip_attr_klass[HIP_ATTR_OVERLINE], 1, start, end));
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_VISIBILITY);
  if (a && !a->value.b)
    par_add_attr (r, hip_attr_new (&hip_attr_klass[HIP_ATTR_INVISIBLE], 0, start, end));

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_FONT_SIZE);
  if (a && gst_ttml_attribute_is_length_present (a, 1) && r->par)
    anamorphic_shapes (render, r, span, a);

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_BACKGROUND_COLOR);
  if (a && (a->value.color & 0xFF)) {
    guint16 cr, cg, cb;
    PangoAttribute *pa;
    smuggled_color (a->value.color, &cr, &cg, &cb);
    pa = pango_attr_background_new (cr, cg, cb);
    pa->start_index = start;
    pa->end_index = end;
    par_add_attr (r, pa);
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_COLOR);
  if (a && (a->value.color & 0xFF)) {
    guint16 cr, cg, cb;
    PangoAttribute *pa;
    smuggled_color (a->value.color, &cr, &cg, &cb);
    pa = pango_attr_foreground_new (cr, cg, cb);
    pa->start_index = start;
    pa->end_index = end;
    par_add_attr (r, pa);
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_UNICODE_BIDI);
  if (a && a->value.unicode_bidi == GST_TTML_UNICODE_BIDI_OVERRIDE) {
    a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_DIRECTION);
    if (a && a->value.direction == GST_TTML_DIRECTION_RTL)
      par_add_attr (r, hip_attr_new (&hip_attr_klass[HIP_ATTR_REVERSE], 1, start, end));
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_FONT_STYLE);
  if (a && a->value.font_style == GST_TTML_FONT_STYLE_REVERSE_OBLIQUE)
    par_add_attr (r,
        hip_attr_new (&hip_attr_klass[HIP_ATTR_REVERSE_OBLIQUE], 1, start, end));

  r->par_plain_len += span->length;
  gst_ttml_style_reset (&merged);
}

/* ---- span plans ---------------------------------------------------
 * span_to_region_slow above merges the region's attribute stack with the span's,
 * configures the region, generates Pango markup and picks the styling Pango markup
 * cannot express -- ~6 us of copying, freeing and printf per active span and
 * buffer, most of what the streaming thread did on animated documents (cfg2: 8
 * spans, of which one or two change per buffer).  All of it is a function of the
 * span's attributes, its region's attribute stack, the parser's residual style
 * (relative paddings look there, like the reference's) and a few numbers; the
 * result -- a "plan" -- is kept under exactly those bytes and applied.  The text
 * itself is not part of it.  Spans with an anamorphic font size (two lengths) take
 * the slow path: their shapes depend on the paragraph so far. */
typedef struct
{
  guint16 r, g, b;
  gint kind;                    /* 0 overline 1 invisible 2 background 3 foreground
                                 * 4 reverse 5 reverse oblique */
} PlanAttr;

typedef struct
{
  guint8 *key;
  gsize key_len;
  gboolean drop;                /* display none, or an unknown region */
  gchar *rid;
  HipRegion proto;              /* what region_configure makes of the merged style */
  gchar *head, *tail;
  gint par_wrap, par_align, par_spacing;
  guint n_attrs;
  PlanAttr attrs[6];
} SpanPlan;

static void
span_plan_free (gpointer data)
{
  SpanPlan *p = data;
  g_free (p->key);
  g_free (p->rid);
  g_free (p->head);
  g_free (p->tail);
  g_free (p);
}

/* Hash of a cache key (some 600 bytes for a span plan, eight per buffer): four independent
 * multiply chains over 32 bytes at a time -- one chain waited 5 cycles per word for itself. */
static guint
bytes_hash (const guint8 *p, gsize n)
{
  guint64 h0 = 0xcbf29ce484222325ull, h1 = 0x9e3779b97f4a7c15ull, h2 = 0xc2b2ae3d27d4eb4full,
      h3 = 0x165667b19e3779f9ull, h;
  gsize i = 0;
  for (; i + 32 <= n; i += 32) {
    guint64 w[4];
    memcpy (w, p + i, 32);
    h0 = (h0 ^ w[0]) * 0x100000001b3ull;
    h1 = (h1 ^ w[1]) * 0x100000001b3ull;
    h2 = (h2 ^ w[2]) * 0x100000001b3ull;
    h3 = (h3 ^ w[3]) * 0x100000001b3ull;
    h0 ^= h0 >> 29;
    h1 ^= h1 >> 31;
    h2 ^= h2 >> 27;
    h3 ^= h3 >> 33;
  }
  h = (h0 ^ (h1 << 1 | h1 >> 63)) * 0x100000001b3ull;
  h = (h ^ h2) * 0x100000001b3ull;
  h = (h ^ h3) * 0x100000001b3ull;
  for (; i + 8 <= n; i += 8) {
    guint64 w;
    memcpy (&w, p + i, 8);
    h = (h ^ w) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  for (; i < n; i++)
    h = (h ^ p[i]) * 0x100000001b3ull;
  h ^= n;
  return (guint) (h ^ (h >> 32));
}

static guint
span_plan_hash (gconstpointer v)
{
  const SpanPlan *p = v;
  return bytes_hash (p->key, p->key_len);
}

static gboolean
span_plan_equal (gconstpointer a, gconstpointer b)
{
  const SpanPlan *x = a, *y = b;
  return x->key_len == y->key_len && memcmp (x->key, y->key, x->key_len) == 0;
}

/* (a plain growing buffer written with inlined copies: some 80 fields per span and eight
 * spans per buffer went through g_byte_array_append, a tenth of gen_buffer on a stream of
 * cached regions) */
typedef struct KeyBuf
{
  guint8 *data;
  gsize len, cap;
} KeyBuf;

static inline guint8 *
key_room (KeyBuf *k, gsize n)
{
  if (G_UNLIKELY (k->len + n > k->cap)) {
    k->cap = (k->len + n) * 2 + 256;
    k->data = g_realloc (k->data, k->cap);
  }
  k->len += n;
  return k->data + k->len - n;
}

static inline void
key_put (KeyBuf *k, gconstpointer p, gsize n)
{
  memcpy (key_room (k, n), p, n);
}

static inline void
key_put_u32 (KeyBuf *k, guint32 v)
{
  memcpy (key_room (k, 4), &v, 4);
}

static inline void
key_put_string (KeyBuf *k, const gchar *str)
{
  const guint32 n = str ? (guint32) strlen (str) + 1 : 0;
  key_put_u32 (k, n);
  if (n)
    key_put (k, str, n);
}

/* the bytes of an attribute list that determine what it means */
static void
key_put_attrs (KeyBuf *k, GList *attrs)
{
  guint32 n = 0;
  GList *l;
  guint8 *count = key_room (k, 4);
  const gsize count_at = (gsize) (count - k->data);
  for (l = attrs; l; l = l->next) {
    const GstTTMLAttribute *a = l->data;
    n++;
    key_put_u32 (k, (guint32) a->type);
    key_put_string (k, a->value.string);
    /* (the anonymous union after `string`: attributes are made zero-filled and
     * copied whole, so equal values are equal bytes) */
    key_put (k, (const guint8 *) &a->value + sizeof (gchar *),
        sizeof a->value - sizeof (gchar *));
  }
  memcpy (k->data + count_at, &n, 4);     /* (the buffer may have moved) */
}

static SpanPlan *
span_plan_build (GstTTMLRender *render, GstTTMLSpan *span, GList *region_attrs,
    gboolean region_known)
{
  GstTTMLBase *base = GST_TTMLBASE (render);
  SpanPlan *p = g_new0 (SpanPlan, 1);
  GstTTMLStyle merged;
  GstTTMLAttribute *a;
  GList *link;
  gchar *font_size;

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_DISPLAY);
  if (a && !a->value.b) {
    p->drop = TRUE;
    return p;
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_REGION);
  p->rid = g_strdup (a ? a->value.string : "anonymous");
  if (a) {
    GstTTMLStyle region_style;
    if (!region_known) {
      p->drop = TRUE;
      return p;
    }
    region_style.attributes = region_attrs;
    gst_ttml_style_copy (&merged, &region_style, FALSE);
    for (link = span->style.attributes; link; link = link->next)
      gst_ttml_attribute_free (
          gst_ttml_style_set_attr (&merged, (GstTTMLAttribute *) link->data));
  } else {
    gst_ttml_style_copy (&merged, &span->style, FALSE);
  }
  region_configure (render, &p->proto, &merged);
  font_size = render->default_font_size
      ? g_strdup (render->default_font_size)
      : g_strdup_printf (" %dpx ",
            base->state.frame_height / base->state.cell_resolution_y);
  gst_ttml_style_gen_pango_markup (&base->state, &merged, &p->head, &p->tail,
      render->default_font_family, font_size);
  g_free (font_size);
  par_values_from_style (render, &merged, &p->par_wrap, &p->par_align, &p->par_spacing);

#define PLAN_ATTR(k_, r_, g_, b_) do { PlanAttr *pa_ = &p->attrs[p->n_attrs++]; \
    pa_->kind = (k_); pa_->r = (r_); pa_->g = (g_); pa_->b = (b_); } while (0)
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_TEXT_DECORATION);
  if (a && (a->value.text_decoration & GST_TTML_TEXT_DECORATION_OVERLINE))
    PLAN_ATTR (0, 0, 0, 0);
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_VISIBILITY);
  if (a && !a->value.b)
    PLAN_ATTR (1, 0, 0, 0);
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_BACKGROUND_COLOR);
  if (a && (a->value.color & 0xFF)) {
    guint16 cr, cg, cb;
    smuggled_color (a->value.color, &cr, &cg, &cb);
    PLAN_ATTR (2, cr, cg, cb);
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_COLOR);
  if (a && (a->value.color & 0xFF)) {
    guint16 cr, cg, cb;
    smuggled_color (a->value.color, &cr, &cg, &cb);
    PLAN_ATTR (3, cr, cg, cb);
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_UNICODE_BIDI);
  if (a && a->value.unicode_bidi == GST_TTML_UNICODE_BIDI_OVERRIDE) {
    a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_DIRECTION);
    if (a && a->value.direction == GST_TTML_DIRECTION_RTL)
      PLAN_ATTR (4, 0, 0, 0);
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_FONT_STYLE);
  if (a && a->value.font_style == GST_TTML_FONT_STYLE_REVERSE_OBLIQUE)
    PLAN_ATTR (5, 0, 0, 0);
#undef PLAN_ATTR
  gst_ttml_style_reset (&merged);
  return p;
}

/* What span_to_region_slow does to the regions, from the plan (same order). */
static void
span_plan_apply (const SpanPlan *p, GList **regions, GstTTMLSpan *span)
{
  HipRegion *r;
  GList *link;
  guint start, end, i;
  gboolean configure = TRUE;

  if (p->drop)
    return;
  link = g_list_find_custom (*regions, p->rid, region_cmp_id);
  if (link) {
    r = link->data;
    configure = !r->layouts && !r->par;   /* only the first contributor configures */
  } else {
    r = g_new0 (HipRegion, 1);
    r->id = g_strdup (p->rid);
  }
  if (configure) {
    const HipRegion *q = &p->proto;
    r->zindex = q->zindex;
    r->opacity = q->opacity;
    r->ox = q->ox; r->oy = q->oy; r->ex = q->ex; r->ey = q->ey;
    r->pox = q->pox; r->poy = q->poy; r->pex = q->pex; r->pey = q->pey;
    r->bg = q->bg;
    r->image = q->image;
    r->image_x = q->image_x;
    r->image_y = q->image_y;
    r->display_align = q->display_align;
    r->outline = q->outline;
    r->overflow_visible = q->overflow_visible;
    r->writing_mode = q->writing_mode;
    r->pending_attrs = NULL;
  }
  if (!link)
    *regions = g_list_insert_sorted (*regions, r, region_cmp_z);

  if (span->length) {
    /* (the paragraph's buffer at its final size: grown by doubling from nothing it was
     * half a dozen reallocs per region and buffer) */
    const gsize hl = strlen (p->head), tl = strlen (p->tail);
    if (!r->par)
      r->par = g_string_sized_new (hl + span->length + tl + 64);
    g_string_append_len (r->par, p->head, hl);
    g_string_append_len (r->par, span->chars, span->length);
    g_string_append_len (r->par, p->tail, tl);
  }
  if (!r->par_vals && !r->par_style.attributes) {
    r->par_vals = TRUE;
    r->par_wrap = p->par_wrap;
    r->par_align = p->par_align;
    r->par_spacing = p->par_spacing;
  }
  start = r->par_plain_len;
  end = start + span->length;
  for (i = 0; i < p->n_attrs; i++) {
    const PlanAttr *pa = &p->attrs[i];
    PangoAttribute *at;
    switch (pa->kind) {
      case 0:
        at = hip_attr_new (&hip_attr_klass[HIP_ATTR_OVERLINE], 1, start, end);
        break;
      case 1:
        at = hip_attr_new (&hip_attr_klass[HIP_ATTR_INVISIBLE], 0, start, end);
        break;
      case 2:
        at = pango_attr_background_new (pa->r, pa->g, pa->b);
        at->start_index = start;
        at->end_index = end;
        break;
      case 3:
        at = pango_attr_foreground_new (pa->r, pa->g, pa->b);
        at->start_index = start;
        at->end_index = end;
        break;
      case 4:
        at = hip_attr_new (&hip_attr_klass[HIP_ATTR_REVERSE], 1, start, end);
        break;
      default:
        at = hip_attr_new (&hip_attr_klass[HIP_ATTR_REVERSE_OBLIQUE], 1, start, end);
        break;
    }
    par_add_attr (r, at);
  }
  r->par_plain_len += span->length;
}

static void
span_to_region (GstTTMLRender *render, GList **regions, GstTTMLSpan *span)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  GstTTMLBase *base = GST_TTMLBASE (render);
  GstTTMLAttribute *a;
  GList *region_attrs = NULL;
  gboolean region_known = FALSE;
  KeyBuf *k;
  SpanPlan probe, *p;
  gint32 ctx[6];

  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_FONT_SIZE);
  if (priv->no_span_plans || (a && gst_ttml_attribute_is_length_present (a, 1))) {
    span_to_region_slow (render, regions, span);
    return;
  }
  a = gst_ttml_style_get_attr (&span->style, GST_TTML_ATTR_REGION);
  if (a && a->value.string)
    region_known = g_hash_table_lookup_extended (base->state.saved_region_attr_stacks,
        a->value.string, NULL, (gpointer *) &region_attrs);
  /* the plan's key: everything the slow path reads */
  if (!priv->plan_key)
    priv->plan_key = g_new0 (KeyBuf, 1);
  k = priv->plan_key;
  k->len = 0;
  ctx[0] = base->state.frame_width;
  ctx[1] = base->state.frame_height;
  ctx[2] = base->state.cell_resolution_x;
  ctx[3] = base->state.cell_resolution_y;
  ctx[4] = (gint32) render->default_display_align;
  ctx[5] = region_known;
  key_put (k, ctx, sizeof ctx);
  key_put_string (k, render->default_font_family);
  key_put_string (k, render->default_font_size);
  ctx[0] = (gint32) render->default_text_align;
  key_put (k, ctx, 4);
  key_put_attrs (k, span->style.attributes);
  key_put_attrs (k, region_attrs);
  key_put_attrs (k, base->state.style.attributes);
  probe.key = k->data;
  probe.key_len = k->len;
  if (!priv->plans)
    priv->plans = g_hash_table_new_full (span_plan_hash, span_plan_equal, NULL, span_plan_free);
  p = g_hash_table_lookup (priv->plans, &probe);
  if (!p) {
    if (g_hash_table_size (priv->plans) >= 1024)   /* bound it: start over */
      g_hash_table_remove_all (priv->plans);
    p = span_plan_build (render, span, region_attrs, region_known);
    p->key = g_memdup (k->data, k->len);
    p->key_len = k->len;
    g_hash_table_add (priv->plans, p);
  }
  span_plan_apply (p, regions, span);
}

/* Regions whose background shows even without text (reference :1400-1425). */
static void
background_region (gpointer key, gpointer value, gpointer user_data)
{
  gpointer *ctx = user_data;
  GstTTMLRender *render = ctx[0];
  GList **regions = ctx[1];
  GstTTMLStyle style;
  GstTTMLAttribute *a;

  style.attributes = value;
  a = gst_ttml_style_get_attr (&style, GST_TTML_ATTR_SHOW_BACKGROUND);
  if (a && a->value.show_background != GST_TTML_SHOW_BACKGROUND_ALWAYS)
    return;
  if (!g_list_find_custom (*regions, key, region_cmp_id)) {
    /* Its place in the paint order now (the z index), the other dozen attributes when no
     * span turns out to configure the region (gen_buffer, after the spans): the first span
     * that reaches a region configures it over again anyway (:647-654), from its own merged
     * style -- on a stream of eight regions with text, eight region_configure calls per
     * buffer for nothing. */
    HipRegion *r = g_new0 (HipRegion, 1);
    r->id = g_strdup (key);
    a = gst_ttml_style_get_attr (&style, GST_TTML_ATTR_ZINDEX);
    r->zindex = a ? a->value.i : 0;
    r->pending_attrs = value;
    *regions = g_list_insert_sorted (*regions, r, region_cmp_z);
  }
}

/* Layout cache: the reference rebuilds every PangoLayout -- markup
 * parse, itemisation, HarfBuzz shaping, line breaking -- for every buffer
 * (:1462-1472), although consecutive scene states mostly differ in one
 * attribute of one region.  A layout is a function of the paragraph's markup,
 * its extra attributes and a few numbers; layouts are kept per document under
 * that key and shared (they are never modified after creation).  Returns FALSE
 * for paragraphs whose attributes are not plain data (anamorphic shapes). */
/* A colour attribute of a paragraph as the draw loop meets it: foreground (1) or
 * background (2) and the Pango colour the alpha travels in (smuggled_color). */
typedef struct
{
  guint16 type, r, g, b;
} ColorSource;

/* What a layout is apart from its colours: its cache key with the colour values
 * of the attribute records (par_add_attr) blanked, and those values in order.
 * Two layouts of equal shape differ in nothing but colour attributes over the
 * same ranges -- same text, fonts, breaks, glyph positions. */
typedef struct
{
  guint8 *key;
  gsize len;
  guint64 hash;
  ColorSource *colors;
  guint n_colors;
} LayoutShape;

static void
layout_shape_free (gpointer data)
{
  LayoutShape *sh = data;
  g_free (sh->key);
  g_free (sh->colors);
  g_free (sh);
}

static GQuark
layout_shape_quark (void)
{
  static GQuark q;
  if (G_UNLIKELY (!q))
    q = g_quark_from_static_string ("flutl-hip-layout-shape");
  return q;
}

/* key: what layout_cache_key made (9 x gint32, the records, the markup) */
static void
layout_attach_shape (PangoLayout *layout, const GString *key)
{
  LayoutShape *sh = g_new0 (LayoutShape, 1);
  gint32 rec_len;
  guint64 h = 0xcbf29ce484222325ull;
  sh->len = key->len;
  sh->key = g_memdup (key->str, key->len);
  memcpy (&rec_len, key->str + 8 * sizeof (gint32), sizeof rec_len);
  sh->colors = g_new (ColorSource, rec_len / (6 * sizeof (gint32)) + 1);
  for (gsize off = 9 * sizeof (gint32); off + 6 * sizeof (gint32) <= 9 * sizeof (gint32) + (gsize) rec_len;
      off += 6 * sizeof (gint32)) {
    gint32 rec[6];
    memcpy (rec, sh->key + off, sizeof rec);
    if (rec[0] == 1 || rec[0] == 2) {
      ColorSource *c = &sh->colors[sh->n_colors++];
      c->type = (guint16) rec[0];
      c->r = (guint16) rec[3];
      c->g = (guint16) rec[4];
      c->b = (guint16) rec[5];
      rec[3] = rec[4] = rec[5] = 0;
      memcpy (sh->key + off, rec, sizeof rec);
    }
  }
  /* ... and the colours of the markup (gst_ttml_style_gen_pango_markup,
   * gstttmlstyle.c:160-170: fgcolor="#RRGGBB" unless white, bgcolor unless black;
   * text is escaped, so a raw quote only ever belongs to the markup).  Pango makes
   * 16-bit components of them by repeating the byte; the colour attributes above
   * override them wherever the alpha is not 0. */
  {
    const gsize m0 = 9 * sizeof (gint32) + (gsize) rec_len;
    guint n_markup = 0;
    for (gsize i = m0; i + 18 <= sh->len; i++)
      if (sh->key[i] == ' ' && (sh->key[i + 1] == 'f' || sh->key[i + 1] == 'b') &&
          memcmp (sh->key + i + 2, "gcolor=\"#", 9) == 0 && sh->key[i + 17] == '"')
        n_markup++;
    sh->colors = g_renew (ColorSource, sh->colors, sh->n_colors + n_markup + 1);
    for (gsize i = m0; i + 18 <= sh->len; i++) {
      if (sh->key[i] != ' ' || (sh->key[i + 1] != 'f' && sh->key[i + 1] != 'b') ||
          memcmp (sh->key + i + 2, "gcolor=\"#", 9) != 0 || sh->key[i + 17] != '"')
        continue;
      guint v[3];
      gboolean ok = TRUE;
      for (guint c = 0; c < 3; c++) {
        const gint hi = g_ascii_xdigit_value (sh->key[i + 11 + 2 * c]);
        const gint lo = g_ascii_xdigit_value (sh->key[i + 12 + 2 * c]);
        ok = ok && hi >= 0 && lo >= 0;
        v[c] = (guint) (hi * 16 + lo);
      }
      if (!ok)
        continue;
      ColorSource *c = &sh->colors[sh->n_colors++];
      c->type = sh->key[i + 1] == 'f' ? 1 : 2;
      c->r = (guint16) (v[0] << 8 | v[0]);
      c->g = (guint16) (v[1] << 8 | v[1]);
      c->b = (guint16) (v[2] << 8 | v[2]);
      memset (sh->key + i + 11, '0', 6);
    }
  }
  for (gsize i = 0; i < sh->len; i++)
    h = (h ^ sh->key[i]) * 0x100000001b3ull;
  sh->hash = h;
  g_object_set_qdata_full (G_OBJECT (layout), layout_shape_quark (), sh, layout_shape_free);
}

static gboolean
layout_cache_key (GString *key, PangoContext *ctx, HipRegion *r, gint w, gint h,
    gboolean wrap, PangoAlignment align, gint spacing)
{
  /* (the context's gravity is part of the key: the reference sets it for the
   * vertical writing modes and never sets it back, and so do we) */
  const gint32 head[9] = { w, h, wrap, (gint32) align, spacing, (gint32) r->writing_mode,
    (gint32) pango_context_get_base_gravity (ctx), (gint32) pango_context_get_gravity_hint (ctx),
    r->par_key ? (gint32) r->par_key->len : 0 };
  if (r->par_unkeyable)
    return FALSE;
  g_string_append_len (key, (const gchar *) head, sizeof head);
  if (r->par_key)
    g_string_append_len (key, r->par_key->str, r->par_key->len);
  g_string_append_len (key, r->par->str, r->par->len);
  return TRUE;
}

static guint
layout_key_hash (gconstpointer v)
{
  const GString *k = v;
  return bytes_hash ((const guint8 *) k->str, k->len);
}

static gboolean
layout_key_equal (gconstpointer a, gconstpointer b)
{
  const GString *x = a, *y = b;
  return x->len == y->len && memcmp (x->str, y->str, x->len) == 0;
}

static void
layout_key_free (gpointer v)
{
  g_string_free (v, TRUE);
}

/* What a paragraph takes from the style of its first span: wrap option, text
 * alignment (a GstTTMLTextAlign) and line height (reference: smuggled to the draw
 * loop as the layout spacing; -1 means "use the line's own extents"). */
static void
par_values_from_style (GstTTMLRender *render, GstTTMLStyle *style, gint *wrap, gint *align,
    gint *spacing)
{
  GstTTMLAttribute *a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_WRAP_OPTION);
  *wrap = a ? (gint) a->value.wrap_option : (gint) GST_TTML_WRAP_OPTION_YES;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_TEXT_ALIGN);
  *align = a ? (gint) a->value.text_align : (gint) render->default_text_align;
  a = gst_ttml_style_get_attr (style, GST_TTML_ATTR_LINE_HEIGHT);
  *spacing = (a && gst_ttml_attribute_is_length_present (a, 0)) ? (gint) norm_len (render, a, 0, 1)
                                                                : -1;
}

/* Paragraph -> PangoLayout (reference :199-289). */
static void
region_flush_paragraph (GstTTMLRender *render, HipRegion *r)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  PangoLayout *layout;
  GstTTMLAttribute *a;
  gint w = r->pex, h = r->pey, spacing = -1;
  GstTTMLWrapOption wrap = GST_TTML_WRAP_OPTION_YES;
  PangoAlignment align = PANGO_ALIGN_LEFT;
  GString *key = NULL;

  if (r->writing_mode > GST_TTML_WRITING_MODE_RLTB) {
    w = r->pey;
    h = r->pex;
  }
  if (!r->par_vals)
    par_values_from_style (render, &r->par_style, &r->par_wrap, &r->par_align, &r->par_spacing);
  wrap = r->par_wrap;
  spacing = r->par_spacing;
  switch (r->par_align) {
    case GST_TTML_TEXT_ALIGN_RIGHT:
    case GST_TTML_TEXT_ALIGN_END:
      align = PANGO_ALIGN_RIGHT;
      break;
    case GST_TTML_TEXT_ALIGN_CENTER:
      align = PANGO_ALIGN_CENTER;
      break;
    default:
      break;
  }
  if (!priv->no_layout_cache) {
    /* (at its final size, more or less: see span_plan_apply) */
    key = g_string_sized_new ((r->par ? r->par->len : 0) + (r->par_key ? r->par_key->len : 0) + 192);
    if (!layout_cache_key (key, render->pango_context, r, w, h,
            wrap == GST_TTML_WRAP_OPTION_YES, align, spacing)) {
      g_string_free (key, TRUE);
      key = NULL;
    }
  }
  layout = key && priv->layouts ? g_hash_table_lookup (priv->layouts, key) : NULL;
  if (layout) {
    g_object_ref (layout);
    g_string_free (key, TRUE);
    if (r->par_attrs) {
      pango_attr_list_unref (r->par_attrs);
      r->par_attrs = NULL;
    }
    goto done;
  }

  layout = pango_layout_new (render->pango_context);
  if (wrap == GST_TTML_WRAP_OPTION_YES)
    pango_layout_set_width (layout, w * PANGO_SCALE);
  pango_layout_set_height (layout, h * PANGO_SCALE);
  pango_layout_set_alignment (layout, align);
  pango_layout_set_spacing (layout, spacing);

  pango_layout_set_markup (layout, r->par->str, -1);

  if (r->par_attrs) {
    PangoAttrList *dst = pango_layout_get_attributes (layout);
    PangoAttrIterator *it = pango_attr_list_get_iterator (r->par_attrs);
    do {
      GSList *l, *attrs = pango_attr_iterator_get_attrs (it);
      for (l = attrs; l; l = l->next)
        pango_attr_list_change (dst, l->data);
      g_slist_free (attrs);
    } while (pango_attr_iterator_next (it));
    pango_attr_iterator_destroy (it);
    pango_layout_set_attributes (layout, dst);
    pango_attr_list_unref (r->par_attrs);
    r->par_attrs = NULL;
  }

  if (key) {
    layout_attach_shape (layout, key);
    if (!priv->layouts)
      priv->layouts = g_hash_table_new_full (layout_key_hash, layout_key_equal, layout_key_free,
          g_object_unref);
    if (g_hash_table_size (priv->layouts) >= 512)   /* bound it: start over */
      g_hash_table_remove_all (priv->layouts);
    g_hash_table_insert (priv->layouts, key, g_object_ref (layout));
  }

done:
  r->layouts = g_list_append (r->layouts, layout);
  g_string_free (r->par, TRUE);
  r->par = NULL;
  r->par_plain_len = 0;
  if (r->par_key)
    g_string_truncate (r->par_key, 0);
  r->par_unkeyable = FALSE;
  r->par_vals = FALSE;
  gst_ttml_style_reset (&r->par_style);
}

/* ---- layout -> primitives (reference :941-1176) ------------------------------
 * fill == TRUE : glyphs and rectangles are recorded as primitives;
 * fill == FALSE: glyph outlines and rectangles are appended to the recorder's
 *                path; span backgrounds still fill (after flushing the pending
 *                path with a stroke), exactly as the reference's "render"
 *                flag does in the outline pass. */
static void
set_smuggled_source (FlutlRecorder *rec, const PangoColor *c)
{
  /* reference :1009-1012: red uses its high byte / 65280, green and blue the
   * full 16 bits / 65535, alpha is the low byte of red / 255 */
  flutl_recorder_set_color (rec, (c->red & 0xFF00) / 65280.0, c->green / 65535.0,
      c->blue / 65535.0, (c->red & 0x00FF) / 255.0);
}

static guint32
smuggled_pixel (const ColorSource *c)
{
  return flutl_recorder_pixel_of ((c->r & 0xFF00) / 65280.0, c->g / 65535.0, c->b / 65535.0,
      (c->r & 0x00FF) / 255.0);
}

/* ... and the primitives drawn from here on say which colour attribute of the
 * region they took their colour from (1-based into priv->color_sources) */
static void
set_attr_source (GstTTMLRender *render, FlutlRecorder *rec, guint16 type, const PangoColor *c)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  GArray *a = priv->color_sources;
  guint i;
  set_smuggled_source (rec, c);
  if (!a)
    a = priv->color_sources = g_array_new (FALSE, FALSE, sizeof (ColorSource));
  for (i = 0; i < a->len; i++) {
    const ColorSource *s = &g_array_index (a, ColorSource, i);
    if (s->type == type && s->r == c->red && s->g == c->green && s->b == c->blue)
      break;
  }
  if (i == a->len) {
    ColorSource s = { type, c->red, c->green, c->blue };
    g_array_append_val (a, s);
  }
  flutl_recorder_set_color_tag (rec, (guint16) MIN (i + 1, 0xFFFF));
}

static void
rect_or_path (FlutlRecorder *rec, gboolean fill, double x, double y, double w,
    double h)
{
  if (fill) {
    flutl_recorder_fill_rect (rec, x, y, w, h);
  } else {
    flutl_recorder_path_rect (rec, x, y, w, h);
    flutl_recorder_stroke (rec);
  }
}

static void
walk_layout (GstTTMLRender *render, FlutlRecorder *rec, PangoLayout *layout,
    gboolean fill, int right_edge, gboolean show_background,
    GstTTMLWritingMode writing_mode)
{
  cairo_t *cr = flutl_recorder_cairo (rec);
  int n_lines = pango_layout_get_line_count (layout);
  int spacing = pango_layout_get_spacing (layout);
  int baseline = pango_layout_get_baseline (layout) / PANGO_SCALE;
  int xoffset = 0, li;
  /* The reference moves its Cairo context about with a save / translate /
   * restore / translate per glyph.  Only the CTM matters to what is recorded, and
   * saving a context copies a whole graphics state (references on the target, the
   * source, the clip, the font...): a third of the time a new region took.  The
   * matrix is tracked here with Cairo's own arithmetic -- cairo_translate IS
   * cairo_matrix_translate on the CTM (_cairo_gstate_translate), a restore is a
   * copy -- and handed to the context only where something reads it. */
  cairo_matrix_t m, stack[3];
  /* ... and its inverse, the way the context keeps it (_cairo_gstate_translate:
   * inverse = inverse x translate (-tx, -ty)): cairo_move_to + cairo_show_text, which
   * Pango draws the digits of a hex box with, go through it.  The context's own
   * inverse at this point is the reference's (the calls so far were the
   * reference's); it is read out exactly: x * 1 + y * 0 is x. */
  cairo_matrix_t minv, stack_inv[3];
  gboolean m_dirty = FALSE;
  cairo_get_matrix (cr, &m);
  {
    double ax = 1, ay = 0, bx = 0, by = 1, ox = 0, oy = 0;
    cairo_device_to_user_distance (cr, &ax, &ay);
    cairo_device_to_user_distance (cr, &bx, &by);
    cairo_device_to_user (cr, &ox, &oy);
    cairo_matrix_init (&minv, ax, ay, bx, by, ox, oy);
  }
  flutl_recorder_set_ctm_inverse (rec, &minv);
#define CTM_SYNC() G_STMT_START { if (m_dirty) { cairo_set_matrix (cr, &m); m_dirty = FALSE; } } G_STMT_END
#define CTM_TRANSLATE(tx, ty) G_STMT_START { \
    cairo_matrix_t t_; \
    cairo_matrix_translate (&m, (tx), (ty)); \
    cairo_matrix_init_translate (&t_, -(tx), -(ty)); \
    cairo_matrix_multiply (&minv, &minv, &t_); \
    m_dirty = TRUE; } G_STMT_END

  if (pango_layout_get_width (layout) == -1) {
    /* no wrapping: alignment is applied by hand against the region width */
    PangoRectangle ext;
    pango_layout_get_pixel_extents (layout, NULL, &ext);
    if (pango_layout_get_alignment (layout) == PANGO_ALIGN_RIGHT)
      xoffset = right_edge - ext.width;
    if (pango_layout_get_alignment (layout) == PANGO_ALIGN_CENTER)
      xoffset = (right_edge - ext.width) / 2;
  }

  for (li = 0; li < n_lines; li++) {
    PangoLayoutLine *line = pango_layout_get_line_readonly (layout, li);
    int pre, post, *ranges, n_ranges, line_dx;
    GSList *rl;

    pango_layout_line_get_x_ranges (line, line->start_index,
        line->start_index + line->length, &ranges, &n_ranges);
    if (spacing == -1) {
      PangoRectangle ext;
      pango_layout_line_get_pixel_extents (line, NULL, &ext);
      pre = -ext.y;
      post = ext.y + ext.height;
    } else {
      pre = baseline;
      post = spacing - baseline;
    }
    line_dx = xoffset + ranges[0] / PANGO_SCALE;     /* integer division */
    g_free (ranges);

    CTM_TRANSLATE (line_dx, writing_mode != GST_TTML_WRITING_MODE_TBLR ? pre : -post);
    stack[0] = m;
    stack_inv[0] = minv;
    for (rl = line->runs; rl; rl = rl->next) {
      PangoGlyphItem *run = rl->data;
      PangoGlyphString *gs = run->glyphs;
      GSList *al;
      double width = 0.0;
      gboolean skip = FALSE, oblique = FALSE;
      gboolean reverse = writing_mode == GST_TTML_WRITING_MODE_RLTB;
      int i;

      for (i = 0; i < gs->num_glyphs; i++)
        width += gs->glyphs[i].geometry.width;
      width /= PANGO_SCALE;
      stack[1] = m;
      stack_inv[1] = minv;
      double saved_color[4];
      const guint16 saved_tag = flutl_recorder_get_color_tag (rec);
      flutl_recorder_get_color (rec, saved_color);

      /* pass 1: colours and span background */
      for (al = run->item->analysis.extra_attrs; al; al = al->next) {
        PangoAttribute *attr = al->data;
        if (attr->klass->type == PANGO_ATTR_FOREGROUND) {
          set_attr_source (render, rec, 1, &((PangoAttrColor *) attr)->color);
        } else if (attr->klass->type == PANGO_ATTR_BACKGROUND) {
          double keep[4];
          const guint16 keep_tag = flutl_recorder_get_color_tag (rec);
          if (!show_background)
            continue;
          CTM_SYNC ();
          if (!fill)
            flutl_recorder_stroke (rec);   /* flush the outline path so far */
          flutl_recorder_get_color (rec, keep);
          set_attr_source (render, rec, 2, &((PangoAttrColor *) attr)->color);
          flutl_recorder_fill_rect (rec, 0, -pre, width, pre + post);
          flutl_recorder_set_color (rec, keep[0], keep[1], keep[2], keep[3]);
          flutl_recorder_set_color_tag (rec, keep_tag);
        } else if (attr->klass == &hip_attr_klass[HIP_ATTR_REVERSE]) {
          reverse = TRUE;
        }
      }

      /* pass 2: decorations, visibility */
      for (al = run->item->analysis.extra_attrs; al; al = al->next) {
        PangoAttribute *attr = al->data;
        PangoFontMetrics *fm;
        double thick, pos;
        CTM_SYNC ();
        switch (attr->klass->type) {
          case PANGO_ATTR_UNDERLINE:
            fm = pango_font_get_metrics (run->item->analysis.font, NULL);
            thick = pango_font_metrics_get_underline_thickness (fm);
            pos = pango_font_metrics_get_underline_position (fm);
            pango_font_metrics_unref (fm);
            rect_or_path (rec, fill, 0, -pos / PANGO_SCALE, width, thick / PANGO_SCALE);
            break;
          case PANGO_ATTR_STRIKETHROUGH:
            fm = pango_font_get_metrics (run->item->analysis.font, NULL);
            thick = pango_font_metrics_get_strikethrough_thickness (fm);
            pos = pango_font_metrics_get_strikethrough_position (fm);
            pango_font_metrics_unref (fm);
            rect_or_path (rec, fill, 0, -pos / PANGO_SCALE, width, thick / PANGO_SCALE);
            break;
          default:
            if (attr->klass == &hip_attr_klass[HIP_ATTR_OVERLINE]) {
              fm = pango_font_get_metrics (run->item->analysis.font, NULL);
              thick = pango_font_metrics_get_underline_thickness (fm);
              pos = pango_font_metrics_get_ascent (fm);
              pango_font_metrics_unref (fm);
              rect_or_path (rec, fill, 0, -pos / PANGO_SCALE, width,
                  thick / PANGO_SCALE);
            } else if (attr->klass == &hip_attr_klass[HIP_ATTR_INVISIBLE]) {
              skip = TRUE;
            } else if (attr->klass == &hip_attr_klass[HIP_ATTR_REVERSE_OBLIQUE]) {
              oblique = TRUE;
            }
            if (attr->klass->type == PANGO_ATTR_SHAPE) {
              /* anamorphic glyph (reference :1101-1120): the scaled outline is
               * filled (or stroked, in the outline pass) as a path */
              HipShape *shape = ((PangoAttrShape *) attr)->data;
              flutl_recorder_path_scaled_glyph (rec, shape->font, shape->glyph,
                  shape->hscale);
              if (fill)
                flutl_recorder_fill (rec);
              else
                flutl_recorder_stroke (rec);
              skip = TRUE;
            }
            break;
        }
      }

      if (!skip) {
        /* one glyph at a time, like the reference (:1138-1161) */
        int n = gs->num_glyphs;
        for (i = 0; i < n; i++) {
          PangoGlyphInfo *gi = &gs->glyphs[reverse ? n - 1 - i : i];
          stack[2] = m;
          stack_inv[2] = minv;
          if (oblique) {
            /* reference :1125, :1141-1143: reverseOblique = x += y / 4
             * (cairo_transform: ctm = shear x ctm, _cairo_gstate_transform) */
            cairo_matrix_t shear, unshear;
            cairo_matrix_init (&shear, 1, 0, 0.25, 1, 0, 0);
            unshear = shear;
            cairo_matrix_invert (&unshear);
            cairo_matrix_multiply (&m, &shear, &m);
            cairo_matrix_multiply (&minv, &minv, &unshear);
            m_dirty = TRUE;
          }
          /* reference: the glyph offsets are applied here AND again by Pango */
          CTM_TRANSLATE (gi->geometry.x_offset / (double) PANGO_SCALE,
              gi->geometry.y_offset / (double) PANGO_SCALE);
          /* (a glyph of the font: its CTM handed over; Pango's boxes and arrows for a glyph
           * no font has are drawn with calls on the context, which then needs it) */
          if (gi->glyph & PANGO_GLYPH_UNKNOWN_FLAG)
            CTM_SYNC ();
          else
            flutl_recorder_set_ctm (rec, &m);
          if (fill) {
            flutl_recorder_show_glyph (rec, run->item->analysis.font, gi);
          } else {
            flutl_recorder_path_glyph (rec, run->item->analysis.font, gi);
          }
          flutl_recorder_set_ctm (rec, NULL);
          m = stack[2];
          minv = stack_inv[2];
          CTM_TRANSLATE (gi->geometry.width / (double) PANGO_SCALE, 0);
        }
      }
      m = stack[1];
      minv = stack_inv[1];
      m_dirty = TRUE;
      /* (restoring the context also restores the source colour) */
      flutl_recorder_set_color (rec, saved_color[0], saved_color[1], saved_color[2],
          saved_color[3]);
      flutl_recorder_set_color_tag (rec, saved_tag);
      CTM_TRANSLATE (width, 0);
    }
    m = stack[0];
    minv = stack_inv[0];
    CTM_TRANSLATE (-line_dx, writing_mode != GST_TTML_WRITING_MODE_TBLR ? post : -pre);
  }
  CTM_SYNC ();
  flutl_recorder_set_ctm_inverse (rec, NULL);
#undef CTM_SYNC
#undef CTM_TRANSLATE
}

static gboolean flush_atlas (GstTTMLRender *render);
static gboolean ensure_ctx (GstTTMLRender *render);
static void pipeline_drain (GstTTMLRender *render);
static gboolean video_cues_drained (GstTTMLRender *render);
static void video_cue_push (GstTTMLRender *render, GstClockTime ts, GstClockTime duration,
    const flutl_region *rg, guint32 n_rg, const flutl_prim *pr, guint32 n_pr);

/* Glyph / outline / blurred-image masks, host copy and HBM: everything that is
 * re-made on demand. */
static void
caches_drop (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  pipeline_drain (render);      /* no frame may still want a mask or the device */
  if (priv->region_cache)       /* (its primitives name atlas offsets) */
    g_ptr_array_set_size (priv->region_cache, 0);
  if (priv->rec)
    flutl_recorder_reset (priv->rec);
  if (priv->sub)
    flutl_recorder_reset (priv->sub);
  if (priv->blurred) {
    g_hash_table_unref (priv->blurred);
    priv->blurred = NULL;
  }
  if (priv->ctx) {
    flutl_ctx_sync (priv->ctx);
    flutl_atlas_reset (priv->ctx);
  }
}

/* Blurred text outline (reference :1193-1229 + gstttmlblur.c): the outline is
 * stroked onto a cleared temporary ARGB32 surface, that surface is blurred,
 * and the result is blitted under the text.  Here: a child recorder records
 * the temporary surface, the device composites + blurs it into an atlas entry
 * (cached per document: it only changes when the outline itself does), and
 * the frame's display list blits that entry. */
static void
blurred_outline (GstTTMLRender *render, HipRegion *r, PangoLayout *layout, guint32 c)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  FlutlRecorder *rec = priv->rec, *sub = priv->sub;
  cairo_matrix_t m;
  PangoRectangle rect;
  const flutl_region *rg;
  const flutl_prim *pr;
  uint32_t n_rg, n_pr;
  BlurredImage *img;
  gchar *key;
  GChecksum *sum;
  int br = (int) ceilf (r->outline.length[1].f);
  /* reference :1216-1217: (surface, int radius = blur * 2, double sigma = blur) */
  int radius = (int) (r->outline.length[1].f * 2);
  double sigma = r->outline.length[1].f;
  int tw, th, dst_x, dst_y;
  guint32 distx, disty;
  gboolean fractional, rotated;
  int32_t box[4] = { 0, 0, 0, 0 };

  pango_layout_get_pixel_extents (layout, NULL, &rect);
  tw = rect.width + 2 * br;
  th = rect.height + 2 * br;
  cairo_get_matrix (flutl_recorder_cairo (rec), &m);
  /* the vertical writing modes paint through a quarter-turn CTM (:1313-1340;
   * cos (pi/2) is 6e-17 there, not 0) */
  rotated = fabs (m.xx) < 1e-9 && fabs (m.yy) < 1e-9 && m.xy == -1.0 && m.yx == 1.0;
  if ((!rotated && (m.xx != 1.0 || m.yy != 1.0 || m.xy != 0.0 || m.yx != 0.0)) ||
      radius > FLUTL_BLUR_MAX_RADIUS || radius < 0 || !(sigma > 0.0) || tw <= 0 || th <= 0 ||
      tw > 32766 || th > 32766 || fabs (m.x0) > 30000.0 || fabs (m.y0) > 30000.0) {
    /* (no CTM other than a translation or the quarter turn of the vertical writing
     * modes ever reaches this point; the radius' limit is the library's, flutl.h: twice
     * the radius at which the reference's kernel no longer fits its thread's stack) */
    GST_FIXME_OBJECT (render, "blurred textOutline of region '%s': a blur of %g px over "
        "%dx%d px is beyond what is implemented; the buffer is dropped", r->id,
        r->outline.length[1].f, tw, th);
    priv->skipped++;
    return;
  }
  if (rotated) {
    /* device (x, y) = (x0 - user_y, y0 + user_x): source pixel (i, j) of the
     * surface set at user (px, py) lands on device pixel (x0 - py - j - 1,
     * y0 + px + i).  The position is an integer here -- the reference REPLACES
     * the CTM for these modes (cairo_set_matrix, :1313-1327), which discards the
     * only fractional translation there is (displayAlign center, :1303-1309) --
     * so Cairo samples with NEAREST: the blit is the surface turned a quarter
     * clockwise. */
    const double U0 = m.y0 + (rect.x - br), V0 = m.x0 - (rect.y - br);
    if (fabs (U0 - floor (U0 + 0.5)) > 1e-9 || fabs (V0 - floor (V0 + 0.5)) > 1e-9) {
      GST_FIXME_OBJECT (render, "blurred textOutline at a fractional position in a "
          "vertical writing mode; outline of region '%s' not drawn", r->id);
      priv->skipped++;
      return;
    }
    dst_x = (int) floor (V0 + 0.5) - th;
    dst_y = (int) floor (U0 + 0.5);
    fractional = FALSE;
    distx = disty = 0;
  } else
  /* reference :1219-1224: the blurred surface is set as source at the INTEGER
   * user position (rect.x - br, rect.y - br) and a rectangle of its extent is
   * filled under the CTM.  If the CTM offset is fractional (e.g. displayAlign
   * center: posy = diff / 2.0) Cairo (a) resamples the surface with pixman's
   * bilinear filter -- constant weights, the transform being a pure
   * translation: translation in 16.16 rounded to nearest, 7 bits of its
   * fraction -- and (b) antialiases the box edges.  Both pinned against the
   * real libraries by tests/golden/kats.npz fracblit_*. */
  {
    const double X0 = m.x0 + (rect.x - br), Y0 = m.y0 + (rect.y - br);
    const gint64 fx = (gint64) floor (-X0 * 65536.0 + 0.5);
    const gint64 fy = (gint64) floor (-Y0 * 65536.0 + 0.5);
    fractional = X0 != floor (X0) || Y0 != floor (Y0);
    distx = (guint32) ((fx >> 9) & 0x7f);
    disty = (guint32) ((fy >> 9) & 0x7f);
    if (fractional) {           /* resampled entry: one pixel more, starts one earlier */
      dst_x = (int) (-1 - (fx >> 16));
      dst_y = (int) (-1 - (fy >> 16));
      box[0] = flutl_recorder_fixed_from_double (X0) - dst_x * 256;
      box[1] = flutl_recorder_fixed_from_double (Y0) - dst_y * 256;
      box[2] = box[0] + tw * 256;
      box[3] = box[1] + th * 256;
    } else {
      dst_x = (int) X0;
      dst_y = (int) Y0;
      distx = disty = 0;
    }
  }
  GST_LOG_OBJECT (render, "blurred outline of region '%s': %dx%d surface%s%s, entry at "
      "(%d, %d), weights %u/%u", r->id, tw, th, rotated ? ", quarter turn" : "",
      fractional ? ", fractional position" : "", dst_x, dst_y, distx, disty);
  if (priv->dump_only) {      /* record-only test facility: the frame is left blank anyway */
    GST_WARNING_OBJECT (render, "blurred outline needs the device; not drawn");
    return;
  }
  if (!ensure_ctx (render)) {
    priv->skipped++;
    return;
  }
  pipeline_drain (render);      /* this path uses the device from the streaming thread */

  /* ---- record the temporary surface --------------------------------------- */
  flutl_recorder_begin_frame (sub, tw, th);
  flutl_recorder_region_begin (sub);
  {
    cairo_t *scr = flutl_recorder_cairo (sub);
    cairo_translate (scr, br, br);
    flutl_recorder_set_color (sub, COMP (c, 24), COMP (c, 16), COMP (c, 8), COMP (c, 0));
    cairo_set_line_width (scr, r->outline.length[0].f * 2);
    walk_layout (render, sub, layout, FALSE, r->pex, TRUE, r->writing_mode);
    flutl_recorder_stroke (sub);
  }
  flutl_recorder_region_end (sub);
  rg = flutl_recorder_regions (sub, &n_rg);
  pr = flutl_recorder_prims (sub, &n_pr);

  /* ---- blurred image: cached by what determines it ------------------------- */
  sum = g_checksum_new (G_CHECKSUM_SHA1);
  g_checksum_update (sum, (const guchar *) pr, n_pr * sizeof (flutl_prim));
  g_checksum_update (sum, (const guchar *) &tw, sizeof tw);
  g_checksum_update (sum, (const guchar *) &th, sizeof th);
  g_checksum_update (sum, (const guchar *) &radius, sizeof radius);
  g_checksum_update (sum, (const guchar *) &sigma, sizeof sigma);
  g_checksum_update (sum, (const guchar *) &distx, sizeof distx);
  g_checksum_update (sum, (const guchar *) &disty, sizeof disty);
  g_checksum_update (sum, (const guchar *) box, sizeof box);
  g_checksum_update (sum, (const guchar *) &rotated, sizeof rotated);
  key = g_strdup (g_checksum_get_string (sum));
  g_checksum_free (sum);
  if (!priv->blurred)
    priv->blurred = g_hash_table_new_full (g_str_hash, g_str_equal, g_free, g_free);
  img = g_hash_table_lookup (priv->blurred, key);
  if (img) {
    g_free (key);
  } else {
    guint32 dev_off = 0, host_off;
    const int gw = tw + (fractional ? 1 : 0), gh = th + (fractional ? 1 : 0);
    const int ew = rotated ? gh : gw, eh = rotated ? gw : gh;     /* (never both) */
    guint8 *copy = g_malloc ((gsize) ew * eh * 4);
    int rc;
    if (!flush_atlas (render)) {          /* the sub list's masks must be in HBM */
      g_free (copy);
      g_free (key);
      priv->skipped++;
      return;
    }
    rc = flutl_atlas_put_blurred (priv->ctx, tw, th, rg, n_rg, pr, n_pr, radius, sigma,
        distx, disty, fractional ? box : NULL, rotated, copy, &dev_off);
    if (rc != FLUTL_OK) {
      GST_ERROR_OBJECT (render, "flutl_atlas_put_blurred: %s", flutl_strerror (rc));
      g_free (copy);
      g_free (key);
      priv->skipped++;
      return;
    }
    host_off = flutl_recorder_atlas_append_device_entry (rec, copy, (gsize) ew * eh * 4);
    g_free (copy);
    if (host_off != dev_off) {
      GST_ERROR_OBJECT (render, "atlas out of sync (%u vs %u)", host_off, dev_off);
      g_free (key);
      priv->skipped++;
      return;
    }
    img = g_new0 (BlurredImage, 1);
    img->offset = dev_off;
    img->w = ew;
    img->h = eh;
    g_hash_table_insert (priv->blurred, key, img);
  }
  /* reference :1219-1224: blit at (rect.x - br, rect.y - br) under the CTM */
  flutl_recorder_paint_atlas_image (rec, img->offset, img->w, img->h, dst_x, dst_y);
}

/* Text outline (reference :1180-1232). */
static void
outline_layout (GstTTMLRender *render, HipRegion *r, PangoLayout *layout)
{
  FlutlRecorder *rec = get_priv (render)->rec;
  cairo_t *cr = flutl_recorder_cairo (rec);
  /* reference :1186: "current colour" is hard-wired to opaque white */
  guint32 c = r->outline.use_current_color ? 0xFFFFFFFF : r->outline.color;
  double keep[4];

  if (r->outline.length[1].unit != GST_TTML_LENGTH_UNIT_NOT_PRESENT) {
    GstTTMLRenderPrivate *priv = get_priv (render);
    g_mutex_lock (&priv->ctx_lock);     /* (the video side may be on the device) */
    blurred_outline (render, r, layout, c);
    g_mutex_unlock (&priv->ctx_lock);
    return;
  }
  flutl_recorder_get_color (rec, keep);
  cairo_save (cr);
  flutl_recorder_set_color (rec, COMP (c, 24), COMP (c, 16), COMP (c, 8), COMP (c, 0));
  cairo_set_line_width (cr, r->outline.length[0].f * 2);
  walk_layout (render, rec, layout, FALSE, r->pex, TRUE, r->writing_mode);
  flutl_recorder_stroke (rec);
  cairo_restore (cr);
  flutl_recorder_set_color (rec, keep[0], keep[1], keep[2], keep[3]);
}

/* One region -> one binning group (reference :1235-1385). */
static void
record_region_body (HipRegion *r, GstTTMLRender *render)
{
  FlutlRecorder *rec = get_priv (render)->rec;
  cairo_t *cr = flutl_recorder_cairo (rec);
  gboolean grouped = r->opacity < 1.0;
  GList *l;

  flutl_recorder_region_begin (rec);
  cairo_save (cr);
  if (grouped)
    flutl_recorder_group_begin (rec, r->ox, r->oy, r->ex, r->ey);

  if (r->bg != 0) {
    flutl_recorder_set_color (rec, COMP (r->bg, 24), COMP (r->bg, 16),
        COMP (r->bg, 8), COMP (r->bg, 0));
    flutl_recorder_fill_rect (rec, r->ox, r->oy, r->ex, r->ey);
  }
  if (r->image)
    flutl_recorder_paint_image (rec, r->image, r->image_x, r->image_y);
  if (!r->overflow_visible)
    flutl_recorder_clip (rec, r->pox, r->poy, r->pex, r->pey);

  cairo_translate (cr, r->pox, r->poy);
  if (r->display_align != GST_TTML_DISPLAY_ALIGN_BEFORE) {
    double height = 0.0, posy;
    for (l = r->layouts; l; l = l->next) {
      PangoRectangle ext;
      pango_layout_get_pixel_extents (l->data, NULL, &ext);
      height += ext.height;
    }
    /* reference :1303: the UNPADDED extent is used */
    posy = r->ey - height;
    if (r->display_align == GST_TTML_DISPLAY_ALIGN_CENTER)
      posy /= 2.0;
    cairo_translate (cr, 0, posy);
  }

  cairo_save (cr);
  if (r->writing_mode == GST_TTML_WRITING_MODE_TBRL ||
      r->writing_mode == GST_TTML_WRITING_MODE_TBLR) {
    /* reference :1313-1340: the CTM is REPLACED by a quarter turn, and the
     * shared Pango context is switched to east gravity -- for good: the
     * reference never switches it back, so later layouts inherit it */
    cairo_matrix_t om;
    cairo_matrix_init_rotate (&om, G_PI / 2);
    cairo_matrix_translate (&om, r->pox,
        r->writing_mode == GST_TTML_WRITING_MODE_TBRL ? -r->pex : -r->poy);
    cairo_set_matrix (cr, &om);
    pango_context_set_base_gravity (render->pango_context, PANGO_GRAVITY_EAST);
    pango_context_set_gravity_hint (render->pango_context, PANGO_GRAVITY_HINT_STRONG);
    pango_cairo_update_context (cr, render->pango_context);
    if (grouped) {
      /* reference: with an opacity the region is drawn on a temporary surface
       * under cairo_translate (-origin) (:1250-1255) -- which this
       * cairo_set_matrix discards as well: the text lands on that surface at
       * its FRAME coordinates, i.e. moved by the region's origin once more.
       * The CTM stays the reference's; the recorder adds the origin to what
       * it has rounded, as the paint of the temporary surface does. */
      flutl_recorder_set_device_shift (rec, (int) r->ox, (int) r->oy);
    }
  }
  {
    for (l = r->layouts; l; l = l->next) {
      gboolean outlined =
          r->outline.length[0].unit != GST_TTML_LENGTH_UNIT_NOT_PRESENT;
      if (outlined)
        outline_layout (render, r, l->data);
      /* reference :1360-1363: default text colour = inverse of the region
       * background, opaque */
      flutl_recorder_set_color (rec, 1.0 - COMP (r->bg, 24), 1.0 - COMP (r->bg, 16),
          1.0 - COMP (r->bg, 8), 1.0);
      walk_layout (render, rec, l->data, TRUE, r->pex, !outlined, r->writing_mode);
    }
  }
  cairo_restore (cr);

  if (grouped)
    flutl_recorder_group_end (rec, r->opacity);
  flutl_recorder_clip_reset (rec);
  cairo_restore (cr);
  flutl_recorder_region_end (rec);
}


/* ---- region cache (display lists across buffers) --------------------
 * The reference re-draws every region for every buffer (gstttmlrender.c:1462-1472),
 * although consecutive scene states mostly differ in one attribute of one region.
 * What a region records is a function of: its geometry and styling, its layouts
 * (shared objects of the layout cache: the same pointer is the same text, fonts,
 * colours and wrapping), the frame size and whether the frame was still "clear"
 * for Cairo when the region began.  Regions are kept under that key with the
 * primitives they recorded, and played back.  Per document and atlas generation
 * (primitives name atlas offsets). */
typedef struct
{
  gint geo[8];
  guint32 bg;
  gdouble opacity;
  gpointer image;
  gint image_xy[2];
  gint display_align, overflow_visible, writing_mode;
  guint32 outline_color;
  gint outline_current;
  gfloat outline_len[2];
  gint outline_unit[2];
  gint frame_w, frame_h, clear_in;
  guint n_layouts;
} RegionKey;

typedef struct
{
  RegionKey key;
  PangoLayout **layouts;        /* referenced */
  guint64 hash;
  flutl_prim *prims;
  guint32 n_prims;
  gboolean has_region;
  flutl_region region;
  gint clear_out;
  guint32 unsupported;
  /* for playing it back in other colours: per primitive the colour attribute it
   * was drawn under (1-based into sources, 0 none), and the key without colours */
  guint16 *tags;
  ColorSource *sources;
  guint n_sources;
  guint64 shape_hash;
  gboolean recolourable;
} RegionEntry;

static void
region_entry_free (gpointer data)
{
  RegionEntry *e = data;
  for (guint i = 0; i < e->key.n_layouts; i++)
    g_object_unref (e->layouts[i]);
  g_free (e->layouts);
  g_free (e->prims);
  g_free (e->tags);
  g_free (e->sources);
  g_free (e);
}

static void
region_key_fill (RegionKey *k, const HipRegion *r, FlutlRecorder *rec, gint w, gint h)
{
  memset (k, 0, sizeof *k);      /* (padding too: keys are compared as bytes) */
  k->geo[0] = r->ox; k->geo[1] = r->oy; k->geo[2] = r->ex; k->geo[3] = r->ey;
  k->geo[4] = r->pox; k->geo[5] = r->poy; k->geo[6] = r->pex; k->geo[7] = r->pey;
  k->bg = r->bg;
  k->opacity = r->opacity;
  k->image = r->image;
  k->image_xy[0] = r->image_x;
  k->image_xy[1] = r->image_y;
  k->display_align = r->display_align;
  k->overflow_visible = r->overflow_visible;
  k->writing_mode = r->writing_mode;
  k->outline_color = r->outline.color;
  k->outline_current = r->outline.use_current_color;
  for (int i = 0; i < 2; i++) {
    k->outline_unit[i] = r->outline.length[i].unit;
    k->outline_len[i] = r->outline.length[i].unit == GST_TTML_LENGTH_UNIT_NOT_PRESENT
        ? 0.f : r->outline.length[i].f;
  }
  k->frame_w = w;
  k->frame_h = h;
  k->clear_in = flutl_recorder_frame_is_clear (rec);
  k->n_layouts = g_list_length (r->layouts);
}

static void
region_replay (GstTTMLRender *render, const HipRegion *r, const RegionEntry *e,
    const flutl_prim *prims)
{
  FlutlRecorder *rec = get_priv (render)->rec;
  if (r->writing_mode == GST_TTML_WRITING_MODE_TBRL ||
      r->writing_mode == GST_TTML_WRITING_MODE_TBLR) {
    pango_context_set_base_gravity (render->pango_context, PANGO_GRAVITY_EAST);
    pango_context_set_gravity_hint (render->pango_context, PANGO_GRAVITY_HINT_STRONG);
    pango_cairo_update_context (flutl_recorder_cairo (rec), render->pango_context);
  }
  flutl_recorder_replay (rec, prims, e->n_prims, e->has_region ? &e->region : NULL,
      e->clear_out, e->unsupported);
}

static RegionEntry *
region_entry_new (GstTTMLRenderPrivate *priv, const HipRegion *r, const RegionKey *key,
    guint64 hash, guint64 shape_hash, const flutl_prim *prims, guint32 n_prims,
    const flutl_region *region, gint clear_out, guint32 unsupported)
{
  RegionEntry *e;
  GList *l;
  guint i;
  if (priv->region_cache->len >= 256)           /* bound it: the least recently used go */
    g_ptr_array_remove_range (priv->region_cache, 0, 64);
  e = g_new0 (RegionEntry, 1);
  e->key = *key;
  e->hash = hash;
  e->shape_hash = shape_hash;
  e->layouts = g_new (PangoLayout *, key->n_layouts + 1);
  for (l = r->layouts, i = 0; l; l = l->next)
    e->layouts[i++] = g_object_ref (l->data);
  e->prims = g_memdup (prims, n_prims * sizeof *prims);
  e->n_prims = n_prims;
  e->has_region = region != NULL;
  if (region)
    e->region = *region;
  e->clear_out = clear_out;
  e->unsupported = unsupported;
  g_ptr_array_add (priv->region_cache, e);
  return e;
}

/* r is e's region but for colours of its layouts?  Then e's primitives in r's
 * colours are what recording r would give: a colour reaches a primitive as its
 * colour word and through two flags, opaque and clear (flutl_recorder_set_color),
 * which follow from the alpha and choose the compositing rule -- so only colours
 * of the same class (opaque, translucent, clear) are exchanged.
 * The colours are matched by value: two attributes of one colour may have fed
 * one glyph run, so they must also end up in one colour. */
static gboolean
region_recolour (GstTTMLRender *render, HipRegion *r, RegionEntry *e, const RegionKey *key,
    guint64 hash, guint64 shape_hash)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  ColorSource to[64];
  gboolean mapped[64] = { FALSE };
  flutl_prim *prims;
  RegionEntry *n;
  GList *l;
  guint k = 0, i, j;
  /* (e itself may go when the new entry is made: the oldest entries of a full cache do) */
  const guint32 n_prims = e->n_prims, unsupported = e->unsupported;
  const guint n_sources = e->n_sources;
  const gboolean has_region = e->has_region;
  const flutl_region region = e->region;
  const gint clear_out = e->clear_out;
  guint16 *tags;
  ColorSource *sources;

  if (n_sources > G_N_ELEMENTS (to))
    return FALSE;
  for (l = r->layouts; l; l = l->next, k++) {
    const LayoutShape *so = g_object_get_qdata (G_OBJECT (e->layouts[k]), layout_shape_quark ());
    const LayoutShape *sn = g_object_get_qdata (l->data, layout_shape_quark ());
    if (!so || !sn)
      return FALSE;
    if (so != sn && (so->hash != sn->hash || so->len != sn->len ||
            so->n_colors != sn->n_colors || memcmp (so->key, sn->key, so->len) != 0))
      return FALSE;
    /* Pango merges adjoining attributes of one value (pango_attr_list_change), and
     * which of two attributes over one range wins then depends on what was merged:
     * the colours may change, not which of them are alike -- among the attributes
     * and the markup's colours both (fuzz seed 510070: a span whose <set> ended took
     * its paragraph's colour, its attribute went into the paragraph's, and the span
     * of the markup showed through). */
    if (so != sn)
      for (i = 0; i < so->n_colors; i++)
        for (j = i + 1; j < so->n_colors; j++)
          if ((memcmp (&so->colors[i], &so->colors[j], sizeof (ColorSource)) == 0) !=
              (memcmp (&sn->colors[i], &sn->colors[j], sizeof (ColorSource)) == 0))
            return FALSE;
    for (i = 0; i < so->n_colors; i++) {
      const ColorSource *co = &so->colors[i], *cn = &sn->colors[i];
      /* (the alpha, k / 255: Cairo's "opaque" is k == 255, "clear" k == 0.  Whether
       * or not anything was drawn in it: a clear colour draws nothing and drops
       * the pending path, and k17's regions found that out) */
      if (((co->r & 0xFF) == 0xFF) != ((cn->r & 0xFF) == 0xFF) ||
          ((co->r & 0xFF) == 0) != ((cn->r & 0xFF) == 0))
        return FALSE;           /* opaque / translucent / clear: another rule */
      /* (every source of that value: an earlier exchange may have made two alike;
       * none: nothing was drawn in it -- a colour of the markup under an attribute,
       * a clear one, the run of a line break) */
      for (j = 0; j < n_sources; j++) {
        if (memcmp (&e->sources[j], co, sizeof *co) != 0)
          continue;
        if (mapped[j] && memcmp (&to[j], cn, sizeof *cn) != 0)
          return FALSE;         /* one colour then, two now */
        mapped[j] = TRUE;
        to[j] = *cn;
      }
    }
  }
  prims = g_memdup (e->prims, n_prims * sizeof *prims);
  tags = g_memdup (e->tags, n_prims * sizeof *tags);
  sources = g_memdup (e->sources, n_sources * sizeof *sources);
  for (j = 0; j < n_sources; j++)
    if (mapped[j])
      sources[j] = to[j];
  for (i = 0; i < n_prims; i++)
    if (tags[i] && mapped[tags[i] - 1])
      prims[i].color = smuggled_pixel (&to[tags[i] - 1]);
  region_replay (render, r, e, prims);
  /* ... and it is a region of its own from now on */
  n = region_entry_new (priv, r, key, hash, shape_hash, prims, n_prims,
      has_region ? &region : NULL, clear_out, unsupported);
  n->recolourable = TRUE;
  n->tags = tags;
  n->sources = sources;
  n->n_sources = n_sources;
  g_free (prims);
  return TRUE;
}

static void
record_region (HipRegion *r, GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  FlutlRecorder *rec = priv->rec;
  RegionKey key;
  RegionEntry *e;
  guint64 hash = 0xcbf29ce484222325ull, shape_hash;
  gboolean all_shapes = TRUE;
  guint32 prim_mark, region_mark, n_prims, unsupported_before;
  const flutl_region *region;
  const flutl_prim *prims;
  int has_placeholder;
  guint skipped_before, i;
  GList *l;

  if (r->par) {
    gint64 t0 = priv->timing ? g_get_monotonic_time () : 0;
    region_flush_paragraph (render, r);
    if (priv->timing && priv->n_timed)
      priv->t_layout += g_get_monotonic_time () - t0;
  }
  if (priv->no_region_cache) {
    record_region_body (r, render);
    return;
  }
  region_key_fill (&key, r, rec, render->base.state.frame_width,
      render->base.state.frame_height);
  for (i = 0; i < sizeof key / sizeof (guint32); i++)
    hash = (hash ^ ((const guint32 *) &key)[i]) * 0x100000001b3ull;
  shape_hash = hash;
  for (l = r->layouts; l; l = l->next) {
    const LayoutShape *sh = g_object_get_qdata (l->data, layout_shape_quark ());
    hash = (hash ^ (guint64) (guintptr) l->data) * 0x100000001b3ull;
    shape_hash = (shape_hash ^ (sh ? sh->hash : (guint64) (guintptr) l->data)) * 0x100000001b3ull;
    all_shapes = all_shapes && sh != NULL;
  }
  if (!priv->region_cache)
    priv->region_cache = g_ptr_array_new_with_free_func (region_entry_free);
  for (i = 0; i < priv->region_cache->len; i++) {
    e = g_ptr_array_index (priv->region_cache, i);
    if (e->hash != hash || memcmp (&e->key, &key, sizeof key) != 0)
      continue;
    guint k = 0;
    for (l = r->layouts; l && e->layouts[k] == l->data; l = l->next)
      k++;
    if (l)
      continue;
    /* the same region as before: its primitives, and what recording it does to
     * the shared Pango context (the reference switches it to east gravity for the
     * vertical writing modes and never back, :1313-1340) */
    region_replay (render, r, e, e->prims);
    priv->region_hits++;
    /* (most recently used last: what goes when the cache is full is the oldest) */
    if (i + 1 < priv->region_cache->len) {
      gpointer *pd = priv->region_cache->pdata;
      memmove (pd + i, pd + i + 1, (priv->region_cache->len - 1 - i) * sizeof *pd);
      pd[priv->region_cache->len - 1] = e;
    }
    return;
  }
  /* the same region in other colours (a <set> on tts:color / tts:backgroundColor
   * of a span -- the one thing most scene states of a subtitle differ in): the
   * primitives of the earlier one with other colour words */
  if (all_shapes && !priv->no_recolour)
    for (i = 0; i < priv->region_cache->len; i++) {
      e = g_ptr_array_index (priv->region_cache, i);
      if (!e->recolourable || e->shape_hash != shape_hash || memcmp (&e->key, &key, sizeof key) != 0)
        continue;
      if (region_recolour (render, r, e, &key, hash, shape_hash)) {
        priv->region_recoloured++;
        return;
      }
    }
  priv->region_misses++;
  flutl_recorder_mark (rec, &prim_mark, &region_mark);
  unsupported_before = flutl_recorder_unsupported (rec);
  skipped_before = priv->skipped;
  if (priv->color_sources)
    g_array_set_size (priv->color_sources, 0);
  {
    gint64 t0 = priv->timing ? g_get_monotonic_time () : 0;
    record_region_body (r, render);
    if (priv->timing && priv->n_timed)
      priv->t_miss += g_get_monotonic_time () - t0;
  }
  prims = flutl_recorder_snapshot (rec, prim_mark, region_mark, &n_prims, &region,
      &has_placeholder);
  if (has_placeholder || priv->skipped != skipped_before)
    return;                     /* not final yet (a mask is still being made), or failed */
  e = region_entry_new (priv, r, &key, hash, shape_hash, prims, n_prims, region,
      flutl_recorder_frame_is_clear (rec), flutl_recorder_unsupported (rec) - unsupported_before);
  /* (blurred outlines are images made of the text in its colours) */
  e->recolourable = all_shapes &&
      r->outline.length[1].unit == GST_TTML_LENGTH_UNIT_NOT_PRESENT;
  e->tags = g_memdup (flutl_recorder_prim_tags (rec) + prim_mark, n_prims * sizeof (guint16));
  for (i = 0; i < n_prims; i++)
    if (e->tags[i] == 0xFFFF || (priv->color_sources && e->tags[i] > priv->color_sources->len))
      e->recolourable = FALSE;  /* (one glyph run in one colour from two attributes) */
  if (priv->color_sources && priv->color_sources->len) {
    e->n_sources = priv->color_sources->len;
    e->sources = g_memdup (priv->color_sources->data, e->n_sources * sizeof (ColorSource));
    /* (a colour that was set but drew nothing -- the run of a line break -- is no
     * source: type 0 matches no attribute) */
    for (guint j = 0; j < e->n_sources; j++) {
      for (i = 0; i < n_prims && e->tags[i] != j + 1; i++)
        ;
      if (i == n_prims)
        e->sources[j].type = 0;
    }
  }
}

/* ---- device side -------------------------------------------------------------- */
/* (flutl_recorder_set_edge_blocks: user = the context) */
static void *
edge_block_acquire (void *user, size_t bytes)
{
  void *blk = NULL;
  return flutl_ctx_edge_block_acquire (user, bytes, &blk) == FLUTL_OK ? blk : NULL;
}

static void
edge_block_release (void *user, void *block)
{
  flutl_ctx_edge_block_release (user, block);
}

/* The HIP device of an instance.  FLUTL_DEVICE=<n>: that one, for every instance of the
 * process (what a launcher with one process per GPU sets).  Else the instances of a process
 * take the node's devices in turn -- the k-th ttmlrender to make its context gets device
 * k mod flutl_device_count (), or the k-th entry of FLUTL_DEVICE_MAP="0,2,4,6" -- so that
 * several streams in one pipeline process spread over the node's GPUs instead of sharing the
 * first (streams are independent: SURVEY 8e, one per GPU). */
static int
pick_device (void)
{
  static gint next_instance = 0;
  const gchar *dev = g_getenv ("FLUTL_DEVICE");
  const gchar *map = g_getenv ("FLUTL_DEVICE_MAP");
  if (dev && *dev)
    return atoi (dev);
  const int k = g_atomic_int_add (&next_instance, 1);
  if (map && *map) {
    gchar **v = g_strsplit (map, ",", -1);
    const guint n = g_strv_length (v);
    const int d = n ? atoi (v[k % n]) : 0;
    g_strfreev (v);
    return d;
  }
  const int n = flutl_device_count ();
  return n > 0 ? k % n : 0;
}

static gboolean
ensure_ctx (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  if (priv->ctx)
    return TRUE;
  if (priv->ctx_failed)
    return FALSE;
  priv->device = pick_device ();
  int rc = flutl_ctx_create (priv->device, &priv->ctx);
  if (rc != FLUTL_OK) {
    priv->ctx_failed = TRUE;
    GST_ELEMENT_ERROR (render, LIBRARY, INIT,
        ("ttmlrender: no usable MI355X/HIP device"),
        ("flutl_ctx_create: %s", flutl_strerror (rc)));
    return FALSE;
  }
  return TRUE;
}

/* Upload the atlas entries recorded since the last call. */
static gboolean
flush_atlas (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  const uint32_t *off_, *size_;
  FlutlPolygonEntry *const *poly_;
  FlutlPolygonEntry **poly;
  uint32_t *off, *size;
  uint32_t i, n;
  gboolean ok = TRUE;
  const gint64 t_flush0 = g_get_monotonic_time ();
  /* (the streaming thread may be adding glyph masks: entries are uploaded in
   * the order they were given their offsets, whichever frame they belong to.
   * The recorder is locked entry by entry -- its atlas may move when it grows --
   * not for the whole upload: the first frame of a cue brings megabytes of
   * outline masks, and the streaming thread recording the next buffer needs the
   * recorder too.  Calls do not overlap: the finisher makes them, or the
   * streaming thread when the finisher has nothing in flight.) */
  flutl_recorder_lock (priv->rec);
  n = flutl_recorder_new_atlas_entries (priv->rec, &off_, &size_, &poly_);
  off = g_memdup (off_, n * sizeof *off);
  size = g_memdup (size_, n * sizeof *size);
  poly = g_memdup (poly_, n * sizeof *poly);
  flutl_recorder_unlock (priv->rec);
  for (i = 0; i < n && ok; i++) {
    uint32_t dev_off = 0;
    size_t atlas_size;
    int rc;
    if (poly[i]) {
      /* an outline as the polygon a worker's stroker made of it: the device scan
       * converts it straight into the entry (no coverage in host memory, no upload) */
      poly[i]->submitted = 1;   /* (a block of the context's arena is its own again, whatever comes back) */
      rc = flutl_atlas_put_polygon (priv->ctx, poly[i]->edges, poly[i]->n_edges, poly[i]->box,
          0, &dev_off);
      priv->n_polygons++;
      priv->polygon_edges += poly[i]->n_edges;
    } else {
      flutl_recorder_lock (priv->rec);
      rc = flutl_atlas_put (priv->ctx, flutl_recorder_atlas (priv->rec, &atlas_size) + off[i],
          size[i], &dev_off);
      flutl_recorder_unlock (priv->rec);
    }
    if (rc != FLUTL_OK || dev_off != off[i]) {
      GST_ELEMENT_ERROR (render, LIBRARY, FAILED, ("ttmlrender: atlas upload failed"),
          ("flutl_atlas_put: %s (offset %u vs %u)", flutl_strerror (rc), dev_off, off[i]));
      ok = FALSE;
    }
  }
  if (priv->timing_trace && n) {
    guint64 bytes = 0;
    for (i = 0; i < n; i++)
      bytes += size[i];
    g_printerr ("ttmlrender-hip atlas flush: %u entries, %" G_GUINT64_FORMAT " bytes, %" G_GINT64_FORMAT
        " us\n", n, bytes, g_get_monotonic_time () - t_flush0);
  }
  for (i = 0; i < n; i++)
    flutl_polygon_entry_free (poly[i]);
  g_free (poly);
  g_free (off);
  g_free (size);
  return ok;
}

/* ---- the finisher thread ------------------------------------------------------- */
typedef struct
{
  gint width, height;
  flutl_region *regions;
  flutl_prim *prims;
  FlutlPatch *patches;
  guint32 n_rg, n_pr, n_patches;
  GstMemory *mem;              /* keeps the frame out of the pool until it is queued */
  FrameSlot *slot;
  Ticket *ticket;
} FrameJob;

static gpointer
finisher_main (gpointer data)
{
  GstTTMLRender *render = data;
  GstTTMLRenderPrivate *priv = get_priv (render);
  FrameJob *job;
  for (;;) {
    gint64 t0;
    int rc;
    /* A stream of buffers a few hundred microseconds apart: poll for a while
     * before sleeping.  A sleeping consumer makes every push a futex wake on the
     * streaming thread (and wakes up late itself); polling, it costs that thread
     * an uncontended lock. */
    job = g_async_queue_try_pop (priv->jobs);
    if (!job) {
      const gint64 t_idle = g_get_monotonic_time ();
      while (!(job = g_async_queue_try_pop (priv->jobs)) &&
          g_get_monotonic_time () - t_idle < 400)
        ;                         /* (g_get_monotonic_time is a vDSO call: ~20 ns) */
      if (!job)
        job = g_async_queue_pop (priv->jobs);
    }
    if (job->slot == NULL) {   /* the stop sentinel */
      g_free (job);
      break;
    }
    t0 = priv->timing ? g_get_monotonic_time () : 0;
    flutl_recorder_resolve (priv->rec, job->regions, job->n_rg, job->prims, job->n_pr,
        job->patches, job->n_patches);
    if (priv->timing) {
      gint64 t1 = g_get_monotonic_time ();
      priv->t_resolve += t1 - t0;
      t0 = t1;
    }
    /* the list is staged first: it goes up to the device with the conversion of the frame's
     * first outline (flush_atlas), not as a copy operation of its own */
    rc = flutl_frame_stage (priv->ctx, job->slot->frame, job->regions, job->n_rg, job->prims,
        job->n_pr);
    if (rc == FLUTL_OK)
      rc = flush_atlas (render) ? FLUTL_OK : FLUTL_ENODEV;
    if (priv->timing) {
      gint64 t1 = g_get_monotonic_time ();
      priv->t_flush += t1 - t0;
      t0 = t1;
    }
    if (rc == FLUTL_OK)
      rc = flutl_frame_submit_staged (priv->ctx, job->slot->frame);
    if (priv->timing)
      priv->t_submit += g_get_monotonic_time () - t0;
    if (rc != FLUTL_OK) {
      job->slot->stale = TRUE;
      GST_ELEMENT_ERROR (render, LIBRARY, FAILED, ("ttmlrender: HIP render failed"),
          ("%s", flutl_strerror (rc)));
    }
    ticket_complete (job->ticket, rc);
    ticket_unref (job->ticket);
    gst_memory_unref (job->mem);
    const gboolean waited = job->n_patches > 0;
    g_free (job->regions);
    g_free (job->prims);
    g_free (job->patches);
    g_free (job);
    g_mutex_lock (&priv->flight_lock);
    priv->in_flight--;
    if (waited)
      priv->waiting_in_flight--;
    g_cond_broadcast (&priv->flight_cond);
    g_mutex_unlock (&priv->flight_lock);
  }
  return NULL;
}

/* Wait until every buffer handed out so far has been queued on the device: the
 * finisher is idle then and the calling (streaming) thread may use `ctx`. */
static void
pipeline_drain (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  if (!priv->finisher)
    return;
  g_mutex_lock (&priv->flight_lock);
  while (priv->in_flight > 0)
    g_cond_wait (&priv->flight_cond, &priv->flight_lock);
  g_mutex_unlock (&priv->flight_lock);
}

static void
pipeline_stop (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  if (!priv->finisher)
    return;
  pipeline_drain (render);
  g_async_queue_push (priv->jobs, g_new0 (FrameJob, 1));    /* slot == NULL: stop */
  g_thread_join (priv->finisher);
  priv->finisher = NULL;
}

/* Buffers that may be ahead of the device.  The pipeline is bound by latency per
 * slot -- an outline takes ~2.7 ms on its worker, plus thread wake-ups -- so the
 * depth sets the rate on a stream with new outlined text per cue; measured at 4K on
 * the GPU box's host, frames made ahead: 4 -> 1.3 k frames/s, 8 -> 1.6-1.7 k, 12 ->
 * 1.8 k, 16 -> 2.1 k.  16, or fewer where that many page-locked frames would exceed
 * ~640 MB (4 at 8K). */
static guint
run_ahead (GstTTMLRenderPrivate *priv, gint width, gint height)
{
  if (priv->max_ahead)
    return priv->max_ahead;
  const guint by_size = (guint) CLAMP (POOL_BUDGET_BYTES / MAX ((gsize) width * height * 4, 1), 3,
      MAX_IN_FLIGHT);
  /* Round 3: an outline is a polygon after 0.3-0.45 ms on its worker and coverage 30 us
   * later, not 3 ms of cairo_stroke: 6 buffers ahead cover that at 11 k buffers/s
   * (tools/ahead_sweep.py: a long stream 10.5-11.8 k at 6, 11.1-11.5 k at 8, 11.4-11.7 k
   * at 16), and every pool frame never used before costs its first write-back, a whole
   * frame over PCIe -- a 120-cue stream ran at 10 k with 6 ahead and at 6 k with 16.
   * (Outlines by Cairo on the workers, FLUTL_HOST_STROKES: the old depth.) */
  if (priv->host_strokes)
    return by_size;
  /* ... as deep as the pool has frames for.  Round 4, with the device at 66 us per cue: a
   * 3 000-cue stream runs 15.1 k frames/s at 16 ahead against 12.3 k at 12 and 11.7 k with the
   * depth growing by a frame every 64 buffers as in round 3 -- a frame is 14 ms of page-locking
   * on whoever makes it, and made while the stream runs (on this thread, or on one of their
   * own: tried) it slows every thread of the process down; so all of them are made when the
   * stream's first outline shows up (gen_buffer: prefill), cleared by the CPU so that their
   * clearing is not 1.3 ms of device time each in front of the first cues. */
  guint have = 0;
  g_mutex_lock (&priv->pool->lock);
  if (priv->pool->prefilled_size == (gsize) width * height * 4)
    have = priv->pool->prefilled_count;
  g_mutex_unlock (&priv->pool->lock);
  return MIN (by_size, CLAMP (have > 2 ? have - 2 : 0, DEVICE_STROKES_AHEAD, DEVICE_STROKES_AHEAD_MAX));
}

/* Hand a recorded frame to the finisher.  The lists are copied: the recorder's
 * arrays are reused by the next buffer. */
static void
pipeline_push (GstTTMLRender *render, gint width, gint height, const flutl_region *rg,
    guint32 n_rg, const flutl_prim *pr, guint32 n_pr, FrameSlot *slot, GstBuffer *buffer,
    Ticket *ticket)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  FrameJob *job = g_new0 (FrameJob, 1);
  uint32_t n_patches = 0;
  const FlutlPatch *patches = flutl_recorder_patches (priv->rec, &n_patches);
  job->width = width;
  job->height = height;
  job->n_rg = n_rg;
  job->n_pr = n_pr;
  job->n_patches = n_patches;
  job->regions = g_memdup (rg, n_rg * sizeof *rg);
  job->prims = g_memdup (pr, n_pr * sizeof *pr);
  job->patches = g_memdup (patches, n_patches * sizeof *patches);
  job->mem = gst_memory_ref (gst_buffer_peek_memory (buffer, 0));
  job->slot = slot;
  job->ticket = ticket_ref (ticket);
  if (!priv->finisher)
    priv->finisher = g_thread_new ("ttmlrender-finish", finisher_main, render);
  /* (as far ahead as the pool goes for frames that wait for outlines: the point is
   * to get to the next outline early.  A complete frame behind them gains nothing
   * from depth, and every further pool frame in use is one more whose first
   * write-back is a whole frame over PCIe, a millisecond at 4K: 3 of those) */
  const guint ahead = n_patches ? run_ahead (priv, width, height)
                                : MIN (run_ahead (priv, width, height), 3);
  g_mutex_lock (&priv->flight_lock);
  if (priv->in_flight >= ahead) {
    const gint64 t_b = g_get_monotonic_time ();
    while (priv->in_flight >= ahead)
      g_cond_wait (&priv->flight_cond, &priv->flight_lock);
    priv->t_push_block += g_get_monotonic_time () - t_b;
  }
  priv->in_flight++;
  priv->n_pushed++;
  if (n_patches)
    priv->waiting_in_flight++;
  g_mutex_unlock (&priv->flight_lock);
  g_async_queue_push (priv->jobs, job);
}

/* a timeline interval recorded for the video side (fused over-video, below) */
typedef struct
{
  GstClockTime start, end;
  flutl_region *regions;
  flutl_prim *prims;
  guint32 n_rg, n_pr;
} VideoCue;

static void
video_cue_free (VideoCue *c)
{
  g_free (c->regions);
  g_free (c->prims);
  g_free (c);
}

static GstBuffer *
gst_ttmlrender_gen_buffer (GstTTMLBase *base, GstClockTime ts, GstClockTime duration)
{
  GstTTMLRender *render = GST_TTMLRENDER (base);
  GstTTMLRenderPrivate *priv = get_priv (render);
  FlutlRecorder *rec = priv->rec;
  gint width = base->state.frame_width, height = base->state.frame_height;
  GList *regions = NULL, *l;
  const flutl_region *rg;
  const flutl_prim *pr;
  uint32_t n_rg, n_pr;
  GstBuffer *buffer = NULL;
  gint64 t_start = 0, t_recorded = 0;

  if (priv->video_pad) {
    /* fused over-video: the frame is the video frame, whose size the video caps bring */
    g_mutex_lock (&priv->video_lock);
    while (!priv->video_have_caps && !priv->video_flushing)
      g_cond_wait (&priv->video_cond, &priv->video_lock);
    width = base->state.frame_width = priv->video_width;
    height = base->state.frame_height = priv->video_height;
    g_mutex_unlock (&priv->video_lock);
  }
  GST_DEBUG_OBJECT (render, "buffer %" GST_TIME_FORMAT " + %" GST_TIME_FORMAT
      ", frame %dx%d", GST_TIME_ARGS (ts), GST_TIME_ARGS (duration), width, height);
  if (width <= 0 || height <= 0 || width > 32767 || height > 32767)
    return NULL;

  /* ---- host: scene state -> display list -------------------------------- */
  t_start = priv->timing ? g_get_monotonic_time () : 0;
  {
    /* The mask caches live as long as the document (reset vfunc).  A live
     * stream is one endless document with new text in every cue: bound the
     * atlas by starting over (masks are re-made on demand) at a frame boundary. */
    size_t atlas_bytes = 0;
    flutl_recorder_lock (rec);
    (void) flutl_recorder_atlas (rec, &atlas_bytes);
    flutl_recorder_unlock (rec);
    /* (a dump keeps one atlas; so do the cues queued for the video side: there the
     * atlas starts over once the video has used up every cue made so far) */
    if (atlas_bytes > priv->atlas_limit && !priv->dump && video_cues_drained (render)) {
      GST_INFO_OBJECT (render, "atlas at %" G_GSIZE_FORMAT " bytes: dropping the mask caches",
          atlas_bytes);
      caches_drop (render);
    }
  }
  /* outlines without a mask yet are rasterised on worker threads and waited for
   * by the finisher, unless the frame is needed complete right here */
  /* (a cue for the video side: its outlines go to the workers as well and are waited for
   * before the cue is queued, below -- they come back as polygons for the device instead of
   * being rasterised by Cairo on this thread) */
  flutl_recorder_set_defer (rec, !priv->sync_mode && !priv->full_d2h &&
      (!priv->dump || priv->dump_defer));
  /* ... as polygons for the device to scan convert (not when the frame is dumped: a
   * scene pack holds the masks' bytes) */
  flutl_recorder_set_device_strokes (rec, !priv->dump && !priv->host_strokes);
  /* ... whose edges the workers write straight into page-locked blocks of the context */
  if (priv->ctx && !priv->edge_blocks_set && !g_getenv ("FLUTL_NO_EDGE_BLOCKS")) {
    flutl_recorder_set_edge_blocks (rec, edge_block_acquire, edge_block_release, priv->ctx);
    priv->edge_blocks_set = TRUE;
  }
  flutl_recorder_begin_frame (rec, width, height);
  if (base->state.saved_region_attr_stacks) {
    gpointer ctx[2] = { render, &regions };
    g_hash_table_foreach (base->state.saved_region_attr_stacks, background_region,
        ctx);
  }
  for (l = base->active_spans; l; l = l->next)
    span_to_region (render, &regions, l->data);
  for (l = regions; l; l = l->next) {
    /* background regions no span has configured */
    HipRegion *r = l->data;
    if (r->pending_attrs) {
      GstTTMLStyle style;
      style.attributes = r->pending_attrs;
      r->pending_attrs = NULL;
      region_configure (render, r, &style);
    }
  }
  if (priv->timing && priv->n_timed)
    priv->t_build += g_get_monotonic_time () - t_start;
  for (l = regions; l; l = l->next) {
    /* a region with an opacity is drawn on a surface of its own size */
    const HipRegion *r = l->data;
    if (r->opacity < 1.0)
      flutl_recorder_reserve (rec, (int) r->ex, (int) r->ey);
  }
  for (l = regions; l; l = l->next)
    record_region (l->data, render);
  g_list_free_full (regions, region_free);

  rg = flutl_recorder_regions (rec, &n_rg);
  pr = flutl_recorder_prims (rec, &n_pr);
  t_recorded = priv->timing ? g_get_monotonic_time () : 0;
  if (flutl_recorder_unsupported (rec) || priv->skipped) {
    /* never push a frame that differs from the reference's: no buffer at all
     * (the base class's error convention, gstttmlbase.c:161-163) */
    GST_ELEMENT_ERROR (render, STREAM, NOT_IMPLEMENTED,
        ("ttmlrender: the document uses a rendering feature that is not implemented"),
        ("%u primitive(s), %u outline(s) could not be reproduced exactly; buffer dropped",
            flutl_recorder_unsupported (rec), priv->skipped));
    priv->skipped = 0;
    return NULL;
  }

  if (priv->dump) {
    size_t atlas_size;
    const uint8_t *atlas;
    uint32_t n_patches = 0;
    const FlutlPatch *patches = flutl_recorder_patches (rec, &n_patches);
    /* (FLUTL_DUMP_DEFER=1, tests: the worker-thread rasteriser, finished right here) */
    GST_LOG_OBJECT (render, "dump: %u placeholder(s) to resolve", n_patches);
    flutl_recorder_resolve (rec, (flutl_region *) rg, n_rg, (flutl_prim *) pr, n_pr, patches,
        n_patches);
    atlas = flutl_recorder_atlas (rec, &atlas_size);
    /* (FLUTL_DUMP_PACK=/dev/null, host-side profiles: everything but the writing) */
    if (!priv->dump_discard)
      flutl_pack_writer_add (priv->dump, width, height, rg, n_rg, pr, n_pr, atlas,
          atlas_size);
  }
  if (priv->video_pad) {
    /* fused over-video: this interval is a cue for the video side; no buffer */
    if (!priv->dump) {
      uint32_t n_patches = 0;
      const FlutlPatch *patches = flutl_recorder_patches (rec, &n_patches);
      if (n_patches)          /* (the cue's outlines, made by the workers meanwhile) */
        flutl_recorder_resolve (rec, (flutl_region *) rg, n_rg, (flutl_prim *) pr, n_pr, patches,
            n_patches);
    }
    if (priv->timing) {         /* (the cue's host time; FLUTL_TIMING's line counts cues here) */
      gint64 t = g_get_monotonic_time ();
      if (priv->n_timed++ == 0)
        priv->t_first_return = t;
      else
        priv->t_record += t - t_start;
      priv->t_last_return = t;
    }
    video_cue_push (render, ts, duration, rg, n_rg, pr, n_pr);
    /* No buffer for the base class to push: what it does after a push is done here.
     * last_out_time is what clamps the intervals of events that come out of order (a
     * <set> inside a span replays its span's timeline): left at 0, such events made
     * cues that reached back in time over cues already made -- 296 instead of 18 for
     * fuzz document 1510398, and a stale one was the first to cover t = 0.5 s
     * (gstttmlbase.c:103-108, :190). */
    base->last_out_time = ts + duration;
    return NULL;
  }
  if (priv->dump_only) {
    /* record-only test facility (no GPU needed): the base class must see a
     * buffer or its timeline bookkeeping diverges; the frame is left blank */
    if (priv->timing) {         /* (host-side profile without a device) */
      gint64 t = g_get_monotonic_time ();
      if (priv->n_timed++ == 0)
        priv->t_first_return = t;
      else
        priv->t_record += t_recorded - t_start;      /* (without writing the dump) */
      priv->t_last_return = t;
    }
    /* (one shared blank frame, read-only: no 33 MB memset per buffer in the
     * host-side profile; freed with the element) */
    {
      const gsize size = (gsize) width * height * 4;
      if (priv->blank_size != size) {
        g_free (priv->blank);     /* (buffers of the old size were pushed and are gone in these tests) */
        priv->blank = g_malloc0 (size);
        priv->blank_size = size;
      }
      return gst_buffer_new_wrapped_full (GST_MEMORY_FLAG_READONLY, priv->blank, size, 0, size,
          NULL, NULL);
    }
  }

  /* ---- device: display list -> frame ------------------------------------- */
  if (!ensure_ctx (render))
    return NULL;
  {
    GstMapInfo map;
    const gsize size = (gsize) width * height * 4;
    const gboolean async = !priv->full_d2h;
    if (async && !priv->sync_mode && !priv->dump) {
      /* 8 frames at first; as many as may be ahead of the device once the stream
       * turns out to bring outlines that take milliseconds on their workers */
      uint32_t n_def = 0;
      (void) flutl_recorder_patches (rec, &n_def);
      const guint ahead = run_ahead (priv, width, height);
      /* (the default of 8 is a number for 4K: by bytes it is the run-ahead budget
       * plus the frame being handed out -- 5 frames at 8K, not a gigabyte of
       * page-locked memory and as much HBM at the first buffer) */
      const guint first = priv->pool_prefill_set ? priv->pool_prefill
          : MIN (priv->pool_prefill, pool_budget_frames (size));
      /* (a stream with outlines: all the frames its run-ahead may use, now -- run_ahead) */
      /* (two more than may be ahead: the one being handed out and one on its way back --
       * a pool that runs dry makes frames in the middle of the stream, 14 ms each) */
      const guint deep = priv->host_strokes ? ahead
          : MIN ((priv->max_ahead ? priv->max_ahead : DEVICE_STROKES_AHEAD_MAX) + 2,
              MAX (pool_budget_frames (size), 2) - 1);
      frame_pool_prefill (priv->pool, priv->ctx, width, height,
          n_def && !priv->pool_prefill_set ? MAX (first, deep) : first);
    }
    FrameSlot *slot = frame_pool_take (priv->pool, priv->ctx, width, height, async);
    int rc = FLUTL_OK;
    uint32_t n_deferred = 0;
    (void) flutl_recorder_patches (rec, &n_deferred);
    g_mutex_lock (&priv->flight_lock);
    /* (running ahead of the finisher pays while it waits for outlines; once
     * the frames in its queue only need queueing on the device, a frame that is
     * complete waits for them -- some tens of microseconds each -- and goes
     * straight to the device again.  Frames that run ahead take pool frames
     * many buffers old, and what differs from those is what crosses PCIe.) */
    const gboolean finisher_idle = priv->in_flight == 0 ||
        (priv->waiting_in_flight == 0 && n_deferred == 0 && !priv->queue_always);
    g_mutex_unlock (&priv->flight_lock);
    /* (a frame with all its masks there, and nothing ahead of it: queued on the
     * device right here -- a thread hop would only cost) */
    if (slot && slot->frame && !priv->sync_mode && !priv->dump &&
        (n_deferred > 0 || !finisher_idle || priv->queue_always)) {
      /* hand the buffer out now; the finisher queues it on the device when its
       * masks are there (mapping the memory waits for both) */
      Ticket *ticket = ticket_new ();
      priv->n_deferred++;
      buffer = frame_pool_wrap (priv->pool, slot, size, ticket);
      pipeline_push (render, width, height, rg, n_rg, pr, n_pr, slot, buffer, ticket);
      ticket_unref (ticket);
    } else {
      uint32_t n_patches = 0;
      const FlutlPatch *patches = flutl_recorder_patches (rec, &n_patches);
      pipeline_drain (render);
      /* (only if page-locked memory ran out after strokes were deferred) */
      flutl_recorder_resolve (rec, (flutl_region *) rg, n_rg, (flutl_prim *) pr, n_pr,
          patches, n_patches);
      if (!flush_atlas (render)) {
        if (slot)
          frame_slot_free (slot);
        return NULL;
      }
      if (slot && slot->frame) {
        buffer = frame_pool_wrap (priv->pool, slot, size, NULL);
        rc = flutl_frame_submit (priv->ctx, slot->frame, rg, n_rg, pr, n_pr);
      } else if (slot) {
        buffer = frame_pool_wrap (priv->pool, slot, size, NULL);
        rc = flutl_render_frame_to_host (priv->ctx, width, height, rg, n_rg, pr, n_pr,
            slot->block, (size_t) width * 4);
      } else {                    /* no page-locked memory: plain system buffer */
        buffer = gst_buffer_new_and_alloc (size);
        gst_buffer_map (buffer, &map, GST_MAP_WRITE);
        rc = flutl_render_frame_to_host (priv->ctx, width, height, rg, n_rg, pr, n_pr,
            map.data, (size_t) width * 4);
        gst_buffer_unmap (buffer, &map);
      }
    }
    if (rc != FLUTL_OK) {
      if (slot)
        slot->stale = TRUE;     /* what it holds is unknown now */
      GST_ELEMENT_ERROR (render, LIBRARY, FAILED, ("ttmlrender: HIP render failed"),
          ("%s", flutl_strerror (rc)));
      gst_buffer_unref (buffer);
      return NULL;
    }
  }
  if (priv->overlay_meta) {
    /* the rectangle shares the frame's memory; ARGB in the composition's terms
     * is B,G,R,A in memory on little endian, i.e. what the frame holds */
    GstBuffer *pix = gst_buffer_copy_region (buffer, GST_BUFFER_COPY_MEMORY, 0, -1);
    GstVideoOverlayRectangle *rect;
    GstVideoOverlayComposition *comp;
    gst_buffer_add_video_meta (pix, GST_VIDEO_FRAME_FLAG_NONE,
        GST_VIDEO_OVERLAY_COMPOSITION_FORMAT_RGB, width, height);
    rect = gst_video_overlay_rectangle_new_raw (pix, 0, 0, width, height,
        GST_VIDEO_OVERLAY_FORMAT_FLAG_PREMULTIPLIED_ALPHA);
    comp = gst_video_overlay_composition_new (rect);
    gst_buffer_add_video_overlay_composition_meta (buffer, comp);
    gst_video_overlay_composition_unref (comp);
    gst_video_overlay_rectangle_unref (rect);
    gst_buffer_unref (pix);
    GST_LOG_OBJECT (render, "attached a %dx%d overlay composition", width, height);
  }
  if (priv->timing) {
    gint64 t_end = g_get_monotonic_time ();
    /* (averages from the second buffer on, like the span: the first one makes
     * every mask of the document) */
    if (priv->timing_trace)     /* FLUTL_TIMING=2: one line per buffer */
      g_printerr ("ttmlrender-hip buffer %u: record_us=%" G_GINT64_FORMAT " device_us=%"
          G_GINT64_FORMAT " hits=%u recoloured=%u misses=%u\n", priv->n_timed,
          t_recorded - t_start, t_end - t_recorded, priv->region_hits, priv->region_recoloured,
          priv->region_misses);
    if (priv->n_timed++ == 0) {
      priv->t_first_return = t_end;
    } else {
      priv->t_record += t_recorded - t_start;
      priv->t_device += t_end - t_recorded;
    }
    priv->t_last_return = t_end;
  }
  return buffer;
}

/* ---- caps (reference :1480-1531) ----------------------------------------------- */
static void
gst_ttmlrender_fixate_caps (GstTTMLBase *base, GstCaps *caps)
{
  GstTTMLRender *render = GST_TTMLRENDER (base);
  GstStructure *s = gst_caps_get_structure (caps, 0);
  while (gst_caps_get_size (caps) > 1)
    gst_caps_remove_structure (caps, gst_caps_get_size (caps) - 1);
  gst_structure_fixate_field_nearest_int (s, "width", render->window_width);
  gst_structure_fixate_field_nearest_int (s, "height", render->window_height);
}

static void
gst_ttmlrender_complete_caps (GstTTMLBase *base, GstCaps *caps)
{
  gst_structure_fixate_field_nearest_fraction (gst_caps_get_structure (caps, 0),
      "pixel-aspect-ratio", base->state.par_num, base->state.par_den);
}

static void
gst_ttmlrender_setcaps (GstTTMLBase *base, GstCaps *caps)
{
  GstStructure *s = gst_caps_get_structure (caps, 0);
  gint w = base->state.frame_width, h = base->state.frame_height;
  gst_structure_get_int (s, "width", &w);
  gst_structure_get_int (s, "height", &h);
  base->state.frame_width = w;
  base->state.frame_height = h;
}

/* ---- properties (reference :1533-1608) ------------------------------------------ */
static void
gst_ttmlrender_get_property (GObject *object, guint id, GValue *value,
    GParamSpec *pspec)
{
  GstTTMLRender *render = GST_TTMLRENDER (object);
  switch (id) {
    case PROP_DEFAULT_FONT_FAMILY:
      g_value_set_string (value, render->default_font_family);
      break;
    case PROP_DEFAULT_FONT_SIZE:
      g_value_set_string (value, render->default_font_size);
      break;
    case PROP_DEFAULT_TEXT_ALIGN:
      g_value_set_enum (value, render->default_text_align);
      break;
    case PROP_DEFAULT_DISPLAY_ALIGN:
      g_value_set_enum (value, render->default_display_align);
      break;
    case PROP_WINDOW_WIDTH:
      g_value_set_uint (value, render->window_width);
      break;
    case PROP_WINDOW_HEIGHT:
      g_value_set_uint (value, render->window_height);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      break;
  }
}

static void
gst_ttmlrender_set_property (GObject *object, guint id, const GValue *value,
    GParamSpec *pspec)
{
  GstTTMLRender *render = GST_TTMLRENDER (object);
  switch (id) {
    case PROP_DEFAULT_FONT_FAMILY:
      g_free (render->default_font_family);
      render->default_font_family = g_value_dup_string (value);
      break;
    case PROP_DEFAULT_FONT_SIZE:
      g_free (render->default_font_size);
      render->default_font_size = g_value_dup_string (value);
      break;
    case PROP_DEFAULT_TEXT_ALIGN:
      render->default_text_align = g_value_get_enum (value);
      break;
    case PROP_DEFAULT_DISPLAY_ALIGN:
      render->default_display_align = g_value_get_enum (value);
      break;
    case PROP_WINDOW_WIDTH:
      /* reference: the window size is also the frame size until caps say
       * otherwise (:1590-1603) */
      render->window_width = g_value_get_uint (value);
      render->base.state.frame_width = render->window_width;
      break;
    case PROP_WINDOW_HEIGHT:
      render->window_height = g_value_get_uint (value);
      render->base.state.frame_height = render->window_height;
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (object, id, pspec);
      break;
  }
}

/* FLUTL_TIMING=1: one line per document on stderr (tools/e2e_timing.py reads it).
 * record = streaming-thread time per buffer from scene state to display list;
 * device = host time per buffer spent queuing the device work, mostly on the
 * finisher thread (or, with FLUTL_FULL_D2H, waiting for it); span = end of the first gen_buffer to end of the last; drained = to
 * the moment every frame of the document was complete in host memory. */
static void
timing_report (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  gint64 t_drained;
  guint64 delta = 0, sj = 0, sb = 0, sq = 0;
  if (!priv->timing || !priv->n_timed)
    return;
  flutl_recorder_stroke_stats (&sj, &sb, &sq);
  uint32_t rm = 0, rc_ = 0;
  uint64_t pp = 0, pe = 0;
  flutl_recorder_run_stats (priv->rec, &rm, &rc_);
  flutl_recorder_prepare_stats (&pp, &pe);
  uint64_t we = 0, wd = 0, wc = 0;
  flutl_recorder_stroke_split (&we, &wd, &wc);
  uint64_t composed_chk = 0;
  const uint64_t composed_n = flutl_recorder_composed_paths (&composed_chk);   /* (process-wide) */
  uint64_t by_cairo = 0, dev_fills = 0;
  flutl_recorder_raster_stats (&by_cairo, &dev_fills);                         /* (process-wide) */
  pipeline_drain (render);
  if (priv->ctx) {
    /* (fused over-video: the video thread uses the context as well) */
    g_mutex_lock (&priv->ctx_lock);
    flutl_ctx_sync (priv->ctx);
    t_drained = g_get_monotonic_time ();
    delta = flutl_ctx_delta_bytes (priv->ctx) - priv->delta_reported;
    priv->delta_reported += delta;
    g_mutex_unlock (&priv->ctx_lock);
  } else {
    t_drained = priv->t_last_return;
  }
  g_printerr ("ttmlrender-hip timing: device=%d buffers=%u record_us=%.1f device_us=%.1f "
      "span_us=%" G_GINT64_FORMAT " drained_us=%" G_GINT64_FORMAT " delta_bytes=%.0f "
      "deferred=%u resolve_wait_us=%.1f atlas_upload_us=%.1f build_us=%.1f layout_us=%.1f region_hits=%u region_misses=%u strokes=%u stroke_us=%.0f stroke_queued_us=%.0f pure_wait_us=%.0f submit_us=%.1f push_block_us=%.1f runs_merged=%u runs_clashed=%u path_us=%.0f extents_us=%.0f miss_us=%.1f recoloured=%u worker_extents_us=%.0f worker_draw_us=%.0f worker_crop_us=%.0f polygons=%u polygon_edges=%" G_GUINT64_FORMAT " polygons_sequential=%u composed_paths=%" G_GUINT64_FORMAT " composed_checked=%" G_GUINT64_FORMAT " masks_by_cairo=%" G_GUINT64_FORMAT " device_fills=%" G_GUINT64_FORMAT " t_first_us=%" G_GINT64_FORMAT " t_end_us=%" G_GINT64_FORMAT "\n",
      priv->device, priv->n_timed, (double) priv->t_record / MAX (priv->n_timed - 1, 1),
      (double) (priv->t_device + priv->t_submit) / MAX (priv->n_timed - 1, 1),
      priv->t_last_return - priv->t_first_return,
      MAX (t_drained, priv->t_last_return) - priv->t_first_return,
      (double) delta / priv->n_timed, priv->n_deferred,
      (double) priv->t_resolve / MAX (priv->n_deferred, 1),
      (double) priv->t_flush / MAX (priv->n_deferred, 1),
      (double) priv->t_build / MAX (priv->n_timed - 1, 1),
      (double) priv->t_layout / MAX (priv->n_timed - 1, 1), priv->region_hits,
      priv->region_misses, (guint) sj, sj ? (double) sb / sj : 0.0, sj ? (double) sq / sj : 0.0,
      (double) flutl_recorder_resolve_wait_us () / MAX (priv->n_deferred, 1),
      (double) priv->t_submit / MAX (priv->n_deferred, 1),
      (double) priv->t_push_block / MAX (priv->n_deferred, 1), rm, rc_,
      sj ? (double) pp / sj : 0.0, sj ? (double) pe / sj : 0.0,
      (double) priv->t_miss / MAX (priv->n_timed - 1, 1), priv->region_recoloured,
      sj ? (double) we / sj : 0.0, sj ? (double) wd / sj : 0.0, sj ? (double) wc / sj : 0.0,
      priv->n_polygons, priv->polygon_edges, priv->ctx ? flutl_ctx_scan_fallbacks (priv->ctx) : 0,
      (guint64) composed_n, (guint64) composed_chk, (guint64) by_cairo, (guint64) dev_fills,
      /* (CLOCK_MONOTONIC: comparable across the elements and processes of one box) */
      priv->t_first_return, MAX (t_drained, priv->t_last_return));
  priv->t_push_block = 0;
  priv->region_hits = priv->region_misses = priv->region_recoloured = 0;
  priv->n_timed = priv->n_deferred = 0;
  priv->n_polygons = 0;
  priv->polygon_edges = 0;
  priv->t_record = priv->t_device = priv->t_submit = priv->t_flush = priv->t_resolve = 0;
  priv->t_build = priv->t_layout = priv->t_miss = 0;
}

/* ---- reset / dispose / init (reference :1610-1750) ------------------------------- */
static void
gst_ttmlrender_reset (GstTTMLBase *base)
{
  GstTTMLRender *render = GST_TTMLRENDER (base);
  GstTTMLRenderPrivate *priv = get_priv (render);
  timing_report (render);       /* (the document ends here: its frames are drained) */
  priv->n_pushed = 0;
  /* per-document caches: decoded images, layouts, glyph/outline atlas (host + HBM) */
  if (priv->images) {
    g_hash_table_unref (priv->images);
    priv->images = NULL;
  }
  if (priv->layouts) {
    g_hash_table_unref (priv->layouts);
    priv->layouts = NULL;
  }
  if (priv->region_cache)       /* (keyed on layouts and decoded images of the document) */
    g_ptr_array_set_size (priv->region_cache, 0);
  if (priv->plans)              /* (they hold decoded images of the document) */
    g_hash_table_remove_all (priv->plans);
  /* (fused over-video: the document's cues are still waiting for their video
   * frames, and they name masks of the atlas) */
  g_mutex_lock (&priv->video_lock);
  const gboolean cues_waiting = priv->video_pad && !g_queue_is_empty (&priv->cues);
  g_mutex_unlock (&priv->video_lock);
  if (!cues_waiting)
    caches_drop (render);
  if (priv->dump)
    flutl_pack_writer_flush (priv->dump);
}

static void
gst_ttmlrender_dispose (GObject *object)
{
  GstTTMLRender *render = GST_TTMLRENDER (object);
  GstTTMLRenderPrivate *priv = get_priv (render);

  gst_ttmlrender_reset (GST_TTMLBASE (render));
  timing_report (render);
  pipeline_stop (render);
  if (priv->dump) {
    flutl_pack_writer_free (priv->dump);
    priv->dump = NULL;
  }
  /* (the recorders first: a polygon entry they still hold gives its block of page-locked
   * memory back to the context) */
  if (priv->sub) {
    flutl_recorder_free (priv->sub);
    priv->sub = NULL;
  }
  if (priv->rec) {
    flutl_recorder_free (priv->rec);
    priv->rec = NULL;
  }
  if (priv->ctx) {
    flutl_ctx_destroy (priv->ctx);
    priv->ctx = NULL;
  }
  if (priv->pool) {
    frame_pool_unref (priv->pool);
    priv->pool = NULL;
  }
  {
    VideoCue *c;
    while ((c = g_queue_pop_head (&priv->cues)))
      video_cue_free (c);
  }
  if (priv->region_cache) {
    g_ptr_array_unref (priv->region_cache);
    priv->region_cache = NULL;
  }
  if (priv->color_sources) {
    g_array_unref (priv->color_sources);
    priv->color_sources = NULL;
  }
  g_free (priv->blank);
  priv->blank = NULL;
  priv->blank_size = 0;
  if (priv->plans) {
    g_hash_table_unref (priv->plans);
    priv->plans = NULL;
  }
  if (priv->plan_key) {
    g_free (priv->plan_key->data);
    g_free (priv->plan_key);
    priv->plan_key = NULL;
  }
  if (priv->jobs) {
    g_async_queue_unref (priv->jobs);
    priv->jobs = NULL;
    g_mutex_clear (&priv->flight_lock);
    g_cond_clear (&priv->flight_cond);
  }
  g_clear_pointer (&render->default_font_family, g_free);
  g_clear_pointer (&render->default_font_size, g_free);
  g_clear_object (&render->pango_context);
  if (render->downloader) {
    fludownloader_helper_downloader_free (render->downloader);
    render->downloader = NULL;
  }
  G_OBJECT_CLASS (gst_ttmlrender_parent_class)->dispose (object);
}

/* ---- fused over-video: the optional request pad "video_sink"  -------
 * The reference renders overlay frames and leaves the mixing with the video to a
 * downstream compositor (plugins/ttml/README.md:47-50).  With FLUTL_VIDEO_PAD=1 in
 * the environment the element class has one more pad template, a request sink
 * pad for BGRA/BGRx video (an environment knob and not a property or an always
 * pad, so that gst-inspect of the default configuration stays identical to the
 * reference's).  When that pad is linked:
 *   - the TTML side still runs the untouched base class, but gen_buffer RECORDS
 *     each timeline interval as a cue (display list + time span) and pushes
 *     nothing;
 *   - every video buffer gets the cue covering its PTS composited into it by the
 *     fused kernel (flutl_blend_over_host: only the rows the regions span cross
 *     PCIe) and leaves on the src pad with the video's caps and timestamps.
 * Frame size = video size: the TTML side waits for the video caps. */
/* gen_buffer in video mode: the interval becomes a cue; nothing is pushed. */
static void
video_cue_push (GstTTMLRender *render, GstClockTime ts, GstClockTime duration,
    const flutl_region *rg, guint32 n_rg, const flutl_prim *pr, guint32 n_pr)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  VideoCue *c = g_new0 (VideoCue, 1);
  c->start = ts;
  c->end = GST_CLOCK_TIME_IS_VALID (duration) ? ts + duration : GST_CLOCK_TIME_NONE;
  c->n_rg = n_rg;
  c->n_pr = n_pr;
  c->regions = g_memdup (rg, n_rg * sizeof *rg);
  c->prims = g_memdup (pr, n_pr * sizeof *pr);
  g_mutex_lock (&priv->video_lock);
  /* The parser runs through a whole document at once: without a bound a long or live
   * stream queues cues (and pins the masks they name) without limit.  The TTML thread
   * waits here while VIDEO_CUES_AHEAD cues are ahead of the video -- unless nothing
   * will ever take them (video at EOS, flushing, shutting down). */
  while (g_queue_get_length (&priv->cues) >= VIDEO_CUES_AHEAD && !priv->video_flushing &&
      !priv->video_eos)
    g_cond_wait (&priv->video_cond, &priv->video_lock);
  g_queue_push_tail (&priv->cues, c);
  g_cond_broadcast (&priv->video_cond);
  g_mutex_unlock (&priv->video_lock);
}

/* TRUE when no cue is waiting for its video frames (always, outside video mode):
 * only then may the masks the cues name be dropped. */
static gboolean
video_cues_drained (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  gboolean drained;
  if (!priv->video_pad)
    return TRUE;
  g_mutex_lock (&priv->video_lock);
  drained = g_queue_is_empty (&priv->cues);
  g_mutex_unlock (&priv->video_lock);
  return drained;
}

static GstFlowReturn
video_chain (GstPad *pad, GstObject *parent, GstBuffer *buffer)
{
  GstTTMLRender *render = GST_TTMLRENDER (parent);
  GstTTMLRenderPrivate *priv = get_priv (render);
  GstTTMLBase *base = GST_TTMLBASE (render);
  const GstClockTime t = GST_BUFFER_PTS (buffer);
  VideoCue *cue = NULL;
  GstFlowReturn ret;

  /* the cue of this frame's time: wait until the TTML side has got that far */
  g_mutex_lock (&priv->video_lock);
  for (;;) {
    VideoCue *c;
    while ((c = g_queue_peek_head (&priv->cues)) && GST_CLOCK_TIME_IS_VALID (c->end) &&
        GST_CLOCK_TIME_IS_VALID (t) && c->end <= t)
    {
      video_cue_free (g_queue_pop_head (&priv->cues));      /* video time only moves on */
      g_cond_broadcast (&priv->video_cond);                 /* (room for the TTML side) */
    }
    c = g_queue_peek_head (&priv->cues);
    if (c && (!GST_CLOCK_TIME_IS_VALID (t) || c->start <= t)) {
      cue = c;
      break;
    }
    if (c || priv->ttml_eos || priv->video_flushing)
      break;                    /* a gap before the next cue, or no more subtitles */
    g_cond_wait (&priv->video_cond, &priv->video_lock);
  }
  if (priv->video_flushing) {
    g_mutex_unlock (&priv->video_lock);
    gst_buffer_unref (buffer);
    return GST_FLOW_FLUSHING;
  }
  if (cue && cue->n_rg && !priv->dump_only) {   /* (record-only test facility: no device) */
    GstMapInfo map;
    int rc;
    const gint w = priv->video_width, h = priv->video_height;
    g_mutex_unlock (&priv->video_lock);         /* (the head cue stays: only this thread pops) */
    buffer = gst_buffer_make_writable (buffer);
    if (!gst_buffer_map (buffer, &map, GST_MAP_READWRITE) || map.size < (gsize) w * h * 4) {
      GST_ELEMENT_ERROR (render, STREAM, FAILED, ("ttmlrender: cannot map the video frame"),
          (NULL));
      gst_buffer_unref (buffer);
      return GST_FLOW_ERROR;
    }
    /* (the TTML thread uses the device too, for blurred outlines: ctx_lock) */
    g_mutex_lock (&priv->ctx_lock);
    rc = ensure_ctx (render) && flush_atlas (render) ? FLUTL_OK : FLUTL_ENODEV;
    if (rc == FLUTL_OK)
      rc = flutl_blend_over_host (priv->ctx, w, h, cue->regions, cue->n_rg, cue->prims,
          cue->n_pr, map.data, (size_t) w * 4);
    g_mutex_unlock (&priv->ctx_lock);
    gst_buffer_unmap (buffer, &map);
    if (rc != FLUTL_OK) {
      GST_ELEMENT_ERROR (render, LIBRARY, FAILED, ("ttmlrender: HIP render failed"),
          ("flutl_blend_over_host: %s", flutl_strerror (rc)));
      gst_buffer_unref (buffer);
      return GST_FLOW_ERROR;
    }
  } else {
    g_mutex_unlock (&priv->video_lock);
  }
  ret = gst_pad_push (base->srcpad, buffer);
  return ret;
}

static gboolean
video_event (GstPad *pad, GstObject *parent, GstEvent *event)
{
  GstTTMLRender *render = GST_TTMLRENDER (parent);
  GstTTMLRenderPrivate *priv = get_priv (render);
  GstTTMLBase *base = GST_TTMLBASE (render);
  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_CAPS: {
      GstCaps *caps;
      GstStructure *st;
      gint w = 0, h = 0;
      gst_event_parse_caps (event, &caps);
      st = gst_caps_get_structure (caps, 0);
      gst_structure_get_int (st, "width", &w);
      gst_structure_get_int (st, "height", &h);
      if (w <= 0 || h <= 0 || w > 32767 || h > 32767) {
        gst_event_unref (event);
        return FALSE;
      }
      /* the video's caps are the output's; the TTML frame is the video frame.
       * (base->state.frame_* is NOT set here but by the first gen_buffer, as in
       * the stand-alone element: the base class normalises a length differently
       * when the frame size is already known while it parses -- fontSize="1c"
       * becomes a cell WIDTH then, gstttmlattribute.c:702-704 -- and the parser
       * runs on the other thread) */
      /* The caps go downstream BEFORE the TTML side is let go (ttml_chain_in_video_mode
       * waits for video_have_caps): the base class negotiates the src pad on the TTML
       * thread right before gen_buffer when the pad has no caps yet
       * (gstttmlbase.c:115, :922-975) -- with the overlay template fixated to the
       * window size, which would land between the video's caps and its buffers. */
      {
        const gboolean ok = gst_pad_push_event (base->srcpad, event);
        g_mutex_lock (&priv->video_lock);
        priv->video_width = w;
        priv->video_height = h;
        priv->video_have_caps = ok;
        g_cond_broadcast (&priv->video_cond);
        g_mutex_unlock (&priv->video_lock);
        return ok;
      }
    }
    case GST_EVENT_FLUSH_START:
      g_mutex_lock (&priv->video_lock);
      priv->video_flushing = TRUE;
      g_cond_broadcast (&priv->video_cond);
      g_mutex_unlock (&priv->video_lock);
      break;
    case GST_EVENT_FLUSH_STOP:
      g_mutex_lock (&priv->video_lock);
      priv->video_flushing = FALSE;
      priv->video_eos = FALSE;
      g_mutex_unlock (&priv->video_lock);
      break;
    case GST_EVENT_EOS:
      g_mutex_lock (&priv->video_lock);
      priv->video_eos = TRUE;             /* (nobody will take further cues) */
      g_cond_broadcast (&priv->video_cond);
      g_mutex_unlock (&priv->video_lock);
      break;
    default:
      break;
  }
  /* stream-start, segment, tags, EOS ...: the output stream is the video stream */
  return gst_pad_push_event (base->srcpad, event);
}

/* The TTML pad's events in video mode: what would announce or end a stream
 * downstream is not forwarded (the video side does that); everything else goes
 * to the base class's handler. */
static gboolean
ttml_event_in_video_mode (GstPad *pad, GstObject *parent, GstEvent *event)
{
  GstTTMLRender *render = GST_TTMLRENDER (parent);
  GstTTMLRenderPrivate *priv = get_priv (render);
  if (priv->video_pad) {
    switch (GST_EVENT_TYPE (event)) {
      case GST_EVENT_EOS:
        g_mutex_lock (&priv->video_lock);
        priv->ttml_eos = TRUE;
        g_cond_broadcast (&priv->video_cond);
        g_mutex_unlock (&priv->video_lock);
        gst_event_unref (event);
        return TRUE;
      case GST_EVENT_STREAM_START:
      case GST_EVENT_TAG:
        gst_event_unref (event);
        return TRUE;
      default:
        break;
    }
  }
  return priv->ttml_event (pad, parent, event);
}

/* The TTML pad's data in video mode: held back until the video's caps are on the src
 * pad, so that the base class finds the pad negotiated and leaves it alone (see
 * video_event). */
static GstFlowReturn
ttml_chain_in_video_mode (GstPad *pad, GstObject *parent, GstBuffer *buffer)
{
  GstTTMLRender *render = GST_TTMLRENDER (parent);
  GstTTMLRenderPrivate *priv = get_priv (render);
  if (priv->video_pad) {
    gboolean flushing;
    g_mutex_lock (&priv->video_lock);
    while (!priv->video_have_caps && !priv->video_flushing && !priv->video_eos)
      g_cond_wait (&priv->video_cond, &priv->video_lock);
    flushing = priv->video_flushing || !priv->video_have_caps;
    g_mutex_unlock (&priv->video_lock);
    if (flushing) {
      gst_buffer_unref (buffer);
      return GST_FLOW_FLUSHING;
    }
  }
  return priv->ttml_chain (pad, parent, buffer);
}

/* Going down: nothing may keep waiting for the other side (pad deactivation takes
 * the streaming threads' locks). */
static GstStateChangeReturn
gst_ttmlrender_change_state (GstElement *element, GstStateChange transition)
{
  GstTTMLRenderPrivate *priv = get_priv (GST_TTMLRENDER (element));
  if (transition == GST_STATE_CHANGE_PAUSED_TO_READY) {
    g_mutex_lock (&priv->video_lock);
    priv->video_flushing = TRUE;
    g_cond_broadcast (&priv->video_cond);
    g_mutex_unlock (&priv->video_lock);
  } else if (transition == GST_STATE_CHANGE_READY_TO_PAUSED) {
    g_mutex_lock (&priv->video_lock);
    priv->video_flushing = FALSE;
    priv->video_eos = priv->ttml_eos = FALSE;
    priv->video_have_caps = FALSE;
    g_mutex_unlock (&priv->video_lock);
  }
  return GST_ELEMENT_CLASS (gst_ttmlrender_parent_class)->change_state (element, transition);
}

static GstPad *
gst_ttmlrender_request_new_pad (GstElement *element, GstPadTemplate *templ,
    const gchar *name, const GstCaps *caps)
{
  GstTTMLRender *render = GST_TTMLRENDER (element);
  GstTTMLRenderPrivate *priv = get_priv (render);
  GstPad *pad;
  if (priv->video_pad || g_strcmp0 (GST_PAD_TEMPLATE_NAME_TEMPLATE (templ), "video_sink"))
    return NULL;
  pad = gst_pad_new_from_template (templ, "video_sink");
  gst_pad_set_chain_function (pad, GST_DEBUG_FUNCPTR (video_chain));
  gst_pad_set_event_function (pad, GST_DEBUG_FUNCPTR (video_event));
  priv->video_pad = pad;
  gst_pad_set_active (pad, TRUE);
  gst_element_add_pad (element, pad);
  return pad;
}

static void
gst_ttmlrender_release_pad (GstElement *element, GstPad *pad)
{
  GstTTMLRenderPrivate *priv = get_priv (GST_TTMLRENDER (element));
  if (pad != priv->video_pad)
    return;
  g_mutex_lock (&priv->video_lock);
  priv->video_flushing = TRUE;
  g_cond_broadcast (&priv->video_cond);
  g_mutex_unlock (&priv->video_lock);
  gst_pad_set_active (pad, FALSE);
  gst_element_remove_pad (element, pad);
  priv->video_pad = NULL;
}

static void
gst_ttmlrender_class_init (GstTTMLRenderClass *klass)
{
  GObjectClass *gobject_class = G_OBJECT_CLASS (klass);
  GstElementClass *element_class = GST_ELEMENT_CLASS (klass);
  GstTTMLBaseClass *base_class = GST_TTMLBASE_CLASS (klass);

  gobject_class->dispose = gst_ttmlrender_dispose;
  gobject_class->set_property = gst_ttmlrender_set_property;
  gobject_class->get_property = gst_ttmlrender_get_property;

  /* The base class creates its libxml2 push parser on the streaming thread
   * (gstttmlbase.c: xmlCreatePushParserCtxt in the sink chain function) and nothing in
   * the plugin ever calls xmlInitParser: with several elements in one process their
   * streaming threads race through libxml2's lazy set-up of its globals, and now and
   * then one of them locks a mutex another is just replacing -- it never returns
   * (caught with tools/hang_hunt.py: a thread in xmlInitializeGlobalState <-
   * xmlCreatePushParserCtxt, the other chains waiting for its preroll for ever; 2 of
   * some 700 runs of four and five chains).  libxml2 asks for exactly this call, once,
   * before any thread uses the library: here, where the class is made. */
  xmlInitParser ();

  g_object_class_install_property (gobject_class, PROP_DEFAULT_FONT_FAMILY,
      g_param_spec_string ("default_font_family", "Default font family",
          "Font family to use when the TTML file does not explicitly set one",
          "Sans", G_PARAM_READWRITE));
  g_object_class_install_property (gobject_class, PROP_DEFAULT_FONT_SIZE,
      g_param_spec_string ("default_font_size", "Default font size",
          "Font size to use when the TTML file does not explicitly set one",
          "default", G_PARAM_READWRITE));
  g_object_class_install_property (gobject_class, PROP_DEFAULT_TEXT_ALIGN,
      g_param_spec_enum ("default_text_align", "Default text alignment",
          "Text alignment to use when the TTML file does not explicitly set one",
          GST_TTML_TEXT_ALIGN_SPEC, GST_TTML_TEXT_ALIGN_LEFT, G_PARAM_READWRITE));
  g_object_class_install_property (gobject_class, PROP_DEFAULT_DISPLAY_ALIGN,
      g_param_spec_enum ("default_display_align", "Default display alignment",
          "Display alignment to use when the TTML file does not explicitly set one",
          GST_TTML_DISPLAY_ALIGN_SPEC, GST_TTML_DISPLAY_ALIGN_BEFORE,
          G_PARAM_READWRITE));
  g_object_class_install_property (gobject_class, PROP_WINDOW_WIDTH,
      g_param_spec_uint ("window-width", "Default window width",
          "Default window width used when rendering", 0, 4096,
          DEFAULT_WINDOW_WIDTH, G_PARAM_READWRITE));
  g_object_class_install_property (gobject_class, PROP_WINDOW_HEIGHT,
      g_param_spec_uint ("window-height", "Default window width",
          "Default window height used when rendering", 0, 4096,
          DEFAULT_WINDOW_HEIGHT, G_PARAM_READWRITE));

  gst_element_class_add_pad_template (element_class,
      gst_static_pad_template_get (&ttmlrender_src_template));
  gst_element_class_set_details_simple (element_class, "TTML subtitle renderer",
      "Codec/Decoder/Subtitle",
      "Render TTML subtitle streams into a video stream (MI355X / HIP compositor)",
      "fluttml-hip");

  if (g_getenv ("FLUTL_VIDEO_PAD")) {
    static GstStaticPadTemplate video_template =
        GST_STATIC_PAD_TEMPLATE ("video_sink", GST_PAD_SINK, GST_PAD_REQUEST,
            GST_STATIC_CAPS ("video/x-raw, format=(string){ BGRA, BGRx }, "
                "width=(int)[1,32767], height=(int)[1,32767]"));
    gst_element_class_add_pad_template (element_class,
        gst_static_pad_template_get (&video_template));
    element_class->request_new_pad = gst_ttmlrender_request_new_pad;
    element_class->release_pad = gst_ttmlrender_release_pad;
    element_class->change_state = gst_ttmlrender_change_state;
  }

  base_class->gen_buffer = gst_ttmlrender_gen_buffer;
  base_class->fixate_caps = gst_ttmlrender_fixate_caps;
  base_class->complete_caps = gst_ttmlrender_complete_caps;
  base_class->src_setcaps = gst_ttmlrender_setcaps;
  base_class->reset = gst_ttmlrender_reset;
}

static void
gst_ttmlrender_init (GstTTMLRender *render)
{
  GstTTMLRenderPrivate *priv = get_priv (render);
  static gsize registered = 0;
  const gchar *dump;

  if (g_once_init_enter (&registered)) {
    hip_attr_klass[HIP_ATTR_OVERLINE].type = pango_attr_type_register ("Overline");
    hip_attr_klass[HIP_ATTR_INVISIBLE].type =
        pango_attr_type_register ("Invisibility");
    hip_attr_klass[HIP_ATTR_REVERSE].type = pango_attr_type_register ("Reverse");
    hip_attr_klass[HIP_ATTR_REVERSE_OBLIQUE].type =
        pango_attr_type_register ("ReverseOblique");
    g_once_init_leave (&registered, 1);
  }

  /* The reference takes its context from pango_cairo_font_map_get_default () here
   * (:1727), in whatever thread creates the element: every ttmlrender of a process
   * made by one thread then shares ONE font map, and a PangoFontMap is not safe
   * for use from several streaming threads at once -- four elements in one
   * gst-launch crash or hang one run in three, the reference's own included.  A font
   * map of its own per element: same backend, same defaults, same fonts. */
  {
    PangoFontMap *fontmap = pango_cairo_font_map_new ();
    render->pango_context = pango_font_map_create_context (fontmap);
    g_object_unref (fontmap);   /* (the context keeps it) */
  }
  /* reference :1739: the instance default is "Serif" whatever the param spec says */
  render->default_font_family = g_strdup ("Serif");
  render->default_font_size = NULL;
  render->window_width = DEFAULT_WINDOW_WIDTH;
  render->window_height = DEFAULT_WINDOW_HEIGHT;
  render->cached_images = NULL;
  render->downloader = fludownloader_helper_downloader_new (NULL);

  priv->rec = flutl_recorder_new ();
  priv->sub = flutl_recorder_new_child (priv->rec);
  priv->blurred = NULL;
  priv->pool = frame_pool_new ();
  priv->ctx = NULL;
  priv->ctx_failed = FALSE;
  priv->images = NULL;
  dump = g_getenv ("FLUTL_DUMP_PACK");
  priv->dump = dump ? flutl_pack_writer_new (dump) : NULL;
  priv->dump_discard = dump && strcmp (dump, "/dev/null") == 0;
  priv->dump_only = g_getenv ("FLUTL_DUMP_ONLY") != NULL;
  priv->dump_defer = g_getenv ("FLUTL_DUMP_DEFER") != NULL;
  priv->timing = g_getenv ("FLUTL_TIMING") != NULL;
  priv->timing_trace = priv->timing && atoi (g_getenv ("FLUTL_TIMING")) >= 2;
  priv->full_d2h = g_getenv ("FLUTL_FULL_D2H") != NULL;
  priv->atlas_limit = (gsize) (g_getenv ("FLUTL_ATLAS_LIMIT_MB")
      ? MAX (1, atoi (g_getenv ("FLUTL_ATLAS_LIMIT_MB"))) : 256) << 20;
  if (g_getenv ("FLUTL_ATLAS_LIMIT_BYTES"))     /* tests: force the start-over path */
    priv->atlas_limit = (gsize) g_ascii_strtoull (g_getenv ("FLUTL_ATLAS_LIMIT_BYTES"), NULL, 10);
  priv->overlay_meta = g_getenv ("FLUTL_OVERLAY_COMPOSITION") != NULL;
  g_mutex_init (&priv->video_lock);
  g_mutex_init (&priv->ctx_lock);
  g_cond_init (&priv->video_cond);
  g_queue_init (&priv->cues);
  /* (the base class has installed its handler on the TTML pad by now) */
  priv->ttml_event = GST_PAD_EVENTFUNC (render->base.sinkpad);
  gst_pad_set_event_function (render->base.sinkpad, ttml_event_in_video_mode);
  priv->ttml_chain = GST_PAD_CHAINFUNC (render->base.sinkpad);
  gst_pad_set_chain_function (render->base.sinkpad, ttml_chain_in_video_mode);
  priv->sync_mode = g_getenv ("FLUTL_SYNC") != NULL;
  priv->queue_always = !(g_getenv ("FLUTL_QUEUE_ALWAYS") && g_getenv ("FLUTL_QUEUE_ALWAYS")[0] == '0');
  priv->max_ahead = g_getenv ("FLUTL_MAX_AHEAD")
      ? (guint) CLAMP (atoi (g_getenv ("FLUTL_MAX_AHEAD")), 1, 4 * MAX_IN_FLIGHT) : 0;
  priv->pool_prefill_set = g_getenv ("FLUTL_POOL_PREFILL") != NULL;
  priv->pool_prefill = g_getenv ("FLUTL_POOL_PREFILL")
      ? (guint) CLAMP (atoi (g_getenv ("FLUTL_POOL_PREFILL")), 0, POOL_MAX_FREE) : 8;
  priv->jobs = g_async_queue_new ();
  g_mutex_init (&priv->flight_lock);
  g_cond_init (&priv->flight_cond);
  priv->host_strokes = g_getenv ("FLUTL_HOST_STROKES") != NULL;
  priv->no_layout_cache = g_getenv ("FLUTL_NO_LAYOUT_CACHE") != NULL;
  priv->no_region_cache = g_getenv ("FLUTL_NO_REGION_CACHE") != NULL;
  priv->no_recolour = g_getenv ("FLUTL_NO_RECOLOUR") != NULL;
  priv->no_span_plans = g_getenv ("FLUTL_NO_SPAN_PLANS") != NULL;
}
