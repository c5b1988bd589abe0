/*
 * flutl_recorder.c -- see flutl_recorder.h.
 *
 * Every rule below that turns a drawing verb into a primitive restates what
 * Cairo 1.16.0 does for the corresponding call of the reference and is pinned
 * by tests against the real library and against frames of the reference plugin
 * (tests/test_oracle_kats.py, tests/test_scenes_vs_reference.py):
 *   - colour: premultiply in double, (c * 65535 + 0.5) -> 16 bit, keep the high
 *     byte (cairo-color.c);
 *   - path -> fixed point: round to nearest 1/256, relative segments are
 *     converted on their own and added (cairo-path-fixed.c);
 *   - a filled box that is not pixel aligned, and any stroke, goes through the
 *     span renderer: opaque colours -- or ANY colour while the target surface
 *     is still clear -- take Cairo's in-place lerp (FLUTL_RULE_LERP), all
 *     other cases go to pixman (cairo-image-compositor.c);
 *   - glyphs land at floor (device position + 0.5) and always go to pixman.
 */
#ifndef _GNU_SOURCE
#define _GNU_SOURCE               /* mremap */
#endif
#include "flutl_recorder.h"
#include "flutl_stroker.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>

#define ATLAS_SLACK 16u
#define ATLAS_ALIGN 16u

typedef struct
{
  void *font;      /* cairo_scaled_font_t *, referenced */
  uint32_t glyph;
  uint32_t used;
  double lin[4];   /* xx, yx, xy, yy of the CTM the glyph is shown under */
  uint32_t offset; /* atlas offset of the tight A8 bitmap */
  int w, h;        /* tight bitmap size; 0 x 0 = blank glyph */
  int bx, by;      /* bitmap top-left relative to the glyph origin */
  int mbox[4];     /* the font's metric box of the glyph (device pixels, relative to
                    * the origin): what Cairo tests against the clip */
} GlyphEntry;

typedef struct
{
  cairo_surface_t *surface;
  uint32_t offset;
  int w, h;
} ImageEntry;

/* stroke masks are re-rasterised for every buffer (like the reference does),
 * but an identical mask is stored in the atlas only once */
typedef struct
{
  uint64_t hash;
  uint32_t offset, bytes;
} MaskEntry;

/* a piece of a pending outline path, with the CTM it is issued under */
typedef struct
{
  PangoFont *font;           /* glyph outline if set ... */
  PangoGlyphInfo gi;
  cairo_scaled_font_t *sfont; /* ... or an anamorphic glyph (cairo_glyph_path under
                               * cairo_scale (hscale, 1)) if set, else a rectangle */
  unsigned long sglyph;
  double hscale;
  double x, y, w, h;
  int npoly;                 /* ... or, if set, a closed polygon of npoly points (x, y, w, h: its box) */
  double *poly;              /* (its own allocation: the ops of a cue's outline are copied and
                              * sorted as a whole, and 192 bytes of points in every one of them --
                              * used by one glyph in a million -- more than doubled their size) */
  cairo_matrix_t m;
} PathOp;

struct StrokeJob;

/* one glyph of the pending run: its tight A8 bitmap in the atlas and where its
 * pixel (0, 0) lands in the frame */
typedef struct RunGlyph
{
  uint32_t offset;
  int w, h, ox, oy;
} RunGlyph;

/* a run seen before: its glyphs (bitmaps + relative positions) and, from the
 * second sighting on, ONE mask holding all of them */
typedef struct RunEntry
{
  uint64_t hash;
  uint8_t *key;
  size_t key_len;
  int state;                 /* 0 seen once; 1 merged; 2 cannot be merged */
  uint32_t offset;           /* atlas offset of the merged mask */
  int w, h;
} RunEntry;

typedef struct
{
  uint64_t hash;
  uint8_t *key;
  size_t key_len;
  struct StrokeJob *job;     /* being rasterised on a worker thread: offset, x, y, w, h
                              * are not known yet (see "deferred strokes" below) */
  int base_x, base_y;        /* integer base the key is relative to, raster coordinates */
  int translatable;          /* made clear of its limits: valid at any integer
                              * translation that stays clear of the limits there */
  int lim[4];                /* the limits it was made under (device box) */
  uint32_t offset;           /* atlas offset of the tight A8 mask */
  int x, y, w, h;            /* box of the mask at (base_x, base_y), raster coordinates */
  int ext[4];                /* extents of the stroke / fill there, unclipped */
} StrokeEntry;

struct FlutlRecorder
{
  FlutlRecorder *parent;      /* child recorders put their masks in the parent's atlas */
  int width, height;
  cairo_surface_t *scratch;   /* A8, sw x sh: the frame, or more (flutl_recorder_reserve) */
  int sw, sh;
  int line_split;             /* FLUTL_LINE_SPLIT=1: see path_to_mask (NOT exact) */
  cairo_t *cr;

  flutl_prim *prims;
  uint32_t n_prims, cap_prims;
  flutl_region *regions;
  uint32_t n_regions, cap_regions;
  uint32_t region_first;
  int in_region;

  uint8_t *atlas;
  size_t atlas_size, atlas_cap;
  uint32_t *new_off, *new_size;
  uint32_t n_new, cap_new;
  FlutlPolygonEntry **new_poly;      /* per new entry: NULL, or the polygon the DEVICE makes it of */
  int device_strokes;                /* outlines leave the workers as polygons (see stroke_job_run) */
  void *(*edge_acquire) (void *user, size_t bytes);    /* flutl_recorder_set_edge_blocks */
  void (*edge_release) (void *user, void *block);
  void *edge_user;

  GlyphEntry *glyphs;
  uint32_t glyph_cap, glyph_count;
  ImageEntry *images;
  uint32_t n_images, cap_images;
  MaskEntry *masks;
  uint32_t n_masks, cap_masks;
  PathOp *ops;
  uint32_t n_ops, cap_ops;
  StrokeEntry *strokes;
  uint32_t n_strokes, cap_strokes;
  /* by hash, open addressing: entry index + 1, 0 empty (a stream of new outlined text has
   * some 470 entries before its caches start over, and every new outline looked through all
   * of them: 2.6 us of each cue, a twentieth of its recording) */
  uint32_t *stroke_index;
  uint32_t stroke_index_cap;
  /* the keys of `strokes` name scaled fonts by address: every such font is
   * referenced here for as long as the cache lives, so that an address cannot
   * come back as a different font */
  cairo_scaled_font_t **key_fonts;
  uint32_t n_key_fonts, cap_key_fonts;
  /* outlines of single glyphs in 24.8 fixed point at origin 0 ("composed outline paths"
   * below): (scaled font, glyph) -> GlyphOutline; the fonts are pinned in key_fonts */
  GHashTable *outlines;
  int check_paths;              /* FLUTL_CHECK_PATHS=1: every composed path against Cairo's */
  int no_compose;               /* FLUTL_NO_COMPOSED_PATHS=1: every path built by Cairo (A/B) */

  /* glyph runs ("glyph runs" below): glyphs shown back to back in one colour under
   * one clip, held back until something else is drawn */
  struct RunGlyph *run;
  uint32_t n_run, cap_run;
  uint32_t run_pixel;
  uint16_t run_tag;             /* colour source of the run (0xFFFF: glyphs of several) */
  int run_clip[4];
  int in_run_flush, no_run_merge;
  struct RunEntry *runs;
  uint32_t n_runs, cap_runs;
  uint32_t *run_index;          /* by hash, as stroke_index (up to 4096 runs were walked per line of new text) */
  uint32_t run_index_cap;
  uint32_t runs_merged, runs_clashed;     /* statistics */

  /* deferred strokes (root recorder only): what the streaming thread and the
   * thread that resolves frames share -- atlas, mask table, stroke cache, the
   * log of new atlas entries -- is guarded by `lock` */
  GRecMutex lock;
  int defer;
  FlutlPatch *patches;
  uint32_t n_patches, cap_patches;
  int region_deferred;        /* the open region holds a placeholder */
  int region_provisional;     /* ... or a run at its first sighting: recorded again, it
                               * will be one primitive (the region cache waits for that) */

  int has_clip, clip[4];      /* x0,y0,x1,y1 */
  int in_group, group[4];
  int shift[2];                 /* frame = CTM coordinates + shift (integers) */
  int frame_clear, group_clear;

  double rgba[4];
  uint32_t pixel;             /* quantised premultiplied a8r8g8b8 */
  int color_opaque, color_clear;
  uint16_t color_tag;           /* who set the colour (element's numbering; 0: not said) */
  const cairo_matrix_t *ctm_inverse;   /* the reference's inverse CTM, if the caller keeps it */
  int ctm_given;                /* the next glyph's CTM is `ctm`, not the context's (flutl_recorder_set_ctm) */
  cairo_matrix_t ctm;
  uint16_t *prim_tags;          /* per primitive, parallel to prims */

  uint32_t unsupported;
};

static void path_ops_clear (FlutlRecorder *rec);
static void raster_origin (const FlutlRecorder *rec, int *ox, int *oy);
static void stroke_job_wait (struct StrokeJob *j);
static void stroke_job_unref (struct StrokeJob *j);
static int effective_clip (FlutlRecorder *rec, int c[4]);
static int mask_prim_place (flutl_prim *p, const StrokeEntry *e, int ox, int oy, const int c[4]);

/* ---- small helpers --------------------------------------------------------- */

static void *
xrealloc (void *p, size_t n)
{
  void *q = realloc (p, n);
  if (!q)
    abort ();
  return q;
}

/* The host copy of the atlas is an anonymous mapping: pages nothing was written to
 * cost nothing and read as zeros.  The space of an entry the device makes (an
 * outline's coverage, half a megabyte per cue at 4K) is only set aside there, and
 * starting over (drop_caches) gives the touched pages back instead of clearing a
 * quarter of a gigabyte. */
static uint8_t *
atlas_map_grow (uint8_t *old, size_t old_cap, size_t cap)
{
  void *p = old ? mremap (old, old_cap, cap, MREMAP_MAYMOVE)
                : mmap (NULL, cap, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) {
    fprintf (stderr, "flutl_recorder: out of memory (atlas of %zu bytes)\n", cap);
    abort ();
  }
  return p;
}

static void
atlas_ensure (FlutlRecorder *rec, size_t need)
{
  if (need <= rec->atlas_cap)
    return;
  size_t cap = rec->atlas_cap ? rec->atlas_cap : (size_t) 1 << 20;
  while (cap < need)
    cap *= 2;
  rec->atlas = atlas_map_grow (rec->atlas, rec->atlas_cap, cap);    /* (new pages: zeros) */
  rec->atlas_cap = cap;
}

static FlutlRecorder *
root_of (FlutlRecorder *rec)
{
  while (rec->parent)
    rec = rec->parent;
  return rec;
}

void
flutl_recorder_lock (FlutlRecorder *rec)
{
  g_rec_mutex_lock (&root_of (rec)->lock);
}

void
flutl_recorder_unlock (FlutlRecorder *rec)
{
  g_rec_mutex_unlock (&root_of (rec)->lock);
}

#define MASK_PAD 4              /* zero bytes in front of a mask and after each of its rows */
static uint32_t
atlas_put_ex (FlutlRecorder *rec, const void *src, size_t bytes, int on_device);

static uint32_t
atlas_put_locked (FlutlRecorder *rec, const void *src, size_t bytes, int on_device);

static uint32_t
atlas_put (FlutlRecorder *rec, const void *src, size_t bytes)
{
  return atlas_put_ex (rec, src, bytes, 0);
}

static uint32_t
atlas_put_ex (FlutlRecorder *rec, const void *src, size_t bytes, int on_device)
{
  if (rec->parent)
    return atlas_put_ex (rec->parent, src, bytes, on_device);
  g_rec_mutex_lock (&rec->lock);
  uint32_t r = atlas_put_locked (rec, src, bytes, on_device);
  g_rec_mutex_unlock (&rec->lock);
  return r;
}

static uint32_t
atlas_put_locked (FlutlRecorder *rec, const void *src, size_t bytes, int on_device)
{
  size_t off = (rec->atlas_size + ATLAS_ALIGN - 1) & ~(size_t) (ATLAS_ALIGN - 1);
  size_t end = off + bytes;
  atlas_ensure (rec, end + ATLAS_SLACK);
  /* (beyond atlas_size nothing has been written since the mapping was made or given
   * back: zeros.  src == NULL: an entry of zeros) */
  if (src)
    memcpy (rec->atlas + off, src, bytes);
  rec->atlas_size = end;
  if (on_device)
    return (uint32_t) off;      /* produced on the device: nothing to upload */
  if (rec->n_new == rec->cap_new) {
    rec->cap_new = rec->cap_new ? rec->cap_new * 2 : 64;
    rec->new_off = xrealloc (rec->new_off, rec->cap_new * sizeof (uint32_t));
    rec->new_size = xrealloc (rec->new_size, rec->cap_new * sizeof (uint32_t));
    rec->new_poly = xrealloc (rec->new_poly, rec->cap_new * sizeof (FlutlPolygonEntry *));
  }
  rec->new_off[rec->n_new] = (uint32_t) off;
  rec->new_size[rec->n_new] = (uint32_t) bytes;
  rec->new_poly[rec->n_new] = NULL;
  rec->n_new++;
  return (uint32_t) off;
}

/* An entry the DEVICE will make (flutl_atlas_put_polygon): its space is set aside
 * here (zeros in the host copy), and it takes its place in the log of new entries,
 * which whoever owns the device works off in order -- offsets are given out in the
 * same order on both sides.  Takes the polygon over.  (with the root's lock) */
static uint32_t
atlas_reserve_polygon (FlutlRecorder *rec, FlutlPolygonEntry *pe)
{
  rec = root_of (rec);
  const size_t bytes = MASK_PAD + ((size_t) pe->w + MASK_PAD) * pe->h;
  const uint32_t off = atlas_put_locked (rec, NULL, bytes, 0);
  rec->new_poly[rec->n_new - 1] = pe;
  return off;
}

static uint64_t
hash_bytes (const uint8_t *p, size_t n)
{
  uint64_t h = 0xcbf29ce484222325ull ^ (uint64_t) n;
  size_t i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t v;
    memcpy (&v, p + i, 8);
    h = (h ^ v) * 0x100000001b3ull;
    h ^= h >> 29;
  }
  for (; i < n; i++)
    h = (h ^ p[i]) * 0x100000001b3ull;
  return h;
}

static uint32_t atlas_put_dedup_hashed (FlutlRecorder *rec, const uint8_t *src, size_t bytes,
    uint64_t h);

static uint32_t
atlas_put_dedup (FlutlRecorder *rec, const uint8_t *src, size_t bytes)
{
  return atlas_put_dedup_hashed (rec, src, bytes, hash_bytes (src, bytes));
}

static uint32_t
atlas_put_dedup_hashed (FlutlRecorder *rec, const uint8_t *src, size_t bytes, uint64_t h)
{
  /* (a child recorder keeps its own table of masks, but the bytes live in the
   * parent's atlas, see atlas_put_ex) */
  FlutlRecorder *root = root_of (rec);
  g_rec_mutex_lock (&root->lock);
  for (uint32_t i = 0; i < rec->n_masks; i++)
    if (rec->masks[i].hash == h && rec->masks[i].bytes == bytes &&
        memcmp (root->atlas + rec->masks[i].offset, src, bytes) == 0) {
      uint32_t found = rec->masks[i].offset;
      g_rec_mutex_unlock (&root->lock);
      return found;
    }
  if (rec->n_masks == rec->cap_masks) {
    rec->cap_masks = rec->cap_masks ? rec->cap_masks * 2 : 64;
    rec->masks = xrealloc (rec->masks, rec->cap_masks * sizeof (MaskEntry));
  }
  uint32_t off = atlas_put (rec, src, bytes);
  rec->masks[rec->n_masks].hash = h;
  rec->masks[rec->n_masks].offset = off;
  rec->masks[rec->n_masks].bytes = (uint32_t) bytes;
  rec->n_masks++;
  g_rec_mutex_unlock (&root->lock);
  return off;
}

/* Store a tight w x h A8 bitmap the way the kernel wants to read it: 4 zero
 * bytes in front of the first row, every row followed by 4 zero bytes (stride
 * w + 4).  Returns the atlas offset of mask pixel (0,0). */
static uint8_t *
mask_padded (const uint8_t *tight, int w, int h, size_t *bytes_out)
{
  size_t stride = (size_t) w + MASK_PAD, bytes = MASK_PAD + stride * h;
  uint8_t *buf = calloc (1, bytes);
  if (!buf)
    abort ();
  for (int y = 0; y < h; y++)
    memcpy (buf + MASK_PAD + y * stride, tight + (size_t) y * w, (size_t) w);
  *bytes_out = bytes;
  return buf;
}

static uint32_t
atlas_put_mask (FlutlRecorder *rec, const uint8_t *tight, int w, int h, int dedup)
{
  size_t bytes;
  uint8_t *buf = mask_padded (tight, w, h, &bytes);
  uint32_t off = dedup ? atlas_put_dedup (rec, buf, bytes) : atlas_put (rec, buf, bytes);
  free (buf);
  return off + MASK_PAD;
}

static uint16_t
color_double_to_short (double d)
{
  return (uint16_t) (d * 65535.0 + 0.5);
}

static double
clamp01 (double v)
{
  return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
}

static int32_t
fixed_from_double (double d)
{
  /* _cairo_fixed_from_double: nearest 1/256, ties to even */
  return (int32_t) nearbyint (d * 256.0);
}

int32_t
flutl_recorder_fixed_from_double (double d)
{
  return fixed_from_double (d);
}

static int
imax (int a, int b)
{
  return a > b ? a : b;
}

static int
imin (int a, int b)
{
  return a < b ? a : b;
}

/* Intersect a device box with clip, group surface and frame.  Returns 0 if
 * nothing is left. */
static int
clip_box (FlutlRecorder *rec, int *x0, int *y0, int *x1, int *y1)
{
  *x0 = imax (*x0, 0);
  *y0 = imax (*y0, 0);
  *x1 = imin (*x1, rec->width);
  *y1 = imin (*y1, rec->height);
  if (rec->has_clip) {
    *x0 = imax (*x0, rec->clip[0]);
    *y0 = imax (*y0, rec->clip[1]);
    *x1 = imin (*x1, rec->clip[2]);
    *y1 = imin (*y1, rec->clip[3]);
  }
  if (rec->in_group) {
    *x0 = imax (*x0, rec->group[0]);
    *y0 = imax (*y0, rec->group[1]);
    *x1 = imin (*x1, rec->group[2]);
    *y1 = imin (*y1, rec->group[3]);
  }
  return *x1 > *x0 && *y1 > *y0;
}

/* Does a device box meet the surface being drawn on (frame, or the region's
 * temporary surface) and the clip?  Cairo returns NOTHING_TO_DO for an
 * operation whose extents do not -- BEFORE it marks the surface as no longer
 * clear (cairo-surface.c: is_clear = FALSE only if status != NOTHING_TO_DO),
 * and "clear" selects the arithmetic of the next antialiased operation
 * (found by tools/fuzz_ttml.py: a line of text clipped away above the region,
 * then a background box with fractional edges). */
static int
box_hits_target (const FlutlRecorder *rec, int x0, int y0, int x1, int y1)
{
  if (x1 <= x0 || y1 <= y0)
    return 0;
  if (rec->in_group) {
    x0 = imax (x0, rec->group[0]);
    y0 = imax (y0, rec->group[1]);
    x1 = imin (x1, rec->group[2]);
    y1 = imin (y1, rec->group[3]);
  } else {
    x0 = imax (x0, 0);
    y0 = imax (y0, 0);
    x1 = imin (x1, rec->width);
    y1 = imin (y1, rec->height);
  }
  if (rec->has_clip) {
    x0 = imax (x0, rec->clip[0]);
    y0 = imax (y0, rec->clip[1]);
    x1 = imin (x1, rec->clip[2]);
    y1 = imin (y1, rec->clip[3]);
  }
  return x1 > x0 && y1 > y0;
}

static void run_flush (FlutlRecorder *rec);

static flutl_prim *
push_prim (FlutlRecorder *rec)
{
  if (rec->n_run && !rec->in_run_flush)
    run_flush (rec);            /* paint order: the held-back glyphs come first */
  if (rec->n_prims == rec->cap_prims) {
    rec->cap_prims = rec->cap_prims ? rec->cap_prims * 2 : 256;
    rec->prims = xrealloc (rec->prims, rec->cap_prims * sizeof (flutl_prim));
    rec->prim_tags = xrealloc (rec->prim_tags, rec->cap_prims * sizeof (uint16_t));
  }
  rec->prim_tags[rec->n_prims] = rec->in_run_flush ? rec->run_tag : rec->color_tag;
  flutl_prim *p = &rec->prims[rec->n_prims++];
  memset (p, 0, sizeof *p);
  return p;
}

static int
target_is_clear (FlutlRecorder *rec)
{
  return rec->in_group ? rec->group_clear : rec->frame_clear;
}

static void
target_touched (FlutlRecorder *rec)
{
  if (rec->in_group)
    rec->group_clear = 0;
  else
    rec->frame_clear = 0;
}

/* ---- life cycle ------------------------------------------------------------ */

FlutlRecorder *
flutl_recorder_new (void)
{
  FlutlRecorder *rec = calloc (1, sizeof *rec);
  if (!rec)
    abort ();
  rec->atlas_size = ATLAS_SLACK;
  rec->line_split = g_getenv ("FLUTL_LINE_SPLIT") != NULL;
  rec->check_paths = g_getenv ("FLUTL_CHECK_PATHS") != NULL;
  rec->no_compose = g_getenv ("FLUTL_NO_COMPOSED_PATHS") != NULL;
  rec->no_run_merge = g_getenv ("FLUTL_NO_RUN_MERGE") != NULL;
  g_rec_mutex_init (&rec->lock);
  return rec;
}

FlutlRecorder *
flutl_recorder_new_child (FlutlRecorder *parent)
{
  FlutlRecorder *rec = flutl_recorder_new ();
  rec->parent = parent;
  return rec;
}

uint32_t
flutl_recorder_next_atlas_offset (FlutlRecorder *rec)
{
  while (rec->parent)
    rec = rec->parent;
  return (uint32_t) ((rec->atlas_size + ATLAS_ALIGN - 1) & ~(size_t) (ATLAS_ALIGN - 1));
}

uint32_t
flutl_recorder_atlas_append_device_entry (FlutlRecorder *rec, const void *data,
    size_t bytes)
{
  return atlas_put_ex (rec, data, bytes, 1);
}

void
flutl_recorder_paint_atlas_image (FlutlRecorder *rec, uint32_t offset, int w, int h,
    int x, int y)
{
  target_touched (rec);
  x += rec->shift[0];           /* (the position came from the CTM, see set_device_shift) */
  y += rec->shift[1];
  int x0 = x, y0 = y, x1 = x + w, y1 = y + h;
  if (!clip_box (rec, &x0, &y0, &x1, &y1))
    return;
  flutl_prim *p = push_prim (rec);
  p->op = FLUTL_OP_IMAGE;
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  p->a = (int32_t) offset;
  p->b = w * 4;
  p->c = x;
  p->d = y;
}

static void
drop_caches (FlutlRecorder *rec)
{
  for (uint32_t i = 0; i < rec->glyph_cap; i++)
    if (rec->glyphs[i].used)
      cairo_scaled_font_destroy (rec->glyphs[i].font);
  free (rec->glyphs);
  rec->glyphs = NULL;
  rec->glyph_cap = rec->glyph_count = 0;
  for (uint32_t i = 0; i < rec->n_images; i++)
    cairo_surface_destroy (rec->images[i].surface);
  rec->n_images = 0;
  rec->n_masks = 0;
  for (uint32_t i = 0; i < rec->n_strokes; i++) {
    free (rec->strokes[i].key);
    if (rec->strokes[i].job) {      /* (the caller has resolved every frame: it is done) */
      stroke_job_wait (rec->strokes[i].job);
      stroke_job_unref (rec->strokes[i].job);
    }
  }
  rec->n_strokes = 0;
  if (rec->stroke_index)
    memset (rec->stroke_index, 0, rec->stroke_index_cap * sizeof *rec->stroke_index);
  if (rec->outlines)            /* (keyed by the addresses of the fonts released next) */
    g_hash_table_remove_all (rec->outlines);
  for (uint32_t i = 0; i < rec->n_key_fonts; i++)
    cairo_scaled_font_destroy (rec->key_fonts[i]);
  rec->n_key_fonts = 0;
  for (uint32_t i = 0; i < rec->n_runs; i++)
    free (rec->runs[i].key);
  rec->n_runs = 0;
  if (rec->run_index)
    memset (rec->run_index, 0, rec->run_index_cap * sizeof *rec->run_index);
  rec->n_run = 0;
  rec->atlas_size = ATLAS_SLACK;
  if (rec->atlas && madvise (rec->atlas, rec->atlas_cap, MADV_DONTNEED) != 0)
    memset (rec->atlas, 0, rec->atlas_cap);
  for (uint32_t i = 0; i < rec->n_new; i++)
    flutl_polygon_entry_free (rec->new_poly[i]);
  rec->n_new = 0;
}

void
flutl_recorder_reset (FlutlRecorder *rec)
{
  drop_caches (rec);
}

void
flutl_recorder_free (FlutlRecorder *rec)
{
  if (!rec)
    return;
  drop_caches (rec);
  if (rec->cr)
    cairo_destroy (rec->cr);
  if (rec->scratch)
    cairo_surface_destroy (rec->scratch);
  free (rec->prims);
  free (rec->prim_tags);
  free (rec->regions);
  if (rec->atlas)
    munmap (rec->atlas, rec->atlas_cap);
  for (uint32_t i = 0; i < rec->n_new; i++)
    flutl_polygon_entry_free (rec->new_poly ? rec->new_poly[i] : NULL);
  free (rec->new_off);
  free (rec->new_size);
  free (rec->new_poly);
  free (rec->images);
  free (rec->masks);
  path_ops_clear (rec);
  free (rec->ops);
  free (rec->strokes);
  free (rec->stroke_index);
  if (rec->outlines)
    g_hash_table_destroy (rec->outlines);
  free (rec->key_fonts);
  free (rec->patches);
  free (rec->run);
  free (rec->runs);
  free (rec->run_index);
  g_rec_mutex_clear (&rec->lock);
  free (rec);
}

void
flutl_recorder_begin_frame (FlutlRecorder *rec, int width, int height)
{
  if (!rec->scratch || rec->width != width || rec->height != height) {
    if (rec->cr)
      cairo_destroy (rec->cr);
    if (rec->scratch)
      cairo_surface_destroy (rec->scratch);
    rec->scratch = cairo_image_surface_create (CAIRO_FORMAT_A8, width, height);
    rec->cr = NULL;
    rec->width = rec->sw = width;
    rec->height = rec->sh = height;
  }
  if (rec->cr)
    cairo_destroy (rec->cr);
  rec->cr = cairo_create (rec->scratch);
  cairo_set_source_rgba (rec->cr, 1, 1, 1, 1);
  rec->n_prims = 0;
  rec->n_regions = 0;
  rec->in_region = 0;
  rec->has_clip = 0;
  rec->in_group = 0;
  rec->frame_clear = 1;
  rec->group_clear = 1;
  rec->unsupported = 0;
  rec->n_patches = 0;
  rec->region_deferred = 0;
  rec->n_run = 0;
  path_ops_clear (rec);
  flutl_recorder_set_color (rec, 0, 0, 0, 1);
}

/* Paths are rasterised on the scratch surface under the bounds the reference
 * draws them under -- and the surface itself is one more bound.  A region with
 * an opacity draws on a temporary surface of ITS size, which may be larger than
 * the frame: the scratch surface must not end before it does (found by
 * tools/fuzz_ttml.py: a glyph outline that crosses x = frame width inside a
 * wider region came out 1/255 different in one pixel).  To be called after
 * flutl_recorder_begin_frame and before anything is drawn: the tracking context
 * is made anew. */
void
flutl_recorder_reserve (FlutlRecorder *rec, int w, int h)
{
  if (w <= rec->sw && h <= rec->sh)
    return;
  rec->sw = imax (rec->sw, w);
  rec->sh = imax (rec->sh, h);
  cairo_destroy (rec->cr);
  cairo_surface_destroy (rec->scratch);
  rec->scratch = cairo_image_surface_create (CAIRO_FORMAT_A8, rec->sw, rec->sh);
  rec->cr = cairo_create (rec->scratch);
  cairo_set_source_rgba (rec->cr, 1, 1, 1, 1);
}

cairo_t *
flutl_recorder_cairo (FlutlRecorder *rec)
{
  return rec->cr;
}

/* ---- regions, groups, clip, colour ----------------------------------------- */

void
flutl_recorder_region_begin (FlutlRecorder *rec)
{
  rec->region_first = rec->n_prims;
  rec->in_region = 1;
  rec->region_deferred = 0;
  rec->region_provisional = 0;
  rec->shift[0] = rec->shift[1] = 0;
}

/* From here to the end of the region, what the CTM of the tracking context
 * maps to lies (dx, dy) whole pixels from its place in the frame.  The
 * reference's vertical writing modes inside an opacity group need this: there
 * the CTM is the bare quarter turn on the region's temporary surface, whose
 * cos (pi/2) = 6e-17 leaves device positions a few ulp off the halves they
 * would be exactly -- so "CTM + origin, then round" and "round, then + origin"
 * place a glyph one pixel apart.  The shift is added after rounding, where the
 * reference's paint of the temporary surface adds it. */
void
flutl_recorder_set_device_shift (FlutlRecorder *rec, int dx, int dy)
{
  rec->shift[0] = dx;
  rec->shift[1] = dy;
}

void
flutl_recorder_region_end (FlutlRecorder *rec)
{
  run_flush (rec);
  uint32_t n = rec->n_prims - rec->region_first;
  rec->in_region = 0;
  if (n == 0)
    return;
  int x0 = 32767, y0 = 32767, x1 = 0, y1 = 0;
  for (uint32_t i = rec->region_first; i < rec->n_prims; i++) {
    const flutl_prim *p = &rec->prims[i];
    if (p->cx1 <= p->cx0 || p->cy1 <= p->cy0)
      continue;
    x0 = imin (x0, p->cx0);
    y0 = imin (y0, p->cy0);
    x1 = imax (x1, p->cx1);
    y1 = imax (y1, p->cy1);
  }
  if ((x1 <= x0 || y1 <= y0) && rec->region_deferred) {
    /* nothing visible YET: a placeholder's mask is still being made;
     * flutl_recorder_resolve gives the region its box */
    x0 = y0 = x1 = y1 = 0;
  } else if (x1 <= x0 || y1 <= y0) {   /* nothing visible */
    rec->n_prims = rec->region_first;
    return;
  }
  /* empty-bbox prims (group markers of an off-frame region) must still lie
   * inside the region box */
  for (uint32_t i = rec->region_first; i < rec->n_prims; i++) {
    flutl_prim *p = &rec->prims[i];
    if (p->cx1 <= p->cx0 || p->cy1 <= p->cy0) {
      p->cx0 = p->cx1 = (int16_t) x0;
      p->cy0 = p->cy1 = (int16_t) y0;
    }
  }
  if (rec->n_regions == rec->cap_regions) {
    rec->cap_regions = rec->cap_regions ? rec->cap_regions * 2 : 32;
    rec->regions = xrealloc (rec->regions, rec->cap_regions * sizeof (flutl_region));
  }
  flutl_region *r = &rec->regions[rec->n_regions++];
  r->x0 = (int16_t) x0;
  r->y0 = (int16_t) y0;
  r->x1 = (int16_t) x1;
  r->y1 = (int16_t) y1;
  r->first = rec->region_first;
  r->count = n;
}

void
flutl_recorder_group_begin (FlutlRecorder *rec, int x, int y, int w, int h)
{
  rec->group[0] = x;
  rec->group[1] = y;
  rec->group[2] = x + imax (w, 0);
  rec->group[3] = y + imax (h, 0);
  int x0 = x, y0 = y, x1 = rec->group[2], y1 = rec->group[3];
  /* the group marker itself is not subject to the region clip */
  int saved = rec->has_clip;
  rec->has_clip = 0;
  rec->in_group = 0;
  if (!clip_box (rec, &x0, &y0, &x1, &y1))
    x0 = y0 = x1 = y1 = 0;
  rec->has_clip = saved;
  flutl_prim *p = push_prim (rec);
  p->op = FLUTL_OP_GROUP_BEGIN;
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  rec->in_group = 1;
  rec->group_clear = 1;
}

void
flutl_recorder_group_end (FlutlRecorder *rec, double opacity)
{
  int x0 = rec->group[0], y0 = rec->group[1], x1 = rec->group[2], y1 = rec->group[3];
  int saved = rec->has_clip;
  rec->has_clip = 0;
  rec->in_group = 0;
  if (!clip_box (rec, &x0, &y0, &x1, &y1))
    x0 = y0 = x1 = y1 = 0;
  rec->has_clip = saved;
  flutl_prim *p = push_prim (rec);
  p->op = FLUTL_OP_GROUP_END;
  /* solid mask pattern of cairo_paint_with_alpha: alpha_short >> 8 */
  p->param = (uint16_t) (color_double_to_short (clamp01 (opacity)) >> 8);
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  /* cairo_paint_with_alpha returns before touching the surface if the alpha
   * is "zero" (CAIRO_ALPHA_IS_ZERO: <= 0x00ff / 0xffff): the frame then still
   * counts as clear for the next antialiased operation (found by
   * tools/fuzz_ttml.py); the composite itself is a no-op either way
   * -- and so does the paint of a temporary surface that nothing was drawn
   * on (cairo-surface.c nothing_to_do: OVER with a clear source) */
  if (!(opacity <= (double) 0x00ff / (double) 0xffff) && !rec->group_clear)
    rec->frame_clear = 0;
}

void
flutl_recorder_clip (FlutlRecorder *rec, int x, int y, int w, int h)
{
  run_flush (rec);
  int c[4] = { x, y, x + imax (w, 0), y + imax (h, 0) };
  if (rec->has_clip) {
    c[0] = imax (c[0], rec->clip[0]);
    c[1] = imax (c[1], rec->clip[1]);
    c[2] = imin (c[2], rec->clip[2]);
    c[3] = imin (c[3], rec->clip[3]);
  }
  memcpy (rec->clip, c, sizeof c);
  rec->has_clip = 1;
}

void
flutl_recorder_clip_reset (FlutlRecorder *rec)
{
  run_flush (rec);
  rec->has_clip = 0;
}

void
flutl_recorder_set_color (FlutlRecorder *rec, double r, double g, double b, double a)
{
  r = clamp01 (r);
  g = clamp01 (g);
  b = clamp01 (b);
  a = clamp01 (a);
  rec->rgba[0] = r;
  rec->rgba[1] = g;
  rec->rgba[2] = b;
  rec->rgba[3] = a;
  uint16_t as = color_double_to_short (a);
  uint16_t rs = color_double_to_short (r * a);
  uint16_t gs = color_double_to_short (g * a);
  uint16_t bs = color_double_to_short (b * a);
  rec->pixel = ((uint32_t) (as >> 8) << 24) | ((uint32_t) (rs >> 8) << 16) |
      ((uint32_t) (gs >> 8) << 8) | (uint32_t) (bs >> 8);
  rec->color_opaque = as >= 0xff00;   /* CAIRO_COLOR_IS_OPAQUE */
  rec->color_clear = as <= 0x00ff;    /* CAIRO_COLOR_IS_CLEAR  */
  rec->color_tag = 0;
}

/* The pixel a colour becomes (what set_color leaves in the primitives). */
uint32_t
flutl_recorder_pixel_of (double r, double g, double b, double a)
{
  r = clamp01 (r);
  g = clamp01 (g);
  b = clamp01 (b);
  a = clamp01 (a);
  uint16_t as = color_double_to_short (a);
  uint16_t rs = color_double_to_short (r * a);
  uint16_t gs = color_double_to_short (g * a);
  uint16_t bs = color_double_to_short (b * a);
  return ((uint32_t) (as >> 8) << 24) | ((uint32_t) (rs >> 8) << 16) |
      ((uint32_t) (gs >> 8) << 8) | (uint32_t) (bs >> 8);
}

/* Colour tags: the element names the source of the colour it has just set (a
 * colour attribute of a layout); every primitive remembers the tag it was drawn
 * under.  A region whose layouts change in nothing but such colours is then the
 * same primitives with other colour words (the element's region cache).
 * set_color clears the tag. */
/* The caller moves the CTM about with cairo_set_matrix (walk_layout) where the
 * reference uses translate / transform calls: the context's own inverse is then
 * an inversion, not the reference's product of inverses.  Whoever needs the
 * reference's (hexbox_show) takes it from here while it is set. */
void
flutl_recorder_set_ctm_inverse (FlutlRecorder *rec, const cairo_matrix_t *inverse)
{
  rec->ctm_inverse = inverse;
}

void
flutl_recorder_set_color_tag (FlutlRecorder *rec, uint16_t tag)
{
  rec->color_tag = tag;
}

uint16_t
flutl_recorder_get_color_tag (FlutlRecorder *rec)
{
  return rec->color_tag;
}

const uint16_t *
flutl_recorder_prim_tags (FlutlRecorder *rec)
{
  run_flush (rec);
  return rec->prim_tags;
}

void
flutl_recorder_get_color (FlutlRecorder *rec, double rgba[4])
{
  memcpy (rgba, rec->rgba, sizeof rec->rgba);
}

/* ---- rectangles ------------------------------------------------------------ */

static int
ctm_is_translation (cairo_t *cr, cairo_matrix_t *m)
{
  cairo_get_matrix (cr, m);
  return m->xx == 1.0 && m->yy == 1.0 && m->xy == 0.0 && m->yx == 0.0;
}

void
flutl_recorder_fill_rect (FlutlRecorder *rec, double x, double y, double w, double h)
{
  cairo_matrix_t m;
  cairo_get_matrix (rec->cr, &m);
  if (rec->color_clear)
    return;               /* OVER with a clear source is a no-op in cairo */
  /* cairo_rectangle: move_to (x, y), then three relative segments; the start
   * point and every segment vector are mapped through the CTM and converted to
   * 24.8 fixed point on their own (cairo-path-fixed.c) */
  double px = x, py = y;
  cairo_matrix_transform_point (&m, &px, &py);
  double ax = w, ay = 0, bx = 0, by = h;
  cairo_matrix_transform_distance (&m, &ax, &ay);
  cairo_matrix_transform_distance (&m, &bx, &by);
  int32_t p0x = fixed_from_double (px), p0y = fixed_from_double (py);
  int32_t p1x = p0x + fixed_from_double (ax), p1y = p0y + fixed_from_double (ay);
  int32_t p2x = p1x + fixed_from_double (bx), p2y = p1y + fixed_from_double (by);
  int32_t p3x = p2x + fixed_from_double (-ax), p3y = p2y + fixed_from_double (-ay);
  /* a box in device space?  (translations, and the 90 degree rotation of the
   * vertical writing modes, whose cos (pi/2) = 6e-17 vanishes in fixed point) */
  int horiz_first = p0y == p1y && p1x == p2x && p2y == p3y && p3x == p0x;
  int vert_first = p0x == p1x && p1y == p2y && p2x == p3x && p3y == p0y;
  if (!horiz_first && !vert_first) {
    rec->unsupported++;   /* general polygon fill: not a rectangle on the device */
    return;
  }
  int32_t fx1 = p0x < p2x ? p0x : p2x, fx2 = p0x < p2x ? p2x : p0x;
  int32_t fy1 = p0y < p2y ? p0y : p2y, fy2 = p0y < p2y ? p2y : p0y;
  fx1 += rec->shift[0] * 256;
  fx2 += rec->shift[0] * 256;
  fy1 += rec->shift[1] * 256;
  fy2 += rec->shift[1] * 256;
  if (fx1 == fx2 || fy1 == fy2)
    return;
  int x0 = fx1 >> 8, y0 = fy1 >> 8, x1 = (fx2 + 255) >> 8, y1 = (fy2 + 255) >> 8;
  if (!clip_box (rec, &x0, &y0, &x1, &y1))
    return;
  int aligned = ((fx1 | fy1 | fx2 | fy2) & 255) == 0;
  flutl_prim *p = push_prim (rec);
  p->op = FLUTL_OP_RECT;
  p->rule = (!aligned && (rec->color_opaque || target_is_clear (rec)))
      ? FLUTL_RULE_LERP : FLUTL_RULE_PIXMAN;
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  p->color = rec->pixel;
  p->a = fx1;
  p->b = fy1;
  p->c = fx2;
  p->d = fy2;
  target_touched (rec);
}

/* ---- glyphs ---------------------------------------------------------------- */

static uint32_t
glyph_hash (const void *font, uint32_t glyph)
{
  uint64_t h = (uint64_t) (uintptr_t) font * 0x9E3779B97F4A7C15ull;
  h ^= (uint64_t) glyph * 0xC2B2AE3D27D4EB4Full;
  return (uint32_t) (h >> 32);
}

static GlyphEntry *glyph_lookup (FlutlRecorder *rec, cairo_scaled_font_t *sf,
    uint32_t glyph, const cairo_matrix_t *m);

static void
glyph_grow (FlutlRecorder *rec)
{
  uint32_t old_cap = rec->glyph_cap;
  GlyphEntry *old = rec->glyphs;
  rec->glyph_cap = old_cap ? old_cap * 2 : 256;
  rec->glyphs = calloc (rec->glyph_cap, sizeof (GlyphEntry));
  if (!rec->glyphs)
    abort ();
  for (uint32_t i = 0; i < old_cap; i++) {
    if (!old[i].used)
      continue;
    uint32_t j = glyph_hash (old[i].font, old[i].glyph) & (rec->glyph_cap - 1);
    while (rec->glyphs[j].used)
      j = (j + 1) & (rec->glyph_cap - 1);
    rec->glyphs[j] = old[i];
  }
  free (old);
}

/* Rasterise one glyph exactly as the image backend will see it: through the
 * scaled font's own glyph cache, at an integer origin, onto a cleared A8
 * surface with an opaque white source (result byte == mask byte). */
static void
glyph_rasterise (FlutlRecorder *rec, cairo_scaled_font_t *sf, GlyphEntry *e)
{
  cairo_glyph_t g = { e->glyph, 0, 0 };
  cairo_text_extents_t ext;
  cairo_scaled_font_glyph_extents (sf, &g, 1, &ext);
  e->w = e->h = 0;
  memset (e->mbox, 0, sizeof e->mbox);
  if (!(ext.width > 0) || !(ext.height > 0))
    return;
  /* device-space box of the glyph under the linear part of the CTM (identity
   * for plain text; a shear for reverseOblique; a rotation for vertical
   * writing modes): Cairo makes a scaled font for that CTM and FreeType
   * transforms the outline before rasterising */
  cairo_matrix_t lin;
  cairo_matrix_init (&lin, e->lin[0], e->lin[1], e->lin[2], e->lin[3], 0, 0);
  double bx0 = 1e30, by0 = 1e30, bx1 = -1e30, by1 = -1e30;
  for (int c = 0; c < 4; c++) {
    double px = ext.x_bearing + ((c & 1) ? ext.width : 0);
    double py = ext.y_bearing + ((c & 2) ? ext.height : 0);
    cairo_matrix_transform_point (&lin, &px, &py);
    if (px < bx0) bx0 = px;
    if (px > bx1) bx1 = px;
    if (py < by0) by0 = py;
    if (py > by1) by1 = py;
  }
  int x0 = (int) floor (bx0) - 3, y0 = (int) floor (by0) - 3;
  int w = (int) ceil (bx1 - bx0) + 8, h = (int) ceil (by1 - by0) + 8;
  e->mbox[0] = (int) floor (bx0);
  e->mbox[1] = (int) floor (by0);
  e->mbox[2] = (int) ceil (bx1);
  e->mbox[3] = (int) ceil (by1);
  cairo_surface_t *s = cairo_image_surface_create (CAIRO_FORMAT_A8, w, h);
  cairo_t *cr = cairo_create (s);
  lin.x0 = -x0;                 /* glyph origin at the integer pixel (-x0, -y0) */
  lin.y0 = -y0;
  cairo_set_matrix (cr, &lin);
  cairo_set_scaled_font (cr, sf);
  cairo_set_source_rgba (cr, 1, 1, 1, 1);
  cairo_show_glyphs (cr, &g, 1);
  cairo_destroy (cr);
  cairo_surface_flush (s);
  const uint8_t *data = cairo_image_surface_get_data (s);
  int stride = cairo_image_surface_get_stride (s);
  int tx0 = w, ty0 = h, tx1 = -1, ty1 = -1;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++)
      if (data[y * stride + x]) {
        if (x < tx0) tx0 = x;
        if (x > tx1) tx1 = x;
        if (y < ty0) ty0 = y;
        if (y > ty1) ty1 = y;
      }
  if (tx1 >= tx0) {
    int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
    uint8_t *tight = malloc ((size_t) tw * th);
    if (!tight)
      abort ();
    for (int y = 0; y < th; y++)
      memcpy (tight + (size_t) y * tw, data + (size_t) (ty0 + y) * stride + tx0, tw);
    e->offset = atlas_put_mask (rec, tight, tw, th, 0);
    free (tight);
    e->w = tw;
    e->h = th;
    e->bx = x0 + tx0;
    e->by = y0 + ty0;
  }
  cairo_surface_destroy (s);
}

static GlyphEntry *
glyph_lookup (FlutlRecorder *rec, cairo_scaled_font_t *sf, uint32_t glyph,
    const cairo_matrix_t *m)
{
  if (rec->glyph_count * 2 >= rec->glyph_cap)
    glyph_grow (rec);
  uint32_t j = glyph_hash (sf, glyph) & (rec->glyph_cap - 1);
  while (rec->glyphs[j].used) {
    GlyphEntry *c = &rec->glyphs[j];
    if (c->font == sf && c->glyph == glyph && c->lin[0] == m->xx && c->lin[1] == m->yx &&
        c->lin[2] == m->xy && c->lin[3] == m->yy)
      return c;
    j = (j + 1) & (rec->glyph_cap - 1);
  }
  GlyphEntry *e = &rec->glyphs[j];
  e->used = 1;
  e->font = cairo_scaled_font_reference (sf);
  e->glyph = glyph;
  e->lin[0] = m->xx;
  e->lin[1] = m->yx;
  e->lin[2] = m->xy;
  e->lin[3] = m->yy;
  rec->glyph_count++;
  glyph_rasterise (rec, sf, e);
  return e;
}

/* One glyph of a scaled font at the user-space point (ux, uy): what
 * cairo_show_glyphs / cairo_show_text do on the image backend -- the point goes
 * through the CTM (a plain addition for a translation) and is rounded half up
 * (_cairo_lround); the mask comes from the scaled font's glyph cache. */
static void
show_scaled_glyph (FlutlRecorder *rec, cairo_scaled_font_t *sf, uint32_t glyph, double cx,
    double cy)
{
  cairo_matrix_t m;
  if (rec->ctm_given)
    m = rec->ctm;
  else
    cairo_get_matrix (rec->cr, &m);
  if (m.xx == 1.0 && m.yy == 1.0 && m.xy == 0.0 && m.yx == 0.0) {
    cx += m.x0;
    cy += m.y0;
  } else {
    cairo_matrix_transform_point (&m, &cx, &cy);
  }
  int ix = (int) floor (cx + 0.5) + rec->shift[0];
  int iy = (int) floor (cy + 0.5) + rec->shift[1];
  GlyphEntry *e = glyph_lookup (rec, sf, glyph, &m);
  if (box_hits_target (rec, ix + e->mbox[0], iy + e->mbox[1], ix + e->mbox[2],
          iy + e->mbox[3]))
    target_touched (rec);
  if (e->w == 0)
    return;
  int ox = ix + e->bx, oy = iy + e->by;
  int c[4];
  if (!effective_clip (rec, c) || ox >= c[2] || oy >= c[3] || ox + e->w <= c[0] ||
      oy + e->h <= c[1])
    return;
  /* held back: glyphs that follow in the same colour under the same clip form a
   * run (run_flush) */
  if (rec->n_run && (rec->run_pixel != rec->pixel || memcmp (rec->run_clip, c, sizeof c) != 0))
    run_flush (rec);
  if (rec->n_run == rec->cap_run) {
    rec->cap_run = rec->cap_run ? rec->cap_run * 2 : 64;
    rec->run = xrealloc (rec->run, rec->cap_run * sizeof (RunGlyph));
  }
  if (rec->n_run == 0)
    rec->run_tag = rec->color_tag;
  else if (rec->run_tag != rec->color_tag)
    rec->run_tag = 0xFFFF;      /* one colour from two sources */
  rec->run_pixel = rec->pixel;
  memcpy (rec->run_clip, c, sizeof c);
  RunGlyph *g = &rec->run[rec->n_run++];
  g->offset = e->offset;
  g->w = e->w;
  g->h = e->h;
  g->ox = ox;
  g->oy = oy;
}

/* ---- glyph runs -----------------------------------------------------------------
 * The reference shows one glyph at a time (gstttmlrender.c:1138-1161) and so did
 * the display list: one MASK primitive per glyph.  On the device a primitive that
 * hits a tile costs ~250 instructions of bookkeeping before any pixel is blended
 * (the wave timeline), so a tile of text was dominated by it.  A
 * run of glyphs in one colour under one clip whose coverages never meet on a
 * pixel composites exactly like ONE mask holding all of them: a glyph leaves
 * every pixel it covers with 0 untouched (pixman: s * 0 + d * 255 / 255 == d).
 * Runs are recognised by their glyphs and relative positions; the first time a
 * run is seen its glyphs are emitted one by one as before (a cue shown once pays
 * nothing), the second time its bitmaps are pasted into one mask -- unless two of
 * them do meet on a pixel, then it stays as it is -- and from then on the run is
 * one primitive. */
static void
run_emit_glyphs (FlutlRecorder *rec)
{
  const int *c = rec->run_clip;
  for (uint32_t i = 0; i < rec->n_run; i++) {
    const RunGlyph *g = &rec->run[i];
    int x0 = imax (g->ox, c[0]), y0 = imax (g->oy, c[1]);
    int x1 = imin (g->ox + g->w, c[2]), y1 = imin (g->oy + g->h, c[3]);
    flutl_prim *p = push_prim (rec);
    p->op = FLUTL_OP_MASK;
    p->rule = FLUTL_RULE_PIXMAN;
    p->cx0 = (int16_t) x0;
    p->cy0 = (int16_t) y0;
    p->cx1 = (int16_t) x1;
    p->cy1 = (int16_t) y1;
    p->color = rec->run_pixel;
    p->param = (x0 > g->ox || x1 < g->ox + g->w) ? FLUTL_MASK_XCLIP : 0;
    p->a = (int32_t) g->offset;
    p->b = g->w + MASK_PAD;
    p->c = g->ox;
    p->d = g->oy;
  }
}

static void
run_flush (FlutlRecorder *rec)
{
  if (rec->n_run == 0 || rec->in_run_flush)
    return;
  rec->in_run_flush = 1;
  const uint32_t n = rec->n_run;
  if (n < 2 || rec->no_run_merge || rec->parent) {
    run_emit_glyphs (rec);
    rec->n_run = 0;
    rec->in_run_flush = 0;
    return;
  }
  /* the run's identity: its bitmaps and where they sit relative to its corner */
  int x0 = rec->run[0].ox, y0 = rec->run[0].oy, x1 = x0, y1 = y0;
  for (uint32_t i = 0; i < n; i++) {
    const RunGlyph *g = &rec->run[i];
    x0 = imin (x0, g->ox);
    y0 = imin (y0, g->oy);
    x1 = imax (x1, g->ox + g->w);
    y1 = imax (y1, g->oy + g->h);
  }
  const size_t key_len = (size_t) n * 5 * sizeof (int32_t);
  int32_t *key = malloc (key_len);
  if (!key)
    abort ();
  for (uint32_t i = 0; i < n; i++) {
    const RunGlyph *g = &rec->run[i];
    key[5 * i] = (int32_t) g->offset;
    key[5 * i + 1] = g->w;
    key[5 * i + 2] = g->h;
    key[5 * i + 3] = g->ox - x0;
    key[5 * i + 4] = g->oy - y0;
  }
  const uint64_t h = hash_bytes ((const uint8_t *) key, key_len);
  g_rec_mutex_lock (&rec->lock);      /* (the atlas is shared with the resolving thread) */
  RunEntry *e = NULL;
  for (uint32_t pos = (uint32_t) h; rec->run_index_cap; pos++) {
    const uint32_t v = rec->run_index[pos & (rec->run_index_cap - 1)];
    if (!v)
      break;
    if (rec->runs[v - 1].hash == h && rec->runs[v - 1].key_len == key_len &&
        memcmp (rec->runs[v - 1].key, key, key_len) == 0) {
      e = &rec->runs[v - 1];
      break;
    }
  }
  if (!e) {                     /* first sighting: remember it, emit as before */
    /* (a live stream without outlines never fills the atlas, so nothing else ever
     * clears this table: bound it -- a run forgotten is a run seen for the first
     * time again, its mask, if it has one, stays where the lists point) */
    if (rec->n_runs >= 4096) {
      for (uint32_t i = 0; i < rec->n_runs; i++)
        free (rec->runs[i].key);
      rec->n_runs = 0;
      if (rec->run_index)
        memset (rec->run_index, 0, rec->run_index_cap * sizeof *rec->run_index);
    }
    if (rec->n_runs == rec->cap_runs) {
      rec->cap_runs = rec->cap_runs ? rec->cap_runs * 2 : 32;
      rec->runs = xrealloc (rec->runs, rec->cap_runs * sizeof (RunEntry));
    }
    e = &rec->runs[rec->n_runs++];
    memset (e, 0, sizeof *e);
    e->hash = h;
    e->key = (uint8_t *) key;
    e->key_len = key_len;
    key = NULL;
    rec->region_provisional = 1;
    if (!rec->run_index) {      /* (8192 places for at most 4096 runs: never more than half full) */
      rec->run_index_cap = 8192;
      rec->run_index = calloc (rec->run_index_cap, sizeof *rec->run_index);
      if (!rec->run_index)
        abort ();
    }
    {
      uint32_t pos = (uint32_t) h;
      while (rec->run_index[pos & (rec->run_index_cap - 1)])
        pos++;
      rec->run_index[pos & (rec->run_index_cap - 1)] = rec->n_runs;
    }
  } else if (e->state == 0) {   /* second sighting: one mask, if coverages never meet */
    const int W = x1 - x0, H = y1 - y0;
    uint8_t *m = calloc ((size_t) W, (size_t) H);
    if (!m)
      abort ();
    int clash = 0;
    for (uint32_t i = 0; i < n && !clash; i++) {
      const RunGlyph *g = &rec->run[i];
      const uint8_t *src = rec->atlas + g->offset;     /* rows g->w + MASK_PAD apart */
      for (int y = 0; y < g->h && !clash; y++) {
        uint8_t *d = m + (size_t) (g->oy - y0 + y) * W + (g->ox - x0);
        const uint8_t *sr = src + (size_t) y * (g->w + MASK_PAD);
        for (int x = 0; x < g->w; x++) {
          if (sr[x]) {
            if (d[x]) {
              clash = 1;
              break;
            }
            d[x] = sr[x];
          }
        }
      }
    }
    if (clash) {
      e->state = 2;
      rec->runs_clashed++;
    } else {
      rec->runs_merged++;
      e->offset = atlas_put_mask (rec, m, W, H, 0);
      e->w = W;
      e->h = H;
      e->state = 1;
    }
    free (m);
  }
  const int state = e->state;
  const uint32_t offset = e->offset;
  const int W = e->w;
  g_rec_mutex_unlock (&rec->lock);
  free (key);
  if (state == 1) {
    const int *c = rec->run_clip;
    int bx0 = imax (x0, c[0]), by0 = imax (y0, c[1]), bx1 = imin (x1, c[2]), by1 = imin (y1, c[3]);
    if (bx1 > bx0 && by1 > by0) {
      flutl_prim *p = push_prim (rec);
      p->op = FLUTL_OP_MASK;
      p->rule = FLUTL_RULE_PIXMAN;
      p->cx0 = (int16_t) bx0;
      p->cy0 = (int16_t) by0;
      p->cx1 = (int16_t) bx1;
      p->cy1 = (int16_t) by1;
      p->color = rec->run_pixel;
      p->param = (bx0 > x0 || bx1 < x1) ? FLUTL_MASK_XCLIP : 0;
      p->a = (int32_t) offset;
      p->b = W + MASK_PAD;
      p->c = x0;
      p->d = y0;
    }
  } else {
    run_emit_glyphs (rec);
  }
  rec->n_run = 0;
  rec->in_run_flush = 0;
}

/* ---- unknown glyphs: Pango's hex box -------------------------------------------
 * For a code point no font covers, pango_cairo_show_glyph_string (reference
 * gstttmlrender.c:1149-1151) draws a box with the code point's hex digits in it.
 * Third-party arithmetic, not in /root/reference: Pango 1.45.3, pangocairo-
 * font.c (_pango_cairo_font_private_get_hex_box_info: the metrics, from a
 * "monospace" mini font of size / 2.2) and pangocairo-render.c
 * (_pango_cairo_renderer_draw_unknown_glyph: a stroked rectangle, then one
 * cairo_show_text per digit), restated here; pinned by the cairo-trace of the
 * reference (oracle/hexbox_trace.md) and by scene k16 (frames of the reference).
 * On an image with DejaVu only this is every CJK / Arabic / Hebrew cue. */
typedef struct
{
  PangoFont *mini;
  int rows;
  double digit_width, digit_height, pad_x, pad_y, line_width, box_height, box_descent;
} HexBox;

static void
hexbox_free (gpointer data)
{
  HexBox *hb = data;
  if (hb->mini)
    g_object_unref (hb->mini);
  g_free (hb);
}

static HexBox *
hexbox_info (PangoFont *font)
{
  static GQuark quark;
  if (!quark)
    quark = g_quark_from_static_string ("flutl-hexbox-info");
  HexBox *hb = g_object_get_qdata (G_OBJECT (font), quark);
  if (hb)
    return hb->mini ? hb : NULL;
  hb = g_new0 (HexBox, 1);
  g_object_set_qdata_full (G_OBJECT (font), quark, hb, hexbox_free);

  cairo_scaled_font_t *sf = pango_cairo_font_get_scaled_font ((PangoCairoFont *) font);
  PangoFontMap *fontmap = pango_font_get_font_map (font);
  if (!sf || cairo_scaled_font_status (sf) != CAIRO_STATUS_SUCCESS || !fontmap)
    return NULL;
  cairo_font_options_t *options = cairo_font_options_create ();
  cairo_matrix_t ctm;
  cairo_scaled_font_get_ctm (sf, &ctm);
  cairo_scaled_font_get_font_options (sf, options);
  const int hinted = cairo_font_options_get_hint_metrics (options) != CAIRO_HINT_METRICS_OFF;
  double scale_x = 1., scale_x_inv = 1., scale_y = 1., scale_y_inv = 1.;
  if (hinted) {
    double x = 1., y = 0.;
    cairo_matrix_transform_distance (&ctm, &x, &y);
    scale_x = sqrt (x * x + y * y);
    scale_x_inv = 1 / scale_x;
    x = 0.;
    y = 1.;
    cairo_matrix_transform_distance (&ctm, &x, &y);
    scale_y = sqrt (x * x + y * y);
    scale_y_inv = 1 / scale_y;
  }
#define HB_HINT(v, sc, inv) (ceil (((v) - 1e-5) * (sc)) * (inv))
  PangoFontDescription *desc = pango_font_describe_with_absolute_size (font);
  const double size = pango_font_description_get_size (desc) / (1. * PANGO_SCALE);
  pango_font_description_unset_fields (desc, PANGO_FONT_MASK_GRAVITY);
  pango_font_description_set_family_static (desc, "monospace");
  int rows = 2;
  double mini_size = size / 2.2;
  if (hinted) {
    mini_size = HB_HINT (mini_size, scale_y, scale_y_inv);
    if (mini_size < 6.0) {
      rows = 1;
      mini_size = MIN (MAX (size - 1, 0), 6.0);
    }
  }
  pango_font_description_set_absolute_size (desc, pango_units_from_double (mini_size));
  {
    PangoContext *context = pango_font_map_create_context (fontmap);
    PangoMatrix pm = { ctm.xx, ctm.xy, ctm.yx, ctm.yy, ctm.x0, ctm.y0 };
    pango_context_set_matrix (context, &pm);
    pango_context_set_language (context, pango_script_get_sample_language (PANGO_SCRIPT_LATIN));
    pango_cairo_context_set_font_options (context, options);
    hb->mini = pango_font_map_load_font (fontmap, context, desc);
    g_object_unref (context);
  }
  pango_font_description_free (desc);
  cairo_font_options_destroy (options);
  cairo_scaled_font_t *smini = hb->mini
      ? pango_cairo_font_get_scaled_font ((PangoCairoFont *) hb->mini) : NULL;
  if (!smini || cairo_scaled_font_status (smini) != CAIRO_STATUS_SUCCESS) {
    g_clear_object (&hb->mini);
    return NULL;
  }
  double width = 0, height = 0;
  for (int i = 0; i < 16; i++) {
    const char c[2] = { "0123456789ABCDEF"[i], 0 };
    cairo_text_extents_t ext;
    cairo_scaled_font_text_extents (smini, c, &ext);
    width = MAX (width, ext.width);
    height = MAX (height, ext.height);
  }
  cairo_font_extents_t fe;
  cairo_scaled_font_extents (sf, &fe);
  if (fe.ascent + fe.descent <= 0) {
    fe.ascent = 14;               /* PANGO_UNKNOWN_GLYPH_HEIGHT */
    fe.descent = 0;
  }
  double pad = (fe.ascent + fe.descent) / 43;
  pad = MIN (pad, mini_size);
  hb->rows = rows;
  hb->digit_width = width;
  hb->digit_height = height;
  hb->pad_x = hb->pad_y = pad;
  if (hinted) {
    hb->digit_width = HB_HINT (hb->digit_width, scale_x, scale_x_inv);
    hb->digit_height = HB_HINT (hb->digit_height, scale_y, scale_y_inv);
    hb->pad_x = HB_HINT (hb->pad_x, scale_x, scale_x_inv);
    hb->pad_y = HB_HINT (hb->pad_y, scale_y, scale_y_inv);
  }
  hb->line_width = MIN (hb->pad_x, hb->pad_y);
  hb->box_height = 3 * hb->pad_y + rows * (hb->pad_y + hb->digit_height);
  if (rows == 1 || hb->box_height <= fe.ascent)
    hb->box_descent = 2 * hb->pad_y;
  else if (hb->box_height <= fe.ascent + fe.descent - 2 * hb->pad_y)
    hb->box_descent = 2 * hb->pad_y + hb->box_height - fe.ascent;
  else
    hb->box_descent = fe.descent * hb->box_height / (fe.ascent + fe.descent);
  if (hinted)
    hb->box_descent = HB_HINT (hb->box_descent, scale_y, scale_y_inv);
#undef HB_HINT
  return hb;
}

/* Code points Pango 1.45 does NOT draw as a hex box when they come as unknown glyphs
 * (_pango_cairo_renderer_draw_unknown_glyph's other branches): line breaks and tabs (an
 * arrow: a filled polygon), spaces (a centred dot: cairo_arc + fill), default ignorables (a
 * short name in a box).  What reaches them through the reference on this stack -- settled by
 * cairo-traces of the reference over 31 such code points in the three DejaVu families
 * (oracle/hexbox_trace.md, "the other branches"): U+2028 LINE SEPARATOR in a font without
 * it (DejaVu Serif, Sans Mono); HarfBuzz gives every unknown space the font's space glyph and
 * hides default ignorables, Pango lays tabs out itself, so the dot and the name boxes only
 * come under PANGO_ATTR_SHOW, which the reference never sets.  The arrows and the dot are
 * restated below; a name box still fails the buffer (flutl_recorder_unsupported). */
static int
hexbox_is_named (gunichar ch)
{
  return ch == 0x00ad || ch == 0x034f || ch == 0x061c || (ch >= 0x200b && ch <= 0x200f) ||
      (ch >= 0x202a && ch <= 0x202e) || (ch >= 0x2060 && ch <= 0x2069) || ch == 0xfeff;
}

/* The arrow like U+21B5 Pango draws for a line break no font has a glyph for ('\n', U+2028,
 * U+2029): nine points, filled.  Pinned point for point by cairo-traces of the reference at
 * nine font sizes in two families (oracle/hexbox_trace.md; scene k21). */
static int
line_break_arrow (const HexBox *hb, const PangoGlyphInfo *gi, double *pts)
{
  const double cx = (double) gi->geometry.x_offset / PANGO_SCALE;
  const double cy = gi->geometry.y_offset == 0
      ? 0.0 : (double) gi->geometry.y_offset / PANGO_SCALE;
  const double gw = (double) gi->geometry.width / PANGO_SCALE;
  const double offset = 0.2 * gw, al = gw - 2 * offset;         /* arrow length */
  const double tl = MIN (hb->digit_width, 0.75 * al);            /* tip length */
  const double ah = al - 0.5 * tl;                                /* arrow height */
  const double tw2 = 2.5 * hb->line_width, lw2 = 0.5 * hb->line_width;
  const double x = cx + offset;
  const double y = cy - (hb->box_height - al) / 2;               /* centre of the tip */
  int n = 0;
#define PT(px, py) do { pts[2 * n] = (px); pts[2 * n + 1] = (py); n++; } while (0)
  PT (x, y);
  PT (x + tl, y - tw2);
  PT (x + tl, y - lw2);
  PT (x + al - lw2, y - lw2);
  PT (x + al - lw2, y - ah);
  PT (x + al + lw2, y - ah);
  PT (x + al + lw2, y + lw2);
  PT (x + tl, y + lw2);
  PT (x + tl, y + tw2);
#undef PT
  return n;
}

static int
glyph_is_line_break_arrow (const PangoGlyphInfo *gi)
{
  const gunichar ch = gi->glyph & ~PANGO_GLYPH_UNKNOWN_FLAG;
  return (gi->glyph & PANGO_GLYPH_UNKNOWN_FLAG) && gi->glyph != PANGO_GLYPH_INVALID_INPUT &&
      (ch == '\n' || ch == 0x2028 || ch == 0x2029);
}

static void
hexbox_show (FlutlRecorder *rec, PangoFont *font, const PangoGlyphInfo *gi)
{
  const gunichar ch = gi->glyph & ~PANGO_GLYPH_UNKNOWN_FLAG;
  /* the point pango_cairo_renderer_show_text_glyphs hands to the box */
  const double cx = (double) gi->geometry.x_offset / PANGO_SCALE;
  const double cy = gi->geometry.y_offset == 0
      ? 0.0 : (double) gi->geometry.y_offset / PANGO_SCALE;
  HexBox *hb;
  if (gi->glyph == (0x20 | PANGO_GLYPH_UNKNOWN_FLAG))
    return;                     /* "no hex boxes for space, please" */
  if (gi->glyph == PANGO_GLYPH_INVALID_INPUT || ch > 0x10FFFF || hexbox_is_named (ch) ||
      !(hb = hexbox_info (font))) {
    rec->unsupported++;
    return;
  }
  if (rec->color_clear)
    return;
  if (ch == 0x2423 || g_unichar_type (ch) == G_UNICODE_SPACE_SEPARATOR) {
    /* "a centered dot": cairo_new_sub_path, cairo_arc (x, y, 1.5 line widths, 0, 2 pi),
     * cairo_close_path, cairo_fill -- the arc as Cairo flattens it is not restated */
    rec->unsupported++;
    return;
  }
  if (ch == '\t') {
    /* (an arrow like U+21A6.  Pango lays tabs out itself: no tab glyph reaches the
     * reference's show call, so there is no trace to pin a restatement on) */
    rec->unsupported++;
    return;
  }
  if (ch == '\n' || ch == 0x2028 || ch == 0x2029) {
    double pts[24];
    flutl_recorder_path_polygon (rec, pts, line_break_arrow (hb, gi, pts));
    flutl_recorder_fill (rec);
    return;
  }
  cairo_scaled_font_t *smini = pango_cairo_font_get_scaled_font ((PangoCairoFont *) hb->mini);
  const int rows = hb->rows, cols = (ch > 0xffff ? 6 : 4) / rows;
  char buf[8];
  g_snprintf (buf, sizeof buf, ch > 0xffff ? "%06X" : "%04X", ch);
  const double width = 3 * hb->pad_x + cols * (hb->digit_width + hb->pad_x);
  double lsb = ((double) gi->geometry.width / PANGO_SCALE - width) * .5;
  lsb = floor (lsb / hb->pad_x) * hb->pad_x;
  /* the frame: cairo_rectangle + cairo_stroke (line join miter, limit 2) */
  {
    cairo_t *cr = rec->cr;
    cairo_save (cr);
    cairo_set_line_width (cr, hb->line_width);
    cairo_set_miter_limit (cr, 2.);
    flutl_recorder_path_rect (rec, cx + lsb + .5 * hb->pad_x,
        cy + hb->box_descent - hb->box_height + hb->line_width * 0.5, width - hb->pad_x,
        hb->box_height - hb->line_width);
    flutl_recorder_stroke (rec);
    cairo_restore (cr);
  }
  /* the digits: cairo_move_to + cairo_show_text with the mini font */
  const double x0 = cx + lsb + hb->pad_x * 2;
  const double y0 = cy + hb->box_descent - hb->pad_y * 2;
  for (int row = 0; row < rows; row++) {
    const double y = y0 - (rows - 1 - row) * (hb->digit_height + hb->pad_y);
    for (int col = 0; col < cols; col++) {
      /* cairo_move_to keeps the point in device space as 24.8 fixed point, and
       * cairo_show_text asks for it back (cairo_get_current_point: through the
       * inverse CTM): the digit sits on a 1/256 px grid, which decides the pixel
       * it is rounded to when that grid point is n + 1/2 -- one digit in 256, or
       * every second column at some sizes (fuzz seed 560001, scale 2: the right
       * digits one pixel off under a vertical writing mode) */
      double x = x0 + col * (hb->digit_width + hb->pad_x), yy = y;
      {
        /* (the inverse as the reference's context holds it: kept up call by call,
         * not inverted afresh -- under a rotated CTM the two differ in the last
         * bit, which decides where a grid point at n + 1/2 goes; fuzz seed 800452) */
        cairo_matrix_t m, inv;
        cairo_get_matrix (rec->cr, &m);
        inv = m;
        if (rec->ctm_inverse)
          inv = *rec->ctm_inverse;
        if (rec->ctm_inverse || cairo_matrix_invert (&inv) == CAIRO_STATUS_SUCCESS) {
          cairo_matrix_transform_point (&m, &x, &yy);
          x = fixed_from_double (x) / 256.0;
          yy = fixed_from_double (yy) / 256.0;
          cairo_matrix_transform_point (&inv, &x, &yy);
        }
      }
      cairo_glyph_t *glyphs = NULL;
      int n = 0;
      if (cairo_scaled_font_text_to_glyphs (smini, x, yy, &buf[row * cols + col], 1, &glyphs,
              &n, NULL, NULL, NULL) != CAIRO_STATUS_SUCCESS)
        continue;
      for (int k = 0; k < n; k++)
        show_scaled_glyph (rec, smini, (uint32_t) glyphs[k].index, glyphs[k].x, glyphs[k].y);
      cairo_glyph_free (glyphs);
    }
  }
}

void
flutl_recorder_show_glyph (FlutlRecorder *rec, PangoFont *font,
    const PangoGlyphInfo *gi)
{
  if (gi->glyph == PANGO_GLYPH_EMPTY)
    return;
  if (gi->glyph & PANGO_GLYPH_UNKNOWN_FLAG) {
    hexbox_show (rec, font, gi);
    return;
  }
  if (rec->color_clear)
    return;
  cairo_scaled_font_t *sf =
      pango_cairo_font_get_scaled_font ((PangoCairoFont *) font);
  if (!sf)
    return;
  /* pango_cairo_renderer: position = current point (none -> 0,0) + offsets */
  show_scaled_glyph (rec, sf, gi->glyph, (double) gi->geometry.x_offset / PANGO_SCALE,
      gi->geometry.y_offset == 0 ? 0.0 : (double) gi->geometry.y_offset / PANGO_SCALE);
}

/* ---- strokes ---------------------------------------------------------------
 * Outline paths are not built eagerly.  The walker hands over the pieces of
 * the path (glyph outlines, rectangles) together with the CTM each piece is
 * issued under; at stroke time the pieces + line width form a key into a cache
 * of coverage masks that lives across buffers.  Only on a miss is the path
 * really built on the tracking context (by the very calls the reference makes:
 * pango_cairo_glyph_string_path / cairo_rectangle) and stroked, once, onto the
 * scratch A8 surface.  The reference re-strokes every outline for every
 * buffer (gstttmlrender.c:1343-1355); that is >90 % of its time on outlined
 * scenes.
 *
 * A mask is valid at any integer translation of the path (device coordinates
 * are 24.8 fixed point, so an integer shift moves every vertex exactly), as
 * long as the stroke was not cut by the borders of the scratch surface. */

static void
path_ops_clear (FlutlRecorder *rec)
{
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    if (rec->ops[i].font)
      g_object_unref (rec->ops[i].font);
    if (rec->ops[i].sfont)
      cairo_scaled_font_destroy (rec->ops[i].sfont);
    free (rec->ops[i].poly);
  }
  rec->n_ops = 0;
}

static PathOp *
path_op_push (FlutlRecorder *rec)
{
  if (rec->n_ops == rec->cap_ops) {
    rec->cap_ops = rec->cap_ops ? rec->cap_ops * 2 : 128;
    rec->ops = xrealloc (rec->ops, rec->cap_ops * sizeof (PathOp));
  }
  PathOp *op = &rec->ops[rec->n_ops++];
  memset (op, 0, sizeof *op);
  if (rec->ctm_given)
    op->m = rec->ctm;
  else
    cairo_get_matrix (rec->cr, &op->m);
  return op;
}

/* The CTM of the next glyph shown or added to the path, given instead of set on the tracking
 * context and read back (a cairo_set_matrix inverts and checks the matrix; per glyph and pass
 * it was a tenth of a region of new text).  NULL: the context's again. */
void
flutl_recorder_set_ctm (FlutlRecorder *rec, const cairo_matrix_t *m)
{
  rec->ctm_given = m != NULL;
  if (m)
    rec->ctm = *m;
}

void
flutl_recorder_path_glyph (FlutlRecorder *rec, PangoFont *font,
    const PangoGlyphInfo *gi)
{
  HexBox *hb;
  if (glyph_is_line_break_arrow (gi) && (hb = hexbox_info (font))) {
    /* reference (Pango 1.45): the arrow's branch of the unknown-glyph renderer ends in
     * cairo_fill whether it was asked to show or only to add to the path -- under
     * pango_cairo_glyph_string_path (gstttmlrender.c:1153-1154) that fills THE WHOLE PENDING
     * PATH, every glyph outline gathered for the stroke so far, in the outline's colour, and
     * leaves the path empty: the text before a line separator gets a filled outline instead
     * of a stroked one.  Reproduced (scene k21; found by comparing with the reference). */
    double pts[24];
    flutl_recorder_path_polygon (rec, pts, line_break_arrow (hb, gi, pts));
    flutl_recorder_fill (rec);
    return;
  }
  PathOp *op = path_op_push (rec);
  op->font = g_object_ref (font);
  op->gi = *gi;
}

void
flutl_recorder_path_scaled_glyph (FlutlRecorder *rec, cairo_scaled_font_t *sfont,
    unsigned long glyph, double hscale)
{
  PathOp *op = path_op_push (rec);
  op->sfont = cairo_scaled_font_reference (sfont);
  op->sglyph = glyph;
  op->hscale = hscale;
}

void
flutl_recorder_path_rect (FlutlRecorder *rec, double x, double y, double w, double h)
{
  PathOp *op = path_op_push (rec);
  op->x = x;
  op->y = y;
  op->w = w;
  op->h = h;
}

/* cairo_move_to + cairo_line_to ... + cairo_close_path (the arrows Pango draws for a tab
 * or a line separator no font has, hexbox_show) */
void
flutl_recorder_path_polygon (FlutlRecorder *rec, const double *pts, int n)
{
  if (n < 3 || n > 12) {
    rec->unsupported++;
    return;
  }
  PathOp *op = path_op_push (rec);
  double x0 = pts[0], y0 = pts[1], x1 = pts[0], y1 = pts[1];
  op->npoly = n;
  op->poly = malloc ((size_t) n * 2 * sizeof (double));
  if (!op->poly)
    abort ();
  memcpy (op->poly, pts, (size_t) n * 2 * sizeof (double));
  for (int i = 1; i < n; i++) {
    x0 = MIN (x0, pts[2 * i]);
    x1 = MAX (x1, pts[2 * i]);
    y0 = MIN (y0, pts[2 * i + 1]);
    y1 = MAX (y1, pts[2 * i + 1]);
  }
  op->x = x0;
  op->y = y0;
  op->w = x1 - x0;
  op->h = y1 - y0;
}

static void
key_font_pin (FlutlRecorder *rec, cairo_scaled_font_t *sf)
{
  if (!sf)
    return;
  for (uint32_t i = 0; i < rec->n_key_fonts; i++)
    if (rec->key_fonts[i] == sf)
      return;
  if (rec->n_key_fonts == rec->cap_key_fonts) {
    rec->cap_key_fonts = rec->cap_key_fonts ? rec->cap_key_fonts * 2 : 8;
    rec->key_fonts = xrealloc (rec->key_fonts, rec->cap_key_fonts * sizeof *rec->key_fonts);
  }
  rec->key_fonts[rec->n_key_fonts++] = cairo_scaled_font_reference (sf);
}

/* The fonts the pending ops name: pinned when their key is stored. */
static void
key_fonts_pin_ops (FlutlRecorder *rec)
{
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    const PathOp *op = &rec->ops[i];
    if (op->sfont)
      key_font_pin (rec, op->sfont);
    else if (op->font)
      key_font_pin (rec, pango_cairo_font_get_scaled_font ((PangoCairoFont *) op->font));
  }
}

/* Serialise what determines the coverage mask, relative to an integer base. */
static size_t
stroke_key (FlutlRecorder *rec, double line_width, int base_x, int base_y,
    uint8_t **out)
{
  size_t per = 10 * sizeof (double) + 2 * sizeof (uint64_t);
  size_t n = 2 * sizeof (double) + rec->n_ops * per;
  uint8_t *k = malloc (n), *p = k;
  if (!k)
    abort ();
  double miter = cairo_get_miter_limit (rec->cr);
  memcpy (p, &line_width, sizeof (double));
  p += sizeof (double);
  memcpy (p, &miter, sizeof (double));
  p += sizeof (double);
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    const PathOp *op = &rec->ops[i];
    double d[10] = { op->m.x0 - base_x, op->m.y0 - base_y, op->m.xx, op->m.yy,
      op->x, op->y, op->w, op->h, op->m.xy, op->m.yx };
    uint64_t u[2] = { 0, 0 };
    if (op->sfont) {
      u[0] = (uint64_t) (uintptr_t) op->sfont ^ 0x5a5a5a5a00000000ull;
      u[1] = (uint64_t) op->sglyph;
      d[4] = op->hscale;
    } else if (op->font) {
      u[0] = (uint64_t) (uintptr_t) pango_cairo_font_get_scaled_font (
          (PangoCairoFont *) op->font);
      u[1] = ((uint64_t) op->gi.glyph << 32) ^
          ((uint64_t) (uint32_t) op->gi.geometry.x_offset << 16) ^
          (uint64_t) (uint32_t) op->gi.geometry.y_offset;
      d[4] = op->gi.geometry.x_offset;
      d[5] = op->gi.geometry.y_offset;
    } else if (op->npoly) {
      /* (its points, hashed: the box alone does not say which polygon) */
      uint64_t hsh = 1469598103934665603ull;
      const uint8_t *b = (const uint8_t *) op->poly;
      for (size_t q = 0; q < (size_t) op->npoly * 2 * sizeof (double); q++)
        hsh = (hsh ^ b[q]) * 1099511628211ull;
      u[0] = hsh;
      u[1] = 0xa5a5000000000000ull | (uint64_t) op->npoly;
    }
    memcpy (p, d, sizeof d);
    p += sizeof d;
    memcpy (p, u, sizeof u);
    p += sizeof u;
  }
  *out = k;
  return n;
}

static void
stroke_emit (FlutlRecorder *rec, const StrokeEntry *e, int dx, int dy)
{
  int gx, gy;
  raster_origin (rec, &gx, &gy);
  /* the surface stops being "clear" if the extents of the stroke meet it,
   * whether or not any coverage does (see box_hits_target) -- after the rule
   * of this very primitive has been chosen */
  const int was_clear = target_is_clear (rec);
  if (box_hits_target (rec, e->ext[0] + dx + gx, e->ext[1] + dy + gy,
          e->ext[2] + dx + gx, e->ext[3] + dy + gy))
    target_touched (rec);
  if (e->w == 0 && !e->job)
    return;
  int c[4];
  if (!effective_clip (rec, c))
    return;
  flutl_prim tmp;
  memset (&tmp, 0, sizeof tmp);
  tmp.op = FLUTL_OP_MASK;
  tmp.rule = (rec->color_opaque || was_clear) ? FLUTL_RULE_LERP : FLUTL_RULE_PIXMAN;
  tmp.color = rec->pixel;
  if (e->job) {
    /* the mask is still being made: a placeholder (an empty primitive, valid as
     * it stands) and a note of how to fill it in */
    tmp.b = 1;
    flutl_prim *p = push_prim (rec);
    *p = tmp;
    if (rec->n_patches == rec->cap_patches) {
      rec->cap_patches = rec->cap_patches ? rec->cap_patches * 2 : 16;
      rec->patches = xrealloc (rec->patches, rec->cap_patches * sizeof (FlutlPatch));
    }
    FlutlPatch *pt = &rec->patches[rec->n_patches++];
    pt->prim = rec->n_prims - 1;
    pt->entry = (uint32_t) (e - rec->strokes);
    pt->ox = dx + gx;
    pt->oy = dy + gy;
    for (int k = 0; k < 4; k++)
      pt->clip[k] = c[k];
    rec->region_deferred = 1;
    return;
  }
  if (!mask_prim_place (&tmp, e, dx + gx, dy + gy, c))
    return;
  *push_prim (rec) = tmp;
}

/* ---- rasterisation of a path into an A8 mask, in three steps ------------------
 * prepare: the pending ops become a real Cairo path on rec->cr; its (generous)
 *          device extents give the box to look at and tell whether the stroke
 *          stays clear of the frame borders (then the mask is valid under any
 *          integer translation);
 * draw:    Cairo's stroker / filler + scan converter write coverage into the
 *          scratch surface -- on the streaming thread, or, for the row-disjoint
 *          lines of one stroke, on worker threads (below);
 * crop:    the covered part of the box moves into the atlas. */
typedef struct
{
  int bx0, by0, bx1, by1;       /* box to look at; empty: nothing to draw */
  int inside;                   /* the path stays clear of the limits */
  int lim[4];                   /* limits: frame, clip and group surface */
  int ext[4];                   /* extents of the stroke / fill, unclipped */
} RasterBox;

/* What bounds a stroke in the reference: the surface it draws on (the frame,
 * or the region's temporary surface when it has an opacity) and the region's
 * clip rectangle.  Cairo hands this box to the polygon as "limits", and an
 * edge that crosses a limit is cut there with rounded coordinates -- which
 * moves coverages by +-1 here and there ALONG that edge, not only at the
 * border (found by tools/fuzz_ttml.py).  So a path is rasterised under exactly
 * these limits, and its mask is position independent only if it stays clear
 * of them. */
/* Paths are rasterised in the coordinates the reference strokes them in: the
 * frame's, or -- inside an opacity group -- those of the region's temporary
 * surface (gstttmlrender.c:1250-1255: cairo_translate (-originx, -originy)).
 * This matters: Cairo's coverage is exact under integer translations in x but
 * NOT in y (+-1 at a few pixels, irregularly: measured with one line of 17 px
 * text at 40 vertical positions, 2 variants of its outline mask), so a moved
 * group keeps its mask, and a mask is otherwise only reused at the same y. */
static gint64 prepare_path_us, prepare_extents_us;   /* statistics: building paths, their extents */

void
flutl_recorder_prepare_stats (uint64_t *path_us, uint64_t *extents_us)
{
  *path_us = (uint64_t) __atomic_load_n (&prepare_path_us, __ATOMIC_RELAXED);
  *extents_us = (uint64_t) __atomic_load_n (&prepare_extents_us, __ATOMIC_RELAXED);
}

static void
raster_origin (const FlutlRecorder *rec, int *ox, int *oy)
{
  *ox = rec->in_group ? rec->group[0] : 0;
  *oy = rec->in_group ? rec->group[1] : 0;
}

static void
current_limits (FlutlRecorder *rec, int lim[4])
{
  int ox, oy;
  raster_origin (rec, &ox, &oy);
  if (rec->in_group) {
    /* the region's temporary surface, which may reach beyond the frame: the
     * frame only cuts the finished group, pixel by pixel */
    lim[0] = rec->group[0];
    lim[1] = rec->group[1];
    lim[2] = rec->group[2];
    lim[3] = rec->group[3];
  } else {
    lim[0] = 0;
    lim[1] = 0;
    lim[2] = rec->width;
    lim[3] = rec->height;
  }
  if (rec->has_clip) {
    lim[0] = imax (lim[0], rec->clip[0]);
    lim[1] = imax (lim[1], rec->clip[1]);
    lim[2] = imin (lim[2], rec->clip[2]);
    lim[3] = imin (lim[3], rec->clip[3]);
  }
  if (lim[2] < lim[0])
    lim[2] = lim[0];
  if (lim[3] < lim[1])
    lim[3] = lim[1];
  lim[0] -= ox;                 /* in raster coordinates */
  lim[2] -= ox;
  lim[1] -= oy;
  lim[3] -= oy;
}

static void raster_box_from_extents (RasterBox *rb, double ex1, double ey1, double ex2, double ey2,
    int sw, int sh);

/* the pending ops become a real Cairo path on rec->cr, under the reference's bounds */
static void
raster_build_path (FlutlRecorder *rec, RasterBox *rb)
{
  cairo_t *cr = rec->cr;
  int ox, oy;
  const gint64 t_prepare = g_get_monotonic_time ();
  raster_origin (rec, &ox, &oy);
  cairo_new_path (cr);
  /* The reference builds its path on a context that carries the region's clip,
   * and cairo_glyph_path DROPS glyphs whose origin lies more than 10 font sizes
   * outside surface extents x clip extents (_cairo_gstate_transform_glyphs_to_
   * backend) -- the far end of a noWrap line that runs a long way out of its
   * region.  Missing or not, they lie outside the clip; but their edges share
   * rows with the visible ones, and Cairo's coverage is not independent of
   * that (found by tools/fuzz_ttml.py at FUZZ_SCALE=2).  So the path is built
   * under the same bounds. */
  current_limits (rec, rb->lim);
  cairo_save (cr);
  cairo_identity_matrix (cr);
  cairo_rectangle (cr, rb->lim[0], rb->lim[1], rb->lim[2] - rb->lim[0], rb->lim[3] - rb->lim[1]);
  cairo_clip (cr);
  cairo_new_path (cr);
  /* CTM coordinates -> raster coordinates; nothing to do (and no rounding of
   * m.x0) where the CTM already is the reference's own */
  ox -= rec->shift[0];
  oy -= rec->shift[1];
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    const PathOp *op = &rec->ops[i];
    cairo_matrix_t m = op->m;
    if (ox)
      m.x0 -= ox;
    if (oy)
      m.y0 -= oy;
    cairo_save (cr);
    cairo_set_matrix (cr, &m);
    if (op->sfont) {
      cairo_glyph_t g = { op->sglyph, 0, 0 };
      cairo_scale (cr, op->hscale, 1.0);
      cairo_set_scaled_font (cr, op->sfont);
      cairo_glyph_path (cr, &g, 1);
    } else if (op->font) {
      PangoGlyphInfo gi = op->gi;
      PangoGlyphString one = { 1, &gi, NULL, 1 };
      pango_cairo_glyph_string_path (cr, op->font, &one);
    } else if (op->npoly) {
      cairo_move_to (cr, op->poly[0], op->poly[1]);
      for (int q = 1; q < op->npoly; q++)
        cairo_line_to (cr, op->poly[2 * q], op->poly[2 * q + 1]);
      cairo_close_path (cr);
    } else {
      cairo_rectangle (cr, op->x, op->y, op->w, op->h);
    }
    cairo_restore (cr);
  }
  cairo_restore (cr);           /* the clip goes, the path stays */
  __atomic_fetch_add (&prepare_path_us, g_get_monotonic_time () - t_prepare, __ATOMIC_RELAXED);
}

static void
raster_prepare (FlutlRecorder *rec, int fill, RasterBox *rb)
{
  cairo_t *cr = rec->cr;
  raster_build_path (rec, rb);
  double ex1, ey1, ex2, ey2;
  const gint64 t_path_done = g_get_monotonic_time ();
  cairo_save (cr);
  cairo_identity_matrix (cr);            /* extents in device space */
  if (fill)
    cairo_fill_extents (cr, &ex1, &ey1, &ex2, &ey2);
  else
    cairo_stroke_extents (cr, &ex1, &ey1, &ex2, &ey2);
  cairo_restore (cr);
  __atomic_fetch_add (&prepare_extents_us, g_get_monotonic_time () - t_path_done, __ATOMIC_RELAXED);
  raster_box_from_extents (rb, ex1, ey1, ex2, ey2, rec->sw, rec->sh);
}

/* rb->lim is set; the rest of the box from the (exact) extents of the stroke / fill */
static void
raster_box_from_extents (RasterBox *rb, double ex1, double ey1, double ex2, double ey2, int sw,
    int sh)
{
  rb->inside = ex1 >= rb->lim[0] + 3 && ey1 >= rb->lim[1] + 3 && ex2 <= rb->lim[2] - 3 &&
      ey2 <= rb->lim[3] - 3;
  rb->ext[0] = (int) floor (ex1);
  rb->ext[1] = (int) floor (ey1);
  rb->ext[2] = (int) ceil (ex2);
  rb->ext[3] = (int) ceil (ey2);
  rb->bx0 = imax (imax ((int) floor (ex1) - 2, rb->lim[0]), 0);
  rb->by0 = imax (imax ((int) floor (ey1) - 2, rb->lim[1]), 0);
  rb->bx1 = imin (imin ((int) ceil (ex2) + 2, rb->lim[2]), sw);
  rb->by1 = imin (imin ((int) ceil (ey2) + 2, rb->lim[3]), sh);
}

/* The covered part of the box, cut out of an A8 surface (which is left clear
 * again): malloc'd tight w x h bytes at (x, y), or NULL if nothing is covered. */
static uint8_t *
crop_coverage (cairo_surface_t *surf, const RasterBox *rb, int *ox, int *oy, int *ow, int *oh)
{
  const int bx0 = rb->bx0, by0 = rb->by0, bx1 = rb->bx1, by1 = rb->by1;
  *ox = *oy = *ow = *oh = 0;
  /* (nothing to look at: a stroke that lies outside the surface altogether -- its
   * exact extents, taken on the worker, can say so where the bounds of them the
   * streaming thread went by could not; fuzz seed 610052) */
  if (bx1 <= bx0 || by1 <= by0)
    return NULL;
  cairo_surface_flush (surf);
  uint8_t *data = cairo_image_surface_get_data (surf);
  int stride = cairo_image_surface_get_stride (surf);
  int tx0 = bx1, ty0 = by1, tx1 = -1, ty1 = -1;
  for (int y = by0; y < by1; y++) {
    const uint8_t *row = data + (size_t) y * stride;
    int first = -1, last = -1;
    /* first / last covered pixel of the row, 8 bytes at a time (most of a
     * text line's bounding box is empty) */
    int x = bx0, xe = bx1;
    while (x < xe && ((uintptr_t) (row + x) & 7) && !row[x])
      x++;
    if (x < xe && !row[x]) {
      while (x + 8 <= xe && *(const uint64_t *) (row + x) == 0)
        x += 8;
      while (x < xe && !row[x])
        x++;
    }
    if (x < xe) {
      first = x;
      while (xe > first && ((uintptr_t) (row + xe) & 7) && !row[xe - 1])
        xe--;
      if (!row[xe - 1]) {
        while (xe - 8 > first && *(const uint64_t *) (row + xe - 8) == 0)
          xe -= 8;
        while (!row[xe - 1])
          xe--;
      }
      last = xe - 1;
    }
    if (first >= 0) {
      if (first < tx0) tx0 = first;
      if (last > tx1) tx1 = last;
      if (y < ty0) ty0 = y;
      ty1 = y;
    }
  }
  *ox = *oy = *ow = *oh = 0;
  if (tx1 < tx0)
    return NULL;
  int w = tx1 - tx0 + 1, h = ty1 - ty0 + 1;
  uint8_t *tight = malloc ((size_t) w * h);
  if (!tight)
    abort ();
  for (int y = 0; y < h; y++) {
    memcpy (tight + (size_t) y * w, data + (size_t) (ty0 + y) * stride + tx0, w);
    memset (data + (size_t) (ty0 + y) * stride + tx0, 0, (size_t) w);
  }
  cairo_surface_mark_dirty_rectangle (surf, tx0, ty0, w, h);
  *ox = tx0;
  *oy = ty0;
  *ow = w;
  *oh = h;
  return tight;
}

static void
raster_crop (FlutlRecorder *rec, StrokeEntry *e, const RasterBox *rb)
{
  int x, y, w, h;
  uint8_t *tight = crop_coverage (rec->scratch, rb, &x, &y, &w, &h);
  if (!tight)
    return;
  e->offset = atlas_put_mask (rec, tight, w, h, 1);
  free (tight);
  e->x = x;
  e->y = y;
  e->w = w;
  e->h = h;
}

/* Build the pending path for real, stroke it and cut the coverage out of the
 * scratch surface.  Returns 0 if the stroke touched the surface borders (the
 * mask is then only valid at this very position). */
static int stroke_draw (FlutlRecorder *rec, StrokeEntry *e, const RasterBox *rbp, int fill);

static int
stroke_rasterise (FlutlRecorder *rec, StrokeEntry *e, int fill)
{
  cairo_t *cr = rec->cr;
  RasterBox rb;
  raster_prepare (rec, fill, &rb);
  e->w = e->h = 0;
  memcpy (e->ext, rb.ext, sizeof e->ext);
  if (rb.bx1 <= rb.bx0 || rb.by1 <= rb.by0) {
    cairo_new_path (cr);
    return rb.inside;
  }
  return stroke_draw (rec, e, &rb, fill);
}

/* (the path is on the context, its box known: Cairo rasterises it) */
static int
stroke_draw (FlutlRecorder *rec, StrokeEntry *e, const RasterBox *rbp, int fill)
{
  cairo_t *cr = rec->cr;
  const RasterBox rb = *rbp;
  /* coverage = what the stroker + scan converter hand to the span renderer */
  cairo_save (cr);
  cairo_set_source_rgba (cr, 1, 1, 1, 1);
  if (!rb.inside) {             /* under the reference's limits, see current_limits */
    cairo_path_t *pth = cairo_copy_path (cr);
    cairo_new_path (cr);
    cairo_save (cr);
    cairo_identity_matrix (cr);
    cairo_rectangle (cr, rb.lim[0], rb.lim[1], rb.lim[2] - rb.lim[0], rb.lim[3] - rb.lim[1]);
    cairo_restore (cr);
    cairo_clip (cr);
    cairo_append_path (cr, pth);
    cairo_path_destroy (pth);
  }
  if (fill)
    cairo_fill (cr);
  else
    cairo_stroke (cr);
  cairo_restore (cr);
  memcpy (e->lim, rb.lim, sizeof e->lim);
  raster_crop (rec, e, &rb);
  return rb.inside;
}

/* ---- the lines of one stroke on worker threads ---------------------------------
 * Each job strokes a copied path (device coordinates: no font or Pango object
 * leaves the streaming thread) into its own cairo surface over its own rows of
 * the scratch memory; the rows of different jobs do not overlap.
 * FLUTL_RASTER_THREADS=n (1 = everything on the streaming thread). */
typedef struct
{
  cairo_path_t *path;
  double line_width;
  uint8_t *data;
  int stride, width;
  int y0, y1;
  int clip;                      /* rasterise under lim (the path crosses it) */
  int lim[4];
  GMutex *lock;
  GCond *cond;
  int *pending;
} LineJob;

static void
line_job_run (gpointer data, gpointer user_data)
{
  LineJob *j = data;
  /* The surface starts at row 0 of the scratch memory, NOT at the job's first
   * row with a device offset: the stroker flattens curves in absolute double
   * coordinates, and a translated copy differs by +-1 in a few pixels per
   * frame (measured).  Surfaces of different jobs therefore alias memory, but
   * a job only writes the pixels its stroke covers, which lie in its own rows. */
  cairo_surface_t *s = cairo_image_surface_create_for_data (
      j->data, CAIRO_FORMAT_A8, j->width, j->y1, j->stride);
  cairo_t *cr;
  cr = cairo_create (s);
  if (j->clip) {
    cairo_rectangle (cr, j->lim[0], j->lim[1], j->lim[2] - j->lim[0], j->lim[3] - j->lim[1]);
    cairo_clip (cr);
  }
  cairo_append_path (cr, j->path);
  cairo_set_source_rgba (cr, 1, 1, 1, 1);
  cairo_set_line_width (cr, j->line_width);
  cairo_stroke (cr);
  cairo_destroy (cr);
  cairo_surface_finish (s);
  cairo_surface_destroy (s);
  if (j->lock) {
    g_mutex_lock (j->lock);
    if (--*j->pending == 0)
      g_cond_signal (j->cond);
    g_mutex_unlock (j->lock);
  }
  (void) user_data;
}

static GThreadPool *raster_pool;
static int raster_threads = -1;
static GMutex raster_pool_lock;

static int
raster_thread_count (void)
{
  g_mutex_lock (&raster_pool_lock);
  if (raster_threads < 0) {
    const char *e = g_getenv ("FLUTL_RASTER_THREADS");
    int n = e ? atoi (e) : (int) g_get_num_processors ();
    raster_threads = n < 1 ? 1 : (n > 8 ? 8 : n);
    if (raster_threads > 1)
      raster_pool = g_thread_pool_new (line_job_run, NULL, raster_threads - 1, TRUE, NULL);
    if (!raster_pool)
      raster_threads = 1;
  }
  g_mutex_unlock (&raster_pool_lock);
  return raster_threads;
}

/* ---- deferred strokes ------------------------------------------------------------
 * cairo_stroke is > 90 % of the reference's time on outlined text
 * (gstttmlrender.c:1212) and was 88 % of this element's for a cue with NEW text.
 * With flutl_recorder_set_defer the streaming thread only PREPARES a stroke it
 * has no mask for -- the path is built by the very calls the reference makes,
 * under the reference's bounds, and its exact extents are taken (they decide
 * "clear" bookkeeping and mask reuse, see stroke_emit) -- and hands the path, in
 * device coordinates, to a worker thread, which strokes it under the same CTM,
 * line width and limits onto a surface of the same size.  The whole stroke, as
 * one cairo_stroke: its rows are not independent (see path_to_mask).  The frame's
 * display list gets a placeholder primitive and a FlutlPatch; the element
 * resolves the patches (flutl_recorder_resolve: wait, put the mask into the
 * atlas, fill the primitive in, recompute the region boxes) when it finishes
 * the frame -- on another thread, while the streaming thread records the next
 * buffers and more strokes run beside this one. */
typedef struct StrokeJob
{
  gint refs;
  cairo_path_t *path;            /* device coordinates */
  int path_cap;                  /* > 0: a composed path in a pooled buffer of that capacity */
  cairo_matrix_t ctm;            /* in force at cairo_stroke */
  double line_width, miter_limit;
  RasterBox rb;
  int sw, sh;                    /* size of the surface to stroke on */
  int exact_pending;             /* rb was made from bounds of the extents: the worker takes
                                  * the exact ones (rb becomes exact before the stroke) */
  gint64 t_pushed;
  int to_polygon;                /* the device makes the mask: the result is the polygon */
  void *(*edge_acquire) (void *user, size_t bytes);    /* (the recorder's, when the job was made) */
  void (*edge_release) (void *user, void *block);
  void *edge_user;
  FlutlPolygonEntry *poly;       /* result (to_polygon): the stroke's polygon, or NULL */
  uint8_t *tight;                /* result: w x h coverage at (x, y), or NULL */
  uint8_t *padded;               /* ... in the atlas's layout, and its hash: made here so */
  size_t padded_bytes;           /* that the thread that resolves frames only copies it  */
  uint64_t padded_hash;
  int x, y, w, h;
  GMutex lock;
  GCond cond;
  int done;
} StrokeJob;

typedef struct
{
  cairo_surface_t *s;
  int w, h;
} WorkerSurface;

static void
worker_surface_free (gpointer data)
{
  WorkerSurface *ws = data;
  if (ws->s)
    cairo_surface_destroy (ws->s);
  free (ws);
}

static GPrivate worker_surface_key = G_PRIVATE_INIT (worker_surface_free);

static void path_buffer_put (cairo_path_t *p, int cap);

static void
stroke_job_unref (StrokeJob *j)
{
  if (!g_atomic_int_dec_and_test (&j->refs))
    return;
  if (j->path && j->path_cap)
    path_buffer_put (j->path, j->path_cap);
  else if (j->path)
    cairo_path_destroy (j->path);
  flutl_polygon_entry_free (j->poly);
  free (j->tight);
  free (j->padded);
  g_mutex_clear (&j->lock);
  g_cond_clear (&j->cond);
  free (j);
}

static void
stroke_job_wait (StrokeJob *j)
{
  g_mutex_lock (&j->lock);
  while (!j->done)
    g_cond_wait (&j->cond, &j->lock);
  g_mutex_unlock (&j->lock);
}

static gint64 stroke_jobs_done, stroke_jobs_us, stroke_jobs_queue_us;   /* statistics */
static gint64 masks_by_cairo;           /* strokes / fills Cairo rasterised on the recording thread */
static gint64 masks_by_cairo_on_workers;        /* ... and strokes Cairo drew on a worker thread */
static gint64 fills_by_device;          /* fills that went to the device as polygons */
static gint64 stroke_jobs_polygons;     /* ... of which left as polygons for the device */

/* (process-wide) masks Cairo rasterised synchronously -- strokes the stroker declined, fills
 * with the device route off -- and fills the device converted */
void
flutl_recorder_raster_stats (uint64_t *by_cairo, uint64_t *device_fills)
{
  *by_cairo = (uint64_t) __atomic_load_n (&masks_by_cairo, __ATOMIC_RELAXED) +
      (uint64_t) __atomic_load_n (&masks_by_cairo_on_workers, __ATOMIC_RELAXED);
  *device_fills = (uint64_t) __atomic_load_n (&fills_by_device, __ATOMIC_RELAXED);
}

uint64_t
flutl_recorder_polygon_jobs (void)
{
  return (uint64_t) __atomic_load_n (&stroke_jobs_polygons, __ATOMIC_RELAXED);
}
static gint64 stroke_jobs_extents_us, stroke_jobs_draw_us, stroke_jobs_crop_us;
static gint64 resolve_wait_us;  /* time flutl_recorder_resolve spent waiting for workers */

uint64_t
flutl_recorder_resolve_wait_us (void)
{
  return (uint64_t) __atomic_load_n (&resolve_wait_us, __ATOMIC_RELAXED);
}

void
flutl_recorder_stroke_stats (uint64_t *jobs, uint64_t *busy_us, uint64_t *queued_us)
{
  *jobs = (uint64_t) __atomic_load_n (&stroke_jobs_done, __ATOMIC_RELAXED);
  *busy_us = (uint64_t) __atomic_load_n (&stroke_jobs_us, __ATOMIC_RELAXED);
  *queued_us = (uint64_t) __atomic_load_n (&stroke_jobs_queue_us, __ATOMIC_RELAXED);
}

/* The stroke as the polygon Cairo makes of it (flutl_stroker.c); the device scan
 * converts it (flutl_atlas_put_polygon) and the mask never exists in host memory.
 * 0: not a stroke the stroker takes (Cairo draws it, below). */
static int
stroke_job_polygon (StrokeJob *j)
{
  FlutlPolygon pg;
  /* what bounds the stroke in the reference: its surface and the region's clip
   * (current_limits), within the surface paths are rasterised on here */
  const int target[4] = { imax (j->rb.lim[0], 0), imax (j->rb.lim[1], 0),
    imin (j->rb.lim[2], j->sw), imin (j->rb.lim[3], j->sh) };
  flutl_polygon_init (&pg);
  if (flutl_stroke_to_polygon (j->path, &j->ctm, j->line_width, j->miter_limit,
          CAIRO_LINE_JOIN_MITER, 0.1, target, &pg) != FLUTL_STROKER_OK) {
    flutl_polygon_fini (&pg);
    return 0;
  }
  /* the stroke's extents as cairo_stroke_extents gives them: those of its polygon
   * before anything is cut to limits */
  if (pg.uext[2] >= pg.uext[0]) {
    raster_box_from_extents (&j->rb, pg.uext[0] / 256.0, pg.uext[1] / 256.0, pg.uext[2] / 256.0,
        pg.uext[3] / 256.0, j->sw, j->sh);
  } else {
    raster_box_from_extents (&j->rb, 0, 0, 0, 0, j->sw, j->sh);
  }
  if (pg.box[2] > pg.box[0] && pg.box[3] > pg.box[1] && pg.n_edges) {
    FlutlPolygonEntry *pe = calloc (1, sizeof *pe);
    if (!pe)
      abort ();
    pe->edges = pg.edges;
    pe->n_edges = pg.n_edges;
    if (j->edge_acquire) {
      /* straight into page-locked memory the device reads: nobody copies the edges on the
       * thread that feeds the device (330 KB per cue of a 4K stream) */
      const size_t bytes = (size_t) pg.n_edges * 7 * sizeof (int32_t);
      void *blk = j->edge_acquire (j->edge_user, bytes);
      if (blk) {
        memcpy (blk, pg.edges, bytes);
        free (pg.edges);
        pe->edges = blk;
        pe->in_block = 1;
        pe->release = j->edge_release;
        pe->release_user = j->edge_user;
      }
    }
    memcpy (pe->box, pg.box, sizeof pe->box);
    pe->w = pg.box[2] - pg.box[0];
    pe->h = pg.box[3] - pg.box[1];
    pg.edges = NULL;
    j->poly = pe;
    j->x = pg.box[0];
    j->y = pg.box[1];
    j->w = pe->w;
    j->h = pe->h;
  }
  flutl_polygon_fini (&pg);
  return 1;
}

static void
stroke_job_run (gpointer data, gpointer user_data)
{
  StrokeJob *j = data;
  const gint64 t_run = g_get_monotonic_time ();
  if (j->to_polygon && stroke_job_polygon (j)) {
    __atomic_fetch_add (&stroke_jobs_polygons, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add (&stroke_jobs_done, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add (&stroke_jobs_us, g_get_monotonic_time () - t_run, __ATOMIC_RELAXED);
    __atomic_fetch_add (&stroke_jobs_queue_us, t_run - j->t_pushed, __ATOMIC_RELAXED);
    g_mutex_lock (&j->lock);
    j->done = 1;
    g_cond_broadcast (&j->cond);
    g_mutex_unlock (&j->lock);
    stroke_job_unref (j);
    return;
  }
  /* (a stroke the stroker declines -- a path of axis-aligned segments only, such as the frame
   * of a hex box, which Cairo strokes as boxes with another rasteriser -- or the device route
   * switched off: Cairo draws it, here on the worker) */
  __atomic_fetch_add (&masks_by_cairo_on_workers, 1, __ATOMIC_RELAXED);
  WorkerSurface *ws = g_private_get (&worker_surface_key);
  if (!ws) {
    ws = calloc (1, sizeof *ws);
    if (!ws)
      abort ();
    g_private_set (&worker_surface_key, ws);
  }
  if (!ws->s || ws->w != j->sw || ws->h != j->sh) {
    /* the surface is one more bound of the rasteriser (flutl_recorder_reserve):
     * the same size as the recorder's, no device offset */
    if (ws->s)
      cairo_surface_destroy (ws->s);
    ws->s = cairo_image_surface_create (CAIRO_FORMAT_A8, j->sw, j->sh);
    ws->w = j->sw;
    ws->h = j->sh;
  }
  cairo_t *cr = cairo_create (ws->s);
  cairo_append_path (cr, j->path);      /* (identity CTM: device coordinates) */
  cairo_set_line_width (cr, j->line_width);
  cairo_set_miter_limit (cr, j->miter_limit);
  if (j->exact_pending) {
    /* what raster_prepare does on the streaming thread in the synchronous path:
     * the stroke's extents in device space (identity CTM), exactly */
    double ex1, ey1, ex2, ey2;
    const gint64 t_e = g_get_monotonic_time ();
    cairo_stroke_extents (cr, &ex1, &ey1, &ex2, &ey2);
    raster_box_from_extents (&j->rb, ex1, ey1, ex2, ey2, j->sw, j->sh);
    __atomic_fetch_add (&stroke_jobs_extents_us, g_get_monotonic_time () - t_e, __ATOMIC_RELAXED);
  }
  if (!j->rb.inside) {          /* under the reference's limits, see current_limits */
    cairo_path_t *pth = cairo_copy_path (cr);
    cairo_new_path (cr);
    cairo_rectangle (cr, j->rb.lim[0], j->rb.lim[1], j->rb.lim[2] - j->rb.lim[0],
        j->rb.lim[3] - j->rb.lim[1]);
    cairo_clip (cr);
    cairo_append_path (cr, pth);
    cairo_path_destroy (pth);
  }
  cairo_set_matrix (cr, &j->ctm);
  cairo_set_source_rgba (cr, 1, 1, 1, 1);
  const gint64 t_d = g_get_monotonic_time ();
  if (j->rb.bx1 > j->rb.bx0 && j->rb.by1 > j->rb.by0)
    cairo_stroke (cr);
  cairo_destroy (cr);
  const gint64 t_c = g_get_monotonic_time ();
  j->tight = crop_coverage (ws->s, &j->rb, &j->x, &j->y, &j->w, &j->h);
  if (j->tight) {
    j->padded = mask_padded (j->tight, j->w, j->h, &j->padded_bytes);
    j->padded_hash = hash_bytes (j->padded, j->padded_bytes);
  }
  __atomic_fetch_add (&stroke_jobs_draw_us, t_c - t_d, __ATOMIC_RELAXED);
  __atomic_fetch_add (&stroke_jobs_crop_us, g_get_monotonic_time () - t_c, __ATOMIC_RELAXED);
  __atomic_fetch_add (&stroke_jobs_done, 1, __ATOMIC_RELAXED);
  __atomic_fetch_add (&stroke_jobs_us, g_get_monotonic_time () - t_run, __ATOMIC_RELAXED);
  __atomic_fetch_add (&stroke_jobs_queue_us, t_run - j->t_pushed, __ATOMIC_RELAXED);
  g_mutex_lock (&j->lock);
  j->done = 1;
  g_cond_broadcast (&j->cond);
  g_mutex_unlock (&j->lock);
  stroke_job_unref (j);
  (void) user_data;
}

static GThreadPool *stroke_pool;
static GMutex stroke_pool_lock;

/* The processors this process may keep busy: the machine's, or what its control group's CPU
 * bandwidth allows where that is less (a container given 16 of a 256-processor host's: 255
 * workers' worth of threads there would only take turns being throttled). */
static int
cpu_budget (void)
{
  int n = (int) g_get_num_processors ();
  FILE *f = fopen ("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    long long quota = 0, period = 0;
    char first[32] = "";
    if (fscanf (f, "%31s %lld", first, &period) == 2 && strcmp (first, "max") != 0 && period > 0) {
      quota = atoll (first);
      if (quota > 0) {
        const int by_quota = (int) ((quota + period - 1) / period);
        if (by_quota >= 1 && by_quota < n)
          n = by_quota;
      }
    }
    fclose (f);
  } else {
    /* (control groups v1) */
    long long quota = -1, period = 0;
    FILE *q = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
    FILE *p = fopen ("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (q && p && fscanf (q, "%lld", &quota) == 1 && fscanf (p, "%lld", &period) == 1 &&
        quota > 0 && period > 0) {
      const int by_quota = (int) ((quota + period - 1) / period);
      if (by_quota >= 1 && by_quota < n)
        n = by_quota;
    }
    if (q)
      fclose (q);
    if (p)
      fclose (p);
  }
  return n;
}


static GThreadPool *
stroke_pool_get (void)
{
  g_mutex_lock (&stroke_pool_lock);
  if (!stroke_pool) {
    const char *e = g_getenv ("FLUTL_STROKE_THREADS");
    int n = e ? atoi (e) : cpu_budget () - 1;
    n = n < 1 ? 1 : (n > 32 ? 32 : n);
    /* (exclusive: its threads are made now and stay.  A shared pool keeps at most two
     * idle threads of the process around -- GLib's max_unused_threads -- and on a stream that
     * needs five workers every now and then it made and ended threads all the time: a
     * pthread_create on the streaming thread per push, its mmap, mprotect and clone a fifth
     * of that thread's time in a region of new outlined text, tools/sprof.c) */
    stroke_pool = g_thread_pool_new (stroke_job_run, NULL, n, TRUE, NULL);
  }
  g_mutex_unlock (&stroke_pool_lock);
  return stroke_pool;
}

void
flutl_recorder_stroke_split (uint64_t *extents_us, uint64_t *draw_us, uint64_t *crop_us)
{
  *extents_us = (uint64_t) __atomic_exchange_n (&stroke_jobs_extents_us, 0, __ATOMIC_RELAXED);
  *draw_us = (uint64_t) __atomic_exchange_n (&stroke_jobs_draw_us, 0, __ATOMIC_RELAXED);
  *crop_us = (uint64_t) __atomic_exchange_n (&stroke_jobs_crop_us, 0, __ATOMIC_RELAXED);
}

void
flutl_recorder_run_stats (FlutlRecorder *rec, uint32_t *merged, uint32_t *clashed)
{
  *merged = rec->runs_merged;
  *clashed = rec->runs_clashed;
}

void
flutl_recorder_set_defer (FlutlRecorder *rec, int on)
{
  rec->defer = on && !rec->parent;
}

/* ---- composed outline paths ---------------------------------------------------------
 * What a new outline costs the streaming thread was, by a sampling profile of a stream
 * with new text in every cue (tools/sprof_stream.py): 8 % building the path on the
 * tracking context (one pango_cairo_glyph_string_path per glyph) and 42 % copying it out
 * again (cairo_copy_path: 100-150 KB allocated, written and freed per cue, every time
 * past the allocator's trim threshold: page faults and brk calls).  Neither is needed
 * where the path is glyphs under translations, the rule and not the exception
 * (gstttmlrender.c:1138-1161): Cairo appends a glyph's outline -- a fixed-point path of
 * its own, kept by the scaled font -- at the glyph's origin ROUNDED TO FIXED POINT
 * (_cairo_scaled_font_glyph_path: _cairo_path_fixed_append (path, glyph_path,
 * _cairo_fixed_from_double (x), _cairo_fixed_from_double (y))), an integer addition per
 * point; and what cairo-path-fixed.c filters out of a path (degenerate and collinear
 * line_tos) it has filtered out of the glyph's own path already, by differences that no
 * translation changes.  So the path is composed here: each glyph's outline taken from
 * Cairo ONCE (the same call at origin 0), then integer additions into a pooled buffer,
 * laid out as the cairo_path_t cairo_copy_path would have returned, element for element.
 * FLUTL_CHECK_PATHS=1 builds every such path both ways and aborts on any difference
 * (tests/test_element_boundary.py; FLUTL_CHECK_PATHS=1 tools/fuzz_ttml.py).
 * The path's extents (cairo_path_extents: points, and the exact bounds of curves whose
 * control points lie outside the box so far -- _cairo_box_add_curve_to) come the same way:
 * the box of each glyph's on-curve points moves with the glyph, and the few curves whose
 * control points leave that box are bounded again at their place by the same arithmetic
 * (flutl_stroker_box_add_curve; skipping a curve whose control points lie inside the box
 * never changes the box, so which curves Cairo happened to skip does not matter). */
typedef struct
{
  cairo_scaled_font_t *sf;
  unsigned long glyph;
} OutlineKey;

typedef struct
{
  OutlineKey key;
  int n_data;                   /* cairo_path_data_t elements (headers + points) */
  int32_t *el;                  /* per element: header {type, length} or point {x, y}, 24.8 */
  int32_t box[4];               /* of the on-curve points; n_data == 0: no outline */
  int n_flagged;
  int *flagged;                 /* element index of the CURVE_TO headers to bound again */
} GlyphOutline;

static gint64 composed_paths, composed_paths_checked;   /* statistics */

uint64_t
flutl_recorder_composed_paths (uint64_t *checked)
{
  if (checked)
    *checked = (uint64_t) composed_paths_checked;
  return (uint64_t) composed_paths;
}

static guint
outline_key_hash (gconstpointer p)
{
  const OutlineKey *k = p;
  return (guint) (((guintptr) k->sf >> 4) * 2654435761u) ^ (guint) (k->glyph * 40503u);
}

static gboolean
outline_key_equal (gconstpointer a, gconstpointer b)
{
  const OutlineKey *x = a, *y = b;
  return x->sf == y->sf && x->glyph == y->glyph;
}

static void
outline_free (gpointer p)
{
  GlyphOutline *o = p;
  free (o->el);
  free (o->flagged);
  free (o);
}

static int
pt_in_box (const int32_t box[4], int32_t x, int32_t y)
{
  return x >= box[0] && x <= box[2] && y >= box[1] && y <= box[3];
}

/* the glyph's outline as Cairo appends it at origin (0, 0): taken once, by the very call
 * raster_build_path makes */
static GlyphOutline *
outline_get (FlutlRecorder *rec, PangoFont *font, cairo_scaled_font_t *sf, PangoGlyph glyph)
{
  OutlineKey key = { sf, glyph };
  if (!rec->outlines)
    rec->outlines = g_hash_table_new_full (outline_key_hash, outline_key_equal, NULL, outline_free);
  GlyphOutline *o = g_hash_table_lookup (rec->outlines, &key);
  if (o)
    return o;
  cairo_t *cr = rec->cr;
  PangoGlyphInfo gi;
  memset (&gi, 0, sizeof gi);
  gi.glyph = glyph;
  PangoGlyphString one = { 1, &gi, NULL, 1 };
  cairo_save (cr);
  cairo_identity_matrix (cr);
  cairo_reset_clip (cr);
  cairo_new_path (cr);
  pango_cairo_glyph_string_path (cr, font, &one);
  cairo_path_t *path = cairo_copy_path (cr);
  cairo_new_path (cr);
  cairo_restore (cr);
  if (!path || path->status != CAIRO_STATUS_SUCCESS) {
    if (path)
      cairo_path_destroy (path);
    return NULL;
  }
  o = calloc (1, sizeof *o);
  if (!o)
    abort ();
  o->key = key;
  o->n_data = path->num_data;
  o->el = malloc ((size_t) (path->num_data ? path->num_data : 1) * 2 * sizeof (int32_t));
  o->flagged = malloc ((size_t) (path->num_data ? path->num_data : 1) * sizeof (int));
  if (!o->el || !o->flagged)
    abort ();
  o->box[0] = o->box[1] = INT32_MAX;
  o->box[2] = o->box[3] = INT32_MIN;
  int ok = 1;
  for (int i = 0; i < path->num_data && ok; i += path->data[i].header.length) {
    const cairo_path_data_t *d = &path->data[i];
    const int npt = d->header.length - 1;
    o->el[2 * i] = (int32_t) d->header.type;
    o->el[2 * i + 1] = d->header.length;
    for (int k = 1; k <= npt; k++) {
      /* (under the identity these are fixed-point values / 256, exactly) */
      const double fx = d[k].point.x * 256.0, fy = d[k].point.y * 256.0;
      if (fx != floor (fx) || fy != floor (fy) || fabs (fx) > 1e9 || fabs (fy) > 1e9)
        ok = 0;
      o->el[2 * (i + k)] = (int32_t) fx;
      o->el[2 * (i + k) + 1] = (int32_t) fy;
    }
    if (npt > 0) {              /* the element's last point is on the curve */
      const int32_t x = o->el[2 * (i + npt)], y = o->el[2 * (i + npt) + 1];
      o->box[0] = x < o->box[0] ? x : o->box[0];
      o->box[1] = y < o->box[1] ? y : o->box[1];
      o->box[2] = x > o->box[2] ? x : o->box[2];
      o->box[3] = y > o->box[3] ? y : o->box[3];
    }
  }
  for (int i = 0, prev_len = 0; i < path->num_data && ok; i += prev_len) {
    if (path->data[i].header.type == CAIRO_PATH_CURVE_TO &&
        (!pt_in_box (o->box, o->el[2 * (i + 1)], o->el[2 * (i + 1) + 1]) ||
         !pt_in_box (o->box, o->el[2 * (i + 2)], o->el[2 * (i + 2) + 1]))) {
      if (prev_len < 2)
        ok = 0;                 /* (a curve without a point before it: never from Cairo) */
      o->flagged[o->n_flagged++] = i;
    }
    prev_len = path->data[i].header.length;
  }
  cairo_path_destroy (path);
  if (!ok)
    o->n_data = -1;             /* (remembered: not composable) */
  key_font_pin (rec, sf);       /* (the key names the font by address) */
  g_hash_table_insert (rec->outlines, &o->key, o);
  return o;
}

/* path buffers go round: a cue's path is 100-150 KB, past the allocator's trim threshold */
static GMutex path_pool_lock;
static cairo_path_t *path_pool[16];
static int path_pool_cap[16], path_pool_n;

static cairo_path_t *
path_buffer_get (int n_data)
{
  cairo_path_t *p = NULL;
  int cap = 0;
  g_mutex_lock (&path_pool_lock);
  if (path_pool_n) {
    path_pool_n--;
    p = path_pool[path_pool_n];
    cap = path_pool_cap[path_pool_n];
  }
  g_mutex_unlock (&path_pool_lock);
  if (!p) {
    p = calloc (1, sizeof *p);
    if (!p)
      abort ();
  }
  if (cap < n_data) {
    cap = n_data + n_data / 2 + 64;
    free (p->data);
    p->data = malloc ((size_t) cap * sizeof (cairo_path_data_t));
    if (!p->data)
      abort ();
  }
  p->status = CAIRO_STATUS_SUCCESS;
  p->num_data = cap;            /* (capacity, until the composer sets the count) */
  return p;
}

static void
path_buffer_put (cairo_path_t *p, int cap)
{
  g_mutex_lock (&path_pool_lock);
  if (path_pool_n < 16) {
    path_pool[path_pool_n] = p;
    path_pool_cap[path_pool_n] = cap;
    path_pool_n++;
    p = NULL;
  }
  g_mutex_unlock (&path_pool_lock);
  if (p) {
    free (p->data);
    free (p);
  }
}

/* The pending ops as the path raster_build_path + cairo_copy_path would give, or NULL
 * where this file does not compose (anything but glyphs under translations whose origins
 * lie within the bounds Cairo keeps glyphs in).  ext: cairo_path_extents of it. */
static cairo_path_t *
compose_outline_path (FlutlRecorder *rec, const RasterBox *rb, double ext[4], int *cap_out)
{
  int ox, oy, total = 0;
  raster_origin (rec, &ox, &oy);
  ox -= rec->shift[0];
  oy -= rec->shift[1];
  if (rec->n_ops == 0)
    return NULL;
  /* pass 1: every op must be composable; how many elements */
  GlyphOutline **outs = rec->n_ops <= 2048 ? g_newa (GlyphOutline *, rec->n_ops) : NULL;
  if (!outs)
    return NULL;                /* (a path of thousands of glyphs: Cairo's way) */
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    const PathOp *op = &rec->ops[i];
    if (!op->font || op->sfont)
      return NULL;
    if (op->m.xx != 1.0 || op->m.yy != 1.0 || op->m.xy != 0.0 || op->m.yx != 0.0)
      return NULL;
    if (op->gi.glyph == PANGO_GLYPH_EMPTY || (op->gi.glyph & PANGO_GLYPH_UNKNOWN_FLAG))
      return NULL;
    cairo_scaled_font_t *sf = pango_cairo_font_get_scaled_font ((PangoCairoFont *) op->font);
    cairo_matrix_t fm;
    if (!sf)
      return NULL;
    /* (Cairo would pick another scaled font under a CTM whose linear part is not the
     * font's own; a font matrix with a translation moves the glyphs) */
    cairo_scaled_font_get_ctm (sf, &fm);
    if (fm.xx != 1.0 || fm.yy != 1.0 || fm.xy != 0.0 || fm.yx != 0.0)
      return NULL;
    cairo_scaled_font_get_font_matrix (sf, &fm);
    if (fm.x0 != 0.0 || fm.y0 != 0.0)
      return NULL;
    outs[i] = outline_get (rec, op->font, sf, op->gi.glyph);
    if (!outs[i] || outs[i]->n_data < 0)
      return NULL;
    total += outs[i]->n_data;
  }
  if (total == 0)
    return NULL;
  cairo_path_t *path = path_buffer_get (total);
  *cap_out = path->num_data;
  cairo_path_data_t *out = path->data;
  int32_t box[4] = { INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN };
  int n = 0;
  for (uint32_t i = 0; i < rec->n_ops; i++) {
    const PathOp *op = &rec->ops[i];
    const GlyphOutline *o = outs[i];
    /* pango_cairo_renderer: position = (no current point: 0, 0) + offsets; then
     * _cairo_gstate_transform_glyphs_to_backend under a translation: + (0 + x0 + 0) */
    double cx = (double) op->gi.geometry.x_offset / PANGO_SCALE;
    double cy = op->gi.geometry.y_offset == 0 ? 0.0 : (double) op->gi.geometry.y_offset / PANGO_SCALE;
    double mx = op->m.x0, my = op->m.y0;
    if (ox)
      mx -= ox;
    if (oy)
      my -= oy;
    if (mx != 0.0 || my != 0.0) {
      cx += mx;
      cy += my;
    }
    /* a glyph whose origin lies far outside surface x clip is dropped by Cairo: only
     * origins inside are composed */
    if (!(cx >= rb->lim[0] && cx <= rb->lim[2] && cy >= rb->lim[1] && cy <= rb->lim[3]) ||
        !(cx >= 0 && cx <= rec->sw && cy >= 0 && cy <= rec->sh)) {
      path_buffer_put (path, *cap_out);
      return NULL;
    }
    if (o->n_data == 0)
      continue;
    const int32_t fx = fixed_from_double (cx), fy = fixed_from_double (cy);
    const int32_t *el = o->el;
    const int base = n;
    for (int k = 0; k < o->n_data;) {
      const int len = el[2 * k + 1];
      out[n].header.type = (cairo_path_data_type_t) el[2 * k];
      out[n].header.length = len;
      for (int q = 1; q < len; q++) {
        out[n + q].point.x = (double) (el[2 * (k + q)] + fx) / 256.0;
        out[n + q].point.y = (double) (el[2 * (k + q) + 1] + fy) / 256.0;
      }
      n += len;
      k += len;
    }
    box[0] = o->box[0] + fx < box[0] ? o->box[0] + fx : box[0];
    box[1] = o->box[1] + fy < box[1] ? o->box[1] + fy : box[1];
    box[2] = o->box[2] + fx > box[2] ? o->box[2] + fx : box[2];
    box[3] = o->box[3] + fy > box[3] ? o->box[3] + fy : box[3];
    for (int f = 0; f < o->n_flagged; f++) {
      /* the curve's start is the last point of the element before it */
      const int k = o->flagged[f];
      int32_t p[8];
      p[0] = el[2 * (k - 1)] + fx;
      p[1] = el[2 * (k - 1) + 1] + fy;
      for (int q = 1; q <= 3; q++) {
        p[2 * q] = el[2 * (k + q)] + fx;
        p[2 * q + 1] = el[2 * (k + q) + 1] + fy;
      }
      flutl_stroker_box_add_curve (box, p);
    }
    (void) base;
  }
  path->num_data = n;
  __atomic_fetch_add (&composed_paths, 1, __ATOMIC_RELAXED);
  ext[0] = box[0] / 256.0;
  ext[1] = box[1] / 256.0;
  ext[2] = box[2] / 256.0;
  ext[3] = box[3] / 256.0;
  return path;
}

/* FLUTL_CHECK_PATHS: the composed path and its extents against what Cairo builds */
static void
composed_path_check (FlutlRecorder *rec, RasterBox *rb, const cairo_path_t *composed,
    const double ext[4])
{
  cairo_t *cr = rec->cr;
  RasterBox tmp = *rb;
  double e[4];
  raster_build_path (rec, &tmp);
  cairo_save (cr);
  cairo_identity_matrix (cr);
  cairo_path_extents (cr, &e[0], &e[1], &e[2], &e[3]);
  cairo_path_t *ref = cairo_copy_path (cr);
  cairo_restore (cr);
  cairo_new_path (cr);
  int bad = !ref || ref->status != CAIRO_STATUS_SUCCESS || ref->num_data != composed->num_data ||
      memcmp (tmp.lim, rb->lim, sizeof tmp.lim) != 0;
  if (!bad)
    for (int i = 0; i < ref->num_data && !bad; i += ref->data[i].header.length) {
      const cairo_path_data_t *a = &ref->data[i], *b = &composed->data[i];
      if (a->header.type != b->header.type || a->header.length != b->header.length)
        bad = 1;
      for (int k = 1; k < a->header.length && !bad; k++)
        if (a[k].point.x != b[k].point.x || a[k].point.y != b[k].point.y)
          bad = 1;
    }
  if (!bad && (e[0] != ext[0] || e[1] != ext[1] || e[2] != ext[2] || e[3] != ext[3]))
    bad = 2;
  if (bad) {
    fprintf (stderr, "flutl_recorder: FLUTL_CHECK_PATHS: composed outline path differs from Cairo's "
        "(%s; %d vs %d elements; extents %.10g %.10g %.10g %.10g vs %.10g %.10g %.10g %.10g; %u ops)\n",
        bad == 2 ? "extents" : "elements", composed->num_data, ref ? ref->num_data : -1,
        ext[0], ext[1], ext[2], ext[3], e[0], e[1], e[2], e[3], rec->n_ops);
    abort ();
  }
  cairo_path_destroy (ref);
  __atomic_fetch_add (&composed_paths_checked, 1, __ATOMIC_RELAXED);
}

/* The stroke of the pending ops, prepared here and drawn elsewhere: the job, or
 * NULL if there is nothing to draw (rb is valid either way).  Touches nothing the
 * resolving thread shares: called WITHOUT the recorder's lock -- building the path
 * and taking its exact extents is most of what a new stroke costs this thread. */
static StrokeJob *
stroke_defer_prepare (FlutlRecorder *rec, RasterBox *rb, int *exact)
{
  cairo_t *cr = rec->cr;
  double px1, py1, px2, py2;
  /* glyphs under translations: the path is composed, not built and copied out */
  cairo_path_t *composed = NULL;
  int composed_cap = 0;
  {
    double ext[4];
    current_limits (rec, rb->lim);
    if (!root_of (rec)->no_compose)
      composed = compose_outline_path (rec, rb, ext, &composed_cap);
    if (composed) {
      px1 = ext[0];
      py1 = ext[1];
      px2 = ext[2];
      py2 = ext[3];
      if (root_of (rec)->check_paths)
        composed_path_check (rec, rb, composed, ext);
    }
  }
  if (!composed)
    raster_build_path (rec, rb);
  /* cairo_stroke_extents runs the stroker over the whole path: 4x what building
   * the path costs, and the worker can do it.  What this thread must decide
   * now -- does the stroke meet its target at all (Cairo's "clear" bookkeeping,
   * stroke_emit), may it be reused under a translation -- follows from bounds of
   * the extents in all but contrived cases: the stroke covers its path and reaches
   * at most half a line width times the miter limit beyond it. */
  if (!composed) {
    cairo_save (cr);
    cairo_identity_matrix (cr);
    cairo_path_extents (cr, &px1, &py1, &px2, &py2);
    cairo_restore (cr);
  }
  const double reach = 0.5 * cairo_get_line_width (cr) * fmax (cairo_get_miter_limit (cr), 1.0) + 2.0;
  int gx, gy;
  raster_origin (rec, &gx, &gy);
  const int in0 = (int) floor (px1), in1 = (int) floor (py1), in2 = (int) ceil (px2),
      in3 = (int) ceil (py2);                 /* inside the extents */
  const int out0 = (int) floor (px1 - reach), out1 = (int) floor (py1 - reach),
      out2 = (int) ceil (px2 + reach), out3 = (int) ceil (py2 + reach);   /* around them */
  const int hit_in = px2 > px1 && py2 > py1 &&
      box_hits_target (rec, in0 + gx, in1 + gy, in2 + gx, in3 + gy);
  const int hit_out = box_hits_target (rec, out0 + gx, out1 + gy, out2 + gx, out3 + gy);
  *exact = 0;
  if (px2 <= px1 || py2 <= py1 || (hit_out && !hit_in) || !(cairo_get_line_width (cr) > 0.0)) {
    /* an empty path or pen, or the bounds disagree: the exact extents, here */
    double ex1, ey1, ex2, ey2;
    cairo_save (cr);
    cairo_identity_matrix (cr);
    if (composed) {             /* (rare: Cairo needs the path after all) */
      cairo_new_path (cr);
      cairo_append_path (cr, composed);
    }
    cairo_stroke_extents (cr, &ex1, &ey1, &ex2, &ey2);
    cairo_restore (cr);
    raster_box_from_extents (rb, ex1, ey1, ex2, ey2, rec->sw, rec->sh);
    *exact = 1;
  } else {
    /* hit_in: the extents meet the target because the path's box does; !hit_out:
     * they miss it because even the outer box does.  Either box gives stroke_emit
     * the right answer for this placement; the exact extents replace it when the
     * job is collected.  Reuse under translation waits for them (inside = 0 unless
     * even the outer box is clear of the limits). */
    const int *b = hit_in ? (const int[4]) { in0, in1, in2, in3 } : (const int[4]) { out0, out1, out2, out3 };
    rb->ext[0] = b[0];
    rb->ext[1] = b[1];
    rb->ext[2] = b[2];
    rb->ext[3] = b[3];
    rb->inside = out0 >= rb->lim[0] + 3 && out1 >= rb->lim[1] + 3 && out2 <= rb->lim[2] - 3 &&
        out3 <= rb->lim[3] - 3;
    rb->bx0 = imax (imax (out0 - 2, rb->lim[0]), 0);
    rb->by0 = imax (imax (out1 - 2, rb->lim[1]), 0);
    rb->bx1 = imin (imin (out2 + 2, rb->lim[2]), rec->sw);
    rb->by1 = imin (imin (out3 + 2, rb->lim[3]), rec->sh);
  }
  if (rb->bx1 <= rb->bx0 || rb->by1 <= rb->by0) {
    cairo_new_path (cr);
    if (composed)
      path_buffer_put (composed, composed_cap);
    return NULL;
  }
  StrokeJob *j = calloc (1, sizeof *j);
  if (!j)
    abort ();
  j->refs = 2;                  /* the cache entry's and the worker's */
  g_mutex_init (&j->lock);
  g_cond_init (&j->cond);
  cairo_get_matrix (cr, &j->ctm);
  j->line_width = cairo_get_line_width (cr);
  j->miter_limit = cairo_get_miter_limit (cr);
  if (composed) {
    j->path = composed;
    j->path_cap = composed_cap;
  } else {
    cairo_save (cr);
    cairo_identity_matrix (cr);
    j->path = cairo_copy_path (cr);
    cairo_restore (cr);
  }
  cairo_new_path (cr);
  if (!j->path || j->path->status != CAIRO_STATUS_SUCCESS)
    abort ();
  j->rb = *rb;
  j->exact_pending = !*exact;
  j->sw = rec->sw;
  j->sh = rec->sh;
  j->to_polygon = root_of (rec)->device_strokes;
  j->edge_acquire = root_of (rec)->edge_acquire;
  j->edge_release = root_of (rec)->edge_release;
  j->edge_user = root_of (rec)->edge_user;
  return j;
}

/* (with the lock) the prepared stroke becomes the entry's */
static void
stroke_defer_commit (FlutlRecorder *rec, StrokeEntry *e, const RasterBox *rb, StrokeJob *j)
{
  e->w = e->h = 0;
  memcpy (e->ext, rb->ext, sizeof e->ext);
  e->translatable = rb->inside;
  if (!j)
    return;
  memcpy (e->lim, rb->lim, sizeof e->lim);
  e->job = j;
  j->t_pushed = g_get_monotonic_time ();
  g_thread_pool_push (stroke_pool_get (), j, NULL);
  (void) rec;
}

/* The frame rectangle a stroke emitted now may reach: frame x clip x group. */
static int
effective_clip (FlutlRecorder *rec, int c[4])
{
  c[0] = -(1 << 20);
  c[1] = -(1 << 20);
  c[2] = 1 << 20;
  c[3] = 1 << 20;
  return clip_box (rec, &c[0], &c[1], &c[2], &c[3]);
}

/* Fill a MASK primitive in from its cache entry (the rule and the colour were
 * chosen when it was emitted).  Returns 0 if nothing of the mask is visible. */
static int
mask_prim_place (flutl_prim *p, const StrokeEntry *e, int ox, int oy, const int c[4])
{
  if (e->w == 0)
    return 0;
  ox += e->x;
  oy += e->y;
  int x0 = imax (ox, c[0]), y0 = imax (oy, c[1]);
  int x1 = imin (ox + e->w, c[2]), y1 = imin (oy + e->h, c[3]);
  if (x1 <= x0 || y1 <= y0)
    return 0;
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  p->param = (x0 > ox || x1 < ox + e->w) ? FLUTL_MASK_XCLIP : 0;
  p->a = (int32_t) e->offset;
  p->b = e->w + MASK_PAD;
  p->c = ox;
  p->d = oy;
  return 1;
}

const FlutlPatch *
flutl_recorder_patches (FlutlRecorder *rec, uint32_t *n)
{
  *n = rec->n_patches;
  return rec->patches;
}

/* Finish a frame's display list (a copy of the recorder's arrays, or the arrays
 * themselves): wait for the masks its placeholders stand for, store them, fill
 * the primitives in and give the regions their boxes -- the same results the
 * synchronous path produces.  Any thread; the recorder may be recording the
 * next frame meanwhile. */
void
flutl_recorder_resolve (FlutlRecorder *rec, flutl_region *regions, uint32_t n_regions,
    flutl_prim *prims, uint32_t n_prims, const FlutlPatch *patches, uint32_t n_patches)
{
  if (n_patches == 0)
    return;
  for (uint32_t k = 0; k < n_patches; k++) {
    const FlutlPatch *pt = &patches[k];
    StrokeEntry snap;
    StrokeJob *j;
    g_rec_mutex_lock (&rec->lock);
    j = rec->strokes[pt->entry].job;
    if (j)
      g_atomic_int_inc (&j->refs);
    g_rec_mutex_unlock (&rec->lock);
    if (j) {
      const gint64 t_w = g_get_monotonic_time ();
      stroke_job_wait (j);
      __atomic_fetch_add (&resolve_wait_us, g_get_monotonic_time () - t_w, __ATOMIC_RELAXED);
    }
    g_rec_mutex_lock (&rec->lock);
    StrokeEntry *e = &rec->strokes[pt->entry];    /* (the array may have moved) */
    if (j && e->job == j) {     /* first to get here: the mask goes into the atlas */
      /* (the worker's extents are the exact ones) */
      memcpy (e->ext, j->rb.ext, sizeof e->ext);
      e->translatable = j->rb.inside;
      if (j->poly) {
        /* the device makes the mask, of the polygon, when the entry's turn comes */
        e->x = j->x;
        e->y = j->y;
        e->w = j->w;
        e->h = j->h;
        e->offset = atlas_reserve_polygon (rec, j->poly) + MASK_PAD;
        j->poly = NULL;
      } else if (j->tight) {
        e->offset = atlas_put_dedup_hashed (rec, j->padded, j->padded_bytes, j->padded_hash) +
            MASK_PAD;
        e->x = j->x;
        e->y = j->y;
        e->w = j->w;
        e->h = j->h;
      }
      e->job = NULL;
      stroke_job_unref (j);     /* the entry's reference */
    }
    snap = *e;
    g_rec_mutex_unlock (&rec->lock);
    if (j)
      stroke_job_unref (j);     /* ours */
    if (pt->prim < n_prims) {
      const int c[4] = { pt->clip[0], pt->clip[1], pt->clip[2], pt->clip[3] };
      mask_prim_place (&prims[pt->prim], &snap, pt->ox, pt->oy, c);
    }
  }
  /* region boxes, as flutl_recorder_region_end makes them */
  for (uint32_t r = 0; r < n_regions; r++) {
    flutl_region *rg = &regions[r];
    int touched = 0;
    for (uint32_t k = 0; k < n_patches && !touched; k++)
      touched = patches[k].prim >= rg->first && patches[k].prim < rg->first + rg->count;
    if (!touched)
      continue;
    int x0 = 32767, y0 = 32767, x1 = 0, y1 = 0;
    for (uint32_t i = rg->first; i < rg->first + rg->count; i++) {
      const flutl_prim *p = &prims[i];
      if (p->cx1 <= p->cx0 || p->cy1 <= p->cy0)
        continue;
      x0 = imin (x0, p->cx0);
      y0 = imin (y0, p->cy0);
      x1 = imax (x1, p->cx1);
      y1 = imax (y1, p->cy1);
    }
    if (x1 <= x0 || y1 <= y0) {         /* nothing visible after all */
      x0 = y0 = x1 = y1 = 0;
      rg->count = 0;
    }
    for (uint32_t i = rg->first; i < rg->first + rg->count; i++) {
      flutl_prim *p = &prims[i];
      if (p->cx1 <= p->cx0 || p->cy1 <= p->cy0) {
        p->cx0 = p->cx1 = (int16_t) x0;
        p->cy0 = p->cy1 = (int16_t) y0;
      }
    }
    rg->x0 = (int16_t) x0;
    rg->y0 = (int16_t) y0;
    rg->x1 = (int16_t) x1;
    rg->y1 = (int16_t) y1;
  }
}

/* The cache entry of one stroke / fill of the ops [rec->ops, rec->ops + rec->n_ops).
 * *dx, *dy = where the mask goes relative to where it was made.  A new entry
 * (*is_new) has no mask yet: the caller rasterises.  (The pointer is valid
 * until the next call.) */
static StrokeEntry *
ops_find_mask (FlutlRecorder *rec, int fill, int *dx, int *dy, int *is_new)
{
  /* a fill does not depend on the line width: key it with a marker instead */
  double lw = fill ? -1.0 : cairo_get_line_width (rec->cr);
  int ox, oy;
  raster_origin (rec, &ox, &oy);
  ox -= rec->shift[0];
  oy -= rec->shift[1];
  /* base in raster coordinates (the key itself only holds offsets from it) */
  int base_x = (int) floor (rec->ops[0].m.x0) - ox, base_y = (int) floor (rec->ops[0].m.y0) - oy;
  uint8_t *key;
  size_t key_len = stroke_key (rec, lw, base_x + ox, base_y + oy, &key);
  uint64_t h = hash_bytes (key, key_len);
  *is_new = 0;
  /* (entries of one hash are met in the order they were made, as the list was walked) */
  for (uint32_t pos = (uint32_t) h; rec->stroke_index_cap; pos++) {
    const uint32_t v = rec->stroke_index[pos & (rec->stroke_index_cap - 1)];
    if (!v)
      break;
    StrokeEntry *e = &rec->strokes[v - 1];
    if (e->hash == h && e->key_len == key_len && memcmp (e->key, key, key_len) == 0) {
      const int tx = base_x - e->base_x, ty = base_y - e->base_y;
      /* A mask made clear of its limits is the stroke at any integer
       * translation that stays clear of the limits in force there; one that
       * crosses them is only good at the same place under the same limits. */
      int lim[4];
      current_limits (rec, lim);
      const int same = tx == 0 && ty == 0 && memcmp (lim, e->lim, sizeof lim) == 0;
      /* (a mask still in the making has no box yet: good at its own place only) */
      const int clear = !e->job && (e->w == 0 ||
          (e->x + tx >= lim[0] + 2 && e->y + ty >= lim[1] + 2 &&
           e->x + tx + e->w <= lim[2] - 2 && e->y + ty + e->h <= lim[3] - 2));
      if (same || (e->translatable && clear && ty == 0)) {
        free (key);
        *dx = tx;
        *dy = ty;
        return e;
      }
    }
  }
  if (rec->n_strokes == rec->cap_strokes) {
    rec->cap_strokes = rec->cap_strokes ? rec->cap_strokes * 2 : 32;
    rec->strokes = xrealloc (rec->strokes, rec->cap_strokes * sizeof (StrokeEntry));
  }
  StrokeEntry *e = &rec->strokes[rec->n_strokes++];
  memset (e, 0, sizeof *e);
  key_fonts_pin_ops (rec);
  e->hash = h;
  e->key = key;
  e->key_len = key_len;
  e->base_x = base_x;
  e->base_y = base_y;
  *dx = *dy = 0;
  *is_new = 1;
  /* into the index: grown to at least twice the entries, everything put in again */
  if (rec->n_strokes * 2 > rec->stroke_index_cap) {
    uint32_t cap = rec->stroke_index_cap ? rec->stroke_index_cap * 2 : 256;
    while (cap < rec->n_strokes * 2)
      cap *= 2;
    free (rec->stroke_index);
    rec->stroke_index = calloc (cap, sizeof *rec->stroke_index);
    if (!rec->stroke_index)
      abort ();
    rec->stroke_index_cap = cap;
    for (uint32_t i = 0; i + 1 < rec->n_strokes; i++) {
      uint32_t pos = (uint32_t) rec->strokes[i].hash;
      while (rec->stroke_index[pos & (cap - 1)])
        pos++;
      rec->stroke_index[pos & (cap - 1)] = i + 1;
    }
  }
  {
    uint32_t pos = (uint32_t) h;
    while (rec->stroke_index[pos & (rec->stroke_index_cap - 1)])
      pos++;
    rec->stroke_index[pos & (rec->stroke_index_cap - 1)] = rec->n_strokes;
  }
  return e;
}

static StrokeEntry *
ops_get_mask (FlutlRecorder *rec, int fill, int *dx, int *dy)
{
  int is_new;
  StrokeEntry *e = ops_find_mask (rec, fill, dx, dy, &is_new);
  if (is_new)
    e->translatable = stroke_rasterise (rec, e, fill);
  return e;
}

/* cairo_fill of the pending path with the mask made by the device: the path as the polygon
 * Cairo makes of it (flutl_fill_to_polygon: every sub-path closed, curves flattened -- no
 * stroker involved, some tens of microseconds on this thread), scan converted like an
 * outline's.  What the reference fills: anamorphic glyphs (gstttmlrender.c:1109-1118) and,
 * through Pango, the line-break arrow.  A path the polygon maker declines (axis-aligned
 * segments only: Cairo fills those as boxes) is rasterised by Cairo as before. */
static int
fill_by_device (FlutlRecorder *rec, StrokeEntry *e)
{
  cairo_t *cr = rec->cr;
  RasterBox rb;
  raster_prepare (rec, 1, &rb);
  e->w = e->h = 0;
  memcpy (e->ext, rb.ext, sizeof e->ext);
  if (rb.bx1 <= rb.bx0 || rb.by1 <= rb.by0) {
    cairo_new_path (cr);
    return rb.inside;
  }
  cairo_save (cr);
  cairo_identity_matrix (cr);
  cairo_path_t *pth = cairo_copy_path (cr);
  cairo_restore (cr);
  const int target[4] = { imax (rb.lim[0], 0), imax (rb.lim[1], 0), imin (rb.lim[2], rec->sw),
    imin (rb.lim[3], rec->sh) };
  FlutlPolygon pg;
  flutl_polygon_init (&pg);
  const int made = pth && pth->status == CAIRO_STATUS_SUCCESS &&
      flutl_fill_to_polygon (pth, cairo_get_tolerance (cr), target, &pg) == FLUTL_STROKER_OK;
  if (pth)
    cairo_path_destroy (pth);
  if (!made) {
    flutl_polygon_fini (&pg);
    return stroke_draw (rec, e, &rb, 1);
  }
  cairo_new_path (cr);
  memcpy (e->lim, rb.lim, sizeof e->lim);
  if (pg.box[2] > pg.box[0] && pg.box[3] > pg.box[1] && pg.n_edges) {
    FlutlPolygonEntry *pe = calloc (1, sizeof *pe);
    if (!pe)
      abort ();
    pe->edges = pg.edges;
    pe->n_edges = pg.n_edges;
    memcpy (pe->box, pg.box, sizeof pe->box);
    pe->w = pg.box[2] - pg.box[0];
    pe->h = pg.box[3] - pg.box[1];
    pg.edges = NULL;
    e->x = pe->box[0];
    e->y = pe->box[1];
    e->w = pe->w;
    e->h = pe->h;
    e->offset = atlas_reserve_polygon (rec, pe) + MASK_PAD;
    __atomic_fetch_add (&fills_by_device, 1, __ATOMIC_RELAXED);
  }
  flutl_polygon_fini (&pg);
  return rb.inside;
}

static void
ops_to_mask (FlutlRecorder *rec, int fill)
{
  int dx, dy, is_new;
  GRecMutex *lk = &root_of (rec)->lock;
  /* (the stroke cache is shared with the thread that resolves frames) */
  g_rec_mutex_lock (lk);
  StrokeEntry *e = ops_find_mask (rec, fill, &dx, &dy, &is_new);
  if (is_new && rec->defer && !fill) {
    const uint32_t idx = (uint32_t) (e - rec->strokes);
    RasterBox rb;
    g_rec_mutex_unlock (lk);
    int exact;
    StrokeJob *j = stroke_defer_prepare (rec, &rb, &exact);
    g_rec_mutex_lock (lk);
    e = &rec->strokes[idx];     /* (only this thread adds entries: it has not moved) */
    stroke_defer_commit (rec, e, &rb, j);
  } else if (is_new && !fill && root_of (rec)->device_strokes) {
    /* not deferred (no pipeline: FLUTL_SYNC, the request-pad mode, a blurred outline's
     * temporary surface), but the device still makes the mask: the stroke's polygon right
     * here -- a tenth of what Cairo's rasteriser would take on this thread */
    const uint32_t idx = (uint32_t) (e - rec->strokes);
    RasterBox rb;
    int exact;
    g_rec_mutex_unlock (lk);
    StrokeJob *j = stroke_defer_prepare (rec, &rb, &exact);
    const int made = j && stroke_job_polygon (j);
    g_rec_mutex_lock (lk);
    e = &rec->strokes[idx];
    if (j && !made) {
      /* a stroke the stroker does not take: Cairo draws it (the ops are still pending) */
      stroke_job_unref (j);
      stroke_job_unref (j);
      __atomic_fetch_add (&masks_by_cairo, 1, __ATOMIC_RELAXED);
      e->translatable = stroke_rasterise (rec, e, fill);
    } else {
      e->w = e->h = 0;
      memcpy (e->ext, j ? j->rb.ext : rb.ext, sizeof e->ext);
      e->translatable = j ? j->rb.inside : rb.inside;
      memcpy (e->lim, rb.lim, sizeof e->lim);
      if (j && j->poly) {
        e->x = j->x;
        e->y = j->y;
        e->w = j->w;
        e->h = j->h;
        e->offset = atlas_reserve_polygon (rec, j->poly) + MASK_PAD;
        j->poly = NULL;
      }
      if (j) {
        stroke_job_unref (j);   /* (made with two references: the entry's and a worker's) */
        stroke_job_unref (j);
      }
    }
  } else if (is_new && fill && root_of (rec)->device_strokes) {
    e->translatable = fill_by_device (rec, e);
  } else if (is_new) {
    __atomic_fetch_add (&masks_by_cairo, 1, __ATOMIC_RELAXED);
    e->translatable = stroke_rasterise (rec, e, fill);
  }
  stroke_emit (rec, e, dx, dy);
  g_rec_mutex_unlock (lk);
}

/* Device rows an op's stroke can touch: its outline's bounding box under the
 * op's matrix, grown by what the pen can reach (half the line width times
 * Cairo's default miter limit of 10) and 2 px of antialiasing. */
static int
op_row_range (const PathOp *op, double reach, double *lo, double *hi)
{
  double x0, y0, x1, y1;
  if (op->sfont)
    return 0;
  if (op->font) {
    PangoRectangle ink;
    pango_font_get_glyph_extents (op->font, op->gi.glyph, &ink, NULL);
    x0 = (ink.x + op->gi.geometry.x_offset) / (double) PANGO_SCALE;
    y0 = (ink.y + op->gi.geometry.y_offset) / (double) PANGO_SCALE;
    x1 = x0 + ink.width / (double) PANGO_SCALE;
    y1 = y0 + ink.height / (double) PANGO_SCALE;
    /* hinted outlines may exceed the unhinted ink box by a pixel */
    x0 -= 1.0; y0 -= 1.0; x1 += 1.0; y1 += 1.0;
  } else {
    x0 = op->x; y0 = op->y; x1 = op->x + op->w; y1 = op->y + op->h;
  }
  double ys[4] = { op->m.yx * x0 + op->m.yy * y0, op->m.yx * x1 + op->m.yy * y0,
    op->m.yx * x0 + op->m.yy * y1, op->m.yx * x1 + op->m.yy * y1 };
  *lo = *hi = ys[0];
  for (int k = 1; k < 4; k++) {
    if (ys[k] < *lo) *lo = ys[k];
    if (ys[k] > *hi) *hi = ys[k];
  }
  *lo += op->m.y0 - reach;
  *hi += op->m.y0 + reach;
  return 1;
}

/* cairo_stroke / cairo_fill of the pending path: one mask, as drawn.
 *
 * FLUTL_LINE_SPLIT=1 (off by default, an experiment kept for the record): a
 * stroke whose contours fall into groups of disjoint device rows (the lines
 * of a paragraph) is rasterised and cached group by group, the cache misses
 * on worker threads -- 1.5x on streams with new outlined text in every cue.
 * It is NOT exact: tools/fuzz_ttml.py found a strike-through rectangle whose
 * edge rows come out a few units different when it is stroked without the
 * glyphs that the reference has pending in the same cairo_stroke, six rows
 * above it.  Cairo's scan converter is not row-local after all. */
static void
path_to_mask (FlutlRecorder *rec, int fill)
{
  if (rec->n_ops == 0)
    return;
  if (rec->color_clear) {
    path_ops_clear (rec);
    return;
  }
  uint32_t n = rec->n_ops;
  int split = !fill && n >= 2 && rec->line_split;
  double *lo = NULL, *hi = NULL;
  uint32_t *grp = NULL;
  uint32_t n_groups = 1;
  if (split) {
    /* tentative grouping: pen radius + antialiasing, no miter allowance */
    const double reach = cairo_get_line_width (rec->cr) * 0.5 + 2.0;
    lo = malloc (sizeof (double) * n * 2);
    grp = malloc (sizeof (uint32_t) * n);
    if (!lo || !grp)
      abort ();
    hi = lo + n;
    for (uint32_t i = 0; i < n && split; i++)
      split = op_row_range (&rec->ops[i], reach, &lo[i], &hi[i]);
  }
  if (split) {
    /* connected components of overlapping row ranges (n is small: O(n^2)) */
    for (uint32_t i = 0; i < n; i++)
      grp[i] = i;
    for (int changed = 1; changed;) {
      changed = 0;
      for (uint32_t i = 0; i < n; i++)
        for (uint32_t j = i + 1; j < n; j++)
          if (grp[i] != grp[j] && lo[i] <= hi[j] && lo[j] <= hi[i]) {
            /* merge: both take the smaller label and the union of the ranges */
            uint32_t a = grp[i] < grp[j] ? grp[i] : grp[j], b = grp[i] ^ grp[j] ^ a;
            double l = lo[i] < lo[j] ? lo[i] : lo[j], h = hi[i] > hi[j] ? hi[i] : hi[j];
            for (uint32_t k = 0; k < n; k++)
              if (grp[k] == b || grp[k] == a) {
                grp[k] = a;
                if (lo[k] > l) lo[k] = l;
                if (hi[k] < h) hi[k] = h;
              }
            changed = 1;
          }
    }
    n_groups = 0;
    for (uint32_t i = 0; i < n; i++)
      if (grp[i] == i)
        n_groups++;
  }
  if (n_groups <= 1) {
    ops_to_mask (rec, fill);
  } else {
    /* stable partition by group label, then one mask per group */
    PathOp *all = rec->ops, *sorted = malloc (sizeof (PathOp) * n);
    uint32_t *idx = malloc (sizeof (uint32_t) * n_groups);   /* entry indices: the */
    int *off = malloc (sizeof (int) * 2 * n_groups);         /* array may be moved  */
    if (!sorted || !idx || !off)
      abort ();
    uint32_t pos = 0, k = 0;
    /* pass 1: cache lookups; a miss gets its entry, its path (copied) and its
     * box, but is not drawn yet */
    StrokeEntry **miss = calloc (n_groups, sizeof *miss);
    RasterBox *mbox = calloc (n_groups, sizeof *mbox);
    cairo_path_t **mpath = calloc (n_groups, sizeof *mpath);
    if (!miss || !mbox || !mpath)
      abort ();
    for (uint32_t g = 0; g < n; g++) {
      if (grp[g] != g)
        continue;
      uint32_t first = pos;
      for (uint32_t i = 0; i < n; i++)
        if (grp[i] == g)
          sorted[pos++] = all[i];
      rec->ops = sorted + first;
      rec->n_ops = pos - first;
      int is_new = 0;
      StrokeEntry *e = ops_find_mask (rec, fill, &off[2 * k], &off[2 * k + 1], &is_new);
      idx[k] = (uint32_t) (e - rec->strokes);
      if (is_new) {
        raster_prepare (rec, fill, &mbox[k]);
        e->w = e->h = 0;
        e->translatable = mbox[k].inside;
        memcpy (e->lim, mbox[k].lim, sizeof e->lim);
        if (mbox[k].bx1 > mbox[k].bx0 && mbox[k].by1 > mbox[k].by0) {
          cairo_save (rec->cr);
          cairo_identity_matrix (rec->cr);      /* copy in device coordinates */
          mpath[k] = cairo_copy_path (rec->cr);
          cairo_restore (rec->cr);
          miss[k] = e;                          /* (index kept in idx[k]) */
        }
        cairo_new_path (rec->cr);
      }
      k++;
    }
    /* pass 2: draw the misses -- in parallel if their boxes (generous: Cairo's
     * stroke extents include the miter allowance) are apart in rows, else one
     * after the other on this thread */
    {
      uint32_t n_miss = 0;
      int parallel = raster_thread_count () > 1;
      for (uint32_t i = 0; i < n_groups; i++)
        if (miss[i]) {
          n_miss++;
          for (uint32_t j = i + 1; j < n_groups; j++)
            if (miss[j] && mbox[i].by0 < mbox[j].by1 && mbox[j].by0 < mbox[i].by1)
              parallel = 0;
          if (!mpath[i] || mpath[i]->status != CAIRO_STATUS_SUCCESS)
            abort ();
        }
      if (n_miss) {
        cairo_surface_flush (rec->scratch);
        uint8_t *data = cairo_image_surface_get_data (rec->scratch);
        int stride = cairo_image_surface_get_stride (rec->scratch);
        GMutex lock;
        GCond cond;
        int pending = (int) n_miss;
        LineJob *jobs = calloc (n_groups, sizeof *jobs);
        if (!jobs)
          abort ();
        parallel = parallel && n_miss > 1;
        g_mutex_init (&lock);
        g_cond_init (&cond);
        uint32_t last = 0;
        for (uint32_t i = 0; i < n_groups; i++)
          if (miss[i]) {
            jobs[i] = (LineJob) { mpath[i], cairo_get_line_width (rec->cr), data, stride,
              rec->width, mbox[i].by0, mbox[i].by1, !mbox[i].inside,
              { mbox[i].lim[0], mbox[i].lim[1], mbox[i].lim[2], mbox[i].lim[3] },
              parallel ? &lock : NULL, &cond, &pending };
            last = i;
          }
        for (uint32_t i = 0; i < n_groups; i++)
          if (miss[i]) {
            if (parallel && i != last)
              g_thread_pool_push (raster_pool, &jobs[i], NULL);
            else
              line_job_run (&jobs[i], NULL);    /* this thread takes one too */
          }
        if (parallel) {
          g_mutex_lock (&lock);
          while (pending > 0)
            g_cond_wait (&cond, &lock);
          g_mutex_unlock (&lock);
        }
        g_mutex_clear (&lock);
        g_cond_clear (&cond);
        free (jobs);
        for (uint32_t i = 0; i < n_groups; i++)
          if (miss[i]) {
            cairo_surface_mark_dirty_rectangle (rec->scratch, 0, mbox[i].by0, rec->width,
                mbox[i].by1 - mbox[i].by0);
            raster_crop (rec, &rec->strokes[idx[i]], &mbox[i]);
            cairo_path_destroy (mpath[i]);
          }
      }
    }
    free (miss);
    free (mbox);
    free (mpath);
    memcpy (all, sorted, sizeof (PathOp) * n);    /* same refs, new order */
    free (sorted);
    rec->ops = all;
    rec->n_ops = n;
    /* the check: covered rows of the masks, one row of margin, pairwise apart */
    int disjoint = 1;
    for (uint32_t i = 0; i < n_groups && disjoint; i++)
      for (uint32_t j = i + 1; j < n_groups && disjoint; j++) {
        const StrokeEntry *a = &rec->strokes[idx[i]], *b = &rec->strokes[idx[j]];
        if (a->w == 0 || b->w == 0)
          continue;
        int a0 = a->y + off[2 * i + 1] - 1, a1 = a->y + off[2 * i + 1] + a->h + 1;
        int b0 = b->y + off[2 * j + 1] - 1, b1 = b->y + off[2 * j + 1] + b->h + 1;
        if (a0 < b1 && b0 < a1)
          disjoint = 0;
      }
    if (disjoint)
      for (uint32_t i = 0; i < n_groups; i++)
        stroke_emit (rec, &rec->strokes[idx[i]], off[2 * i], off[2 * i + 1]);
    else
      ops_to_mask (rec, fill);                    /* one piece, as drawn */
    free (idx);
    free (off);
  }
  free (lo);
  free (grp);
  path_ops_clear (rec);
}

void
flutl_recorder_stroke (FlutlRecorder *rec)
{
  path_to_mask (rec, 0);
}

/* cairo_fill of the pending path (anamorphic glyphs, :1109-1117): coverage from
 * the polygon scan converter, same rule selection as a stroke. */
void
flutl_recorder_fill (FlutlRecorder *rec)
{
  path_to_mask (rec, 1);
}

/* ---- images ---------------------------------------------------------------- */

void
flutl_recorder_paint_image (FlutlRecorder *rec, cairo_surface_t *img, int x, int y)
{
  if (cairo_surface_get_type (img) != CAIRO_SURFACE_TYPE_IMAGE) {
    rec->unsupported++;
    return;
  }
  cairo_format_t fmt = cairo_image_surface_get_format (img);
  if (fmt != CAIRO_FORMAT_ARGB32 && fmt != CAIRO_FORMAT_RGB24) {
    rec->unsupported++;
    return;
  }
  ImageEntry *e = NULL;
  for (uint32_t i = 0; i < rec->n_images; i++)
    if (rec->images[i].surface == img)
      e = &rec->images[i];
  if (!e) {
    int w = cairo_image_surface_get_width (img);
    int h = cairo_image_surface_get_height (img);
    int stride = cairo_image_surface_get_stride (img);
    cairo_surface_flush (img);
    const uint8_t *data = cairo_image_surface_get_data (img);
    uint32_t *tight = malloc ((size_t) w * h * 4);
    if (!tight)
      abort ();
    for (int r = 0; r < h; r++) {
      memcpy (tight + (size_t) r * w, data + (size_t) r * stride, (size_t) w * 4);
      if (fmt == CAIRO_FORMAT_RGB24)
        for (int c = 0; c < w; c++)
          tight[(size_t) r * w + c] |= 0xff000000u;
    }
    if (rec->n_images == rec->cap_images) {
      rec->cap_images = rec->cap_images ? rec->cap_images * 2 : 8;
      rec->images = xrealloc (rec->images, rec->cap_images * sizeof (ImageEntry));
    }
    e = &rec->images[rec->n_images++];
    e->surface = cairo_surface_reference (img);
    e->w = w;
    e->h = h;
    e->offset = atlas_put (rec, tight, (size_t) w * h * 4);
    free (tight);
  }
  int x0 = x, y0 = y, x1 = x + e->w, y1 = y + e->h;
  if (box_hits_target (rec, x0, y0, x1, y1))
    target_touched (rec);
  if (!clip_box (rec, &x0, &y0, &x1, &y1))
    return;
  flutl_prim *p = push_prim (rec);
  p->op = FLUTL_OP_IMAGE;
  p->cx0 = (int16_t) x0;
  p->cy0 = (int16_t) y0;
  p->cx1 = (int16_t) x1;
  p->cy1 = (int16_t) y1;
  p->a = (int32_t) e->offset;
  p->b = e->w * 4;
  p->c = x;
  p->d = y;
}

/* ---- region replay (the element's region cache) --------------------------------
 * A region whose inputs have not changed since an earlier buffer records the
 * same primitives: the element keeps them and plays them back instead of walking
 * the layouts again.  mark: where the next region's output will start; snapshot:
 * what a region recorded since its mark (prims, its region record if it got one,
 * whether a placeholder is among them -- then it is not final); replay: append a
 * snapshot as this frame's next region. */
void
flutl_recorder_mark (FlutlRecorder *rec, uint32_t *prim_mark, uint32_t *region_mark)
{
  *prim_mark = rec->n_prims;
  *region_mark = rec->n_regions;
}

int
flutl_recorder_frame_is_clear (FlutlRecorder *rec)
{
  return rec->frame_clear;
}

const flutl_prim *
flutl_recorder_snapshot (FlutlRecorder *rec, uint32_t prim_mark, uint32_t region_mark,
    uint32_t *n_prims, const flutl_region **region, int *has_placeholder)
{
  *n_prims = rec->n_prims - prim_mark;
  *region = rec->n_regions > region_mark ? &rec->regions[region_mark] : NULL;
  *has_placeholder = rec->region_provisional;
  for (uint32_t k = 0; k < rec->n_patches; k++)
    if (rec->patches[k].prim >= prim_mark)
      *has_placeholder = 1;
  return rec->prims + prim_mark;
}

void
flutl_recorder_replay (FlutlRecorder *rec, const flutl_prim *prims, uint32_t n_prims,
    const flutl_region *region, int frame_clear_after, uint32_t unsupported)
{
  const uint32_t first = rec->n_prims;
  if (rec->n_prims + n_prims > rec->cap_prims) {
    while (rec->n_prims + n_prims > rec->cap_prims)
      rec->cap_prims = rec->cap_prims ? rec->cap_prims * 2 : 256;
    rec->prims = xrealloc (rec->prims, rec->cap_prims * sizeof (flutl_prim));
    rec->prim_tags = xrealloc (rec->prim_tags, rec->cap_prims * sizeof (uint16_t));
  }
  if (n_prims) {                /* (a region that recorded nothing replays nothing) */
    memcpy (rec->prims + first, prims, n_prims * sizeof (flutl_prim));
    memset (rec->prim_tags + first, 0, n_prims * sizeof (uint16_t));
  }
  rec->n_prims += n_prims;
  if (region) {
    if (rec->n_regions == rec->cap_regions) {
      rec->cap_regions = rec->cap_regions ? rec->cap_regions * 2 : 32;
      rec->regions = xrealloc (rec->regions, rec->cap_regions * sizeof (flutl_region));
    }
    flutl_region *r = &rec->regions[rec->n_regions++];
    *r = *region;
    r->first = first;
    r->count = n_prims;
  }
  rec->frame_clear = frame_clear_after;
  rec->unsupported += unsupported;
}

/* ---- results ---------------------------------------------------------------- */

const flutl_region *
flutl_recorder_regions (FlutlRecorder *rec, uint32_t *n)
{
  *n = rec->n_regions;
  return rec->regions;
}

const flutl_prim *
flutl_recorder_prims (FlutlRecorder *rec, uint32_t *n)
{
  run_flush (rec);
  *n = rec->n_prims;
  return rec->prims;
}

const uint8_t *
flutl_recorder_atlas (FlutlRecorder *rec, size_t *size)
{
  /* [0, atlas_size) is data, followed by ATLAS_SLACK readable zero bytes */
  size_t need = rec->atlas_size + ATLAS_SLACK;
  atlas_ensure (rec, need);
  *size = need;
  return rec->atlas;
}

uint32_t
flutl_recorder_new_atlas_entries (FlutlRecorder *rec, const uint32_t **offsets,
    const uint32_t **sizes, FlutlPolygonEntry *const **polygons)
{
  uint32_t n = rec->n_new;
  *offsets = rec->new_off;
  *sizes = rec->new_size;
  *polygons = rec->new_poly;
  rec->n_new = 0;               /* (the polygons are the caller's now) */
  return n;
}

void
flutl_polygon_entry_free (FlutlPolygonEntry *pe)
{
  if (!pe)
    return;
  if (!pe->in_block)
    free (pe->edges);
  else if (!pe->submitted && pe->release)
    pe->release (pe->release_user, pe->edges);
  free (pe);
}

void
flutl_recorder_set_edge_blocks (FlutlRecorder *rec, void *(*acquire) (void *user, size_t bytes),
    void (*release) (void *user, void *block), void *user)
{
  rec = root_of (rec);
  rec->edge_acquire = acquire;
  rec->edge_release = release;
  rec->edge_user = user;
}

void
flutl_recorder_set_device_strokes (FlutlRecorder *rec, int on)
{
  rec->device_strokes = on && !rec->parent;
}

uint32_t
flutl_recorder_unsupported (FlutlRecorder *rec)
{
  return rec->unsupported;
}
