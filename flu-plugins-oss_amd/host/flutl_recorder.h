/*
 * flutl_recorder.h -- host-side display-list recorder of the fluttml-hip
 * render element.
 *
 * The reference draws every primitive straight into a Cairo image surface
 * (plugins/ttml/gstttmlrender.c:1235-1385 show_regions, :941-1176 show_layout,
 * :1180-1232 render_outline).  This recorder offers the same small set of
 * drawing verbs, but RECORDS them as flutl_prim entries (include/flutl.h) with
 * the clip, the group surface bounds, the colour quantisation and the blend
 * rule already resolved, so that the device kernel only does pixel arithmetic.
 *
 * Cairo is used here for two things only, never for pixels of the frame:
 *   - a scratch A8 surface + cairo_t that tracks the CTM with Cairo's own
 *     matrix arithmetic and accumulates outline paths;
 *   - rasterising A8 masks (glyphs through the scaled-font glyph cache,
 *     strokes through the stroker + scan converter) into the atlas.
 */
#ifndef FLUTL_RECORDER_H
#define FLUTL_RECORDER_H

#include <cairo.h>
#include <pango/pangocairo.h>
#include <stdint.h>
#include "flutl.h"

typedef struct FlutlRecorder FlutlRecorder;

FlutlRecorder *flutl_recorder_new (void);
void flutl_recorder_free (FlutlRecorder *rec);

/* Drop the atlas and the glyph cache (gst_ttmlrender_reset, :1610-1621). */
void flutl_recorder_reset (FlutlRecorder *rec);

/* Start a new frame of width x height; keeps atlas + glyph cache. */
void flutl_recorder_begin_frame (FlutlRecorder *rec, int width, int height);
void flutl_recorder_reserve (FlutlRecorder *rec, int w, int h);

/* The tracking context: callers issue cairo_save/translate/restore and path
 * calls on it exactly as the reference does on render->cairo. */
cairo_t *flutl_recorder_cairo (FlutlRecorder *rec);

/* One TTML region == one binning group of the display list. */
void flutl_recorder_region_begin (FlutlRecorder *rec);
void flutl_recorder_set_device_shift (FlutlRecorder *rec, int dx, int dy);
void flutl_recorder_region_end (FlutlRecorder *rec);

/* Opacity group = the temporary region surface of :1250-1256 / :1375-1382.
 * Primitives recorded in between are also clipped to the surface bounds. */
void flutl_recorder_group_begin (FlutlRecorder *rec, int x, int y, int w, int h);
void flutl_recorder_group_end (FlutlRecorder *rec, double opacity);

/* cairo_rectangle + cairo_clip with integer device coordinates (:1279-1284);
 * intersects with the current clip, like cairo_clip.  clip_reset drops it
 * (cairo_restore at the end of show_regions). */
void flutl_recorder_clip (FlutlRecorder *rec, int x, int y, int w, int h);
void flutl_recorder_clip_reset (FlutlRecorder *rec);

/* cairo_set_source_rgba. */
void flutl_recorder_set_color (FlutlRecorder *rec, double r, double g, double b,
    double a);
void flutl_recorder_get_color (FlutlRecorder *rec, double rgba[4]);
/* colour tags (see flutl_recorder.c): the source of the colour just set, kept per primitive */
uint32_t flutl_recorder_pixel_of (double r, double g, double b, double a);
void flutl_recorder_set_color_tag (FlutlRecorder *rec, uint16_t tag);
void flutl_recorder_set_ctm_inverse (FlutlRecorder *rec, const cairo_matrix_t *inverse);
/* the CTM of the next show_glyph / path_glyph of a KNOWN glyph, given instead of set on the
 * tracking context (NULL: the context's again) */
void flutl_recorder_set_ctm (FlutlRecorder *rec, const cairo_matrix_t *m);
uint16_t flutl_recorder_get_color_tag (FlutlRecorder *rec);
const uint16_t *flutl_recorder_prim_tags (FlutlRecorder *rec);

/* cairo_rectangle (x, y, w, h) + cairo_fill under the tracking context's CTM. */
void flutl_recorder_fill_rect (FlutlRecorder *rec, double x, double y, double w,
    double h);

/* pango_cairo_show_glyph_string for ONE glyph at the CTM origin (:1149-1151). */
void flutl_recorder_show_glyph (FlutlRecorder *rec, PangoFont *font,
    const PangoGlyphInfo *gi);

/* Outline pass (:1153-1154, :1053-1092 with render == FALSE): the pieces of the
 * path, each under the tracking context's current CTM.  They are only turned
 * into a real Cairo path if the stroke-mask cache misses. */
void flutl_recorder_path_glyph (FlutlRecorder *rec, PangoFont *font,
    const PangoGlyphInfo *gi);
void flutl_recorder_path_polygon (FlutlRecorder *rec, const double *pts, int n);
void flutl_recorder_path_rect (FlutlRecorder *rec, double x, double y, double w,
    double h);
/* anamorphic glyph (:1105-1112): cairo_scale (hscale, 1) + cairo_set_scaled_font
 * + cairo_glyph_path of one glyph at the CTM origin */
void flutl_recorder_path_scaled_glyph (FlutlRecorder *rec, cairo_scaled_font_t *sfont,
    unsigned long glyph, double hscale);
/* cairo_fill of the pending path (:1117) */
void flutl_recorder_fill (FlutlRecorder *rec);

/* cairo_stroke of the pending path with the tracking context's line width
 * (:1212, and the intermediate strokes of show_layout in path mode). */
void flutl_recorder_stroke (FlutlRecorder *rec);

/* cairo_set_source_surface (img, x, y) + cairo_paint (:1271-1276). */
void flutl_recorder_paint_image (FlutlRecorder *rec, cairo_surface_t *img, int x,
    int y);

/* ---- blurred outlines: a child recorder records the temporary surface of
 * gstttmlrender.c:1196-1199 as its own display list; its masks go to the
 * parent's atlas.  The device blurs it into a new atlas entry
 * (flutl_atlas_put_blurred); the host copy of that entry is appended here
 * without being queued for upload, and blitted with paint_atlas_image. */
FlutlRecorder *flutl_recorder_new_child (FlutlRecorder *parent);
uint32_t flutl_recorder_next_atlas_offset (FlutlRecorder *rec);
uint32_t flutl_recorder_atlas_append_device_entry (FlutlRecorder *rec,
    const void *data, size_t bytes);
void flutl_recorder_paint_atlas_image (FlutlRecorder *rec, uint32_t offset, int w,
    int h, int x, int y);

/* _cairo_fixed_from_double: device coordinate -> Cairo's 24.8 fixed point */
int32_t flutl_recorder_fixed_from_double (double d);

/* ---- results of the frame -------------------------------------------------*/
const flutl_region *flutl_recorder_regions (FlutlRecorder *rec, uint32_t *n);
const flutl_prim *flutl_recorder_prims (FlutlRecorder *rec, uint32_t *n);
/* host copy of the atlas: bytes [0, size), incl. the 16-byte leading slack */
const uint8_t *flutl_recorder_atlas (FlutlRecorder *rec, size_t *size);
/* An atlas entry the device makes of a polygon (flutl_atlas_put_polygon): the
 * outline of a text as the worker's stroker left it.  edges: n_edges x 7 int32 (24.8
 * fixed point cairo_edge_t), box: where Cairo converts it (raster pixels), w x h its
 * size; the entry has the MASK layout (4 + (w + 4) * h bytes, zero padded). */
typedef struct
{
  int32_t *edges;
  uint32_t n_edges;
  int32_t box[4];
  int w, h;
  /* edges in a block of the device context's page-locked arena (see
   * flutl_recorder_set_edge_blocks): not free()d; given back through `release` unless
   * `submitted` (flutl_atlas_put_polygon has taken the block) */
  int in_block, submitted;
  void (*release) (void *user, void *block);
  void *release_user;
} FlutlPolygonEntry;
void flutl_polygon_entry_free (FlutlPolygonEntry *pe);
/* Where the stroke workers put a polygon's edges: acquire (user, bytes) returns a block the
 * device can read in place (flutl_ctx_edge_block_acquire) or NULL (the edges then stay in
 * ordinary memory and are copied when the entry is made).  NULL functions: never.  The
 * functions must stay callable until every polygon entry made under them is freed. */
void flutl_recorder_set_edge_blocks (FlutlRecorder *rec, void *(*acquire) (void *user, size_t bytes),
    void (*release) (void *user, void *block), void *user);
/* Outlines without a mask leave the worker threads as polygons for the device instead
 * of as coverage rasterised by Cairo (only where strokes are deferred, set_defer). */
void flutl_recorder_set_device_strokes (FlutlRecorder *rec, int on);
/* outline paths composed from cached glyph outlines instead of built by Cairo and copied
 * out (process-wide); *checked: how many of them FLUTL_CHECK_PATHS compared with Cairo's */
uint64_t flutl_recorder_composed_paths (uint64_t *checked);
uint64_t flutl_recorder_polygon_jobs (void);
void flutl_recorder_raster_stats (uint64_t *by_cairo, uint64_t *device_fills);

/* atlas entries appended since the last call (for incremental upload);
 * returns the number of entries; entry i spans [offsets[i], offsets[i]+sizes[i]);
 * polygons[i] != NULL: the device makes that entry (the caller owns the polygon now:
 * flutl_polygon_entry_free), else its bytes are in the host copy */
uint32_t flutl_recorder_new_atlas_entries (FlutlRecorder *rec,
    const uint32_t **offsets, const uint32_t **sizes, FlutlPolygonEntry *const **polygons);
/* number of primitives dropped because a feature is not implemented yet */
uint32_t flutl_recorder_unsupported (FlutlRecorder *rec);

/* ---- region replay (flutl_recorder.c, "region replay") ------------------------- */
void flutl_recorder_mark (FlutlRecorder *rec, uint32_t *prim_mark, uint32_t *region_mark);
int flutl_recorder_frame_is_clear (FlutlRecorder *rec);
const flutl_prim *flutl_recorder_snapshot (FlutlRecorder *rec, uint32_t prim_mark,
    uint32_t region_mark, uint32_t *n_prims, const flutl_region **region,
    int *has_placeholder);
void flutl_recorder_replay (FlutlRecorder *rec, const flutl_prim *prims, uint32_t n_prims,
    const flutl_region *region, int frame_clear_after, uint32_t unsupported);

/* ---- deferred strokes (flutl_recorder.c, "deferred strokes") -----------------
 * With set_defer (root recorders only) a stroke the cache has no mask for is
 * prepared on the calling thread and rasterised on a worker thread; the frame's
 * list then holds placeholder primitives, described by patches, and is finished
 * by flutl_recorder_resolve -- typically on another thread, on a COPY of the
 * frame's arrays, while the recorder goes on with the next frame.  What the two
 * threads share is guarded by the recorder's lock; whoever reads the atlas or
 * the list of new atlas entries while frames are being resolved elsewhere holds
 * it (flutl_recorder_lock).  flutl_recorder_reset must only be called when every
 * frame has been resolved. */
typedef struct
{
  uint32_t prim;             /* index of the placeholder in the frame's prims */
  uint32_t entry;            /* stroke cache entry whose mask it takes */
  int32_t ox, oy;            /* frame position of the mask's raster origin */
  int16_t clip[4];           /* frame x clip x group surface, when it was emitted */
} FlutlPatch;

void flutl_recorder_set_defer (FlutlRecorder *rec, int on);
const FlutlPatch *flutl_recorder_patches (FlutlRecorder *rec, uint32_t *n);
void flutl_recorder_resolve (FlutlRecorder *rec, flutl_region *regions, uint32_t n_regions,
    flutl_prim *prims, uint32_t n_prims, const FlutlPatch *patches, uint32_t n_patches);
/* process-wide statistics of the stroke workers: jobs done, time they ran, time they
 * waited in the pool's queue (microseconds) */
void flutl_recorder_stroke_stats (uint64_t *jobs, uint64_t *busy_us, uint64_t *queued_us);
uint64_t flutl_recorder_resolve_wait_us (void);
void flutl_recorder_prepare_stats (uint64_t *path_us, uint64_t *extents_us);
/* glyph runs turned into one mask so far / left as they are because two glyphs meet */
void flutl_recorder_stroke_split (uint64_t *extents_us, uint64_t *draw_us, uint64_t *crop_us);
void flutl_recorder_run_stats (FlutlRecorder *rec, uint32_t *merged, uint32_t *clashed);
void flutl_recorder_lock (FlutlRecorder *rec);
void flutl_recorder_unlock (FlutlRecorder *rec);

#endif /* FLUTL_RECORDER_H */
