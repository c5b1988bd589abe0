/*
 * flutl.h -- C-ABI of the MI355X-native ttmlrender compositor ("fluttml-hip").
 *
 * This is the drop-in boundary UNDER the reference's GStreamer element
 * (plugins/ttml/gstttmlrender.c).  The reference element's per-buffer work,
 * gst_ttmlrender_gen_buffer (gstttmlrender.c:1427-1478), is split in two:
 *
 *   host  (C, Pango for shaping, Cairo only as A8 mask rasteriser)
 *         builds a DISPLAY LIST: an ordered list of pixel primitives with the
 *         region scissor, z-order and colour quantisation already baked in;
 *   device (hand-written HIP, gfx950) composites that list into the
 *         premultiplied B,G,R,A frame, every pixel stored exactly once.
 *
 * Each entry point below names the reference call site(s) it replaces.
 * Plain pointers and sizes only; no GStreamer, torch or HIP types.
 *
 * All functions return 0 on success or a negative FLUTL_E* code.  There is no
 * CPU fallback: without a usable HIP device every compute entry point returns
 * FLUTL_ENODEV.
 */
#ifndef FLUTL_H
#define FLUTL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLUTL_ABI_VERSION 2

/* ---- error codes --------------------------------------------------------- */
#define FLUTL_OK       0
#define FLUTL_EINVAL  -22 /* bad argument / malformed display list          */
#define FLUTL_ENOMEM  -12 /* hipMalloc / malloc failed                       */
#define FLUTL_ENODEV  -19 /* no HIP device, or HIP runtime error             */
#define FLUTL_ERANGE  -34 /* primitive references bytes outside the atlas    */

/* ---- display-list IR ------------------------------------------------------
 *
 * Pixel format everywhere: premultiplied a8r8g8b8 in a little-endian uint32,
 * i.e. memory order B,G,R,A == Cairo CAIRO_FORMAT_ARGB32 == what the
 * reference pushes downstream (gstttmlrender.c:1446-1448; caps say "BGRA").
 *
 * Primitive ops.  "coverage" is an 8-bit value per destination pixel.
 */
#define FLUTL_OP_RECT        1u /* coverage from a 24.8 fixed-point box: replaces
                                 * cairo_rectangle+cairo_fill at gstttmlrender.c
                                 * :1265-1267 (region bg, pixel aligned), :1025-1027
                                 * (span bg), :1053-1092 (under/over/strike lines).
                                 * a,b,c,d = x1,y1,x2,y2 in 1/256 px.             */
#define FLUTL_OP_MASK        2u /* coverage from an A8 mask in the atlas: replaces
                                 * pango_cairo_show_glyph_string (:1150, one glyph)
                                 * and cairo_stroke of the outline path (:1212).
                                 * a = atlas byte offset, b = row stride (bytes),
                                 * c,d = destination x,y of mask pixel (0,0).
                                 * Atlas contract: every mask row is followed by
                                 * >= 3 zero bytes and the first row preceded by
                                 * >= 3 zero bytes, so that the kernel's 4-byte
                                 * windows read 0 left and right of the mask, and
                                 * the FIRST 16 BYTES OF THE ATLAS ARE ZERO (lanes
                                 * outside a mask read them instead); stride
                                 * < 2^23; if
                                 * [cx0,cx1) is narrower than the mask rows (the
                                 * clip cut it) set FLUTL_MASK_XCLIP in param and
                                 * the kernel masks the columns itself.           */
#define FLUTL_MASK_XCLIP     1u /* param bit of a MASK prim, see above             */
#define FLUTL_OP_IMAGE       3u /* premultiplied ARGB32 image in the atlas, OVER:
                                 * replaces set_source_surface+paint (:1272-1275)
                                 * and the blurred-outline blit (:1219-1224).
                                 * a,b,c,d as for MASK (stride in bytes).         */
#define FLUTL_OP_GROUP_BEGIN 4u /* region opacity group opens: the temp surface of
                                 * :1250-1255.  Following prims paint into the
                                 * group buffer, which starts transparent.        */
#define FLUTL_OP_GROUP_END   5u /* cairo_paint_with_alpha (:1378-1380):
                                 * dst = (group IN param) OVER dst.               */

/* Blend rules (which third-party arithmetic the reference reaches). */
#define FLUTL_RULE_PIXMAN 0u /* pixman: d = s*m + d*(255 - sa*m), MUL_UN8 rounding
                              * (t = a*b + 0x80; (t + (t >> 8)) >> 8), saturating add */
#define FLUTL_RULE_LERP   1u /* cairo in-place span renderer for OPAQUE solid
                              * sources (fill/stroke with partial coverage):
                              * d = mul(s,m) + mul(d,255-m) with 0x7f rounding       */

typedef struct flutl_prim {
  uint8_t  op;     /* FLUTL_OP_*                                              */
  uint8_t  rule;   /* FLUTL_RULE_*                                            */
  uint16_t param;  /* GROUP_END: 8-bit group alpha; MASK: FLUTL_MASK_XCLIP  */
  int16_t  cx0, cy0, cx1, cy1; /* destination bbox [cx0,cx1) x [cy0,cy1) in px,
                                * already intersected with clip, group surface
                                * and frame; nothing is touched outside it     */
  uint32_t color;  /* premultiplied a8r8g8b8 source (RECT, MASK)              */
  int32_t  a, b, c, d; /* op-specific, see FLUTL_OP_*                         */
} flutl_prim; /* 32 bytes */

/* One binning group == the primitives of one TTML region, in paint order.
 * Regions are stored in paint (z) order: gstttmlrender.c:659-660, :1466-1467. */
typedef struct flutl_region {
  int16_t  x0, y0, x1, y1; /* union of the prim bboxes (for tile culling)     */
  uint32_t first;          /* index of first prim                             */
  uint32_t count;          /* number of prims                                 */
} flutl_region; /* 16 bytes */

/* Output tile of the compositor kernel (the north star's 64x8): one wave
 * composites two of them stacked, a 64x16 "supertile", strip (64x4) by strip. */
#define FLUTL_TILE_W 64
#define FLUTL_TILE_H 8

/* ---- host-side validation (no device needed) ----------------------------- */

/* Checks a display list for structural validity against a W x H frame and an
 * atlas of atlas_bytes: ops/rules known, bboxes inside the frame, mask/image
 * reads inside the atlas, groups balanced per region.  Launching an invalid
 * list is refused by every render entry point (a stray read would fault the
 * GPU). */
int flutl_list_validate (uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, size_t atlas_bytes);

/* ---- stateless launches on caller-owned device memory --------------------
 * For callers that manage HBM themselves (a device-memory GstAllocator, the
 * test-suite, bench.py).  All pointers are DEVICE pointers; `stream` is a
 * hipStream_t (NULL = default stream).  The list must have been validated on
 * the host with flutl_list_validate(). */

/* One frame: the device side of gst_ttmlrender_gen_buffer
 * (gstttmlrender.c:1446-1467: CLEAR + region loop), one launch. */
int flutl_launch_frame (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas);

/* The same, compositing only the rows [row0, row1) of the frame, widened to
 * whole 16-row supertiles (FLUTL_TILE_H * 2); nothing else of dst is written.
 * A subtitle frame is transparent outside its regions: the element composites
 * the rows its regions span and lets flutl_launch_delta_window take the rest as
 * transparent. */
int flutl_launch_frame_rows (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas, uint32_t row0, uint32_t row1);

/* ... taking the window's supertile rows in the order of `row_order` (device:
 * flutl_row_order's table), heaviest first.  A single frame is one round of
 * waves, as long as its slowest; rows dealt top to bottom start the text of the
 * lowest region last.  Speed only: any order gives the same pixels. */
int flutl_launch_frame_ordered (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas, uint32_t row0, uint32_t row1,
    const uint16_t *row_order);
/* ... and with the delta read-back fused into the store: the rows [row0, row1) of the
 * frame are composited and COMPARED with `twin` (HBM; what the host block holds);
 * the segments that differ (flutl_delta_segment_bytes ()) are written into the twin
 * and into `host` (page-locked, device mapped: over PCIe), everything else is left
 * alone.  The frame is never written to a scratch surface and read back (what
 * flutl_launch_frame_ordered + flutl_launch_delta_window cost: 8 more bytes of HBM
 * traffic per pixel and a launch).  The window must cover what the twin holds as
 * well as what the list draws.  park: a scratch frame of the same geometry (what a
 * translucent group sets aside while it draws).  This is what flutl_frame_submit
 * launches per buffer (replaces gstttmlrender.c:1442-1472 + the push of the frame). */
int flutl_launch_frame_delta (void *stream, uint32_t width, uint32_t height,
    void *twin, void *host, void *park, size_t stride_bytes,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas, uint32_t row0, uint32_t row1,
    const uint16_t *row_order, uint64_t *changed_segments);
/* Host helper: the supertile rows of the window [row0, row1), as indices relative
 * to its first one, sorted by how much coverage work the (validated) list brings
 * to each, descending.  order_out: room for (height + 15) / 16 entries; returns
 * the number written. */
uint32_t flutl_row_order (uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, uint32_t row0, uint32_t row1,
    uint16_t *order_out);

/* A batch of independent frames (different scene states / streams) in ONE
 * launch: frame i uses regions[region_first[i] .. +region_count[i]) and is
 * written to dst + i*dst_frame_bytes.  Prim indices inside regions are
 * absolute into `prims`. */
typedef struct flutl_frame_desc {
  uint32_t region_first;
  uint32_t region_count;
  uint64_t dst_offset;   /* byte offset of this frame from `dst`             */
} flutl_frame_desc;

int flutl_launch_batch (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes,
    const flutl_frame_desc *frames /* device */, uint32_t n_frames,
    const flutl_region *regions, const flutl_prim *prims,
    const uint8_t *atlas);

/* The same batch, composited OVER video frames in the same launch: frame i of
 * the result is  overlay_i OVER video_i  (pixman OVER of premultiplied pixels,
 * saturating), where video_i = video + frames[i].dst_offset has the geometry
 * of the destination (same stride) and `video` != `dst`.  This is the mixing
 * step the reference leaves to a downstream element (plugins/ttml/README.md:
 * 47-50, "videomixer"/"compositor" after ttmlrender); fused here the overlay
 * frame never exists in memory.  Opaque video in, opaque video out. */
int flutl_launch_batch_over (void *stream, uint32_t width, uint32_t height,
    void *dst, const void *video, size_t stride_bytes,
    const flutl_frame_desc *frames /* device */, uint32_t n_frames,
    const flutl_region *regions, const flutl_prim *prims,
    const uint8_t *atlas);

/* One frame OVER one video frame, rows [row0, row1) only (whole supertiles): the
 * launch behind the element's video_sink pad. */
int flutl_launch_frame_over_rows (void *stream, uint32_t width, uint32_t height,
    void *dst, const void *video, size_t stride_bytes,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas, uint32_t row0, uint32_t row1);

/* Gaussian blur of a premultiplied ARGB32 image with pixman's convolution
 * semantics: replaces gst_ttml_blur_image_surface (gstttmlblur.c:71-110).
 * `kernel_fixed` = (2r+1)^2 pixman 16.16 weights (device), as produced by
 * flutl_blur_kernel(). src/dst are w x h, stride in bytes; taps outside the
 * source read transparent (PIXMAN_REPEAT_NONE). */
int flutl_launch_blur (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t stride_bytes,
    const int32_t *kernel_fixed, int radius);

/* Host helper: replaces gst_ttml_blur_create_gaussian_kernel
 * (gstttmlblur.c:27-67).  Writes (2r+1)^2 16.16 weights to out. */
int flutl_blur_kernel (int radius, double sigma, int32_t *out);

/* pixman PIXMAN_FILTER_BILINEAR resample of a premultiplied ARGB32 image under
 * a pure fractional translation: replaces what cairo_paint does at
 * gstttmlrender.c:1219-1224 when the CTM offset is not an integer.  distx /
 * disty = the 7-bit weights of the right / lower neighbour; dst is
 * (w+1) x (h+1): dst (i, j) mixes src (i-1, j-1), (i, j-1), (i-1, j), (i, j),
 * transparent outside the source (PIXMAN_REPEAT_NONE).
 * box (host pointer, 4 ints, or NULL): the reference paints the surface
 * through cairo_rectangle + cairo_fill of its extent (:1221-1224); {x1, y1, x2,
 * y2} is that rectangle in Cairo's 24.8 fixed point relative to dst (0, 0),
 * and its antialiased edge coverage multiplies the sample (pixman IN). */
int flutl_launch_resample (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t src_stride_bytes, size_t dst_stride_bytes,
    uint32_t distx, uint32_t disty, const int32_t *box);

/* Quarter turn clockwise of an ARGB32 image, w x h -> h x w: what cairo_fill
 * of a surface pattern produces under the 90-degree CTM of the vertical
 * writing modes (gstttmlrender.c:1313-1340, NEAREST filter at integer
 * positions): dst (row i, column h-1-j) = src (row j, column i).  box: as for
 * flutl_launch_resample, in the coordinates of dst, or NULL. */
int flutl_launch_rotate_cw (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t src_stride_bytes, size_t dst_stride_bytes,
    const int32_t *box);

/* Delta read-back: compares the rows [row0, row1) of a freshly composited frame
 * (`fresh`, device, tight rows of 4*width bytes) with `twin` (device, same
 * layout) a segment (flutl_delta_segment_bytes ()) at a time and writes the segments that differ into `twin`
 * and into `host` -- page-locked host memory mapped into the device (the device
 * address of a flutl_host_alloc block), same layout.  With twin == host's
 * content before the call, both equal `fresh` on those rows afterwards; only
 * what changed crosses PCIe.  This is the "dirty-rect D2H into pre-zeroed
 * buffers" of a pooled output frame (replaces the whole-frame write of
 * gstttmlrender.c:1442-1452).  changed_segments (device, or NULL): incremented
 * by the number of segments written (flutl_delta_segment_bytes () each). */
uint32_t flutl_delta_segment_bytes (void);
int flutl_launch_delta (void *stream, const void *fresh, void *twin, void *host,
    uint32_t width, uint32_t height, uint32_t row0, uint32_t row1,
    uint64_t *changed_segments);
/* ... where only the rows [fresh_row0, fresh_row1) of `fresh` were composited
 * (flutl_launch_frame_rows): the new frame counts as transparent on the others. */
int flutl_launch_delta_window (void *stream, const void *fresh, void *twin, void *host,
    uint32_t width, uint32_t height, uint32_t row0, uint32_t row1,
    uint32_t fresh_row0, uint32_t fresh_row1, uint64_t *changed_segments);

/* ---- context API (library-owned HBM; what the GStreamer element uses) ---- */

typedef struct flutl_ctx flutl_ctx;     /* one per element instance / thread  */
typedef struct flutl_list flutl_list;   /* a display list resident in HBM     */

/* Create a context on HIP device `device` with its own stream.  Replaces the
 * per-instance Cairo state set up in gst_ttmlrender_init (gstttmlrender.c
 * :1723-1750).  Re-entrant: several contexts may be used from different
 * threads concurrently. */
int  flutl_ctx_create (int device, flutl_ctx **out);
void flutl_ctx_destroy (flutl_ctx *ctx);

/* Atlas: one growing device buffer of A8 masks / ARGB32 images per context.
 * flutl_atlas_put appends `bytes` (16-byte aligned) and returns its offset.
 * flutl_atlas_reset drops all entries: gst_ttmlrender_reset (:1610-1621). */
int  flutl_atlas_put (flutl_ctx *ctx, const void *src, size_t bytes,
    uint32_t *offset_out);
void flutl_atlas_reset (flutl_ctx *ctx);
size_t flutl_atlas_size (const flutl_ctx *ctx);

/* Blurred text outline (gst_ttmlrender_render_outline with a blur radius,
 * gstttmlrender.c:1193-1229): the display list `regions/prims` is composited
 * on the device into a cleared w x h ARGB32 temporary surface (the
 * cairo_image_surface_create of :1196), that surface is convolved with the
 * Gaussian kernel of gst_ttml_blur_image_surface (gstttmlblur.c:71-110,
 * radius/sigma as its arguments) and the result becomes a NEW ATLAS ENTRY
 * (premultiplied ARGB32, stride 4*w) whose offset is returned: the frame's
 * display list then blits it with an IMAGE primitive (:1219-1224).
 * distx / disty (0..127): when the blit position is fractional Cairo samples
 * the blurred surface with pixman's bilinear filter, whose 7-bit weights are
 * constant for a pure translation; the entry is then the blurred surface
 * resampled by them, (w+1) x (h+1), stride 4*(w+1), blitted at integer
 * coordinates (see flutl_launch_resample, also for `box`).  0, 0, NULL: the
 * blurred surface itself.  rotate_cw: the entry is turned a quarter clockwise
 * (flutl_launch_rotate_cw) -- after the resample, whose weights are then along
 * the surface's own axes, and `box` is in the turned (device) axes.
 * If host_copy is not NULL the entry's bytes are also copied there (for
 * scene-pack dumps).  radius <= FLUTL_BLUR_MAX_RADIUS -- twice what the reference survives:
 * gstttmlblur.c:38 keeps (2 radius + 1)^2 doubles on the stack (g_newa), 8 MB of it at a
 * radius of 511.  No limit on the surface: the convolution is not separable in the
 * reference's fixed point, so its cost is w * h * (2 radius + 1)^2 taps whatever one does
 * (the reference's pixman needs minutes per frame for large ones); it is queued as launches
 * of at most 2e10 taps each (flutl_launch_blur). */
#define FLUTL_BLUR_MAX_RADIUS 1024
int  flutl_atlas_put_blurred (flutl_ctx *ctx, uint32_t w, uint32_t h,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims,
    int radius, double sigma, uint32_t distx, uint32_t disty, const int32_t *box,
    int rotate_cw, void *host_copy, uint32_t *offset_out);

/* Outline coverage on the device.  The reference draws tts:textOutline with
 * cairo_set_line_width + cairo_stroke over the glyph paths of a layout
 * (gst_ttmlrender_render_outline, gstttmlrender.c:1209-1212); under that call Cairo's
 * image backend turns the stroke into a polygon and scan converts it
 * (cairo-tor-scan-converter.c, 15 x 256 samples per pixel) -- where the reference
 * spends > 90 % of its time on outlined text.  The polygon is made on the host
 * (host/flutl_stroker.c, Cairo's own arithmetic); this converts it: `edges` are
 * n_edges x 7 int32 {p1.x, p1.y, p2.x, p2.y, top, bottom, dir} in 24.8 fixed point
 * (a cairo_edge_t), box = {x0, y0, x1, y1} in pixels is where Cairo converts (the
 * polygon's extents within surface, clip and approximate stroke extents;
 * 0 <= x0 < x1 <= x0 + 32767; wider than 4096 it is converted in column windows, the
 * same coverage).  The coverage becomes a NEW ATLAS ENTRY in the layout
 * MASK primitives read: 4 zero bytes, then box height rows of (width + 4) bytes, each
 * width coverage bytes + 4 zero bytes; *offset_out is the entry's offset (pixel (0,0)
 * of the box is at offset + 4).  Bit exact with Cairo 1.16 for any polygon, also
 * where Cairo's result depends on the order its sort left equal keys in. */
int  flutl_atlas_put_polygon (flutl_ctx *ctx, const int32_t *edges, uint32_t n_edges,
    const int32_t box[4], int even_odd, uint32_t *offset_out);
/* Page-locked memory the device reads an outline's edges from (flutl_atlas_put_polygon
 * copies edges that lie elsewhere into such a block; a caller that lets its stroke workers
 * write them there directly saves that copy on the thread that feeds the device).
 * acquire: thread safe, never waits for the device; FLUTL_ENOMEM when `bytes` exceed a block
 * (1 MiB) or none is free right now -- the caller then keeps its edges in ordinary memory.
 * A block passed to flutl_atlas_put_polygon (at its first byte) belongs to the context again
 * whatever that call returns; release is for a block that never got there. */
int  flutl_ctx_edge_block_acquire (flutl_ctx *ctx, size_t bytes, void **block_out);
void flutl_ctx_edge_block_release (flutl_ctx *ctx, void *block);
/* Reserve the space such an entry takes WITHOUT making it (the caller keeps offsets in
 * step with a host-side atlas): bytes = 4 + (w + 4) * h. */
size_t flutl_polygon_entry_bytes (uint32_t w, uint32_t h);
/* Statistics: polygons of this context that needed the sequential converter. */
uint32_t flutl_ctx_scan_fallbacks (flutl_ctx *ctx);

/* Stateless: the same conversion on caller-owned device memory.  edges_dev: the edges
 * in device memory; scratch: flutl_scan_scratch_bytes (...) bytes of device memory;
 * dst: (x1 - x0) x (y1 - y0) bytes at dst_stride.  mode 0: one workgroup per pixel row,
 * and the sequential converter behind it for a polygon the rows flagged; 1: rows only
 * (flutl_scan_status then says whether they flagged: bit 0 an undecided tie, bit 1 too
 * many edges in a row); 2: the sequential converter alone.
 * | FLUTL_SCAN_CLEAN: the caller has zeroed the first flutl_scan_zero_bytes () bytes of
 * `scratch` once; every conversion leaves them zero, so a scratch buffer that serves one
 * conversion after the other (as the context's does) needs no memset in front of each. */
#define FLUTL_SCAN_CLEAN 16
size_t flutl_scan_scratch_bytes (uint32_t n_edges, uint32_t width, uint32_t height);
size_t flutl_scan_zero_bytes (void);
int  flutl_launch_scan (void *stream, const int32_t *edges_dev, uint32_t n_edges,
    const int32_t box[4], int even_odd, void *scratch, void *dst, size_t dst_stride, int mode);
int  flutl_scan_status (void *stream, const void *scratch, uint32_t n_edges, uint32_t height,
    uint32_t *status_out);

/* Upload (validate + H2D) a display list for a W x H frame. */
int  flutl_list_create (flutl_ctx *ctx, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims, flutl_list **out);
void flutl_list_destroy (flutl_ctx *ctx, flutl_list *list);

/* Render into a device buffer (asynchronous on the context's stream). */
int  flutl_render (flutl_ctx *ctx, const flutl_list *list,
    void *dst_dev, size_t dst_stride_bytes);
/* Render and copy to host memory (synchronous): what gen_buffer needs to fill
 * a system-memory GstBuffer (gstttmlrender.c:1442-1444, :1474-1477). */
int  flutl_render_to_host (flutl_ctx *ctx, const flutl_list *list,
    void *dst_host, size_t dst_stride_bytes);
int  flutl_ctx_sync (flutl_ctx *ctx);

/* The whole per-buffer device step in one call, with no allocation in steady
 * state: validate the list, copy it into the context's reusable list buffer,
 * composite, copy the frame to dst_host (page-locked memory from
 * flutl_host_alloc makes that copy a plain DMA), wait.  This is what
 * gst_ttmlrender_gen_buffer's Cairo section (gstttmlrender.c:1446-1477) becomes. */
int  flutl_render_frame_to_host (flutl_ctx *ctx, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims,
    void *dst_host, size_t dst_stride_bytes);
/* The same, copying back only the rows [row_bands[2i], row_bands[2i+1]) of the
 * frame (n_bands pairs, inside [0, height)): the "dirty-rect D2H into
 * pre-zeroed buffers" of a pooled output frame -- rows outside every region
 * are transparent, so a buffer that was zero there stays right without being
 * copied.  The caller must include the rows its buffer was written in before.
 * n_bands == 0 copies the whole frame. */
int  flutl_render_frame_rows_to_host (flutl_ctx *ctx, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims,
    void *dst_host, size_t dst_stride_bytes,
    const uint32_t *row_bands, uint32_t n_bands);
/* ... and of those rows only the columns [col0, col1): the regions of a subtitle
 * frame rarely reach the left and right borders of the picture. */
int  flutl_render_frame_window_to_host (flutl_ctx *ctx, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims,
    void *dst_host, size_t dst_stride_bytes,
    const uint32_t *row_bands, uint32_t n_bands, uint32_t col0, uint32_t col1);

/* Fused over-video for a video frame in HOST memory (what the element's optional
 * video_sink pad receives; the mixing the reference leaves to a downstream
 * compositor, plugins/ttml/README.md:47-50): the display list is composited OVER
 * the frame (pixman OVER of premultiplied pixels) and the result written back IN
 * PLACE.  Only the rows the list's regions span travel: up, through the fused
 * kernel, and back; a list without regions leaves the frame untouched and costs
 * nothing.  Synchronous.  video_host: width x height BGRA/BGRx pixels (opaque),
 * rows stride_bytes apart. */
int  flutl_blend_over_host (flutl_ctx *ctx, uint32_t width, uint32_t height,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims,
    void *video_host, size_t stride_bytes);

/* ---- pooled output frames, completed asynchronously ----------------------
 * A flutl_frame is one output frame of the element: a page-locked host block
 * (what downstream maps: system memory, tight rows of 4*width bytes, initially
 * transparent) and its twin in HBM (what a HIP-aware consumer reads: the same
 * pixels, device resident).  flutl_frame_submit queues, on the context's
 * stream and without waiting: the display list's upload, the composite into the
 * context's scratch frame and the delta read-back into block + twin
 * (flutl_launch_delta over the rows this or the previous content of the frame
 * can occupy); it returns as soon as the work is queued, so the caller records
 * the next buffer while this one renders (the reference renders synchronously
 * inside gen_buffer, gstttmlrender.c:1446-1477).  flutl_frame_wait blocks until
 * the last submission into the frame has completed (any thread; this is what
 * the element's GstMemory map does).  At most a few submissions are in flight
 * per context; a further one waits for the oldest.  A frame may outlive its
 * context.  The list memory may be reused by the caller right after submit. */
typedef struct flutl_frame flutl_frame;
int   flutl_frame_create (flutl_ctx *ctx, uint32_t width, uint32_t height,
    flutl_frame **out);
/* ... with the host block cleared by the calling thread instead of by the device in stream
 * order (quiet != 0): for frames made in the background of a running stream. */
int   flutl_frame_create_quiet (flutl_ctx *ctx, uint32_t width, uint32_t height, int quiet,
    flutl_frame **out);
/* Make the memory of `count` frames of width x height in one go: the frames created
 * afterwards take their host block and twin from it (cleared already; a block is reused when
 * its frame is destroyed) -- one huge-page mapping registered with the device instead of a
 * page-locked allocation per frame, 2 ms per 4K frame instead of 5.5-14 ms.  Blocks already
 * reserved and free count.  FLUTL_ENOMEM: nothing reserved, frames are made one by one as
 * before.  The reservation lives until the context and the last frame made of it are gone. */
int   flutl_ctx_reserve_frames (flutl_ctx *ctx, uint32_t width, uint32_t height, uint32_t count);
void  flutl_frame_destroy (flutl_frame *frame);
void *flutl_frame_host (const flutl_frame *frame);
const void *flutl_frame_device (const flutl_frame *frame);
int   flutl_frame_submit (flutl_ctx *ctx, flutl_frame *frame,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims);
/* flutl_frame_submit in two steps, for a frame that comes with NEW outlines: stage (the
 * list is validated, copied to page-locked memory, its rows ordered), then the caller makes
 * the frame's atlas entries (flutl_atlas_put_polygon ...), then submit_staged queues the
 * composite.  The staged list goes up to the device with the first scan conversion launched
 * in between (the conversion's kernel copies it) instead of as a copy operation of its own:
 * one stream operation and its gap less per cue of a stream with new outlined text.  One
 * staged frame per context at a time. */
int   flutl_frame_stage (flutl_ctx *ctx, flutl_frame *frame,
    const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, uint32_t n_prims);
int   flutl_frame_submit_staged (flutl_ctx *ctx, flutl_frame *frame);
int   flutl_frame_wait (flutl_frame *frame);
/* Somebody has WRITTEN into the frame's host block (a downstream element that works in
 * place on writable memory, FLUTL_WRITABLE): block and twin no longer hold the same pixels.
 * The next flutl_frame_submit brings the twin up to the block (one copy of the frame up, in
 * stream order) and composites the whole frame, so that the delta read-back puts every pixel
 * right again.  Any thread. */
void  flutl_frame_mark_written (flutl_frame *frame);
/* bytes all submissions on this context have written to host blocks so far
 * (drains the context's stream; statistics) */
uint64_t flutl_ctx_delta_bytes (flutl_ctx *ctx);

/* Page-locked host memory for output frames: a D2H copy into it is a true
 * asynchronous DMA at PCIe rate (pageable memory goes through a bounce buffer
 * and takes page faults on first touch).  The element wraps such blocks as
 * GstMemory and recycles them (the reference allocates a fresh system buffer
 * per frame, gstttmlrender.c:1442-1443).  Not tied to a context's lifetime. */
int  flutl_host_alloc (size_t bytes, void **out);
void flutl_host_free (void *ptr);

/* Library / device info (never fails; device_count 0 without a GPU). */
int         flutl_abi_version (void);
int         flutl_device_count (void);
const char *flutl_strerror (int code);

#ifdef __cplusplus
}
#endif
#endif /* FLUTL_H */
