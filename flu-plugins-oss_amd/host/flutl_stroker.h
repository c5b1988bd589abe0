/* flutl_stroker.h -- a stroked path as the polygon Cairo 1.16 makes of it (see
 * flutl_stroker.c).  Host side of the outline rasteriser: the polygon's edges go to
 * the device, which scan converts them (include/flutl.h: flutl_atlas_put_polygon). */
#ifndef FLUTL_STROKER_H
#define FLUTL_STROKER_H

#include <cairo.h>
#include <stdint.h>

enum
{
  FLUTL_STROKER_OK = 0,
  FLUTL_STROKER_ENOMEM = -1,
  FLUTL_STROKER_UNSUPPORTED = -2,  /* a path or matrix this file does not reproduce */
  FLUTL_STROKER_RECTILINEAR = -3   /* Cairo strokes such paths as boxes, not as a polygon */
};

/* A polygon as cairo-polygon.c holds it: edges of 7 int32 {p1.x, p1.y, p2.x, p2.y, top,
 * bottom, dir} in 24.8 fixed point (a cairo_edge_t: the line, the part of it that counts,
 * its winding direction) -- the layout flutl_atlas_put_polygon takes. */
typedef struct
{
  int32_t *edges;
  uint32_t n_edges, cap;
  int32_t extents[4];           /* x0, y0, x1, y1 of the edges' end points (cut at top / bottom) */
  int32_t uext[4];              /* ... before anything was cut to limits */
  int has_limits;
  int32_t limits[4];            /* the box edges were cut to (cairo_polygon_t.limits) */
  int box[4];                   /* pixels: the box Cairo scan converts the polygon in (all 0: nothing) */
} FlutlPolygon;

void flutl_polygon_init (FlutlPolygon *pg);
void flutl_polygon_fini (FlutlPolygon *pg);

/* cairo_stroke (path in device coordinates, appended under an identity CTM; `ctm` is
 * the matrix the pen lives in) as a polygon.  tolerance: cairo_get_tolerance (0.1).
 * target: {x0, y0, x1, y1} in pixels, the surface cut by the clip rectangle -- Cairo cuts
 * the polygon's edges to it (polygon limits) when the stroke may reach beyond it, and
 * that changes coverage INSIDE as well (a cut edge starts a new sample row).
 * polygon->box is where the coverage lands. */
int flutl_stroke_to_polygon (const cairo_path_t *path, const cairo_matrix_t *ctm,
    double line_width, double miter_limit, cairo_line_join_t line_join, double tolerance,
    const int target[4], FlutlPolygon *polygon);

/* cairo_fill (path in device coordinates, default fill rule: winding) as a polygon; target,
 * tolerance and polygon->box as above.  FLUTL_STROKER_RECTILINEAR: Cairo fills such a path
 * as boxes. */
int flutl_fill_to_polygon (const cairo_path_t *path, double tolerance, const int target[4],
    FlutlPolygon *polygon);

/* The pixel box Cairo converts the polygon in (before surface and clip cut it):
 * extents rounded out.  0 for an empty polygon. */
int flutl_polygon_box (const FlutlPolygon *pg, int box[4]);

/* cairo_path_fixed_t.extents taking in a curve (_cairo_box_add_curve_to: its end points and
 * its extreme points), 24.8 fixed point: box {x0, y0, x1, y1}, p the curve's four points */
void flutl_stroker_box_add_curve (int32_t box[4], const int32_t p[8]);

#endif
