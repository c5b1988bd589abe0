/*
 * flutl_stroker.c -- a stroked path as a polygon, exactly as Cairo 1.16 makes it.
 *
 * The reference draws tts:textOutline with cairo_set_line_width + cairo_stroke over the
 * glyph paths of a layout (plugins/ttml/gstttmlrender.c:1209-1212); under that call
 * Cairo's image backend (1) turns the path into a polygon -- two offset contours per
 * closed sub-path, pen fans along the curves, miter joins -- and (2) scan converts
 * the polygon.  (2) is the pixel work and runs on the GPU here (scan_rows_kernel & co.,
 * csrc/flutl_scan.hip); this file is (1): geometry on the host, in the doubles and
 * the 24.8 fixed point Cairo uses, so that the polygon is the SAME polygon, edge for
 * edge.  Third-party arithmetic absent from /root/reference: cairo 1.16.0
 * cairo-path-stroke-polygon.c, cairo-pen.c, cairo-spline.c, cairo-slope.c,
 * cairo-contour.c, cairo-polygon.c, cairo-path-fixed.c; restated from the published
 * algorithm and pinned against the real library by oracle/stroke_check.c (random
 * paths and real glyph outlines through cairo_stroke, byte for byte).
 *
 * Supported: what the element strokes -- closed and open sub-paths of lines and
 * cubic curves, miter joins, butt caps, no dashes, a CTM that is a similarity
 * (translation, rotation, reflection, uniform scale: see flutl_stroke_to_polygon for why
 * not more), the target box Cairo cuts the polygon to.  Anything else returns
 * FLUTL_STROKER_UNSUPPORTED (a path of axis-aligned rectangles only:
 * FLUTL_STROKER_RECTILINEAR) and the caller lets Cairo itself stroke it, as before
 * round 3 (flutl_recorder.c: stroke_job_run) -- never a frame that differs.
 */
#include "flutl_stroker.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t x, y; } Pt;
typedef struct { int32_t dx, dy; } Slope;
typedef struct { Pt p1, p2; } Box;

typedef struct
{
  Pt ccw, point, cw;
  Slope dev_vector;
  double dev_slope_x, dev_slope_y;
} Face;

typedef struct
{
  Pt *pts;
  int n, cap;
  int direction;
} Contour;

typedef struct
{
  Pt point;
  Slope slope_cw, slope_ccw;
} PenVertex;

typedef struct
{
  FlutlPolygon *polygon;
  const cairo_matrix_t *ctm, *ctm_inverse;
  int ctm_inverse_is_identity;
  double tolerance, half_line_width, miter_limit, spline_cusp_tolerance;
  int line_join;
  int ctm_det_positive;
  PenVertex *pen;
  int pen_n;
  Pt first_point;
  int has_initial_sub_path, has_current_face, has_first_face;
  Face current_face, first_face;
  int has_bounds;
  Box bounds;
  Contour cw, ccw;
  int failed;
  Pt fill_current, fill_last_move;      /* cairo_fill (flutl_fill_to_polygon): cairo_filler_t */
} Stroker;

/* ---- fixed point, slopes ------------------------------------------------------- */
static inline int32_t
fixed_from_double (double d)
{
  /* cairo-fixed-private.h: the 1.5 * 2^(52 - 8) trick, round half to even */
  union { double d; int32_t i[2]; } u;
  u.d = d + 26388279066624.0;
  return u.i[0];
}

static inline double
fixed_to_double (int32_t f)
{
  return ((double) f) / 256.0;
}

static inline void
slope_init (Slope *s, const Pt *a, const Pt *b)
{
  s->dx = b->x - a->x;
  s->dy = b->y - a->y;
}

static int
slope_compare (const Slope *a, const Slope *b)
{
  const int64_t ady_bdx = (int64_t) a->dy * b->dx;
  const int64_t bdy_adx = (int64_t) b->dy * a->dx;
  if (ady_bdx != bdy_adx)
    return ady_bdx < bdy_adx ? -1 : 1;
  /* zero vectors all compare equal, and more positive than any non-zero vector */
  if (a->dx == 0 && a->dy == 0 && b->dx == 0 && b->dy == 0)
    return 0;
  if (a->dx == 0 && a->dy == 0)
    return 1;
  if (b->dx == 0 && b->dy == 0)
    return -1;
  /* equal, or differing by exactly pi */
  if ((a->dx ^ b->dx) < 0 || (a->dy ^ b->dy) < 0) {
    if (a->dx > 0 || (a->dx == 0 && a->dy > 0))
      return -1;
    else
      return +1;
  }
  return 0;
}

/* ---- contours and the polygon ---------------------------------------------------- */
static void
contour_push (Stroker *st, Contour *c, const Pt *p)
{
  if (c->n == c->cap) {
    int cap = c->cap ? c->cap * 2 : 256;
    Pt *np = realloc (c->pts, (size_t) cap * sizeof *np);
    if (!np) {
      st->failed = 1;
      return;
    }
    c->pts = np;
    c->cap = cap;
  }
  c->pts[c->n++] = *p;
}

/* (cairo-path-stroke-polygon.c:contour_add_point: its within_tolerance test is
 * disabled in 1.16 -- every point is kept) */
#define contour_add_point(st, c, p) contour_push (st, c, p)

static void
polygon_push (Stroker *st, const Pt *p1, const Pt *p2, int top, int bottom, int dir)
{
  FlutlPolygon *pg = st->polygon;
  if (pg->n_edges == pg->cap) {
    uint32_t cap = pg->cap ? pg->cap * 2 : 1024;
    int32_t *ne = realloc (pg->edges, (size_t) cap * 7 * sizeof *ne);
    if (!ne) {
      st->failed = 1;
      return;
    }
    pg->edges = ne;
    pg->cap = cap;
  }
  int32_t *e = pg->edges + (size_t) pg->n_edges++ * 7;
  e[0] = p1->x; e[1] = p1->y; e[2] = p2->x; e[3] = p2->y;
  e[4] = top; e[5] = bottom; e[6] = dir;
}

/* _cairo_edge_compute_intersection_{x_for_y,y_for_x}: "_cairo_fixed_mul_div_floor",
 * which is C's truncating division in this build (seen in the binary: plain idiv) */
static int32_t
edge_x_for_y (const Pt *p1, const Pt *p2, int32_t y)
{
  if (y == p1->y)
    return p1->x;
  if (y == p2->y)
    return p2->x;
  int32_t x = p1->x;
  const int32_t dy = p2->y - p1->y;
  if (dy != 0) {
    const int64_t num = (int64_t) (y - p1->y) * (p2->x - p1->x);
    x += (int32_t) (num / dy);
  }
  return x;
}

static int32_t
edge_y_for_x (const Pt *p1, const Pt *p2, int32_t x)
{
  if (x == p1->x)
    return p1->y;
  if (x == p2->x)
    return p2->y;
  int32_t y = p1->y;
  const int32_t dx = p2->x - p1->x;
  if (dx != 0) {
    const int64_t num = (int64_t) (x - p1->x) * (p2->y - p1->y);
    y += (int32_t) (num / dx);
  }
  return y;
}

static void
add_edge (Stroker *st, const Pt *p1, const Pt *p2, int top, int bottom, int dir)
{
  FlutlPolygon *pg = st->polygon;
  polygon_push (st, p1, p2, top, bottom, dir);
  if (top < pg->extents[1])
    pg->extents[1] = top;
  if (bottom > pg->extents[3])
    pg->extents[3] = bottom;
  if (p1->x < pg->extents[0] || p1->x > pg->extents[2]) {
    int32_t x = p1->x;
    if (top != p1->y)
      x = edge_x_for_y (p1, p2, top);
    if (x < pg->extents[0])
      pg->extents[0] = x;
    if (x > pg->extents[2])
      pg->extents[2] = x;
  }
  if (p2->x < pg->extents[0] || p2->x > pg->extents[2]) {
    int32_t x = p2->x;
    if (bottom != p2->y)
      x = edge_x_for_y (p1, p2, bottom);
    if (x < pg->extents[0])
      pg->extents[0] = x;
    if (x > pg->extents[2])
      pg->extents[2] = x;
  }
}

static void
add_clipped_edge (Stroker *st, const Pt *p1, const Pt *p2, const int top, const int bottom,
    const int dir)
{
  const FlutlPolygon *pg = st->polygon;
  const Box limits = { { pg->limits[0], pg->limits[1] }, { pg->limits[2], pg->limits[3] } };
  Pt bot_left, top_right;
  int32_t top_y, bot_y, pleft, pright;

  if (top >= limits.p2.y)
    return;
  if (bottom <= limits.p1.y)
    return;
  bot_left.x = limits.p1.x;
  bot_left.y = limits.p2.y;
  top_right.x = limits.p2.x;
  top_right.y = limits.p1.y;
  top_y = top > limits.p1.y ? top : limits.p1.y;
  bot_y = bottom < limits.p2.y ? bottom : limits.p2.y;
  pleft = p1->x < p2->x ? p1->x : p2->x;
  pright = p1->x > p2->x ? p1->x : p2->x;

  if (limits.p1.x <= pleft && pright <= limits.p2.x) {
    add_edge (st, p1, p2, top_y, bot_y, dir);
  } else if (pright <= limits.p1.x) {
    add_edge (st, &limits.p1, &bot_left, top_y, bot_y, dir);
  } else if (limits.p2.x <= pleft) {
    add_edge (st, &top_right, &limits.p2, top_y, bot_y, dir);
  } else {
    int32_t left_y, right_y;
    const int top_left_to_bottom_right = (p1->x <= p2->x) == (p1->y <= p2->y);
    if (top_left_to_bottom_right) {
      if (pleft >= limits.p1.x) {
        left_y = top_y;
      } else {
        left_y = edge_y_for_x (p1, p2, limits.p1.x);
        if (edge_x_for_y (p1, p2, left_y) < limits.p1.x)
          left_y++;
      }
      left_y = left_y < bot_y ? left_y : bot_y;
      if (top_y < left_y) {
        add_edge (st, &limits.p1, &bot_left, top_y, left_y, dir);
        top_y = left_y;
      }
      if (pright <= limits.p2.x) {
        right_y = bot_y;
      } else {
        right_y = edge_y_for_x (p1, p2, limits.p2.x);
        if (edge_x_for_y (p1, p2, right_y) > limits.p2.x)
          right_y--;
      }
      right_y = right_y > top_y ? right_y : top_y;
      if (bot_y > right_y) {
        add_edge (st, &top_right, &limits.p2, right_y, bot_y, dir);
        bot_y = right_y;
      }
    } else {
      if (pright <= limits.p2.x) {
        right_y = top_y;
      } else {
        right_y = edge_y_for_x (p1, p2, limits.p2.x);
        if (edge_x_for_y (p1, p2, right_y) > limits.p2.x)
          right_y++;
      }
      right_y = right_y < bot_y ? right_y : bot_y;
      if (top_y < right_y) {
        add_edge (st, &top_right, &limits.p2, top_y, right_y, dir);
        top_y = right_y;
      }
      if (pleft >= limits.p1.x) {
        left_y = bot_y;
      } else {
        left_y = edge_y_for_x (p1, p2, limits.p1.x);
        if (edge_x_for_y (p1, p2, left_y) < limits.p1.x)
          left_y--;
      }
      left_y = left_y > top_y ? left_y : top_y;
      if (bot_y > left_y) {
        add_edge (st, &limits.p1, &bot_left, left_y, bot_y, dir);
        bot_y = left_y;
      }
    }
    if (top_y != bot_y)
      add_edge (st, p1, p2, top_y, bot_y, dir);
  }
}

static void
polygon_add_edge (Stroker *st, const Pt *p1, const Pt *p2, int dir)
{
  FlutlPolygon *pg = st->polygon;
  if (p1->y == p2->y)           /* horizontal edges do not count */
    return;
  /* (the polygon's extents had it no limits: what cairo_stroke_extents reports) */
  for (int k = 0; k < 2; k++) {
    const Pt *q = k ? p2 : p1;
    if (q->x < pg->uext[0]) pg->uext[0] = q->x;
    if (q->y < pg->uext[1]) pg->uext[1] = q->y;
    if (q->x > pg->uext[2]) pg->uext[2] = q->x;
    if (q->y > pg->uext[3]) pg->uext[3] = q->y;
  }
  if (p1->y > p2->y) {
    const Pt *t = p1;
    p1 = p2;
    p2 = t;
    dir = -dir;
  }
  if (pg->has_limits) {
    if (p2->y <= pg->limits[1])
      return;
    if (p1->y >= pg->limits[3])
      return;
    add_clipped_edge (st, p1, p2, p1->y, p2->y, dir);
  } else {
    add_edge (st, p1, p2, p1->y, p2->y, dir);
  }
}

static void
polygon_add_contour (Stroker *st, const Contour *c)
{
  if (c->n <= 1)
    return;
  const Pt *prev = &c->pts[0];
  for (int i = 0; i < c->n; i++) {
    polygon_add_edge (st, prev, &c->pts[i], c->direction);
    prev = &c->pts[i];
  }
}

/* ---- the pen (cairo-pen.c) ------------------------------------------------------------ */
static int
matrix_has_unity_scale (const cairo_matrix_t *m)
{
  const double eps = 1.0 / 256.0;
  const double det = m->xx * m->yy - m->yx * m->xy;
  if (fabs (det * det - 1.0) < eps) {
    if (fabs (m->xy) < eps && fabs (m->yx) < eps)
      return 1;
    if (fabs (m->xx) < eps && fabs (m->yy) < eps)
      return 1;
  }
  return 0;
}

static double
transformed_circle_major_axis (const cairo_matrix_t *m, double radius)
{
  if (matrix_has_unity_scale (m))
    return radius;
  const double a = m->xx, b = m->yx, c = m->xy, d = m->yy;
  const double i = a * a + b * b;
  const double j = c * c + d * d;
  const double f = 0.5 * (i + j);
  const double g = 0.5 * (i - j);
  const double h = a * c + b * d;
  return radius * sqrt (f + hypot (g, h));
}

static int
pen_vertices_needed (double tolerance, double radius, const cairo_matrix_t *m)
{
  const double major_axis = transformed_circle_major_axis (m, radius);
  int n;
  if (tolerance >= 4 * major_axis) {
    n = 1;
  } else if (tolerance >= major_axis) {
    n = 4;
  } else {
    n = (int) ceil (2 * M_PI / acos (1 - tolerance / major_axis));
    if (n % 2)
      n++;
    if (n < 4)
      n = 4;
  }
  return n;
}

static int
pen_init (Stroker *st, double radius, double tolerance, const cairo_matrix_t *ctm)
{
  const int reflect = (ctm->xx * ctm->yy - ctm->yx * ctm->xy) < 0.;
  st->pen_n = pen_vertices_needed (tolerance, radius, ctm);
  st->pen = malloc ((size_t) st->pen_n * sizeof *st->pen);
  if (!st->pen)
    return 0;
  for (int i = 0; i < st->pen_n; i++) {
    double theta = 2 * M_PI * i / (double) st->pen_n, dx, dy;
    if (reflect)
      theta = -theta;
    dx = radius * cos (theta);
    dy = radius * sin (theta);
    cairo_matrix_transform_distance (ctm, &dx, &dy);
    st->pen[i].point.x = fixed_from_double (dx);
    st->pen[i].point.y = fixed_from_double (dy);
  }
  for (int i = 0, i_prev = st->pen_n - 1; i < st->pen_n; i_prev = i++) {
    PenVertex *prev = &st->pen[i_prev], *v = &st->pen[i], *next = &st->pen[(i + 1) % st->pen_n];
    slope_init (&v->slope_cw, &prev->point, &v->point);
    slope_init (&v->slope_ccw, &v->point, &next->point);
  }
  return 1;
}

static void
pen_find_active_cw_vertices (const Stroker *st, const Slope *in, const Slope *out, int *start,
    int *stop)
{
  const int n = st->pen_n;
  int lo = 0, hi = n;
  int i = (lo + hi) >> 1;
  do {
    if (slope_compare (&st->pen[i].slope_cw, in) < 0)
      lo = i;
    else
      hi = i;
    i = (lo + hi) >> 1;
  } while (hi - lo > 1);
  if (slope_compare (&st->pen[i].slope_cw, in) < 0)
    if (++i == n)
      i = 0;
  *start = i;

  if (slope_compare (out, &st->pen[i].slope_ccw) >= 0) {
    lo = i;
    hi = i + n;
    i = (lo + hi) >> 1;
    do {
      int j = i;
      if (j >= n)
        j -= n;
      if (slope_compare (&st->pen[j].slope_cw, out) > 0)
        hi = i;
      else
        lo = i;
      i = (lo + hi) >> 1;
    } while (hi - lo > 1);
    if (i >= n)
      i -= n;
  }
  *stop = i;
}

static void
pen_find_active_ccw_vertices (const Stroker *st, const Slope *in, const Slope *out, int *start,
    int *stop)
{
  const int n = st->pen_n;
  int lo = 0, hi = n;
  int i = (lo + hi) >> 1;
  do {
    if (slope_compare (in, &st->pen[i].slope_ccw) < 0)
      lo = i;
    else
      hi = i;
    i = (lo + hi) >> 1;
  } while (hi - lo > 1);
  if (slope_compare (in, &st->pen[i].slope_ccw) < 0)
    if (++i == n)
      i = 0;
  *start = i;

  if (slope_compare (&st->pen[i].slope_cw, out) <= 0) {
    lo = i;
    hi = i + n;
    i = (lo + hi) >> 1;
    do {
      int j = i;
      if (j >= n)
        j -= n;
      if (slope_compare (out, &st->pen[j].slope_ccw) > 0)
        hi = i;
      else
        lo = i;
      i = (lo + hi) >> 1;
    } while (hi - lo > 1);
    if (i >= n)
      i -= n;
  }
  *stop = i;
}

/* ---- faces, joins (cairo-path-stroke-polygon.c) ------------------------------------------ */
static inline int
box_contains_point (const Box *b, const Pt *p)
{
  return p->x >= b->p1.x && p->x <= b->p2.x && p->y >= b->p1.y && p->y <= b->p2.y;
}

static void
add_fan (Stroker *st, const Slope *in_vector, const Slope *out_vector, const Pt *midpt,
    int clockwise, Contour *c)
{
  int start, stop;
  if (st->has_bounds && !box_contains_point (&st->bounds, midpt))
    return;
  if (clockwise) {
    pen_find_active_cw_vertices (st, in_vector, out_vector, &start, &stop);
    while (start != stop) {
      Pt p = *midpt;
      p.x += st->pen[start].point.x;
      p.y += st->pen[start].point.y;
      contour_add_point (st, c, &p);
      if (++start == st->pen_n)
        start = 0;
    }
  } else {
    pen_find_active_ccw_vertices (st, in_vector, out_vector, &start, &stop);
    while (start != stop) {
      Pt p = *midpt;
      p.x += st->pen[start].point.x;
      p.y += st->pen[start].point.y;
      contour_add_point (st, c, &p);
      if (start-- == 0)
        start += st->pen_n;
    }
  }
}

static inline int
join_is_clockwise (const Face *in, const Face *out)
{
  return slope_compare (&in->dev_vector, &out->dev_vector) < 0;
}

static int
slope_compare_sgn (double dx1, double dy1, double dx2, double dy2)
{
  const double c = (dx1 * dy2 - dx2 * dy1);
  if (c > 0)
    return 1;
  if (c < 0)
    return -1;
  return 0;
}

static void
inner_join (Stroker *st, const Face *in, const Face *out, int clockwise)
{
  const Pt *outpt;
  Contour *inner;
  if (clockwise) {
    inner = &st->ccw;
    outpt = &out->ccw;
  } else {
    inner = &st->cw;
    outpt = &out->cw;
  }
  contour_add_point (st, inner, &in->point);
  contour_add_point (st, inner, outpt);
}

static void
inner_close (Stroker *st, const Face *in, Face *out)
{
  const Pt *inpt;
  Contour *inner;
  if (join_is_clockwise (in, out)) {
    inner = &st->ccw;
    inpt = &out->ccw;
  } else {
    inner = &st->cw;
    inpt = &out->cw;
  }
  contour_add_point (st, inner, &in->point);
  contour_add_point (st, inner, inpt);
  if (inner->n)
    inner->pts[0] = inner->pts[inner->n - 1];
}

/* the miter of two faces: 1 and the point if it is to be used */
static int
miter_point (const Stroker *st, const Face *in, const Face *out, const Pt *inpt, const Pt *outpt,
    Pt *p)
{
  const double in_dot_out = in->dev_slope_x * out->dev_slope_x + in->dev_slope_y * out->dev_slope_y;
  const double ml = st->miter_limit;
  if (2 <= ml * ml * (1 + in_dot_out)) {
    double x1, y1, x2, y2, mx, my, dx1, dx2, dy1, dy2, ix, iy, fdx1, fdy1, fdx2, fdy2, mdx, mdy;
    x1 = fixed_to_double (inpt->x);
    y1 = fixed_to_double (inpt->y);
    dx1 = in->dev_slope_x;
    dy1 = in->dev_slope_y;
    x2 = fixed_to_double (outpt->x);
    y2 = fixed_to_double (outpt->y);
    dx2 = out->dev_slope_x;
    dy2 = out->dev_slope_y;
    my = (((x2 - x1) * dy1 * dy2 - y2 * dx2 * dy1 + y1 * dx1 * dy2) / (dx1 * dy2 - dx2 * dy1));
    if (fabs (dy1) >= fabs (dy2))
      mx = (my - y1) * dx1 / dy1 + x1;
    else
      mx = (my - y2) * dx2 / dy2 + x2;
    ix = fixed_to_double (in->point.x);
    iy = fixed_to_double (in->point.y);
    fdx1 = x1 - ix;
    fdy1 = y1 - iy;
    fdx2 = x2 - ix;
    fdy2 = y2 - iy;
    mdx = mx - ix;
    mdy = my - iy;
    if (slope_compare_sgn (fdx1, fdy1, mdx, mdy) != slope_compare_sgn (fdx2, fdy2, mdx, mdy)) {
      p->x = fixed_from_double (mx);
      p->y = fixed_from_double (my);
      return 1;
    }
  }
  return 0;
}

static void
outer_close (Stroker *st, const Face *in, const Face *out)
{
  const Pt *inpt, *outpt;
  Contour *outer;
  int clockwise;
  if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x &&
      in->ccw.y == out->ccw.y)
    return;
  clockwise = join_is_clockwise (in, out);
  if (clockwise) {
    inpt = &in->cw;
    outpt = &out->cw;
    outer = &st->cw;
  } else {
    inpt = &in->ccw;
    outpt = &out->ccw;
    outer = &st->ccw;
  }
  /* (within_tolerance is constant FALSE in 1.16) */
  switch (st->line_join) {
    case CAIRO_LINE_JOIN_ROUND:
      if ((in->dev_slope_x * out->dev_slope_x + in->dev_slope_y * out->dev_slope_y) <
          st->spline_cusp_tolerance) {
        add_fan (st, &in->dev_vector, &out->dev_vector, &in->point, clockwise, outer);
        break;
      }
      /* fall through */
    case CAIRO_LINE_JOIN_MITER:
    default: {
      Pt p;
      if (miter_point (st, in, out, inpt, outpt, &p)) {
        if (outer->n) {
          outer->pts[outer->n - 1] = p;
          outer->pts[0] = p;
        }
        return;
      }
      break;
    }
    case CAIRO_LINE_JOIN_BEVEL:
      break;
  }
  contour_add_point (st, outer, outpt);
}

static void
outer_join (Stroker *st, const Face *in, const Face *out, int clockwise)
{
  const Pt *inpt, *outpt;
  Contour *outer;
  if (in->cw.x == out->cw.x && in->cw.y == out->cw.y && in->ccw.x == out->ccw.x &&
      in->ccw.y == out->ccw.y)
    return;
  if (clockwise) {
    inpt = &in->cw;
    outpt = &out->cw;
    outer = &st->cw;
  } else {
    inpt = &in->ccw;
    outpt = &out->ccw;
    outer = &st->ccw;
  }
  switch (st->line_join) {
    case CAIRO_LINE_JOIN_ROUND:
      add_fan (st, &in->dev_vector, &out->dev_vector, &in->point, clockwise, outer);
      break;
    case CAIRO_LINE_JOIN_MITER:
    default: {
      Pt p;
      if (miter_point (st, in, out, inpt, outpt, &p)) {
        if (outer->n)
          outer->pts[outer->n - 1] = p;
        return;
      }
      break;
    }
    case CAIRO_LINE_JOIN_BEVEL:
      break;
  }
  contour_add_point (st, outer, outpt);
}

static inline double
normalize_slope (double *dx, double *dy)
{
  const double dx0 = *dx, dy0 = *dy;
  double mag;
  if (dx0 == 0.0) {
    *dx = 0.0;
    if (dy0 > 0.0) {
      mag = dy0;
      *dy = 1.0;
    } else {
      mag = -dy0;
      *dy = -1.0;
    }
  } else if (dy0 == 0.0) {
    *dy = 0.0;
    if (dx0 > 0.0) {
      mag = dx0;
      *dx = 1.0;
    } else {
      mag = -dx0;
      *dx = -1.0;
    }
  } else {
    mag = hypot (dx0, dy0);
    *dx = dx0 / mag;
    *dy = dy0 / mag;
  }
  return mag;
}

static void
compute_face (const Pt *point, const Slope *dev_slope, Stroker *st, Face *face)
{
  double face_dx, face_dy;
  Pt offset_ccw, offset_cw;
  double slope_dx, slope_dy;

  slope_dx = fixed_to_double (dev_slope->dx);
  slope_dy = fixed_to_double (dev_slope->dy);
  (void) normalize_slope (&slope_dx, &slope_dy);
  face->dev_slope_x = slope_dx;
  face->dev_slope_y = slope_dy;

  if (!st->ctm_inverse_is_identity) {
    cairo_matrix_transform_distance (st->ctm_inverse, &slope_dx, &slope_dy);
    (void) normalize_slope (&slope_dx, &slope_dy);
    if (st->ctm_det_positive) {
      face_dx = -slope_dy * st->half_line_width;
      face_dy = slope_dx * st->half_line_width;
    } else {
      face_dx = slope_dy * st->half_line_width;
      face_dy = -slope_dx * st->half_line_width;
    }
    cairo_matrix_transform_distance (st->ctm, &face_dx, &face_dy);
  } else {
    face_dx = -slope_dy * st->half_line_width;
    face_dy = slope_dx * st->half_line_width;
  }

  offset_ccw.x = fixed_from_double (face_dx);
  offset_ccw.y = fixed_from_double (face_dy);
  offset_cw.x = -offset_ccw.x;
  offset_cw.y = -offset_ccw.y;

  face->ccw = *point;
  face->ccw.x += offset_ccw.x;
  face->ccw.y += offset_ccw.y;
  face->point = *point;
  face->cw = *point;
  face->cw.x += offset_cw.x;
  face->cw.y += offset_cw.y;
  face->dev_vector = *dev_slope;
}

static void
add_cap (Stroker *st, const Face *f, Contour *c)
{
  /* butt caps only (the default; the reference never sets another) */
  contour_add_point (st, c, &f->cw);
}

static void
add_leading_cap (Stroker *st, const Face *face, Contour *c)
{
  Face reversed = *face;
  Pt t;
  /* the initial cap needs an outward facing vector: reverse everything */
  reversed.dev_vector.dx = -reversed.dev_vector.dx;
  reversed.dev_vector.dy = -reversed.dev_vector.dy;
  t = reversed.cw;
  reversed.cw = reversed.ccw;
  reversed.ccw = t;
  add_cap (st, &reversed, c);
}

static void
add_caps (Stroker *st)
{
  /* (a degenerate sub-path only draws something with round caps) */
  if (st->has_current_face)
    add_cap (st, &st->current_face, &st->ccw);
  polygon_add_contour (st, &st->ccw);
  st->ccw.n = 0;
  if (st->has_first_face) {
    contour_push (st, &st->ccw, &st->first_face.cw);
    add_leading_cap (st, &st->first_face, &st->ccw);
    polygon_add_contour (st, &st->ccw);
    st->ccw.n = 0;
  }
  polygon_add_contour (st, &st->cw);
  st->cw.n = 0;
}

/* ---- the path's verbs -------------------------------------------------------------------- */
static void
stroker_move_to (Stroker *st, const Pt *point)
{
  add_caps (st);
  st->has_first_face = 0;
  st->has_current_face = 0;
  st->has_initial_sub_path = 0;
  st->first_point = *point;
  st->current_face.point = *point;
}

static void
stroker_line_to (Stroker *st, const Pt *point)
{
  Face start;
  Pt *p1 = &st->current_face.point;
  Slope dev_slope;

  st->has_initial_sub_path = 1;
  if (p1->x == point->x && p1->y == point->y)
    return;
  slope_init (&dev_slope, p1, point);
  compute_face (p1, &dev_slope, st, &start);

  if (st->has_current_face) {
    int clockwise = slope_compare (&st->current_face.dev_vector, &start.dev_vector);
    if (clockwise) {
      clockwise = clockwise < 0;
      /* (within_tolerance is constant FALSE: always joined) */
      outer_join (st, &st->current_face, &start, clockwise);
      inner_join (st, &st->current_face, &start, clockwise);
    }
  } else {
    if (!st->has_first_face) {
      st->first_face = start;
      st->has_first_face = 1;
    }
    st->has_current_face = 1;
    contour_add_point (st, &st->cw, &start.cw);
    contour_add_point (st, &st->ccw, &start.ccw);
  }

  st->current_face = start;
  st->current_face.point = *point;
  st->current_face.ccw.x += dev_slope.dx;
  st->current_face.ccw.y += dev_slope.dy;
  st->current_face.cw.x += dev_slope.dx;
  st->current_face.cw.y += dev_slope.dy;
  contour_add_point (st, &st->cw, &st->current_face.cw);
  contour_add_point (st, &st->ccw, &st->current_face.ccw);
}

static void
spline_to (Stroker *st, const Pt *point, const Slope *tangent)
{
  Face face;
  if ((tangent->dx | tangent->dy) == 0) {
    Contour *outer;
    Pt t;
    int clockwise;
    face = st->current_face;
    face.dev_vector.dx = -face.dev_vector.dx;
    face.dev_vector.dy = -face.dev_vector.dy;
    t = face.cw;
    face.cw = face.ccw;
    face.ccw = t;
    clockwise = join_is_clockwise (&st->current_face, &face);
    outer = clockwise ? &st->cw : &st->ccw;
    add_fan (st, &st->current_face.dev_vector, &face.dev_vector, &st->current_face.point,
        clockwise, outer);
  } else {
    compute_face (point, tangent, st, &face);
    if ((face.dev_slope_x * st->current_face.dev_slope_x +
            face.dev_slope_y * st->current_face.dev_slope_y) < st->spline_cusp_tolerance) {
      Contour *outer;
      const int clockwise = join_is_clockwise (&st->current_face, &face);
      st->current_face.cw.x += face.point.x - st->current_face.point.x;
      st->current_face.cw.y += face.point.y - st->current_face.point.y;
      contour_add_point (st, &st->cw, &st->current_face.cw);
      st->current_face.ccw.x += face.point.x - st->current_face.point.x;
      st->current_face.ccw.y += face.point.y - st->current_face.point.y;
      contour_add_point (st, &st->ccw, &st->current_face.ccw);
      outer = clockwise ? &st->cw : &st->ccw;
      add_fan (st, &st->current_face.dev_vector, &face.dev_vector, &st->current_face.point,
          clockwise, outer);
    }
    contour_add_point (st, &st->cw, &face.cw);
    contour_add_point (st, &st->ccw, &face.ccw);
  }
  st->current_face = face;
}

/* ---- cairo-spline.c ------------------------------------------------------------------ */
typedef struct { Pt a, b, c, d; } Knots;

typedef struct
{
  Stroker *st;
  Knots knots;
  Slope initial_slope, final_slope;
  Pt last_point;
  int filling;                  /* cairo_fill: the points go to filler_line_to */
} Spline;

static int
spline_init (Spline *s, Stroker *st, const Pt *a, const Pt *b, const Pt *c, const Pt *d)
{
  /* if both tangents are zero, this is just a straight line */
  if (a->x == b->x && a->y == b->y && c->x == d->x && c->y == d->y)
    return 0;
  s->st = st;
  s->filling = 0;
  s->knots.a = *a;
  s->knots.b = *b;
  s->knots.c = *c;
  s->knots.d = *d;
  if (a->x != b->x || a->y != b->y)
    slope_init (&s->initial_slope, &s->knots.a, &s->knots.b);
  else if (a->x != c->x || a->y != c->y)
    slope_init (&s->initial_slope, &s->knots.a, &s->knots.c);
  else if (a->x != d->x || a->y != d->y)
    slope_init (&s->initial_slope, &s->knots.a, &s->knots.d);
  else
    return 0;
  if (c->x != d->x || c->y != d->y)
    slope_init (&s->final_slope, &s->knots.c, &s->knots.d);
  else if (b->x != d->x || b->y != d->y)
    slope_init (&s->final_slope, &s->knots.b, &s->knots.d);
  else
    return 0;
  return 1;
}

static void polygon_add_edge (Stroker *st, const Pt *p1, const Pt *p2, int dir);

/* _cairo_filler_line_to (cairo-path-fill.c): _cairo_polygon_add_external_edge */
static void
filler_line_to (Stroker *st, const Pt *point)
{
  polygon_add_edge (st, &st->fill_current, point, 1);
  st->fill_current = *point;
}

static void
spline_add_point (Spline *s, const Pt *point, const Pt *knot)
{
  Pt *prev = &s->last_point;
  Slope slope;
  if (prev->x == point->x && prev->y == point->y)
    return;
  slope_init (&slope, point, knot);
  s->last_point = *point;
  if (s->filling)
    filler_line_to (s->st, point);
  else
    spline_to (s->st, point, &slope);
}

static inline void
lerp_half (const Pt *a, const Pt *b, Pt *r)
{
  r->x = a->x + ((b->x - a->x) >> 1);
  r->y = a->y + ((b->y - a->y) >> 1);
}

static void
de_casteljau (Knots *s1, Knots *s2)
{
  Pt ab, bc, cd, abbc, bccd, fin;
  lerp_half (&s1->a, &s1->b, &ab);
  lerp_half (&s1->b, &s1->c, &bc);
  lerp_half (&s1->c, &s1->d, &cd);
  lerp_half (&ab, &bc, &abbc);
  lerp_half (&bc, &cd, &bccd);
  lerp_half (&abbc, &bccd, &fin);
  s2->a = fin;
  s2->b = bccd;
  s2->c = cd;
  s2->d = s1->d;
  s1->b = ab;
  s1->c = abbc;
  s1->d = fin;
}

static double
spline_error_squared (const Knots *k)
{
  double bdx, bdy, berr, cdx, cdy, cerr;
  bdx = fixed_to_double (k->b.x - k->a.x);
  bdy = fixed_to_double (k->b.y - k->a.y);
  cdx = fixed_to_double (k->c.x - k->a.x);
  cdy = fixed_to_double (k->c.y - k->a.y);
  if (k->a.x != k->d.x || k->a.y != k->d.y) {
    double dx, dy, u, v;
    dx = fixed_to_double (k->d.x - k->a.x);
    dy = fixed_to_double (k->d.y - k->a.y);
    v = dx * dx + dy * dy;
    u = bdx * dx + bdy * dy;
    if (u <= 0) {
    } else if (u >= v) {
      bdx -= dx;
      bdy -= dy;
    } else {
      bdx -= u / v * dx;
      bdy -= u / v * dy;
    }
    u = cdx * dx + cdy * dy;
    if (u <= 0) {
    } else if (u >= v) {
      cdx -= dx;
      cdy -= dy;
    } else {
      cdx -= u / v * dx;
      cdy -= u / v * dy;
    }
  }
  berr = bdx * bdx + bdy * bdy;
  cerr = cdx * cdx + cdy * cdy;
  return berr > cerr ? berr : cerr;
}

static void
spline_decompose_into (Knots *s1, double tolerance_squared, Spline *result)
{
  Knots s2;
  if (spline_error_squared (s1) < tolerance_squared) {
    spline_add_point (result, &s1->a, &s1->b);
    return;
  }
  de_casteljau (s1, &s2);
  spline_decompose_into (s1, tolerance_squared, result);
  spline_decompose_into (&s2, tolerance_squared, result);
}

static int
spline_intersects (const Pt *a, const Pt *b, const Pt *c, const Pt *d, const Box *box)
{
  Box bounds;
  if (box_contains_point (box, a) || box_contains_point (box, b) || box_contains_point (box, c) ||
      box_contains_point (box, d))
    return 1;
  bounds.p2 = bounds.p1 = *a;
  const Pt *pts[3] = { b, c, d };
  for (int i = 0; i < 3; i++) {
    if (pts[i]->x < bounds.p1.x) bounds.p1.x = pts[i]->x;
    if (pts[i]->x > bounds.p2.x) bounds.p2.x = pts[i]->x;
    if (pts[i]->y < bounds.p1.y) bounds.p1.y = pts[i]->y;
    if (pts[i]->y > bounds.p2.y) bounds.p2.y = pts[i]->y;
  }
  if (bounds.p2.x <= box->p1.x || bounds.p1.x >= box->p2.x || bounds.p2.y <= box->p1.y ||
      bounds.p1.y >= box->p2.y)
    return 0;
  return 1;
}

static void
stroker_curve_to (Stroker *st, const Pt *b, const Pt *c, const Pt *d)
{
  Spline spline;
  Face face;
  if (st->has_bounds && !spline_intersects (&st->current_face.point, b, c, d, &st->bounds)) {
    stroker_line_to (st, d);
    return;
  }
  if (!spline_init (&spline, st, &st->current_face.point, b, c, d)) {
    stroker_line_to (st, d);
    return;
  }
  compute_face (&st->current_face.point, &spline.initial_slope, st, &face);
  if (st->has_current_face) {
    const int clockwise = join_is_clockwise (&st->current_face, &face);
    outer_join (st, &st->current_face, &face, clockwise);
    inner_join (st, &st->current_face, &face, clockwise);
  } else {
    if (!st->has_first_face) {
      st->first_face = face;
      st->has_first_face = 1;
    }
    st->has_current_face = 1;
    contour_add_point (st, &st->cw, &face.cw);
    contour_add_point (st, &st->ccw, &face.ccw);
  }
  st->current_face = face;

  {
    Knots s1 = spline.knots;
    spline.last_point = s1.a;
    spline_decompose_into (&s1, st->tolerance * st->tolerance, &spline);
    spline_to (st, &spline.knots.d, &spline.final_slope);
  }
}

static void
stroker_close_path (Stroker *st)
{
  stroker_line_to (st, &st->first_point);
  if (st->has_first_face && st->has_current_face) {
    outer_close (st, &st->current_face, &st->first_face);
    inner_close (st, &st->current_face, &st->first_face);
    polygon_add_contour (st, &st->cw);
    polygon_add_contour (st, &st->ccw);
    st->cw.n = 0;
    st->ccw.n = 0;
  } else {
    add_caps (st);
  }
  st->has_initial_sub_path = 0;
  st->has_first_face = 0;
  st->has_current_face = 0;
}

/* ---- cairo-path-fixed.c: what a path becomes when it is appended to a context ---------- */
enum { OP_MOVE_TO, OP_LINE_TO, OP_CURVE_TO, OP_CLOSE_PATH };

typedef struct
{
  uint8_t *ops;
  Pt *pts;
  int n_ops, n_pts, cap_ops, cap_pts;
  Pt current_point, last_move_point;
  int has_current_point, needs_move_to, has_curve_to;
  int stroke_is_rectilinear;
  int has_extents;
  Box extents;                  /* cairo_path_fixed_t.extents: points, and the bounds of curves */
  int failed;
} PathFixed;

static inline void
box_add_point (Box *b, const Pt *p)
{
  if (p->x < b->p1.x)
    b->p1.x = p->x;
  if (p->x > b->p2.x)
    b->p2.x = p->x;
  if (p->y < b->p1.y)
    b->p1.y = p->y;
  if (p->y > b->p2.y)
    b->p2.y = p->y;
}

/* cairo-spline.c:_cairo_spline_bound feeding _cairo_box_add_spline_point: the curve's
 * end points and its extreme points in x and y (roots of the derivative in (0, 1)) */
static void
box_add_curve_bound (Box *box, const Pt *p0, const Pt *p1, const Pt *p2, const Pt *p3)
{
  const double x0 = fixed_to_double (p0->x), y0 = fixed_to_double (p0->y);
  const double x1 = fixed_to_double (p1->x), y1 = fixed_to_double (p1->y);
  const double x2 = fixed_to_double (p2->x), y2 = fixed_to_double (p2->y);
  const double x3 = fixed_to_double (p3->x), y3 = fixed_to_double (p3->y);
  double a, b, c, t[4];
  int t_num = 0;
#define ADD(t0) \
  { \
    const double _t0 = (t0); \
    if (0 < _t0 && _t0 < 1) \
      t[t_num++] = _t0; \
  }
#define FIND_EXTREMES(a, b, c) \
  { \
    if (a == 0) { \
      if (b != 0) \
        ADD (-c / (2 * b)); \
    } else { \
      const double b2 = b * b; \
      const double delta = b2 - a * c; \
      if (delta > 0) { \
        int feasible; \
        const double _2ab = 2 * a * b; \
        if (_2ab >= 0) \
          feasible = delta > b2 && delta < a * a + b2 + _2ab; \
        else if (-b / a >= 1) \
          feasible = delta < b2 && delta > a * a + b2 + _2ab; \
        else \
          feasible = delta < b2 || delta < a * a + b2 + _2ab; \
        if (feasible) { \
          const double sqrt_delta = sqrt (delta); \
          ADD ((-b - sqrt_delta) / a); \
          ADD ((-b + sqrt_delta) / a); \
        } \
      } else if (delta == 0) { \
        ADD (-b / a); \
      } \
    } \
  }
  a = -x0 + 3 * x1 - 3 * x2 + x3;
  b = x0 - 2 * x1 + x2;
  c = -x0 + x1;
  FIND_EXTREMES (a, b, c);
  a = -y0 + 3 * y1 - 3 * y2 + y3;
  b = y0 - 2 * y1 + y2;
  c = -y0 + y1;
  FIND_EXTREMES (a, b, c);
#undef ADD
#undef FIND_EXTREMES
  box_add_point (box, p0);
  for (int i = 0; i < t_num; i++) {
    Pt p;
    const double t_1_0 = t[i], t_0_1 = 1 - t_1_0;
    const double t_2_0 = t_1_0 * t_1_0, t_0_2 = t_0_1 * t_0_1;
    const double t_3_0 = t_2_0 * t_1_0, t_2_1_3 = t_2_0 * t_0_1 * 3, t_1_2_3 = t_1_0 * t_0_2 * 3,
        t_0_3 = t_0_1 * t_0_2;
    const double x = x0 * t_0_3 + x1 * t_1_2_3 + x2 * t_2_1_3 + x3 * t_3_0;
    const double y = y0 * t_0_3 + y1 * t_1_2_3 + y2 * t_2_1_3 + y3 * t_3_0;
    p.x = fixed_from_double (x);
    p.y = fixed_from_double (y);
    box_add_point (box, &p);
  }
  box_add_point (box, p3);
}

/* _cairo_box_add_curve_to for a caller that keeps a path's extents itself (the
 * recorder's composed outline paths): box {x0, y0, x1, y1}, p = the curve's four points */
void
flutl_stroker_box_add_curve (int32_t box[4], const int32_t p[8])
{
  Box b = { { box[0], box[1] }, { box[2], box[3] } };
  const Pt p0 = { p[0], p[1] }, p1 = { p[2], p[3] }, p2 = { p[4], p[5] }, p3 = { p[6], p[7] };
  box_add_curve_bound (&b, &p0, &p1, &p2, &p3);
  box[0] = b.p1.x;
  box[1] = b.p1.y;
  box[2] = b.p2.x;
  box[3] = b.p2.y;
}

static void
pf_add (PathFixed *pf, int op, const Pt *pts, int n)
{
  if (pf->n_ops == pf->cap_ops) {
    int cap = pf->cap_ops ? pf->cap_ops * 2 : 512;
    uint8_t *no = realloc (pf->ops, (size_t) cap);
    if (!no) {
      pf->failed = 1;
      return;
    }
    pf->ops = no;
    pf->cap_ops = cap;
  }
  if (pf->n_pts + n > pf->cap_pts) {
    int cap = pf->cap_pts ? pf->cap_pts * 2 : 1024;
    Pt *np = realloc (pf->pts, (size_t) cap * sizeof *np);
    if (!np) {
      pf->failed = 1;
      return;
    }
    pf->pts = np;
    pf->cap_pts = cap;
  }
  pf->ops[pf->n_ops++] = (uint8_t) op;
  for (int i = 0; i < n; i++)
    pf->pts[pf->n_pts++] = pts[i];
}

static inline int
pf_last_op (const PathFixed *pf)
{
  return pf->n_ops ? pf->ops[pf->n_ops - 1] : -1;
}

static void
pf_drop_line_to (PathFixed *pf)
{
  pf->n_ops--;
  pf->n_pts--;
}

static void
pf_move_to (PathFixed *pf, int32_t x, int32_t y)
{
  /* _cairo_path_fixed_new_sub_path */
  if (!pf->needs_move_to)
    pf->needs_move_to = 1;
  pf->has_current_point = 1;
  pf->current_point.x = x;
  pf->current_point.y = y;
  pf->last_move_point = pf->current_point;
}

static void
pf_move_to_apply (PathFixed *pf)
{
  if (!pf->needs_move_to)
    return;
  pf->needs_move_to = 0;
  if (pf->has_extents) {
    box_add_point (&pf->extents, &pf->current_point);
  } else {
    pf->extents.p1 = pf->extents.p2 = pf->current_point;
    pf->has_extents = 1;
  }
  pf->last_move_point = pf->current_point;
  pf_add (pf, OP_MOVE_TO, &pf->current_point, 1);
}

static void
pf_line_to (PathFixed *pf, int32_t x, int32_t y)
{
  Pt point = { x, y };
  if (!pf->has_current_point) {
    pf_move_to (pf, x, y);
    return;
  }
  pf_move_to_apply (pf);
  /* a line_to that does not move is dropped, unless it follows a move_to (a dot) */
  if (pf_last_op (pf) != OP_MOVE_TO) {
    if (x == pf->current_point.x && y == pf->current_point.y)
      return;
  }
  /* a line_to that continues the previous one replaces its point */
  if (pf_last_op (pf) == OP_LINE_TO) {
    const Pt *p = &pf->pts[pf->n_pts - 2];
    if (p->x == pf->current_point.x && p->y == pf->current_point.y) {
      pf_drop_line_to (pf);     /* the previous element was degenerate */
    } else {
      Slope prev, self;
      slope_init (&prev, p, &pf->current_point);
      slope_init (&self, &pf->current_point, &point);
      /* _cairo_slope_equal && !_cairo_slope_backwards */
      if ((int64_t) prev.dy * self.dx == (int64_t) self.dy * prev.dx &&
          !((int64_t) prev.dx * self.dx + (int64_t) prev.dy * self.dy < 0))
        pf_drop_line_to (pf);
    }
  }
  if (pf->stroke_is_rectilinear)
    pf->stroke_is_rectilinear = pf->current_point.x == x || pf->current_point.y == y;
  pf->current_point = point;
  box_add_point (&pf->extents, &point);
  pf_add (pf, OP_LINE_TO, &point, 1);
}

static void
pf_curve_to (PathFixed *pf, int32_t x0, int32_t y0, int32_t x1, int32_t y1, int32_t x2, int32_t y2)
{
  Pt point[3];
  /* a curve that does not move at all is a line_to */
  if (pf->current_point.x == x2 && pf->current_point.y == y2) {
    if (x1 == x2 && x0 == x2 && y1 == y2 && y0 == y2) {
      pf_line_to (pf, x2, y2);
      return;
    }
  }
  if (!pf->has_current_point)
    pf_move_to (pf, x0, y0);
  pf_move_to_apply (pf);
  if (pf_last_op (pf) == OP_LINE_TO) {
    const Pt *p = &pf->pts[pf->n_pts - 2];
    if (p->x == pf->current_point.x && p->y == pf->current_point.y)
      pf_drop_line_to (pf);
  }
  point[0].x = x0; point[0].y = y0;
  point[1].x = x1; point[1].y = y1;
  point[2].x = x2; point[2].y = y2;
  /* _cairo_box_add_curve_to */
  box_add_point (&pf->extents, &point[2]);
  if (!box_contains_point (&pf->extents, &point[0]) || !box_contains_point (&pf->extents, &point[1]))
    box_add_curve_bound (&pf->extents, &pf->current_point, &point[0], &point[1], &point[2]);
  pf->current_point = point[2];
  pf->has_curve_to = 1;
  pf->stroke_is_rectilinear = 0;
  pf_add (pf, OP_CURVE_TO, point, 3);
}

static void
pf_close_path (PathFixed *pf)
{
  if (!pf->has_current_point)
    return;
  /* a line_to, to resolve any degeneracy; taken away again if it was added */
  pf_line_to (pf, pf->last_move_point.x, pf->last_move_point.y);
  if (pf_last_op (pf) == OP_LINE_TO)
    pf_drop_line_to (pf);
  pf->needs_move_to = 1;
  pf_add (pf, OP_CLOSE_PATH, NULL, 0);
}

/* ---- entry point ---------------------------------------------------------------------- */
void
flutl_polygon_init (FlutlPolygon *pg)
{
  memset (pg, 0, sizeof *pg);
  pg->extents[0] = pg->extents[1] = INT32_MAX;
  pg->extents[2] = pg->extents[3] = INT32_MIN;
  pg->uext[0] = pg->uext[1] = INT32_MAX;
  pg->uext[2] = pg->uext[3] = INT32_MIN;
}

void
flutl_polygon_fini (FlutlPolygon *pg)
{
  free (pg->edges);
  pg->edges = NULL;
  pg->n_edges = pg->cap = 0;
}

static int path_to_polygon (const cairo_path_t *path, const cairo_matrix_t *ctm, double line_width,
    double miter_limit, cairo_line_join_t line_join, double tolerance, const int target[4],
    FlutlPolygon *polygon, int fill);

int
flutl_stroke_to_polygon (const cairo_path_t *path, const cairo_matrix_t *ctm, double line_width,
    double miter_limit, cairo_line_join_t line_join, double tolerance, const int target[4],
    FlutlPolygon *polygon)
{
  return path_to_polygon (path, ctm, line_width, miter_limit, line_join, tolerance, target, polygon, 0);
}

/* cairo_fill of the path (in device coordinates) as the polygon Cairo makes of it
 * (cairo-path-fill.c: _cairo_path_fixed_fill_to_polygon; cairo-spans-compositor.c:
 * _cairo_spans_compositor_fill): every sub-path closed, curves flattened to `tolerance`,
 * edges cut to the composite's box where the path's extents reach beyond the target.
 * FLUTL_STROKER_RECTILINEAR: a path of axis-aligned segments only, which Cairo fills as
 * boxes (another rasteriser).  Pinned against the real cairo_fill: oracle/stroke_check.c
 * modes 4-7. */
int
flutl_fill_to_polygon (const cairo_path_t *path, double tolerance, const int target[4],
    FlutlPolygon *polygon)
{
  cairo_matrix_t id;
  cairo_matrix_init_identity (&id);
  return path_to_polygon (path, &id, 1.0, 10.0, CAIRO_LINE_JOIN_MITER, tolerance, target, polygon, 1);
}

static int
path_to_polygon (const cairo_path_t *path, const cairo_matrix_t *ctm, double line_width,
    double miter_limit, cairo_line_join_t line_join, double tolerance, const int target[4],
    FlutlPolygon *polygon, int fill)
{
  Stroker st;
  PathFixed pf;
  cairo_matrix_t inverse = *ctm;
  int rc = FLUTL_STROKER_OK;

  if (!path || path->status != CAIRO_STATUS_SUCCESS)
    return FLUTL_STROKER_UNSUPPORTED;
  /* Round and bevel joins are restated above and match Cairo's polygon, but Cairo
   * then converts the polygon inside its APPROXIMATE stroke extents (path extents
   * grown by half a line width -- unscaled when the CTM's determinant happens to be
   * +-1, _cairo_matrix_has_unity_scale), which can cut such a stroke.  With miter
   * joins those extents grow by sqrt(2) * miter limit line widths and never cut.  The
   * reference never sets a join (the default is miter). */
  if (line_join != CAIRO_LINE_JOIN_MITER || !(miter_limit >= 1.0) || !(line_width > 0.0))
    return FLUTL_STROKER_UNSUPPORTED;
  /* ... and for the same reason only a CTM that is a similarity (translation,
   * rotation, reflection, uniform scale): under an anisotropic one the miter limit is
   * tested on device-space angles while the pen is a user-space circle, and a miter
   * can reach beyond those approximate extents (seen: 2 of 160 000 random strokes).
   * The element's CTMs at stroke time are translations and quarter turns
   * (gstttmlrender.c:1286-1340; a font's horizontal scale is in the path already). */
  if (fabs (ctm->xx) != fabs (ctm->yy) || fabs (ctm->xy) != fabs (ctm->yx) ||
      ctm->xx * ctm->xy + ctm->yx * ctm->yy != 0.0)
    return FLUTL_STROKER_UNSUPPORTED;
  if (cairo_matrix_invert (&inverse) != CAIRO_STATUS_SUCCESS)
    return FLUTL_STROKER_UNSUPPORTED;

  /* the path as the context holds it (24.8 fixed point, cairo-path-fixed.c) */
  memset (&pf, 0, sizeof pf);
  pf.needs_move_to = 1;
  pf.stroke_is_rectilinear = 1;
  for (int i = 0; i < path->num_data; i += path->data[i].header.length) {
    const cairo_path_data_t *d = &path->data[i];
    switch (d->header.type) {
      case CAIRO_PATH_MOVE_TO:
        pf_move_to (&pf, fixed_from_double (d[1].point.x), fixed_from_double (d[1].point.y));
        break;
      case CAIRO_PATH_LINE_TO:
        pf_line_to (&pf, fixed_from_double (d[1].point.x), fixed_from_double (d[1].point.y));
        break;
      case CAIRO_PATH_CURVE_TO:
        pf_curve_to (&pf, fixed_from_double (d[1].point.x), fixed_from_double (d[1].point.y),
            fixed_from_double (d[2].point.x), fixed_from_double (d[2].point.y),
            fixed_from_double (d[3].point.x), fixed_from_double (d[3].point.y));
        break;
      case CAIRO_PATH_CLOSE_PATH:
        pf_close_path (&pf);
        break;
    }
  }
  if (pf.failed) {
    free (pf.ops);
    free (pf.pts);
    return FLUTL_STROKER_ENOMEM;
  }
  if (pf.stroke_is_rectilinear && pf.n_ops) {
    /* Cairo strokes such a path as boxes (_cairo_path_fixed_stroke_rectilinear_to_boxes),
     * a different rasteriser: left to the caller */
    free (pf.ops);
    free (pf.pts);
    return FLUTL_STROKER_RECTILINEAR;
  }

  memset (&st, 0, sizeof st);
  st.polygon = polygon;
  st.ctm = ctm;
  st.ctm_inverse = &inverse;
  st.ctm_inverse_is_identity = inverse.xx == 1.0 && inverse.yx == 0.0 && inverse.xy == 0.0 &&
      inverse.yy == 1.0 && inverse.x0 == 0.0 && inverse.y0 == 0.0;
  st.tolerance = tolerance;
  st.half_line_width = line_width / 2.;
  st.miter_limit = miter_limit;
  st.line_join = line_join;
  st.spline_cusp_tolerance = 1 - tolerance / st.half_line_width;
  st.spline_cusp_tolerance *= st.spline_cusp_tolerance;
  st.spline_cusp_tolerance *= 2;
  st.spline_cusp_tolerance -= 1;
  st.ctm_det_positive = (ctm->xx * ctm->yy - ctm->yx * ctm->xy) >= 0.0;
  st.cw.direction = 1;
  st.ccw.direction = -1;

  polygon->box[0] = polygon->box[1] = polygon->box[2] = polygon->box[3] = 0;
  if (!pf.has_extents)          /* an empty path */
    goto out;
  {
    /* What Cairo composites within (cairo-composite-rectangles.c): the surface cut by
     * the clip (`target`) and by the stroke's APPROXIMATE extents -- the path's extents
     * grown by _cairo_stroke_style_max_distance_from_path.  Only when those approximate
     * extents reach beyond the target does the polygon get limits (the box just made),
     * and the stroker bounds (cairo-spans-compositor.c:_cairo_spans_compositor_stroke). */
    double style_expansion = 0.5, dx, dy;
    if (line_join == CAIRO_LINE_JOIN_MITER && !pf.stroke_is_rectilinear &&
        style_expansion < M_SQRT2 * miter_limit)
      style_expansion = M_SQRT2 * miter_limit;
    style_expansion *= line_width;
    if (fill)                   /* _cairo_path_fixed_approximate_fill_extents: the path's own */
      style_expansion = 0.0;
    if (matrix_has_unity_scale (ctm)) {
      dx = dy = style_expansion;
    } else {
      dx = style_expansion * hypot (ctm->xx, ctm->xy);
      dy = style_expansion * hypot (ctm->yy, ctm->yx);
    }
    const int32_t fdx = fixed_from_double (dx), fdy = fixed_from_double (dy);
    /* _cairo_path_fixed_approximate_stroke_extents + _cairo_box_round_to_rectangle */
    const int64_t ex0 = (int64_t) pf.extents.p1.x - fdx, ey0 = (int64_t) pf.extents.p1.y - fdy;
    const int64_t ex1 = (int64_t) pf.extents.p2.x + fdx, ey1 = (int64_t) pf.extents.p2.y + fdy;
    int mask[4] = { (int) (ex0 >> 8), (int) (ey0 >> 8), (int) ((ex1 + 255) >> 8),
      (int) ((ey1 + 255) >> 8) };
    int ub[4] = { mask[0] > target[0] ? mask[0] : target[0], mask[1] > target[1] ? mask[1] : target[1],
      mask[2] < target[2] ? mask[2] : target[2], mask[3] < target[3] ? mask[3] : target[3] };
    if (ub[2] <= ub[0] || ub[3] <= ub[1])
      goto out;                 /* nothing to do */
    memcpy (polygon->box, ub, sizeof ub);
    st.has_bounds = !(mask[0] >= target[0] && mask[1] >= target[1] && mask[2] <= target[2] &&
        mask[3] <= target[3]);
    polygon->has_limits = st.has_bounds;
    if (st.has_bounds) {
      for (int k = 0; k < 4; k++)
        polygon->limits[k] = ub[k] * 256;
      st.bounds.p1.x = polygon->limits[0] - fdx;
      st.bounds.p2.x = polygon->limits[2] + fdx;
      st.bounds.p1.y = polygon->limits[1] - fdy;
      st.bounds.p2.y = polygon->limits[3] + fdy;
    }
  }

  if (fill) {
    /* _cairo_path_fixed_fill_to_polygon: move_to closes the sub-path before it, a curve
     * that misses the limits is its chord, the end closes the last sub-path */
    const Pt *p = pf.pts;
    Box limit = { { polygon->limits[0], polygon->limits[1] }, { polygon->limits[2], polygon->limits[3] } };
    st.fill_current.x = st.fill_current.y = 0;
    st.fill_last_move = st.fill_current;
    for (int i = 0; i < pf.n_ops; i++) {
      switch (pf.ops[i]) {
        case OP_MOVE_TO:
          filler_line_to (&st, &st.fill_last_move);
          st.fill_current = st.fill_last_move = p[0];
          p += 1;
          break;
        case OP_LINE_TO:
          filler_line_to (&st, &p[0]);
          p += 1;
          break;
        case OP_CURVE_TO: {
          Spline spline;
          if ((polygon->has_limits && !spline_intersects (&st.fill_current, &p[0], &p[1], &p[2], &limit)) ||
              !spline_init (&spline, &st, &st.fill_current, &p[0], &p[1], &p[2])) {
            filler_line_to (&st, &p[2]);
          } else {
            Knots s1 = spline.knots;
            spline.filling = 1;
            spline.last_point = s1.a;
            spline_decompose_into (&s1, tolerance * tolerance, &spline);
            filler_line_to (&st, &spline.knots.d);
          }
          p += 3;
          break;
        }
        case OP_CLOSE_PATH:
          filler_line_to (&st, &st.fill_last_move);
          break;
      }
    }
    filler_line_to (&st, &st.fill_last_move);
  } else if (pf.has_curve_to || line_join == CAIRO_LINE_JOIN_ROUND) {
    if (!pen_init (&st, st.half_line_width, tolerance, ctm)) {
      rc = FLUTL_STROKER_ENOMEM;
      goto out;
    }
    /* a pen reduced to one point: nothing to draw */
    if (st.pen_n <= 1)
      goto out;
  }

  if (!fill) {
    const Pt *p = pf.pts;
    for (int i = 0; i < pf.n_ops; i++) {
      switch (pf.ops[i]) {
        case OP_MOVE_TO:
          stroker_move_to (&st, &p[0]);
          p += 1;
          break;
        case OP_LINE_TO:
          stroker_line_to (&st, &p[0]);
          p += 1;
          break;
        case OP_CURVE_TO:
          stroker_curve_to (&st, &p[0], &p[1], &p[2]);
          p += 3;
          break;
        case OP_CLOSE_PATH:
          stroker_close_path (&st);
          break;
      }
    }
    add_caps (&st);
  }
  if (st.failed)
    rc = FLUTL_STROKER_ENOMEM;
  /* the box the polygon is converted in: its own extents within what is composited
   * (trim_extents_to_polygon) */
  {
    int pb[4];
    if (rc == FLUTL_STROKER_OK && flutl_polygon_box (polygon, pb)) {
      if (pb[0] > polygon->box[0]) polygon->box[0] = pb[0];
      if (pb[1] > polygon->box[1]) polygon->box[1] = pb[1];
      if (pb[2] < polygon->box[2]) polygon->box[2] = pb[2];
      if (pb[3] < polygon->box[3]) polygon->box[3] = pb[3];
      if (polygon->box[2] <= polygon->box[0] || polygon->box[3] <= polygon->box[1])
        polygon->box[0] = polygon->box[1] = polygon->box[2] = polygon->box[3] = 0;
    } else {
      polygon->box[0] = polygon->box[1] = polygon->box[2] = polygon->box[3] = 0;
    }
  }
  goto done;

out:
  polygon->box[0] = polygon->box[1] = polygon->box[2] = polygon->box[3] = 0;
done:
  free (st.cw.pts);
  free (st.ccw.pts);
  free (st.pen);
  free (pf.ops);
  free (pf.pts);
  return rc;
}

int
flutl_polygon_box (const FlutlPolygon *pg, int box[4])
{
  if (pg->n_edges == 0 || pg->extents[2] < pg->extents[0])
    return 0;
  /* _cairo_box_round_to_rectangle */
  box[0] = pg->extents[0] >> 8;
  box[1] = pg->extents[1] >> 8;
  box[2] = (int) (((int64_t) pg->extents[2] + 255) >> 8);
  box[3] = (int) (((int64_t) pg->extents[3] + 255) >> 8);
  return 1;
}
