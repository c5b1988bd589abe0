/*
 * flutl_tor_oracle.c -- TEST INFRASTRUCTURE (never linked into the product).
 *
 * CPU restatement of the antialiased polygon scan converter that Cairo's image
 * backend runs under cairo_stroke / cairo_fill -- the step that turns the outline
 * of a tts:textOutline into coverage, and where the reference spends most of its
 * time (call site: gst_ttmlrender_render_outline, plugins/ttml/gstttmlrender.c:
 * 1209-1212, cairo_set_line_width + cairo_stroke).
 *
 * Third-party arithmetic absent from /root/reference: cairo 1.16.0,
 * src/cairo-tor-scan-converter.c ("tor": 15 sample rows x 256 sample columns per
 * pixel, GRID_AREA_TO_ALPHA (c) = (c + (c << 4) + 256) >> 9).  Restated here as a
 * sequential emulation of its published algorithm, list operations and all,
 * because the order of edges with equal sample positions is part of its
 * behaviour (it decides which edge bounds a span on rows converted analytically):
 *   - an edge (24.8 fixed point) is clipped to [ymin, ymax) sample rows; x at the
 *     centre of its first sample row and the advances per sample row / per pixel
 *     row are exact quotient + remainder pairs over dy * 15 * 512;
 *   - `cell` is x rounded to the nearest sample column; the active list is kept
 *     sorted by cell (merge sort of the edges starting on a sample row, insertion
 *     sort after every step);
 *   - a pixel row none of whose edges starts, ends or changes order is converted
 *     analytically (exact trapezoid areas of the span-bounding edges); any other
 *     row by 15 sample rows with non-zero (or even-odd) winding spans;
 *   - per pixel: covered_height (sample rows) and uncovered_area accumulate in
 *     16-bit cells; alpha from the running cover.
 * Pinned against the real libcairo of this image (cairo 1.16.0):
 *   - oracle/tor_check.c: random polygons through cairo_fill on an A8 surface, on grids
 *     that make Cairo's sort meet equal keys all the time, differentially against this
 *     file (43 000 polygons when it was written; 3 000 in the CPU suite,
 *     tests/test_live_reference.py);
 *   - oracle/stroke_check.c: whole strokes, cairo_stroke against host/flutl_stroker.c +
 *     this file (720 000; 3 500 in the CPU suite);
 *   - tests/golden/polygons.npz (oracle/polygon_kats.c: 321 polygons with the real
 *     Cairo's coverage): tests/test_oracle_kats.py::test_tor_restatement_reproduces_
 *     cairos_coverage, and the HIP kernels against the same in tests/test_gpu_scan.py.
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GRID_Y 15
#define ALPHA(c) ((uint8_t) (((c) * 17 + 256) >> 9))     /* (c may be negative: no shift) */

typedef struct Edge
{
  struct Edge *next, *prev;
  int ytop, height_left, dir, cell;
  int32_t xq;
  int64_t xr;
  int32_t dq;
  int64_t dr;
  int32_t fq;
  int64_t fr;
  int64_t dy;
} Edge;

typedef struct
{
  int16_t uncovered_area, covered_height;
  uint8_t used;
} Cell;

typedef struct
{
  int xmin, xmax, ymin, ymax;   /* grid units */
  int xmin_i, xmax_i;           /* pixels */
  Edge **y_buckets;
  Edge *edges;
  Edge head, tail;
  int min_height, is_vertical;
  Cell *cells;                  /* one per pixel column xmin_i - 1 .. xmax_i (dense) */
  int cell_lo;                  /* pixel column of cells[0] */
  int n_cells;
} Tor;

static int
div_trunc_i (int a, int b)
{
  return a / b;
}

static void
normalise (int32_t *q, int64_t *r, int64_t dy)
{
  if (*r < 0) {
    --*q;
    *r += dy;
  } else if (*r >= dy) {
    ++*q;
    *r -= dy;
  }
}

/* cells are found by pixel column; columns outside the dense window are clamped
 * into two catch-alls (left of xmin: only covered_height matters; right of xmax:
 * never read) */
static Cell *
cell_at (Tor *t, int ix)
{
  int k = ix - t->cell_lo;
  if (k < 0)
    k = 0;
  if (k >= t->n_cells)
    k = t->n_cells - 1;
  t->cells[k].used = 1;
  return &t->cells[k];
}

static void
add_subspan (Tor *t, int x1, int x2)
{
  const int fx1 = x1 & 0xff, fx2 = x2 & 0xff;
  const int ix1 = x1 >> 8, ix2 = x2 >> 8;
  if (ix1 != ix2) {
    Cell *a = cell_at (t, ix1), *b = cell_at (t, ix2);
    /* (two cells that fall into the same catch-all cancel exactly as in Cairo,
     * where they are distinct cells left of xmin: only their covered_height is
     * summed there) */
    a->uncovered_area += 2 * fx1;
    a->covered_height += 1;
    b->uncovered_area -= 2 * fx2;
    b->covered_height -= 1;
  } else {
    Cell *a = cell_at (t, ix1);
    a->uncovered_area += 2 * (fx1 - fx2);
  }
}

/* one edge of a span on an analytically converted row: exact areas of the
 * trapezoid between the edge and the left border of every pixel it crosses */
static void
render_edge (Tor *t, Edge *e, int sign)
{
  int32_t x1q = e->xq, x2q;
  int64_t x1r = e->xr, x2r;
  if (e->dy == 0) {
    x2q = x1q;
    x2r = x1r;
  } else {
    e->xq += e->fq;
    e->xr += e->fr;
    normalise (&e->xq, &e->xr, e->dy);
    e->cell = e->xq + (e->dy / 2 <= e->xr);
    /* from the centre of the first sample row to the top of the pixel row */
    const int32_t hq = e->dq / 2;
    const int64_t hr = e->dr / 2;
    x1q -= hq;
    x1r -= hr;
    normalise (&x1q, &x1r, e->dy);
    x2q = e->xq - hq;
    x2r = e->xr - hr;
    normalise (&x2q, &x2r, e->dy);
  }
  int fx1 = x1q & 0xff, fx2 = x2q & 0xff;
  int ix1 = x1q >> 8, ix2 = x2q >> 8;
  if (ix1 == ix2) {
    Cell *c = cell_at (t, ix1);
    c->covered_height += (int16_t) (sign * GRID_Y);
    c->uncovered_area += (int16_t) (sign * GRID_Y * (fx1 + fx2));
    return;
  }
  /* orient left to right: (lo) ... (hi) */
  int32_t loq = x2q, hiq = x1q;
  int64_t lor = x2r, hir = x1r;
  int ixlo = ix2, ixhi = ix1, fxlo = fx2, fxhi = fx1;
  if (ix1 < ix2) {
    loq = x1q; hiq = x2q; lor = x1r; hir = x2r;
    ixlo = ix1; ixhi = ix2; fxlo = fx1; fxhi = fx2;
  }
  const int64_t D = e->dy;
  const int64_t tx = (hir - lor) + (int64_t) (hiq - loq) * D;
  int ix = ixlo + 1;
  const int64_t a = (int64_t) (int32_t) ((uint32_t) ix << 8) * D - ((int64_t) loq * D + lor);
  int64_t yq = (a * GRID_Y) / tx, yr = (a * GRID_Y) % tx;
  {
    Cell *c = cell_at (t, ixlo);
    const int h = (int) yq * sign;
    c->covered_height += (int16_t) h;
    c->uncovered_area += (int16_t) ((fxlo + 256) * h);
  }
  int y = (int) yq;
  if (ix < ixhi) {
    const int64_t sq = (D * GRID_Y * 256) / tx, sr = (D * GRID_Y * 256) % tx;
    do {
      int y2 = (int) sq + y;
      yr += sr;
      if (yr >= tx) {
        y2++;
        yr -= tx;
      }
      Cell *c = cell_at (t, ix);
      const int h = (y2 - y) * sign;
      c->covered_height += (int16_t) h;
      c->uncovered_area += (int16_t) (h * 256);
      y = y2;
      ix++;
    } while (ix != ixhi);
  }
  {
    Cell *c = cell_at (t, ixhi);
    const int h = sign * (GRID_Y - y);
    c->covered_height += (int16_t) h;
    c->uncovered_area += (int16_t) (fxhi * h);
  }
}

static Edge *
merge_sorted_edges (Edge *head_a, Edge *head_b)
{
  Edge *head, **next, *prev;
  int x;
  prev = head_a->prev;
  next = &head;
  if (head_a->cell <= head_b->cell) {
    head = head_a;
  } else {
    head = head_b;
    head_b->prev = prev;
    goto start_with_b;
  }
  for (;;) {
    x = head_b->cell;
    while (head_a != NULL && head_a->cell <= x) {
      prev = head_a;
      next = &head_a->next;
      head_a = head_a->next;
    }
    head_b->prev = prev;
    *next = head_b;
    if (head_a == NULL)
      return head;
  start_with_b:
    x = head_a->cell;
    while (head_b != NULL && head_b->cell <= x) {
      prev = head_b;
      next = &head_b->next;
      head_b = head_b->next;
    }
    head_a->prev = prev;
    *next = head_a;
    if (head_b == NULL)
      return head;
  }
}

static Edge *
sort_edges (Edge *list, unsigned level, Edge **head_out)
{
  Edge *head_other = list->next, *remaining;
  if (head_other == NULL) {
    *head_out = list;
    return NULL;
  }
  remaining = head_other->next;
  if (list->cell <= head_other->cell) {
    *head_out = list;
    head_other->next = NULL;
  } else {
    *head_out = head_other;
    head_other->prev = list->prev;
    head_other->next = list;
    list->prev = head_other;
    list->next = NULL;
  }
  for (unsigned i = 0; i < level && remaining; i++) {
    remaining = sort_edges (remaining, i, &head_other);
    *head_out = merge_sorted_edges (*head_out, head_other);
  }
  return remaining;
}

static void
merge_bucket (Tor *t, Edge *bucket)
{
  Edge *sorted;
  sort_edges (bucket, UINT_MAX, &sorted);
  t->head.next = merge_sorted_edges (t->head.next, sorted);
}

static void
step_sub (Edge *e)
{
  if (e->dy == 0)
    return;
  e->xq += e->dq;
  e->xr += e->dr;
  normalise (&e->xq, &e->xr, e->dy);
  e->cell = e->xq + (e->dy / 2 <= e->xr);
}

static void
sub_row (Tor *t, unsigned mask)
{
  Edge *e = t->head.next;
  int xstart = INT_MIN, prev_x = INT_MIN, winding = 0;
  while (e != &t->tail) {
    Edge *next = e->next;
    const int xend = e->cell;
    if (--e->height_left) {
      step_sub (e);
      if (e->cell < prev_x) {
        Edge *pos = e->prev;
        pos->next = next;
        next->prev = pos;
        do
          pos = pos->prev;
        while (e->cell < pos->cell);
        pos->next->prev = e;
        e->next = pos->next;
        e->prev = pos;
        pos->next = e;
      } else {
        prev_x = e->cell;
      }
      t->min_height = -1;
    } else {
      e->prev->next = next;
      next->prev = e->prev;
    }
    winding += e->dir;
    if ((winding & mask) == 0) {
      if (next->cell != xend) {
        if (xend != xstart)
          add_subspan (t, xstart, xend);
        xstart = INT_MIN;
      }
    } else if (xstart == INT_MIN) {
      xstart = xend;
    }
    e = next;
  }
}

static void
dec (Tor *t, Edge *e, int h)
{
  e->height_left -= h;
  if (e->height_left == 0) {
    e->prev->next = e->next;
    e->next->prev = e->prev;
    t->min_height = -1;
  }
}

static void
full_row (Tor *t, unsigned mask)
{
  Edge *left = t->head.next;
  while (left != &t->tail) {
    Edge *right;
    int winding;
    dec (t, left, GRID_Y);
    winding = left->dir;
    right = left->next;
    for (;;) {
      dec (t, right, GRID_Y);
      winding += right->dir;
      if ((winding & mask) == 0 && right->next->cell != right->cell)
        break;
      if (right->dy) {
        right->xq += right->fq;
        right->xr += right->fr;
        normalise (&right->xq, &right->xr, right->dy);
        right->cell = right->xq + (right->dy / 2 <= right->xr);
      }
      right = right->next;
    }
    render_edge (t, left, +1);
    render_edge (t, right, -1);
    left = right->next;
  }
}

static int
can_do_full_row (Tor *t)
{
  if (t->min_height <= 0) {
    int min_height = INT_MAX, is_vertical = 1;
    for (Edge *e = t->head.next; e != NULL; e = e->next) {
      if (e->height_left < min_height)
        min_height = e->height_left;
      is_vertical &= e->dy == 0;
    }
    t->is_vertical = is_vertical;
    t->min_height = min_height;
  }
  if (t->min_height < GRID_Y)
    return 0;
  int prev_x = INT_MIN;
  for (Edge *e = t->head.next; e != &t->tail; e = e->next) {
    int cell = e->cell;
    if (e->dy) {
      int32_t q = e->xq + e->fq;
      int64_t r = e->xr + e->fr;
      normalise (&q, &r, e->dy);
      cell = q + (e->dy / 2 <= r);
    }
    if (cell < prev_x)
      return 0;
    prev_x = cell;
  }
  return 1;
}

/* the row's cells -> alpha, into `height` rows of the mask */
static void
blit (Tor *t, uint8_t *mask, int stride, int y, int height, int y0)
{
  /* cells[0] collects everything left of xmin_i: its covered_height is the cover
   * the first pixel starts with */
  int16_t cover = 0;
  int any = 0;
  for (int k = 0; k < t->n_cells; k++)
    any |= t->cells[k].used;
  if (!any)
    return;
  uint8_t *row = mask + (size_t) (y - y0) * stride;
  cover = (int16_t) (t->cells[0].covered_height * 512);
  for (int x = t->xmin_i; x < t->xmax_i; x++) {
    const Cell *c = &t->cells[x - t->cell_lo];
    cover = (int16_t) (cover + c->covered_height * 512);
    const int16_t area = (int16_t) (cover - c->uncovered_area);
    row[x - t->xmin_i] = ALPHA ((int) area);
  }
  for (int r = 1; r < height; r++)
    memcpy (row + (size_t) r * stride, row, (size_t) (t->xmax_i - t->xmin_i));
}

/* The box Cairo converts a polygon in: the polygon's extents -- end points of its
 * edges, cut at top / bottom (cairo-polygon.c:_add_edge) -- rounded out to pixels
 * (trim_extents_to_polygon, cairo-spans-compositor.c).  It matters: on a row that is
 * converted analytically an edge is extended to the row's top and bottom, beyond
 * its own ends, and what that puts outside the box is dropped.  The caller
 * intersects it with surface and clip.  Returns 0 for an empty polygon. */
int
flutl_tor_polygon_box (const int32_t *edges, int n, int box[4])
{
  int64_t x0 = INT32_MAX, y0 = INT32_MAX, x1 = INT32_MIN, y1 = INT32_MIN;
  for (int i = 0; i < n; i++) {
    const int32_t *e = edges + (size_t) i * 7;
    const int64_t dy = (int64_t) e[3] - e[1];
    if (e[4] < y0) y0 = e[4];
    if (e[5] > y1) y1 = e[5];
    for (int k = 0; k < 2; k++) {
      const int32_t py = k ? e[3] : e[1], at = k ? e[5] : e[4];
      int64_t x = k ? e[2] : e[0];
      if (x >= x0 && x <= x1)
        continue;
      if (at != py && dy) {
        /* _cairo_edge_compute_intersection_x_for_y (C's truncating division) */
        x = e[0] + ((int64_t) (at - e[1]) * ((int64_t) e[2] - e[0])) / dy;
      }
      if (x < x0) x0 = x;
      if (x > x1) x1 = x;
    }
  }
  if (n == 0 || x1 < x0)
    return 0;
  box[0] = (int) (x0 >> 8);
  box[1] = (int) (y0 >> 8);
  box[2] = (int) ((x1 + 255) >> 8);
  box[3] = (int) ((y1 + 255) >> 8);
  return 1;
}

/* edges: n x 7 int32 {p1.x, p1.y, p2.x, p2.y, top, bottom, dir} in 24.8 fixed point
 * (a cairo_edge_t).  The converter's box is [xmin, xmax) x [ymin, ymax) pixels;
 * mask (stride bytes per row) receives coverage for that box, row ymin first.
 * even_odd: 0 = non-zero winding.  Returns 0, or -1 when out of memory. */
int
flutl_tor_oracle (const int32_t *edges, int n, int xmin, int ymin, int xmax, int ymax,
    int even_odd, uint8_t *mask, int stride)
{
  Tor t;
  memset (&t, 0, sizeof t);
  if (xmax <= xmin || ymax <= ymin)
    return 0;
  for (int y = 0; y < ymax - ymin; y++)
    memset (mask + (size_t) y * stride, 0, (size_t) (xmax - xmin));
  t.xmin = xmin * 256;
  t.xmax = xmax * 256;
  t.ymin = ymin * GRID_Y;
  t.ymax = ymax * GRID_Y;
  t.xmin_i = xmin;
  t.xmax_i = xmax;
  const int h = ymax - ymin;
  t.y_buckets = calloc ((size_t) h + 1, sizeof *t.y_buckets);
  t.edges = calloc ((size_t) n + 1, sizeof *t.edges);
  t.cell_lo = xmin - 1;
  t.n_cells = xmax - xmin + 2;
  t.cells = calloc ((size_t) t.n_cells, sizeof *t.cells);
  if (!t.y_buckets || !t.edges || !t.cells) {
    free (t.y_buckets);
    free (t.edges);
    free (t.cells);
    return -1;
  }
  t.head.cell = INT_MIN;
  t.head.height_left = INT_MAX;
  t.head.next = &t.tail;
  t.tail.prev = &t.head;
  t.tail.cell = INT_MAX;
  t.tail.height_left = INT_MAX;
  t.min_height = 0;
  t.is_vertical = 1;

  for (int i = 0; i < n; i++) {
    const int32_t *in = edges + (size_t) i * 7;
    int ytop = (int) (((int64_t) in[4] * GRID_Y + 128) >> 8);
    int ybot = (int) (((int64_t) in[5] * GRID_Y + 128) >> 8);
    if (ytop < t.ymin)
      ytop = t.ymin;
    if (ybot > t.ymax)
      ybot = t.ymax;
    if (ybot <= ytop)
      continue;
    Edge *e = &t.edges[i];
    e->ytop = ytop;
    e->height_left = ybot - ytop;
    int32_t p1x = in[0], p1y = in[1], p2x = in[2], p2y = in[3];
    if (p2y > p1y) {
      e->dir = in[6];
    } else {
      int32_t s;
      s = p1x; p1x = p2x; p2x = s;
      s = p1y; p1y = p2y; p2y = s;
      e->dir = -in[6];
    }
    if (p2x == p1x) {
      e->cell = p2x;
      e->xq = p1x;
      e->xr = 0;
      e->dq = 0; e->dr = 0; e->fq = 0; e->fr = 0;
      e->dy = 0;
    } else {
      const int64_t dx = (int64_t) p2x - p1x;
      const int64_t Dy = ((int64_t) p2y - p1y) * GRID_Y * 512;
      /* (multiplications, not shifts: dx may be negative) */
      const int64_t Ex = dx * 256;
      e->dq = (int32_t) ((dx * 131072) / Dy);
      e->dr = (dx * 131072) % Dy;
      const int64_t num = ((int64_t) (2 * ytop + 1) * 256 - 2 * (int64_t) GRID_Y * p1y) * Ex;
      e->xq = (int32_t) (num / Dy) + p1x;
      e->xr = num % Dy;
      normalise (&e->xq, &e->xr, Dy);
      if (e->height_left >= GRID_Y) {
        e->fq = (int32_t) ((Ex * GRID_Y * 512) / Dy);
        e->fr = (Ex * GRID_Y * 512) % Dy;
      } else {
        e->fq = 0;
        e->fr = 0;
      }
      e->dy = Dy;
      e->cell = e->xq + (Dy / 2 <= e->xr);
    }
    Edge **b = &t.y_buckets[div_trunc_i (ytop - t.ymin, GRID_Y)];
    e->next = *b;
    *b = e;
  }

  const unsigned mask_w = even_odd ? 1u : ~0u;
  Edge *buckets[GRID_Y];
  memset (buckets, 0, sizeof buckets);
  int j;
  for (int i = 0; i < h; i = j) {
    int do_full_row = 0;
    j = i + 1;
    if (t.y_buckets[i]) {
      /* the edges that start in this pixel row, by sample row */
      int max_suby = 0;
      const int row_top = (i + ymin) * GRID_Y;
      for (Edge *e = t.y_buckets[i], *nx; e; e = nx) {
        nx = e->next;
        const int suby = e->ytop - row_top;
        if (buckets[suby])
          buckets[suby]->prev = e;
        e->next = buckets[suby];
        e->prev = NULL;
        buckets[suby] = e;
        if (e->height_left < t.min_height)
          t.min_height = e->height_left;
        t.is_vertical &= e->dy == 0;
        if (suby > max_suby)
          max_suby = suby;
      }
      if (max_suby == 0) {
        if (buckets[0]) {
          merge_bucket (&t, buckets[0]);
          buckets[0] = NULL;
        }
        do_full_row = can_do_full_row (&t);
      }
    } else {
      if (t.head.next == &t.tail) {
        t.min_height = INT_MAX;
        t.is_vertical = 1;
        for (; j < h && !t.y_buckets[j]; j++)
          ;
        continue;
      }
      do_full_row = can_do_full_row (&t);
    }
    for (int k = 0; k < t.n_cells; k++) {
      t.cells[k].uncovered_area = 0;
      t.cells[k].covered_height = 0;
      t.cells[k].used = 0;
    }
    if (do_full_row) {
      full_row (&t, mask_w);
      if (t.is_vertical) {
        while (j < h && t.y_buckets[j] == NULL && t.min_height >= 2 * GRID_Y) {
          t.min_height -= GRID_Y;
          j++;
        }
        if (j != i + 1) {
          const int count = (j - (i + 1)) * GRID_Y;
          for (Edge *e = t.head.next, *nx; e != &t.tail; e = nx) {
            nx = e->next;
            dec (&t, e, count);
          }
        }
      }
    } else {
      for (int sub = 0; sub < GRID_Y; sub++) {
        if (buckets[sub]) {
          merge_bucket (&t, buckets[sub]);
          buckets[sub] = NULL;
        }
        sub_row (&t, mask_w);
      }
    }
    blit (&t, mask, stride, i + ymin, j - i, ymin);
    t.min_height -= GRID_Y;
  }
  free (t.y_buckets);
  free (t.edges);
  free (t.cells);
  return 0;
}
