/*
 * tor_check.c -- TEST INFRASTRUCTURE, builder container only (needs the cairo dev
 * files under /opt/conda).  Differential check of oracle/flutl_tor_oracle.c against
 * the REAL libcairo 1.16.0: random polygons are filled through cairo_fill on an A8
 * surface (the call shape that reaches Cairo's tor scan converter with the
 * polygon's own segments as edges) and through the restatement, byte for byte.
 *     tor_check [n_polygons] [seed]
 */
#include <cairo.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int flutl_tor_polygon_box (const int32_t *edges, int n, int box[4]);
int flutl_tor_oracle (const int32_t *edges, int n, int xmin, int ymin, int xmax, int ymax,
    int even_odd, uint8_t *mask, int stride);

static uint64_t rng_state;
static uint64_t
rnd (void)
{
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

#define MAXP 4096

int
main (int argc, char **argv)
{
  const int n_poly = argc > 1 ? atoi (argv[1]) : 2000;
  rng_state = argc > 2 ? strtoull (argv[2], NULL, 0) : 1;
  const int W = 160, H = 120;
  cairo_surface_t *s = cairo_image_surface_create (CAIRO_FORMAT_A8, W, H);
  const int stride = cairo_image_surface_get_stride (s);
  uint8_t *mine = malloc ((size_t) stride * H);
  static int32_t px[MAXP], py[MAXP], edges[MAXP * 7];
  int bad = 0;
  for (int it = 0; it < n_poly; it++) {
    const int kind = (int) (rnd () % 8);
    const int even_odd = (int) (rnd () % 4 == 0);
    int n_sub = 1 + (int) (rnd () % 3), n_e = 0;
    cairo_t *cr = cairo_create (s);
    cairo_set_operator (cr, CAIRO_OPERATOR_CLEAR);
    cairo_paint (cr);
    cairo_set_operator (cr, CAIRO_OPERATOR_OVER);
    cairo_set_source_rgba (cr, 1, 1, 1, 1);
    cairo_set_fill_rule (cr, even_odd ? CAIRO_FILL_RULE_EVEN_ODD : CAIRO_FILL_RULE_WINDING);
    int degenerate = 0;
    for (int sp = 0; sp < n_sub && !degenerate; sp++) {
      int n = 3 + (int) (rnd () % (kind == 7 ? 200 : 12));
      /* coordinates on the 1/256 grid, strictly inside the surface */
      const int snap = kind == 1 ? 256 : kind == 2 ? 128 : kind == 3 ? 17 : 1;
      for (int i = 0; i < n; i++) {
        int x = 4 * 256 + (int) (rnd () % ((W - 8) * 256));
        int y = 4 * 256 + (int) (rnd () % ((H - 8) * 256));
        if (kind == 4 && i)          /* nearly vertical runs */
          x = px[i - 1] + (int) (rnd () % 64) - 32;
        if (kind == 5 && i)          /* nearly horizontal runs */
          y = py[i - 1] + (int) (rnd () % 64) - 32;
        if (kind == 6 && i) {        /* short edges: a wobbly blob */
          x = px[i - 1] + (int) (rnd () % 1024) - 512;
          y = py[i - 1] + (int) (rnd () % 1024) - 512;
        }
        if (kind == 7) {             /* many small edges around a circle-ish path */
          x = W * 128 + (int) ((W * 100.0) * __builtin_cos (i * 6.283185 / n)) + (int) (rnd () % 300);
          y = H * 128 + (int) ((H * 100.0) * __builtin_sin (i * 6.283185 / n)) + (int) (rnd () % 300);
        }
        x -= x % snap;
        y -= y % snap;
        if (x < 1024) x = 1024;
        if (y < 1024) y = 1024;
        if (x > (W - 4) * 256) x = (W - 4) * 256;
        if (y > (H - 4) * 256) y = (H - 4) * 256;
        px[i] = x;
        py[i] = y;
      }
      /* Cairo's path merges a line_to that continues the previous one and drops
       * one that does not move: keep to polygons without either */
      for (int i = 0; i < n; i++) {
        const int a = i, b = (i + 1) % n, c = (i + 2) % n;
        if (px[a] == px[b] && py[a] == py[b])
          degenerate = 1;
        const int64_t cr_ = (int64_t) (px[b] - px[a]) * (py[c] - py[b]) -
            (int64_t) (py[b] - py[a]) * (px[c] - px[b]);
        if (cr_ == 0)
          degenerate = 1;
      }
      if (degenerate)
        break;
      cairo_move_to (cr, px[0] / 256.0, py[0] / 256.0);
      for (int i = 1; i < n; i++)
        cairo_line_to (cr, px[i] / 256.0, py[i] / 256.0);
      cairo_close_path (cr);
      for (int i = 0; i < n; i++) {
        int32_t x1 = px[i], y1 = py[i], x2 = px[(i + 1) % n], y2 = py[(i + 1) % n], dir = 1;
        if (y1 == y2)
          continue;
        if (y1 > y2) {
          int32_t t;
          t = x1; x1 = x2; x2 = t;
          t = y1; y1 = y2; y2 = t;
          dir = -1;
        }
        int32_t *e = edges + n_e++ * 7;
        e[0] = x1; e[1] = y1; e[2] = x2; e[3] = y2; e[4] = y1; e[5] = y2; e[6] = dir;
      }
    }
    if (degenerate) {
      cairo_destroy (cr);
      it--;
      continue;
    }
    cairo_fill (cr);
    cairo_destroy (cr);
    cairo_surface_flush (s);
    const uint8_t *ref = cairo_image_surface_get_data (s);
    int box[4];
    memset (mine, 0, (size_t) stride * H);
    if (flutl_tor_polygon_box (edges, n_e, box))
      flutl_tor_oracle (edges, n_e, box[0], box[1], box[2], box[3], even_odd,
          mine + (size_t) box[1] * stride + box[0], stride);
    int diff = 0, fx = -1, fy = -1;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        if (ref[y * stride + x] != mine[y * stride + x]) {
          if (!diff) {
            fx = x;
            fy = y;
          }
          diff++;
        }
    if (diff) {
      bad++;
      if (bad <= 10)
        printf ("polygon %d (kind %d, %s, %d edges): %d px differ, first (%d,%d) cairo %d mine %d\n",
            it, kind, even_odd ? "even-odd" : "winding", n_e, diff, fx, fy,
            ref[fy * stride + fx], mine[fy * stride + fx]);
    }
  }
  printf ("%d polygons, %d differ\n", n_poly, bad);
  return bad != 0;
}
