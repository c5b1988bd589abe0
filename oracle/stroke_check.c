/*
 * stroke_check.c -- TEST INFRASTRUCTURE, builder container only (needs the cairo dev
 * files under /opt/conda).  Differential check of the product's host-side stroker
 * (flu-plugins-oss_amd/host/flutl_stroker.c: path -> polygon) against the REAL
 * libcairo 1.16.0: random paths are stroked through cairo_stroke on an A8 surface --
 * the call the reference makes for tts:textOutline (gstttmlrender.c:1209-1212) -- and
 * through stroker + oracle/flutl_tor_oracle.c, byte for byte.
 *     stroke_check [n_paths] [seed] [mode]
 * mode: 0 random paths inside the surface, 1 crossing a clip box, 2 lines of text
 * (DejaVu glyph outlines, what the element strokes), 3 text crossing a clip box;
 * + 4: the same paths FILLED (cairo_fill against flutl_fill_to_polygon: what the reference
 * does with anamorphic glyphs, gstttmlrender.c:1109-1118, and Pango with its line-break arrow)
 */
#include <cairo.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../flu-plugins-oss_amd/host/flutl_stroker.h"

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

static double
frand (double lo, double hi)
{
  return lo + (hi - lo) * (double) (rnd () >> 11) / 9007199254740992.0;
}

int
main (int argc, char **argv)
{
  const int n_paths = argc > 1 ? atoi (argv[1]) : 2000;
  rng_state = argc > 2 ? strtoull (argv[2], NULL, 0) : 1;
  const int fill = argc > 3 ? (atoi (argv[3]) & 4) != 0 : 0;
  const int mode = argc > 3 ? atoi (argv[3]) & 3 : 0;
  const int text = mode >= 2;
  const int W = text ? 1000 : 200, H = text ? 320 : 150;
  cairo_surface_t *s = cairo_image_surface_create (CAIRO_FORMAT_A8, W, H);
  const int stride = cairo_image_surface_get_stride (s);
  uint8_t *mine = malloc ((size_t) stride * H);
  int bad = 0, skipped = 0;
  unsigned long long total_edges = 0, total_cov = 0;
  for (int it = 0; it < n_paths; it++) {
    cairo_t *cr = cairo_create (s);
    cairo_set_operator (cr, CAIRO_OPERATOR_CLEAR);
    cairo_paint (cr);
    cairo_set_operator (cr, CAIRO_OPERATOR_OVER);
    cairo_set_source_rgba (cr, 1, 1, 1, 1);
    const int kind = (int) (rnd () % 6);
    const double margin = (mode & 1) ? -30 : 40;
    const double lw = text ? (rnd () & 1 ? 2.0 * (1 + (int) (rnd () % 5)) : frand (0.5, 12.0))
        : kind == 5 ? frand (0.3, 2.0) : frand (1.0, 12.0);
    const int n_sub = 1 + (int) (rnd () % 3);
    if (text) {
      /* one or two lines of text, as the element builds an outline's path: glyph by
       * glyph at integer or fractional positions, optionally scaled horizontally */
      static const char *fonts[] = { "DejaVu Sans", "DejaVu Serif", "DejaVu Sans Mono" };
      static const char *words[] = { "Subtitles", "Quick brown fox", "jumps over", "AVATAR",
        "g@&%$#", "Ill1|!", "OoQ0", "WwMm", "xyz{}[]", "~^`'\"", "Ünïcödé ßø" };
      const double size = frand (12, 150);
      cairo_select_font_face (cr, fonts[rnd () % 3], rnd () % 3 == 0 ? CAIRO_FONT_SLANT_ITALIC
          : CAIRO_FONT_SLANT_NORMAL, rnd () & 1 ? CAIRO_FONT_WEIGHT_BOLD : CAIRO_FONT_WEIGHT_NORMAL);
      cairo_set_font_size (cr, size);
      const int n_lines = 1 + (int) (rnd () % 2);
      for (int ln = 0; ln < n_lines; ln++) {
        double x = frand ((mode & 1) ? -40 : 30, 200), y = frand (size, H - 20);
        if (rnd () & 1) {
          x = floor (x);
          y = floor (y);
        }
        cairo_save (cr);
        cairo_translate (cr, x, y);
        if (rnd () % 4 == 0)
          cairo_scale (cr, frand (0.5, 1.5), 1.0);
        cairo_move_to (cr, 0, 0);
        cairo_text_path (cr, words[rnd () % 11]);
        if (rnd () % 3 == 0) {      /* a decoration rectangle in the same path */
          cairo_rectangle (cr, 0, floor (size * 0.1), floor (size * 2), floor (size * 0.07) + 1);
        }
        cairo_restore (cr);
      }
    } else
    /* the path, in device coordinates on the 1/256 grid */
    for (int sp = 0; sp < n_sub; sp++) {
      const int n = 2 + (int) (rnd () % 7);
      double x = frand (margin, W - margin), y = frand (margin, H - margin);
      cairo_move_to (cr, floor (x * 256) / 256, floor (y * 256) / 256);
      for (int i = 0; i < n; i++) {
        const double r = kind == 4 ? 6 : 60;
        double px[3], py[3];
        for (int k = 0; k < 3; k++) {
          px[k] = x + frand (-r, r);
          py[k] = y + frand (-r, r);
          if (px[k] < margin) px[k] = margin;
          if (px[k] > W - margin) px[k] = W - margin;
          if (py[k] < margin) py[k] = margin;
          if (py[k] > H - margin) py[k] = H - margin;
          px[k] = floor (px[k] * 256) / 256;
          py[k] = floor (py[k] * 256) / 256;
        }
        if (kind == 0 || (kind >= 2 && (rnd () & 1)))
          cairo_line_to (cr, px[2], py[2]);
        else
          cairo_curve_to (cr, px[0], py[0], px[1], py[1], px[2], py[2]);
        x = px[2];
        y = py[2];
      }
      if (kind != 3 || (rnd () & 1))
        cairo_close_path (cr);
    }
    cairo_matrix_t ctm;
    switch (rnd () % 5) {
      case 0: cairo_matrix_init_identity (&ctm); break;
      case 1: cairo_matrix_init_translate (&ctm, frand (-50, 50), frand (-50, 50)); break;
      case 2: { const double k = frand (0.5, 2); cairo_matrix_init_scale (&ctm, k, k); break; }
      case 3: cairo_matrix_init_rotate (&ctm, frand (0, 6.28)); break;
      default:
        cairo_matrix_init (&ctm, 6.123233995736766e-17, 1, -1, 6.123233995736766e-17,
            frand (0, 100), frand (0, 100));
        break;
    }
    /* (miter joins only: what the reference uses, and what the stroker takes -- see
     * flutl_stroke_to_polygon) */
    const cairo_line_join_t join = CAIRO_LINE_JOIN_MITER;
    const double miter = rnd () % 4 == 0 ? frand (1, 4) : 10.0;
    cairo_path_t *path = cairo_copy_path (cr);
    int lim[4] = { 0, 0, W, H };
    if (mode & 1) {
      lim[0] = (int) (rnd () % 40);
      lim[1] = (int) (rnd () % 40);
      lim[2] = W - (int) (rnd () % (text ? 500 : 40));
      lim[3] = H - (int) (rnd () % (text ? 150 : 40));
      cairo_new_path (cr);
      cairo_rectangle (cr, lim[0], lim[1], lim[2] - lim[0], lim[3] - lim[1]);
      cairo_clip (cr);
      cairo_append_path (cr, path);
    }
    cairo_set_matrix (cr, &ctm);
    cairo_set_line_width (cr, lw);
    cairo_set_line_join (cr, join);
    cairo_set_miter_limit (cr, miter);
    if (fill)
      cairo_fill (cr);
    else
      cairo_stroke (cr);
    cairo_destroy (cr);
    cairo_surface_flush (s);
    const uint8_t *ref = cairo_image_surface_get_data (s);

    FlutlPolygon pg;
    flutl_polygon_init (&pg);
    int rc = fill ? flutl_fill_to_polygon (path, 0.1, lim, &pg)
                  : flutl_stroke_to_polygon (path, &ctm, lw, miter, join, 0.1, lim, &pg);
    cairo_path_destroy (path);
    if (rc != FLUTL_STROKER_OK) {
      skipped++;
      flutl_polygon_fini (&pg);
      continue;
    }
    const int *box = pg.box;
    memset (mine, 0, (size_t) stride * H);
    if (box[2] > box[0] && box[3] > box[1])
      flutl_tor_oracle (pg.edges, (int) pg.n_edges, box[0], box[1], box[2], box[3], 0,
          mine + (size_t) box[1] * stride + box[0], stride);
    int diff = 0, fx = -1, fy = -1, maxd = 0;
    total_edges += pg.n_edges;
    for (int y = 0; y < H; y++)
      for (int x = 0; x < W; x++)
        if (total_cov += ref[y * stride + x] != 0, ref[y * stride + x] != mine[y * stride + x]) {
          int d = abs ((int) ref[y * stride + x] - (int) mine[y * stride + x]);
          if (d > maxd) maxd = d;
          if (!diff) {
            fx = x;
            fy = y;
          }
          diff++;
        }
    if (diff) {
      bad++;
      if (bad <= 12)
        printf ("path %d (kind %d, lw %.3f, join %d, miter %.2f, ctm %g %g %g %g, %u edges): %d px differ "
            "(max %d), first (%d,%d) cairo %d mine %d\n", it, kind, lw, (int) join, miter, ctm.xx, ctm.yx,
            ctm.xy, ctm.yy, pg.n_edges, diff, maxd, fx, fy, ref[fy * stride + fx], mine[fy * stride + fx]);
    }
    flutl_polygon_fini (&pg);
  }
  printf ("%d paths, %d differ, %d skipped (%llu edges and %llu covered pixels per path)\n", n_paths, bad,
      skipped, total_edges / (n_paths ? n_paths : 1), total_cov / (n_paths ? n_paths : 1));
  return bad != 0;
}
