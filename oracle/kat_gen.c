/*
 * kat_gen.c -- TEST INFRASTRUCTURE.  Known-answer-table generator.
 *
 * Drives the REAL libcairo 1.16.0 / libpixman 0.40.0 of this image through
 * exactly the call shapes the reference uses on its render path
 * (plugins/ttml/gstttmlrender.c, call sites quoted per table) and dumps
 * (input, output) tables.  oracle/make_kats.py packs them into
 * tests/golden/kats.npz; tests then hold oracle/flutl_oracle.c and the HIP
 * kernel to these answers.  Runs only in the builder container (needs the
 * cairo/pixman dev files under /opt/conda); its outputs are committed.
 *
 * Output: one raw little-endian file per table in argv[1].
 */
#include <cairo.h>
#include <pixman.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t
splitmix64 (void)
{
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static uint32_t
rand_premul (void)
{
  uint64_t r = splitmix64 ();
  uint32_t a = r & 0xff;
  if (((r >> 60) & 3) == 0)
    a = 255;
  if (((r >> 60) & 3) == 1)
    a = (r >> 56) & 1 ? 0 : 128;
  uint32_t c0 = ((r >> 8) & 0xff) * a / 255, c1 = ((r >> 16) & 0xff) * a / 255,
           c2 = ((r >> 24) & 0xff) * a / 255;
  return (a << 24) | (c0 << 16) | (c1 << 8) | c2;
}

static FILE *
open_out (const char *dir, const char *name)
{
  char path[1024];
  snprintf (path, sizeof path, "%s/%s", dir, name);
  FILE *f = fopen (path, "wb");
  if (!f) {
    perror (path);
    exit (1);
  }
  return f;
}

static void
w32 (FILE *f, uint32_t v)
{
  fwrite (&v, 4, 1, f);
}

static void
wf64 (FILE *f, double v)
{
  fwrite (&v, 8, 1, f);
}

/* A W x H ARGB32 surface over caller memory */
static cairo_surface_t *
surf (uint32_t *px, int w, int h)
{
  return cairo_image_surface_create_for_data (
      (unsigned char *) px, CAIRO_FORMAT_ARGB32, w, h, w * 4);
}

#define GET_COMP(c, offs) ((((c) >> offs) & 255) / 255.0)

/* Table 1: colour quantisation.  set_source_rgba(GET_CAIRO_COMP...) + aligned
 * rectangle fill over a cleared pixel -- gstttmlrender.c:1259-1267.
 * record: rgba8888 (0xRRGGBBAA as the reference packs it), out pixel */
static void
kat_color (const char *dir)
{
  FILE *f = open_out (dir, "color.u32");
  for (int i = 0; i < 20000; i++) {
    uint32_t c = (uint32_t) splitmix64 ();
    if (i < 256)
      c = 0xffffff00u | (uint32_t) i;   /* white, all alphas */
    else if (i < 512)
      c = ((uint32_t) (i - 256) << 24) | 0x00804080u;
    else if (i < 768)
      c = ((uint32_t) (i - 512) << 8) | 0x120000c0u;
    uint32_t px = 0;
    cairo_surface_t *s = surf (&px, 1, 1);
    cairo_t *cr = cairo_create (s);
    cairo_set_source_rgba (cr, GET_COMP (c, 24), GET_COMP (c, 16),
        GET_COMP (c, 8), GET_COMP (c, 0));
    cairo_rectangle (cr, 0, 0, 1, 1);
    cairo_fill (cr);
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    w32 (f, c);
    w32 (f, px);
  }
  fclose (f);
}

/* Table 2: aligned solid fill OVER an existing pixel (region background over
 * earlier regions, :1259-1267).  record: rgba8888, dst, out */
static void
kat_fill (const char *dir)
{
  FILE *f = open_out (dir, "fill.u32");
  for (int i = 0; i < 70000; i++) {
    uint32_t c = (uint32_t) splitmix64 ();
    uint32_t d = rand_premul ();
    if (i < 65536) {
      /* systematic: alpha x dst channel */
      c = 0x40c0ff00u | (uint32_t) (i >> 8);
      uint32_t v = i & 0xff;
      d = 0xff000000u | (v << 16) | (v << 8) | v;
    }
    uint32_t px = d;
    cairo_surface_t *s = surf (&px, 1, 1);
    cairo_t *cr = cairo_create (s);
    cairo_set_source_rgba (cr, GET_COMP (c, 24), GET_COMP (c, 16),
        GET_COMP (c, 8), GET_COMP (c, 0));
    cairo_rectangle (cr, 0, 0, 1, 1);
    cairo_fill (cr);
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    w32 (f, c);
    w32 (f, d);
    w32 (f, px);
  }
  fclose (f);
}

/* Table 3: solid source through an A8 mask, OVER.  This is the arithmetic of
 * a glyph composite (pango_cairo_show_glyph_string, :1150; pixman
 * over_n_8_8888).  Done 64 px wide so the SIMD path is the one exercised.
 * record: rgba8888, mask, dst, out */
static void
kat_mask (const char *dir)
{
  FILE *f = open_out (dir, "mask.u32");
  enum { N = 64 };
  for (int it = 0; it < 1200; it++) {
    uint32_t c = (uint32_t) splitmix64 ();
    if (it % 3 == 0)
      c |= 0xff;                /* opaque colours too */
    if (it < 256)
      c = 0xffffff00u | (uint32_t) it;
    uint32_t px[N], d0[N];
    uint8_t m[N];
    for (int i = 0; i < N; i++) {
      d0[i] = px[i] = rand_premul ();
      m[i] = (uint8_t) splitmix64 ();
      if (it < 256) {
        m[i] = (uint8_t) (i * 4 + (it & 3));
        d0[i] = px[i] = 0;
      } else if (it < 512) {
        m[i] = (uint8_t) (it - 256);
      }
    }
    cairo_surface_t *s = surf (px, N, 1);
    cairo_surface_t *ms = cairo_image_surface_create_for_data (
        m, CAIRO_FORMAT_A8, N, 1, N);
    cairo_t *cr = cairo_create (s);
    cairo_set_source_rgba (cr, GET_COMP (c, 24), GET_COMP (c, 16),
        GET_COMP (c, 8), GET_COMP (c, 0));
    cairo_mask_surface (cr, ms, 0, 0);
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    cairo_surface_destroy (ms);
    for (int i = 0; i < N; i++) {
      w32 (f, c);
      w32 (f, m[i]);
      w32 (f, d0[i]);
      w32 (f, px[i]);
    }
  }
  fclose (f);
}

/* Table 4: region opacity group: temp ARGB32 surface painted with
 * cairo_paint_with_alpha (:1375-1382).  record: opacity(f64 bits lo,hi),
 * group px, dst, out */
static void
kat_group (const char *dir)
{
  FILE *f = open_out (dir, "group.bin");
  enum { N = 64 };
  for (int it = 0; it < 700; it++) {
    double op = (it < 101) ? it / 100.0
        : (it < 357) ? (it - 101) / 255.0
        : (double) (splitmix64 () >> 11) / 9007199254740992.0;
    uint32_t px[N], d0[N], g[N];
    for (int i = 0; i < N; i++) {
      d0[i] = px[i] = (it % 4 == 0) ? 0 : rand_premul ();
      g[i] = rand_premul ();
      if (it < 101)
        g[i] = 0xff000000u | (uint32_t) (i * 4) * 0x010101u;
    }
    cairo_surface_t *s = surf (px, N, 1);
    cairo_surface_t *gs = surf (g, N, 1);
    cairo_t *cr = cairo_create (s);
    cairo_set_source_surface (cr, gs, 0, 0);
    cairo_paint_with_alpha (cr, op);
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    cairo_surface_destroy (gs);
    for (int i = 0; i < N; i++) {
      wf64 (f, op);
      w32 (f, g[i]);
      w32 (f, d0[i]);
      w32 (f, px[i]);
      w32 (f, 0);
    }
  }
  fclose (f);
}

/* Table 5: un-aligned rectangle fill (span background :1025-1027 with a
 * fractional CTM translation; under/over/strike lines :1053-1092).
 * The rectangle is issued exactly like the reference: translate, then
 * cairo_rectangle(0, -pre, width, pre+post) + fill.
 * record: tx,ty,x,y,w,h (f64) rgba8888 u32, pad u32, then dst 12x12 u32, out
 * 12x12 u32 */
static void
kat_rect (const char *dir)
{
  FILE *f = open_out (dir, "rect.bin");
  enum { W = 12, H = 12 };
  for (int it = 0; it < 1000; it++) {
    double tx, ty, x, y, w, h;
    uint32_t c = (uint32_t) splitmix64 ();
    if (it & 1)
      c |= 0xff;
    if (it < 512) {
      /* sweep every 1/256 offset on x (even) or y (odd) */
      int k = it >> 1;
      tx = 2 + ((it & 1) ? 0.0 : k / 256.0);
      ty = 3 + ((it & 1) ? k / 256.0 : 0.0);
      x = 0; y = 0; w = 5; h = 4;
    } else if (it < 768) {
      int k = it - 512;
      tx = 1.5; ty = 2.25; x = 0; y = -1.0;
      w = 3 + k / 256.0; h = 2 + (k * 7 % 256) / 256.0;
    } else {
      tx = (double) (splitmix64 () % 4096) / 1024.0;
      ty = (double) (splitmix64 () % 4096) / 1024.0 + 2;
      x = 0;
      y = -(double) (splitmix64 () % 2048) / 1024.0;
      w = (double) (splitmix64 () % 6000) / 1000.0;
      h = (double) (splitmix64 () % 6000) / 1000.0;
      if (it % 7 == 0) { tx = floor (tx); ty = floor (ty); y = floor (y);
        w = floor (w) + 1; h = floor (h) + 1; }
      if (it % 11 == 0) { tx = 1.0 / 3.0; ty = 2 + 2.0 / 3.0; w = 4.7; }
    }
    uint32_t px[W * H], d0[W * H];
    for (int i = 0; i < W * H; i++)
      d0[i] = px[i] = (it % 3 == 0) ? 0 : rand_premul ();
    cairo_surface_t *s = surf (px, W, H);
    cairo_t *cr = cairo_create (s);
    cairo_translate (cr, tx, ty);
    cairo_set_source_rgba (cr, GET_COMP (c, 24), GET_COMP (c, 16),
        GET_COMP (c, 8), GET_COMP (c, 0));
    cairo_rectangle (cr, x, y, w, h);
    cairo_fill (cr);
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    wf64 (f, tx); wf64 (f, ty); wf64 (f, x); wf64 (f, y); wf64 (f, w);
    wf64 (f, h);
    w32 (f, c); w32 (f, 0);
    fwrite (d0, 4, W * H, f);
    fwrite (px, 4, W * H, f);
  }
  fclose (f);
}

/* Table 6: stroked path (text outline, :1206-1212): the same path is stroked
 * (a) with solid white onto a cleared A8 surface -> the coverage mask the
 * product's host side extracts, and (b) with the outline colour onto ARGB32
 * over an existing background -> what the reference produces.  Pins both
 * "A8 render == coverage" and the opaque-vs-translucent blend rule.
 * record: rgba8888, line_width(f64), then mask 48x32 u8, dst 48x32 u32, out */
static void
kat_stroke (const char *dir)
{
  FILE *f = open_out (dir, "stroke.bin");
  enum { W = 48, H = 32 };
  for (int it = 0; it < 60; it++) {
    uint32_t c = (uint32_t) splitmix64 ();
    if (it % 2 == 0)
      c |= 0xff;
    if (it == 0)
      c = 0x000000ffu;
    double lw = 1.0 + (it % 7);
    uint8_t m[W * H];
    uint32_t px[W * H], d0[W * H];
    memset (m, 0, sizeof m);
    for (int i = 0; i < W * H; i++)
      d0[i] = px[i] = (it % 3 == 0) ? 0 : rand_premul ();
    double cx = 10 + (it % 5) * 0.37, cy = 8 + (it % 3) * 0.5;
    for (int pass = 0; pass < 2; pass++) {
      cairo_surface_t *s = pass == 0
          ? cairo_image_surface_create_for_data (m, CAIRO_FORMAT_A8, W, H, W)
          : surf (px, W, H);
      cairo_t *cr = cairo_create (s);
      if (pass == 0)
        cairo_set_source_rgba (cr, 1, 1, 1, 1);
      else
        cairo_set_source_rgba (cr, GET_COMP (c, 24), GET_COMP (c, 16),
            GET_COMP (c, 8), GET_COMP (c, 0));
      cairo_set_line_width (cr, lw);
      cairo_translate (cr, cx, cy);
      cairo_move_to (cr, 0, 0);
      cairo_curve_to (cr, 10, -6, 20, 18, 28, 4);
      cairo_line_to (cr, 30, 16);
      cairo_close_path (cr);
      cairo_move_to (cr, 3, 12);
      cairo_line_to (cr, 25.5, 12);
      cairo_stroke (cr);
      cairo_destroy (cr);
      cairo_surface_finish (s);
      cairo_surface_destroy (s);
    }
    w32 (f, c);
    w32 (f, 0);
    wf64 (f, lw);
    fwrite (m, 1, W * H, f);
    fwrite (d0, 4, W * H, f);
    fwrite (px, 4, W * H, f);
  }
  fclose (f);
}

/* Table 7: premultiplied ARGB32 image OVER (smpte:backgroundImage paint,
 * :1271-1276, and the blurred outline blit :1219-1224).
 * record: src, dst, out */
static void
kat_image (const char *dir)
{
  FILE *f = open_out (dir, "image.u32");
  enum { N = 64 };
  for (int it = 0; it < 600; it++) {
    uint32_t px[N], d0[N], g[N];
    for (int i = 0; i < N; i++) {
      d0[i] = px[i] = (it % 4 == 0) ? 0 : rand_premul ();
      g[i] = rand_premul ();
    }
    cairo_surface_t *s = surf (px, N, 1);
    cairo_surface_t *gs = surf (g, N, 1);
    cairo_t *cr = cairo_create (s);
    cairo_set_source_surface (cr, gs, 0, 0);
    if (it & 1) {
      cairo_paint (cr);
    } else {
      cairo_rectangle (cr, 0, 0, N, 1);
      cairo_fill (cr);
    }
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    cairo_surface_destroy (gs);
    for (int i = 0; i < N; i++) {
      w32 (f, g[i]);
      w32 (f, d0[i]);
      w32 (f, px[i]);
    }
  }
  fclose (f);
}

/* Table 8: pixman convolution blur, issued exactly like
 * gst_ttml_blur_image_surface (plugins/ttml/gstttmlblur.c:71-110) with a
 * kernel built like gst_ttml_blur_create_gaussian_kernel (:27-67).
 * record: radius i32, pad, sigma f64, kernel (2r+1)^2 i32, src 40x24 u32,
 * out 40x24 u32 */
static void
kat_blur (const char *dir)
{
  FILE *f = open_out (dir, "blur.bin");
  enum { W = 40, H = 24 };
  const double blurs[] = { 0.5, 1.0, 1.5, 2.0, 3.0, 4.0 };
  for (int it = 0; it < 12; it++) {
    double blur = blurs[it % 6];
    int radius = (int) (blur * 2);      /* int radius arg of :1216-1217 */
    double sigma = blur;
    const int size = 2 * radius + 1, n = size * size;
    double *tmp = malloc (sizeof (double) * n), sum = 0;
    const double scale2 = 2.0 * sigma * sigma, scale1 = 1.0 / (M_PI * scale2);
    int i = 0;
    for (int x = -radius; x <= radius; ++x)
      for (int y = -radius; y <= radius; ++y, ++i) {
        tmp[i] = scale1 * exp (-((double) (x * x) + (double) (y * y)) / scale2);
        sum += tmp[i];
      }
    pixman_fixed_t *params = malloc (sizeof (pixman_fixed_t) * (n + 2));
    params[0] = pixman_int_to_fixed (size);
    params[1] = pixman_int_to_fixed (size);
    for (i = 0; i < n; ++i)
      params[2 + i] = pixman_double_to_fixed (tmp[i] / sum);
    uint32_t src[W * H], dst[W * H];
    memset (dst, 0, sizeof dst);
    for (i = 0; i < W * H; i++) {
      int x = i % W, y = i / W;
      if (it < 6)
        src[i] = (x > 8 && x < 30 && y > 5 && y < 18 && ((x / 3 + y / 2) & 1))
            ? 0xff102030u : 0;
      else
        src[i] = rand_premul ();
    }
    pixman_image_t *ps = pixman_image_create_bits (PIXMAN_a8r8g8b8, W, H, src, W * 4);
    pixman_image_set_filter (ps, PIXMAN_FILTER_CONVOLUTION, params, n + 2);
    pixman_image_t *pd = pixman_image_create_bits (PIXMAN_a8r8g8b8, W, H, dst, W * 4);
    pixman_image_composite (PIXMAN_OP_SRC, ps, NULL, pd, 0, 0, 0, 0, 0, 0, W, H);
    pixman_image_unref (ps);
    pixman_image_unref (pd);
    w32 (f, (uint32_t) radius);
    w32 (f, 0);
    wf64 (f, sigma);
    fwrite (params + 2, 4, n, f);
    fwrite (src, 4, W * H, f);
    fwrite (dst, 4, W * H, f);
    free (params);
    free (tmp);
  }
  fclose (f);
}

/* Table 9: an ARGB32 surface painted at a FRACTIONAL device position, issued
 * like the blurred-outline blit of gstttmlrender.c:1219-1224 (set_source_surface
 * at integer (px, py) under a CTM translated by doubles, then cairo_paint):
 * cairo picks the bilinear filter, pixman resamples with 7-bit weights.
 * Odd records paint; even records fill the rectangle of the surface's extent,
 * as the reference does: the box edges then add an antialiased coverage.
 * record: tx, ty f64; px, py i32; src SW x SH u32; dst0, out DW x DH u32 */
static void
kat_fracblit (const char *dir)
{
  FILE *f = open_out (dir, "fracblit.bin");
  enum { SW = 10, SH = 7, DW = 24, DH = 16 };
  const double fr[] = { 0.5, 0.25, 0.75, 0.125, 1.0 / 3.0, 0.99, 0.004, 0.0 };
  for (int it = 0; it < 160; it++) {
    uint32_t src[SW * SH], d0[DW * DH], px[DW * DH];
    for (int i = 0; i < SW * SH; i++)
      src[i] = (it % 5 == 4) ? 0xff804020u : rand_premul ();
    for (int i = 0; i < DW * DH; i++)
      d0[i] = px[i] = (it % 3 == 0) ? 0 : rand_premul ();
    uint64_t r = splitmix64 ();
    double tx, ty;
    if (it < 64) {
      tx = 3 + fr[it % 8];
      ty = 2 + fr[(it / 8) % 8];
    } else {
      tx = ((r & 0xfffff) / 1048576.0) * 9.0 - 1.5;
      ty = (((r >> 20) & 0xfffff) / 1048576.0) * 6.0 - 1.5;
      if (it % 7 == 0)
        ty = (double) (int) ty;       /* fractional in x only */
      if (it >= 128) {
        /* just below / above a 7-bit weight boundary: pins how cairo rounds
         * the translation to 16.16 */
        int kx = (int) ((r >> 48) & 127), ky = (int) ((r >> 55) & 127);
        tx = 3 + (kx * 512 + ((it & 1) ? 0.4 : -0.4)) / 65536.0;
        ty = 2 + (ky * 512 + ((it & 2) ? 0.6 : -0.6)) / 65536.0;
      }
    }
    int ox = (int) ((r >> 40) & 3) - 1, oy = (int) ((r >> 44) & 3) - 1;
    cairo_surface_t *s = surf (px, DW, DH);
    cairo_surface_t *is = surf (src, SW, SH);
    cairo_t *cr = cairo_create (s);
    cairo_translate (cr, tx, ty);
    cairo_set_source_surface (cr, is, ox, oy);
    if (it & 1) {
      cairo_paint (cr);
    } else {                        /* exactly as :1219-1224: a fill of the extent */
      cairo_rectangle (cr, ox, oy, SW, SH);
      cairo_fill (cr);
    }
    cairo_destroy (cr);
    cairo_surface_finish (s);
    cairo_surface_destroy (s);
    cairo_surface_destroy (is);
    wf64 (f, tx);
    wf64 (f, ty);
    w32 (f, (uint32_t) ox);
    w32 (f, (uint32_t) oy);
    fwrite (src, 4, SW * SH, f);
    fwrite (d0, 4, DW * DH, f);
    fwrite (px, 4, DW * DH, f);
  }
  fclose (f);
}

int
main (int argc, char **argv)
{
  if (argc < 2) {
    fprintf (stderr, "usage: kat_gen OUTDIR\n");
    return 2;
  }
  printf ("cairo %s pixman %s\n", cairo_version_string (),
      pixman_version_string ());
  kat_color (argv[1]);
  kat_fill (argv[1]);
  kat_mask (argv[1]);
  kat_group (argv[1]);
  kat_rect (argv[1]);
  kat_stroke (argv[1]);
  kat_image (argv[1]);
  kat_blur (argv[1]);
  kat_fracblit (argv[1]);
  return 0;
}
