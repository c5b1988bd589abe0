/*
 * flutl_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the pixel arithmetic that the reference's
 * `ttmlrender` element reaches through Cairo 1.16.0 / pixman 0.40.0 on its
 * per-buffer path (gst_ttmlrender_gen_buffer, plugins/ttml/gstttmlrender.c
 * :1427-1478), expressed over the display-list IR of include/flutl.h.
 *
 * The arithmetic lives in third-party libraries whose sources are NOT under
 * /root/reference (pangocairo >= 1.16.0, pixman-1 >= 0.18.0 --
 * plugins/ttml/meson.build:30-31; the oracle stack in this image is cairo
 * 1.16.0 / pixman 0.40.0 / pango 1.45.3).  It is restated here from the
 * published algorithms and PINNED two ways (tests/test_oracle_*.py):
 *   1. known-answer tables produced by the real libcairo/libpixman of this
 *      image (oracle/kat_gen.c -> tests/golden/kat_*.bin);
 *   2. whole frames rendered by the reference plugin itself, built from
 *      /root/reference by oracle/build_ref.sh (tests/golden/scenes/...).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * call into this file.  The product (flu-plugins-oss_amd/) never does.
 *
 * Structure is deliberately different from the HIP kernel: this walks the
 * primitives one at a time over whole surfaces (as Cairo does), with a real
 * temporary group surface, whereas the kernel walks tiles.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "../include/flutl.h"

/* ---- pixman 8-bit arithmetic (pixman-combine32.h: MUL_UN8, UN8x4_*) ------ */

static inline uint32_t mul_un8 (uint32_t a, uint32_t b)
{
  uint32_t t = a * b + 0x80;
  return (t + (t >> 8)) >> 8;
}

static inline uint32_t add_sat8 (uint32_t a, uint32_t b)
{
  uint32_t t = a + b;
  return t > 255 ? 255 : t;
}

/* x_c = x_c * a / 255 on all four channels */
static inline uint32_t un8x4_mul_un8 (uint32_t x, uint32_t a)
{
  return mul_un8 (x & 0xff, a) | (mul_un8 ((x >> 8) & 0xff, a) << 8) |
      (mul_un8 ((x >> 16) & 0xff, a) << 16) | (mul_un8 (x >> 24, a) << 24);
}

/* pixman OVER: d = s + d * (255 - alpha(s)) / 255, saturating */
static inline uint32_t pixman_over (uint32_t s, uint32_t d)
{
  uint32_t ia = 255 - (s >> 24);
  uint32_t r = 0;
  for (int sh = 0; sh < 32; sh += 8)
    r |= add_sat8 ((s >> sh) & 0xff, mul_un8 ((d >> sh) & 0xff, ia)) << sh;
  return r;
}

/* (solid s IN mask m) OVER d  -- pixman over_n_8_8888 and the glyph path */
static inline uint32_t pixman_over_mask (uint32_t s, uint32_t m, uint32_t d)
{
  if (m == 0)
    return d;
  return pixman_over (un8x4_mul_un8 (s, m), d);
}

/* ---- cairo in-place lerp for opaque solid sources -------------------------
 * cairo-image-compositor.c: mul8_8 / lerp8x4 use ONE_HALF = 0x7f (not 0x80). */

static inline uint32_t cairo_mul8 (uint32_t a, uint32_t b)
{
  uint32_t t = a * b + 0x7f;
  return ((t >> 8) + t) >> 8;
}

static inline uint32_t cairo_lerp (uint32_t s, uint32_t m, uint32_t d)
{
  if (m == 0)
    return d;
  if (m == 255)
    return s;
  uint32_t im = 255 - m, r = 0;
  for (int sh = 0; sh < 32; sh += 8)
    r |= add_sat8 (cairo_mul8 ((s >> sh) & 0xff, m),
        cairo_mul8 ((d >> sh) & 0xff, im)) << sh;
  return r;
}

uint32_t flutl_oracle_blend (uint32_t rule, uint32_t s, uint32_t m, uint32_t d)
{
  return rule == FLUTL_RULE_LERP ? cairo_lerp (s, m, d)
                                 : pixman_over_mask (s, m, d);
}

uint32_t flutl_oracle_over (uint32_t s, uint32_t d)
{
  return pixman_over (s, d);
}

/* (group pixel g IN alpha) OVER d -- pixman over_8888_n_8888 */
uint32_t flutl_oracle_group (uint32_t g, uint32_t alpha, uint32_t d)
{
  if (g == 0)
    return d;
  return pixman_over (un8x4_mul_un8 (g, alpha), d);
}

/* ---- cairo colour quantisation -------------------------------------------
 * cairo-color.c: _cairo_color_double_to_short (d) = d * 65535.0 + 0.5;
 * _cairo_color_compute_shorts premultiplies in double first; the image
 * backend then keeps the high byte of each 16-bit channel. */
uint32_t flutl_oracle_color (double r, double g, double b, double a)
{
  if (r < 0) r = 0; if (r > 1) r = 1;
  if (g < 0) g = 0; if (g > 1) g = 1;
  if (b < 0) b = 0; if (b > 1) b = 1;
  if (a < 0) a = 0; if (a > 1) a = 1;
  uint16_t as = (uint16_t) (a * 65535.0 + 0.5);
  uint16_t rs = (uint16_t) (r * a * 65535.0 + 0.5);
  uint16_t gs = (uint16_t) (g * a * 65535.0 + 0.5);
  uint16_t bs = (uint16_t) (b * a * 65535.0 + 0.5);
  return ((uint32_t) (as >> 8) << 24) | ((uint32_t) (rs >> 8) << 16) |
      ((uint32_t) (gs >> 8) << 8) | (uint32_t) (bs >> 8);
}

/* ---- cairo rectangular scan converter ------------------------------------
 * cairo-rectangular-scan-converter.c: a box in 24.8 fixed point covers pixel
 * (px,py) with area = (x overlap in 1/256) * (y overlap in 1/256) in
 * [0,65536]; the 8-bit coverage handed to the span renderer is
 * min(255, area >> 8).  (Pinned by a full 257x257 sweep against cairo 1.16.0;
 * the more obvious (area*255+32768)>>16 is NOT what cairo does.) */
static inline int overlap256 (int lo, int hi, int p)
{
  int a = p * 256, b = a + 256;
  if (lo > a) a = lo;
  if (hi < b) b = hi;
  return b > a ? b - a : 0;
}

uint32_t flutl_oracle_rect_coverage (int fx1, int fy1, int fx2, int fy2,
    int px, int py)
{
  uint32_t area = (uint32_t) overlap256 (fx1, fx2, px) *
      (uint32_t) overlap256 (fy1, fy2, py);
  area >>= 8;
  return area > 255 ? 255 : area;
}

/* ---- the compositor -------------------------------------------------------
 * Restates the paint sequence of gst_ttmlrender_gen_buffer (:1446-1467):
 * clear, then regions in list order; inside a region the optional opacity
 * group of gst_ttmlrender_show_regions (:1250-1256, :1375-1382). */

static uint32_t prim_coverage (const flutl_prim *p, const uint8_t *atlas,
    int x, int y)
{
  if (p->op == FLUTL_OP_RECT)
    return flutl_oracle_rect_coverage (p->a, p->b, p->c, p->d, x, y);
  /* MASK */
  return atlas[(size_t) p->a + (size_t) (y - p->d) * (size_t) p->b +
      (size_t) (x - p->c)];
}

int flutl_oracle_render (uint32_t width, uint32_t height, uint32_t *dst,
    size_t dst_stride_bytes, const flutl_region *regions, uint32_t n_regions,
    const flutl_prim *prims, const uint8_t *atlas)
{
  size_t stride = dst_stride_bytes / 4;
  uint32_t *group = NULL;
  size_t group_cap = 0;

  for (uint32_t y = 0; y < height; y++)
    memset (dst + y * stride, 0, (size_t) width * 4);

  for (uint32_t r = 0; r < n_regions; r++) {
    const flutl_region *rg = &regions[r];
    int in_group = 0;
    int gx0 = rg->x0, gy0 = rg->y0, gw = rg->x1 - rg->x0, gh = rg->y1 - rg->y0;

    for (uint32_t i = rg->first; i < rg->first + rg->count; i++) {
      const flutl_prim *p = &prims[i];
      switch (p->op) {
        case FLUTL_OP_GROUP_BEGIN: {
          size_t need = (size_t) (gw > 0 ? gw : 0) * (size_t) (gh > 0 ? gh : 0);
          if (need > group_cap) {
            free (group);
            group = (uint32_t *) malloc (need * 4 + 4);
            if (!group)
              return FLUTL_ENOMEM;
            group_cap = need;
          }
          memset (group, 0, need * 4);
          in_group = 1;
          break;
        }
        case FLUTL_OP_GROUP_END:
          for (int y = p->cy0; y < p->cy1; y++)
            for (int x = p->cx0; x < p->cx1; x++) {
              uint32_t g = group[(size_t) (y - gy0) * gw + (x - gx0)];
              uint32_t *d = &dst[(size_t) y * stride + x];
              *d = flutl_oracle_group (g, p->param, *d);
            }
          in_group = 0;
          break;
        case FLUTL_OP_RECT:
        case FLUTL_OP_MASK:
          for (int y = p->cy0; y < p->cy1; y++)
            for (int x = p->cx0; x < p->cx1; x++) {
              uint32_t *d = in_group
                  ? &group[(size_t) (y - gy0) * gw + (x - gx0)]
                  : &dst[(size_t) y * stride + x];
              uint32_t m = prim_coverage (p, atlas, x, y);
              *d = flutl_oracle_blend (p->rule, p->color, m, *d);
            }
          break;
        case FLUTL_OP_IMAGE:
          for (int y = p->cy0; y < p->cy1; y++)
            for (int x = p->cx0; x < p->cx1; x++) {
              uint32_t *d = in_group
                  ? &group[(size_t) (y - gy0) * gw + (x - gx0)]
                  : &dst[(size_t) y * stride + x];
              uint32_t s;
              memcpy (&s, atlas + (size_t) p->a +
                  (size_t) (y - p->d) * (size_t) p->b +
                  (size_t) (x - p->c) * 4, 4);
              *d = pixman_over (s, *d);
            }
          break;
        default:
          free (group);
          return FLUTL_EINVAL;
      }
    }
  }
  free (group);
  return FLUTL_OK;
}

/* ---- blur ------------------------------------------------------------------
 * gst_ttml_blur_create_gaussian_kernel (plugins/ttml/gstttmlblur.c:27-67):
 * G(x,y) = 1/(2 pi s^2) exp(-(x^2+y^2)/(2 s^2)), normalised, converted with
 * pixman_double_to_fixed (truncation towards zero of d * 65536). */
int flutl_oracle_blur_kernel (int radius, double sigma, int32_t *out)
{
  const double scale2 = 2.0 * sigma * sigma;
  const double scale1 = 1.0 / (M_PI * scale2);
  const int size = 2 * radius + 1;
  const int n = size * size;
  double *tmp = (double *) malloc (sizeof (double) * (size_t) n);
  double sum = 0;
  int i = 0;
  if (!tmp)
    return FLUTL_ENOMEM;
  for (int x = -radius; x <= radius; ++x)
    for (int y = -radius; y <= radius; ++y, ++i) {
      const double u = x * x;
      const double v = y * y;
      tmp[i] = scale1 * exp (-(u + v) / scale2);
      sum += tmp[i];
    }
  for (i = 0; i < n; ++i)
    out[i] = (int32_t) ((tmp[i] / sum) * 65536.0);
  free (tmp);
  return FLUTL_OK;
}

/* gst_ttml_blur_image_surface (gstttmlblur.c:71-110) = pixman
 * PIXMAN_FILTER_CONVOLUTION + PIXMAN_OP_SRC with identity transform and
 * REPEAT_NONE.  pixman-bits-image.c (0.40.0) bits_image_fetch_pixel_convolution:
 * sample position is the pixel centre; x1 = int(x - e - ((cw-1)<<16)/2);
 * per channel acc += chan * f (f == 0 taps skipped); result
 * (acc + 0x8000) >> 16 clipped to [0,255]. */
int flutl_oracle_blur (const uint32_t *src, uint32_t *dst, uint32_t w,
    uint32_t h, size_t stride_bytes, const int32_t *kernel, int radius)
{
  const int size = 2 * radius + 1;
  size_t stride = stride_bytes / 4;
  for (int y = 0; y < (int) h; y++)
    for (int x = 0; x < (int) w; x++) {
      /* centre (x + 0.5) in 16.16 minus e minus (size-1)/2 -> tap origin */
      int32_t fx = (x << 16) + 0x8000 - 1 - (((size - 1) << 16) >> 1);
      int32_t fy = (y << 16) + 0x8000 - 1 - (((size - 1) << 16) >> 1);
      int x1 = fx >> 16, y1 = fy >> 16;
      int32_t acc[4] = { 0, 0, 0, 0 };
      const int32_t *k = kernel;
      for (int j = y1; j < y1 + size; j++)
        for (int i = x1; i < x1 + size; i++) {
          int32_t f = *k++;
          if (!f)
            continue;
          if (i < 0 || j < 0 || i >= (int) w || j >= (int) h)
            continue;
          uint32_t p = src[(size_t) j * stride + i];
          acc[0] += (int32_t) (p & 0xff) * f;
          acc[1] += (int32_t) ((p >> 8) & 0xff) * f;
          acc[2] += (int32_t) ((p >> 16) & 0xff) * f;
          acc[3] += (int32_t) (p >> 24) * f;
        }
      uint32_t o = 0;
      for (int c = 0; c < 4; c++) {
        int32_t v = (acc[c] + 0x8000) >> 16;
        if (v < 0) v = 0;
        if (v > 255) v = 255;
        o |= (uint32_t) v << (8 * c);
      }
      dst[(size_t) y * stride + x] = o;
    }
  return FLUTL_OK;
}

/* An ARGB32 surface blitted at a fractional device position (the blurred
 * outline of gstttmlrender.c:1219-1224 under a CTM with fractional x0/y0):
 * cairo 1.16 selects PIXMAN_FILTER_BILINEAR; pixman 0.40 samples at the pixel
 * centre, x1 = x * 65536 + t (t = the pattern translation in 16.16, rounded to
 * nearest from the double), weights = 7 bits of the fraction, constant over
 * the image because the transform is a pure translation
 * (pixman-inlines.h bilinear_interpolation, BILINEAR_INTERPOLATION_BITS = 7).
 *
 * flutl_oracle_frac_origin: device origin (double) -> integer destination of
 * resampled pixel (0,0) and the 7-bit weight of the right/lower neighbour.
 * flutl_oracle_resample: w x h -> (w+1) x (h+1); pixel (i,j) mixes source
 * pixels (i-1,j-1), (i,j-1), (i-1,j), (i,j), transparent outside. */
void flutl_oracle_frac_origin (double origin, int *base, uint32_t *dist)
{
  const int64_t t = (int64_t) floor (-origin * 65536.0 + 0.5);
  *base = (int) (-1 - (t >> 16));
  *dist = (uint32_t) ((t >> 9) & 0x7f);
}

static uint32_t bilinear_px (uint32_t tl, uint32_t tr, uint32_t bl, uint32_t br,
    uint32_t distx, uint32_t disty)
{
  distx <<= 1;
  disty <<= 1;
  const uint32_t dxy = distx * disty, dxiy = (distx << 8) - dxy,
      dixy = (disty << 8) - dxy,
      dixiy = 256 * 256 - (disty << 8) - (distx << 8) + dxy;
  uint32_t r = 0;
  for (int sh = 0; sh < 32; sh += 8) {
    const uint32_t c = ((tl >> sh) & 255) * dixiy + ((tr >> sh) & 255) * dxiy +
        ((bl >> sh) & 255) * dixy + ((br >> sh) & 255) * dxy;
    r |= ((c >> 16) & 255) << sh;
  }
  return r;
}

/* box: NULL, or the 24.8 box {x1, y1, x2, y2} of the cairo_rectangle + cairo_fill
 * that the reference paints the surface through (:1221-1224), relative to
 * resampled pixel (0,0): its antialiased edge coverage (rectangular scan
 * converter, see flutl_oracle_rect_coverage) multiplies the sample (pixman IN
 * with an a8 mask: MUL_UN8 per channel) before the OVER. */
void flutl_oracle_resample (const uint32_t *src, uint32_t w, uint32_t h,
    uint32_t distx, uint32_t disty, const int32_t *box,
    uint32_t *dst /* (w+1) x (h+1) */)
{
#define SRC(x, y) (((x) >= 0 && (y) >= 0 && (x) < (int) w && (y) < (int) h) \
    ? src[(size_t) (y) * w + (x)] : 0u)
  for (int j = 0; j <= (int) h; j++)
    for (int i = 0; i <= (int) w; i++)
    {
      uint32_t v = bilinear_px (SRC (i - 1, j - 1), SRC (i, j - 1),
          SRC (i - 1, j), SRC (i, j), distx, disty);
      if (box) {
        const uint32_t cov = flutl_oracle_rect_coverage (box[0], box[1], box[2], box[3], i, j);
        v = un8x4_mul_un8 (v, cov);
      }
      dst[(size_t) j * (w + 1) + i] = v;
    }
#undef SRC
}

/* ---- array forms (so that Python tests can check whole tables quickly) --- */
void flutl_oracle_blend_n (uint32_t rule, const uint32_t *s, const uint32_t *m,
    const uint32_t *d, uint32_t *out, size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = flutl_oracle_blend (rule, s[i], m[i], d[i]);
}

void flutl_oracle_over_n (const uint32_t *s, const uint32_t *d, uint32_t *out,
    size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = pixman_over (s[i], d[i]);
}

void flutl_oracle_group_n (const uint32_t *g, const uint32_t *alpha,
    const uint32_t *d, uint32_t *out, size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = flutl_oracle_group (g[i], alpha[i], d[i]);
}

/* colour as the reference packs it: 0xRRGGBBAA with each byte / 255.0
 * (GET_CAIRO_COMP, gstttmlrender.c:1178) */
uint32_t flutl_oracle_color_rgba8888 (uint32_t c)
{
  return flutl_oracle_color (((c >> 24) & 255) / 255.0, ((c >> 16) & 255) / 255.0,
      ((c >> 8) & 255) / 255.0, (c & 255) / 255.0);
}

void flutl_oracle_color_rgba8888_n (const uint32_t *c, uint32_t *out, size_t n)
{
  for (size_t i = 0; i < n; i++)
    out[i] = flutl_oracle_color_rgba8888 (c[i]);
}

/* paint_with_alpha opacity -> 8-bit mask alpha (solid mask pattern colour,
 * alpha_short >> 8) */
uint32_t flutl_oracle_opacity_to_alpha (double opacity)
{
  if (opacity < 0) opacity = 0;
  if (opacity > 1) opacity = 1;
  return ((uint16_t) (opacity * 65535.0 + 0.5)) >> 8;
}

/* _cairo_fixed_from_double: round to nearest 1/256, ties to even (the
 * 1.5 * 2^(52-8) magic-number addition) */
int32_t flutl_oracle_fixed_from_double (double d)
{
  return (int32_t) nearbyint (d * 256.0);
}
