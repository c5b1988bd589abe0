// flutl_kernels.hip -- gfx950 (CDNA4, MI355X) kernels of the fluttml compositor.
//
// Device side of gst_ttmlrender_gen_buffer (reference plugins/ttml/
// gstttmlrender.c:1427-1478): CLEAR + the z-ordered region loop of
// gst_ttmlrender_show_regions (:1235-1385), over the display-list IR of
// include/flutl.h.  The per-pixel arithmetic is the reference's third-party
// arithmetic (pixman 0.40.0 OVER/IN, cairo 1.16.0 span lerp and rectangular
// scan converter) -- bit exact, integer only.  No MFMA: this is byte-wise
// blending bound by HBM stores.
//
// Work decomposition (see DESIGN.md for the measurements behind it)
//   * the blend granule is the 64x8 output tile: a wave instruction covers a
//     64x4 strip, lane l owning the 4 adjacent pixels (16 bytes) at column
//     4*(l%16), row l/16, so a wave store writes four 256-byte row segments;
//   * one workgroup (2 wavefronts) owns a 64x32 "supertile" = 4 stacked tiles:
//     region culling and primitive binning are done once per supertile and
//     each lane carries 4 row groups, which amortises the dependent
//     round trips (region table -> primitives -> mask bytes) that bound a
//     one-tile-per-workgroup kernel (v1: 41 us/4K frame, latency bound);
//   * every output pixel is written exactly once, from registers (there is no
//     clear pass and no read of the destination);
//   * supertiles are dealt to XCDs in contiguous bands (blocks b and b+8 share
//     an XCD's L2), so a band's glyph masks are fetched into one L2 only;
//   * regions are culled 64 at a time (one per lane + ballot), a region's
//     primitives are binned by the 128 lanes in parallel (ballot + prefix
//     popcount -> ordered, compacted 32-byte prims in LDS), then replayed in
//     paint order with the next primitive's mask bytes already in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flutl.h"

namespace {

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));

constexpr int kThreads = 64;           // one wavefront per workgroup

// ---- packed 2 x 16-bit channel arithmetic ---------------------------------
// A pixel is kept unpacked as two registers: rb = {B, R}, ag = {G, A}, one
// 8-bit channel per 16-bit lane, so v_pk_mad_u16 & co process two channels.

struct Px {
  u16x2 rb, ag;
};

__device__ __forceinline__ u16x2 as_u16x2 (uint32_t v)
{
  return __builtin_bit_cast (u16x2, v);
}
__device__ __forceinline__ uint32_t as_u32 (u16x2 v)
{
  return __builtin_bit_cast (uint32_t, v);
}
__device__ __forceinline__ Px unpack (uint32_t p)
{
  Px r;
  r.rb = as_u16x2 (p & 0x00ff00ffu);
  r.ag = as_u16x2 ((p >> 8) & 0x00ff00ffu);
  return r;
}
__device__ __forceinline__ uint32_t pack (Px p)
{
  return as_u32 (p.rb) | (as_u32 (p.ag) << 8);
}
__device__ __forceinline__ u16x2 splat (uint32_t v)
{
  return as_u16x2 (v | (v << 16));
}

// pixman MUL_UN8 on two channels: t = x*a + 0x80; (t + (t >> 8)) >> 8
__device__ __forceinline__ u16x2 mul_un8 (u16x2 x, u16x2 a)
{
  u16x2 t = x * a + (u16x2) (0x80);
  return (t + (t >> 8)) >> 8;
}
// cairo mul8_8 (cairo-image-compositor.c): same with ONE_HALF = 0x7f
__device__ __forceinline__ u16x2 mul_7f (u16x2 x, u16x2 a)
{
  u16x2 t = x * a + (u16x2) (0x7f);
  return (t + (t >> 8)) >> 8;
}
__device__ __forceinline__ u16x2 add_sat (u16x2 a, u16x2 b)
{
  return __builtin_elementwise_min ((u16x2) (a + b), (u16x2) (255));
}

// d = s OVER d (pixman), s premultiplied
__device__ __forceinline__ void over (Px &d, Px s)
{
  uint32_t a = as_u32 (s.ag) >> 16;
  u16x2 ia = splat (255u - a);
  d.rb = add_sat (s.rb, mul_un8 (d.rb, ia));
  d.ag = add_sat (s.ag, mul_un8 (d.ag, ia));
}

// d = (s IN m) OVER d   -- pixman over_n_8_8888 / glyph compositing
// Branch free: m == 0 gives sm = 0 and 0 OVER d == d exactly (MUL_UN8 (x, 255)
// == x), so lanes without coverage need no early-out.
__device__ __forceinline__ void blend_pixman (Px &d, Px s, uint32_t m)
{
  u16x2 mm = splat (m);
  Px sm;
  sm.rb = mul_un8 (s.rb, mm);
  sm.ag = mul_un8 (s.ag, mm);
  over (d, sm);
}

// cairo in-place lerp for opaque solid sources (0x7f rounding)
// Branch free as well: mul_7f (x, 255) == x and mul_7f (x, 0) == 0, so m == 0
// leaves d and m == 255 yields s exactly, as cairo's special cases do.
__device__ __forceinline__ void blend_lerp (Px &d, Px s, uint32_t m)
{
  u16x2 mm = splat (m), im = splat (255u - m);
  d.rb = add_sat (mul_7f (s.rb, mm), mul_7f (d.rb, im));
  d.ag = add_sat (mul_7f (s.ag, mm), mul_7f (d.ag, im));
}

// cairo rectangular scan converter: overlap of [lo,hi) (24.8) with pixel p
__device__ __forceinline__ int overlap256 (int lo, int hi, int p)
{
  int a = max (lo, p << 8), b = min (hi, (p << 8) + 256);
  return max (b - a, 0);
}

struct FrameParams {
  uint32_t width, height;
  uint32_t st_x, st_y;                  // supertiles per row / column
  uint32_t n_frames;
  uint32_t region_first, region_count;  // used when frames == nullptr
  uint32_t per_xcd;                     // ceil(total_supertiles / 8)
  uint64_t stride;                      // bytes
};

// Workgroup footprint: a 64 x 32 "supertile" = four stacked 64x8 output tiles.
// Wave w owns rows [16w, 16w+16); lane l owns, in each of the wave's 4 row
// groups r, the 4 pixels at column 4*(l%16), row 4*r + l/16 -- so one store
// instruction of a wave writes four 256-byte row segments.
constexpr int kSuperW = FLUTL_TILE_W;        // 64
constexpr int kRG = 4;                       // row groups per wave
constexpr int kSuperH = 4 * kRG;             // kRG strips of 4 rows

// A primitive held wave-uniformly in SGPRs.  During binning every lane holds
// one candidate primitive (32 bytes = 8 VGPRs); a hit is broadcast to the
// whole wave with 8 v_readlane -- no LDS list, no scalar loads.
struct PrimU {
  uint32_t op, rule, param;
  int cx0, cy0, cx1, cy1;
  uint32_t color;
  int a, b, c, d;
};

__device__ __forceinline__ uint32_t rdl (uint32_t v, int l)
{
  return (uint32_t) __builtin_amdgcn_readlane ((int) v, l);
}

__device__ __forceinline__ PrimU prim_from_lane (const uint4 &lo, const uint4 &hi, int l)
{
  const uint32_t w0 = rdl (lo.x, l), w1 = rdl (lo.y, l), w2 = rdl (lo.z, l);
  PrimU p;
  p.op = w0 & 0xffu;
  p.rule = (w0 >> 8) & 0xffu;
  p.param = w0 >> 16;
  p.cx0 = (int16_t) (w1 & 0xffffu);
  p.cy0 = (int16_t) (w1 >> 16);
  p.cx1 = (int16_t) (w2 & 0xffffu);
  p.cy1 = (int16_t) (w2 >> 16);
  p.color = rdl (lo.w, l);
  p.a = (int) rdl (hi.x, l);
  p.b = (int) rdl (hi.y, l);
  p.c = (int) rdl (hi.z, l);
  p.d = (int) rdl (hi.w, l);
  return p;
}

// The strip's 4 pixels per lane are kept UNPACKED (rb / ag, 2 x 16 bit each):
// a primitive then costs no unpack/pack, only the final store packs.

// splat byte k of w into both 16-bit lanes: {w.k, 0, w.k, 0} (one v_perm_b32)
template <int K>
__device__ __forceinline__ u16x2 splat_byte (uint32_t w)
{
  return as_u16x2 (__builtin_amdgcn_perm (0u, w, 0x0c000c00u | (uint32_t) K | ((uint32_t) K << 16)));
}

// (solid s IN m) OVER t, pixman rule, for a VALIDATED premultiplied colour
// (every channel <= alpha, checked by flutl_list_validate): then
// s*m + t*(255 - sa*m) cannot exceed 255 and the saturating add is not needed.
//   OPAQUE: alpha(s) == 255, so alpha(s IN m) == m and 255 - m == 255 ^ m.
template <bool OPAQUE, int K>
__device__ __forceinline__ void px_pixman (Px &t, Px s, uint32_t w)
{
  const u16x2 mm = splat_byte<K> (w);
  const u16x2 sm_rb = mul_un8 (s.rb, mm);
  const u16x2 sm_ag = mul_un8 (s.ag, mm);
  u16x2 ia;
  if (OPAQUE)
    ia = as_u16x2 (as_u32 (mm) ^ 0x00ff00ffu);
  else   // {a, 0, a, 0} from byte 2 of sm_ag, then 255 - a
    ia = as_u16x2 (0x00ff00ffu - __builtin_amdgcn_perm (0u, as_u32 (sm_ag), 0x0c020c02u));
  t.rb = sm_rb + mul_un8 (t.rb, ia);
  t.ag = sm_ag + mul_un8 (t.ag, ia);
}

// cairo in-place lerp (0x7f rounding); mul_7f (255, m) == m, so the sum never
// exceeds 255 either.  RB_ZERO: the colour has no red/blue (black outlines).
template <bool RB_ZERO, int K>
__device__ __forceinline__ void px_lerp (Px &t, Px s, uint32_t w)
{
  const u16x2 mm = splat_byte<K> (w);
  const u16x2 im = as_u16x2 (as_u32 (mm) ^ 0x00ff00ffu);
  if (RB_ZERO)
    t.rb = mul_7f (t.rb, im);
  else
    t.rb = mul_7f (s.rb, mm) + mul_7f (t.rb, im);
  t.ag = mul_7f (s.ag, mm) + mul_7f (t.ag, im);
}

template <bool LERP, bool FLAG>
__device__ __forceinline__ void strip_blend (Px (&t)[4], Px s, uint32_t w)
{
  if (LERP) {
    px_lerp<FLAG, 0> (t[0], s, w);
    px_lerp<FLAG, 1> (t[1], s, w);
    px_lerp<FLAG, 2> (t[2], s, w);
    px_lerp<FLAG, 3> (t[3], s, w);
  } else {
    px_pixman<FLAG, 0> (t[0], s, w);
    px_pixman<FLAG, 1> (t[1], s, w);
    px_pixman<FLAG, 2> (t[2], s, w);
    px_pixman<FLAG, 3> (t[3], s, w);
  }
}

// The 4 A8 coverage bytes of a MASK primitive for this lane's 4 pixels on row
// y.  Issued one primitive ahead of use: OVER is order dependent, the loads
// are not.  Branch free AND select free: lanes that the prim does not touch
// (and every lane, for other ops) load the first dword of the atlas, which is
// zero by contract (include/flutl.h: leading slack), so nothing depends on the
// loaded value until the blend -- a select on the value would make the
// compiler wait for each load right where it is issued.
// Bytes left/right of the mask read its zero padding (atlas contract).
__device__ __forceinline__ uint32_t mask_fetch (const PrimU &p,
    const uint8_t *__restrict__ atlas, int x, int y)
{
  // wave uniform: no rows for other ops
  const uint32_t rows = p.op == FLUTL_OP_MASK ? (uint32_t) (p.cy1 - p.cy0) : 0u;
  const bool valid = (uint32_t) (y - p.cy0) < rows &&
      (uint32_t) (x + 3 - p.cx0) < (uint32_t) (p.cx1 - p.cx0 + 3);
  // rows and strides are < 2^15: 24-bit multiply-add (full rate)
  const uint32_t off = __umul24 ((uint32_t) (y - p.d), (uint32_t) p.b) +
      (uint32_t) (p.a - p.c) + (uint32_t) x;
  uint32_t w;
  __builtin_memcpy (&w, atlas + (valid ? off : 0u), 4);
  return w;
}

// Per-lane pixel predicates as a byte mask over the lane's 4 pixels, by
// unsigned range checks and shifts (VALU only: compare + s_and chains on
// 64-bit lane masks would load the scalar unit instead).
__device__ __forceinline__ uint32_t pixel_mask (const PrimU &p, int x, bool rowin)
{
  const int klo = min (max (p.cx0 - x, 0), 4);   // first pixel inside
  const int khi = min (max (p.cx1 - x, 0), 4);   // one past the last
  const uint32_t mlo = (uint32_t) ((1ull << (8 * klo)) - 1ull);
  const uint32_t mhi = (uint32_t) ((1ull << (8 * khi)) - 1ull);
  return rowin ? (mhi & ~mlo) : 0u;
}

// The lane's 4 row groups of a MASK primitive, fetched together (4 loads in
// flight) one primitive ahead of their use.
struct MaskFetch {
  uint32_t w[kRG];
};

__device__ __forceinline__ MaskFetch mask_fetch4 (const PrimU &p,
    const uint8_t *__restrict__ atlas, int x, int y0)
{
  MaskFetch f;
#pragma unroll
  for (int r = 0; r < kRG; r++)
    f.w[r] = mask_fetch (p, atlas, x, y0 + 4 * r);
  return f;
}

// blend one strip (4 packed pixels of the lane) with coverage dword w
__device__ __forceinline__ void strip_apply (uint32_t (&tp)[4], const PrimU &p, uint32_t w)
{
  Px t[4] = { unpack (tp[0]), unpack (tp[1]), unpack (tp[2]), unpack (tp[3]) };
  const Px s = unpack (p.color);
  if (p.rule == FLUTL_RULE_LERP) {
    if ((p.color & 0x00ff00ffu) == 0u)
      strip_blend<true, true> (t, s, w);
    else
      strip_blend<true, false> (t, s, w);
  } else {
    if ((p.color >> 24) == 255u)
      strip_blend<false, true> (t, s, w);
    else
      strip_blend<false, false> (t, s, w);
  }
#pragma unroll
  for (int k = 0; k < 4; k++)
    tp[k] = pack (t[k]);
  // do not let the scheduler interleave the blends of different strips: it
  // would trade occupancy for ILP that the other waves already provide
  __builtin_amdgcn_sched_barrier (0);
}

__global__ __launch_bounds__ (kThreads) void flutl_composite_kernel (
    FrameParams fp, uint8_t *__restrict__ dst,
    const flutl_frame_desc *__restrict__ frames,
    const flutl_region *__restrict__ regions,
    const flutl_prim *__restrict__ prims, const uint8_t *__restrict__ atlas)
{
  // frame pixels parked during an opacity group: lane-private slots, so no
  // barrier; this is the kernel's only LDS
  __shared__ uint4 s_park[kRG][kThreads];

  // One wavefront per workgroup: nothing is shared between waves.
  const int lane = threadIdx.x;

  // Blocks b and b+8 share an XCD (and its L2): XCD k gets the k-th contiguous
  // band of supertiles, so the glyph masks and display-list entries of a band
  // are fetched into one L2 only.  Speed only; any placement is correct.
  const uint32_t st_per_frame = fp.st_x * fp.st_y;
  const uint32_t total = st_per_frame * fp.n_frames;
  const uint32_t logical = (blockIdx.x & 7u) * fp.per_xcd + (blockIdx.x >> 3);
  if (logical >= total)
    return;
  const uint32_t frame = logical / st_per_frame;
  const uint32_t st = logical - frame * st_per_frame;
  const uint32_t sty = st / fp.st_x;
  const uint32_t stx = st - sty * fp.st_x;

  uint32_t rfirst = fp.region_first, rcount = fp.region_count;
  uint64_t dst_off = 0;
  if (frames) {
    rfirst = frames[frame].region_first;
    rcount = frames[frame].region_count;
    dst_off = frames[frame].dst_offset;
  }

  const int sx0 = (int) stx * kSuperW, sy0 = (int) sty * kSuperH;
  const int x_lane = sx0 + (lane & 15) * 4;
  const int y_lane = sy0 + (lane >> 4);         // row in strip 0; strip r: + 4r

  // t = the surface being painted, 4 strips x 4 packed pixels per lane: the
  // frame, or -- between GROUP_BEGIN and GROUP_END -- the region's group
  // surface while the frame pixels are parked in s_park.
  uint32_t t[kRG][4];
#pragma unroll
  for (int r = 0; r < kRG; r++)
#pragma unroll
    for (int k = 0; k < 4; k++)
      t[r][k] = 0u;

  // ---- regions, 64 at a time: one per lane, culled against the supertile ----
  for (uint32_t rbase = 0; rbase < rcount; rbase += 64) {
    // lanes past the end re-read the last region and are masked out
    const uint32_t ridx = min (rbase + (uint32_t) lane, rcount - 1u);
    const uint4 rv = *reinterpret_cast<const uint4 *> (&regions[rfirst + ridx]);
    const int rx0 = (int16_t) (rv.x & 0xffffu), ry0r = (int16_t) (rv.x >> 16);
    const int rx1 = (int16_t) (rv.y & 0xffffu), ry1r = (int16_t) (rv.y >> 16);
    const bool rhit = rbase + lane < rcount && rx0 < sx0 + kSuperW && rx1 > sx0 &&
        ry0r < sy0 + kSuperH && ry1r > sy0 && rv.w != 0;
    unsigned long long rmask = __ballot (rhit);
    while (rmask) {
      const int rl = __builtin_ctzll (rmask);
      rmask &= rmask - 1;
      const uint32_t pfirst = rdl (rv.z, rl), pcount = rdl (rv.w, rl);

      // ---- the region's prims, 64 at a time: bin and replay fused ----------
      for (uint32_t pbase = 0; pbase < pcount; pbase += 64) {
        const uint32_t pi = pbase + (uint32_t) lane;
        const uint4 *pp = reinterpret_cast<const uint4 *> (
            &prims[pfirst + min (pi, pcount - 1u)]);
        const uint4 lo = pp[0], hi = pp[1];
        const int bx0 = (int16_t) (lo.y & 0xffffu), by0 = (int16_t) (lo.y >> 16);
        const int bx1 = (int16_t) (lo.z & 0xffffu), by1 = (int16_t) (lo.z >> 16);
        const bool hit = pi < pcount && bx0 < sx0 + kSuperW && bx1 > sx0 &&
            by0 < sy0 + kSuperH && by1 > sy0;
        unsigned long long hits = __ballot (hit);
        if (!hits)
          continue;

        // replay the hits in paint order (= lane order); the next hit's mask
        // bytes are requested before the current one is blended
        PrimU nxt = prim_from_lane (lo, hi, __builtin_ctzll (hits));
        hits &= hits - 1;
        MaskFetch fnxt = mask_fetch4 (nxt, atlas, x_lane, y_lane);
        for (;;) {
          // Re-materialise the lane coordinates per primitive: left visible,
          // LICM hoists dozens of x+k / y+4r derived VGPRs out of the loops.
          int x = x_lane, y0 = y_lane;
          asm volatile ("" : "+v"(x), "+v"(y0));
          const PrimU p = nxt;
          const MaskFetch f = fnxt;
          const bool more = hits != 0;
          if (more) {
            nxt = prim_from_lane (lo, hi, __builtin_ctzll (hits));
            hits &= hits - 1;
            fnxt = mask_fetch4 (nxt, atlas, x, y0);
          }

          if (p.op == FLUTL_OP_GROUP_BEGIN) {
            uint32_t zero = 0;
            asm volatile ("" : "+v"(zero));    // no zero vector kept alive
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              s_park[r][lane] = make_uint4 (t[r][0], t[r][1], t[r][2], t[r][3]);
#pragma unroll
              for (int k = 0; k < 4; k++)
                t[r][k] = zero;
            }
          } else if (p.op == FLUTL_OP_MASK) {
            // ---- solid colour through an A8 mask ---------------------------
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              const int ry0 = sy0 + 4 * r;           // scalar row test
              if (p.cy1 <= ry0 || p.cy0 >= ry0 + 4)
                continue;
              uint32_t w = f.w[r];
              if (p.param & FLUTL_MASK_XCLIP)
                w &= pixel_mask (p, x, true);
              if (__ballot (w != 0) == 0)
                continue;                            // nothing lands in this strip
              strip_apply (t[r], p, w);
            }
          } else if (p.op == FLUTL_OP_RECT) {
            // ---- solid colour through a 24.8 box ---------------------------
            const bool xfull = ((p.a + 255) >> 8) <= sx0 && (p.c >> 8) >= sx0 + kSuperW &&
                p.cx0 <= sx0 && p.cx1 >= sx0 + kSuperW;
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              const int ry0 = sy0 + 4 * r;
              if (p.cy1 <= ry0 || p.cy0 >= ry0 + 4)
                continue;
              const bool full = xfull && ((p.b + 255) >> 8) <= ry0 &&
                  (p.d >> 8) >= ry0 + 4 && p.cy0 <= ry0 && p.cy1 >= ry0 + 4;
              if (full) {
                // coverage 255 on the whole strip: t = s OVER t with a uniform
                // s.  Opaque s, LERP, or nothing painted yet: t = s exactly.
                if ((p.color >> 24) == 255u || p.rule == FLUTL_RULE_LERP ||
                    __ballot ((t[r][0] | t[r][1] | t[r][2] | t[r][3]) != 0) == 0) {
#pragma unroll
                  for (int k = 0; k < 4; k++)
                    t[r][k] = p.color;
                } else {
                  const Px sc = unpack (p.color);
#pragma unroll
                  for (int k = 0; k < 4; k++) {
                    Px v = unpack (t[r][k]);
                    over (v, sc);
                    t[r][k] = pack (v);
                  }
                }
                continue;
              }
              // cairo rectangular scan converter coverage
              const int y = y0 + 4 * r;
              const bool rowin = (uint32_t) (y - p.cy0) < (uint32_t) (p.cy1 - p.cy0);
              const int oy = overlap256 (p.b, p.d, y);
              uint32_t w = 0;
#pragma unroll
              for (int k = 0; k < 4; k++)
                w |= min ((uint32_t) (overlap256 (p.a, p.c, x + k) * oy) >> 8, 255u) << (8 * k);
              w &= pixel_mask (p, x, rowin);
              if (__ballot (w != 0) == 0)
                continue;
              strip_apply (t[r], p, w);
            }
          } else {
            // ---- per-pixel premultiplied source, OVER: IMAGE / GROUP_END ----
            const bool is_end = p.op == FLUTL_OP_GROUP_END;
            const u16x2 al = splat (p.param);
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              const int ry0 = sy0 + 4 * r;
              const bool touched = !(p.cy1 <= ry0 || p.cy0 >= ry0 + 4);
              if (!is_end && !touched)
                continue;
              const int y = y0 + 4 * r;
              const bool rowin = (uint32_t) (y - p.cy0) < (uint32_t) (p.cy1 - p.cy0);
              const uint32_t pmask = pixel_mask (p, x, rowin);
              uint32_t sv[4] = { 0u, 0u, 0u, 0u };
              if (is_end) {
                // the group surface is what was painted since GROUP_BEGIN;
                // swap the parked frame pixels back in
                const uint4 dd = s_park[r][lane];
#pragma unroll
                for (int k = 0; k < 4; k++)
                  sv[k] = ((pmask >> (8 * k)) & 1u) ? t[r][k] : 0u;
                t[r][0] = dd.x;
                t[r][1] = dd.y;
                t[r][2] = dd.z;
                t[r][3] = dd.w;
              } else {
                const uint32_t rowoff = (uint32_t) p.a +
                    (uint32_t) (y - p.d) * (uint32_t) p.b + (uint32_t) (x - p.c) * 4u;
#pragma unroll
                for (int k = 0; k < 4; k++)
                  if ((pmask >> (8 * k)) & 1u)
                    sv[k] = *reinterpret_cast<const uint32_t *> (
                        atlas + (rowoff + 4u * (uint32_t) k));
              }
              if (__ballot ((sv[0] | sv[1] | sv[2] | sv[3]) != 0) == 0)
                continue;   // 0 OVER d == d
#pragma unroll
              for (int k = 0; k < 4; k++) {
                Px sp = unpack (sv[k]);
                if (is_end) {   // (group IN alpha)
                  sp.rb = mul_un8 (sp.rb, al);
                  sp.ag = mul_un8 (sp.ag, al);
                }
                Px v = unpack (t[r][k]);
                over (v, sp);
                t[r][k] = pack (v);
              }
            }
          }
          if (!more)
            break;
        }
      }
    }
  }

  // ---- store: every pixel exactly once --------------------------------------
  if (x_lane < (int) fp.width) {
#pragma unroll
    for (int r = 0; r < kRG; r++) {
      const int y = y_lane + 4 * r;
      if (y >= (int) fp.height)
        continue;
      uint8_t *out = dst + dst_off + (size_t) y * fp.stride + (size_t) x_lane * 4;
      if (x_lane + 4 <= (int) fp.width && ((reinterpret_cast<uintptr_t> (out) & 15) == 0)) {
        *reinterpret_cast<uint4 *> (out) = make_uint4 (t[r][0], t[r][1], t[r][2], t[r][3]);
      } else {
        for (int k = 0; k < 4 && x_lane + k < (int) fp.width; k++)
          reinterpret_cast<uint32_t *> (out)[k] = t[r][k];
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Blur: pixman PIXMAN_FILTER_CONVOLUTION + OP_SRC, identity transform,
// REPEAT_NONE (reference plugins/ttml/gstttmlblur.c:71-110).  One workgroup
// per 32x8 output tile; the (32+2r) x (8+2r) source halo and the (2r+1)^2
// 16.16 weights are staged in LDS; accumulation is exact int32:
//   acc_c += chan_c * f;  out_c = clip ((acc_c + 0x8000) >> 16, 0, 255).
constexpr int kBlurTW = 32, kBlurTH = 8, kBlurMaxR = 16;

__global__ __launch_bounds__ (256) void flutl_blur_kernel_dev (
    const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int w, int h,
    int stride_px, const int32_t *__restrict__ kern, int radius)
{
  extern __shared__ __attribute__ ((aligned (16))) uint32_t smem[];
  const int size = 2 * radius + 1;
  const int hw = kBlurTW + 2 * radius, hh = kBlurTH + 2 * radius;
  uint32_t *s_src = smem;                       // hw * hh
  int32_t *s_k = reinterpret_cast<int32_t *> (smem + hw * hh);  // size*size
  const int tx0 = blockIdx.x * kBlurTW, ty0 = blockIdx.y * kBlurTH;
  const int tid = threadIdx.x;

  for (int i = tid; i < size * size; i += 256)
    s_k[i] = kern[i];
  // pixman: tap origin = int ((x + 0.5) - e - (size-1)/2) = x - radius
  for (int i = tid; i < hw * hh; i += 256) {
    const int sx = tx0 - radius + i % hw, sy = ty0 - radius + i / hw;
    s_src[i] = (sx >= 0 && sy >= 0 && sx < w && sy < h)
        ? src[(size_t) sy * stride_px + sx] : 0u;
  }
  __syncthreads ();

  const int lx = tid % kBlurTW, ly = tid / kBlurTW;
  const int x = tx0 + lx, y = ty0 + ly;
  if (x >= w || y >= h)
    return;
  int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int j = 0; j < size; j++)
    for (int i = 0; i < size; i++) {
      const int32_t f = s_k[j * size + i];
      const uint32_t p = s_src[(ly + j) * hw + lx + i];
      a0 += (int32_t) (p & 0xff) * f;
      a1 += (int32_t) ((p >> 8) & 0xff) * f;
      a2 += (int32_t) ((p >> 16) & 0xff) * f;
      a3 += (int32_t) (p >> 24) * f;
    }
  uint32_t o0 = (uint32_t) min (max ((a0 + 0x8000) >> 16, 0), 255);
  uint32_t o1 = (uint32_t) min (max ((a1 + 0x8000) >> 16, 0), 255);
  uint32_t o2 = (uint32_t) min (max ((a2 + 0x8000) >> 16, 0), 255);
  uint32_t o3 = (uint32_t) min (max ((a3 + 0x8000) >> 16, 0), 255);
  // hipcc (ROCm 7.2) fuses two of these clamps into gfx950's
  // v_ashr_pk_u8_i32 and then ORs the other channels into the result as if
  // its upper 16 bits were zero; on MI355X they are not (observed: R |= B).
  // Keep the four clamps opaque so that the plain med3 form is emitted.
  asm volatile ("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
  dst[(size_t) y * stride_px + x] = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}

int launch_tiles (hipStream_t stream, uint32_t width, uint32_t height, void *dst,
    size_t stride, const flutl_frame_desc *frames, uint32_t n_frames,
    uint32_t region_first, uint32_t region_count, const flutl_region *regions,
    const flutl_prim *prims, const uint8_t *atlas)
{
  if (!dst || width == 0 || height == 0 || n_frames == 0 || width > 32767 ||
      height > 32767 || (stride & 3) || stride < (size_t) width * 4)
    return FLUTL_EINVAL;
  FrameParams fp;
  fp.width = width;
  fp.height = height;
  fp.st_x = (width + kSuperW - 1) / kSuperW;
  fp.st_y = (height + kSuperH - 1) / kSuperH;
  fp.n_frames = n_frames;
  fp.region_first = region_first;
  fp.region_count = region_count;
  const uint64_t total = (uint64_t) fp.st_x * fp.st_y * n_frames;
  if (total > 0x7fffffffull)
    return FLUTL_EINVAL;
  fp.per_xcd = (uint32_t) ((total + 7) / 8);
  fp.stride = stride;
  const uint32_t grid = fp.per_xcd * 8u;   // one workgroup per supertile
  hipLaunchKernelGGL (flutl_composite_kernel, dim3 (grid), dim3 (kThreads),
      0, stream, fp, static_cast<uint8_t *> (dst), frames, regions, prims, atlas);
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}

}  // namespace

extern "C" int flutl_launch_frame (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas)
{
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst,
      dst_stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas);
}

extern "C" int flutl_launch_batch (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_frame_desc *frames,
    uint32_t n_frames, const flutl_region *regions, const flutl_prim *prims,
    const uint8_t *atlas)
{
  if (!frames)
    return FLUTL_EINVAL;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst,
      dst_stride_bytes, frames, n_frames, 0, 0, regions, prims, atlas);
}

extern "C" int flutl_launch_blur (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t stride_bytes, const int32_t *kernel_fixed,
    int radius)
{
  if (!src || !dst || !kernel_fixed || w == 0 || h == 0 || radius < 0 ||
      radius > kBlurMaxR || (stride_bytes & 3) || stride_bytes < (size_t) w * 4)
    return FLUTL_EINVAL;
  const int size = 2 * radius + 1;
  const size_t lds = ((size_t) (kBlurTW + 2 * radius) * (kBlurTH + 2 * radius) +
      (size_t) size * size) * 4;
  dim3 grid ((w + kBlurTW - 1) / kBlurTW, (h + kBlurTH - 1) / kBlurTH);
  hipLaunchKernelGGL (flutl_blur_kernel_dev, grid, dim3 (256), lds,
      static_cast<hipStream_t> (stream), static_cast<const uint32_t *> (src),
      static_cast<uint32_t *> (dst), (int) w, (int) h, (int) (stride_bytes / 4),
      kernel_fixed, radius);
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}
