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
// Work decomposition (DESIGN.md 5 has the measurements behind every choice)
//   * one wavefront = one workgroup = one 64x16 "supertile" (two stacked 64x8
//     tiles of the north star): a wave instruction covers a 64x4 strip, lane l
//     owning the 4 adjacent pixels (16 bytes) at column 4*(l%16), row l/16, so a
//     wave store writes four 256-byte row segments; 4 strips per wave;
//   * every output pixel is written exactly once (no clear pass, no read of the
//     destination); the supertile's pixels live in 4 KB of LDS, lane-private
//     16-byte slots, so no barrier and no bank conflict;
//   * strips that only background boxes have met are not pixels at all but one
//     colour in a scalar register ("uniform"), through opacity groups as well;
//   * regions are culled 64 at a time (one per lane + ballot); a hit region's
//     primitives are loaded 64 at a time, one per lane, bbox-tested, balloted,
//     and the hits replayed in paint order, each broadcast to SGPRs by
//     v_readlane, with the next hit's mask bytes already in flight;
//   * workgroups are dealt to the 8 XCDs round robin: whole frames per XCD for
//     batches, interleaved supertile rows otherwise (see the kernel).
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

// Ablation builds for profiling notes; 0 = the product.
#ifndef FLUTL_EXP
#define FLUTL_EXP 0
#endif


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
  // {0, A, 0, G} in one v_perm_b32 instead of shift + and
  r.ag = as_u16x2 (__builtin_amdgcn_perm (0u, p, 0x0c030c01u));
  return r;
}
__device__ __forceinline__ uint32_t pack (Px p)
{
  // {A, R, G, B} = {ag.byte2, rb.byte2, ag.byte0, rb.byte0}: one v_perm_b32
  // (the compiler makes shift + or, two instructions, of the plain expression)
  return __builtin_amdgcn_perm (as_u32 (p.ag), as_u32 (p.rb), 0x06020400u);
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
// (cairo's mul8_8, cairo-image-compositor.c, is the same with ONE_HALF = 0x7f: mul_rnd below)
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
  uint32_t frame_first;                 // first frame of this launch
  uint32_t xcd_frames, row_mul, row_add;   // workgroup mapping, see the kernel
  uint32_t row_base, row_end;           // supertile rows [row_base, row_end) are rendered
  uint32_t magic_stx;                   // ceil(2^32 / st_x), 0 if st_x == 1
  uint64_t stride;                      // bytes
};

// Workgroup (= wave) footprint: a 64 x 16 "supertile" = two stacked 64x8 output
// tiles = kRG strips of 4 rows.  Lane l owns, in each strip r, the 4 pixels at
// column 4*(l%16), row 4*r + l/16 -- so one store instruction of the wave
// writes four 256-byte row segments.
constexpr int kSuperW = FLUTL_TILE_W;        // 64
#ifndef FLUTL_KRG
#define FLUTL_KRG 4
#endif
constexpr int kRG = FLUTL_KRG;               // row groups per wave
constexpr int kSuperH = 4 * kRG;             // kRG strips of 4 rows
constexpr uint32_t kStInterior = 1u << 31;   // see `st` in the kernel
static_assert (kRG <= 7, "strip masks: 8 bits wide, the last one 7");

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

// The 8 raw dwords of lane l's primitive; decoded (cheap scalar ops) only
// where it is used, so that the prefetched NEXT primitive costs 8 SGPRs, not 12.
struct PrimRaw {
  uint32_t w[8];
};

__device__ __forceinline__ PrimRaw prim_raw_from_lane (const uint4 &lo, const uint4 &hi, int l)
{
  PrimRaw q;
  q.w[0] = rdl (lo.x, l);
  q.w[1] = rdl (lo.y, l);
  q.w[2] = rdl (lo.z, l);
  q.w[3] = rdl (lo.w, l);
  q.w[4] = rdl (hi.x, l);
  q.w[5] = rdl (hi.y, l);
  q.w[6] = rdl (hi.z, l);
  q.w[7] = rdl (hi.w, l);
  return q;
}

__device__ __forceinline__ PrimU prim_decode (const PrimRaw &q)
{
  const uint32_t w0 = q.w[0], w1 = q.w[1], w2 = q.w[2];
  PrimU p;
  p.op = w0 & 0xffu;
  p.rule = (w0 >> 8) & 0xffu;
  p.param = w0 >> 16;
  p.cx0 = (int16_t) (w1 & 0xffffu);
  p.cy0 = (int16_t) (w1 >> 16);
  p.cx1 = (int16_t) (w2 & 0xffffu);
  p.cy1 = (int16_t) (w2 >> 16);
  p.color = q.w[3];
  p.a = (int) q.w[4];
  p.b = (int) q.w[5];
  p.c = (int) q.w[6];
  p.d = (int) q.w[7];
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

// The blend of a whole primitive is chosen ONCE (kernel: BLEND_STRIPS), not per
// strip: the scalar unit, one per CU, is as scarce here as the vector units.
//   kBlendOpaque  every LERP, and PIXMAN with an opaque colour: both are
//                 s * m + t * (255 - m) with each product rounded on its own;
//                 only the rounding constant differs (0x7f cairo / 0x80 pixman),
//                 passed in `rnd` (alpha (s IN m) == m exactly when alpha (s) == 255);
//   kBlendBlack   the same for opaque black (the usual outline): s * m is
//                 {0, 0, 0, m} without a multiplication;
//   kBlendPixman  PIXMAN with a translucent colour: t * (255 - alpha (s IN m)).
enum { kBlendOpaque = 0, kBlendBlack = 1, kBlendPixman = 2 };

__device__ __forceinline__ u16x2 mul_rnd (u16x2 x, u16x2 a, u16x2 rnd)
{
  u16x2 t = x * a + rnd;
  return (t + (t >> 8)) >> 8;
}

template <int V, int K>
__device__ __forceinline__ void px_blend (Px &t, Px s, uint32_t w, u16x2 rnd)
{
  if (V == kBlendPixman) {
    px_pixman<false, K> (t, s, w);
    return;
  }
  const u16x2 mm = splat_byte<K> (w);
  const u16x2 im = as_u16x2 (as_u32 (mm) ^ 0x00ff00ffu);
  if (V == kBlendBlack) {
    t.rb = mul_rnd (t.rb, im, rnd);
    t.ag = as_u16x2 (as_u32 (mm) & 0xffff0000u) + mul_rnd (t.ag, im, rnd);
  } else {
    t.rb = mul_rnd (s.rb, mm, rnd) + mul_rnd (t.rb, im, rnd);
    t.ag = mul_rnd (s.ag, mm, rnd) + mul_rnd (t.ag, im, rnd);
  }
}

template <int V>
__device__ __forceinline__ void strip_blend_v (uint4 &v, Px s, uint32_t w, u16x2 rnd)
{
#if FLUTL_EXP == 1   // ablation: no blend arithmetic
  v.x ^= w;
  return;
#endif
  Px t[4] = { unpack (v.x), unpack (v.y), unpack (v.z), unpack (v.w) };
  px_blend<V, 0> (t[0], s, w, rnd);
  px_blend<V, 1> (t[1], s, w, rnd);
  px_blend<V, 2> (t[2], s, w, rnd);
  px_blend<V, 3> (t[3], s, w, rnd);
  v = make_uint4 (pack (t[0]), pack (t[1]), pack (t[2]), pack (t[3]));
}

// The A8 coverage bytes of a MASK primitive for this lane's 4 pixels in each
// of its kRG strips (rows y0 + 4r).  Issued one primitive ahead of use: OVER
// is order dependent, the loads are not.  Select free: lanes that the prim
// does not touch load the first dword of the atlas, which is zero by contract
// (include/flutl.h: leading slack), so nothing depends on the loaded value
// until the blend -- a select on the value would make the compiler wait for
// each load where it is issued.  Other ops load nothing.
// Bytes left/right of the mask read its zero padding (atlas contract).
typedef uint32_t u32x4 __attribute__ ((ext_vector_type (4)));

struct MaskFetch {
  uint32_t w[kRG];
};
typedef uint32_t __attribute__ ((aligned (1))) u32_unaligned;

__device__ __forceinline__ MaskFetch mask_fetch4 (const PrimU &p,
    const uint8_t *__restrict__ atlas, int x, int y0)
{
  MaskFetch f;
#pragma unroll
  for (int r = 0; r < kRG; r++)
    f.w[r] = 0u;
  if (p.op != FLUTL_OP_MASK)    // wave uniform
    return f;
  const uint32_t rows = (uint32_t) (p.cy1 - p.cy0);
  const bool colin = (uint32_t) (x + 3 - p.cx0) < (uint32_t) (p.cx1 - p.cx0 + 3);
  // |rows| and strides are < 2^15: 24-bit multiply (full rate).  SIGNED: strip
  // 0 may lie above the mask (y0 < p.d) while a later strip of the same lane
  // is valid, and off0 + r * step must then still be exact modulo 2^32.
  const uint32_t off0 = (uint32_t) __mul24 (y0 - p.d, p.b) +
      (uint32_t) (p.a - p.c) + (uint32_t) x;
  const uint32_t step = 4u * (uint32_t) p.b;
#pragma unroll
  for (int r = 0; r < kRG; r++) {
    const bool valid = colin & ((uint32_t) (y0 + 4 * r - p.cy0) < rows);
    uint32_t off = valid ? off0 + (uint32_t) r * step : 0u;
    // keep the select on the 32-bit offset (not on a 64-bit pointer): the load
    // then takes the atlas base from SGPRs and this VGPR as its offset
    asm volatile ("" : "+v"(off));
#if FLUTL_EXP == 3   // ablation: no mask loads
    f.w[r] = off;
#else
    f.w[r] = *reinterpret_cast<const u32_unaligned *> (atlas + off);
#endif
  }
  return f;
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

// Bit r set for the strips r of the supertile with
// floor (lo / 4) <= r < floor (hi / 4), rows relative to the supertile.
// (Touch test: lo = first row, hi = last row + 4.  Cover test: lo = first
// row + 3, hi = one past the last row.)  Wave uniform, scalar unit.
__device__ __forceinline__ int clamp_scalar (int v, int hi)
{
  // two scalar ops; written as min (max ()) it becomes v_med3_i32 +
  // v_readfirstlane, i.e. vector-unit work for a wave-uniform value
  int a = max (v, 0);
  asm volatile ("" : "+s"(a));
  return min (a, hi);
}

__device__ __forceinline__ uint32_t strip_range (int lo, int hi, int nmax)
{
  const int a = clamp_scalar (lo >> 2, nmax), b = clamp_scalar (hi >> 2, nmax);
  return b > a ? ((1u << b) - 1u) & ~((1u << a) - 1u) : 0u;
}

// Parking of frame pixels in the destination frame during an opacity group
// (see the kernel).  whole = the supertile is inside the frame and the rows
// are 16-byte aligned (wave uniform); else nx / ny = pixels left to the frame
// edge from this lane's first pixel: pixels outside the frame are never
// stored, nor read back.  The read-back is by the lane that stored, at the
// same address, later in program order: no fence is needed for that.
__device__ __forceinline__ void park_store (uint8_t *o, uint4 v, bool whole, int nx, int ny)
{
  if (whole) {
    *reinterpret_cast<uint4 *> (o) = v;
    return;
  }
  if (ny <= 0)
    return;
  uint32_t *q = reinterpret_cast<uint32_t *> (o);
  const uint32_t e[4] = { v.x, v.y, v.z, v.w };
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (k < nx)
      q[k] = e[k];
}

__device__ __forceinline__ uint4 park_load (const uint8_t *o, bool whole, int nx, int ny)
{
  if (whole)
    return *reinterpret_cast<const uint4 *> (o);
  const uint32_t *q = reinterpret_cast<const uint32_t *> (o);
  uint32_t e[4] = { 0u, 0u, 0u, 0u };
  if (ny > 0) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (k < nx)
        e[k] = q[k];
  }
  return make_uint4 (e[0], e[1], e[2], e[3]);
}

__device__ __forceinline__ uint4 splat4 (uint32_t c)
{
  return make_uint4 (c, c, c, c);
}

// pixman arithmetic on wave-uniform colours: done once on the vector unit
// (there is no scalar packed arithmetic) and handed back to an SGPR
__device__ __forceinline__ uint32_t uniform_in (uint32_t c, u16x2 al)
{
  Px sp = unpack (c);
  sp.rb = mul_un8 (sp.rb, al);
  sp.ag = mul_un8 (sp.ag, al);
  return (uint32_t) __builtin_amdgcn_readfirstlane ((int) pack (sp));
}
__device__ __forceinline__ uint32_t uniform_over (uint32_t d, uint32_t s)
{
  Px v = unpack (d);
  over (v, unpack (s));
  return (uint32_t) __builtin_amdgcn_readfirstlane ((int) pack (v));
}

#ifndef FLUTL_DELTA_LANES
#define FLUTL_DELTA_LANES 4      // delta read-back: 64-byte segments (8: 128, 16: 256 bytes)
#endif
// At least 6 waves per SIMD (4 KB of LDS per wave allows 8): measured 5 -> 6
// waves +4 %, 6 = 7 = 8 -- and at 8 the 96-SGPR budget makes the compiler
// spill scalars into VGPR lanes, +60 VALU per wave on a VALU-bound kernel.
#ifndef FLUTL_WAVES
#define FLUTL_WAVES 6
#endif
// strips a primitive must meet in their uniform state for the coverage lookup
// table to be made (1 ... 4: within 0.5 % of each other on cfg2, which is write bound)
#ifndef FLUTL_LUT_MIN
#define FLUTL_LUT_MIN 2
#endif
// VIDEO: the finished overlay pixel is composited OVER the pixel of a video
// frame (same geometry as dst) in the store epilogue -- the "blend over the
// video buffer" of the north star, fused so that the overlay never exists in
// memory.  Separate instantiation: the plain overlay path pays nothing.
// (Tried: a second instantiation for frames that end on whole strips -- width a
// multiple of 64, height of 4 -- with no lane-divergent branch left in the
// replay loop, so that the compiler does not "structurise" its wave-uniform
// dispatch into flag juggling.  It came out with MORE scalar branches, not
// fewer, and 2 % slower; dropped.)
// ORDERED (launches of fewer than 8 frames through the element's path): supertile
// rows are taken in the order of a table made on the host, heaviest first.  A
// single frame is one round of waves; its time is the time of the LAST wave, and
// with rows dealt top to bottom the text of the bottom region starts last.
// DELTA (the element's per-buffer launch, flutl_frame_submit): dst is the HBM twin
// of a pooled host frame and is not overwritten but COMPARED with, segment by
// segment, in the store epilogue; the segments that differ go into the twin and
// straight into the page-locked host block over PCIe (`host`, device mapped).  The
// delta read-back fused into its producer: the frame is never written to a scratch
// surface and read back (66 MB of HBM traffic and a launch per 4K buffer).  What a
// translucent group parks (see below) goes to `park`, a scratch frame of the same
// geometry, since the twin must stay what it is until the comparison.
template <bool VIDEO, bool ORDERED, bool DELTA>
__attribute__ ((amdgpu_waves_per_eu (FLUTL_WAVES)))
__global__ __launch_bounds__ (kThreads) void flutl_composite_kernel (
    FrameParams fp, uint8_t *__restrict__ dst, const uint8_t *__restrict__ video,
    const uint16_t *__restrict__ row_order,
    const flutl_frame_desc *__restrict__ frames,
    const flutl_region *__restrict__ regions,
    const flutl_prim *__restrict__ prims, const uint8_t *__restrict__ atlas,
    uint8_t *__restrict__ host, uint8_t *__restrict__ park,
    unsigned long long *__restrict__ changed_segments)
{
  // The supertile's pixels live in LDS, not in registers.  Every lane only
  // ever touches its own 16-byte slots, so there is no barrier and no bank
  // conflict (ds_*_b128, lane-linear).
  // Why LDS: with the 16 pixels of a lane in VGPRs across four nested loops
  // and ~20 control-flow joins per primitive, the register allocator kept up
  // to three copies of them alive (118 VGPRs even with the blend arithmetic
  // removed, ~450 v_mov in the loop).  A blend is ~100 VALU per strip; the
  // LDS read + write it now costs is 16 LDS cycles per CU.
  // ONE surface only (4 KB, 8 waves per SIMD): between GROUP_BEGIN and
  // GROUP_END it holds the region's group surface, and the frame pixels
  // painted so far -- none, unless regions overlap -- are parked in the
  // lane's own pixels of the destination frame, which are written for good
  // at the end anyway.
  __shared__ uint4 s_surf[kRG][kThreads];
  // Blend results by coverage, for the strips a primitive meets while they are
  // still one uniform colour (the outline of a text over its background box):
  // there the result depends on the coverage byte alone, so it is computed once
  // for the 256 coverages -- 4 per lane -- and looked up per pixel (BLEND_STRIPS).
  __shared__ uint32_t s_lut[256];

  // One wavefront per workgroup: nothing is shared between waves.
  const int lane = threadIdx.x;
#if FLUTL_EXP == 20   // diagnosis (tools/wave_timeline.py): when does each wave start and end
  const unsigned long long t_wave_start = __builtin_readcyclecounter ();
#endif

  // Workgroup -> (frame, supertile).  Workgroups are handed to the 8 XCDs
  // round robin (blockIdx.x & 7 is the XCD: grid.x is a multiple of 8), and
  // -- measured, DESIGN.md 5.3 -- in lockstep: workgroup n+1 is not placed
  // before workgroup n, so the 8 workgroups of a round should cost the same.
  //  * batches: XCD k renders whole frames (xcd_frames of them); frames of
  //    a batch are states of one stream with the same layout, so the 8
  //    workgroups of a round are the same supertile of 8 sibling frames;
  //  * fewer than 8 frames: XCD k renders the supertile rows k, k + 8, ... of
  //    every frame, so a round is one column of 8 adjacent rows (a contiguous
  //    band per XCD would leave 6 XCDs waiting on the 2 that hold the text of
  //    a typical subtitle frame); a glyph mask is then fetched into the L2 of
  //    each XCD whose rows it crosses.
  // Speed only; any placement is correct.  No run-time division: the launch
  // code chooses the grid so that these are multiply-adds.
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t i = blockIdx.x >> 3;
  const uint32_t g = fp.magic_stx ? __umulhi (i, fp.magic_stx) : i;   // i / st_x
  const uint32_t stx = i - g * fp.st_x;
  uint32_t sty = g * fp.row_mul + xcd * fp.row_add;      // within the row window
  if (fp.row_base + sty >= fp.row_end)
    return;
  if (ORDERED)
    sty = row_order[sty];       // (wave uniform: a scalar load)
  sty += fp.row_base;
  const uint32_t frame = fp.frame_first + xcd * fp.xcd_frames + blockIdx.y;

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

  // Wave-uniform bookkeeping, packed into ONE scalar register `st` (the kernel
  // sits at the 102-SGPR limit of the architecture; a spilled scalar costs a
  // v_writelane in every wave):
  //   bits  0.. 7  "touched":  strip r of the current surface holds pixels in LDS;
  //   bits  8..15  "solid":    strip r is one uniform colour `base` all over
  //                (never together with touched): a background box that covers
  //                the strip.  Such a strip costs scalar bookkeeping only -- no
  //                LDS write, no LDS read, and a box under an opacity group is
  //                faded once per supertile, not per pixel -- until something
  //                with coverage lands on it;
  //   bits 16..23  "touched_frame": strips of the FRAME that held pixels when
  //                the current opacity group began;
  //   bits 24..30  "solid_frame": strips of the FRAME that were uniform then
  //                (colour: base_frame);
  //   bit  31      the whole supertile is inside the frame and its rows are
  //                16-byte aligned ("interior").
  // A strip that is neither touched nor solid is transparent and is neither
  // initialised nor read.
  uint32_t base = 0, base_frame = 0;
  uint32_t st = (sx0 + kSuperW <= (int) fp.width && sy0 + kSuperH <= (int) fp.height &&
      ((reinterpret_cast<uintptr_t> (dst) | (VIDEO ? reinterpret_cast<uintptr_t> (video) : 0) |
          (DELTA ? reinterpret_cast<uintptr_t> (host) | reinterpret_cast<uintptr_t> (park) : 0) |
          dst_off | fp.stride) & 15u) == 0) ? kStInterior : 0u;   // scalar

#define ST_TOUCHED      (st & 0xffu)
#define ST_SOLID        ((st >> 8) & 0xffu)
#define ST_T(r)         ((st >> (r)) & 1u)
#define ST_S(r)         ((st >> (8 + (r))) & 1u)
#define ST_INTERIOR     ((st & kStInterior) != 0u)
#define STRIP_UNIFORM(r) (ST_S (r) ? base : 0u)        /* scalar */
#define STRIP_LOAD(r)  (ST_T (r) ? s_surf[r][lane] : splat4 (STRIP_UNIFORM (r)))
#define STRIP_STORE(r, v) do { s_surf[r][lane] = (v); \
    st = (st | (1u << (r))) & ~(0x100u << (r)); } while (0)
  // the strips of mask m become the uniform colour c (one `base` per supertile:
  // a second colour, e.g. where two regions meet inside it, goes to LDS)
  // (inside an opacity group, a strip whose LDS slots the group has written to
  // while the frame's pixels of that strip are parked stays in LDS: GROUP_END
  // tells by the touched bit whether the slots still hold the frame's pixels)
#define STRIPS_SET_UNIFORM(m, c) do { \
    const uint32_t c_ = (c); \
    uint32_t m_ = (m); \
    const uint32_t keep_ = m_ & st & (st >> 16) & 0xffu; \
    if (keep_) { \
      _Pragma ("unroll") for (int r_ = 0; r_ < kRG; r_++) \
        if ((keep_ >> r_) & 1u) s_surf[r_][lane] = splat4 (c_); \
      m_ &= ~keep_; \
    } \
    if ((ST_SOLID & ~m_) == 0u || base == c_) { \
      base = c_; st = (st | (m_ << 8)) & ~m_; \
    } else { \
      _Pragma ("unroll") for (int r_ = 0; r_ < kRG; r_++) \
        if ((m_ >> r_) & 1u) STRIP_STORE (r_, splat4 (c_)); \
    } } while (0)

  // this lane's pixels of strip r in the destination frame (x, y0 = the lane's
  // first pixel of strip 0; recomputed where needed, not kept in registers)
#define STRIP_OUT(r, x, y0) \
  (dst + dst_off + (size_t) ((y0) + 4 * (r)) * fp.stride + (size_t) (x) * 4)
  // ... and where a group parks them
#define PARK_OUT(r, x, y0) \
  ((DELTA ? park : dst) + dst_off + (size_t) ((y0) + 4 * (r)) * fp.stride + (size_t) (x) * 4)

#if FLUTL_EXP == 2    // ablation: store floor, no display list at all
  rcount = 0;
#endif
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
#if FLUTL_EXP == 4    // ablation: region cull only
      base ^= pfirst + pcount;
      continue;
#endif

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
#if FLUTL_EXP == 5    // ablation: prim chunk load + bbox test only
        base ^= (uint32_t) hits ^ rdl (hi.x, 0);
        continue;
#endif

#if FLUTL_EXP != 11   // (ablation 11: without this short cut)
        // An opacity group of which only its background box meets this supertile, and covers
        // it: GROUP_BEGIN, RECT, GROUP_END are then one colour -- the box's, IN the group's
        // alpha -- OVER a frame that is itself one colour or empty here: the three primitives'
        // worth of decoding and dispatch (some 400 scalar + vector instructions) become 60.
        // (The 8K stress frames are 16 overlapping translucent regions: most supertiles of a
        // region hold no text.)  Anything else -- pixels in LDS, a box edge, a strip of another
        // colour -- takes the general path below; the arithmetic is the same.
        if (__builtin_popcountll (hits) == 3 && ST_TOUCHED == 0u &&
            (ST_SOLID == 0u || ST_SOLID == (1u << kRG) - 1u)) {
          const int l0 = __builtin_ctzll (hits), l2 = 63 - __builtin_clzll (hits);
          const int l1 = __builtin_ctzll (hits & (hits - 1));
          const uint32_t w0b = rdl (lo.x, l0), w0r = rdl (lo.x, l1), w0e = rdl (lo.x, l2);
          if ((w0b & 0xffu) == FLUTL_OP_GROUP_BEGIN && (w0r & 0xffu) == FLUTL_OP_RECT &&
              (w0e & 0xffu) == FLUTL_OP_GROUP_END) {
            const uint32_t r1 = rdl (lo.y, l1), r2 = rdl (lo.z, l1);
            const int ra = (int) rdl (hi.x, l1), rb = (int) rdl (hi.y, l1);
            const int rc = (int) rdl (hi.z, l1), rd = (int) rdl (hi.w, l1);
            const uint32_t e1 = rdl (lo.y, l2), e2 = rdl (lo.z, l2);
            const int lox = max ((ra + 255) >> 8, (int) (int16_t) (r1 & 0xffffu));
            const int loy = max ((rb + 255) >> 8, (int) (int16_t) (r1 >> 16));
            const int hix = min (rc >> 8, (int) (int16_t) (r2 & 0xffffu));
            const int hiy = min (rd >> 8, (int) (int16_t) (r2 >> 16));
            const bool rfull = lox <= sx0 && loy <= sy0 && hix >= sx0 + kSuperW && hiy >= sy0 + kSuperH;
            const bool gfull = (int) (int16_t) (e1 & 0xffffu) <= sx0 && (int) (int16_t) (e1 >> 16) <= sy0 &&
                (int) (int16_t) (e2 & 0xffffu) >= sx0 + kSuperW && (int) (int16_t) (e2 >> 16) >= sy0 + kSuperH;
            if (rfull && gfull) {
              const uint32_t cin = uniform_in (rdl (lo.w, l1), splat (w0e >> 16));
              if (ST_SOLID) {
                base = uniform_over (base, cin);
              } else {
                base = cin;
                st |= ((1u << kRG) - 1u) << 8;
              }
              continue;
            }
          }
        }
#endif
        // replay the hits in paint order (= lane order); the next hit's mask
        // bytes are requested before the current one is blended
        PrimRaw nxt = prim_raw_from_lane (lo, hi, __builtin_ctzll (hits));
        hits &= hits - 1;
        MaskFetch fnxt = mask_fetch4 (prim_decode (nxt), atlas, x_lane, y_lane);
        for (;;) {
          // Re-materialise the lane coordinates per primitive: left visible,
          // LICM hoists dozens of x+k / y+4r derived VGPRs out of the loops.
          int x = x_lane, y0 = y_lane;
          asm volatile ("" : "+v"(x), "+v"(y0));
          const PrimU p = prim_decode (nxt);
          const MaskFetch f = fnxt;
          const bool more = hits != 0;
          if (more) {
            nxt = prim_raw_from_lane (lo, hi, __builtin_ctzll (hits));
            hits &= hits - 1;
            fnxt = mask_fetch4 (prim_decode (nxt), atlas, x, y0);
          }
#if FLUTL_EXP == 6    // ablation: replay loop without the op bodies
          base ^= p.color ^ rdl (f.w[0] ^ f.w[1] ^ f.w[2] ^ f.w[3], 0);
          if (!more)
            break;
          continue;
#endif

#if FLUTL_EXP == 7      // ablations: one kind of op body removed
          const bool skip_op = p.op == FLUTL_OP_GROUP_END || p.op == FLUTL_OP_IMAGE;
#elif FLUTL_EXP == 8
          const bool skip_op = p.op == FLUTL_OP_RECT;
#elif FLUTL_EXP == 9
          const bool skip_op = p.op == FLUTL_OP_MASK;
#else
          const bool skip_op = false;
#endif
          if (skip_op) {
            base ^= (p.color ^ rdl (f.w[0] ^ f.w[1] ^ f.w[2] ^ f.w[3], 0)) & 1u;
          } else if (p.op == FLUTL_OP_MASK || p.op == FLUTL_OP_RECT) {
            // ---- solid colour through 8-bit coverage: an A8 mask, or the
            //      24.8 box of cairo's rectangular scan converter -------------
            // strips the clipped bbox touches: a bit mask made once per
            // primitive with scalar ops
            uint32_t smask;
            uint32_t w[kRG];
            if (p.op == FLUTL_OP_RECT) {
              // strips the box covers completely (coverage 255 all over); the
              // usual supertile lies wholly inside a background box
              uint32_t fmask = 0u;
              if (((p.a + 255) >> 8) <= sx0 && (p.c >> 8) >= sx0 + kSuperW &&
                  p.cx0 <= sx0 && p.cx1 >= sx0 + kSuperW) {
                const int fy0 = max ((p.b + 255) >> 8, p.cy0), fy1 = min (p.d >> 8, p.cy1);
                fmask = (fy0 <= sy0 && fy1 >= sy0 + kSuperH) ? (1u << kRG) - 1u
                    : strip_range (fy0 - sy0 + 3, fy1 - sy0, kRG);
              }
              smask = fmask == (1u << kRG) - 1u ? 0u
                  : strip_range (p.cy0 - sy0, p.cy1 - sy0 + 3, kRG);
              if (fmask) {
                // t = s OVER t with a uniform s.  Opaque s, LERP, or nothing
                // painted yet: t = s exactly.
                const bool replace = (p.color >> 24) == 255u || p.rule == FLUTL_RULE_LERP;
                const uint32_t plain = replace ? fmask : fmask & ~(ST_TOUCHED | ST_SOLID);
                const uint32_t tinted = fmask & ~plain & ST_SOLID;   // uniform under a translucent box
                const uint32_t under = fmask & ~plain & ST_TOUCHED;  // pixels under a translucent box
                if (tinted) {
                  const uint32_t c = uniform_over (base, p.color);
                  if ((ST_SOLID & ~tinted) == 0u) {
                    base = c;
                  } else {
#pragma unroll
                    for (int r = 0; r < kRG; r++)
                      if ((tinted >> r) & 1u)
                        STRIP_STORE (r, splat4 (c));
                  }
                }
                if (plain)
                  STRIPS_SET_UNIFORM (plain, p.color);
                if (under) {
                  const Px sc = unpack (p.color);
#pragma unroll
                  for (int r = 0; r < kRG; r++) {
                    if (((under >> r) & 1u) == 0u)
                      continue;
                    const uint4 d = s_surf[r][lane];
                    Px v0 = unpack (d.x), v1 = unpack (d.y), v2 = unpack (d.z), v3 = unpack (d.w);
                    over (v0, sc);
                    over (v1, sc);
                    over (v2, sc);
                    over (v3, sc);
                    s_surf[r][lane] = make_uint4 (pack (v0), pack (v1), pack (v2), pack (v3));
                  }
                }
                smask &= ~fmask;
              }
              // what is left are the edges of the box: coverage from cairo's
              // rectangular scan converter
              if (smask) {
#pragma unroll
                for (int r = 0; r < kRG; r++) {
                  w[r] = 0u;
                  if (((smask >> r) & 1u) == 0u)
                    continue;
                  const int y = y0 + 4 * r;
                  const bool rowin = (uint32_t) (y - p.cy0) < (uint32_t) (p.cy1 - p.cy0);
                  const int oy = overlap256 (p.b, p.d, y);
                  uint32_t c = 0;
#pragma unroll
                  for (int k = 0; k < 4; k++)
                    c |= min ((uint32_t) (overlap256 (p.a, p.c, x + k) * oy) >> 8, 255u) << (8 * k);
                  w[r] = c & pixel_mask (p, x, rowin);
                }
              }
            } else {
              smask = strip_range (p.cy0 - sy0, p.cy1 - sy0 + 3, kRG);
#pragma unroll
              for (int r = 0; r < kRG; r++)
                w[r] = f.w[r];
              if (p.param & FLUTL_MASK_XCLIP) {
                const uint32_t pm = pixel_mask (p, x, true);
#pragma unroll
                for (int r = 0; r < kRG; r++)
                  w[r] &= pm;
              }
            }
            if (smask) {
              const Px sc = unpack (p.color);
              const u16x2 rnd = as_u16x2 (p.rule == FLUTL_RULE_LERP ? 0x007f007fu : 0x00800080u);
              // strips where coverage lands at all, and those of them that are
              // still uniform
              uint32_t cmask = 0u;
#pragma unroll
              for (int r = 0; r < kRG; r++)
                if (((smask >> r) & 1u) && __ballot (w[r] != 0) != 0)
                  cmask |= 1u << r;
              uint32_t lmask = cmask & ST_SOLID;
#if FLUTL_EXP == 10      // ablation: no lookup table
              lmask = 0u;
#endif
              if (__builtin_popcount (lmask) < FLUTL_LUT_MIN)
                lmask = 0u;           // one strip: the table would cost more than it saves
              cmask &= ~lmask;
#define BLEND_STRIPS(V) \
              if (lmask) { \
                _Pragma ("unroll") for (int j = 0; j < 4; j++) { \
                  uint32_t m = (uint32_t) lane + 64u * (uint32_t) j; \
                  asm volatile ("" : "+v"(m));    /* or everything derived from it is hoisted to the kernel's top */ \
                  Px t = unpack (base); \
                  px_blend<V, 0> (t, sc, m, rnd); \
                  s_lut[m] = pack (t); \
                } \
                _Pragma ("unroll") for (int r = 0; r < kRG; r++) { \
                  if (((lmask >> r) & 1u) == 0u) \
                    continue; \
                  const uint32_t ww = w[r]; \
                  STRIP_STORE (r, make_uint4 (s_lut[ww & 0xffu], s_lut[(ww >> 8) & 0xffu], \
                      s_lut[(ww >> 16) & 0xffu], s_lut[ww >> 24])); \
                } \
              } \
              _Pragma ("unroll") for (int r = 0; r < kRG; r++) { \
                if (((cmask >> r) & 1u) == 0u) \
                  continue; \
                uint4 v = STRIP_LOAD (r); \
                strip_blend_v<V> (v, sc, w[r], rnd); \
                STRIP_STORE (r, v); \
              }
              if (p.color == 0xff000000u) {
                BLEND_STRIPS (kBlendBlack)
              } else if ((p.color >> 24) == 255u || p.rule == FLUTL_RULE_LERP) {
                BLEND_STRIPS (kBlendOpaque)
              } else {
                BLEND_STRIPS (kBlendPixman)
              }
#undef BLEND_STRIPS
            }
          } else if (p.op == FLUTL_OP_GROUP_BEGIN) {
            // the region's temporary surface starts transparent; what the
            // frame holds here so far (overlapping regions only) is parked
#pragma unroll
            for (int r = 0; r < kRG; r++)
              if (ST_T (r))
                park_store (PARK_OUT (r, x, y0), s_surf[r][lane], ST_INTERIOR,
                    (int) fp.width - x, (int) fp.height - (y0 + 4 * r));
            // touched_frame = touched, solid_frame = solid, nothing in the group yet
            base_frame = base;
            st = (st & kStInterior) | (ST_TOUCHED << 16) | (ST_SOLID << 24);
          } else if (p.op == FLUTL_OP_GROUP_END) {
            // ---- (group surface IN alpha) OVER frame ---------------------------
            // back to the frame; the group surface becomes the source.  A strip
            // the group did not write to LDS still holds its frame pixels there,
            // and a uniform frame strip is uniform still.
            const uint32_t sg = st;     // the group's touched / solid, the frame's of before
            const uint32_t base_group = base;
            const uint32_t lds_frame = (sg >> 16) & 0xffu, solid_frame = (sg >> 24) & 0x7fu;
            st = (sg & kStInterior) | (lds_frame & ~sg & 0xffu) | (solid_frame << 8);
            base = base_frame;
            const u16x2 al = splat (p.param);
            // uniform group strips, wholly inside the group's box, over frame
            // strips that are uniform or empty: scalar bookkeeping (all of a
            // supertile that only background boxes have met, however many
            // regions overlap there)
            const bool xfull = p.cx0 <= sx0 && p.cx1 >= sx0 + kSuperW;
            const uint32_t inside = xfull ? strip_range (p.cy0 - sy0 + 3, p.cy1 - sy0, kRG) : 0u;
            const uint32_t um = (sg >> 8) & inside & ~lds_frame;
            if (um) {
              const uint32_t cin = uniform_in (base_group, al);
              const uint32_t um_over = um & solid_frame;
              if (um_over) {
                const uint32_t cover = uniform_over (base_frame, cin);
                if ((ST_SOLID & ~um_over) == 0u) {
                  base = cover;
                } else {
#pragma unroll
                  for (int r = 0; r < kRG; r++)
                    if ((um_over >> r) & 1u)
                      STRIP_STORE (r, splat4 (cover));
                }
              }
              if (um & ~solid_frame)
                STRIPS_SET_UNIFORM (um & ~solid_frame, cin);
            }
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              if (((((sg | (sg >> 8)) & ~um) >> r) & 1u) == 0u)
                continue;
              const bool src_solid = (sg >> (8 + r)) & 1u;
              const bool frame_lds = (lds_frame >> r) & 1u;
              const bool frame_solid = (solid_frame >> r) & 1u;
              const int y = y0 + 4 * r;
              const uint4 g = src_solid ? splat4 (base_group) : s_surf[r][lane];
              uint32_t sv[4] = { g.x, g.y, g.z, g.w };
              if (((inside >> r) & 1u) == 0u) {   // cut by the group's box
                const bool rowin = (uint32_t) (y - p.cy0) < (uint32_t) (p.cy1 - p.cy0);
                const uint32_t pmask = pixel_mask (p, x, rowin);
                sv[0] = (pmask & 0x1u) ? g.x : 0u;
                sv[1] = (pmask & 0x100u) ? g.y : 0u;
                sv[2] = (pmask & 0x10000u) ? g.z : 0u;
                sv[3] = (pmask & 0x1000000u) ? g.w : 0u;
              }
              uint4 d;
              if (frame_lds) {
                if (src_solid) {
                  d = s_surf[r][lane];          // never overwritten by the group
                } else {
                  d = park_load (PARK_OUT (r, x, y0), ST_INTERIOR,
                      (int) fp.width - x, (int) fp.height - y);
                  STRIP_STORE (r, d);           // the frame strip is back
                }
              } else {
                d = splat4 (frame_solid ? base_frame : 0u);
              }
              if (__ballot ((sv[0] | sv[1] | sv[2] | sv[3]) != 0) == 0)
                continue;   // 0 OVER d == d
              if (!frame_lds && !frame_solid) {
                // nothing under the group here (regions that do not overlap,
                // the usual case): x OVER 0 == x, only the IN remains
                uint32_t o[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                  Px sp = unpack (sv[k]);
                  sp.rb = mul_un8 (sp.rb, al);
                  sp.ag = mul_un8 (sp.ag, al);
                  o[k] = pack (sp);
                }
                STRIP_STORE (r, make_uint4 (o[0], o[1], o[2], o[3]));
                continue;
              }
              uint32_t dv[4] = { d.x, d.y, d.z, d.w };
#pragma unroll
              for (int k = 0; k < 4; k++) {
                Px sp = unpack (sv[k]);
                sp.rb = mul_un8 (sp.rb, al);
                sp.ag = mul_un8 (sp.ag, al);
                Px v = unpack (dv[k]);
                over (v, sp);
                dv[k] = pack (v);
              }
              STRIP_STORE (r, make_uint4 (dv[0], dv[1], dv[2], dv[3]));
            }
          } else {
            // ---- IMAGE: per-pixel premultiplied source, OVER --------------------
#pragma unroll
            for (int r = 0; r < kRG; r++) {
              const int ry0 = sy0 + 4 * r;
              if (p.cy1 <= ry0 || p.cy0 >= ry0 + 4)
                continue;
              const int y = y0 + 4 * r;
              const bool rowin = (uint32_t) (y - p.cy0) < (uint32_t) (p.cy1 - p.cy0);
              const uint32_t pmask = pixel_mask (p, x, rowin);
              uint32_t sv[4];
              const uint32_t rowoff = (uint32_t) p.a +
                  (uint32_t) (y - p.d) * (uint32_t) p.b + (uint32_t) (x - p.c) * 4u;
              // select free, like the mask bytes: pixels outside the image read
              // the leading zeros of the atlas (contract: 16 of them)
#pragma unroll
              for (int k = 0; k < 4; k++) {
                uint32_t off = ((pmask >> (8 * k)) & 1u) ? rowoff + 4u * (uint32_t) k : 0u;
                asm volatile ("" : "+v"(off));
                sv[k] = *reinterpret_cast<const uint32_t *> (atlas + off);
              }
              if (__ballot ((sv[0] | sv[1] | sv[2] | sv[3]) != 0) == 0)
                continue;   // 0 OVER d == d
              const uint4 d = STRIP_LOAD (r);
              uint32_t dv[4] = { d.x, d.y, d.z, d.w };
#pragma unroll
              for (int k = 0; k < 4; k++) {
                Px v = unpack (dv[k]);
                over (v, unpack (sv[k]));
                dv[k] = pack (v);
              }
              STRIP_STORE (r, make_uint4 (dv[0], dv[1], dv[2], dv[3]));
            }
          }
          if (!more)
            break;
        }
      }
    }
  }

  // ---- store: every pixel exactly once --------------------------------------
  // (validated lists have balanced groups, so the frame is current here)
  uint8_t *const out = STRIP_OUT (0, x_lane, y_lane);
  const uint8_t *const vin = VIDEO
      ? video + dst_off + (size_t) y_lane * fp.stride + (size_t) x_lane * 4 : nullptr;
  if (DELTA) {
    // compare with the twin; a segment = kSegLanes lanes x 16 bytes of one row (a
    // row of the supertile is 16 lanes): whole segments keep the PCIe writes at
    // full payload size
    constexpr uint32_t kSegLanes = FLUTL_DELTA_LANES;
    static_assert (kSegLanes == 16 || kSegLanes == 8 || kSegLanes == 4, "segments within a row");
    uint8_t *const hout = host + (out - dst);
    uint32_t n_seg = 0;         // scalar
    if (ST_INTERIOR) {
#pragma unroll
      for (int r = 0; r < kRG; r++) {
        const uint4 v = STRIP_LOAD (r);
        const size_t o = (size_t) (4 * r) * fp.stride;
        const uint4 t = *reinterpret_cast<const uint4 *> (out + o);
        const bool diff = (v.x != t.x) | (v.y != t.y) | (v.z != t.z) | (v.w != t.w);
        const unsigned long long b = __ballot (diff);
        if (b == 0ull)
          continue;
        if (((b >> (lane & (64u - kSegLanes))) & ((1ull << kSegLanes) - 1ull)) != 0ull) {
          *reinterpret_cast<uint4 *> (out + o) = v;
          *reinterpret_cast<uint4 *> (hout + o) = v;
        }
#pragma unroll
        for (uint32_t k = 0; k < 64u; k += kSegLanes)
          n_seg += ((b >> k) & ((1ull << kSegLanes) - 1ull)) != 0ull;
      }
    } else if (x_lane < (int) fp.width) {
      // frame edge or rows that are not 16-byte multiples: pixel by pixel
#pragma unroll
      for (int r = 0; r < kRG; r++) {
        if (y_lane + 4 * r >= (int) fp.height)
          continue;
        const uint4 v = STRIP_LOAD (r);
        const uint32_t e[4] = { v.x, v.y, v.z, v.w };
        uint32_t *tw = reinterpret_cast<uint32_t *> (out + (size_t) (4 * r) * fp.stride);
        uint32_t *ho = reinterpret_cast<uint32_t *> (hout + (size_t) (4 * r) * fp.stride);
        bool any = false;
        for (int k = 0; k < 4 && x_lane + k < (int) fp.width; k++)
          if (tw[k] != e[k]) {
            tw[k] = e[k];
            ho[k] = e[k];
            any = true;
          }
        const unsigned long long b = __ballot (any);
        for (uint32_t k = 0; k < 64u; k += kSegLanes)
          n_seg += ((b >> k) & ((1ull << kSegLanes) - 1ull)) != 0ull;
      }
    }
    if (changed_segments && n_seg && lane == 0)
      atomicAdd (changed_segments, (unsigned long long) n_seg);
  } else if (ST_INTERIOR) {
#pragma unroll
    for (int r = 0; r < kRG; r++) {
      // streaming (nt) stores: the frame is written once and never read here,
      // so it should not evict glyph masks and list entries from the L2
      // (measured +6 % on the whole kernel; the store-only floor is unchanged)
      uint4 v = STRIP_LOAD (r);
      if (VIDEO) {
        const u32x4 q = __builtin_nontemporal_load (
            reinterpret_cast<const u32x4 *> (vin + (size_t) (4 * r) * fp.stride));
        if (ST_T (r) | ST_S (r)) {
          Px d0 = unpack (q.x), d1 = unpack (q.y), d2 = unpack (q.z), d3 = unpack (q.w);
          over (d0, unpack (v.x));
          over (d1, unpack (v.y));
          over (d2, unpack (v.z));
          over (d3, unpack (v.w));
          v = make_uint4 (pack (d0), pack (d1), pack (d2), pack (d3));
        } else {
          v = make_uint4 (q.x, q.y, q.z, q.w);      // nothing painted: the video pixel
        }
      }
      const u32x4 nv = { v.x, v.y, v.z, v.w };
      __builtin_nontemporal_store (nv, reinterpret_cast<u32x4 *> (out + (size_t) (4 * r) * fp.stride));
    }
  } else if (x_lane < (int) fp.width) {
    // frame edge or unaligned destination: per-pixel bounds
#pragma unroll
    for (int r = 0; r < kRG; r++) {
      if (y_lane + 4 * r >= (int) fp.height)
        continue;
      const uint4 v = STRIP_LOAD (r);
      const uint32_t e[4] = { v.x, v.y, v.z, v.w };
      for (int k = 0; k < 4 && x_lane + k < (int) fp.width; k++) {
        uint32_t o = e[k];
        if (VIDEO) {
          Px d = unpack (reinterpret_cast<const uint32_t *> (vin + (size_t) (4 * r) * fp.stride)[k]);
          over (d, unpack (o));
          o = pack (d);
        }
        reinterpret_cast<uint32_t *> (out + (size_t) (4 * r) * fp.stride)[k] = o;
      }
    }
  }
#if FLUTL_EXP == 20
  // the wave's first 16 bytes of its supertile become (start, end) clock values
  if (lane == 0 && sx0 + 4 <= (int) fp.width && sy0 < (int) fp.height) {
    __builtin_amdgcn_s_waitcnt (0);
    const unsigned long long t_wave_end = __builtin_readcyclecounter ();
    unsigned long long *q = reinterpret_cast<unsigned long long *> (
        dst + dst_off + (size_t) sy0 * fp.stride + (size_t) sx0 * 4);
    q[0] = t_wave_start;
    q[1] = t_wave_end;
  }
#endif
#undef STRIP_OUT
#undef PARK_OUT
#undef STRIP_LOAD
#undef STRIP_STORE
#undef STRIPS_SET_UNIFORM
#undef STRIP_UNIFORM
#undef ST_TOUCHED
#undef ST_SOLID
#undef ST_T
#undef ST_S
#undef ST_INTERIOR
}

// ---------------------------------------------------------------------------
// Blur: pixman PIXMAN_FILTER_CONVOLUTION + OP_SRC, identity transform,
// REPEAT_NONE (reference plugins/ttml/gstttmlblur.c:71-110).  One workgroup
// per 32x8 output tile; the (32+2r) x (8+2r) source halo and the (2r+1)^2
// 16.16 weights are staged in LDS; accumulation is exact int32:
//   acc_c += chan_c * f;  out_c = clip ((acc_c + 0x8000) >> 16, 0, 255).
constexpr int kBlurTW = 32, kBlurTH = 8;
constexpr int kBlurMaxR = 16;      // LDS-tiled kernel (<= 18 KB of LDS)
constexpr int kBlurMaxRWide = FLUTL_BLUR_MAX_RADIUS;   // flutl_blur_wide_kernel_dev: taps straight from L1/L2

__global__ __launch_bounds__ (256) void flutl_blur_kernel_dev (
    const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int w, int h,
    int stride_px, const int32_t *__restrict__ kern, int radius, int tile_row0)
{
  extern __shared__ __attribute__ ((aligned (16))) uint32_t smem[];
  const int size = 2 * radius + 1;
  const int hw = kBlurTW + 2 * radius, hh = kBlurTH + 2 * radius;
  uint32_t *s_src = smem;                       // hw * hh
  int32_t *s_k = reinterpret_cast<int32_t *> (smem + hw * hh);  // size*size
  const int tx0 = blockIdx.x * kBlurTW, ty0 = (tile_row0 + blockIdx.y) * kBlurTH;
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

// The same convolution for radii whose halo does not fit the LDS tile (blur
// lengths above 8 px, rare): one pixel per lane, taps and weights read through
// the L1/L2.  Same arithmetic, same results; O((2r+1)^2) loads per pixel.
__global__ __launch_bounds__ (256) void flutl_blur_wide_kernel_dev (
    const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int w, int h,
    int stride_px, const int32_t *__restrict__ kern, int radius, int row0)
{
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= w || y >= h)
    return;
  const int size = 2 * radius + 1;
  int32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  for (int j = 0; j < size; j++) {
    const int sy = y - radius + j;
    if (sy < 0 || sy >= h)
      continue;
    const uint32_t *row = src + (size_t) sy * stride_px;
    const int32_t *krow = kern + j * size;
    const int i0 = max (0, radius - x), i1 = min (size, w - x + radius);
    for (int i = i0; i < i1; i++) {
      const int32_t f = krow[i];
      const uint32_t p = row[x - radius + i];
      a0 += (int32_t) (p & 0xff) * f;
      a1 += (int32_t) ((p >> 8) & 0xff) * f;
      a2 += (int32_t) ((p >> 16) & 0xff) * f;
      a3 += (int32_t) (p >> 24) * f;
    }
  }
  uint32_t o0 = (uint32_t) min (max ((a0 + 0x8000) >> 16, 0), 255);
  uint32_t o1 = (uint32_t) min (max ((a1 + 0x8000) >> 16, 0), 255);
  uint32_t o2 = (uint32_t) min (max ((a2 + 0x8000) >> 16, 0), 255);
  uint32_t o3 = (uint32_t) min (max ((a3 + 0x8000) >> 16, 0), 255);
  asm volatile ("" : "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));   // see above
  dst[(size_t) y * stride_px + x] = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
}

// Bilinear resample by CONSTANT 7-bit weights: an ARGB32 surface blitted at a
// fractional device position (cairo picks PIXMAN_FILTER_BILINEAR; the transform
// is a pure translation, so every pixel has the same weights).  pixman-inlines.h
// bilinear_interpolation with BILINEAR_INTERPOLATION_BITS = 7, REPEAT_NONE:
// out (i, j) mixes src (i-1, j-1), (i, j-1), (i-1, j), (i, j); w x h -> (w+1) x (h+1).
// The reference paints the surface through a cairo_rectangle + cairo_fill of its
// extent (gstttmlrender.c:1221-1224): the box edges, fractional too, add the
// rectangular scan converter's coverage as an a8 mask (sample IN coverage,
// MUL_UN8 per channel); box = that rectangle in 24.8 relative to out (0, 0).
// HBM bound, 8 bytes per output pixel with the re-reads served by the L1/L2.
struct ResampleBox {
  int x1, y1, x2, y2;
  int enabled;
};

__global__ __launch_bounds__ (256) void flutl_resample_kernel_dev (
    const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int w, int h,
    int src_stride_px, int dst_stride_px, uint32_t distx, uint32_t disty, ResampleBox box)
{
  const int i = blockIdx.x * 64 + (threadIdx.x & 63);
  const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (i > w || j > h)
    return;
  auto at = [&] (int x, int y) -> uint32_t {
    return (x >= 0 && y >= 0 && x < w && y < h) ? src[(size_t) y * src_stride_px + x] : 0u;
  };
  const uint32_t tl = at (i - 1, j - 1), tr = at (i, j - 1), bl = at (i - 1, j), br = at (i, j);
  const uint32_t dx = distx << 1, dy = disty << 1;
  const uint32_t dxy = dx * dy, dxiy = (dx << 8) - dxy, dixy = (dy << 8) - dxy,
      dixiy = 256u * 256u - (dy << 8) - (dx << 8) + dxy;
  uint32_t r = 0;
#pragma unroll
  for (int sh = 0; sh < 32; sh += 8) {
    const uint32_t c = ((tl >> sh) & 255u) * dixiy + ((tr >> sh) & 255u) * dxiy +
        ((bl >> sh) & 255u) * dixy + ((br >> sh) & 255u) * dxy;
    r |= ((c >> 16) & 255u) << sh;
  }
  if (box.enabled) {
    const uint32_t cov = min ((uint32_t) (overlap256 (box.x1, box.x2, i) *
        overlap256 (box.y1, box.y2, j)) >> 8, 255u);
    Px v = unpack (r);
    const u16x2 cc = splat (cov);
    v.rb = mul_un8 (v.rb, cc);
    v.ag = mul_un8 (v.ag, cc);
    r = pack (v);
  }
  dst[(size_t) j * dst_stride_px + i] = r;
}

// Quarter turn clockwise of an ARGB32 image (w x h -> h x w): a surface painted
// through the 90-degree CTM of the vertical writing modes (gstttmlrender.c:
// 1313-1340) at an integer device position is sampled with the NEAREST filter,
// i.e. it is a permutation of its pixels: dst (row i, column h-1-j) = src (j, i).
// box: as in the resample kernel, the filled rectangle's edge coverage, here in
// the coordinates of the TURNED image (fractional positions: the image was
// resampled in its own space first, then turned, then clipped softly).
__global__ __launch_bounds__ (256) void flutl_rotate_cw_kernel_dev (
    const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, int w, int h,
    int src_stride_px, int dst_stride_px, ResampleBox box)
{
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);   // dst column, < h
  const int r = blockIdx.y * 4 + (threadIdx.x >> 6);    // dst row, < w
  if (c >= h || r >= w)
    return;
  uint32_t v = src[(size_t) (h - 1 - c) * src_stride_px + r];
  if (box.enabled) {
    const uint32_t cov = min ((uint32_t) (overlap256 (box.x1, box.x2, c) *
        overlap256 (box.y1, box.y2, r)) >> 8, 255u);
    Px q = unpack (v);
    const u16x2 cc = splat (cov);
    q.rb = mul_un8 (q.rb, cc);
    q.ag = mul_un8 (q.ag, cc);
    v = pack (q);
  }
  dst[(size_t) r * dst_stride_px + c] = v;
}

// ---------------------------------------------------------------------------
// Delta read-back ("dirty-rect D2H into pre-zeroed buffers", taken
// to its end).  A pooled output frame is a page-locked host block plus its twin
// in HBM; both always hold the same pixels.  After a frame has been composited
// into the context's scratch frame, this kernel compares it with the twin, 256
// bytes (16 lanes x 16 bytes of one row) at a time, and writes the segments
// that differ into the twin AND straight into the host block over PCIe (the
// block is device-mapped): a <set> that recolours one line of text moves a few
// hundred KB instead of 33 MB, with no staging copy, no size read-back and no
// host-side bookkeeping of rectangles.  Whole 256-byte segments keep the PCIe
// writes at full payload size.  HBM bound: reads 8 bytes per pixel of the rows
// it is run on.
//
// Rows [y0, y1) only: the caller passes the union of what the twin holds and
// what the new frame holds (both are transparent elsewhere).
// WIDE: rows are 16-byte aligned multiples of 16 bytes; else per-pixel chunks.
template <bool WIDE>
__global__ __launch_bounds__ (256) void flutl_delta_kernel (
    const uint8_t *__restrict__ fresh, uint8_t *__restrict__ twin,
    uint8_t *__restrict__ host, uint32_t row_bytes, uint32_t chunks_per_row,
    uint32_t y0, uint32_t n_rows, uint32_t fresh_y0, uint32_t fresh_y1,
    unsigned long long *__restrict__ changed_segments)
{
  // thread -> (row, 16-byte chunk); a wave covers 4 segments of 16 chunks, which
  // may lie in two rows (chunks_per_row is padded to a multiple of 16 so that a
  // segment never straddles rows)
  const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
  const uint32_t row = idx / chunks_per_row, c = idx - row * chunks_per_row;
  const bool live = row < n_rows && c * 16u < row_bytes;
  const size_t off = (size_t) (y0 + row) * row_bytes + (size_t) c * 16u;
  // rows outside [fresh_y0, fresh_y1) were not composited this time: the new
  // frame is transparent there, whatever `fresh` still holds
  const bool drawn = y0 + row >= fresh_y0 && y0 + row < fresh_y1;
  uint4 v = make_uint4 (0, 0, 0, 0), t = v;
  uint32_t n_dw = 0;              // dwords of this chunk that exist (WIDE: 4)
  if (live) {
    if (WIDE) {
      n_dw = 4;
      if (drawn)
        v = *reinterpret_cast<const uint4 *> (fresh + off);
      t = *reinterpret_cast<const uint4 *> (twin + off);
    } else {
      n_dw = min (4u, (row_bytes - c * 16u) / 4u);
      const uint32_t *fv = reinterpret_cast<const uint32_t *> (fresh + off);
      const uint32_t *tv = reinterpret_cast<const uint32_t *> (twin + off);
      uint32_t a[4] = { 0, 0, 0, 0 }, b[4] = { 0, 0, 0, 0 };
      for (uint32_t k = 0; k < n_dw; k++) {
        a[k] = drawn ? fv[k] : 0u;
        b[k] = tv[k];
      }
      v = make_uint4 (a[0], a[1], a[2], a[3]);
      t = make_uint4 (b[0], b[1], b[2], b[3]);
    }
  }
  const bool diff = (v.x != t.x) | (v.y != t.y) | (v.z != t.z) | (v.w != t.w);
  const unsigned long long b = __ballot (diff);
  const uint32_t lane = threadIdx.x & 63u;
  // my segment (FLUTL_DELTA_LANES lanes x 16 bytes) changed
  constexpr uint32_t kSegLanes = FLUTL_DELTA_LANES;
  const bool seg = ((b >> (lane & (64u - kSegLanes))) & ((1ull << kSegLanes) - 1ull)) != 0ull;
  if (seg && live) {
    if (WIDE) {
      *reinterpret_cast<uint4 *> (twin + off) = v;
      *reinterpret_cast<uint4 *> (host + off) = v;
    } else {
      const uint32_t e[4] = { v.x, v.y, v.z, v.w };
      for (uint32_t k = 0; k < n_dw; k++) {
        reinterpret_cast<uint32_t *> (twin + off)[k] = e[k];
        reinterpret_cast<uint32_t *> (host + off)[k] = e[k];
      }
    }
  }
  if (changed_segments && b != 0ull && lane == 0u) {     // one atomic per wave
    uint32_t n = 0;
#pragma unroll
    for (uint32_t k = 0; k < 64u; k += kSegLanes)
      n += ((b >> k) & ((1ull << kSegLanes) - 1ull)) != 0ull;
    atomicAdd (changed_segments, (unsigned long long) n);
  }
}

int launch_tiles (hipStream_t stream, uint32_t width, uint32_t height, void *dst,
    const void *video, size_t stride, const flutl_frame_desc *frames, uint32_t n_frames,
    uint32_t region_first, uint32_t region_count, const flutl_region *regions,
    const flutl_prim *prims, const uint8_t *atlas, uint32_t row0 = 0, uint32_t row1 = 0,
    const uint16_t *row_order = nullptr, void *delta_host = nullptr, void *delta_park = nullptr,
    unsigned long long *delta_segments = nullptr)
{
  if (!dst || width == 0 || height == 0 || n_frames == 0 || width > 32767 ||
      height > 32767 || (stride & 3) || stride < (size_t) width * 4)
    return FLUTL_EINVAL;
  FrameParams fp;
  fp.width = width;
  fp.height = height;
  fp.st_x = (width + kSuperW - 1) / kSuperW;
  fp.st_y = (height + kSuperH - 1) / kSuperH;
  // rows [row0, row1) only (row1 == 0: all), widened to whole supertiles
  fp.row_base = row1 ? row0 / kSuperH : 0u;
  fp.row_end = row1 ? (min (row1, height) + kSuperH - 1) / kSuperH : fp.st_y;
  if (fp.row_end <= fp.row_base)
    return FLUTL_OK;
  const uint32_t n_rows = fp.row_end - fp.row_base;
  if (row_order && (n_frames != 1 || video))
    return FLUTL_EINVAL;        // one table, one frame
  if (delta_host && (!row_order || !delta_park))
    return FLUTL_EINVAL;        // the fused delta read-back: single ordered frames
  fp.n_frames = n_frames;
  fp.region_first = region_first;
  fp.region_count = region_count;
  if (n_frames > 8u * 65535u)
    return FLUTL_EINVAL;
  const uint32_t per_frame = fp.st_x * n_rows;      // <= 512 * 2048
  fp.magic_stx = fp.st_x > 1u
      ? (uint32_t) ((0x100000000ull + fp.st_x - 1u) / fp.st_x) : 0u;
  fp.stride = stride;
  // (every launch is checked on its own, so that a refused launch is reported for
  // the part of the batch it belongs to; a fault inside a kernel is asynchronous
  // and surfaces at the caller's next synchronisation)
  hipError_t launch_error = hipSuccess;
  // (tried for single frames, tools/b1_variants.sh: capping the resident waves
  // with extra LDS per workgroup so that rounds overlap their store and compute
  // phases -- 14.4 us -> 15.5 ... 24.8; wave priorities by dispatch order: no change)
#define LAUNCH(grid) do { \
    if (launch_error != hipSuccess) \
      break; \
    if (video) \
      hipLaunchKernelGGL ((flutl_composite_kernel<true, false, false>), grid, dim3 (kThreads), 0, stream, fp, \
          static_cast<uint8_t *> (dst), static_cast<const uint8_t *> (video), \
          static_cast<const uint16_t *> (nullptr), frames, regions, prims, atlas, \
          static_cast<uint8_t *> (nullptr), static_cast<uint8_t *> (nullptr), \
          static_cast<unsigned long long *> (nullptr)); \
    else if (delta_host) \
      hipLaunchKernelGGL ((flutl_composite_kernel<false, true, true>), grid, dim3 (kThreads), 0, stream, fp, \
          static_cast<uint8_t *> (dst), static_cast<const uint8_t *> (nullptr), row_order, \
          frames, regions, prims, atlas, static_cast<uint8_t *> (delta_host), \
          static_cast<uint8_t *> (delta_park), delta_segments); \
    else if (row_order) \
      hipLaunchKernelGGL ((flutl_composite_kernel<false, true, false>), grid, dim3 (kThreads), 0, stream, fp, \
          static_cast<uint8_t *> (dst), static_cast<const uint8_t *> (nullptr), row_order, \
          frames, regions, prims, atlas, static_cast<uint8_t *> (nullptr), \
          static_cast<uint8_t *> (nullptr), static_cast<unsigned long long *> (nullptr)); \
    else \
      hipLaunchKernelGGL ((flutl_composite_kernel<false, false, false>), grid, dim3 (kThreads), 0, stream, fp, \
          static_cast<uint8_t *> (dst), static_cast<const uint8_t *> (nullptr), \
          static_cast<const uint16_t *> (nullptr), frames, regions, prims, atlas, \
          static_cast<uint8_t *> (nullptr), static_cast<uint8_t *> (nullptr), \
          static_cast<unsigned long long *> (nullptr)); \
    launch_error = hipGetLastError (); \
  } while (0)
  // one workgroup (= one wave) per supertile; grid.x is a multiple of 8 so
  // that blockIdx.x & 7 is the XCD of the workgroup in every row of the grid
#if defined(FLUTL_MAP) && FLUTL_MAP == 3   // experiment: rows interleaved for every batch size
  const uint32_t whole = 0, rest = n_frames;
#else
  const uint32_t whole = n_frames / 8u, rest = n_frames % 8u;
#endif
  if (whole) {          // XCD k renders frames [k * whole, (k + 1) * whole)
    fp.frame_first = 0;
    fp.xcd_frames = whole;
    fp.row_mul = 1;
    fp.row_add = 0;
    LAUNCH (dim3 (per_frame * 8u, whole));
  }
  if (rest) {           // XCD k renders rows k, k + 8, ... of each remaining frame
    fp.frame_first = whole * 8u;
    fp.xcd_frames = 0;
    fp.row_mul = 8;
    fp.row_add = 1;
    LAUNCH (dim3 ((n_rows + 7u) / 8u * fp.st_x * 8u, rest));
  }
#undef LAUNCH
  return launch_error == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}

}  // namespace

extern "C" uint32_t flutl_delta_segment_bytes (void)
{
  return FLUTL_DELTA_LANES * 16u;
}

extern "C" int flutl_launch_frame (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas)
{
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, nullptr,
      dst_stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas);
}

extern "C" int flutl_launch_frame_rows (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas, uint32_t row0,
    uint32_t row1)
{
  if (row0 > row1 || row1 > height)
    return FLUTL_EINVAL;
  if (row0 == row1)
    return FLUTL_OK;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, nullptr,
      dst_stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas, row0, row1);
}

extern "C" int flutl_launch_frame_ordered (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas, uint32_t row0,
    uint32_t row1, const uint16_t *row_order)
{
  if (row0 > row1 || row1 > height || !row_order)
    return FLUTL_EINVAL;
  if (row0 == row1)
    return FLUTL_OK;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, nullptr,
      dst_stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas, row0, row1, row_order);
}

extern "C" int flutl_launch_frame_delta (void *stream, uint32_t width, uint32_t height,
    void *twin, void *host, void *park, size_t stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas, uint32_t row0,
    uint32_t row1, const uint16_t *row_order, uint64_t *changed_segments)
{
  if (row0 > row1 || row1 > height || !row_order || !twin || !host || !park)
    return FLUTL_EINVAL;
  if (row0 == row1)
    return FLUTL_OK;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, twin, nullptr,
      stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas, row0, row1, row_order,
      host, park, reinterpret_cast<unsigned long long *> (changed_segments));
}

extern "C" int flutl_launch_frame_over_rows (void *stream, uint32_t width, uint32_t height,
    void *dst, const void *video, size_t stride_bytes, const flutl_region *regions,
    uint32_t n_regions, const flutl_prim *prims, const uint8_t *atlas, uint32_t row0,
    uint32_t row1)
{
  if (!video || video == dst || row0 > row1 || row1 > height)
    return FLUTL_EINVAL;
  if (row0 == row1)
    return FLUTL_OK;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, video,
      stride_bytes, nullptr, 1, 0, n_regions, regions, prims, atlas, row0, row1);
}

extern "C" int flutl_launch_batch (void *stream, uint32_t width, uint32_t height,
    void *dst, size_t dst_stride_bytes, const flutl_frame_desc *frames,
    uint32_t n_frames, const flutl_region *regions, const flutl_prim *prims,
    const uint8_t *atlas)
{
  if (!frames)
    return FLUTL_EINVAL;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, nullptr,
      dst_stride_bytes, frames, n_frames, 0, 0, regions, prims, atlas);
}

extern "C" int flutl_launch_batch_over (void *stream, uint32_t width, uint32_t height,
    void *dst, const void *video, size_t stride_bytes, const flutl_frame_desc *frames,
    uint32_t n_frames, const flutl_region *regions, const flutl_prim *prims,
    const uint8_t *atlas)
{
  if (!frames || !video || video == dst)
    return FLUTL_EINVAL;
  return launch_tiles (static_cast<hipStream_t> (stream), width, height, dst, video,
      stride_bytes, frames, n_frames, 0, 0, regions, prims, atlas);
}

extern "C" int flutl_launch_blur (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t stride_bytes, const int32_t *kernel_fixed,
    int radius)
{
  if (!src || !dst || !kernel_fixed || w == 0 || h == 0 || radius < 0 ||
      radius > kBlurMaxRWide || (stride_bytes & 3) || stride_bytes < (size_t) w * 4)
    return FLUTL_EINVAL;
  // A launch per band of rows: the convolution is not separable in the reference's fixed
  // point, (2r + 1)^2 taps per pixel it is -- a 4K surface under a 300 pixel radius is 3e12 of
  // them --, and no single kernel should run for seconds.  Bands of at most 2e10 taps
  // (FLUTL_BLUR_BAND_TAPS: the tests' way to have many bands on a small surface), queued one
  // behind the other.
  const int size = 2 * radius + 1;
  long long band_taps = 20000000000ll;
  if (const char *e = getenv ("FLUTL_BLUR_BAND_TAPS"))
    if (atoll (e) > 0)
      band_taps = atoll (e);
  const long long row_taps = (long long) w * size * size;
  if (radius > kBlurMaxR) {
    long long rows = band_taps / row_taps;
    rows = rows > (long long) h ? (long long) h : rows;
    rows = rows < 4 ? 4 : (rows + 3) & ~3ll;       // (whole workgroups of 4 rows)
    for (uint32_t y0 = 0; y0 < h; y0 += (uint32_t) rows) {
      const uint32_t n = h - y0 < (uint32_t) rows ? h - y0 : (uint32_t) rows;
      hipLaunchKernelGGL (flutl_blur_wide_kernel_dev, dim3 ((w + 63) / 64, (n + 3) / 4),
          dim3 (256), 0, static_cast<hipStream_t> (stream),
          static_cast<const uint32_t *> (src), static_cast<uint32_t *> (dst), (int) w,
          (int) h, (int) (stride_bytes / 4), kernel_fixed, radius, (int) y0);
      if (hipGetLastError () != hipSuccess)
        return FLUTL_ENODEV;
    }
    return FLUTL_OK;
  }
  const size_t lds = ((size_t) (kBlurTW + 2 * radius) * (kBlurTH + 2 * radius) +
      (size_t) size * size) * 4;
  const uint32_t tiles_x = (w + kBlurTW - 1) / kBlurTW, tiles_y = (h + kBlurTH - 1) / kBlurTH;
  long long band = band_taps / (row_taps * kBlurTH);
  band = band < 1 ? 1 : (band > 65535 ? 65535 : band);
  for (uint32_t t0 = 0; t0 < tiles_y; t0 += (uint32_t) band) {
    const uint32_t n = tiles_y - t0 < (uint32_t) band ? tiles_y - t0 : (uint32_t) band;
    hipLaunchKernelGGL (flutl_blur_kernel_dev, dim3 (tiles_x, n), dim3 (256), lds,
        static_cast<hipStream_t> (stream), static_cast<const uint32_t *> (src),
        static_cast<uint32_t *> (dst), (int) w, (int) h, (int) (stride_bytes / 4),
        kernel_fixed, radius, (int) t0);
    if (hipGetLastError () != hipSuccess)
      return FLUTL_ENODEV;
  }
  return FLUTL_OK;
}

extern "C" int flutl_launch_resample (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t src_stride_bytes, size_t dst_stride_bytes,
    uint32_t distx, uint32_t disty, const int32_t *box)
{
  if (!src || !dst || w == 0 || h == 0 || w > 32766 || h > 32766 || distx > 127 ||
      disty > 127 || (src_stride_bytes & 3) || (dst_stride_bytes & 3) ||
      src_stride_bytes < (size_t) w * 4 || dst_stride_bytes < (size_t) (w + 1) * 4)
    return FLUTL_EINVAL;
  ResampleBox b = { 0, 0, 0, 0, 0 };
  if (box) {
    // 24.8 coordinates of at most a frame: the overlap product stays below 2^31
    for (int k = 0; k < 4; k++)
      if (box[k] < -(1 << 24) || box[k] > (1 << 24))
        return FLUTL_EINVAL;
    b = ResampleBox { box[0], box[1], box[2], box[3], 1 };
  }
  dim3 grid ((w + 1 + 63) / 64, (h + 1 + 3) / 4);
  hipLaunchKernelGGL (flutl_resample_kernel_dev, grid, dim3 (256), 0,
      static_cast<hipStream_t> (stream), static_cast<const uint32_t *> (src),
      static_cast<uint32_t *> (dst), (int) w, (int) h, (int) (src_stride_bytes / 4),
      (int) (dst_stride_bytes / 4), distx, disty, b);
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}

extern "C" int flutl_launch_rotate_cw (void *stream, const void *src, void *dst,
    uint32_t w, uint32_t h, size_t src_stride_bytes, size_t dst_stride_bytes,
    const int32_t *box)
{
  if (!src || !dst || w == 0 || h == 0 || w > 32767 || h > 32767 ||
      (src_stride_bytes & 3) || (dst_stride_bytes & 3) ||
      src_stride_bytes < (size_t) w * 4 || dst_stride_bytes < (size_t) h * 4)
    return FLUTL_EINVAL;
  ResampleBox b = { 0, 0, 0, 0, 0 };
  if (box) {
    for (int k = 0; k < 4; k++)
      if (box[k] < -(1 << 24) || box[k] > (1 << 24))
        return FLUTL_EINVAL;
    b = ResampleBox { box[0], box[1], box[2], box[3], 1 };
  }
  hipLaunchKernelGGL (flutl_rotate_cw_kernel_dev, dim3 ((h + 63) / 64, (w + 3) / 4),
      dim3 (256), 0, static_cast<hipStream_t> (stream), static_cast<const uint32_t *> (src),
      static_cast<uint32_t *> (dst), (int) w, (int) h, (int) (src_stride_bytes / 4),
      (int) (dst_stride_bytes / 4), b);
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}

extern "C" int flutl_launch_delta (void *stream, const void *fresh, void *twin, void *host,
    uint32_t width, uint32_t height, uint32_t row0, uint32_t row1,
    uint64_t *changed_segments)
{
  return flutl_launch_delta_window (stream, fresh, twin, host, width, height, row0, row1, 0,
      height, changed_segments);
}

extern "C" int flutl_launch_delta_window (void *stream, const void *fresh, void *twin,
    void *host, uint32_t width, uint32_t height, uint32_t row0, uint32_t row1,
    uint32_t fresh_row0, uint32_t fresh_row1, uint64_t *changed_segments)
{
  if (!fresh || !twin || !host || width == 0 || height == 0 || width > 32767 ||
      height > 32767 || row0 > row1 || row1 > height || fresh_row0 > fresh_row1)
    return FLUTL_EINVAL;
  if (row0 == row1)
    return FLUTL_OK;
  const uint32_t row_bytes = width * 4u;
  const bool wide = (row_bytes & 15u) == 0u &&
      ((reinterpret_cast<uintptr_t> (fresh) | reinterpret_cast<uintptr_t> (twin) |
        reinterpret_cast<uintptr_t> (host)) & 15u) == 0u;
  const uint32_t chunks = (((row_bytes + 15u) / 16u) + 15u) & ~15u;   // whole segments per row
  const uint64_t threads = (uint64_t) chunks * (row1 - row0);
  const dim3 grid ((uint32_t) ((threads + 255u) / 256u));
  if (wide)
    hipLaunchKernelGGL (flutl_delta_kernel<true>, grid, dim3 (256), 0,
        static_cast<hipStream_t> (stream), static_cast<const uint8_t *> (fresh),
        static_cast<uint8_t *> (twin), static_cast<uint8_t *> (host), row_bytes, chunks,
        row0, row1 - row0, fresh_row0, fresh_row1,
        reinterpret_cast<unsigned long long *> (changed_segments));
  else
    hipLaunchKernelGGL (flutl_delta_kernel<false>, grid, dim3 (256), 0,
        static_cast<hipStream_t> (stream), static_cast<const uint8_t *> (fresh),
        static_cast<uint8_t *> (twin), static_cast<uint8_t *> (host), row_bytes, chunks,
        row0, row1 - row0, fresh_row0, fresh_row1,
        reinterpret_cast<unsigned long long *> (changed_segments));
  return hipGetLastError () == hipSuccess ? FLUTL_OK : FLUTL_ENODEV;
}
