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
// Work decomposition
//   * one workgroup (2 wavefronts = 128 lanes) per 64x8 output tile;
//     lane l owns the 4 horizontally adjacent pixels (16 bytes) at
//     row l/16, column 4*(l%16): a wave stores four 256-byte row segments;
//   * every output pixel is written exactly once, from registers (there is no
//     clear pass and no read of the destination);
//   * tiles are dealt to XCDs in contiguous bands (blocks b and b+8 share an
//     XCD's L2), so a band's glyph masks are fetched into one L2 only;
//   * the grid is persistent (<= 16 workgroups per CU) and grid-strides over
//     tiles: workgroup dispatch rate would otherwise bound a 16k-tile frame;
//   * per tile, regions are culled 64 at a time (one per lane + ballot), a
//     region's primitives are binned by the 128 lanes in parallel (ballot +
//     prefix popcount -> ordered, compacted 32-byte prims in LDS), then
//     replayed in paint order with the next primitive's mask bytes already
//     in flight.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "flutl.h"

namespace {

typedef unsigned short u16x2 __attribute__ ((ext_vector_type (2)));

constexpr int kThreads = 128;          // 2 wavefronts of 64
constexpr int kChunk = kThreads;       // prims binned per round

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
__device__ __forceinline__ void blend_pixman (Px &d, Px s, uint32_t m)
{
  if (m == 0)
    return;
  u16x2 mm = splat (m);
  Px sm;
  sm.rb = mul_un8 (s.rb, mm);
  sm.ag = mul_un8 (s.ag, mm);
  over (d, sm);
}

// cairo in-place lerp for opaque solid sources (0x7f rounding)
__device__ __forceinline__ void blend_lerp (Px &d, Px s, uint32_t m)
{
  if (m == 0)
    return;
  if (m == 255) {
    d = s;
    return;
  }
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
  uint32_t tiles_x, tiles_y;
  uint32_t n_frames;
  uint32_t region_first, region_count;  // used when frames == nullptr
  uint32_t per_xcd;                     // ceil(total_tiles / 8)
  uint64_t stride;                      // bytes
};

// A primitive held wave-uniformly (SGPRs): read from the LDS bin list and
// broadcast with v_readfirstlane so that all control flow on it is scalar.
struct PrimU {
  uint32_t op, rule, param;
  int cx0, cy0, cx1, cy1;
  uint32_t color;
  int a, b, c, d;
};

__device__ __forceinline__ uint32_t rfl (uint32_t v)
{
  return (uint32_t) __builtin_amdgcn_readfirstlane ((int) v);
}

__device__ __forceinline__ PrimU prim_from_lds (const uint4 *s_prim, uint32_t i)
{
  const uint4 lo = s_prim[2 * i], hi = s_prim[2 * i + 1];
  PrimU p;
  const uint32_t w0 = rfl (lo.x), w1 = rfl (lo.y), w2 = rfl (lo.z);
  p.op = w0 & 0xffu;
  p.rule = (w0 >> 8) & 0xffu;
  p.param = w0 >> 16;
  p.cx0 = (int16_t) (w1 & 0xffffu);
  p.cy0 = (int16_t) (w1 >> 16);
  p.cx1 = (int16_t) (w2 & 0xffffu);
  p.cy1 = (int16_t) (w2 >> 16);
  p.color = rfl (lo.w);
  p.a = (int) rfl (hi.x);
  p.b = (int) rfl (hi.y);
  p.c = (int) rfl (hi.z);
  p.d = (int) rfl (hi.w);
  return p;
}

// What a primitive needs from memory for this lane's 4 pixels.  Issued one
// primitive ahead of its use so the load latency hides under the blend of the
// previous primitive (OVER is order dependent, the loads are not).
struct Fetched {
  uint32_t v[4];
};

__device__ __forceinline__ Fetched prim_fetch (const PrimU &p,
    const uint8_t *__restrict__ atlas, int x, int y)
{
  Fetched f = { { 0u, 0u, 0u, 0u } };
  const bool any = y >= p.cy0 && y < p.cy1 && x < p.cx1 && x + 4 > p.cx0;
  if (p.op == FLUTL_OP_MASK) {
    if (any) {
      // 4 consecutive A8 bytes at an arbitrary byte address.  The validated
      // list keeps this window inside the atlas allocation (16 bytes of slack
      // on both sides); bytes outside the bbox are masked off below.
      const uint8_t *ptr = atlas + (uint32_t) p.a +
          (size_t) (y - p.d) * (uint32_t) p.b + (x - p.c);
      __builtin_memcpy (&f.v[0], ptr, 4);
    }
  } else if (p.op == FLUTL_OP_IMAGE) {
    if (any) {
      const uint8_t *row = atlas + (uint32_t) p.a +
          (size_t) (y - p.d) * (uint32_t) p.b;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (x + k >= p.cx0 && x + k < p.cx1)
          f.v[k] = *reinterpret_cast<const uint32_t *> (
              row + (size_t) (x + k - p.c) * 4);
    }
  }
  return f;
}

__device__ __forceinline__ void blend4 (Px (&t)[4], const uint32_t (&cov)[4],
    Px s, uint32_t rule)
{
  if (rule == FLUTL_RULE_LERP) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      blend_lerp (t[k], s, cov[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++)
      blend_pixman (t[k], s, cov[k]);
  }
}

__device__ __forceinline__ void prim_apply (const PrimU &p, const Fetched &f,
    Px (&d)[4], Px (&g)[4], bool &in_group, int x, int y)
{
  if (p.op == FLUTL_OP_GROUP_BEGIN) {
    in_group = true;
#pragma unroll
    for (int k = 0; k < 4; k++)
      g[k] = unpack (0);
    return;
  }
  const bool rowin = y >= p.cy0 && y < p.cy1;
  if (p.op == FLUTL_OP_GROUP_END) {
    in_group = false;
    const u16x2 al = splat (p.param);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = rowin && x + k >= p.cx0 && x + k < p.cx1;
      if (in && (as_u32 (g[k].rb) | as_u32 (g[k].ag)) != 0) {
        Px s;
        s.rb = mul_un8 (g[k].rb, al);
        s.ag = mul_un8 (g[k].ag, al);
        over (d[k], s);
      }
    }
    return;
  }
  if (p.op == FLUTL_OP_IMAGE) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // pixels outside the bbox were fetched as 0: 0 OVER d == d
      if (f.v[k] != 0) {
        if (in_group)
          over (g[k], unpack (f.v[k]));
        else
          over (d[k], unpack (f.v[k]));
      }
    }
    return;
  }
  // RECT / MASK: 8-bit coverage for the lane's 4 pixels
  uint32_t cov[4];
  if (p.op == FLUTL_OP_RECT) {
    const int oy = rowin ? overlap256 (p.b, p.d, y) : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = x + k >= p.cx0 && x + k < p.cx1;
      const uint32_t area = (uint32_t) (overlap256 (p.a, p.c, x + k) * oy) >> 8;
      cov[k] = in ? min (area, 255u) : 0u;
    }
  } else {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool in = x + k >= p.cx0 && x + k < p.cx1;
      cov[k] = in ? ((f.v[0] >> (8 * k)) & 0xffu) : 0u;
    }
  }
  // wave-level early out: nothing of this prim lands in this wave's pixels
  if (__ballot ((cov[0] | cov[1] | cov[2] | cov[3]) != 0) == 0)
    return;
  const Px s = unpack (p.color);
  if (in_group)
    blend4 (g, cov, s, p.rule);
  else
    blend4 (d, cov, s, p.rule);
}

__global__ __launch_bounds__ (kThreads) void flutl_composite_kernel (
    FrameParams fp, uint8_t *__restrict__ dst,
    const flutl_frame_desc *__restrict__ frames,
    const flutl_region *__restrict__ regions,
    const flutl_prim *__restrict__ prims, const uint8_t *__restrict__ atlas)
{
  __shared__ uint4 s_prim[kChunk * 2];   // binned prims of the current chunk
  __shared__ uint32_t s_cnt[2];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const uint32_t tiles_per_frame = fp.tiles_x * fp.tiles_y;
  const uint32_t total = tiles_per_frame * fp.n_frames;

  // Persistent grid-stride loop.  Blocks b and b+8 share an XCD (and its L2):
  // XCD k walks the k-th contiguous band of tiles, so the glyph masks and
  // display-list entries of a band are fetched into one L2 only.
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t jstride = gridDim.x >> 3;
  for (uint32_t j = blockIdx.x >> 3; j < fp.per_xcd; j += jstride) {
    const uint32_t logical = xcd * fp.per_xcd + j;
    if (logical >= total)
      break;
    const uint32_t frame = logical / tiles_per_frame;
    const uint32_t tile = logical - frame * tiles_per_frame;
    const uint32_t ty = tile / fp.tiles_x;
    const uint32_t tx = tile - ty * fp.tiles_x;

    uint32_t rfirst = fp.region_first, rcount = fp.region_count;
    uint64_t dst_off = 0;
    if (frames) {
      rfirst = frames[frame].region_first;
      rcount = frames[frame].region_count;
      dst_off = frames[frame].dst_offset;
    }

    const int tx0 = (int) tx * FLUTL_TILE_W, ty0 = (int) ty * FLUTL_TILE_H;
    const int x = tx0 + (tid & 15) * 4;
    const int y = ty0 + (tid >> 4);

    Px d[4], g[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      d[k] = unpack (0);
      g[k] = unpack (0);
    }

    // ---- cull regions: 64 at a time, one per lane, ballot -----------------
    for (uint32_t rbase = 0; rbase < rcount; rbase += 64) {
      uint4 rv = make_uint4 (0, 0, 0, 0);
      bool rhit = false;
      if (rbase + lane < rcount) {
        rv = *reinterpret_cast<const uint4 *> (&regions[rfirst + rbase + lane]);
        const int rx0 = (int16_t) (rv.x & 0xffffu), ry0 = (int16_t) (rv.x >> 16);
        const int rx1 = (int16_t) (rv.y & 0xffffu), ry1 = (int16_t) (rv.y >> 16);
        rhit = rx0 < tx0 + FLUTL_TILE_W && rx1 > tx0 &&
            ry0 < ty0 + FLUTL_TILE_H && ry1 > ty0 && rv.w != 0;
      }
      unsigned long long rmask = __ballot (rhit);
      while (rmask) {
        const int rl = __builtin_ctzll (rmask);
        rmask &= rmask - 1;
        const uint32_t pfirst = (uint32_t) __builtin_amdgcn_readlane ((int) rv.z, rl);
        const uint32_t pcount = (uint32_t) __builtin_amdgcn_readlane ((int) rv.w, rl);
        bool in_group = false;

        for (uint32_t base = 0; base < pcount; base += kChunk) {
          // ---- bin: lane t tests prim base+t, hits are compacted (in paint
          // order) into LDS as whole 32-byte primitives --------------------
          const uint32_t pi = base + (uint32_t) tid;
          bool hit = false;
          uint4 lo = make_uint4 (0, 0, 0, 0), hi = lo;
          if (pi < pcount) {
            const uint4 *pp = reinterpret_cast<const uint4 *> (&prims[pfirst + pi]);
            lo = pp[0];
            hi = pp[1];
            const int cx0 = (int16_t) (lo.y & 0xffffu), cy0 = (int16_t) (lo.y >> 16);
            const int cx1 = (int16_t) (lo.z & 0xffffu), cy1 = (int16_t) (lo.z >> 16);
            hit = cx0 < tx0 + FLUTL_TILE_W && cx1 > tx0 &&
                cy0 < ty0 + FLUTL_TILE_H && cy1 > ty0;
          }
          const unsigned long long bal = __ballot (hit);
          if (lane == 0)
            s_cnt[wave] = (uint32_t) __popcll (bal);
          __syncthreads ();
          const uint32_t pos = (wave ? s_cnt[0] : 0u) +
              (uint32_t) __popcll (bal & ((1ull << lane) - 1ull));
          const uint32_t n_hit = s_cnt[0] + s_cnt[1];
          if (hit) {
            s_prim[2 * pos] = lo;
            s_prim[2 * pos + 1] = hi;
          }
          __syncthreads ();

          // ---- replay in paint order, fetching one primitive ahead -------
          if (n_hit) {
            PrimU cur = prim_from_lds (s_prim, 0);
            Fetched fcur = prim_fetch (cur, atlas, x, y);
            for (uint32_t i = 0; i < n_hit; i++) {
              PrimU nxt = cur;
              Fetched fnxt = fcur;
              if (i + 1 < n_hit) {
                nxt = prim_from_lds (s_prim, i + 1);
                fnxt = prim_fetch (nxt, atlas, x, y);
              }
              prim_apply (cur, fcur, d, g, in_group, x, y);
              cur = nxt;
              fcur = fnxt;
            }
          }
          __syncthreads ();  // s_prim / s_cnt are reused
        }
      }
    }

    // ---- store: every pixel exactly once ----------------------------------
    if (y < (int) fp.height && x < (int) fp.width) {
      uint8_t *out = dst + dst_off + (size_t) y * fp.stride + (size_t) x * 4;
      const uint4 v = make_uint4 (pack (d[0]), pack (d[1]), pack (d[2]), pack (d[3]));
      if (x + 4 <= (int) fp.width && ((reinterpret_cast<uintptr_t> (out) & 15) == 0)) {
        *reinterpret_cast<uint4 *> (out) = v;
      } else {
        const uint32_t vv[4] = { v.x, v.y, v.z, v.w };
        for (int k = 0; k < 4 && x + k < (int) fp.width; k++)
          reinterpret_cast<uint32_t *> (out)[k] = vv[k];
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

// Workgroups per CU of the persistent grid (16 = the 32-wave CU limit for
// 2-wave workgroups).  FLUTL_WG_PER_CU overrides it for tuning runs.
uint32_t wg_per_cu ()
{
  static const uint32_t v = [] {
    const char *e = getenv ("FLUTL_WG_PER_CU");
    const long n = e ? strtol (e, nullptr, 10) : 0;
    return (uint32_t) (n >= 1 && n <= 64 ? n : 16);
  }();
  return v;
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
  fp.tiles_x = (width + FLUTL_TILE_W - 1) / FLUTL_TILE_W;
  fp.tiles_y = (height + FLUTL_TILE_H - 1) / FLUTL_TILE_H;
  fp.n_frames = n_frames;
  fp.region_first = region_first;
  fp.region_count = region_count;
  const uint64_t total = (uint64_t) fp.tiles_x * fp.tiles_y * n_frames;
  if (total > 0x7fffffffull)
    return FLUTL_EINVAL;
  fp.per_xcd = (uint32_t) ((total + 7) / 8);
  fp.stride = stride;
  // persistent grid: at most kWgPerCu workgroups per CU, a multiple of 8 (XCDs)
  uint32_t grid = fp.per_xcd * 8u;
  const uint32_t cap = 256u * wg_per_cu ();
  if (grid > cap)
    grid = cap;
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
